// Per-thread building blocks of the bootstrap kernels (gfx950), written so the same code also
// compiles for the host: the host build runs the identical index maps and arithmetic thread by
// thread, which is how the LDS layouts are checked without a GPU.
//
// Arithmetic.  The negacyclic transform runs over two primes p0, p1 just under 5*2^47 (both
// 1 mod 8192, p0*p1 > 2^98.6).  Residues are exact integers carried in f64 registers and kept
// lazily reduced (|x| < 13 p < 2^53).  A modular product is a Barrett reduction whose quotient
// estimate comes from the FP64 pipe and whose remainder is recovered exactly with two FMAs:
//     h = b*w; l = fma(b,w,-h); q = rint(h/p); r = fma(-q,p,h) + l        (b*w = h + l exactly)
// On gfx950 v_fma_f64 / v_mul_f64 / v_add_f64 issue at the same (half) rate as v_mul_lo_u32 and
// v_mad_u64_u32, so this 6-instruction product is several times cheaper than a 64-bit Montgomery
// product (about 11 multiplier instructions plus carries); see csrc/microbench/valu_rates.hip.
//
// Work split for one polynomial pair (N = 2048, 256 threads, 8 coefficients per thread and
// polynomial): four register passes (radix 8, 8, 8, 4) with an LDS exchange between passes.
//   pass 1 owns e = t + 256 j           (strides 1024, 512, 256)
//   pass 2 owns e = 256 (t>>5) + (t&31) + 32 j   (strides 128, 64, 32)
//   pass 3 owns e = 32 (t>>2) + (t&3) + 4 j      (strides 16, 8, 4)
//   pass 4 owns e = 8 t + j             (strides 2, 1)
// The forward transform is Cooley-Tukey with the twist merged (natural in, bit-reversed out);
// the inverse is Gentleman-Sande through the same passes backwards.  Each exchange has its own
// padding so that both its write and its read are bank-conflict free for 8-byte LDS accesses.
//
// Which exchanges cross wavefronts.  With t = 64 wave + lane, passes 2, 3 and 4 all own
// coefficients whose two top index bits equal the wave number: wave w works on e in
// [512 w, 512 w + 512), and every padding maps that range into the wave's own LDS region
// [576 w, 576 w + 576).  Only the exchange between passes 1 and 2 moves data between waves and
// needs a workgroup barrier on either side; the other two are private to a wave, whose LDS
// operations execute in program order, so a compiler-level fence is all they need (sync_after).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define LFHE_HD __host__ __device__ __forceinline__
#else
#define LFHE_HD inline
#endif

#pragma clang fp contract(off)

namespace lfhe {

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;

constexpr int N = 2048;
constexpr int LOGN = 11;
constexpr int NT = 256;          // threads per ciphertext
constexpr int E = 8;             // coefficients per thread and polynomial
constexpr int POLY_STRIDE = 2304;  // padded polynomial in LDS, in 8-byte words
constexpr int LDS_WORDS = 2 * POLY_STRIDE;

constexpr double P0 = 703687441727489.0;
constexpr double P1 = 703687441702913.0;
constexpr u64 P0_U = 703687441727489ull;
constexpr u64 P1_U = 703687441702913ull;

struct Prime {
    double p, pinv;
};

// b * w mod p, b any integer with |b| < 13 p, |w| <= p/2.  Result within (0.5 + 0.08 |b|/p) p.
LFHE_HD double mmul(double b, double w, const Prime& P) {
    double h = b * w;
    double l = __builtin_fma(b, w, -h);
    double q = __builtin_rint(h * P.pinv);
    double c = __builtin_fma(-q, P.p, h);
    return c + l;
}

// centred representative of x, |x| < 2^53
LFHE_HD double mnorm(double x, const Prime& P) {
    return __builtin_fma(-__builtin_rint(x * P.pinv), P.p, x);
}

// Host-only instrumentation (the phase emulation defines LFHE_TRACK_MAX): largest magnitude any
// butterfly or product operand reaches, to check the lazy-reduction bound |x| < 2^53.
#if defined(LFHE_TRACK_MAX) && !defined(__HIP_DEVICE_COMPILE__)
inline double& track_max_ref() { static double m = 0; return m; }
inline void track_max(double v) { double a = v < 0 ? -v : v; if (a > track_max_ref()) track_max_ref() = a; }
#else
LFHE_HD void track_max(double) {}
#endif

// Table loads take a 32-bit unsigned byte offset from a wave-uniform base, which lets the compiler
// use the scalar-base + 32-bit-offset addressing mode instead of 64-bit address arithmetic per load.
LFHE_HD double ldt(const double* __restrict__ base, unsigned idx) {
#ifdef LFHE_ABLATE_NO_TABLE   // timing experiment only: results are wrong
    (void)base;
    return (double)idx;
#else
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + idx * 8u);
#endif
}

// twiddle load
LFHE_HD double ldtw(const double* __restrict__ base, unsigned idx) {
#ifdef LFHE_ABLATE_NO_TW   // timing experiment only: results are wrong
    (void)base;
    return (double)idx;
#else
    return ldt(base, idx);
#endif
}

LFHE_HD void ct_bfly(double& a, double& b, double w, const Prime& P) {
    track_max(b);
    double t = mmul(b, w, P);
    b = a - t;
    a = a + t;
    track_max(a); track_max(b);
}

LFHE_HD void gs_bfly(double& a, double& b, double w, const Prime& P) {
    double u = a, v = b;
    a = u + v;
    track_max(a); track_max(u - v);
    b = mmul(u - v, w, P);
}

// ---- element ownership per pass
LFHE_HD unsigned own1(unsigned t, unsigned j) { return t + 256 * j; }
LFHE_HD unsigned own2(unsigned t, unsigned j) { return ((t >> 5) << 8) + (t & 31) + 32 * j; }
LFHE_HD unsigned own3(unsigned t, unsigned j) { return ((t >> 2) << 5) + (t & 3) + 4 * j; }
LFHE_HD unsigned own4(unsigned t, unsigned j) { return 8 * t + j; }
// ---- paddings per exchange
LFHE_HD unsigned pad12(unsigned e) { return e + 64 * (e >> 9); }
LFHE_HD unsigned pad23(unsigned e) { return e + 4 * (e >> 5); }
LFHE_HD unsigned pad34(unsigned e) { return e + (e >> 3); }

// Twiddles of one pass, loaded ahead of the pass (the loads are issued a phase early so that
// their latency is covered by the barrier and the LDS exchange in between).
//   radix-8 pass with first-stage table index b: w[0] = tw[b], w[1..2] = tw[2b..2b+1], w[3..6] = tw[4b..4b+3]
//   last pass (strides 2 and 1):                 w[0..1] = tw[512+2t..], w[2..5] = tw[1024+4t..]
LFHE_HD void load_tw8(const double* __restrict__ tw, unsigned b, double (&w)[7]) {
    w[0] = ldtw(tw, b);
    w[1] = ldtw(tw, 2 * b); w[2] = ldtw(tw, 2 * b + 1);
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) w[3 + k] = ldtw(tw, 4 * b + k);
}
LFHE_HD void load_tw4x2(const double* __restrict__ tw, unsigned t, double (&w)[7]) {
    w[0] = ldtw(tw, 512 + 2 * t); w[1] = ldtw(tw, 512 + 2 * t + 1);
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) w[2 + k] = ldtw(tw, 1024 + 4 * t + k);
}

// Radix-8 Cooley-Tukey on the 8 registers of one pass (tw is the bit-reversed table,
// tw[m + i] = psi^bitrev(m + i)).
LFHE_HD void ct_radix8(double (&x)[8], const double (&w)[7], const Prime& P) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ct_bfly(x[j], x[j + 4], w[0], P);
    ct_bfly(x[0], x[2], w[1], P); ct_bfly(x[1], x[3], w[1], P);
    ct_bfly(x[4], x[6], w[2], P); ct_bfly(x[5], x[7], w[2], P);
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) ct_bfly(x[2 * k], x[2 * k + 1], w[3 + k], P);
}
LFHE_HD void ct_radix8(double (&x)[8], const double* __restrict__ tw, unsigned b, const Prime& P) {
    double w[7];
    load_tw8(tw, b, w);
    ct_radix8(x, w, P);
}

// last forward pass: strides 2 and 1 on 8 consecutive coefficients
LFHE_HD void ct_radix4x2(double (&x)[8], const double (&w)[7], const Prime& P) {
    ct_bfly(x[0], x[2], w[0], P); ct_bfly(x[1], x[3], w[0], P);
    ct_bfly(x[4], x[6], w[1], P); ct_bfly(x[5], x[7], w[1], P);
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) ct_bfly(x[2 * k], x[2 * k + 1], w[2 + k], P);
}
LFHE_HD void ct_radix4x2(double (&x)[8], const double* __restrict__ tw, unsigned t, const Prime& P) {
    double w[7];
    load_tw4x2(tw, t, w);
    ct_radix4x2(x, w, P);
}
LFHE_HD void ct_radix4x2_bounded(double (&x)[8], const double (&w)[7], const Prime& P) {
    x[0] = mnorm(x[0], P);
    x[4] = mnorm(x[4], P);
    ct_radix4x2(x, w, P);
}

// Last forward pass of a transform whose outputs meet a wide operand in the point-wise stage (the
// grouped blind rotation: the step's GGSW entry, |B| < 2.2 p): registers 0 and 4 are the only values
// that reach the outputs of the pass without going through a product, so normalising these two
// first leaves every output below 3.4 p (two products of operands below 10.2 p on top of 0.5 p) --
// two normalisations per polynomial instead of eight on the outputs.  |B d| < 7.5 p^2 keeps the
// quotient estimate of the product below 2^53 and the point-wise sum of two products below 4.5 p,
// which the inverse transform's schedule takes (its first stage accepts up to 6.3 p).
LFHE_HD void ct_radix4x2_bounded(double (&x)[8], const double (&w)[7], const Prime& P);

LFHE_HD void norm8(double (&x)[8], const Prime& P) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = mnorm(x[j], P);
}

// normalise the registers whose bit is set in MASK
template <unsigned MASK>
LFHE_HD void norm_sel(double (&x)[8], const Prime& P) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (MASK & (1u << j)) x[j] = mnorm(x[j], P);
}

// Where the inverse transform normalises.  A Gentleman-Sande stage turns (a, b) into (a + b,
// product): the sums grow, the products come back small, so after a few stages only the registers
// that collected sums are large, and which ones they are depends on the register index only.
// The masks below (register bits, one per stage of each pass) are the cheapest schedule that keeps
// every value below 2^53 = 12.79 p in the worst case -- fourteen normalisations per polynomial
// instead of three times eight -- and leaves the result below 2 p.  tools/lazy_bounds.py derives
// the bounds (worst 11.75 p) and finds the schedule; tests/test_host_logic.py ties it to these
// constants.
constexpr unsigned GS_NORM_PASS4[2] = {0x55, 0x22};
constexpr unsigned GS_NORM_PASS3[3] = {0x00, 0x00, 0x01};
constexpr unsigned GS_NORM_PASS2[3] = {0x00, 0x11, 0x0E};
constexpr unsigned GS_NORM_PASS1[3] = {0x00, 0x00, 0x03};
// The grouped blind rotation canonicalises (or lifts) every output of the inverse transform anyway
// and only needs it below 3 p: one normalisation fewer in pass 2, result below 2.5 p (13 per polynomial).
constexpr unsigned GS_NORM_PASS2_GROUPED[3] = {0x00, 0x11, 0x06};

template <unsigned M1, unsigned M2>
LFHE_HD void gs_radix4x2(double (&x)[8], const double (&w)[7], const Prime& P) {
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) gs_bfly(x[2 * k], x[2 * k + 1], w[2 + k], P);
    norm_sel<M1>(x, P);
    gs_bfly(x[0], x[2], w[0], P); gs_bfly(x[1], x[3], w[0], P);
    gs_bfly(x[4], x[6], w[1], P); gs_bfly(x[5], x[7], w[1], P);
    norm_sel<M2>(x, P);
}
template <unsigned M1, unsigned M2>
LFHE_HD void gs_radix4x2(double (&x)[8], const double* __restrict__ tw, unsigned t, const Prime& P) {
    double w[7];
    load_tw4x2(tw, t, w);
    gs_radix4x2<M1, M2>(x, w, P);
}

// M1..M3: registers normalised after the first, second and third stage of the pass
template <unsigned M1, unsigned M2, unsigned M3>
LFHE_HD void gs_radix8(double (&x)[8], const double (&w)[7], const Prime& P) {
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) gs_bfly(x[2 * k], x[2 * k + 1], w[3 + k], P);
    norm_sel<M1>(x, P);
    gs_bfly(x[0], x[2], w[1], P); gs_bfly(x[1], x[3], w[1], P);
    gs_bfly(x[4], x[6], w[2], P); gs_bfly(x[5], x[7], w[2], P);
    norm_sel<M2>(x, P);
#pragma unroll
    for (int j = 0; j < 4; ++j) gs_bfly(x[j], x[j + 4], w[0], P);
    norm_sel<M3>(x, P);
}
template <unsigned M1, unsigned M2, unsigned M3>
LFHE_HD void gs_radix8(double (&x)[8], const double* __restrict__ tw, unsigned b, const Prime& P) {
    double w[7];
    load_tw8(tw, b, w);
    gs_radix8<M1, M2, M3>(x, w, P);
}

// table index of the first stage of each radix-8 pass
LFHE_HD unsigned base2(unsigned t) { return 8 + (t >> 5); }
LFHE_HD unsigned base3(unsigned t) { return 64 + (t >> 2); }

// ---- LDS exchange halves.  `buf` is one polynomial's region (POLY_STRIDE words).
#ifdef LFHE_ABLATE_NO_LDS   // timing experiment only: results are wrong
#define LFHE_XCHG(NAME, OWN, PAD)                                                         \
    LFHE_HD void st_##NAME(double*, int, const double (&)[8]) {}                          \
    LFHE_HD void ld_##NAME(const double*, int, double (&)[8]) {}
#else
#define LFHE_XCHG(NAME, OWN, PAD)                                                         \
    LFHE_HD void st_##NAME(double* buf, unsigned t, const double (&x)[8]) {               \
        _Pragma("unroll") for (unsigned j = 0; j < 8; ++j) buf[PAD(OWN(t, j))] = x[j];    \
    }                                                                                     \
    LFHE_HD void ld_##NAME(const double* buf, unsigned t, double (&x)[8]) {               \
        _Pragma("unroll") for (unsigned j = 0; j < 8; ++j) x[j] = buf[PAD(OWN(t, j))];    \
    }
#endif
LFHE_XCHG(o1p12, own1, pad12)
LFHE_XCHG(o2p12, own2, pad12)
LFHE_XCHG(o2p23, own2, pad23)
LFHE_XCHG(o3p23, own3, pad23)
LFHE_XCHG(o3p34, own3, pad34)
LFHE_XCHG(o4p34, own4, pad34)
#undef LFHE_XCHG

// ---- conversions
// exact integer |v| < 2^51 held in f64 -> two's complement 64-bit
LFHE_HD i64 f64_to_i64(double v) {
    const double magic = 6755399441055744.0;  // 2^52 + 2^51
    double s = v + magic;
    i64 bits;
    __builtin_memcpy(&bits, &s, 8);
    return bits - 0x4338000000000000ll;
}

// signed digit of the one-level base-2^23 decomposition (closest representative; the tie keeps
// +2^22, matching tfhe-rs SignedDecomposer)
LFHE_HD int decomp23(u64 v) {
    u32 hi = (u32)((v + (1ull << 40)) >> 32);
    u32 u = hi >> 9;
    return (int)u - (int)((u > (1u << 22)) ? (1u << 23) : 0u);
}

// torus value (centred) -> residue mod p as an exact f64, |result| <= p/2 (+1 ulp of slack)
LFHE_HD double torus_to_res(u64 v, double two32_mod_p, const Prime& P) {
    i64 s = (i64)v;
    double hi = (double)(int)(s >> 32);          // signed high word
    double lo = (double)(u32)(u64)s;             // unsigned low word
    return mnorm(mmul(hi, two32_mod_p, P) + lo, P);
}

// CRT lift of (y0 mod p0, y1 mod p1), both centred, to the integer in (-P/2, P/2), wrapped to 2^64
LFHE_HD u64 crt_wrap(double y0, double y1, double p0inv_mod_p1, const Prime& Q1) {
    double k = mnorm(mmul(y1 - y0, p0inv_mod_p1, Q1), Q1);
    return (u64)f64_to_i64(y0) + P0_U * (u64)f64_to_i64(k);
}

// The same lift with a look at where it landed.  The lift tells integers apart inside (-P/2, P/2), P =
// p0 p1; the quotient k = (value - y0) / p0 is centred in (-p1/2, p1/2), so a value near the end of
// the window has |k| near p1/2.  A step of the grouped key over 2^64 stays 149 standard deviations
// inside the window; should one ever leave it, it leaves by far less than the guard band |k| >
// k_guard (the tail beyond the window falls off on a scale 2^14 times smaller than the band), so it
// lands IN the band and is seen.  Nothing is ever flagged by chance either: the band starts 130
// standard deviations out.
LFHE_HD u64 crt_wrap_checked(double y0, double y1, double p0inv_mod_p1, const Prime& Q1, double k_guard, bool& near_edge) {
    double k = mnorm(mmul(y1 - y0, p0inv_mod_p1, Q1), Q1);
    near_edge = near_edge || __builtin_fabs(k) > k_guard;
    return (u64)f64_to_i64(y0) + P0_U * (u64)f64_to_i64(k);
}

}  // namespace lfhe
