// The external product  out_c = sum_j dig_j (*) BSK_i[j][c]  (mod X^N+1, mod 2^64)  for one
// ciphertext, cut into barrier-delimited phases.  Every phase is a per-thread function; the GPU
// kernels run phase, __syncthreads(), phase, ...; the host build runs each phase for all 256
// threads in turn.  Same code, same LDS image.
#pragma once
#include <type_traits>

#include "pbs_device.h"

namespace lfhe {

// Everything a kernel needs besides its operands; passed by value as a kernel argument.
struct Tables {
    const double* tw_fwd[2];   // [N] psi^bitrev(i), centred, per prime
    const double* tw_inv[2];   // [N] psi^-bitrev(i), centred
    Prime pr[2];
    double two32[2];           // 2^32 mod p, centred
    double ninv[2];            // N^-1 mod p, centred
    double p0inv_mod_p1;       // centred
    const double* mono;        // [2N] psi^k over prime 0, centred, then [8] psi^(N k / 4) (grouped blind rotation)
    const double* mono1;       // the same over prime 1 (grouped key over 2^64)
};

// Bootstrapping key on the device: f64 residues, N^-1 folded in, transform (bit-reversed) order,
// laid out so that each 16-byte load of the point-wise stage is contiguous across a wavefront:
//   [i][prime][q = j_in*2 + c_out][chunk c4][thread t][2]   holds position e = 8 t + 2 c4 + h.
constexpr int BSK_POLY = N;                 // doubles per (i, prime, q)
constexpr int BSK_ROW = 2 * 4 * BSK_POLY;   // doubles per i
LFHE_HD unsigned bsk_off(unsigned r, unsigned q, unsigned c4, unsigned t) { return ((r * 4 + q) * 4 + c4) * (NT * 2) + t * 2; }

struct EpState {
    double tw[7];      // twiddles of the next pass, loaded one phase ahead
    int dig[2][E];     // in: decomposition digits, pass-1 ownership
    double x[2][E];    // working residues
    double y0[2][E];   // prime-0 result, pass-1 ownership
    u64 out[2][E];     // result, pass-1 ownership: added to (ACCUM) or overwritten
};

constexpr int EP_PHASES = 24;  // 12 per prime
constexpr int EPQ_PHASES = 12; // single ring modulus

// What has to separate phase PH from the next one.  Phases 0 and 10 publish the exchange between
// passes 1 and 2, which crosses wavefronts, and phase 11 reads it: workgroup barriers.  Phase 5 reads
// exactly the words phase 6 rewrites, thread by thread: nothing.  Every other hand-over stays inside
// one wavefront's LDS region (pbs_device.h): program order within the wave is enough.
enum class Sync { None, Wave, Block };
constexpr Sync sync_after(int PH) {
    const int ph = PH % 12;
    return (ph == 0 || ph >= 10) ? Sync::Block : ph == 5 ? Sync::None : Sync::Wave;
}

// The same product over the single ring modulus q' = p0 (bootstrapping key and accumulator in
// Z_q'): phases 0-11 only, digits handed over as doubles in s.x, result (|y| <= p0, any
// representative) left in s.x; key row layout [q = j_in*2 + c_out][c4][t][2].
constexpr int BSKQ_ROW = 4 * BSK_POLY;      // doubles per i
constexpr u64 QP_U = P0_U;

// phase p of the external product.  `lds` holds two padded polynomials.
template <int PH, bool ACCUM = true, bool SINGLE = false>
LFHE_HD void ep_phase(EpState& s, double* lds, unsigned t, const Tables& T, const double* __restrict__ bsk_i) {
    static_assert(!SINGLE || PH < 12, "the single-modulus product has twelve phases");
    constexpr int r = PH / 12;
    constexpr int ph = PH % 12;
    const Prime P = T.pr[r];
#if defined(__HIP_DEVICE_COMPILE__)
    // Everything a thread addresses (twiddles, key row, LDS slots) depends on its index only, not
    // on the blind-rotation step, so the compiler would hoist the loads or at least their 64-bit
    // addresses out of the step loop and keep hundreds of registers live across it.  Making the
    // thread index and a zero table offset opaque per phase turns them back into a few integer
    // instructions and per-step loads served by L1/L2 (the pointers themselves stay kernel
    // arguments, so the loads remain global, not flat).
    unsigned zero = 0;
    asm volatile("" : "+v"(t));
    asm volatile("" : "+s"(zero));
    const double* __restrict__ twf = T.tw_fwd[r] + zero;
    const double* __restrict__ twi = T.tw_inv[r] + zero;
#else
    const double* __restrict__ twf = T.tw_fwd[r];
    const double* __restrict__ twi = T.tw_inv[r];
#endif
    double* b0 = lds;
    double* b1 = lds + POLY_STRIDE;

#ifdef LFHE_ABLATE_NO_BSK   // timing experiment only: results are wrong
#define LFHE_LDK(b, o) ((double)(o))
#else
#define LFHE_LDK(b, o) ldt(b, o)
#endif
    if constexpr (ph == 0) {  // digits -> residues, forward pass 1, publish
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if constexpr (!SINGLE) {
#pragma unroll
                for (int j = 0; j < E; ++j) s.x[q][j] = (double)s.dig[q][j];
            }
            ct_radix8(s.x[q], twf, 1, P);
        }
        st_o1p12(b0, t, s.x[0]); st_o1p12(b1, t, s.x[1]);
        load_tw8(twf, base2(t), s.tw);
    } else if constexpr (ph == 1) {
        ld_o2p12(b0, t, s.x[0]); ld_o2p12(b1, t, s.x[1]);
        ct_radix8(s.x[0], s.tw, P); ct_radix8(s.x[1], s.tw, P);
    } else if constexpr (ph == 2) {
        st_o2p23(b0, t, s.x[0]); st_o2p23(b1, t, s.x[1]);
        load_tw8(twf, base3(t), s.tw);
    } else if constexpr (ph == 3) {
        ld_o3p23(b0, t, s.x[0]); ld_o3p23(b1, t, s.x[1]);
        ct_radix8(s.x[0], s.tw, P); ct_radix8(s.x[1], s.tw, P);
    } else if constexpr (ph == 4) {
        st_o3p34(b0, t, s.x[0]); st_o3p34(b1, t, s.x[1]);
        load_tw4x2(twf, t, s.tw);
    } else if constexpr (ph == 5) {  // forward pass 4, point-wise with the key row, inverse pass 4
        ld_o4p34(b0, t, s.x[0]); ld_o4p34(b1, t, s.x[1]);
        ct_radix4x2(s.x[0], s.tw, P); ct_radix4x2(s.x[1], s.tw, P);
        load_tw4x2(twi, t, s.tw);   // for the inverse pass below: in flight during the point-wise stage
        // key row, two coefficients (one 16-byte load per polynomial) at a time, the next chunk's
        // loads in flight while this one is multiplied; the fences keep the compiler from hoisting
        // all sixteen loads (64 registers) to the top of the phase
        double kc[4][2], kn[4][2];
#pragma unroll
        for (unsigned q = 0; q < 4; ++q) { const unsigned o = bsk_off(r, q, 0, t); kc[q][0] = LFHE_LDK(bsk_i, o); kc[q][1] = LFHE_LDK(bsk_i, o + 1); }
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            if (c4 < 3) {
#pragma unroll
                for (unsigned q = 0; q < 4; ++q) { const unsigned o = bsk_off(r, q, c4 + 1, t); kn[q][0] = LFHE_LDK(bsk_i, o); kn[q][1] = LFHE_LDK(bsk_i, o + 1); }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = 2 * c4 + h;
                const double d0 = s.x[0][j], d1 = s.x[1][j];
                s.x[0][j] = mmul(d0, kc[0][h], P) + mmul(d1, kc[2][h], P);
                s.x[1][j] = mmul(d0, kc[1][h], P) + mmul(d1, kc[3][h], P);
            }
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::: "memory");
#endif
#pragma unroll
            for (int q = 0; q < 4; ++q) { kc[q][0] = kn[q][0]; kc[q][1] = kn[q][1]; }
        }
        gs_radix4x2<GS_NORM_PASS4[0], GS_NORM_PASS4[1]>(s.x[0], s.tw, P);
        gs_radix4x2<GS_NORM_PASS4[0], GS_NORM_PASS4[1]>(s.x[1], s.tw, P);
    } else if constexpr (ph == 6) {
        st_o4p34(b0, t, s.x[0]); st_o4p34(b1, t, s.x[1]);
        load_tw8(twi, base3(t), s.tw);
    } else if constexpr (ph == 7) {
        ld_o3p34(b0, t, s.x[0]); ld_o3p34(b1, t, s.x[1]);
        gs_radix8<GS_NORM_PASS3[0], GS_NORM_PASS3[1], GS_NORM_PASS3[2]>(s.x[0], s.tw, P);     // stages 3-5
        gs_radix8<GS_NORM_PASS3[0], GS_NORM_PASS3[1], GS_NORM_PASS3[2]>(s.x[1], s.tw, P);
    } else if constexpr (ph == 8) {
        st_o3p23(b0, t, s.x[0]); st_o3p23(b1, t, s.x[1]);
        load_tw8(twi, base2(t), s.tw);
    } else if constexpr (ph == 9) {
        ld_o2p23(b0, t, s.x[0]); ld_o2p23(b1, t, s.x[1]);
        gs_radix8<GS_NORM_PASS2[0], GS_NORM_PASS2[1], GS_NORM_PASS2[2]>(s.x[0], s.tw, P);     // stages 6-8
        gs_radix8<GS_NORM_PASS2[0], GS_NORM_PASS2[1], GS_NORM_PASS2[2]>(s.x[1], s.tw, P);
    } else if constexpr (ph == 10) {
        st_o2p12(b0, t, s.x[0]); st_o2p12(b1, t, s.x[1]);
    } else {  // ph == 11: inverse pass 1, then keep (prime 0) or lift (prime 1)
        static_assert(ph == 11, "phase out of range");
        ld_o1p12(b0, t, s.x[0]); ld_o1p12(b1, t, s.x[1]);
        // stages 9-11: the outputs (|y| < 2 p) go to the CRT lift or the accumulator as they are; the
        // lift only needs residues, not centred ones, because |true value| <= 2^97 < 0.33 p0 p1
        gs_radix8<GS_NORM_PASS1[0], GS_NORM_PASS1[1], GS_NORM_PASS1[2]>(s.x[0], twi, 1, P);
        gs_radix8<GS_NORM_PASS1[0], GS_NORM_PASS1[1], GS_NORM_PASS1[2]>(s.x[1], twi, 1, P);
        if constexpr (SINGLE) {
        } else if constexpr (r == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) s.y0[q][j] = s.x[q][j];
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const u64 v = crt_wrap(s.y0[q][j], s.x[q][j], T.p0inv_mod_p1, T.pr[1]);
                    if constexpr (ACCUM) s.out[q][j] += v; else s.out[q][j] = v;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Blind rotation with two mask elements per step (single ring modulus).  The key holds, per pair of
// key bits (s, s'), GGSWs G0..G3 of (1-s)(1-s'), s(1-s'), (1-s)s', s s' as four consecutive rows of
// the single-modulus layout; the step is  acc <- B [.] acc  with
//     B = G0 + X^a G1 + X^a' G2 + X^(a+a') G3 .
// In the transform domain X^e is the point-wise factor psi^(e (2 bitrev(pos) + 1)) at position pos,
// so B is formed point by point from the four key rows and powers of psi, and the step costs ONE
// external product: the transforms of phases 0-4 and 6-11 above, with ep_group2_phase5 in place of
// phase 5's point-wise stage.
// ------------------------------------------------------------------------------------------------
constexpr int MONO_WORDS = 2 * N;

LFHE_HD unsigned rev8(unsigned t) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(t) >> 24;
#else
    unsigned r = ((t & 0xF0u) >> 4) | ((t & 0x0Fu) << 4);
    r = ((r & 0xCCu) >> 2) | ((r & 0x33u) << 2);
    return ((r & 0xAAu) >> 1) | ((r & 0x55u) << 1);
#endif
}

// The factors psi^(e u) of X^e at the two positions 8 t + 2 c4 + {0, 1}, whose exponents are
// u = u_t + 512 r and u_t + 512 (r + 4) with r = bitrev3(2 c4) in {0, 2, 1, 3}, u_t odd:
// psi^(e u_t) comes from the table of all 2N powers once per thread (`base`); psi^(512 e r) =
// w8^(e r mod 8) is one of the eight 8th roots of unity, the same for every thread of the
// ciphertext (a scalar); and psi^(2048 e) = (-1)^e turns the first position's power into the
// second's.  One modular product per pair of positions, none for c4 = 0.
template <int C4>
LFHE_HD void mono_pair(double base, const double* __restrict__ w8, unsigned e, const Prime& P, double& m0, double& m1) {
    constexpr unsigned R[4] = {0, 2, 1, 3};
    m0 = C4 == 0 ? base : mmul(base, w8[(e * R[C4]) & 7u], P);
    // m1 = (-1)^e m0 as a flip of the sign bit (an exclusive-or with a scalar, where a select on a
    // 64-bit value costs two conditional moves)
    u64 bits;
    __builtin_memcpy(&bits, &m0, 8);
    bits ^= (u64)(e & 1u) << 63;
    __builtin_memcpy(&m1, &bits, 8);
}

// One entry of the step's GGSW, built up as its four key words arrive (row m = 0..3), m1 = psi^(a u)
// and m2 = psi^(a' u) being the monomial factors at this position.
//   rows as generated (K0..K3):   B = (K0 + m1 K1) + m2 (K2 + m1 K3)                 three products
//   FACTORED rows (a, b, d, c):   B = d (m1 + a)(m2 + b) + c                          two products
// with a = K2/K3, b = K1/K3, d = K3, c = K0 - K1 K2 / K3 computed once when the key is loaded
// (k_bsk_factor): the same residue, 15 instead of 21 FP64 instructions per entry.  |B| < 2.2 p either way.
template <bool FACTORED>
LFHE_HD void ggsw_entry_row(int m, double& B, double& S, double k, double m1, double m2, const Prime& P) {
    if constexpr (FACTORED) {
        if (m == 0) B = m1 + k;
        else if (m == 1) B = mmul(B, m2 + k, P);     // both operands below 1.1 p: within the product's range
        else if (m == 2) B = mmul(B, k, P);
        else B += k;
    } else {
        if (m == 0) B = k;
        else if (m == 1) B += mmul(m1, k, P);
        else if (m == 2) S = k;
        else B += mmul(S + mmul(m1, k, P), m2, P);   // |S + ...| < 1.1 p: the wide operand
    }
}

// ONE of the four terms of the step's GGSW -- K0, m1 K1, m2 K2, m1 m2 K3 -- built from the same rows in
// the same order.  Used by the slow path of the grouped key over 2^64 (k_blind_rotate_g64_pair<.., SAFE>),
// which multiplies the accumulator by each term separately: every such product is the product with
// ONE GGSW and lifts exactly without condition.  From factored rows (a, b, d, c) the key words come
// back as K3 = d, K2 = a d, K1 = b d, K0 = c + a b d.
template <bool FACTORED, int TERM>
LFHE_HD void ggsw_term_row(int m, double& B, double k, double m1, double m2, const Prime& P) {
    if constexpr (FACTORED) {
        if constexpr (TERM == 0) {
            if (m == 0) B = k; else if (m == 1 || m == 2) B = mmul(B, k, P); else B += k;
        } else if constexpr (TERM == 1) {
            if (m == 1) B = mmul(m1, k, P); else if (m == 2) B = mmul(B, k, P);
        } else if constexpr (TERM == 2) {
            if (m == 0) B = mmul(m2, k, P); else if (m == 2) B = mmul(B, k, P);
        } else {
            if (m == 0) B = mmul(m1, m2, P); else if (m == 2) B = mmul(B, k, P);
        }
    } else {
        if (m == TERM) B = TERM == 0 ? k : TERM == 1 ? mmul(m1, k, P) : TERM == 2 ? mmul(m2, k, P) : mmul(mmul(m1, m2, P), k, P);
    }
}

// The same for three key bits per step: ONE of the eight terms K0, m1 K1, m2 K2, m1 m2 K3 (TERM 0..3) and
// m3 K4, m3 m1 K5, m3 m2 K6, m3 m1 m2 K7 (TERM 4..7) of the step's GGSW entry, from the factored word pairs
// in their load order w = 0..3 = (a1, b1), (d1, c1), (a0, b0), (d0, c0); the key words come back as
// K3 = d, K2 = a d, K1 = b d, K0 = c + a b d per half.  T holds the partial value between the two pairs of
// the term's half; B is complete after w = 3.  |B| < 1.1 p.
template <int TERM>
LFHE_HD void ggsw3_term_row(int w, double& B, double& T, double k0, double k1, double m1, double m2, double m3,
                            const Prime& P) {
    constexpr int HALF = TERM >> 2, SUB = TERM & 3;
    const int first = HALF ? 0 : 2;
    if (w == first) {           // (a, b)
        T = SUB == 0 ? mmul(k0, k1, P) : SUB == 1 ? k1 : SUB == 2 ? k0 : mmul(m1, m2, P);
    } else if (w == first + 1) {   // (d, c)
        double v = mmul(T, k0, P);
        if (SUB == 0) v += k1;
        else if (SUB == 1) v = mmul(v, m1, P);
        else if (SUB == 2) v = mmul(v, m2, P);
        T = HALF ? mmul(v, m3, P) : v;
    }
    if (w == 3) B = T;
}

template <bool FACTORED = false>
LFHE_HD void ep_group2_phase5(EpState& s, double* lds, unsigned t, const Tables& T,
                              const double* __restrict__ rows, unsigned a1, unsigned a2) {
    const Prime P = T.pr[0];
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned zero = 0;
    asm volatile("" : "+v"(t));
    asm volatile("" : "+s"(zero));
    const double* __restrict__ twi = T.tw_inv[0] + zero;
    const double* __restrict__ mono = T.mono + zero;
#else
    const double* __restrict__ twi = T.tw_inv[0];
    const double* __restrict__ mono = T.mono;
#endif
    // psi^(e u_t) for the two exponents, u_t = 2 rev8(t) + 1: two gathers, in flight during the
    // last forward pass (X^(a + a') is never formed)
    const unsigned ut = 2 * rev8(t) + 1;
    const unsigned am[2] = {a1, a2};
    double base[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) base[m] = ldt(mono, (am[m] * ut) & (2 * N - 1));
    double* b0 = lds;
    double* b1 = lds + POLY_STRIDE;
    ld_o4p34(b0, t, s.x[0]); ld_o4p34(b1, t, s.x[1]);
    ct_radix4x2_bounded(s.x[0], s.tw, P); ct_radix4x2_bounded(s.x[1], s.tw, P);   // |d| < 3.4 p
    load_tw4x2(twi, t, s.tw);   // for the inverse pass below
    // position 8 t + j evaluates at psi^(2 bitrev11(8 t + j) + 1) = psi^(512 bitrev3(j) + 2 rev8(t) + 1)
    const double* __restrict__ w8 = mono + MONO_WORDS;
    // key words of unit u = 4 c4 + m: row m, the four polynomials q, positions 8 t + 2 c4 + {0, 1};
    // the next unit's four 16-byte loads are in flight while this one is multiplied
    double kc[4][2], kn[4][2];
#pragma unroll
    for (unsigned q = 0; q < 4; ++q) { const unsigned o = bsk_off(0, q, 0, t); kc[q][0] = LFHE_LDK(rows, o); kc[q][1] = LFHE_LDK(rows, o + 1); }
    auto chunk = [&](auto c4_tag) {
        constexpr int c4 = decltype(c4_tag)::value;
        double B[4][2], S[4][2], M1[2], M2[2];
        mono_pair<c4>(base[0], w8, am[0], P, M1[0], M1[1]);
        mono_pair<c4>(base[1], w8, am[1], P, M2[0], M2[1]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int u = 4 * c4 + m;
            if (u < 15) {
                const unsigned c4n = (unsigned)(u + 1) / 4, mn = (unsigned)(u + 1) % 4;
#pragma unroll
                for (unsigned q = 0; q < 4; ++q) {
                    const unsigned o = mn * (unsigned)BSKQ_ROW + bsk_off(0, q, c4n, t);
                    kn[q][0] = LFHE_LDK(rows, o); kn[q][1] = LFHE_LDK(rows, o + 1);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) ggsw_entry_row<FACTORED>(m, B[q][h], S[q][h], kc[q][h], M1[h], M2[h], P);
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::: "memory");
#endif
#pragma unroll
            for (int q = 0; q < 4; ++q) { kc[q][0] = kn[q][0]; kc[q][1] = kn[q][1]; }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = 2 * c4 + h;
            // |B| < 2.2 p times |d| < 3.4 p (ct_radix4x2_bounded): within the product's range
            const double d0 = s.x[0][j], d1 = s.x[1][j];
            s.x[0][j] = mmul(B[0][h], d0, P) + mmul(B[2][h], d1, P);
            s.x[1][j] = mmul(B[1][h], d0, P) + mmul(B[3][h], d1, P);
        }
    };
    chunk(std::integral_constant<int, 0>{});
    chunk(std::integral_constant<int, 1>{});
    chunk(std::integral_constant<int, 2>{});
    chunk(std::integral_constant<int, 3>{});
    gs_radix4x2<GS_NORM_PASS4[0], GS_NORM_PASS4[1]>(s.x[0], s.tw, P);
    gs_radix4x2<GS_NORM_PASS4[0], GS_NORM_PASS4[1]>(s.x[1], s.tw, P);
}

// coefficient e of X^a * poly: index to read and whether to negate it
LFHE_HD unsigned rot_index(unsigned e, unsigned a, bool& neg) {
    const unsigned ar = a & (N - 1);
    const bool wrap = e < ar;
    neg = ((a & N) != 0) != wrap;
    return wrap ? e + N - ar : e - ar;
}

// ---- ring modulus q' <-> 2^64
// round(v * q' / 2^64) as a centred exact double
LFHE_HD double torus_to_qp(u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u64 hi = __umul64hi(v, QP_U);
#else
    const u64 hi = (u64)(((unsigned __int128)v * QP_U) >> 64);
#endif
    u64 r = hi + ((v * QP_U) >> 63);
    if (r == QP_U) r = 0;
    return r > QP_U / 2 ? -(double)(QP_U - r) : (double)r;
}
// round(r * 2^64 / q') for the centred residue c, by restoring division (64 steps)
LFHE_HD u64 qp_to_torus(double c) {
    u64 r = c < 0 ? QP_U - (u64)(-c) : (u64)c;
    u64 w = 0;
#pragma nounroll
    for (int i = 0; i < 64; ++i) {
        r <<= 1;
        w <<= 1;
        if (r >= QP_U) { r -= QP_U; w |= 1; }
    }
    return w + (2 * r >= QP_U ? 1 : 0);
}

LFHE_HD u32 modswitch_2n(u64 x) {  // round to Z_{2N}
    u64 t = x >> (64 - (LOGN + 1) - 1);
    return (u32)(((t + 1) >> 1) & (2 * N - 1));
}

}  // namespace lfhe
