// HIP kernels of the bootstrap path for gfx950 (MI355X): bootstrapping-key preparation (forward
// negacyclic NTT), inverse NTT, GGSW external product, blind rotation with fused modulus switch
// and sample extraction, LWE key switch, and the small linear LWE kernels the radix layer needs.
//
// One 256-thread workgroup owns one ciphertext; its accumulator lives in registers, its
// decomposed polynomials go through four register passes with LDS exchanges in between
// (pbs_device.h / pbs_phases.h), and the NTT-domain key row of the current iteration is streamed
// from L2/HBM with contiguous 16-byte loads.  Workgroups of different ciphertexts walk the key in
// near lock-step, so each row is fetched from HBM about once per XCD and served from L2 to the
// rest.
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdexcept>

#include <cstdlib>
#include <utility>

#include "engine.h"
#include "pbs_phases.h"

#ifndef LFHE_BR_WAVES
// Register budget of the blind-rotation kernel, as waves per SIMD.  Measured on MI355X with 4096
// ciphertexts: budget 2 (the compiler then uses 166 VGPRs, so three workgroups still fit per CU)
// 163 ms; budget 3 (156 VGPRs, same occupancy, tighter schedule) 191 ms; budget 4 (128 VGPRs with
// 116 B of scratch) 201 ms.  The kernel is bound by FP64-rate VALU issue, not by latency.
#define LFHE_BR_WAVES 2
#endif

namespace lfhe {

extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
static constexpr size_t LDS_BYTES = (size_t)LDS_WORDS * sizeof(double);

// The separation a hand-over between two phases needs (pbs_phases.h: sync_after).  Inside one
// wavefront LDS operations execute in program order, so a hand-over private to a wave only has to
// keep the compiler from moving the loads above the stores.
#ifdef LFHE_ABLATE_NO_BARRIER   // timing experiment only: results are wrong
template <Sync K> __device__ __forceinline__ void phase_sync() {}
#else
template <Sync K>
__device__ __forceinline__ void phase_sync() {
    if constexpr (K == Sync::Block) {
        __syncthreads();
    } else if constexpr (K == Sync::Wave) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
#endif
__device__ __forceinline__ void wave_sync() { phase_sync<Sync::Wave>(); }

template <int... PH>
__device__ __forceinline__ void ep_run_seq(EpState& s, double* lds, unsigned t, const Tables& T,
                                           const double* __restrict__ bsk_i,
                                           std::integer_sequence<int, PH...>) {
    ((ep_phase<PH>(s, lds, t, T, bsk_i), __syncthreads()), ...);
}

__device__ __forceinline__ void ep_run(EpState& s, double* lds, unsigned t, const Tables& T,
                                       const double* __restrict__ bsk_i) {
    ep_run_seq(s, lds, t, T, bsk_i, std::make_integer_sequence<int, EP_PHASES>{});
}

// ------------------------------------------------------------------------------------------------
// forward transform of torus polynomials, both primes.
//   mode 0: out[(poly*2 + r)*N + e]  centred residues in transform order (for parity checks)
//   mode 1: bootstrapping-key layout with N^-1 folded in; poly = i*4 + q
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) k_ntt_forward(Tables T, const u64* __restrict__ in,
                                                    double* __restrict__ out, int mode) {
    double* buf = lds_dyn;
    const unsigned t = threadIdx.x;
    const size_t poly = blockIdx.x;
    u64 v[E];
#pragma unroll
    for (int j = 0; j < E; ++j) v[j] = in[poly * N + own1(t, j)];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const Prime P = T.pr[r];
        const double* __restrict__ tw = T.tw_fwd[r];
        double x[E];
#pragma unroll
        for (int j = 0; j < E; ++j) x[j] = torus_to_res(v[j], T.two32[r], P);
        ct_radix8(x, tw, 1, P);
        st_o1p12(buf, t, x); __syncthreads(); ld_o2p12(buf, t, x); __syncthreads();
        ct_radix8(x, tw, base2(t), P);
        st_o2p23(buf, t, x); __syncthreads(); ld_o3p23(buf, t, x); __syncthreads();
        ct_radix8(x, tw, base3(t), P);
        st_o3p34(buf, t, x); __syncthreads(); ld_o4p34(buf, t, x); __syncthreads();
        ct_radix4x2(x, tw, t, P);
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < E; ++j) out[(poly * 2 + r) * N + own4(t, j)] = mnorm(x[j], P);
        } else {
            const size_t i = poly >> 2;
            const int q = (int)(poly & 3);
            double* row = out + i * BSK_ROW;
#pragma unroll
            for (int j = 0; j < E; ++j)
                row[bsk_off(r, q, j >> 1, t) + (j & 1)] = mnorm(mmul(x[j], T.ninv[r], P), P);
        }
    }
}

// inverse transform + CRT lift: in[(poly*2 + r)*N + e] residues in transform order -> torus
__global__ void __launch_bounds__(NT) k_ntt_inverse(Tables T, const double* __restrict__ in,
                                                    u64* __restrict__ out) {
    double* buf = lds_dyn;
    const unsigned t = threadIdx.x;
    const size_t poly = blockIdx.x;
    double y0[E];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const Prime P = T.pr[r];
        const double* __restrict__ tw = T.tw_inv[r];
        double x[E];
#pragma unroll
        for (int j = 0; j < E; ++j) x[j] = mmul(in[(poly * 2 + r) * N + own4(t, j)], T.ninv[r], P);
        gs_radix4x2<0xFF, 0xFF>(x, tw, t, P);
        st_o4p34(buf, t, x); __syncthreads(); ld_o3p34(buf, t, x); __syncthreads();
        gs_radix8<0, 0xFF, 0xFF>(x, tw, base3(t), P);
        st_o3p23(buf, t, x); __syncthreads(); ld_o2p23(buf, t, x); __syncthreads();
        gs_radix8<0, 0xFF, 0xFF>(x, tw, base2(t), P);
        st_o2p12(buf, t, x); __syncthreads(); ld_o1p12(buf, t, x); __syncthreads();
        gs_radix8<0, 0xFF, 0xFF>(x, tw, 1, P);
        if (r == 0) {
#pragma unroll
            for (int j = 0; j < E; ++j) y0[j] = x[j];
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j)
                out[poly * N + own1(t, j)] = crt_wrap(y0[j], x[j], T.p0inv_mod_p1, T.pr[1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out = BSK[ggsw_index[ct]] [.] glwe[ct]    (one workgroup per ciphertext)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT, 2) k_external_product(Tables T, const double* __restrict__ bsk,
                                                            const u32* __restrict__ ggsw_index,
                                                            const u64* __restrict__ glwe,
                                                            u64* __restrict__ out) {
    const unsigned t = threadIdx.x;
    const size_t ct = blockIdx.x;
    EpState s;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            s.dig[q][j] = decomp23(glwe[(ct * 2 + q) * N + own1(t, j)]);
            s.out[q][j] = 0;
        }
    ep_run(s, lds_dyn, t, T, bsk + (size_t)ggsw_index[ct] * BSK_ROW);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < E; ++j) out[(ct * 2 + q) * N + own1(t, j)] = s.out[q][j];
}

// ------------------------------------------------------------------------------------------------
// blind rotation: modulus switch of the small ciphertext, acc = X^-b~ * lut, n CMux steps,
// then either the whole accumulator (glwe_out) or its sample-extracted LWE (lwe_out).
//
// The accumulator (2 x 2048 words) stays on the CU for all n steps: each thread holds its 16
// coefficients in registers and republishes them after every step in the LDS exchange buffer
// (free between steps), from which the next step takes its rotated reads.  A workspace in global
// memory was measured slower: with a thousand resident workgroups it no longer fits the L2 and
// every step pays a round trip for 64 KB per ciphertext.
// ------------------------------------------------------------------------------------------------
template <int... PH>
__device__ __forceinline__ void ep_run_assign(EpState& s, double* lds, unsigned t, const Tables& T,
                                              const double* __restrict__ bsk_i,
                                              std::integer_sequence<int, PH...>) {
    // nothing after the last phase: the caller publishes its result first
    ((ep_phase<PH, false>(s, lds, t, T, bsk_i), phase_sync<(PH + 1 < EP_PHASES ? sync_after(PH) : Sync::None)>()), ...);
}

__global__ void __launch_bounds__(NT, LFHE_BR_WAVES) k_blind_rotate(Tables T, const double* __restrict__ bsk,
                                                        const u64* __restrict__ lwe_small, int n,
                                                        const u64* __restrict__ luts,
                                                        const u32* __restrict__ lut_index,
                                                        u64* __restrict__ glwe_out,
                                                        u64* __restrict__ lwe_out, BlockMap omap) {
    const unsigned t = threadIdx.x;
    const size_t ct = blockIdx.x;
    const u64* __restrict__ lwe = lwe_small + ct * (size_t)(n + 1);
    const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct % omap.ops : ct] : 0) * N;

    // The accumulator never leaves the CU: every thread keeps its own 16 coefficients in registers
    // (pass-1 ownership) and publishes them between steps in the LDS exchange buffer, which serves
    // the rotated reads of the next step.  The global workspace is written once, at the end.
    u64* acc_lds[2] = {reinterpret_cast<u64*>(lds_dyn), reinterpret_cast<u64*>(lds_dyn + POLY_STRIDE)};
    u64 acc[2][E];
    {
        const unsigned bt = modswitch_2n(lwe[n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const unsigned e = own1(t, j);
            bool neg;
            const u64 v = lut[rot_index(e, a, neg)];
            acc[0][j] = 0;
            acc[1][j] = neg ? (u64)0 - v : v;
            acc_lds[0][pad12(e)] = acc[0][j];
            acc_lds[1][pad12(e)] = acc[1][j];
        }
    }
    __syncthreads();

    EpState s;
    for (int i = 0; i < n; ++i) {
        const unsigned a = modswitch_2n(lwe[i]);
        if (a == 0) continue;  // X^0 - 1 = 0: this CMux adds exactly zero
        unsigned tt = t;
        asm volatile("" : "+v"(tt));  // recompute the addresses per step (see ep_phase)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                bool neg;
                const u64 r = acc_lds[q][pad12(rot_index(own1(tt, j), a, neg))];
                s.dig[q][j] = decomp23((neg ? (u64)0 - r : r) - acc[q][j]);
            }
        __syncthreads();  // every rotated read is done before the first phase overwrites the buffer
        ep_run_assign(s, lds_dyn, t, T, bsk + (size_t)i * BSK_ROW, std::make_integer_sequence<int, EP_PHASES>{});
        // a thread's last-pass reads and its accumulator writes hit the same LDS words: no barrier between
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                acc[q][j] += s.out[q][j];
                acc_lds[q][pad12(own1(tt, j))] = acc[q][j];
            }
        __syncthreads();  // the new accumulator is visible to the whole workgroup before the next step
    }
    if (glwe_out) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) glwe_out[(ct * 2 + q) * N + own1(t, j)] = acc[q][j];
    }
    if (lwe_out) {  // sample extraction of coefficient 0: every thread places its own coefficients
        u64* o = lwe_out + omap.block(ct) * (size_t)(N + 1);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const unsigned e = own1(t, j);
            if (e == 0) { o[0] = acc[0][j]; o[N] = acc[1][j]; }
            else o[N - e] = (u64)0 - acc[0][j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The bootstrap over the single ring modulus q' = p0 (pbs_phases.h): bootstrapping key and
// accumulator are residues mod q', one transform per polynomial, no lift.  The accumulator
// workspace holds centred residues as doubles.
// ------------------------------------------------------------------------------------------------
#ifndef LFHE_BRQ_WAVES
#define LFHE_BRQ_WAVES 2
#endif

// key preparation: residues in [0, q') -> transform domain, N^-1 folded in, key-row layout
__global__ void __launch_bounds__(NT) k_bskq_prepare(Tables T, const u64* __restrict__ in, double* __restrict__ out) {
    double* buf = lds_dyn;
    const unsigned t = threadIdx.x;
    const size_t poly = blockIdx.x;
    const Prime P = T.pr[0];
    const double* __restrict__ tw = T.tw_fwd[0];
    double x[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const u64 v = in[poly * N + own1(t, j)];
        x[j] = v > QP_U / 2 ? -(double)(QP_U - v) : (double)v;
    }
    ct_radix8(x, tw, 1, P);
    st_o1p12(buf, t, x); __syncthreads(); ld_o2p12(buf, t, x); __syncthreads();
    ct_radix8(x, tw, base2(t), P);
    st_o2p23(buf, t, x); __syncthreads(); ld_o3p23(buf, t, x); __syncthreads();
    ct_radix8(x, tw, base3(t), P);
    st_o3p34(buf, t, x); __syncthreads(); ld_o4p34(buf, t, x); __syncthreads();
    ct_radix4x2(x, tw, t, P);
    double* row = out + (poly >> 2) * BSKQ_ROW;
    const unsigned q = (unsigned)(poly & 3);
#pragma unroll
    for (int j = 0; j < E; ++j) row[bsk_off(0, q, j >> 1, t) + (j & 1)] = mnorm(mmul(x[j], T.ninv[0], P), P);
}

template <int... PH>
__device__ __forceinline__ void epq_run(EpState& s, double* lds, unsigned t, const Tables& T,
                                        const double* __restrict__ bsk_i, std::integer_sequence<int, PH...>) {
    ((ep_phase<PH, false, true>(s, lds, t, T, bsk_i), phase_sync<(PH + 1 < EPQ_PHASES ? sync_after(PH) : Sync::None)>()), ...);
}

// phases 1..11 (the kernel runs phase 0 itself)
template <int... PH>
__device__ __forceinline__ void epq_run_from1(EpState& s, double* lds, unsigned t, const Tables& T,
                                              const double* __restrict__ bsk_i, std::integer_sequence<int, PH...>) {
    // workgroup barriers throughout: measured faster than wave-level fences at full occupancy (see LFHE_G2_BLOCK_MASK)
    ((ep_phase<PH + 1, false, true>(s, lds, t, T, bsk_i), phase_sync<(PH + 2 < EPQ_PHASES ? Sync::Block : Sync::None)>()), ...);
}

// centred residue of an exact integer |x| < 3 q' (the quotient estimate is exact in this range)
__device__ __forceinline__ double qp_canon(double x, const Prime& P) { return mnorm(x, P); }

// out = BSK[ggsw_index[ct]] [.] glwe[ct] over Z_q'; glwe and out hold residues in [0, q')
__global__ void __launch_bounds__(NT, 2) k_external_product_q(Tables T, const double* __restrict__ bsk,
                                                              const u32* __restrict__ ggsw_index,
                                                              const u64* __restrict__ glwe, u64* __restrict__ out,
                                                              double inv_g) {
    const unsigned t = threadIdx.x;
    const size_t ct = blockIdx.x;
    const Prime P = T.pr[0];
    EpState s;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const u64 v = glwe[(ct * 2 + q) * N + own1(t, j)];
            const double c = v > QP_U / 2 ? -(double)(QP_U - v) : (double)v;
            s.x[q][j] = __builtin_rint(c * inv_g);
        }
    epq_run(s, lds_dyn, t, T, bsk + (size_t)ggsw_index[ct] * BSKQ_ROW, std::make_integer_sequence<int, EPQ_PHASES>{});
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double c = qp_canon(s.x[q][j], P);
            out[(ct * 2 + q) * N + own1(t, j)] = c < 0 ? QP_U - (u64)(-c) : (u64)c;
        }
}

__global__ void __launch_bounds__(NT, LFHE_BRQ_WAVES) k_blind_rotate_q(Tables T, const double* __restrict__ bsk,
                                                                       const u64* __restrict__ lwe_small, int n,
                                                                       const u64* __restrict__ luts,
                                                                       const u32* __restrict__ lut_index,
                                                                       u64* __restrict__ glwe_out,
                                                                       u64* __restrict__ lwe_out, BlockMap omap,
                                                                       double inv_g) {
    const unsigned t = threadIdx.x;
    const size_t ct = blockIdx.x;
    const Prime P = T.pr[0];
    const u64* __restrict__ lwe = lwe_small + ct * (size_t)(n + 1);
    const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct % omap.ops : ct] : 0) * N;

    // The accumulator never leaves the CU: every thread keeps its own 16 coefficients in registers
    // (pass-1 ownership) and publishes them between steps in the LDS exchange buffer, which serves
    // the rotated reads of the next step.  The global workspace is written once, at the end.
    double* acc_lds[2] = {lds_dyn, lds_dyn + POLY_STRIDE};
    double acc[2][E];
    {
        const unsigned bt = modswitch_2n(lwe[n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const unsigned e = own1(t, j);
            bool neg;
            const double v = torus_to_qp(lut[rot_index(e, a, neg)]);
            acc[0][j] = 0.0;
            acc[1][j] = neg ? -v : v;
            acc_lds[0][pad12(e)] = acc[0][j];
            acc_lds[1][pad12(e)] = acc[1][j];
        }
    }
    __syncthreads();

    EpState s;
    for (int i = 0; i < n; ++i) {
        const unsigned a = modswitch_2n(lwe[i]);
        if (a == 0) continue;  // X^0 - 1 = 0: this CMux adds exactly zero
        unsigned tt = t;
        asm volatile("" : "+v"(tt));  // recompute the addresses per step (see ep_phase)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                bool neg;
                const double r = acc_lds[q][pad12(rot_index(own1(tt, j), a, neg))];
                const double d = (neg ? -r : r) - acc[q][j];
                s.x[q][j] = __builtin_rint(qp_canon(d, P) * inv_g);
            }
        // first forward pass in registers, then publish: every thread must have read its rotated
        // coefficients before the buffer is overwritten
        ct_radix8(s.x[0], T.tw_fwd[0], 1, P);
        ct_radix8(s.x[1], T.tw_fwd[0], 1, P);
        __syncthreads();
        st_o1p12(acc_lds[0], tt, s.x[0]); st_o1p12(acc_lds[1], tt, s.x[1]);
        load_tw8(T.tw_fwd[0], base2(tt), s.tw);   // twiddles of the next pass, one phase ahead
        __syncthreads();
        epq_run_from1(s, lds_dyn, t, T, bsk + (size_t)i * BSKQ_ROW, std::make_integer_sequence<int, EPQ_PHASES - 1>{});
        // a thread's last-pass reads and its accumulator writes hit the same LDS words: no barrier between
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                acc[q][j] = qp_canon(acc[q][j] + s.x[q][j], P);
                acc_lds[q][pad12(own1(tt, j))] = acc[q][j];
            }
        __syncthreads();  // the new accumulator is visible to the whole workgroup before the next step
    }
    if (glwe_out) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const double c = acc[q][j];
                glwe_out[(ct * 2 + q) * N + own1(t, j)] = c < 0 ? QP_U - (u64)(-c) : (u64)c;
            }
    }
    if (lwe_out) {  // sample extraction of coefficient 0, every word switched back to 2^64
        u64* o = lwe_out + omap.block(ct) * (size_t)(N + 1);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const unsigned e = own1(t, j);
            if (e == 0) { o[0] = qp_to_torus(acc[0][j]); o[N] = qp_to_torus(acc[1][j]); }
            else o[N - e] = qp_to_torus(-acc[0][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS image of the 512-thread kernels.  One such workgroup owns a CU, so besides the two exchange
// buffers it keeps the twiddle tables of passes 2-4 (and the monomial table of the grouped key) in
// LDS.  Vector memory loads return in order: with the tables in LDS the key words are the only
// vector loads of a step, so the NEXT step's key words can be requested as soon as the point-wise
// stage has consumed this step's (a whole step of latency cover, in the same registers) without
// any twiddle load queueing behind them.
// ------------------------------------------------------------------------------------------------
constexpr int WIDE_TWF = LDS_WORDS;                 // word offsets
constexpr int WIDE_TWI = WIDE_TWF + N;
constexpr int WIDE_MONO = WIDE_TWI + N;
// one-bit kernel: exchange buffers of the inverse transform and the published accumulator behind the tables
constexpr int WIDE_BACK = WIDE_MONO;
constexpr int WIDE_ACC = WIDE_BACK + LDS_WORDS;
static constexpr size_t WIDE_LDS_BYTES = (size_t)(WIDE_ACC + LDS_WORDS) * sizeof(double);
constexpr int WIDE2_BACK = WIDE_MONO + MONO_WORDS + 8;   // grouped kernel: exchange buffers of the inverse transform
static constexpr size_t WIDE2_LDS_BYTES = (size_t)(WIDE2_BACK + LDS_WORDS) * sizeof(double);

__device__ __forceinline__ void wide_fill_tables(const Tables& T, bool mono) {
    for (unsigned i = threadIdx.x; i < (unsigned)N; i += 2 * NT) {
        lds_dyn[WIDE_TWF + i] = T.tw_fwd[0][i];
        lds_dyn[WIDE_TWI + i] = T.tw_inv[0][i];
    }
    if (mono)
        for (unsigned i = threadIdx.x; i < (unsigned)MONO_WORDS + 8; i += 2 * NT) lds_dyn[WIDE_MONO + i] = T.mono[i];
}

// The seven twiddles of pass 1 are the same for every thread and every step: held in scalar
// registers for the whole kernel instead of being loaded (and waited for) in every step.
__device__ __forceinline__ double uniform_f64(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void load_tw8_uniform(const double* __restrict__ tw, double (&w)[7]) {
    load_tw8(tw, 1, w);
#pragma unroll
    for (int k = 0; k < 7; ++k) w[k] = uniform_f64(w[k]);
}

// ------------------------------------------------------------------------------------------------
// The same blind rotation for launches too small to fill the GPU (one operation's bootstrap layer
// is often a few dozen ciphertexts): 512 threads per ciphertext, threads 0-255 own the mask
// polynomial, 256-511 the body polynomial.  Each half decomposes, transforms and inverse-transforms
// its own polynomial in its own LDS buffer; the only coupling is the point-wise stage, where each
// half publishes its transformed digits and reads the other half's.  A step has half the
// instructions per thread of k_blind_rotate_q and one more barrier.  Results are bit-identical.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(2 * NT, 1) k_blind_rotate_q_wide(Tables T, const double* __restrict__ bsk,
                                                                   const u64* __restrict__ lwe_small, int n,
                                                                   const u64* __restrict__ luts,
                                                                   const u32* __restrict__ lut_index,
                                                                   u64* __restrict__ glwe_out,
                                                                   u64* __restrict__ lwe_out, BlockMap omap,
                                                                   double inv_g) {
    const unsigned h = threadIdx.x >> 8;   // polynomial of this half (wave-uniform)
    const unsigned t0 = threadIdx.x & (NT - 1);
    const size_t ct = blockIdx.x;
    const Prime P = T.pr[0];
    const u64* __restrict__ lwe = lwe_small + ct * (size_t)(n + 1);
    const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct % omap.ops : ct] : 0) * N;
    double* mine = lds_dyn + h * POLY_STRIDE;
    double* theirs = lds_dyn + (1 - h) * POLY_STRIDE;
    const double* __restrict__ twf = lds_dyn + WIDE_TWF;   // passes 2-4; pass 1's live in scalar registers
    const double* __restrict__ twi = lds_dyn + WIDE_TWI;
    // The inverse transform and the published accumulator (the next step's rotated reads) have
    // buffers of their own, which leaves four workgroup barriers per step: after pass 1 of the
    // forward transform is published, after the transformed digits are, before pass 1 of the
    // inverse transform reads, and after the new accumulator is published.
    double* back = lds_dyn + WIDE_BACK + h * POLY_STRIDE;
    double* accb = lds_dyn + WIDE_ACC + h * POLY_STRIDE;
    wide_fill_tables(T, false);

    double acc[E], x[E], w[7], w1f[7], w1i[7];
    load_tw8_uniform(T.tw_fwd[0], w1f);
    load_tw8_uniform(T.tw_inv[0], w1i);
    {
        const unsigned bt = modswitch_2n(lwe[n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const unsigned e = own1(t0, j);
            bool neg;
            const double v = torus_to_qp(lut[rot_index(e, a, neg)]);
            acc[j] = h ? (neg ? -v : v) : 0.0;
            accb[pad12(e)] = acc[j];
        }
    }
    __syncthreads();

    // Key row of a step: kx multiplies this half's own digits, ky the other half's.  The row of step
    // i + 1 is requested right after the point-wise stage of step i has used up the registers.
    double kx[E], ky[E];
    auto load_key = [&](int i, unsigned t) {
        const double* __restrict__ row = bsk + (size_t)i * BSKQ_ROW;
#pragma unroll
        for (unsigned c4 = 0; c4 < 4; ++c4) {
            const unsigned o0 = bsk_off(0, 3 * h, c4, t), o1 = bsk_off(0, 2 - h, c4, t);
#pragma unroll
            for (unsigned k = 0; k < 2; ++k) { kx[2 * c4 + k] = ldt(row, o0 + k); ky[2 * c4 + k] = ldt(row, o1 + k); }
        }
    };
    // first step at or after i that is not the identity (X^0 - 1 = 0: such a CMux adds exactly zero)
    auto next_step = [&](int i) {
        while (i < n && modswitch_2n(lwe[i]) == 0) ++i;
        return i;
    };
    int i = next_step(0);
    if (i < n) load_key(i, t0);

    while (i < n) {
        const unsigned a = modswitch_2n(lwe[i]);
        unsigned t = t0;
        asm volatile("" : "+v"(t));  // addresses are recomputed per step instead of living in registers
#pragma unroll
        for (int j = 0; j < E; ++j) {
            bool neg;
            const double r = accb[pad12(rot_index(own1(t, j), a, neg))];
            x[j] = __builtin_rint(qp_canon((neg ? -r : r) - acc[j], P) * inv_g);
        }
        ct_radix8(x, w1f, P);
        st_o1p12(mine, t, x); load_tw8(twf, base2(t), w);
        __syncthreads();
        ld_o2p12(mine, t, x); ct_radix8(x, w, P);
        wave_sync();
        st_o2p23(mine, t, x); load_tw8(twf, base3(t), w);
        wave_sync();
        ld_o3p23(mine, t, x); ct_radix8(x, w, P);
        wave_sync();
        st_o3p34(mine, t, x); load_tw4x2(twf, t, w);
        wave_sync();
        ld_o4p34(mine, t, x); ct_radix4x2(x, w, P);
        // publish the transformed digits (a thread rewrites the words it has just read) and fetch
        // the other polynomial's
        st_o4p34(mine, t, x); load_tw4x2(twi, t, w);
        __syncthreads();
        {
            double y[E];
            ld_o4p34(theirs, t, y);
            // out_h = D_h K[h][h] + D_(1-h) K[1-h][h]; key polynomial index q = j_in * 2 + c_out
            double o[E];
#pragma unroll
            for (int j = 0; j < E; ++j) o[j] = mmul(x[j], kx[j], P) + mmul(y[j], ky[j], P);
#pragma unroll
            for (int j = 0; j < E; ++j) x[j] = o[j];
        }
        // the next step's loads must stay below the last use of this step's words (same registers):
        // their addresses are made to depend on the point-wise results
        unsigned tk = t;
        asm volatile("" : "+v"(tk) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
        const int inext = next_step(i + 1);
        if (inext < n) load_key(inext, tk);
        gs_radix4x2<GS_NORM_PASS4[0], GS_NORM_PASS4[1]>(x, w, P);
        st_o4p34(back, t, x); load_tw8(twi, base3(t), w);
        wave_sync();
        ld_o3p34(back, t, x); gs_radix8<GS_NORM_PASS3[0], GS_NORM_PASS3[1], GS_NORM_PASS3[2]>(x, w, P);
        wave_sync();
        st_o3p23(back, t, x); load_tw8(twi, base2(t), w);
        wave_sync();
        ld_o2p23(back, t, x); gs_radix8<GS_NORM_PASS2[0], GS_NORM_PASS2[1], GS_NORM_PASS2[2]>(x, w, P);
        wave_sync();
        st_o2p12(back, t, x);
        __syncthreads();
        ld_o1p12(back, t, x); gs_radix8<GS_NORM_PASS1[0], GS_NORM_PASS1[1], GS_NORM_PASS1[2]>(x, w1i, P);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            acc[j] = qp_canon(acc[j] + x[j], P);
            accb[pad12(own1(t, j))] = acc[j];
        }
        __syncthreads();                      // the new accumulator is visible before the next step's rotated reads
        i = inext;
    }

    if (glwe_out) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double c = acc[j];
            glwe_out[(ct * 2 + h) * N + own1(t0, j)] = c < 0 ? QP_U - (u64)(-c) : (u64)c;
        }
    }
    if (lwe_out) {  // sample extraction of coefficient 0, every word switched back to 2^64
        u64* o = lwe_out + omap.block(ct) * (size_t)(N + 1);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const unsigned e = own1(t0, j);
            if (h == 0) o[e == 0 ? 0 : N - e] = qp_to_torus(e == 0 ? acc[j] : -acc[j]);
            else if (e == 0) o[N] = qp_to_torus(acc[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Blind rotation with two mask elements per step (grouped key, pbs_phases.h: ep_group2_phase5).
// The accumulator lives in registers only: a step decomposes it, transforms the digits, forms the
// step's GGSW point by point from three key rows and the monomial factors, multiplies, transforms
// back and REPLACES the accumulator -- no rotated reads, hence no accumulator copy in LDS.  (n + 1) / 2 steps of one external product each.
//
// LOCKSTEP: workgroup barriers after every phase, also where the data flow needs a wave-level fence
// only.  A launch that keeps more than one round of workgroups busy is FASTER with them (4096
// ciphertexts: 48.0 ms against 49.8 ms with fences, 49.3 ms with only the barrier before the
// point-wise stage kept; 2048: 24.3 against 24.6 ms): waves that stay in lock step walk the key rows
// together.  Up to one round of workgroups the fences win (384: 6.5 against 6.9 ms, 768: 10.1
// against 10.5 ms).
// ------------------------------------------------------------------------------------------------
#define LFHE_G2_SYNC() phase_sync<Sync::Block>()
template <bool LOCKSTEP>
constexpr Sync g2_sync_after(int PH) { return LOCKSTEP ? Sync::Block : sync_after(PH); }
template <bool LOCKSTEP, int... PH>
__device__ __forceinline__ void epq_run_range(EpState& s, double* lds, unsigned t, const Tables& T,
                                              std::integer_sequence<int, PH...>) {
    ((ep_phase<PH, false, true>(s, lds, t, T, nullptr), phase_sync<g2_sync_after<LOCKSTEP>(PH)>()), ...);
}
template <int FIRST, int... K>
constexpr auto phase_range(std::integer_sequence<int, K...>) { return std::integer_sequence<int, (FIRST + K)...>{}; }

// The twiddles of passes 1 and 2 -- table entries 1..63 of either direction, 1 KiB together -- sit
// in LDS behind the exchange buffers (four workgroups still fit a CU): they are what every thread
// asks for with few distinct addresses, and taking them off the vector memory path leaves it to the
// key rows and the pass-3/4 twiddles.
#ifndef LFHE_G2_TW_SMALL
#define LFHE_G2_TW_SMALL 64
#endif
constexpr int G2_TW_SMALL = LFHE_G2_TW_SMALL;
static constexpr size_t G2_LDS_BYTES = LDS_BYTES + 2 * G2_TW_SMALL * sizeof(double);
// twiddles of pass 3 (first-stage index b in [64, 128)): entries below G2_TW_SMALL from LDS
__device__ __forceinline__ void load_tw8_pass3(const double* __restrict__ small, const double* __restrict__ tw,
                                               unsigned b, double (&w)[7]) {
    if constexpr (G2_TW_SMALL >= 256) {
        w[0] = small[b]; w[1] = small[2 * b]; w[2] = small[2 * b + 1];
#pragma unroll
        for (unsigned k = 0; k < 4; ++k) w[3 + k] = ldtw(tw, 4 * b + k);
    } else {
        load_tw8(tw, b, w);
    }
}

template <bool LOCKSTEP, bool FACTORED>
__global__ void __launch_bounds__(NT, 4) k_blind_rotate_q2(Tables T, const double* __restrict__ bsk,
                                                                        const u64* __restrict__ lwe_small, int n,
                                                                        const u64* __restrict__ luts,
                                                                        const u32* __restrict__ lut_index,
                                                                        u64* __restrict__ glwe_out,
                                                                        u64* __restrict__ lwe_out, BlockMap omap,
                                                                        double inv_g) {
    const unsigned t = threadIdx.x;
    const size_t ct = blockIdx.x;
    const Prime P = T.pr[0];
    const u64* __restrict__ lwe = lwe_small + ct * (size_t)(n + 1);
    const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct % omap.ops : ct] : 0) * N;

    double acc[2][E];
    {
        const unsigned bt = modswitch_2n(lwe[n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            bool neg;
            const double v = torus_to_qp(lut[rot_index(own1(t, j), a, neg)]);
            acc[0][j] = 0.0;
            acc[1][j] = neg ? -v : v;
        }
    }
    const double* __restrict__ twf_small = lds_dyn + LDS_WORDS;   // entries 0..63 of the forward table
    const double* __restrict__ twi_small = twf_small + G2_TW_SMALL;
    for (unsigned i = t; i < (unsigned)G2_TW_SMALL; i += NT) {
        lds_dyn[LDS_WORDS + i] = T.tw_fwd[0][i];
        lds_dyn[LDS_WORDS + G2_TW_SMALL + i] = T.tw_inv[0][i];
    }
    LFHE_G2_SYNC();

    EpState s;
    double* b0 = lds_dyn;
    double* b1 = lds_dyn + POLY_STRIDE;
    const int groups = (n + 1) / 2;
    for (int g = 0; g < groups; ++g) {
        const unsigned a0 = modswitch_2n(lwe[2 * g]);
        const unsigned a1 = 2 * g + 1 < n ? modswitch_2n(lwe[2 * g + 1]) : 0u;
        unsigned tt = t;
        asm volatile("" : "+v"(tt));  // recompute the addresses per step (see ep_phase)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) s.x[q][j] = __builtin_rint(acc[q][j] * inv_g);
        // phase 0
        ct_radix8(s.x[0], twf_small, 1, P);
        ct_radix8(s.x[1], twf_small, 1, P);
        st_o1p12(b0, tt, s.x[0]); st_o1p12(b1, tt, s.x[1]);
        load_tw8(twf_small, base2(tt), s.tw);   // twiddles of the next pass, one phase ahead
        LFHE_G2_SYNC();
        epq_run_range<LOCKSTEP>(s, lds_dyn, t, T, phase_range<1>(std::make_integer_sequence<int, 1>{}));
        // phases 2, 6, 8 and 11 as in ep_phase, with twiddles from LDS
        asm volatile("" : "+v"(tt));
        st_o2p23(b0, tt, s.x[0]); st_o2p23(b1, tt, s.x[1]);
        load_tw8_pass3(twf_small, T.tw_fwd[0], base3(tt), s.tw);
        phase_sync<g2_sync_after<LOCKSTEP>(2)>();
        epq_run_range<LOCKSTEP>(s, lds_dyn, t, T, phase_range<3>(std::make_integer_sequence<int, 2>{}));
        ep_group2_phase5<FACTORED>(s, lds_dyn, t, T, bsk + (size_t)4 * g * BSKQ_ROW, a0, a1);
        phase_sync<g2_sync_after<LOCKSTEP>(5)>();   // the next phase rewrites the words this thread has just read
        asm volatile("" : "+v"(tt));
        st_o4p34(b0, tt, s.x[0]); st_o4p34(b1, tt, s.x[1]);
        load_tw8_pass3(twi_small, T.tw_inv[0], base3(tt), s.tw);
        phase_sync<g2_sync_after<LOCKSTEP>(6)>();
        epq_run_range<LOCKSTEP>(s, lds_dyn, t, T, phase_range<7>(std::make_integer_sequence<int, 1>{}));
        asm volatile("" : "+v"(tt));
        st_o3p23(b0, tt, s.x[0]); st_o3p23(b1, tt, s.x[1]);
        load_tw8(twi_small, base2(tt), s.tw);
        phase_sync<g2_sync_after<LOCKSTEP>(8)>();
        epq_run_range<LOCKSTEP>(s, lds_dyn, t, T, phase_range<9>(std::make_integer_sequence<int, 2>{}));
        asm volatile("" : "+v"(tt));
        ld_o1p12(b0, tt, s.x[0]); ld_o1p12(b1, tt, s.x[1]);
        gs_radix8<GS_NORM_PASS1[0], GS_NORM_PASS1[1], GS_NORM_PASS1[2]>(s.x[0], twi_small, 1, P);
        gs_radix8<GS_NORM_PASS1[0], GS_NORM_PASS1[1], GS_NORM_PASS1[2]>(s.x[1], twi_small, 1, P);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) acc[q][j] = qp_canon(s.x[q][j], P);
        LFHE_G2_SYNC();  // the last pass's reads are done before the next step overwrites the buffer
    }
    if (glwe_out) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const double c = acc[q][j];
                glwe_out[(ct * 2 + q) * N + own1(t, j)] = c < 0 ? QP_U - (u64)(-c) : (u64)c;
            }
    }
    if (lwe_out) {  // sample extraction of coefficient 0, every word switched back to 2^64
        u64* o = lwe_out + omap.block(ct) * (size_t)(N + 1);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const unsigned e = own1(t, j);
            if (e == 0) { o[0] = qp_to_torus(acc[0][j]); o[N] = qp_to_torus(acc[1][j]); }
            else o[N - e] = qp_to_torus(-acc[0][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The grouped blind rotation with TWO ciphertexts per workgroup.  Every thread owns the same
// positions of both ciphertexts: the key words of a step are loaded once and multiplied into both
// (half the L2 -> CU key traffic per ciphertext), the twiddles of a pass are loaded once for four
// polynomials instead of two, LDS addresses are computed once, and a barrier separates twice the
// work.  Two workgroups per CU (four ciphertexts per CU, as with one ciphertext per workgroup), so
// the register budget is 256 per thread; each ciphertext keeps exchange buffers of its own.
// Arithmetic, index maps and results are those of k_blind_rotate_q2, bit for bit.
// An odd batch: the last workgroup computes its only ciphertext twice and stores it once.
// ------------------------------------------------------------------------------------------------
constexpr int G2P_NC = 2;
static constexpr size_t G2P_LDS_BYTES = G2P_NC * LDS_BYTES + 2 * G2_TW_SMALL * sizeof(double);

template <bool LOCKSTEP, bool FACTORED>
__global__ void __launch_bounds__(NT, 2) k_blind_rotate_q2_pair(Tables T, const double* __restrict__ bsk,
                                                               const u64* __restrict__ lwe_small, int n,
                                                               const u64* __restrict__ luts,
                                                               const u32* __restrict__ lut_index,
                                                               u64* __restrict__ glwe_out,
                                                               u64* __restrict__ lwe_out, BlockMap omap,
                                                               double inv_g, unsigned count) {
    constexpr int NC = G2P_NC;
    const unsigned t = threadIdx.x;
    const Prime P = T.pr[0];
    size_t ct[NC];
    const u64* __restrict__ lwe[NC];
    double x[NC][2][E];   // accumulator between steps, working residues inside a step
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const size_t want = (size_t)blockIdx.x * NC + c;
        ct[c] = want < count ? want : (size_t)count - 1;
        lwe[c] = lwe_small + ct[c] * (size_t)(n + 1);
        const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct[c] % omap.ops : ct[c]] : 0) * N;
        const unsigned bt = modswitch_2n(lwe[c][n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            bool neg;
            const double v = torus_to_qp(lut[rot_index(own1(t, j), a, neg)]);
            x[c][0][j] = 0.0;
            x[c][1][j] = neg ? -v : v;
        }
    }
    const double* __restrict__ twf_small = lds_dyn + NC * LDS_WORDS;   // entries 0..63 of the forward table
    const double* __restrict__ twi_small = twf_small + G2_TW_SMALL;
    for (unsigned i = t; i < (unsigned)G2_TW_SMALL; i += NT) {
        lds_dyn[NC * LDS_WORDS + i] = T.tw_fwd[0][i];
        lds_dyn[NC * LDS_WORDS + G2_TW_SMALL + i] = T.tw_inv[0][i];
    }
    __syncthreads();
    // The twiddles of passes 3 and 4 (26 values per thread, both directions) depend on the thread
    // alone: with 256 registers to spend they stay in registers for the whole blind rotation, which
    // leaves the key words as the step's only vector memory loads besides three monomial gathers.
    double w3f[7], w4f[7], w3i[7], w4i[7];
    load_tw8(T.tw_fwd[0], base3(t), w3f);
    load_tw4x2(T.tw_fwd[0], t, w4f);
    load_tw8(T.tw_inv[0], base3(t), w3i);
    load_tw4x2(T.tw_inv[0], t, w4i);
    w4f[6] = 0.0; w4i[6] = 0.0;

    // polynomial q of ciphertext c
    auto buf = [&](int c, int q) { return lds_dyn + (2 * c + q) * POLY_STRIDE; };
    const int groups = (n + 1) / 2;
    for (int g = 0; g < groups; ++g) {
        unsigned am[NC][2];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            am[c][0] = modswitch_2n(lwe[c][2 * g]);
            am[c][1] = 2 * g + 1 < n ? modswitch_2n(lwe[c][2 * g + 1]) : 0u;
        }
        // addresses are recomputed per phase instead of living in registers across the step loop
        unsigned tt = t, zero = 0;
        asm volatile("" : "+v"(tt));
        asm volatile("" : "+s"(zero));
        const double* __restrict__ mono = T.mono + zero;
        const double* __restrict__ rows = bsk + (size_t)4 * g * BSKQ_ROW;
        double w[7];
        // ---- forward transform of the digits, four polynomials per pass
        load_tw8(twf_small, 1, w);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int j = 0; j < E; ++j) x[c][q][j] = __builtin_rint(x[c][q][j] * inv_g);
                ct_radix8(x[c][q], w, P);
                st_o1p12(buf(c, q), tt, x[c][q]);
            }
        load_tw8(twf_small, base2(tt), w);
        __syncthreads();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o2p12(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ct_radix8(x[c][q], w, P);
        phase_sync<g2_sync_after<LOCKSTEP>(1)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) st_o2p23(buf(c, q), tt, x[c][q]);
        phase_sync<g2_sync_after<LOCKSTEP>(2)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o3p23(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ct_radix8(x[c][q], w3f, P);
        phase_sync<g2_sync_after<LOCKSTEP>(3)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) st_o3p34(buf(c, q), tt, x[c][q]);
        phase_sync<g2_sync_after<LOCKSTEP>(4)>();
        // ---- last forward pass, the step's GGSW formed point by point and applied, first inverse pass
        asm volatile("" : "+v"(tt));
        {
            const unsigned ut = 2 * rev8(tt) + 1;
            // psi^(a u_t), psi^(a' u_t): X^(a + a') is never formed, the step's GGSW is evaluated as
            // (K0 + X^a K1) + X^a' (K2 + X^a K3)
            double base[NC][2];
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int m = 0; m < 2; ++m) base[c][m] = ldt(mono, (am[c][m] * ut) & (2 * N - 1));
            // key words of unit u = 4 c4 + m: row m, the four polynomials q, positions 8 t + 2 c4 + {0, 1};
            // the next unit's four 16-byte loads are in flight while this one is multiplied into both
            // ciphertexts; the first unit's are requested here, ahead of the last forward pass
            double kc[4][2], kn[4][2];
#pragma unroll
            for (unsigned q = 0; q < 4; ++q) { const unsigned o = bsk_off(0, q, 0, tt); kc[q][0] = LFHE_LDK(rows, o); kc[q][1] = LFHE_LDK(rows, o + 1); }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) ld_o4p34(buf(c, q), tt, x[c][q]);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) ct_radix4x2_bounded(x[c][q], w4f, P);   // |d| < 3.4 p
            const double* __restrict__ w8 = mono + MONO_WORDS;
            auto chunk = [&](auto c4_tag) {
                constexpr int c4 = decltype(c4_tag)::value;
                double B[NC][4][2], S[NC][4][2], M1[NC][2], M2[NC][2];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    mono_pair<c4>(base[c][0], w8, am[c][0], P, M1[c][0], M1[c][1]);
                    mono_pair<c4>(base[c][1], w8, am[c][1], P, M2[c][0], M2[c][1]);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int u = 4 * c4 + m;
                    if (u < 15) {
                        const unsigned c4n = (unsigned)(u + 1) / 4, mn = (unsigned)(u + 1) % 4;
#pragma unroll
                        for (unsigned q = 0; q < 4; ++q) {
                            const unsigned o = mn * (unsigned)BSKQ_ROW + bsk_off(0, q, c4n, tt);
                            kn[q][0] = LFHE_LDK(rows, o); kn[q][1] = LFHE_LDK(rows, o + 1);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                ggsw_entry_row<FACTORED>(m, B[c][q][h], S[c][q][h], kc[q][h], M1[c][h], M2[c][h], P);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 4; ++q) { kc[q][0] = kn[q][0]; kc[q][1] = kn[q][1]; }
                }
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int j = 2 * c4 + h;
                        const double d0 = x[c][0][j], d1 = x[c][1][j];
                        x[c][0][j] = mmul(B[c][0][h], d0, P) + mmul(B[c][2][h], d1, P);
                        x[c][1][j] = mmul(B[c][1][h], d0, P) + mmul(B[c][3][h], d1, P);
                    }
            };
            chunk(std::integral_constant<int, 0>{});
            chunk(std::integral_constant<int, 1>{});
            chunk(std::integral_constant<int, 2>{});
            chunk(std::integral_constant<int, 3>{});
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) gs_radix4x2<GS_NORM_PASS4[0], GS_NORM_PASS4[1]>(x[c][q], w4i, P);
        }
        phase_sync<g2_sync_after<LOCKSTEP>(5)>();   // the next phase rewrites the words this thread has just read
        // ---- inverse transform
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) st_o4p34(buf(c, q), tt, x[c][q]);
        phase_sync<g2_sync_after<LOCKSTEP>(6)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o3p34(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) gs_radix8<GS_NORM_PASS3[0], GS_NORM_PASS3[1], GS_NORM_PASS3[2]>(x[c][q], w3i, P);
        phase_sync<g2_sync_after<LOCKSTEP>(7)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) st_o3p23(buf(c, q), tt, x[c][q]);
        load_tw8(twi_small, base2(tt), w);
        phase_sync<g2_sync_after<LOCKSTEP>(8)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o2p23(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) gs_radix8<GS_NORM_PASS2_GROUPED[0], GS_NORM_PASS2_GROUPED[1], GS_NORM_PASS2_GROUPED[2]>(x[c][q], w, P);
        phase_sync<g2_sync_after<LOCKSTEP>(9)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) st_o2p12(buf(c, q), tt, x[c][q]);
        load_tw8(twi_small, 1, w);
        __syncthreads();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o1p12(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                gs_radix8<GS_NORM_PASS1[0], GS_NORM_PASS1[1], GS_NORM_PASS1[2]>(x[c][q], w, P);
#pragma unroll
                for (int j = 0; j < E; ++j) x[c][q][j] = qp_canon(x[c][q][j], P);
            }
        __syncthreads();  // the last pass's reads are done before the next step overwrites the buffers
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c > 0 && (size_t)blockIdx.x * NC + c >= count) break;   // odd batch: the duplicate is not stored
        if (glwe_out) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const double v = x[c][q][j];
                    glwe_out[(ct[c] * 2 + q) * N + own1(t, j)] = v < 0 ? QP_U - (u64)(-v) : (u64)v;
                }
        }
        if (lwe_out) {  // sample extraction of coefficient 0, every word switched back to 2^64
            u64* o = lwe_out + omap.block(ct[c]) * (size_t)(N + 1);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const unsigned e = own1(t, j);
                if (e == 0) { o[0] = qp_to_torus(x[c][0][j]); o[N] = qp_to_torus(x[c][1][j]); }
                else o[N - e] = qp_to_torus(-x[c][0][j]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// THREE key bits per step (prime ring modulus).  Per triple of key bits (s, s', s'') the key holds eight
// GGSWs, GGSW m of the product over the bits of s_i (bit i of m set) or 1 - s_i, and a step is
//     acc <- (sum over m of X^(a [m&1] + a' [m&2] + a'' [m&4]) G_m) [.] acc :
// 306 steps of one external product instead of 459.  Point-wise the step's GGSW entry is
//     B = A0 + m3 A1,   A0 = K0 + m1 K1 + m2 K2 + m1 m2 K3,   A1 = K4 + m1 K5 + m2 K6 + m1 m2 K7
// with m1, m2, m3 the monomial factors of the three exponents at the position, and each A is held
// factored as d (m1 + a)(m2 + b) + c like the pairs' image: five modular products per entry, 24 per
// position with the external product itself (12 with pairs), against a third fewer transforms --
// 12 % fewer FP64 instructions per bootstrap, a third more key words.
//
// Key image (k_bsk_factor3 writes it): per group and per position e = 8 t + p the eight words of an
// entry q travel as four 16-byte pairs w = 0..3 = (a1, b1), (d1, c1), (a0, b0), (d0, c0):
//     [group][p 0..7][w 0..3][q 0..3][thread][2]       (doubles; 65536 per group)
// so that one unit of four loads (one per q) feeds one position and the per-entry state stays at two
// values (B and the running m3 A1) -- the registers of the pairs' kernel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double mod_inverse(double x, const Prime& P, u64 p);
constexpr int BSK3_GROUP = 8 * BSKQ_ROW;    // doubles per group of eight GGSWs
LFHE_HD unsigned bsk3_off(unsigned p, unsigned w, unsigned q, unsigned t) { return (((p * 4 + w) * 4 + q) * NT + t) * 2; }

// eight GGSW rows in the transform layout (src: [m][prime][q][c4][t][2], `src_row` doubles per GGSW: BSKQ_ROW
// with the prime ring modulus, BSK_ROW with both primes) -> the factored image of prime r
// (dst: [group][prime][p][w][q][t][2], `dst_group` doubles per group)
__global__ void __launch_bounds__(256) k_bsk_factor3(Tables T, const double* __restrict__ src, double* __restrict__ dst,
                                                     size_t groups, int* __restrict__ degenerate, unsigned r,
                                                     size_t src_row, size_t dst_group) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;     // (group, q, position e)
    if (id >= groups * 4 * N) return;
    const size_t g = id / (4 * N);
    const unsigned q = (unsigned)(id / N % 4), e = (unsigned)(id % N);
    const unsigned t = e >> 3, pp = e & 7;
    const Prime P = T.pr[r];
    const double* in = src + g * 8 * src_row + bsk_off(r, q, pp >> 1, t) + (pp & 1);
    double k[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) k[m] = mnorm(mnorm(in[(size_t)m * src_row], P), P);
    double out[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const double k0 = k[4 * half], k1 = k[4 * half + 1], k2 = k[4 * half + 2], k3 = k[4 * half + 3];
        if (k3 == 0.0) { atomicOr(degenerate, 1); return; }
        const double inv = mod_inverse(k3, P, r ? P1_U : P0_U);
        const double a = mnorm(mmul(k2, inv, P), P), b = mnorm(mmul(k1, inv, P), P);
        const double c = mnorm(k0 - mmul(k1, a, P), P);
        // pairs in load order: (a1, b1), (d1, c1), (a0, b0), (d0, c0)
        const int base = half == 1 ? 0 : 4;
        out[base] = a; out[base + 1] = b; out[base + 2] = k3; out[base + 3] = c;
    }
    double* o = dst + g * dst_group + (size_t)r * BSK3_GROUP;
#pragma unroll
    for (unsigned w = 0; w < 4; ++w) {
        const unsigned at = bsk3_off(pp, w, q, t);
        o[at] = out[2 * w];
        o[at + 1] = out[2 * w + 1];
    }
}

// primes: 1 (image of the prime ring modulus) or 2 (both primes of the 2^64 arithmetic, src rows of BSK_ROW)
void launch_bsk_factor3(const Tables& T, const double* src, double* dst, size_t groups, int* degenerate, int primes,
                        hipStream_t st) {
    const size_t words = groups * 4 * N;
    for (int r = 0; r < primes; ++r)
        hipLaunchKernelGGL(k_bsk_factor3, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, T, src, dst, groups, degenerate,
                           (unsigned)r, (size_t)(primes == 2 ? BSK_ROW : BSKQ_ROW), (size_t)primes * BSK3_GROUP);
}

// NC ciphertexts per workgroup (2: throughput, as k_blind_rotate_q2_pair; 1: launches that would leave
// compute units without a workgroup otherwise).  Transforms, exchanges and fences are the pairs' kernel's.
template <int NC>
__global__ void __launch_bounds__(NT, 2) k_blind_rotate_q3(Tables T, const double* __restrict__ bsk,
                                                          const u64* __restrict__ lwe_small, int n,
                                                          const u64* __restrict__ luts,
                                                          const u32* __restrict__ lut_index,
                                                          u64* __restrict__ glwe_out,
                                                          u64* __restrict__ lwe_out, BlockMap omap,
                                                          double inv_g, unsigned count) {
    constexpr bool LOCKSTEP = false;
    const unsigned t = threadIdx.x;
    const Prime P = T.pr[0];
    size_t ct[NC];
    const u64* __restrict__ lwe[NC];
    double x[NC][2][E];   // accumulator between steps, working residues inside a step
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const size_t want = (size_t)blockIdx.x * NC + c;
        ct[c] = want < count ? want : (size_t)count - 1;
        lwe[c] = lwe_small + ct[c] * (size_t)(n + 1);
        const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct[c] % omap.ops : ct[c]] : 0) * N;
        const unsigned bt = modswitch_2n(lwe[c][n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            bool neg;
            const double v = torus_to_qp(lut[rot_index(own1(t, j), a, neg)]);
            x[c][0][j] = 0.0;
            x[c][1][j] = neg ? -v : v;
        }
    }
    const double* __restrict__ twf_small = lds_dyn + NC * LDS_WORDS;   // entries 0..63 of the forward table
    const double* __restrict__ twi_small = twf_small + G2_TW_SMALL;
    for (unsigned i = t; i < (unsigned)G2_TW_SMALL; i += NT) {
        lds_dyn[NC * LDS_WORDS + i] = T.tw_fwd[0][i];
        lds_dyn[NC * LDS_WORDS + G2_TW_SMALL + i] = T.tw_inv[0][i];
    }
    __syncthreads();
    // twiddles of passes 3 and 4 stay in registers for the whole blind rotation (as in the pairs' kernel)
    double w3f[7], w4f[7], w3i[7], w4i[7];
    load_tw8(T.tw_fwd[0], base3(t), w3f);
    load_tw4x2(T.tw_fwd[0], t, w4f);
    load_tw8(T.tw_inv[0], base3(t), w3i);
    load_tw4x2(T.tw_inv[0], t, w4i);
    w4f[6] = 0.0; w4i[6] = 0.0;

    auto buf = [&](int c, int q) { return lds_dyn + (2 * c + q) * POLY_STRIDE; };
    const int groups = (n + 2) / 3;
    for (int g = 0; g < groups; ++g) {
        unsigned am[NC][3];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 3; ++i) am[c][i] = 3 * g + i < n ? modswitch_2n(lwe[c][3 * g + i]) : 0u;
        unsigned tt = t, zero = 0;
        asm volatile("" : "+v"(tt));
        asm volatile("" : "+s"(zero));
        const double* __restrict__ mono = T.mono + zero;
        const double* __restrict__ rows = bsk + (size_t)g * BSK3_GROUP;
        double w[7];
        // ---- forward transform of the digits
        load_tw8(twf_small, 1, w);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int j = 0; j < E; ++j) x[c][q][j] = __builtin_rint(x[c][q][j] * inv_g);
                ct_radix8(x[c][q], w, P);
                st_o1p12(buf(c, q), tt, x[c][q]);
            }
        load_tw8(twf_small, base2(tt), w);
        __syncthreads();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o2p12(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ct_radix8(x[c][q], w, P);
        phase_sync<g2_sync_after<LOCKSTEP>(1)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) st_o2p23(buf(c, q), tt, x[c][q]);
        phase_sync<g2_sync_after<LOCKSTEP>(2)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o3p23(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ct_radix8(x[c][q], w3f, P);
        phase_sync<g2_sync_after<LOCKSTEP>(3)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) st_o3p34(buf(c, q), tt, x[c][q]);
        phase_sync<g2_sync_after<LOCKSTEP>(4)>();
        // ---- last forward pass, the step's GGSW formed point by point and applied, first inverse pass
        asm volatile("" : "+v"(tt));
        {
            const unsigned ut = 2 * rev8(tt) + 1;
            double base[NC][3];
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int i = 0; i < 3; ++i) base[c][i] = ldt(mono, (am[c][i] * ut) & (2 * N - 1));
            // unit u = 4 p + w: position 8 t + p, word pair w, the four polynomials q; the next unit's four
            // 16-byte loads are in flight while this one is multiplied into the ciphertexts
            double kc[4][2], kn[4][2];
#pragma unroll
            for (unsigned q = 0; q < 4; ++q) { const unsigned o = bsk3_off(0, 0, q, tt); kc[q][0] = LFHE_LDK(rows, o); kc[q][1] = LFHE_LDK(rows, o + 1); }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) ld_o4p34(buf(c, q), tt, x[c][q]);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) ct_radix4x2_bounded(x[c][q], w4f, P);   // |d| < 3.4 p
            const double* __restrict__ w8 = mono + MONO_WORDS;
            auto chunk = [&](auto c4_tag) {
                constexpr int c4 = decltype(c4_tag)::value;
                double M[NC][3][2];
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int i = 0; i < 3; ++i) mono_pair<c4>(base[c][i], w8, am[c][i], P, M[c][i][0], M[c][i][1]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p = 2 * c4 + h;
                    double B[NC][4], Tm[NC][4];
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        const int u = 4 * p + wd;
                        if (u < 31) {
                            const unsigned pn = (unsigned)(u + 1) / 4, wn = (unsigned)(u + 1) % 4;
#pragma unroll
                            for (unsigned q = 0; q < 4; ++q) {
                                const unsigned o = bsk3_off(pn, wn, q, tt);
                                kn[q][0] = LFHE_LDK(rows, o); kn[q][1] = LFHE_LDK(rows, o + 1);
                            }
                        }
#pragma unroll
                        for (int c = 0; c < NC; ++c)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const double m1 = M[c][0][h], m2 = M[c][1][h], m3 = M[c][2][h];
                                if (wd == 0) Tm[c][q] = mmul(m1 + kc[q][0], m2 + kc[q][1], P);              // (m1 + a1)(m2 + b1)
                                else if (wd == 1) Tm[c][q] = mmul(mmul(Tm[c][q], kc[q][0], P) + kc[q][1], m3, P);   // m3 A1
                                else if (wd == 2) B[c][q] = mmul(m1 + kc[q][0], m2 + kc[q][1], P);          // (m1 + a0)(m2 + b0)
                                else B[c][q] = (mmul(B[c][q], kc[q][0], P) + kc[q][1]) + Tm[c][q];          // A0 + m3 A1, |B| < 1.8 p
                            }
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int q = 0; q < 4; ++q) { kc[q][0] = kn[q][0]; kc[q][1] = kn[q][1]; }
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const double d0 = x[c][0][p], d1 = x[c][1][p];
                        x[c][0][p] = mmul(B[c][0], d0, P) + mmul(B[c][2], d1, P);
                        x[c][1][p] = mmul(B[c][1], d0, P) + mmul(B[c][3], d1, P);
                    }
                }
            };
            chunk(std::integral_constant<int, 0>{});
            chunk(std::integral_constant<int, 1>{});
            chunk(std::integral_constant<int, 2>{});
            chunk(std::integral_constant<int, 3>{});
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) gs_radix4x2<GS_NORM_PASS4[0], GS_NORM_PASS4[1]>(x[c][q], w4i, P);
        }
        phase_sync<g2_sync_after<LOCKSTEP>(5)>();   // the next phase rewrites the words this thread has just read
        // ---- inverse transform
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) st_o4p34(buf(c, q), tt, x[c][q]);
        phase_sync<g2_sync_after<LOCKSTEP>(6)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o3p34(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) gs_radix8<GS_NORM_PASS3[0], GS_NORM_PASS3[1], GS_NORM_PASS3[2]>(x[c][q], w3i, P);
        phase_sync<g2_sync_after<LOCKSTEP>(7)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) st_o3p23(buf(c, q), tt, x[c][q]);
        load_tw8(twi_small, base2(tt), w);
        phase_sync<g2_sync_after<LOCKSTEP>(8)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o2p23(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) gs_radix8<GS_NORM_PASS2_GROUPED[0], GS_NORM_PASS2_GROUPED[1], GS_NORM_PASS2_GROUPED[2]>(x[c][q], w, P);
        phase_sync<g2_sync_after<LOCKSTEP>(9)>();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) st_o2p12(buf(c, q), tt, x[c][q]);
        load_tw8(twi_small, 1, w);
        __syncthreads();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o1p12(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                gs_radix8<GS_NORM_PASS1[0], GS_NORM_PASS1[1], GS_NORM_PASS1[2]>(x[c][q], w, P);
#pragma unroll
                for (int j = 0; j < E; ++j) x[c][q][j] = qp_canon(x[c][q][j], P);
            }
        __syncthreads();  // the last pass's reads are done before the next step overwrites the buffers
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c > 0 && (size_t)blockIdx.x * NC + c >= count) break;   // odd batch: the duplicate is not stored
        if (glwe_out) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const double v = x[c][q][j];
                    glwe_out[(ct[c] * 2 + q) * N + own1(t, j)] = v < 0 ? QP_U - (u64)(-v) : (u64)v;
                }
        }
        if (lwe_out) {  // sample extraction of coefficient 0, every word switched back to 2^64
            u64* o = lwe_out + omap.block(ct[c]) * (size_t)(N + 1);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const unsigned e = own1(t, j);
                if (e == 0) { o[0] = qp_to_torus(x[c][0][j]); o[N] = qp_to_torus(x[c][1][j]); }
                else o[N - e] = qp_to_torus(-x[c][0][j]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The grouped key over 2^64.  Four GGSWs per pair of key bits as above, but encryptions over the
// torus 2^64 itself: ciphertexts, accumulator and key stay in the published ring, the step
//     acc <- (G0 + X^a G1 + X^a' G2 + X^(a+a') G3) [.] acc
// runs once per prime and the two results are lifted to 2^64 -- half the transforms of the
// one-bit-per-step blind rotation over 2^64.  Two ciphertexts per workgroup share key words,
// twiddles and addresses as in k_blind_rotate_q2_pair; the first prime's result waits in registers
// while the second prime runs.
//
// Range of the lift.  A coefficient of the step's result is a sum of 2 N products of a digit
// (|d| <= 2^22) and a coefficient of the step's GGSW, which is a sum of four key words (< 2^65):
// the worst case is 2^99, above the +-2^97.6 the two primes tell apart (one GGSW alone stays below
// 2^97, which is why the one-bit step is exact without condition).  The key words are uniform: the
// sum has standard deviation 2^90.4 and the window ends at 149 standard deviations, a probability
// far below 2^-10000 per coefficient against the 2^-128 failure probability of the scheme itself.
// Inside the window the result is exact and equals the CPU restatement's bit for bit.
// ------------------------------------------------------------------------------------------------
constexpr int G64_TW_WORDS = 4 * G2_TW_SMALL;   // entries 0..63 of the forward and inverse tables of either prime
constexpr size_t g64_lds_bytes(int nc) { return (size_t)nc * LDS_BYTES + G64_TW_WORDS * sizeof(double); }

// One prime's share of a step for the NC (two, or one for launches that do not fill the GPU) ciphertexts of a workgroup: digits in x (pass-1
// ownership) -> residues of the external product in x (|x| < 2 p, pass-1 ownership).  Ends with the
// workgroup barrier that frees the exchange buffers.
// TERM: -1 the whole step's GGSW; 0..2^BITS - 1 one of its terms alone (the slow path, see below).
// BITS: key bits per step, 2 (rows: four GGSW rows of BSK_ROW words) or 3 (rows: the group's factored image
// of both primes as k_bsk_factor3 lays it out; the point-wise stage is k_blind_rotate_q3's).
// R < 0: the prime is picked at run time (r_rt, the same for every lane of a wavefront) -- the 512-thread kernel,
// whose halves each take one prime, with exchange buffers (lds_base) of their own.
template <int R, bool FACTORED, int NC, int TERM = -1, int BITS = 2>
__device__ __forceinline__ void g64_prime_pass(double (&x)[NC][2][E], const Tables& T,
                                               const double* __restrict__ rows, const unsigned (&am)[NC][BITS],
                                               unsigned t, double* __restrict__ lds_base = lds_dyn,
                                               const double* __restrict__ tw_small = lds_dyn + NC * LDS_WORDS,
                                               unsigned r_rt = 0) {
    static_assert(BITS == 2 || (BITS == 3 && FACTORED), "three key bits per step come with the factored image only");
    const unsigned r = R >= 0 ? (unsigned)R : r_rt;
    const Prime P = r ? T.pr[1] : T.pr[0];
    const double* __restrict__ twf_small = tw_small + 2 * r * G2_TW_SMALL;
    const double* __restrict__ twi_small = twf_small + G2_TW_SMALL;
    auto buf = [&](int c, int q) { return lds_base + (2 * c + q) * POLY_STRIDE; };
    // addresses are recomputed per phase instead of living in registers across the step loop
    unsigned tt = t, zero = 0;
    asm volatile("" : "+v"(tt));
    asm volatile("" : "+s"(zero));
    const double* __restrict__ mono = (r == 0 ? T.mono : T.mono1) + zero;
    const double* __restrict__ twf = (r ? T.tw_fwd[1] : T.tw_fwd[0]) + zero;
    const double* __restrict__ twi = (r ? T.tw_inv[1] : T.tw_inv[0]) + zero;
    double w[7];
    // ---- forward transform of the digits, four polynomials per pass
    load_tw8(twf_small, 1, w);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ct_radix8(x[c][q], w, P);
            st_o1p12(buf(c, q), tt, x[c][q]);
        }
    load_tw8(twf_small, base2(tt), w);
    __syncthreads();
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) ld_o2p12(buf(c, q), tt, x[c][q]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) ct_radix8(x[c][q], w, P);
    phase_sync<sync_after(1)>();
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) st_o2p23(buf(c, q), tt, x[c][q]);
    load_tw8(twf, base3(tt), w);   // the twiddles of a pass are requested one hand-over ahead
    phase_sync<sync_after(2)>();
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) ld_o3p23(buf(c, q), tt, x[c][q]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) ct_radix8(x[c][q], w, P);
    phase_sync<sync_after(3)>();
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) st_o3p34(buf(c, q), tt, x[c][q]);
    load_tw4x2(twf, tt, w);
    phase_sync<sync_after(4)>();
    // ---- last forward pass, the step's GGSW formed point by point and applied, first inverse pass
    asm volatile("" : "+v"(tt));
    {
        const unsigned ut = 2 * rev8(tt) + 1;
        double base[NC][BITS];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int m = 0; m < BITS; ++m) base[c][m] = ldt(mono, (am[c][m] * ut) & (2 * N - 1));
        // key words of unit u = 4 c4 + m: GGSW m of the group, this prime, the four polynomials q,
        // positions 8 t + 2 c4 + {0, 1}; the next unit's loads are in flight while this one is multiplied
        // (three bits: unit u = 4 p + w, position 8 t + p, word pair w of this prime's factored image)
        const double* __restrict__ rows3 = rows + (BITS == 3 ? r * BSK3_GROUP : 0);
        double kc[4][2], kn[4][2];
#pragma unroll
        for (unsigned q = 0; q < 4; ++q) {
            const unsigned o = BITS == 3 ? bsk3_off(0, 0, q, tt) : bsk_off(r, q, 0, tt);
            kc[q][0] = LFHE_LDK(rows3, o); kc[q][1] = LFHE_LDK(rows3, o + 1);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ld_o4p34(buf(c, q), tt, x[c][q]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) ct_radix4x2_bounded(x[c][q], w, P);   // |d| < 3.4 p
        load_tw4x2(twi, tt, w);   // for the inverse pass below: in flight during the point-wise stage
        const double* __restrict__ w8 = mono + MONO_WORDS;
        auto chunk = [&](auto c4_tag) {
            constexpr int c4 = decltype(c4_tag)::value;
            if constexpr (BITS == 3) {
                double M[NC][3][2];
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int i = 0; i < 3; ++i) mono_pair<c4>(base[c][i], w8, am[c][i], P, M[c][i][0], M[c][i][1]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p = 2 * c4 + h;
                    double B[NC][4], Tm[NC][4];
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        const int u = 4 * p + wd;
                        if (u < 31) {
                            const unsigned pn = (unsigned)(u + 1) / 4, wn = (unsigned)(u + 1) % 4;
#pragma unroll
                            for (unsigned q = 0; q < 4; ++q) {
                                const unsigned o = bsk3_off(pn, wn, q, tt);
                                kn[q][0] = LFHE_LDK(rows3, o); kn[q][1] = LFHE_LDK(rows3, o + 1);
                            }
                        }
#pragma unroll
                        for (int c = 0; c < NC; ++c)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const double m1 = M[c][0][h], m2 = M[c][1][h], m3 = M[c][2][h];
                                if constexpr (TERM < 0) {
                                    if (wd == 0) Tm[c][q] = mmul(m1 + kc[q][0], m2 + kc[q][1], P);              // (m1 + a1)(m2 + b1)
                                    else if (wd == 1) Tm[c][q] = mmul(mmul(Tm[c][q], kc[q][0], P) + kc[q][1], m3, P);   // m3 A1
                                    else if (wd == 2) B[c][q] = mmul(m1 + kc[q][0], m2 + kc[q][1], P);          // (m1 + a0)(m2 + b0)
                                    else B[c][q] = (mmul(B[c][q], kc[q][0], P) + kc[q][1]) + Tm[c][q];          // A0 + m3 A1, |B| < 1.8 p
                                } else {
                                    ggsw3_term_row<TERM>(wd, B[c][q], Tm[c][q], kc[q][0], kc[q][1], m1, m2, m3, P);
                                }
                            }
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int q = 0; q < 4; ++q) { kc[q][0] = kn[q][0]; kc[q][1] = kn[q][1]; }
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const double d0 = x[c][0][p], d1 = x[c][1][p];
                        x[c][0][p] = mmul(B[c][0], d0, P) + mmul(B[c][2], d1, P);
                        x[c][1][p] = mmul(B[c][1], d0, P) + mmul(B[c][3], d1, P);
                    }
                }
            } else {
            double B[NC][4][2], S[NC][4][2], M1[NC][2], M2[NC][2];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                mono_pair<c4>(base[c][0], w8, am[c][0], P, M1[c][0], M1[c][1]);
                mono_pair<c4>(base[c][1], w8, am[c][1], P, M2[c][0], M2[c][1]);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int u = 4 * c4 + m;
                if (u < 15) {
                    const unsigned c4n = (unsigned)(u + 1) / 4, mn = (unsigned)(u + 1) % 4;
#pragma unroll
                    for (unsigned q = 0; q < 4; ++q) {
                        const unsigned o = mn * (unsigned)BSK_ROW + bsk_off(r, q, c4n, tt);
                        kn[q][0] = LFHE_LDK(rows, o); kn[q][1] = LFHE_LDK(rows, o + 1);
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if constexpr (TERM < 0) ggsw_entry_row<FACTORED>(m, B[c][q][h], S[c][q][h], kc[q][h], M1[c][h], M2[c][h], P);
                            else ggsw_term_row<FACTORED, TERM>(m, B[c][q][h], kc[q][h], M1[c][h], M2[c][h], P);
                        }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) { kc[q][0] = kn[q][0]; kc[q][1] = kn[q][1]; }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 2 * c4 + h;
                    const double d0 = x[c][0][j], d1 = x[c][1][j];
                    x[c][0][j] = mmul(B[c][0][h], d0, P) + mmul(B[c][2][h], d1, P);
                    x[c][1][j] = mmul(B[c][1][h], d0, P) + mmul(B[c][3][h], d1, P);
                }
            }
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) gs_radix4x2<GS_NORM_PASS4[0], GS_NORM_PASS4[1]>(x[c][q], w, P);
    }
    phase_sync<sync_after(5)>();
    // ---- inverse transform
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) st_o4p34(buf(c, q), tt, x[c][q]);
    load_tw8(twi, base3(tt), w);
    phase_sync<sync_after(6)>();
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) ld_o3p34(buf(c, q), tt, x[c][q]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) gs_radix8<GS_NORM_PASS3[0], GS_NORM_PASS3[1], GS_NORM_PASS3[2]>(x[c][q], w, P);
    phase_sync<sync_after(7)>();
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) st_o3p23(buf(c, q), tt, x[c][q]);
    load_tw8(twi_small, base2(tt), w);
    phase_sync<sync_after(8)>();
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) ld_o2p23(buf(c, q), tt, x[c][q]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) gs_radix8<GS_NORM_PASS2_GROUPED[0], GS_NORM_PASS2_GROUPED[1], GS_NORM_PASS2_GROUPED[2]>(x[c][q], w, P);
    phase_sync<sync_after(9)>();
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) st_o2p12(buf(c, q), tt, x[c][q]);
    load_tw8(twi_small, 1, w);
    __syncthreads();
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) ld_o1p12(buf(c, q), tt, x[c][q]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) gs_radix8<GS_NORM_PASS1[0], GS_NORM_PASS1[1], GS_NORM_PASS1[2]>(x[c][q], w, P);
    __syncthreads();  // the last pass's reads are done before the buffers are written again
}

// Every lift of the fast path is looked at (crt_wrap_checked): a ciphertext with a coefficient in the
// guard band at the end of the lift's window is marked in `fault`, and the SAFE instantiation -- launched
// right behind the fast one over the same ciphertexts, its workgroups leaving at once unless their
// ciphertext is marked -- bootstraps it again with the step taken apart into its four terms
//     acc <- G0 [.] acc + G1 [.] X^a acc + G2 [.] X^a' acc + G3 [.] X^(a+a') acc ,
// the same polynomial before the reduction to 2^64, each term the product with ONE GGSW (below 2^97:
// exact without condition), summed in wrapping 64-bit words: the same bits as an exact evaluation of
// the fast path's formula, at four times the transforms.  The order on the stream does the rest;
// the host is not involved.  `fallbacks` counts the ciphertexts that took the slow path.
// BITS = 3: three key bits per step (eight terms; the group's image is k_bsk_factor3's for both primes).
template <bool FACTORED, int NC, bool SAFE = false, int BITS = 2>
__global__ void __launch_bounds__(NT, 2) k_blind_rotate_g64_pair(Tables T, const double* __restrict__ bsk,
                                                                const u64* __restrict__ lwe_small, int n,
                                                                const u64* __restrict__ luts,
                                                                const u32* __restrict__ lut_index,
                                                                u64* __restrict__ glwe_out,
                                                                u64* __restrict__ lwe_out, BlockMap omap,
                                                                unsigned count, int* __restrict__ fault,
                                                                double k_guard, unsigned long long* __restrict__ fallbacks) {
    static_assert(!SAFE || NC == 1, "the slow path takes one ciphertext per workgroup");
    const unsigned t = threadIdx.x;
    if constexpr (SAFE) {
        if (!fault[blockIdx.x]) return;   // the same for every thread of the workgroup
        if (t == 0) atomicAdd(fallbacks, 1ull);
    }
    size_t ct[NC];
    const u64* __restrict__ lwe[NC];
    u64 acc[NC][2][E];
    bool near_edge[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        near_edge[c] = false;
        const size_t want = (size_t)blockIdx.x * NC + c;
        ct[c] = want < count ? want : (size_t)count - 1;
        lwe[c] = lwe_small + ct[c] * (size_t)(n + 1);
        const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct[c] % omap.ops : ct[c]] : 0) * N;
        const unsigned bt = modswitch_2n(lwe[c][n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            bool neg;
            const u64 v = lut[rot_index(own1(t, j), a, neg)];
            acc[c][0][j] = 0;
            acc[c][1][j] = neg ? (u64)0 - v : v;
        }
    }
    for (unsigned i = t; i < (unsigned)G64_TW_WORDS; i += NT) {
        const unsigned r = i / (2 * G2_TW_SMALL), k = i % (2 * G2_TW_SMALL);
        lds_dyn[NC * LDS_WORDS + i] = k < (unsigned)G2_TW_SMALL ? T.tw_fwd[r][k] : T.tw_inv[r][k - G2_TW_SMALL];
    }
    __syncthreads();

    const int groups = (n + BITS - 1) / BITS;
    for (int g = 0; g < groups; ++g) {
        unsigned am[NC][BITS];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < BITS; ++i) am[c][i] = BITS * g + i < n ? modswitch_2n(lwe[c][BITS * g + i]) : 0u;
        const double* __restrict__ rows = bsk + (size_t)g * (BITS == 3 ? 2 * BSK3_GROUP : 4 * BSK_ROW);
        int dig[NC][2][E];
        double x[NC][2][E], y0[NC][2][E];
        auto digits_to_x = [&]() {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < E; ++j) x[c][q][j] = (double)dig[c][q][j];
        };
        auto keep_prime0 = [&]() {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < E; ++j) y0[c][q][j] = x[c][q][j];
        };
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) dig[c][q][j] = decomp23(acc[c][q][j]);
        if constexpr (!SAFE) {
            digits_to_x();
            g64_prime_pass<0, FACTORED, NC, -1, BITS>(x, T, rows, am, t);
            keep_prime0();
            digits_to_x();
            g64_prime_pass<1, FACTORED, NC, -1, BITS>(x, T, rows, am, t);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < E; ++j)
                        acc[c][q][j] = crt_wrap_checked(y0[c][q][j], x[c][q][j], T.p0inv_mod_p1, T.pr[1], k_guard, near_edge[c]);
        } else {
            u64 sum[2][E];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) sum[q][j] = 0;
            auto term = [&](auto term_tag) {
                constexpr int TERM = decltype(term_tag)::value;
                digits_to_x();
                g64_prime_pass<0, FACTORED, NC, TERM, BITS>(x, T, rows, am, t);
                keep_prime0();
                digits_to_x();
                g64_prime_pass<1, FACTORED, NC, TERM, BITS>(x, T, rows, am, t);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < E; ++j) sum[q][j] += crt_wrap(y0[0][q][j], x[0][q][j], T.p0inv_mod_p1, T.pr[1]);
            };
            term(std::integral_constant<int, 0>{});
            term(std::integral_constant<int, 1>{});
            term(std::integral_constant<int, 2>{});
            term(std::integral_constant<int, 3>{});
            if constexpr (BITS == 3) {
                term(std::integral_constant<int, 4>{});
                term(std::integral_constant<int, 5>{});
                term(std::integral_constant<int, 6>{});
                term(std::integral_constant<int, 7>{});
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) acc[0][q][j] = sum[q][j];
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c > 0 && (size_t)blockIdx.x * NC + c >= count) break;   // odd batch: the duplicate is not stored
        if constexpr (!SAFE) {
            // any thread of the workgroup that saw a lift near the edge marks the ciphertext (plain vector store)
            if (near_edge[c]) fault[ct[c]] = 1;
        }
        if (glwe_out) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) glwe_out[(ct[c] * 2 + q) * N + own1(t, j)] = acc[c][q][j];
        }
        if (lwe_out) {  // sample extraction of coefficient 0
            u64* o = lwe_out + omap.block(ct[c]) * (size_t)(N + 1);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const unsigned e = own1(t, j);
                if (e == 0) { o[0] = acc[c][0][j]; o[N] = acc[c][1][j]; }
                else o[N - e] = (u64)0 - acc[c][0][j];
            }
        }
    }
}

// The grouped key over 2^64 for launches of up to one ciphertext per CU: 512 threads per ciphertext, one half
// per PRIME.  The two primes' shares of a step are independent until the lift, so each half runs the whole
// one-prime pass (digits -> transform -> the step's GGSW -> inverse) with exchange buffers of its own, at the
// same time as the other; then thread t of either half hands its 16 residues to thread t of the other through
// the (now idle) exchange buffers and both lift all 16 coefficients -- the same words in both halves, so the
// accumulator, the digits and the guard's verdict never have to be exchanged.  Bit-identical to
// k_blind_rotate_g64_pair; a ciphertext the guard marks goes to that kernel's slow path like any other.
constexpr size_t G64_WIDE_LDS_BYTES = 2 * LDS_BYTES + G64_TW_WORDS * sizeof(double);

template <bool FACTORED, int BITS>
__global__ void __launch_bounds__(2 * NT, 1) k_blind_rotate_g64_wide(Tables T, const double* __restrict__ bsk,
                                                                     const u64* __restrict__ lwe_small, int n,
                                                                     const u64* __restrict__ luts,
                                                                     const u32* __restrict__ lut_index,
                                                                     u64* __restrict__ glwe_out,
                                                                     u64* __restrict__ lwe_out, BlockMap omap,
                                                                     int* __restrict__ fault, double k_guard) {
    const unsigned h = threadIdx.x >> 8;   // prime of this half (the same for every lane of a wavefront)
    const unsigned t = threadIdx.x & (NT - 1);
    const size_t ct = blockIdx.x;
    const u64* __restrict__ lwe = lwe_small + ct * (size_t)(n + 1);
    const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct % omap.ops : ct] : 0) * N;
    double* __restrict__ mine = lds_dyn + h * LDS_WORDS;
    const double* __restrict__ theirs = lds_dyn + (1 - h) * LDS_WORDS;
    const double* __restrict__ tw_small = lds_dyn + 2 * LDS_WORDS;
    u64 acc[1][2][E];
    bool near_edge = false;
    {
        const unsigned bt = modswitch_2n(lwe[n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            bool neg;
            const u64 v = lut[rot_index(own1(t, j), a, neg)];
            acc[0][0][j] = 0;
            acc[0][1][j] = neg ? (u64)0 - v : v;
        }
    }
    for (unsigned i = threadIdx.x; i < (unsigned)G64_TW_WORDS; i += 2 * NT) {
        const unsigned r = i / (2 * G2_TW_SMALL), k = i % (2 * G2_TW_SMALL);
        lds_dyn[2 * LDS_WORDS + i] = k < (unsigned)G2_TW_SMALL ? T.tw_fwd[r][k] : T.tw_inv[r][k - G2_TW_SMALL];
    }
    __syncthreads();

    const int groups = (n + BITS - 1) / BITS;
    for (int g = 0; g < groups; ++g) {
        unsigned am[1][BITS];
#pragma unroll
        for (int i = 0; i < BITS; ++i) am[0][i] = BITS * g + i < n ? modswitch_2n(lwe[BITS * g + i]) : 0u;
        const double* __restrict__ rows = bsk + (size_t)g * (BITS == 3 ? 2 * BSK3_GROUP : 4 * BSK_ROW);
        double x[1][2][E];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) x[0][q][j] = (double)decomp23(acc[0][q][j]);
        g64_prime_pass<-1, FACTORED, 1, -1, BITS>(x, T, rows, am, t, mine, tw_small, h);   // ends with a workgroup barrier
        // residues of this half's prime to the partner thread: word (q, j) of thread t at [q][j][t]
        unsigned tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) mine[q * POLY_STRIDE + j * NT + tt] = x[0][q][j];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const double other = theirs[q * POLY_STRIDE + j * NT + tt];
                const double y0 = h ? other : x[0][q][j], y1 = h ? x[0][q][j] : other;
                acc[0][q][j] = crt_wrap_checked(y0, y1, T.p0inv_mod_p1, T.pr[1], k_guard, near_edge);
            }
        __syncthreads();   // the partner has read before the next step's first pass writes the buffers
    }
    if (h == 0) {
        if (near_edge) fault[ct] = 1;   // any thread that saw a lift near the edge marks the ciphertext (plain vector store)
        if (glwe_out) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) glwe_out[(ct * 2 + q) * N + own1(t, j)] = acc[0][q][j];
        }
        if (lwe_out) {  // sample extraction of coefficient 0
            u64* o = lwe_out + omap.block(ct) * (size_t)(N + 1);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const unsigned e = own1(t, j);
                if (e == 0) { o[0] = acc[0][0][j]; o[N] = acc[0][1][j]; }
                else o[N - e] = (u64)0 - acc[0][0][j];
            }
        }
    }
}

// The grouped key over 2^64 for launches of at most one ciphertext per TWO CUs: the primes of a ciphertext on two
// compute units of one XCD.  Each of the two workgroups (256 threads; blocks b and b + 8 land on the same XCD)
// runs the one-prime pass of a step with the CU to itself and hands its 4096 residues to the other through the
// XCD's L2 once per step: write-through stores, drained per wave, a workgroup barrier, one lane raises the
// workgroup's step counter and polls the partner's, a barrier, loads that bypass the L1.  Payload slots alternate
// with the step, so a workgroup that is one step ahead does not overwrite what its partner still reads (it cannot
// be two ahead: it waits for the partner's counter of the step, raised after the partner has read the step before).
// Both workgroups lift all coefficients and so hold the same accumulator.  The poll is bounded: a workgroup whose
// partner does not show up (not resident because something else holds the CUs) marks the ciphertext in the guard's
// flags and leaves, its partner does the same in its turn, and the slow path that follows every launch of this key
// bootstraps the ciphertext -- no wave ever waits without end.  Bit-identical to the other kernels of the key.
typedef double v2d_t __attribute__((ext_vector_type(2)));
constexpr int G64_SPLIT_WORDS = 2 * N;                      // residues one workgroup hands over per step (both polynomials)
constexpr size_t g64_split_payload_words(size_t cts) { return cts * 2 * 2 * (size_t)G64_SPLIT_WORDS; }   // [ct][half][slot]
constexpr size_t g64_split_flag_words(size_t cts) { return cts * 2 * 32; }                                // one 128-byte line each

__device__ __forceinline__ void st8_l2(v2d_t* p, const v2d_t (&v)[8]) {
    // eight 16-byte write-through stores (lane-contiguous, NT lanes apart), drained once before the wave goes on
    v2d_t *p1 = p + NT, *p2 = p + 2 * NT, *p3 = p + 3 * NT, *p4 = p + 4 * NT, *p5 = p + 5 * NT, *p6 = p + 6 * NT, *p7 = p + 7 * NT;
    asm volatile("global_store_dwordx4 %0, %8, off sc1\n"
                 "global_store_dwordx4 %1, %9, off sc1\n"
                 "global_store_dwordx4 %2, %10, off sc1\n"
                 "global_store_dwordx4 %3, %11, off sc1\n"
                 "global_store_dwordx4 %4, %12, off sc1\n"
                 "global_store_dwordx4 %5, %13, off sc1\n"
                 "global_store_dwordx4 %6, %14, off sc1\n"
                 "global_store_dwordx4 %7, %15, off sc1\n"
                 "s_waitcnt vmcnt(0)"
                 :: "v"(p), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7),
                    "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]) : "memory");
}
__device__ __forceinline__ void ld8_l2(const v2d_t* p, v2d_t (&v)[8]) {
    const v2d_t *p1 = p + NT, *p2 = p + 2 * NT, *p3 = p + 3 * NT, *p4 = p + 4 * NT, *p5 = p + 5 * NT, *p6 = p + 6 * NT, *p7 = p + 7 * NT;
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n"
                 "global_load_dwordx4 %1, %9, off sc1\n"
                 "global_load_dwordx4 %2, %10, off sc1\n"
                 "global_load_dwordx4 %3, %11, off sc1\n"
                 "global_load_dwordx4 %4, %12, off sc1\n"
                 "global_load_dwordx4 %5, %13, off sc1\n"
                 "global_load_dwordx4 %6, %14, off sc1\n"
                 "global_load_dwordx4 %7, %15, off sc1\n"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7) : "memory");
}
__device__ __forceinline__ void flag_store_l2(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc1\ns_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned flag_load_l2(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\ns_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <bool FACTORED, int BITS>
__global__ void __launch_bounds__(NT, 2) k_blind_rotate_g64_split(Tables T, const double* __restrict__ bsk,
                                                                 const u64* __restrict__ lwe_small, int n,
                                                                 const u64* __restrict__ luts,
                                                                 const u32* __restrict__ lut_index,
                                                                 u64* __restrict__ glwe_out,
                                                                 u64* __restrict__ lwe_out, BlockMap omap, unsigned count,
                                                                 int* __restrict__ fault, double k_guard,
                                                                 double* __restrict__ payload, unsigned* __restrict__ flags,
                                                                 unsigned poll_limit) {
    const unsigned b = blockIdx.x;
    const unsigned h = (b >> 3) & 1u;                      // prime of this workgroup
    const size_t ct = (size_t)(b >> 4) * 8 + (b & 7u);     // blocks b and b + 8: one ciphertext, one XCD
    if (ct >= count) return;                               // the same for the whole workgroup and for its partner
    const unsigned t = threadIdx.x;
    const u64* __restrict__ lwe = lwe_small + ct * (size_t)(n + 1);
    const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct % omap.ops : ct] : 0) * N;
    v2d_t* mine = reinterpret_cast<v2d_t*>(payload + (ct * 2 + h) * 2 * (size_t)G64_SPLIT_WORDS);
    const v2d_t* theirs = reinterpret_cast<const v2d_t*>(payload + (ct * 2 + (1 - h)) * 2 * (size_t)G64_SPLIT_WORDS);
    unsigned* my_flag = flags + (ct * 2 + h) * 32;
    const unsigned* their_flag = flags + (ct * 2 + (1 - h)) * 32;
    __shared__ unsigned s_dead;
    if (t == 0) s_dead = 0;
    u64 acc[1][2][E];
    bool near_edge = false;
    {
        const unsigned bt = modswitch_2n(lwe[n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            bool neg;
            const u64 v = lut[rot_index(own1(t, j), a, neg)];
            acc[0][0][j] = 0;
            acc[0][1][j] = neg ? (u64)0 - v : v;
        }
    }
    for (unsigned i = t; i < (unsigned)G64_TW_WORDS; i += NT) {
        const unsigned r = i / (2 * G2_TW_SMALL), k = i % (2 * G2_TW_SMALL);
        lds_dyn[LDS_WORDS + i] = k < (unsigned)G2_TW_SMALL ? T.tw_fwd[r][k] : T.tw_inv[r][k - G2_TW_SMALL];
    }
    __syncthreads();

    const int groups = (n + BITS - 1) / BITS;
    for (int g = 0; g < groups; ++g) {
        unsigned am[1][BITS];
#pragma unroll
        for (int i = 0; i < BITS; ++i) am[0][i] = BITS * g + i < n ? modswitch_2n(lwe[BITS * g + i]) : 0u;
        const double* __restrict__ rows = bsk + (size_t)g * (BITS == 3 ? 2 * BSK3_GROUP : 4 * BSK_ROW);
        double x[1][2][E];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) x[0][q][j] = (double)decomp23(acc[0][q][j]);
        g64_prime_pass<-1, FACTORED, 1, -1, BITS>(x, T, rows, am, t, lds_dyn, lds_dyn + LDS_WORDS, h);
        // hand-over: 16-byte word k = 4 q + j / 2 of thread t at [slot][k][t]
        const unsigned step = (unsigned)g + 1u;
        const unsigned slot = step & 1u;
        unsigned tt = t;
        asm volatile("" : "+v"(tt));
        {
            v2d_t v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[4 * q + k].x = x[0][q][2 * k]; v[4 * q + k].y = x[0][q][2 * k + 1]; }
            st8_l2(mine + (size_t)slot * (G64_SPLIT_WORDS / 2) + tt, v);   // returns when this wave's stores have left
        }
        __syncthreads();                                   // ... and every wave's have
        if (t == 0) {
            flag_store_l2(my_flag, step);
            unsigned tries = 0;
            while (flag_load_l2(their_flag) < step) {
                __builtin_amdgcn_s_sleep(1);
                if (++tries > poll_limit) { s_dead = 1; break; }
            }
        }
        __syncthreads();                                   // the poll has matched before anyone loads
        if (s_dead) {                                      // the same for every thread: the slow path takes this ciphertext
            if (t == 0) fault[ct] = 1;
            return;
        }
        {
            v2d_t u[8];
            ld8_l2(theirs + (size_t)slot * (G64_SPLIT_WORDS / 2) + tt, u);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * k + e;
                        const double other = e ? u[4 * q + k].y : u[4 * q + k].x;
                        const double y0 = h ? other : x[0][q][j], y1 = h ? x[0][q][j] : other;
                        acc[0][q][j] = crt_wrap_checked(y0, y1, T.p0inv_mod_p1, T.pr[1], k_guard, near_edge);
                    }
        }
    }
    if (h == 0) {
        if (near_edge) fault[ct] = 1;   // any thread that saw a lift near the edge marks the ciphertext (plain vector store)
        if (glwe_out) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) glwe_out[(ct * 2 + q) * N + own1(t, j)] = acc[0][q][j];
        }
        if (lwe_out) {  // sample extraction of coefficient 0
            u64* o = lwe_out + omap.block(ct) * (size_t)(N + 1);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const unsigned e = own1(t, j);
                if (e == 0) { o[0] = acc[0][0][j]; o[N] = acc[0][1][j]; }
                else o[N - e] = (u64)0 - acc[0][0][j];
            }
        }
    }
}

// The grouped blind rotation for launches too small to fill the GPU: 512 threads per ciphertext, one
// half per polynomial as in k_blind_rotate_q_wide.  Each half forms the two entries of the step's
// GGSW that feed its own output polynomial.  Results are bit-identical to k_blind_rotate_q2.
// The inverse transform has exchange buffers of its own, which leaves three workgroup barriers per
// step: after pass 1 of the forward transform has been published, after the transformed digits
// have been published for the other half, and before pass 1 of the inverse transform reads.  Every
// later write into a buffer is separated from the last cross-wave read of it by one of the three.
template <bool FACTORED>
__global__ void __launch_bounds__(2 * NT, 1) k_blind_rotate_q2_wide(Tables T, const double* __restrict__ bsk,
                                                                    const u64* __restrict__ lwe_small, int n,
                                                                    const u64* __restrict__ luts,
                                                                    const u32* __restrict__ lut_index,
                                                                    u64* __restrict__ glwe_out,
                                                                    u64* __restrict__ lwe_out, BlockMap omap,
                                                                    double inv_g) {
    const unsigned h = threadIdx.x >> 8;   // polynomial of this half (wave-uniform)
    const unsigned t0 = threadIdx.x & (NT - 1);
    const size_t ct = blockIdx.x;
    const Prime P = T.pr[0];
    const u64* __restrict__ lwe = lwe_small + ct * (size_t)(n + 1);
    const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct % omap.ops : ct] : 0) * N;
    double* mine = lds_dyn + h * POLY_STRIDE;
    double* theirs = lds_dyn + (1 - h) * POLY_STRIDE;
    double* back = lds_dyn + WIDE2_BACK + h * POLY_STRIDE;
    const double* __restrict__ twf = lds_dyn + WIDE_TWF;   // passes 2-4; pass 1's live in scalar registers
    const double* __restrict__ twi = lds_dyn + WIDE_TWI;
    const double* __restrict__ mono = lds_dyn + WIDE_MONO;
    wide_fill_tables(T, true);

    double acc[E], x[E], w[7], w1f[7], w1i[7];
    load_tw8_uniform(T.tw_fwd[0], w1f);
    load_tw8_uniform(T.tw_inv[0], w1i);
    {
        const unsigned bt = modswitch_2n(lwe[n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            bool neg;
            const double v = torus_to_qp(lut[rot_index(own1(t0, j), a, neg)]);
            acc[j] = h ? (neg ? -v : v) : 0.0;
        }
    }
    __syncthreads();

    const int groups = (n + 1) / 2;
    // Key words of a step (four rows): kx multiplies this half's own digits, ky the other half's;
    // polynomial index q = j_in * 2 + c_out.  The rows of step g + 1 are requested right after the
    // point-wise stage of step g has used up the registers.
    double kx[4][E], ky[4][E];
    auto load_key = [&](int g, unsigned t) {
        const double* __restrict__ rows = bsk + (size_t)4 * g * BSKQ_ROW;
#pragma unroll
        for (unsigned m = 0; m < 4; ++m)
#pragma unroll
            for (unsigned c4 = 0; c4 < 4; ++c4) {
                const unsigned o0 = m * (unsigned)BSKQ_ROW + bsk_off(0, 3 * h, c4, t);
                const unsigned o1 = m * (unsigned)BSKQ_ROW + bsk_off(0, 2 - h, c4, t);
#pragma unroll
                for (unsigned k = 0; k < 2; ++k) { kx[m][2 * c4 + k] = ldt(rows, o0 + k); ky[m][2 * c4 + k] = ldt(rows, o1 + k); }
            }
    };
    // the two exponents of step g (every step runs, also one whose exponents are both zero: the control
    // flow does not depend on the ciphertext)
    auto exponent0 = [&](int g) { return g < groups ? modswitch_2n(lwe[2 * g]) : 0u; };
    auto exponent1 = [&](int g) { return 2 * g + 1 < n ? modswitch_2n(lwe[2 * g + 1]) : 0u; };
    int g = 0;
    unsigned a0 = exponent0(0), a1 = exponent1(0);
    if (g < groups) load_key(g, t0);

    while (g < groups) {
        // exponents of the following step: requested now, looked at after the point-wise stage
        unsigned na0 = exponent0(g + 1), na1 = exponent1(g + 1);
        unsigned t = t0;
        asm volatile("" : "+v"(t));  // addresses are recomputed per step instead of living in registers
#pragma unroll
        for (int j = 0; j < E; ++j) x[j] = __builtin_rint(acc[j] * inv_g);
        ct_radix8(x, w1f, P);
        st_o1p12(mine, t, x); load_tw8(twf, base2(t), w);
        __syncthreads();
        ld_o2p12(mine, t, x); ct_radix8(x, w, P);
        wave_sync();
        st_o2p23(mine, t, x); load_tw8(twf, base3(t), w);
        wave_sync();
        ld_o3p23(mine, t, x); ct_radix8(x, w, P);
        wave_sync();
        st_o3p34(mine, t, x); load_tw4x2(twf, t, w);
        wave_sync();
        ld_o4p34(mine, t, x); ct_radix4x2_bounded(x, w, P);   // |d| < 3.4 p for the wide operand of the point-wise stage
        // publish the transformed digits (a thread rewrites the words it has just read) and fetch
        // the other polynomial's
        st_o4p34(mine, t, x); load_tw4x2(twi, t, w);
        __syncthreads();
        {
            double y[E];
            ld_o4p34(theirs, t, y);
            const unsigned ut = 2 * rev8(t) + 1;
            const unsigned am[2] = {a0, a1};
            const double* __restrict__ w8 = mono + MONO_WORDS;
            double base[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) base[m] = ldt(mono, (am[m] * ut) & (2 * N - 1));
            double o[E];
            auto chunk = [&](auto c4_tag) {
                constexpr int c4 = decltype(c4_tag)::value;
                double bx[2], by[2], sx[2], sy[2], M1[2], M2[2];
                mono_pair<c4>(base[0], w8, am[0], P, M1[0], M1[1]);
                mono_pair<c4>(base[1], w8, am[1], P, M2[0], M2[1]);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        ggsw_entry_row<FACTORED>(m, bx[k], sx[k], kx[m][2 * c4 + k], M1[k], M2[k], P);
                        ggsw_entry_row<FACTORED>(m, by[k], sy[k], ky[m][2 * c4 + k], M1[k], M2[k], P);
                    }
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    o[2 * c4 + k] = mmul(bx[k], x[2 * c4 + k], P) + mmul(by[k], y[2 * c4 + k], P);
            };
            chunk(std::integral_constant<int, 0>{});
            chunk(std::integral_constant<int, 1>{});
            chunk(std::integral_constant<int, 2>{});
            chunk(std::integral_constant<int, 3>{});
#pragma unroll
            for (int j = 0; j < E; ++j) x[j] = o[j];
        }
        // the next step's loads must stay below the last use of this step's words (same registers):
        // their addresses are made to depend on the point-wise results
        unsigned tk = t;
        asm volatile("" : "+v"(tk) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
        const int gnext = g + 1;
        if (gnext < groups) load_key(gnext, tk);
        gs_radix4x2<GS_NORM_PASS4[0], GS_NORM_PASS4[1]>(x, w, P);
        st_o4p34(back, t, x); load_tw8(twi, base3(t), w);
        wave_sync();
        ld_o3p34(back, t, x); gs_radix8<GS_NORM_PASS3[0], GS_NORM_PASS3[1], GS_NORM_PASS3[2]>(x, w, P);
        wave_sync();
        st_o3p23(back, t, x); load_tw8(twi, base2(t), w);
        wave_sync();
        ld_o2p23(back, t, x); gs_radix8<GS_NORM_PASS2_GROUPED[0], GS_NORM_PASS2_GROUPED[1], GS_NORM_PASS2_GROUPED[2]>(x, w, P);
        wave_sync();
        st_o2p12(back, t, x);
        __syncthreads();
        ld_o1p12(back, t, x); gs_radix8<GS_NORM_PASS1[0], GS_NORM_PASS1[1], GS_NORM_PASS1[2]>(x, w1i, P);
#pragma unroll
        for (int j = 0; j < E; ++j) acc[j] = qp_canon(x[j], P);
        g = gnext; a0 = na0; a1 = na1;
    }

    if (glwe_out) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double c = acc[j];
            glwe_out[(ct * 2 + h) * N + own1(t0, j)] = c < 0 ? QP_U - (u64)(-c) : (u64)c;
        }
    }
    if (lwe_out) {  // sample extraction of coefficient 0, every word switched back to 2^64
        u64* o = lwe_out + omap.block(ct) * (size_t)(N + 1);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const unsigned e = own1(t0, j);
            if (h == 0) o[e == 0 ? 0 : N - e] = qp_to_torus(e == 0 ? acc[j] : -acc[j]);
            else if (e == 0) o[N] = qp_to_torus(acc[j]);
        }
    }
}

// Three key bits per step for launches too small to fill the GPU: 512 threads per ciphertext, one half
// per polynomial, as k_blind_rotate_q2_wide.  A step's key words for one thread are 128 doubles, twice
// what the registers can hold a step ahead: four positions' worth (64 doubles) are kept.  Positions
// 0..3 of step g + 1 are requested as the point-wise stage of step g releases the registers (a whole
// transform of cover), positions 4..7 of a step as its own positions 0..3 are consumed (three
// positions of arithmetic of cover).  The loads are pinned behind the last use of the words they
// replace by making their addresses depend on the point-wise result.
__global__ void __launch_bounds__(2 * NT, 1) k_blind_rotate_q3_wide(Tables T, const double* __restrict__ bsk,
                                                                    const u64* __restrict__ lwe_small, int n,
                                                                    const u64* __restrict__ luts,
                                                                    const u32* __restrict__ lut_index,
                                                                    u64* __restrict__ glwe_out,
                                                                    u64* __restrict__ lwe_out, BlockMap omap,
                                                                    double inv_g) {
    const unsigned h = threadIdx.x >> 8;   // polynomial of this half (wave-uniform)
    const unsigned t0 = threadIdx.x & (NT - 1);
    const size_t ct = blockIdx.x;
    const Prime P = T.pr[0];
    const u64* __restrict__ lwe = lwe_small + ct * (size_t)(n + 1);
    const u64* __restrict__ lut = luts + (size_t)(lut_index ? lut_index[omap.ops ? ct % omap.ops : ct] : 0) * N;
    double* mine = lds_dyn + h * POLY_STRIDE;
    double* theirs = lds_dyn + (1 - h) * POLY_STRIDE;
    double* back = lds_dyn + WIDE2_BACK + h * POLY_STRIDE;
    const double* __restrict__ twf = lds_dyn + WIDE_TWF;   // passes 2-4; pass 1's live in scalar registers
    const double* __restrict__ twi = lds_dyn + WIDE_TWI;
    const double* __restrict__ mono = lds_dyn + WIDE_MONO;
    wide_fill_tables(T, true);

    double acc[E], x[E], w[7], w1f[7], w1i[7];
    load_tw8_uniform(T.tw_fwd[0], w1f);
    load_tw8_uniform(T.tw_inv[0], w1i);
    {
        const unsigned bt = modswitch_2n(lwe[n]);
        const unsigned a = (2 * N - bt) & (2 * N - 1);  // multiply by X^-b~
#pragma unroll
        for (int j = 0; j < E; ++j) {
            bool neg;
            const double v = torus_to_qp(lut[rot_index(own1(t0, j), a, neg)]);
            acc[j] = h ? (neg ? -v : v) : 0.0;
        }
    }
    __syncthreads();

    const int groups = (n + 2) / 3;
    // key words in flight or in use: slot s holds position s or s + 4; [w][entry: 0 own digits, 1 the other half's][2]
    double kw[4][4][2][2];
    auto load_pos = [&](int g, int p, int slot, unsigned t) {
        const double* __restrict__ rows = bsk + (size_t)g * BSK3_GROUP;
#pragma unroll
        for (unsigned wd = 0; wd < 4; ++wd) {
            const unsigned ox = bsk3_off((unsigned)p, wd, 3 * h, t), oy = bsk3_off((unsigned)p, wd, 2 - h, t);
#pragma unroll
            for (unsigned k = 0; k < 2; ++k) { kw[slot][wd][0][k] = ldt(rows, ox + k); kw[slot][wd][1][k] = ldt(rows, oy + k); }
        }
    };
    auto exponent = [&](int g, int i) { return 3 * g + i < n ? modswitch_2n(lwe[3 * g + i]) : 0u; };
    int g = 0;
    unsigned a0 = exponent(0, 0), a1 = exponent(0, 1), a2 = exponent(0, 2);
#pragma unroll
    for (int p = 0; p < 4; ++p) load_pos(0, p, p, t0);

    while (g < groups) {
        // exponents of the following step: requested now, looked at after the point-wise stage
        unsigned na0 = exponent(g + 1, 0), na1 = exponent(g + 1, 1), na2 = exponent(g + 1, 2);
        unsigned t = t0;
        asm volatile("" : "+v"(t));  // addresses are recomputed per step instead of living in registers
#pragma unroll
        for (int j = 0; j < E; ++j) x[j] = __builtin_rint(acc[j] * inv_g);
        ct_radix8(x, w1f, P);
        st_o1p12(mine, t, x); load_tw8(twf, base2(t), w);
        __syncthreads();
        ld_o2p12(mine, t, x); ct_radix8(x, w, P);
        wave_sync();
        st_o2p23(mine, t, x); load_tw8(twf, base3(t), w);
        wave_sync();
        ld_o3p23(mine, t, x); ct_radix8(x, w, P);
        wave_sync();
        st_o3p34(mine, t, x); load_tw4x2(twf, t, w);
        wave_sync();
        ld_o4p34(mine, t, x); ct_radix4x2_bounded(x, w, P);   // |d| < 3.4 p for the wide operand of the point-wise stage
        st_o4p34(mine, t, x); load_tw4x2(twi, t, w);
        __syncthreads();
        {
            double y[E];
            ld_o4p34(theirs, t, y);
            const unsigned ut = 2 * rev8(t) + 1;
            const unsigned am[3] = {a0, a1, a2};
            const double* __restrict__ w8 = mono + MONO_WORDS;
            double base[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) base[m] = ldt(mono, (am[m] * ut) & (2 * N - 1));
            const int gnext = g + 1;
            auto chunk = [&](auto c4_tag) {
                constexpr int c4 = decltype(c4_tag)::value;
                double M[3][2];
#pragma unroll
                for (int i = 0; i < 3; ++i) mono_pair<c4>(base[i], w8, am[i], P, M[i][0], M[i][1]);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int p = 2 * c4 + k, slot = p & 3;
                    double b[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        // B = A0 + m3 A1 from the pairs (a1, b1), (d1, c1), (a0, b0), (d0, c0); |B| < 1.8 p
                        double tm = mmul(M[0][k] + kw[slot][0][e][0], M[1][k] + kw[slot][0][e][1], P);
                        tm = mmul(mmul(tm, kw[slot][1][e][0], P) + kw[slot][1][e][1], M[2][k], P);
                        const double a = mmul(M[0][k] + kw[slot][2][e][0], M[1][k] + kw[slot][2][e][1], P);
                        b[e] = (mmul(a, kw[slot][3][e][0], P) + kw[slot][3][e][1]) + tm;
                    }
                    x[p] = mmul(b[0], x[p], P) + mmul(b[1], y[p], P);
                    // this slot's words are used up: request the ones that take their place
                    unsigned tk = t;
                    asm volatile("" : "+v"(tk) : "v"(x[p]));
                    if (p < 4) load_pos(g, p + 4, slot, tk);
                    else if (gnext < groups) load_pos(gnext, p - 4, slot, tk);
                }
            };
            chunk(std::integral_constant<int, 0>{});
            chunk(std::integral_constant<int, 1>{});
            chunk(std::integral_constant<int, 2>{});
            chunk(std::integral_constant<int, 3>{});
        }
        gs_radix4x2<GS_NORM_PASS4[0], GS_NORM_PASS4[1]>(x, w, P);
        st_o4p34(back, t, x); load_tw8(twi, base3(t), w);
        wave_sync();
        ld_o3p34(back, t, x); gs_radix8<GS_NORM_PASS3[0], GS_NORM_PASS3[1], GS_NORM_PASS3[2]>(x, w, P);
        wave_sync();
        st_o3p23(back, t, x); load_tw8(twi, base2(t), w);
        wave_sync();
        ld_o2p23(back, t, x); gs_radix8<GS_NORM_PASS2_GROUPED[0], GS_NORM_PASS2_GROUPED[1], GS_NORM_PASS2_GROUPED[2]>(x, w, P);
        wave_sync();
        st_o2p12(back, t, x);
        __syncthreads();
        ld_o1p12(back, t, x); gs_radix8<GS_NORM_PASS1[0], GS_NORM_PASS1[1], GS_NORM_PASS1[2]>(x, w1i, P);
#pragma unroll
        for (int j = 0; j < E; ++j) acc[j] = qp_canon(x[j], P);
        g = g + 1; a0 = na0; a1 = na1; a2 = na2;
    }

    if (glwe_out) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double c = acc[j];
            glwe_out[(ct * 2 + h) * N + own1(t0, j)] = c < 0 ? QP_U - (u64)(-c) : (u64)c;
        }
    }
    if (lwe_out) {  // sample extraction of coefficient 0, every word switched back to 2^64
        u64* o = lwe_out + omap.block(ct) * (size_t)(N + 1);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const unsigned e = own1(t0, j);
            if (h == 0) o[e == 0 ? 0 : N - e] = qp_to_torus(e == 0 ? acc[j] : -acc[j]);
            else if (e == 0) o[N] = qp_to_torus(acc[j]);
        }
    }
}

void launch_bskq_prepare(const Tables& T, const u64* in, double* out, size_t polys, hipStream_t st) {
    hipLaunchKernelGGL(k_bskq_prepare, dim3((unsigned)polys), dim3(NT), LDS_BYTES, st, T, in, out);
}
void launch_external_product_q(const Tables& T, const double* bsk, const u32* idx, const u64* glwe, u64* out,
                               size_t count, double inv_g, hipStream_t st) {
    hipLaunchKernelGGL(k_external_product_q, dim3((unsigned)count), dim3(NT), LDS_BYTES, st, T, bsk, idx, glwe, out,
                       inv_g);
}
// ------------------------------------------------------------------------------------------------
// Factored form of a grouped key image (pbs_phases.h: ggsw_entry_row).  In place, per group of four
// rows and per word:  (K0, K1, K2, K3) -> (a, b, d, c) = (K2/K3, K1/K3, K3, K0 - K1 K2 / K3)  mod q'.
// The rows keep their layout and their N^-1 factor (the ratios do not carry it, d and c do).  A word
// whose K3 is zero has no factored form: the flag is raised and the caller rebuilds the plain image.
// The grouped key over 2^64 is factored the same way, residue by residue of either prime.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double mod_inverse(double x, const Prime& P, u64 p) {   // x canonical, non-zero: x^(p - 2)
    double r = 1.0;
#pragma unroll 1
    for (int bit = 49; bit >= 0; --bit) {
        r = mnorm(mmul(r, r, P), P);
        if ((p - 2) >> bit & 1) r = mnorm(mmul(r, x, P), P);
    }
    return r;
}

// row_words: BSKQ_ROW (single modulus) or BSK_ROW (both primes: the first half of a row belongs to prime 0)
__global__ void __launch_bounds__(256) k_bsk_factor(Tables T, double* __restrict__ img, size_t groups, size_t row_words,
                                                    int* __restrict__ degenerate) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= groups * row_words) return;
    const size_t e = id % row_words;
    const int r = e >= (size_t)BSKQ_ROW ? 1 : 0;
    const Prime P = T.pr[r];
    double* w = img + (id / row_words) * 4 * row_words + e;
    const double k0 = w[0], k1 = w[row_words], k2 = w[2 * row_words];
    const double k3 = mnorm(mnorm(w[3 * row_words], P), P);
    if (k3 == 0.0) { atomicOr(degenerate, 1); return; }
    const double inv = mod_inverse(k3, P, r ? P1_U : P0_U);
    const double a = mnorm(mmul(k2, inv, P), P), b = mnorm(mmul(k1, inv, P), P);
    const double c = mnorm(k0 - mmul(k1, a, P), P);   // K1 K2 / K3 = K1 a
    w[0] = a;
    w[row_words] = b;
    w[2 * row_words] = k3;
    w[3 * row_words] = c;
}

void launch_bsk_factor(const Tables& T, double* img, size_t groups, size_t row_words, int* degenerate, hipStream_t st) {
    const size_t words = groups * row_words;
    hipLaunchKernelGGL(k_bsk_factor, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, T, img, groups, row_words,
                       degenerate);
}

// more than 64 KiB of dynamic LDS has to be asked for, once per device the kernels run on
template <class K>
static bool ask_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes) == hipSuccess;
}
void blind_rotate_setup(int dev) {
    static std::atomic<uint64_t> lds_asked{0};
    if (dev < 0 || dev >= 64) throw std::runtime_error("device index out of range");
    if (lds_asked.load(std::memory_order_acquire) >> dev & 1) return;
    if (hipSetDevice(dev) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
    auto need = [](bool ok, const char* kernel, size_t bytes) {
        if (!ok)
            throw std::runtime_error(std::string(kernel) + " needs " + std::to_string(bytes) +
                                     " bytes of LDS per workgroup and the device refused the request");
    };
    need(ask_lds(k_blind_rotate_q2_pair<true, false>, G2P_LDS_BYTES), "k_blind_rotate_q2_pair", G2P_LDS_BYTES);
    need(ask_lds(k_blind_rotate_q2_pair<false, false>, G2P_LDS_BYTES), "k_blind_rotate_q2_pair", G2P_LDS_BYTES);
    need(ask_lds(k_blind_rotate_q2_pair<true, true>, G2P_LDS_BYTES), "k_blind_rotate_q2_pair", G2P_LDS_BYTES);
    need(ask_lds(k_blind_rotate_q2_pair<false, true>, G2P_LDS_BYTES), "k_blind_rotate_q2_pair", G2P_LDS_BYTES);
    need(ask_lds(k_blind_rotate_q3<2>, G2P_LDS_BYTES), "k_blind_rotate_q3", G2P_LDS_BYTES);
    need(ask_lds(k_blind_rotate_q3_wide, WIDE2_LDS_BYTES), "k_blind_rotate_q3_wide", WIDE2_LDS_BYTES);
    need(ask_lds(k_blind_rotate_g64_pair<false, 2>, g64_lds_bytes(2)), "k_blind_rotate_g64_pair", g64_lds_bytes(2));
    need(ask_lds(k_blind_rotate_g64_pair<true, 2>, g64_lds_bytes(2)), "k_blind_rotate_g64_pair", g64_lds_bytes(2));
    need(ask_lds(k_blind_rotate_g64_pair<true, 1, true>, g64_lds_bytes(1)), "k_blind_rotate_g64_pair (slow path)", g64_lds_bytes(1));
    need(ask_lds(k_blind_rotate_g64_pair<false, 1, true>, g64_lds_bytes(1)), "k_blind_rotate_g64_pair (slow path)", g64_lds_bytes(1));
    need(ask_lds(k_blind_rotate_g64_pair<true, 2, false, 3>, g64_lds_bytes(2)), "k_blind_rotate_g64_pair (three key bits)", g64_lds_bytes(2));
    need(ask_lds(k_blind_rotate_g64_pair<true, 1, true, 3>, g64_lds_bytes(1)), "k_blind_rotate_g64_pair (three key bits, slow path)", g64_lds_bytes(1));
    need(ask_lds(k_blind_rotate_g64_split<true, 3>, g64_lds_bytes(1)), "k_blind_rotate_g64_split", g64_lds_bytes(1));
    need(ask_lds(k_blind_rotate_g64_split<true, 2>, g64_lds_bytes(1)), "k_blind_rotate_g64_split", g64_lds_bytes(1));
    need(ask_lds(k_blind_rotate_g64_split<false, 2>, g64_lds_bytes(1)), "k_blind_rotate_g64_split", g64_lds_bytes(1));
    need(ask_lds(k_blind_rotate_g64_wide<true, 3>, G64_WIDE_LDS_BYTES), "k_blind_rotate_g64_wide", G64_WIDE_LDS_BYTES);
    need(ask_lds(k_blind_rotate_g64_wide<true, 2>, G64_WIDE_LDS_BYTES), "k_blind_rotate_g64_wide", G64_WIDE_LDS_BYTES);
    need(ask_lds(k_blind_rotate_g64_wide<false, 2>, G64_WIDE_LDS_BYTES), "k_blind_rotate_g64_wide", G64_WIDE_LDS_BYTES);
    need(ask_lds(k_blind_rotate_q2_wide<false>, WIDE2_LDS_BYTES), "k_blind_rotate_q2_wide", WIDE2_LDS_BYTES);
    need(ask_lds(k_blind_rotate_q2_wide<true>, WIDE2_LDS_BYTES), "k_blind_rotate_q2_wide", WIDE2_LDS_BYTES);
    need(ask_lds(k_blind_rotate_q_wide, WIDE_LDS_BYTES), "k_blind_rotate_q_wide", WIDE_LDS_BYTES);
    lds_asked.fetch_or(1ull << dev, std::memory_order_release);
}

// grouping: 1, 2 (four GGSWs per pair of key bits as generated) or 3 (the same in factored form)
template <bool FACTORED>
static void launch_grouped(const Tables& T, const double* bsk, const u64* lwe_small, int n, const u64* luts,
                           const u32* lut_index, u64* glwe_out, u64* lwe_out, size_t count, const BlockMap& omap,
                           double inv_g, int wide_below, int lockstep_above, int pair_above, hipStream_t st) {
    if (count <= (size_t)wide_below) {
        hipLaunchKernelGGL(k_blind_rotate_q2_wide<FACTORED>, dim3((unsigned)count), dim3(2 * NT), WIDE2_LDS_BYTES, st, T, bsk,
                           lwe_small, n, luts, lut_index, glwe_out, lwe_out, omap, inv_g);
        return;
    }
    if (pair_above >= 0 && count > (size_t)pair_above) {
        const unsigned wgs = (unsigned)((count + G2P_NC - 1) / G2P_NC);
        // with two ciphertexts per workgroup the wave-level fences win at every launch size (4096: 42.7
        // against 43.5 ms with a workgroup barrier after every phase); the lock-step instantiation stays for A/B runs
        static const bool pair_lockstep = std::getenv("LFHE_PAIR_LOCKSTEP") != nullptr;
        if (pair_lockstep && count > (size_t)lockstep_above)
            hipLaunchKernelGGL((k_blind_rotate_q2_pair<true, FACTORED>), dim3(wgs), dim3(NT), G2P_LDS_BYTES, st, T, bsk,
                               lwe_small, n, luts, lut_index, glwe_out, lwe_out, omap, inv_g, (unsigned)count);
        else
            hipLaunchKernelGGL((k_blind_rotate_q2_pair<false, FACTORED>), dim3(wgs), dim3(NT), G2P_LDS_BYTES, st, T, bsk,
                               lwe_small, n, luts, lut_index, glwe_out, lwe_out, omap, inv_g, (unsigned)count);
        return;
    }
    if (count > (size_t)lockstep_above)
        hipLaunchKernelGGL((k_blind_rotate_q2<true, FACTORED>), dim3((unsigned)count), dim3(NT), G2_LDS_BYTES, st, T, bsk,
                           lwe_small, n, luts, lut_index, glwe_out, lwe_out, omap, inv_g);
    else
        hipLaunchKernelGGL((k_blind_rotate_q2<false, FACTORED>), dim3((unsigned)count), dim3(NT), G2_LDS_BYTES, st, T, bsk,
                           lwe_small, n, luts, lut_index, glwe_out, lwe_out, omap, inv_g);
}

void launch_blind_rotate_q(const Tables& T, const double* bsk, const u64* lwe_small, int n, const u64* luts,
                           const u32* lut_index, u64* glwe_out, u64* lwe_out, size_t count,
                           const BlockMap& omap, double inv_g, int wide_below, int lockstep_above, int grouping,
                           int pair_above, hipStream_t st) {
    if (count <= (size_t)wide_below) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) throw std::runtime_error("hipGetDevice failed");
        blind_rotate_setup(dev);   // a no-op after the engine's own call at start-up
    }
    if (grouping == 4) {   // three key bits per step: 512 threads per ciphertext up to one per CU, else two ciphertexts per
                           // workgroup, or one when that would leave CUs idle
        if (count <= (size_t)wide_below)
            hipLaunchKernelGGL(k_blind_rotate_q3_wide, dim3((unsigned)count), dim3(2 * NT), WIDE2_LDS_BYTES, st, T, bsk, lwe_small,
                               n, luts, lut_index, glwe_out, lwe_out, omap, inv_g);
        else if (pair_above >= 0 && count > (size_t)pair_above)
            hipLaunchKernelGGL(k_blind_rotate_q3<2>, dim3((unsigned)((count + 1) / 2)), dim3(NT), G2P_LDS_BYTES, st, T, bsk,
                               lwe_small, n, luts, lut_index, glwe_out, lwe_out, omap, inv_g, (unsigned)count);
        else
            hipLaunchKernelGGL(k_blind_rotate_q3<1>, dim3((unsigned)count), dim3(NT), G2_LDS_BYTES, st, T, bsk, lwe_small, n,
                               luts, lut_index, glwe_out, lwe_out, omap, inv_g, (unsigned)count);
        return;
    }
    if (grouping == 3) {
        launch_grouped<true>(T, bsk, lwe_small, n, luts, lut_index, glwe_out, lwe_out, count, omap, inv_g, wide_below,
                             lockstep_above, pair_above, st);
        return;
    }
    if (grouping == 2) {
        launch_grouped<false>(T, bsk, lwe_small, n, luts, lut_index, glwe_out, lwe_out, count, omap, inv_g, wide_below,
                              lockstep_above, pair_above, st);
        return;
    }
    // up to two ciphertexts per CU: latency matters more than throughput, use 512 threads each
    if (count <= (size_t)wide_below)
        hipLaunchKernelGGL(k_blind_rotate_q_wide, dim3((unsigned)count), dim3(2 * NT), WIDE_LDS_BYTES, st, T, bsk, lwe_small,
                           n, luts, lut_index, glwe_out, lwe_out, omap, inv_g);
    else
        hipLaunchKernelGGL(k_blind_rotate_q, dim3((unsigned)count), dim3(NT), LDS_BYTES, st, T, bsk, lwe_small, n, luts,
                           lut_index, glwe_out, lwe_out, omap, inv_g);
}

// ------------------------------------------------------------------------------------------------
// LWE key switch, big (N) -> small (n):  out = (0,...,0,b) - sum_i sum_lv d[i][lv] * KSK[i][lv].
// Digits are shifted to u = d + B/2 >= 0 so the inner product is unsigned; the constant
// (B/2) * sum KSK is precomputed per column (colsum).  One thread per output column (coalesced
// key reads), KS_CT ciphertexts per workgroup share every key word read.
// ------------------------------------------------------------------------------------------------
constexpr int KS_CT = 16;      // ciphertexts per workgroup: every key word read serves this many


__device__ __forceinline__ u32 ks_digits_packed(u64 x, int base_log, int levels) {
    // shifted digits u = d + B/2, one per byte, level lv in byte lv (lv = 0 most significant)
    const int rep = base_log * levels;
    u64 state = ((x >> (64 - rep - 1)) + 1) >> 1;
    state &= (1ull << rep) - 1;
    const u64 mask = (1ull << base_log) - 1;
    u32 packed = 0;
    for (int lv = levels - 1; lv >= 0; --lv) {
        u64 o = state & mask;
        state >>= base_log;
        u64 carry = (((o - 1) | state) & o) >> (base_log - 1);
        state += carry;
        const int d = (int)o - (int)(carry << base_log);
        packed |= (u32)(d + (1 << (base_log - 1))) << (8 * lv);
    }
    return packed;
}

__global__ void __launch_bounds__(256) k_ksk_colsum(const u64* __restrict__ ksk, int rows, int cols,
                                                    int half_base, u64* __restrict__ colsum) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols) return;
    u64 s = 0;
    for (int r = 0; r < rows; ++r) s += ksk[(size_t)r * cols + col];
    colsum[col] = s * (u64)half_base;
}

// Stage 1: the shifted digits of every input coefficient, packed one per byte, laid out
// [coefficient][group of KS_CT ciphertexts][KS_CT] so that stage 2 fetches the sixteen words of a
// (coefficient, group) pair with one scalar load.
__global__ void __launch_bounds__(256) k_ks_digits(const u64* __restrict__ in, u32* __restrict__ dig, int count,
                                                   int big, int base_log, int levels, BlockMap imap) {
    const int groups = (count + KS_CT - 1) / KS_CT;
    const size_t total = (size_t)groups * KS_CT * big;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(k % big);            // consecutive threads read consecutive coefficients
        const int ct = (int)(k / big);
        u32 pk = 0;
        if (ct < count) pk = ks_digits_packed(in[imap.block(ct) * (size_t)(big + 1) + i], base_log, levels);
        dig[((size_t)i * groups + ct / KS_CT) * KS_CT + ct % KS_CT] = pk;
    }
}

// Stage 2: one thread per output column, KS_CT ciphertexts per workgroup.  The digits are
// wave-uniform, so they live in scalar registers and their extraction runs on the scalar unit;
// the vector unit is left with two 32x32->64 multiply-adds per 64-bit product (low and high key
// half, recombined at the end).  gridDim.z > 1 splits the input coefficients between workgroups
// (small batches would otherwise run a few workgroups through 8192 key rows each): every slice
// then adds its share into `out`, which the launcher has zeroed, with 64-bit atomics; wrapping
// sums do not depend on the order.
template <int LEVELS>
__global__ void __launch_bounds__(256) k_keyswitch(const u64* __restrict__ ksk,
                                                   const u64* __restrict__ colsum,
                                                   const u64* __restrict__ in, const u32* __restrict__ dig,
                                                   u64* __restrict__ out, int count, int big, int n,
                                                   BlockMap imap) {
    const int cols = n + 1;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int colc = col < cols ? col : cols - 1;   // lanes past the last column repeat it and drop the result
    const int ct0 = blockIdx.y * KS_CT;
    const int groups = gridDim.y;
    const int per = (big + gridDim.z - 1) / gridDim.z;
    const int i_begin = blockIdx.z * per;
    const int i_end = min(big, i_begin + per);
    u64 lo[KS_CT], hi[KS_CT];
#pragma unroll
    for (int c = 0; c < KS_CT; ++c) { lo[c] = 0; hi[c] = 0; }
    // key words of the next coefficient are in flight while this one is multiplied
    u64 kn[LEVELS];
    if (i_begin < i_end) {
#pragma unroll
        for (int lv = 0; lv < LEVELS; ++lv) kn[lv] = ksk[((size_t)i_begin * LEVELS + lv) * cols + colc];
    }
    for (int i = i_begin; i < i_end; ++i) {
        u64 k[LEVELS];
#pragma unroll
        for (int lv = 0; lv < LEVELS; ++lv) k[lv] = kn[lv];
        if (i + 1 < i_end) {
#pragma unroll
            for (int lv = 0; lv < LEVELS; ++lv) kn[lv] = ksk[((size_t)(i + 1) * LEVELS + lv) * cols + colc];
        }
        const u32* __restrict__ d = dig + ((size_t)i * groups + blockIdx.y) * KS_CT;
        u32 w[KS_CT];
#pragma unroll
        for (int c = 0; c < KS_CT; ++c) w[c] = d[c];
#pragma unroll
        for (int lv = 0; lv < LEVELS; ++lv) {
            const u32 klo = (u32)k[lv], khi = (u32)(k[lv] >> 32);
#pragma unroll
            for (int c = 0; c < KS_CT; ++c) {
                const u32 u = (w[c] >> (8 * lv)) & 0xffu;
                lo[c] += (u64)u * klo;
                hi[c] += (u64)u * khi;
            }
        }
    }
    if (col < cols) {
        // the shifted digits overcount by (B/2) * sum of all key rows: added once, by slice 0
        const u64 cs = blockIdx.z == 0 ? colsum[col] : 0;
#pragma unroll
        for (int c = 0; c < KS_CT; ++c) {
            const int ct = ct0 + c;
            if (ct < count) {
                const u64 b = (col == n && blockIdx.z == 0) ? in[imap.block(ct) * (size_t)(big + 1) + big] : 0;
                const u64 v = b - (lo[c] + (hi[c] << 32)) + cs;
                if (gridDim.z == 1) out[(size_t)ct * cols + col] = v;
                else atomicAdd(reinterpret_cast<unsigned long long*>(out + (size_t)ct * cols + col), (unsigned long long)v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out = ca * a + cb * b + (add_body on the body word), count ciphertexts of dim1 words each.
// b may be null.  Everything wraps mod 2^64.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lwe_linear(const u64* __restrict__ a, const u64* __restrict__ b,
                                                    u64* __restrict__ out, size_t total, int dim1, u64 ca, u64 cb,
                                                    u64 add_body) {
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) {
        u64 v = ca * a[k];
        if (b) v += cb * b[k];
        if ((int)(k % (size_t)dim1) == dim1 - 1) v += add_body;
        out[k] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

void launch_ntt_forward(const Tables& T, const u64* in, double* out, size_t polys, int mode, hipStream_t st) {
    hipLaunchKernelGGL(k_ntt_forward, dim3((unsigned)polys), dim3(NT), POLY_STRIDE * sizeof(double), st, T, in, out, mode);
}
void launch_ntt_inverse(const Tables& T, const double* in, u64* out, size_t polys, hipStream_t st) {
    hipLaunchKernelGGL(k_ntt_inverse, dim3((unsigned)polys), dim3(NT), POLY_STRIDE * sizeof(double), st, T, in, out);
}
void launch_external_product(const Tables& T, const double* bsk, const u32* idx, const u64* glwe, u64* out,
                             size_t count, hipStream_t st) {
    hipLaunchKernelGGL(k_external_product, dim3((unsigned)count), dim3(NT), LDS_BYTES, st, T, bsk, idx, glwe, out);
}
size_t g64_split_payload_bytes(size_t cts) { return g64_split_payload_words(cts) * sizeof(double); }
size_t g64_split_flag_bytes(size_t cts) { return g64_split_flag_words(cts) * sizeof(unsigned); }

void launch_blind_rotate(const Tables& T, const double* bsk, const u64* lwe_small, int n, const u64* luts,
                         const u32* lut_index, u64* glwe_out, u64* lwe_out, size_t count,
                         const BlockMap& omap, int grouping, int pair_above, int wide_below, const G64Guard& guard,
                         hipStream_t st) {
    if (grouping >= 2) {
        // four GGSWs per pair of key bits (3: in factored form; 4: eight per triple, factored); two ciphertexts per workgroup, one when
        // the launch would leave CUs without a workgroup otherwise (pair_above: as for the prime modulus)
        if (!guard.fault || count > guard.capacity) throw std::logic_error("lift-guard flags missing or too few for this launch");
        if (hipMemsetAsync(guard.fault, 0, count * sizeof(int), st) != hipSuccess) throw std::runtime_error("hipMemsetAsync failed");
        const bool pair = pair_above >= 0 && count > (size_t)pair_above;
        const dim3 grid((unsigned)(pair ? (count + 1) / 2 : count));
        auto go = [&](auto kernel, int nc, dim3 g) {
            hipLaunchKernelGGL(kernel, g, dim3(NT), g64_lds_bytes(nc), st, T, bsk, lwe_small, n, luts, lut_index, glwe_out,
                               lwe_out, omap, (unsigned)count, guard.fault, guard.k_guard, guard.fallbacks);
        };
        if (guard.xchg && count <= guard.split_below) {   // up to one ciphertext per two CUs: one CU per prime
            if (hipMemsetAsync(guard.xchg_flags, 0, g64_split_flag_words(count) * sizeof(unsigned), st) != hipSuccess)
                throw std::runtime_error("hipMemsetAsync failed");
            auto split = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3((unsigned)(16 * ((count + 7) / 8))), dim3(NT), g64_lds_bytes(1), st, T, bsk, lwe_small,
                                   n, luts, lut_index, glwe_out, lwe_out, omap, (unsigned)count, guard.fault, guard.k_guard,
                                   guard.xchg, guard.xchg_flags, guard.poll_limit);
            };
            if (grouping == 4) split(k_blind_rotate_g64_split<true, 3>);
            else if (grouping == 3) split(k_blind_rotate_g64_split<true, 2>);
            else split(k_blind_rotate_g64_split<false, 2>);
        }
        else if (count <= (size_t)wide_below) {   // up to one ciphertext per CU: 512 threads each, one half per prime
            auto wide = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3((unsigned)count), dim3(2 * NT), G64_WIDE_LDS_BYTES, st, T, bsk, lwe_small, n, luts,
                                   lut_index, glwe_out, lwe_out, omap, guard.fault, guard.k_guard);
            };
            if (grouping == 4) wide(k_blind_rotate_g64_wide<true, 3>);
            else if (grouping == 3) wide(k_blind_rotate_g64_wide<true, 2>);
            else wide(k_blind_rotate_g64_wide<false, 2>);
        }
        else if (grouping == 4) { if (pair) go(k_blind_rotate_g64_pair<true, 2, false, 3>, 2, grid); else go(k_blind_rotate_g64_pair<true, 1, false, 3>, 1, grid); }
        else if (grouping == 3) { if (pair) go(k_blind_rotate_g64_pair<true, 2>, 2, grid); else go(k_blind_rotate_g64_pair<true, 1>, 1, grid); }
        else { if (pair) go(k_blind_rotate_g64_pair<false, 2>, 2, grid); else go(k_blind_rotate_g64_pair<false, 1>, 1, grid); }
        // the slow path for whatever the fast one marked: one workgroup per ciphertext, gone at once when unmarked
        if (grouping == 4) go(k_blind_rotate_g64_pair<true, 1, true, 3>, 1, dim3((unsigned)count));
        else if (grouping == 3) go(k_blind_rotate_g64_pair<true, 1, true>, 1, dim3((unsigned)count));
        else go(k_blind_rotate_g64_pair<false, 1, true>, 1, dim3((unsigned)count));
        return;
    }
    hipLaunchKernelGGL(k_blind_rotate, dim3((unsigned)count), dim3(NT), LDS_BYTES, st, T, bsk, lwe_small, n, luts,
                       lut_index, glwe_out, lwe_out, omap);
}
void launch_ksk_colsum(const u64* ksk, int rows, int cols, int half_base, u64* colsum, hipStream_t st) {
    hipLaunchKernelGGL(k_ksk_colsum, dim3((cols + 255) / 256), dim3(256), 0, st, ksk, rows, cols, half_base, colsum);
}
void launch_keyswitch(const u64* ksk, const u64* colsum, const u64* in, u32* dig_scratch, u64* out, size_t count,
                      int big, int n, int base_log, int levels, const BlockMap& imap, hipStream_t st) {
    const unsigned colblocks = (unsigned)((n + 1 + 255) / 256), groups = (unsigned)((count + KS_CT - 1) / KS_CT);
    const size_t total = (size_t)groups * KS_CT * big;
    unsigned dblocks = (unsigned)((total + 255) / 256);
    if (dblocks > 8192) dblocks = 8192;
    hipLaunchKernelGGL(k_ks_digits, dim3(dblocks), dim3(256), 0, st, in, dig_scratch, (int)count, big, base_log, levels,
                       imap);
    // enough workgroups to cover the GPU a few times over
    unsigned split = 1;
    while (split * 2 <= 64 && colblocks * groups * split < 1024) split *= 2;
    if (split > 1) (void)hipMemsetAsync(out, 0, count * (size_t)(n + 1) * sizeof(u64), st);
    const dim3 grid(colblocks, groups, split);
    switch (levels) {
        case 1: hipLaunchKernelGGL(k_keyswitch<1>, grid, dim3(256), 0, st, ksk, colsum, in, dig_scratch, out, (int)count, big, n, imap); break;
        case 2: hipLaunchKernelGGL(k_keyswitch<2>, grid, dim3(256), 0, st, ksk, colsum, in, dig_scratch, out, (int)count, big, n, imap); break;
        case 3: hipLaunchKernelGGL(k_keyswitch<3>, grid, dim3(256), 0, st, ksk, colsum, in, dig_scratch, out, (int)count, big, n, imap); break;
        default: hipLaunchKernelGGL(k_keyswitch<4>, grid, dim3(256), 0, st, ksk, colsum, in, dig_scratch, out, (int)count, big, n, imap); break;
    }
}
size_t ks_valu_max_count() { return (size_t)65535 * KS_CT; }
size_t keyswitch_scratch_words(size_t count, int big) { return ((count + KS_CT - 1) / KS_CT) * KS_CT * (size_t)big; }

// ------------------------------------------------------------------------------------------------
// linear layer of a radix program: for every instance and every LinOp,
//   arena[inst][op.out] = sum_k coef[k] * arena[inst][src[k]] + cst * delta (body word only)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lin_batch(u64* __restrict__ arena, const LinOp* __restrict__ ops, int nops,
                                                   size_t inst_stride_blocks, u64 delta) {
    const size_t inst = blockIdx.x / nops;
    const LinOp op = ops[blockIdx.x % nops];
    u64* base = arena + inst * inst_stride_blocks * (size_t)(N + 1);
    u64* o = base + (size_t)op.out * (N + 1);
    for (int w = threadIdx.x; w <= N; w += blockDim.x) {
        u64 v = (w == N) ? (u64)(i64)op.cst * delta : 0;
        for (uint32_t k = 0; k < op.n; ++k) v += (u64)(i64)op.coef[k] * base[(size_t)op.src[k] * (N + 1) + w];
        o[w] = v;
    }
}

// trivial (noiseless) blocks: all-zero mask, body = bodies[inst][j]; used for clear right operands
__global__ void __launch_bounds__(256) k_fill_trivial(u64* __restrict__ arena, size_t inst_stride_blocks,
                                                      u32 first_block, u32 m, const u64* __restrict__ bodies) {
    const size_t inst = blockIdx.x / m;
    const u32 j = blockIdx.x % m;
    u64* o = arena + (inst * inst_stride_blocks + first_block + j) * (size_t)(N + 1);
    for (int w = threadIdx.x; w <= N; w += blockDim.x) o[w] = (w == N) ? bodies[inst * m + j] : 0;
}

void launch_fill_trivial(u64* arena, size_t instances, size_t inst_stride_blocks, u32 first_block, u32 m,
                         const u64* bodies, hipStream_t st) {
    hipLaunchKernelGGL(k_fill_trivial, dim3((unsigned)(instances * m)), dim3(256), 0, st, arena, inst_stride_blocks,
                       first_block, m, bodies);
}

// Ciphertext arrays as the caller holds them (per ciphertext: header words, then its blocks) <-> arena
// blocks.  One workgroup per (block, ciphertext); `idx` lists the arena block of every ciphertext
// block, or is null for consecutive blocks starting at `first`.  With these the executor moves a
// whole operand or result array across PCIe in ONE copy instead of one copy per block and instance.
__global__ void __launch_bounds__(256) k_blocks_move(u64* __restrict__ arena, size_t inst_stride_blocks,
                                                     const u32* __restrict__ idx, u32 first, u64* __restrict__ flat,
                                                     size_t ct_words, u32 header_words, int to_arena) {
    const u32 j = blockIdx.x;
    const size_t inst = blockIdx.y;
    u64* a = arena + (inst * inst_stride_blocks + (idx ? idx[j] : first + j)) * (size_t)(N + 1);
    u64* f = flat + inst * ct_words + header_words + (size_t)j * (N + 1);
    if (to_arena) { for (int w = threadIdx.x; w <= N; w += blockDim.x) a[w] = f[w]; }
    else { for (int w = threadIdx.x; w <= N; w += blockDim.x) f[w] = a[w]; }
}

void launch_blocks_move(u64* arena, size_t instances, size_t inst_stride_blocks, const u32* idx, u32 first, u32 nblocks,
                        u64* flat, size_t ct_words, u32 header_words, bool to_arena, hipStream_t st) {
    if (!instances || !nblocks) return;
    hipLaunchKernelGGL(k_blocks_move, dim3(nblocks, (unsigned)instances), dim3(256), 0, st, arena, inst_stride_blocks, idx,
                       first, flat, ct_words, header_words, to_arena ? 1 : 0);
}

// The same between the arena and ciphertexts that live one by one in device memory (the context's
// ciphertext store): ptrs[inst] is the first block of ciphertext `inst`, its blocks back to back.
__global__ void __launch_bounds__(256) k_blocks_move_ptrs(u64* __restrict__ arena, size_t inst_stride_blocks,
                                                          const u32* __restrict__ idx, u32 first,
                                                          u64* const* __restrict__ ptrs, int to_arena) {
    const u32 j = blockIdx.x;
    const size_t inst = blockIdx.y;
    u64* a = arena + (inst * inst_stride_blocks + (idx ? idx[j] : first + j)) * (size_t)(N + 1);
    u64* f = ptrs[inst] + (size_t)j * (N + 1);
    if (to_arena) { for (int w = threadIdx.x; w <= N; w += blockDim.x) a[w] = f[w]; }
    else { for (int w = threadIdx.x; w <= N; w += blockDim.x) f[w] = a[w]; }
}

void launch_blocks_move_ptrs(u64* arena, size_t instances, size_t inst_stride_blocks, const u32* idx, u32 first,
                             u32 nblocks, u64* const* ptrs, bool to_arena, hipStream_t st) {
    if (!instances || !nblocks) return;
    hipLaunchKernelGGL(k_blocks_move_ptrs, dim3(nblocks, (unsigned)instances), dim3(256), 0, st, arena, inst_stride_blocks,
                       idx, first, ptrs, to_arena ? 1 : 0);
}

void launch_lin_batch(u64* arena, const LinOp* ops, int nops, size_t instances, size_t inst_stride_blocks, u64 delta,
                      hipStream_t st) {
    hipLaunchKernelGGL(k_lin_batch, dim3((unsigned)(instances * nops)), dim3(256), 0, st, arena, ops, nops,
                       inst_stride_blocks, delta);
}

// ------------------------------------------------------------------------------------------------
// modulus-switch noise reduction (ms_noise.h): one workgroup per small ciphertext, one wavefront
// per candidate, WAVES candidates per round in index order (-1 = the ciphertext as it is).  The
// choice is the first candidate in index order within the bound, whatever WAVES is: four per round
// waste the least work in a long batch, sixteen per round need a quarter of the rounds when a
// bootstrap layer is a few dozen ciphertexts and the slowest search sets the launch time.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ i64 wave_sum_i64(i64 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_ms_noise_reduce(u64* __restrict__ lwe_small, int n,
                                                                const u64* __restrict__ zeros, MsParams m,
                                                                int* __restrict__ chosen_out) {
    extern __shared__ u64 ms_ct[];  // n + 1 words
    __shared__ u64 est[WAVES];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    u64* ct = lwe_small + (size_t)blockIdx.x * (n + 1);
    for (int j = t; j <= n; j += 64 * WAVES) ms_ct[j] = ct[j];
    __syncthreads();
    u64 best = ~0ull;
    int best_idx = -1, chosen = -2;
    for (int base = -1; base < (int)m.zcount && chosen == -2; base += WAVES) {
        const int cand = base + wave;
        if (cand < (int)m.zcount) {
            const u64* z = zeros + (size_t)(cand < 0 ? 0 : cand) * (n + 1);
            i64 s1 = 0, s2 = 0;
            for (int j = lane; j <= n; j += 64) {
                const u64 a = ms_ct[j] + (cand < 0 ? 0 : z[j]);
                const i64 e = ms_round_error(a, m.log2_2n);
                if (j == n) s1 += 2 * e;
                else { s1 -= e; const i64 h = e >> 26; s2 += h * h; }
            }
            s1 = wave_sum_i64(s1);
            s2 = wave_sum_i64(s2);
            if (lane == 0) est[wave] = ms_estimate(s1, (u64)s2, m);
        }
        __syncthreads();
        for (int w = 0; w < WAVES && base + w < (int)m.zcount; ++w) {
            const u64 e = est[w];
            if (e <= m.bound_fx) { chosen = base + w; break; }
            if (e < best) { best = e; best_idx = base + w; }
        }
        __syncthreads();
    }
    if (chosen == -2) chosen = best_idx;
    if (chosen >= 0) {
        const u64* z = zeros + (size_t)chosen * (n + 1);
        for (int j = t; j <= n; j += 64 * WAVES) ct[j] = ms_ct[j] + z[j];
    }
    if (chosen_out && t == 0) chosen_out[blockIdx.x] = chosen;
}

void launch_ms_noise_reduce(u64* lwe_small, int n, const u64* zeros, const MsParams& m, int* chosen, size_t count,
                            hipStream_t st) {
    // Candidates per round = waves per workgroup.  The search of one ciphertext ends at the first
    // candidate within the bound (about one in eighteen is), so the launch ends with its unluckiest
    // ciphertext, some 150 candidates deep among 4096: measured per 4096 ciphertexts 296-341 us with
    // four candidates per round, 160-169 with eight, 155-164 with sixteen.  Very long batches are bound
    // by the candidates evaluated past the chosen one instead and take eight.
    if (count <= 8192)
        hipLaunchKernelGGL(k_ms_noise_reduce<16>, dim3((unsigned)count), dim3(1024), (size_t)(n + 1) * sizeof(u64), st,
                           lwe_small, n, zeros, m, chosen);
    else
        hipLaunchKernelGGL(k_ms_noise_reduce<8>, dim3((unsigned)count), dim3(512), (size_t)(n + 1) * sizeof(u64), st,
                           lwe_small, n, zeros, m, chosen);
}

void launch_lwe_linear(const u64* a, const u64* b, u64* out, size_t count, int dim1, u64 ca, u64 cb, u64 add_body,
                       hipStream_t st) {
    const size_t total = count * (size_t)dim1;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_lwe_linear, dim3(blocks), dim3(256), 0, st, a, b, out, total, dim1, ca, cb, add_body);
}

}  // namespace lfhe
