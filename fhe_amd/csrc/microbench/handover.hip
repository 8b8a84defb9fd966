// The hand-over between two register passes of the transform, inside one wavefront: through LDS
// (what the kernels do) against cross-lane register moves (what a shuffle-based transform would do).
//
// The exchange between passes 2 and 3 (and 3 and 4) permutes, within a wavefront, a 3-bit register
// index with three lane bits: an 8 x 8 transpose between registers and lanes, of 64-bit values.
//   LDS form:     8 ds_write_b64 + 8 ds_read_b64 per polynomial (+ address arithmetic); the LDS
//                 pipe works beside the vector ALU.
//   shuffle form: three butterfly stages; per stage and register pair a select of what to send, the
//                 cross-lane move of a 64-bit value (two 32-bit moves), and the selects that place
//                 what arrived: every one of them a vector-ALU instruction (or an LDS-crossbar
//                 ds_bpermute), in a kernel whose bound is vector-ALU issue.
// Both forms run R hand-overs, each followed by the same radix-8 pass of exact f64 butterflies, so
// that the hand-over competes with the arithmetic for issue slots as it does in the blind rotation.
// Output: time per hand-over + pass for either form, and for the pass alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#include "../pbs_device.h"
using namespace lfhe;

extern __shared__ __attribute__((aligned(16))) double lds[];

__device__ __forceinline__ void pass(double (&x)[8], const double (&w)[7], const Prime& P) { ct_radix8(x, w, P); norm8(x, P); }

// One stage of the register <-> lane transpose done with the cheapest cross-lane instructions gfx950
// has: the values a of the lanes whose bit `m` is set change places with the values b of their
// partners (lane ^ m).  m = 16: v_permlane16_swap, one instruction per 32-bit half.  m = 8 and 4
// (inside a row of sixteen lanes): a copy and two DPP moves restricted to banks of four lanes.
template <int M>
__device__ __forceinline__ void swap_stage32(unsigned& a, unsigned& b) {
    if constexpr (M == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r[0]; b = r[1];
    } else {
        // lanes with the bit set are banks {2,3} for m = 8 and banks {1,3} for m = 4
        constexpr int up_banks = M == 8 ? 0xC : 0xA, low_banks = 0xF ^ up_banks;
        constexpr int from_low = M == 8 ? 0x128 : 0x114;   // row_ror:8 / row_shr:4: read lane - m (mod 16 for ror)
        constexpr int from_up = M == 8 ? 0x128 : 0x104;    // row_ror:8 / row_shl:4: read lane + m
        const unsigned t = a;
        a = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)b, from_low, 0xF, up_banks, false);
        b = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)t, from_up, 0xF, low_banks, false);
    }
}
template <int M>
__device__ __forceinline__ void swap_stage(double& a, double& b) {
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    swap_stage32<M>(alo, blo);
    swap_stage32<M>(ahi, bhi);
    a = __hiloint2double((int)ahi, (int)alo);
    b = __hiloint2double((int)bhi, (int)blo);
}

// mode 0: pass only; 1: LDS hand-over (the kernels' own st/ld with their padding); 2: hand-over with
// __shfl_xor as the compiler lowers it; 3: hand-over with permlane swaps and DPP moves
template <int MODE>
__global__ void __launch_bounds__(256) k_handover(double* out, int reps, Prime P, double seed) {
    const unsigned t = threadIdx.x;
    double x[8], w[7];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (double)((t * 8 + j) * 2654435761u % 1000003u);
#pragma unroll
    for (int k = 0; k < 7; ++k) w[k] = seed + 17.0 * k + (double)(t >> 2);
    for (int r = 0; r < reps; ++r) {
        if (MODE == 1) {
            unsigned tt = t;
            asm volatile("" : "+v"(tt));
            st_o2p23(lds, tt, x);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            ld_o3p23(lds, tt, x);
        }
        if (MODE == 2) {
            // registers j (3 bits) <-> lane bits 4, 3, 2: stage (lane mask m, register stride s)
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const int m = 4 << st, s = 1 << st;
                const bool up = (t & m) != 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j & s) continue;
                    const double send = up ? x[j] : x[j + s];
                    const double recv = __shfl_xor(send, m, 64);
                    if (up) x[j] = recv; else x[j + s] = recv;
                }
            }
        }
        else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!(j & 1)) swap_stage<4>(x[j], x[j + 1]);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!(j & 2)) swap_stage<8>(x[j], x[j + 2]);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!(j & 4)) swap_stage<16>(x[j], x[j + 4]);
        }
        pass(x, w, P);
    }
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += x[j];
    out[blockIdx.x * 256 + t] = acc;
}

// the two shuffle forms implement the same permutation: checked once on lane * 8 + j
__global__ void __launch_bounds__(64) k_check(int* bad) {
    const unsigned t = threadIdx.x;
    double a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = b[j] = (double)(t * 8 + j);
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        const int m = 4 << st, s = 1 << st;
        const bool up = (t & m) != 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j & s) continue;
            const double send = up ? a[j] : a[j + s];
            const double recv = __shfl_xor(send, m, 64);
            if (up) a[j] = recv; else a[j + s] = recv;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (!(j & 1)) swap_stage<4>(b[j], b[j + 1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) if (!(j & 2)) swap_stage<8>(b[j], b[j + 2]);
#pragma unroll
    for (int j = 0; j < 8; ++j) if (!(j & 4)) swap_stage<16>(b[j], b[j + 4]);
    int wrong = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // register j of lane t holds what register (lane bits 4..2) of lane (t with bits 4..2 := j) held
        const unsigned src_lane = (t & ~0x1Cu) | ((unsigned)j << 2), src_reg = (t >> 2) & 7;
        wrong += a[j] != (double)(src_lane * 8 + src_reg);
        wrong += b[j] != a[j];
    }
    if (wrong) atomicAdd(bad, wrong);
}

int main() {
    {
        int* d_bad; CK(hipMalloc(&d_bad, 4)); CK(hipMemset(d_bad, 0, 4));
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d_bad);
        int bad = -1; CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        printf("shuffle forms: %s (%d mismatches against the register <-> lane transpose)\n", bad == 0 ? "agree" : "DISAGREE", bad);
        if (bad) return 2;
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, wgs = cus * 4;   // four workgroups per CU, as the blind rotation
    double* d_out; CK(hipMalloc(&d_out, sizeof(double) * wgs * 256));
    const Prime P{P0, 1.0 / P0};
    const int reps = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"radix-8 pass alone", "LDS hand-over + pass", "__shfl_xor hand-over + pass",
                            "permlane/DPP hand-over + pass"};
    double ms[4];
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {   // first run warms up
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_handover<0>, dim3(wgs), dim3(256), POLY_STRIDE * 8, 0, d_out, reps, P, 3.0);
            if (mode == 1) hipLaunchKernelGGL(k_handover<1>, dim3(wgs), dim3(256), POLY_STRIDE * 8, 0, d_out, reps, P, 3.0);
            if (mode == 2) hipLaunchKernelGGL(k_handover<2>, dim3(wgs), dim3(256), POLY_STRIDE * 8, 0, d_out, reps, P, 3.0);
            if (mode == 3) hipLaunchKernelGGL(k_handover<3>, dim3(wgs), dim3(256), POLY_STRIDE * 8, 0, d_out, reps, P, 3.0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float f; CK(hipEventElapsedTime(&f, e0, e1)); ms[mode] = f;
        }
        printf("%-28s %8.3f ms for %d rounds on %d workgroups: %.1f ns per round and workgroup slot\n", names[mode], ms[mode], reps,
               wgs, ms[mode] * 1e6 / reps);
    }
    printf("hand-over cost per round: LDS %.1f ns, __shfl_xor %.1f ns, permlane/DPP %.1f ns (pass alone %.1f ns)\n",
           (ms[1] - ms[0]) * 1e6 / reps, (ms[2] - ms[0]) * 1e6 / reps, (ms[3] - ms[0]) * 1e6 / reps, ms[0] * 1e6 / reps);
    return 0;
}
