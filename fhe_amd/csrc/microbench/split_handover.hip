// What would it cost to spread ONE ciphertext's blind rotation over TWO compute units?
//
// The 512-thread blind rotation gives each polynomial of the accumulator to one half of a workgroup;
// the halves meet once per step, at the point-wise stage, where each needs the other's transformed
// digits (2048 doubles = 16 KB each way).  Inside one workgroup that exchange goes through LDS.  Put
// the halves on two CUs (256 threads each, a CU to itself) and it has to go through the XCD's L2:
// every storing wave drains its write-through stores, one lane raises a flag, the partner polls the
// flag and loads the 16 KB past its L1 -- 459 times per bootstrap.
//
// This program measures exactly that, with the arithmetic of a step standing in between the
// exchanges (radix-8 passes of exact f64 butterflies, `work` of them per step; a blind-rotation step
// is about 30 per thread):
//   mode 0  one workgroup of 512 threads per pair, exchange through LDS        (what the kernel does)
//   mode 1  two workgroups of 256 threads per pair on one XCD (blocks b, b + 8), exchange through L2
//   mode 2  the same with no exchange at all                                    (the work alone, 1 wave per SIMD)
// Every word handed over carries a tag (step, index) that the reader checks, so a stale read counts
// as an error; the poll gives up after a bounded number of tries and reports it (no wave waits
// forever).  Both partners of a pair sit on one XCD; the launch must not exceed one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#include "../pbs_device.h"
using namespace lfhe;

typedef double v2d __attribute__((ext_vector_type(2)));
extern __shared__ __attribute__((aligned(16))) double lds[];

constexpr int WORDS = 2048;            // doubles handed over per direction and step
constexpr int SLOT = WORDS / 2;        // v2d per slot
constexpr unsigned POLL_LIMIT = 1u << 20;

__device__ __forceinline__ void st4_sc1(v2d* p, const v2d (&v)[4]) {
    // four 16-byte write-through stores, 4 KB apart (lane-contiguous within each), then drained
    v2d *p1 = p + 256, *p2 = p + 512, *p3 = p + 768;
    asm volatile("global_store_dwordx4 %0, %4, off sc1\n"
                 "global_store_dwordx4 %1, %5, off sc1\n"
                 "global_store_dwordx4 %2, %6, off sc1\n"
                 "global_store_dwordx4 %3, %7, off sc1\n"
                 "s_waitcnt vmcnt(0)"
                 :: "v"(p), "v"(p1), "v"(p2), "v"(p3), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory");
}
__device__ __forceinline__ void ld4_sc1(const v2d* p, v2d (&v)[4]) {
    const v2d *p1 = p + 256, *p2 = p + 512, *p3 = p + 768;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n"
                 "global_load_dwordx4 %1, %5, off sc1\n"
                 "global_load_dwordx4 %2, %6, off sc1\n"
                 "global_load_dwordx4 %3, %7, off sc1\n"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ void flag_store(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc1\ns_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned flag_load(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\ns_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__device__ __forceinline__ void work_pass(double (&x)[8], const double (&w)[7], const Prime& P, int reps) {
    for (int r = 0; r < reps; ++r) { ct_radix8(x, w, P); norm8(x, P); }
}

// tag of word i of the slot that half h of a pair publishes in step s
__device__ __forceinline__ double tag(unsigned s, unsigned h, unsigned i) { return (double)(s * 8192u + h * 4096u + i); }

template <int MODE>
__global__ void __launch_bounds__(MODE == 0 ? 512 : 256, 1)
k_split(v2d* __restrict__ payload, unsigned* __restrict__ flags, int pairs, int steps, int work, Prime P, double seed,
        double* __restrict__ out, unsigned* __restrict__ errors) {
    unsigned pair, h, t;
    if (MODE == 0) { pair = blockIdx.x; h = threadIdx.x >> 8; t = threadIdx.x & 255; }
    else { const unsigned b = blockIdx.x; pair = (b / 16) * 8 + (b % 8); h = (b / 8) & 1; t = threadIdx.x; }
    if ((int)pair >= pairs) return;
    double x[8], w[7];
    for (int j = 0; j < 8; ++j) x[j] = seed + t + 256 * j;
    for (int k = 0; k < 7; ++k) w[k] = seed * (k + 2) + 17.0;
    unsigned bad = 0, dead = 0;
    __shared__ unsigned s_dead;
    if (threadIdx.x == 0) s_dead = 0;
    __syncthreads();
    // [pair][half][slot parity][SLOT]; flags one 128-byte line per (pair, half)
    v2d* mine = payload + ((size_t)(pair * 2 + h) * 2) * SLOT;
    const v2d* theirs = payload + ((size_t)(pair * 2 + (1 - h)) * 2) * SLOT;
    unsigned* my_flag = flags + (size_t)(pair * 2 + h) * 32;
    const unsigned* their_flag = flags + (size_t)(pair * 2 + (1 - h)) * 32;
    for (int s = 1; s <= steps; ++s) {
        work_pass(x, w, P, work);
        v2d v[4], u[4];
        for (int k = 0; k < 4; ++k) { v[k].x = tag(s, h, 2 * (t + 256 * k)); v[k].y = x[k] + x[k + 4]; }
        if (MODE == 0) {
            v2d* l = reinterpret_cast<v2d*>(lds);
            for (int k = 0; k < 4; ++k) l[h * SLOT + t + 256 * k] = v[k];
            __syncthreads();
            for (int k = 0; k < 4; ++k) u[k] = l[(1 - h) * SLOT + t + 256 * k];
            __syncthreads();
        } else if (MODE == 1) {
            st4_sc1(mine + (s & 1) * SLOT + t, v);          // returns when this wave's stores have left
            __syncthreads();                                 // ... and every wave's have
            if (threadIdx.x == 0) {
                flag_store(my_flag, (unsigned)s);
                unsigned tries = 0;
                while (!dead && flag_load(their_flag) < (unsigned)s) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++tries > POLL_LIMIT) { dead = 1; s_dead = 1; }
                }
            }
            __syncthreads();                                 // the poll has matched before anyone loads
            dead = s_dead;
            if (dead) break;
            ld4_sc1(theirs + (s & 1) * SLOT + t, u);
        } else {
            for (int k = 0; k < 4; ++k) u[k] = v[k];
        }
        if (MODE != 2)
            for (int k = 0; k < 4; ++k) bad += u[k].x != tag(s, 1 - h, 2 * (t + 256 * k));
        for (int k = 0; k < 4; ++k) x[k] += u[k].y * 1e-300;
    }
    if (bad) atomicAdd(errors, bad);
    if (dead && threadIdx.x == 0) atomicAdd(errors + 1, 1u);
    double sum = 0;
    for (int j = 0; j < 8; ++j) sum += x[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int steps = 459;
    const Prime P{P0, 1.0 / P0};
    v2d* d_payload; unsigned *d_flags, *d_err; double* d_out;
    const int max_pairs = cus / 2;
    CK(hipMalloc(&d_payload, sizeof(v2d) * (size_t)max_pairs * 4 * SLOT));
    CK(hipMalloc(&d_flags, 128 * (size_t)max_pairs * 2));
    CK(hipMalloc(&d_err, 8));
    CK(hipMalloc(&d_out, sizeof(double) * 512 * (size_t)cus));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_split<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WORDS * 8 + 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_split<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_split<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"one 512-thread workgroup, LDS", "two 256-thread workgroups, L2", "two 256-thread workgroups, no exchange"};
    printf("%d CUs; %d steps per launch; per step and pair (us)\n", cus, steps);
    for (int pairs : {16, 64, max_pairs}) {
        for (int work : {0, 10, 30}) {
            double us[3] = {0, 0, 0};
            for (int mode = 0; mode < 3; ++mode) {
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(d_flags, 0, 128 * (size_t)max_pairs * 2));
                    CK(hipMemset(d_err, 0, 8));
                    CK(hipDeviceSynchronize());
                    const unsigned grid1 = 16 * ((pairs + 7) / 8);
                    CK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(k_split<0>, dim3(pairs), dim3(512), 2 * WORDS * 8 + 65536, 0, d_payload, d_flags, pairs, steps, work, P, 3.0, d_out, d_err);
                    if (mode == 1) hipLaunchKernelGGL(k_split<1>, dim3(grid1), dim3(256), 90 * 1024, 0, d_payload, d_flags, pairs, steps, work, P, 3.0, d_out, d_err);
                    if (mode == 2) hipLaunchKernelGGL(k_split<2>, dim3(grid1), dim3(256), 90 * 1024, 0, d_payload, d_flags, pairs, steps, work, P, 3.0, d_out, d_err);
                    CK(hipGetLastError());
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float f; CK(hipEventElapsedTime(&f, e0, e1));
                    us[mode] = f * 1e3 / steps;
                    unsigned err[2]; CK(hipMemcpy(err, d_err, 8, hipMemcpyDeviceToHost));
                    if (err[0] || err[1]) { printf("mode %d pairs %d work %d: %u stale words, %u timed-out workgroups\n", mode, pairs, work, err[0], err[1]); return 2; }
                }
            }
            printf("pairs %3d work %2d passes:  %s %6.2f | %s %6.2f | %s %6.2f | the exchange through L2 adds %5.2f\n",
                   pairs, work, names[0], us[0], names[1], us[1], names[2], us[2], us[1] - us[2]);
        }
    }
    (void)argc; (void)argv;
    return 0;
}
