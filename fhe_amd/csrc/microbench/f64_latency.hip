// Dependent-issue behaviour of the FP64 pipe on gfx950: throughput of v_fma_f64 as a function of
// the number of independent chains per wave (ILP) and of waves per SIMD (TLP).
// Output: rows "waves_per_simd chains cycles_per_instr_per_simd" at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int NV>
__global__ void __launch_bounds__(64) chain(double* out, int iters, double y, double z, long long* clk) {
    double x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = 1.0 + threadIdx.x * 1e-3 + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
        }
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

typedef void (*kern_t)(double*, int, double, double, long long*);

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* d_out; long long* d_clk;
    CK(hipMalloc(&d_out, sizeof(double) * cus * 4 * 8 * 64)); CK(hipMalloc(&d_clk, 8));
    kern_t ks[] = {chain<1>, chain<2>, chain<3>, chain<4>, chain<6>, chain<8>};
    int nvs[] = {1, 2, 3, 4, 6, 8};
    const int iters = 20000;
    printf("# waves/SIMD chains  s_memtime-cycles per v_fma_f64 per wave   (wave-64, 64-thread blocks)\n");
    for (int wps = 1; wps <= 4; ++wps)
        for (int k = 0; k < 6; ++k) {
            const int blocks = cus * 4 * wps;  // one 64-thread block = one wave; 4 SIMDs per CU
            hipLaunchKernelGGL(ks[k], dim3(blocks), dim3(64), 0, 0, d_out, iters, 0.999, 1e-9, d_clk);
            CK(hipDeviceSynchronize());
            long long c; CK(hipMemcpy(&c, d_clk, 8, hipMemcpyDeviceToHost));
            const double per = (double)c / ((double)iters * 16 * nvs[k]);
            printf("%d %d  %.2f cycles/instr/wave  -> %.2f cycles/instr/SIMD\n", wps, nvs[k], per, per / wps);
        }
    return 0;
}
