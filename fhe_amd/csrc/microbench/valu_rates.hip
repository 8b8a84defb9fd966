// Register-resident VALU rate calibration for gfx950.
//
// Two families of measurements:
//   (1) raw issue rate of single instructions (inline asm, independent chains);
//   (2) throughput of complete NTT butterflies (a,b) -> (a + w*b, a - w*b) mod p in the
//       candidate exact arithmetics for the negacyclic transform.
// Everything stays in registers: the numbers are the VALU ceiling a kernel built on that
// arithmetic can approach, and are what the bootstrap kernels' roofline fraction is priced
// against.
//
// Output: one line per measurement, "name  Gop/s  slots", where slots = full-rate lane-slots per
// op = (256 CU * 128 lanes * clock) / rate with the clock taken from a v_add_u32 calibration.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef uint32_t u32;
typedef uint64_t u64;
typedef unsigned __int128 u128;

constexpr int NV = 8;        // independent chains per lane
constexpr int UNROLL = 8;    // ops per chain per loop trip

// ------------------------------------------------------------------ raw instructions
#define RAW32_KERNEL(NAME, ASMSTR)                                                    \
__global__ void __launch_bounds__(256) raw_##NAME(u32* out, int iters, u32 seed) {    \
    u32 x[NV]; u32 y = seed | 1u, z = seed * 2654435761u + 12345u;                    \
    _Pragma("unroll") for (int i = 0; i < NV; ++i) x[i] = threadIdx.x * 977u + i * 131u + seed; \
    for (int it = 0; it < iters; ++it) {                                              \
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                          \
            _Pragma("unroll") for (int i = 0; i < NV; ++i)                            \
                asm volatile(ASMSTR : "+v"(x[i]) : "v"(y), "v"(z));                   \
        }                                                                             \
    }                                                                                 \
    u32 s = 0; _Pragma("unroll") for (int i = 0; i < NV; ++i) s ^= x[i];              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                   \
}

RAW32_KERNEL(v_add_u32,        "v_add_u32 %0, %0, %1")
RAW32_KERNEL(v_add3_u32,       "v_add3_u32 %0, %0, %1, %2")
RAW32_KERNEL(v_xor_b32,        "v_xor_b32 %0, %0, %1")
RAW32_KERNEL(v_mul_lo_u32,     "v_mul_lo_u32 %0, %0, %1")
RAW32_KERNEL(v_mul_hi_u32,     "v_mul_hi_u32 %0, %0, %1")
RAW32_KERNEL(v_mul_u32_u24,    "v_mul_u32_u24 %0, %0, %1")
RAW32_KERNEL(v_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
RAW32_KERNEL(v_mad_u32_u24,    "v_mad_u32_u24 %0, %0, %1, %2")
RAW32_KERNEL(v_mad_i32_i24,    "v_mad_i32_i24 %0, %0, %1, %2")
RAW32_KERNEL(v_fma_f32,        "v_fma_f32 %0, %0, %1, %2")
RAW32_KERNEL(v_mul_f32,        "v_mul_f32 %0, %0, %1")
RAW32_KERNEL(v_rndne_f32,      "v_rndne_f32 %0, %0")
RAW32_KERNEL(v_alignbit_b32,   "v_alignbit_b32 %0, %0, %1, 24")
RAW32_KERNEL(v_min_u32,        "v_min_u32 %0, %0, %1")
RAW32_KERNEL(v_cndmask_b32,    "v_cndmask_b32 %0, %0, %1, vcc")
RAW32_KERNEL(v_mad_u32_u16,    "v_mad_u32_u16 %0, %0, %1, %2")
RAW32_KERNEL(v_pk_mul_lo_u16,  "v_pk_mul_lo_u16 %0, %0, %1")
RAW32_KERNEL(v_pk_mad_u16,     "v_pk_mad_u16 %0, %0, %1, %2")
RAW32_KERNEL(v_dot4_u32_u8,    "v_dot4_u32_u8 %0, %0, %1, %2")
RAW32_KERNEL(v_cvt_f32_u32,    "v_cvt_f32_u32 %0, %0")
RAW32_KERNEL(v_cvt_u32_f32,    "v_cvt_u32_f32 %0, %0")

#define RAW64_KERNEL(NAME, ASMSTR, CLOB)                                              \
__global__ void __launch_bounds__(256) raw_##NAME(u32* out, int iters, u32 seed) {    \
    u64 x[NV]; u64 y = ((u64)seed << 20) | 0x3ff0000000000001ull;                     \
    u64 z = 0x3ff0000000000000ull + seed;                                             \
    _Pragma("unroll") for (int i = 0; i < NV; ++i)                                    \
        x[i] = 0x3ff0000000000000ull + threadIdx.x * 977u + i * 131u + seed;          \
    for (int it = 0; it < iters; ++it) {                                              \
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                          \
            _Pragma("unroll") for (int i = 0; i < NV; ++i)                            \
                asm volatile(ASMSTR : "+v"(x[i]) : "v"(y), "v"(z) : CLOB);            \
        }                                                                             \
    }                                                                                 \
    u64 s = 0; _Pragma("unroll") for (int i = 0; i < NV; ++i) s ^= x[i];              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);             \
}

RAW64_KERNEL(v_fma_f64,       "v_fma_f64 %0, %0, %1, %2", "memory")
RAW64_KERNEL(v_mul_f64,       "v_mul_f64 %0, %0, %1", "memory")
RAW64_KERNEL(v_add_f64,       "v_add_f64 %0, %0, %1", "memory")
RAW64_KERNEL(v_rndne_f64,     "v_rndne_f64 %0, %0", "memory")
RAW64_KERNEL(v_lshlrev_b64,   "v_lshlrev_b64 %0, 3, %0", "memory")
RAW64_KERNEL(v_lshl_add_u64,  "v_lshl_add_u64 %0, %0, 0, %1", "memory")
RAW64_KERNEL(v_pk_fma_f32,    "v_pk_fma_f32 %0, %0, %1, %2", "memory")
RAW64_KERNEL(v_pk_mul_f32,    "v_pk_mul_f32 %0, %0, %1", "memory")
RAW64_KERNEL(v_pk_add_f32,    "v_pk_add_f32 %0, %0, %1", "memory")

// 32x32+64 -> 64 with carry out in vcc.
__global__ void __launch_bounds__(256) raw_v_mad_u64_u32(u32* out, int iters, u32 seed) {
    u64 x[NV]; u32 y = seed | 1u, z = seed * 2654435761u + 12345u;
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = threadIdx.x * 977u + i * 131u + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(y), "v"(z) : "vcc");
        }
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}

// 64-bit add as carry pair.
__global__ void __launch_bounds__(256) raw_add64_pair(u32* out, int iters, u32 seed) {
    u64 x[NV]; u64 y = ((u64)seed << 32) | 0x9e3779b9u;
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = threadIdx.x * 977u + i * 131u + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                u32 lo = (u32)x[i], hi = (u32)(x[i] >> 32);
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                             : "+v"(lo), "+v"(hi) : "v"((u32)y), "v"((u32)(y >> 32)) : "vcc");
                x[i] = ((u64)hi << 32) | lo;
            }
        }
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}

// ------------------------------------------------------------------ butterflies
// Each returns (a,b) <- (a + w*b, a - w*b) mod p in its own representation.

// (M) 64-bit Montgomery, p < 2^62, values in [0,p).
struct Mont64 {
    static constexpr u64 P = 0x3fffffffffc80001ull;  // 2^62 - 57*2^16 + 1 shape; primality is irrelevant to timing
    static constexpr u64 PINV_NEG = 0x3fffffffffc7ffffull;  // placeholder constant, timing only
    __device__ static inline u64 mul(u64 a, u64 b) {
        u128 t = (u128)a * b;
        u64 m = (u64)t * PINV_NEG;
        u128 u = t + (u128)m * P;
        u64 r = (u64)(u >> 64);
        return r >= P ? r - P : r;
    }
    __device__ static inline void bfly(u64& a, u64& b, u64 w) {
        u64 t = mul(b, w);
        u64 s = a + t; s = s >= P ? s - P : s;
        u64 d = a >= t ? a - t : a + P - t;
        a = s; b = d;
    }
};

// (G) Goldilocks p = 2^64 - 2^32 + 1, values in [0, 2^64) (non-canonical allowed).
struct Gold64 {
    static constexpr u64 EPS = 0xffffffffull;  // 2^32 - 1 = 2^64 mod p
    __device__ static inline u64 reduce128(u128 x) {
        u64 lo = (u64)x, hi = (u64)(x >> 64);
        u32 hh = (u32)(hi >> 32), hl = (u32)hi;
        u64 t = lo - hh;
        if (lo < (u64)hh) t -= EPS;          // borrow: wrapped by 2^64 == EPS
        u64 m = ((u64)hl << 32) - hl;         // hl * (2^32 - 1)
        u64 r = t + m;
        if (r < m) r += EPS;                  // carry
        return r;
    }
    __device__ static inline u64 mul(u64 a, u64 b) { return reduce128((u128)a * b); }
    __device__ static inline void bfly(u64& a, u64& b, u64 w) {
        u64 t = mul(b, w);
        u64 s = a + t; if (s < t) s += EPS;
        u64 d = a - t; if (a < t) d -= EPS;
        a = s; b = d;
    }
};

// (S) 32-bit Shoup/Harvey lazy butterfly, p < 2^30, values in [0,4p).
struct Shoup32 {
    static constexpr u32 P = 0x3fffd001u;  // shape only
    __device__ static inline void bfly(u32& a, u32& b, u32 w, u32 wp) {
        u32 q = __umulhi(b, wp);
        u32 t = b * w - q * P;                 // [0,2p)
        u32 a2 = a - 2 * P; a = a2 < a ? a2 : a;  // min(a, a-2p) unsigned -> [0,2p)
        b = a - t + 2 * P;
        a = a + t;
    }
};

// (U) 24-bit multiplier path, p < 2^22, values in [0,4p) < 2^24.
__device__ static inline u32 mul_u24(u32 a, u32 b) { u32 r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ static inline u32 mulhi_u24(u32 a, u32 b) { u32 r; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
struct U24 {
    static constexpr u32 P = 0x3fd001u;
    __device__ static inline void bfly(u32& a, u32& b, u32 w, u32 wp) {
        u32 lo = mul_u24(b, wp);
        u32 hi = mulhi_u24(b, wp);
        u32 q = __builtin_amdgcn_alignbit(hi, lo, 24);
        u32 t = mul_u24(b, w) - mul_u24(q, P);
        u32 a2 = a - 2 * P; a = a2 < a ? a2 : a;
        b = a - t + 2 * P;
        a = a + t;
    }
};

#pragma clang fp contract(off)
// (F) centred f32, p < 2^21; exact as long as |values| < 2^24.
struct F32C {
    static constexpr float P = 2093057.0f, PINV = 1.0f / 2093057.0f;
    __device__ static inline void bfly(float& a, float& b, float w, float wp) {
        float h = b * w;
        float l = __builtin_fmaf(b, w, -h);
        float q = __builtin_rintf(b * wp);
        float t = __builtin_fmaf(-q, P, h) + l;
        float s = a + t, d = a - t;
        a = s; b = d;
    }
};

// (D) centred f64, p < 2^50; exact as long as |values| < 2^53.
struct F64C {
    static constexpr double P = 1125899906826241.0, PINV = 1.0 / 1125899906826241.0;
    __device__ static inline void bfly(double& a, double& b, double w, double wp) {
        double h = b * w;
        double l = __builtin_fma(b, w, -h);
        double q = __builtin_rint(b * wp);
        double t = __builtin_fma(-q, P, h) + l;
        double s = a + t, d = a - t;
        a = s; b = d;
    }
};

template <class A, class T>
__global__ void __launch_bounds__(256) bfly1(u32* out, int iters, u32 seed) {  // twiddle only
    T a[NV], b[NV];
    T w = (T)(seed | 1u);
#pragma unroll
    for (int i = 0; i < NV; ++i) { a[i] = (T)(threadIdx.x * 977u + i * 131u + seed); b[i] = (T)(i * 7u + threadIdx.x + 3u); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < NV; ++i) A::bfly(a[i], b[i], w);
            w = w + (T)2;
        }
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s = s + a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s;
}

template <class A, class T>
__global__ void __launch_bounds__(256) bfly2(u32* out, int iters, u32 seed) {  // twiddle + Shoup companion
    T a[NV], b[NV];
    T w = (T)(seed | 1u), wp = (T)((seed >> 3) | 1u);
    if constexpr (sizeof(T) == 4 && !__is_same(T, u32)) { wp = (T)1e-7f * (T)seed; }
    if constexpr (sizeof(T) == 8) { wp = (T)1e-16 * (T)seed; }
#pragma unroll
    for (int i = 0; i < NV; ++i) { a[i] = (T)((threadIdx.x * 977u + i * 131u + seed) & 0xfffffu); b[i] = (T)((i * 7u + threadIdx.x + 3u) & 0xfffffu); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < NV; ++i) A::bfly(a[i], b[i], w, wp);
            w = w + (T)2;
        }
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s = s + a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s;
}

// ------------------------------------------------------------------ driver
typedef void (*kern_t)(u32*, int, u32);

static int time_kernel(const char* name, kern_t k, u32* d_out, int blocks, int iters, double* gops) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, iters / 8, 12345u);  // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u + rep);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double ops = (double)blocks * 256.0 * iters * UNROLL * NV;
    *gops = ops / (best * 1e-3) / 1e9;
    (void)name;
    return 0;
}

int main(int argc, char** argv) {
    int wg_per_cu = argc > 1 ? atoi(argv[1]) : 8;
    int iters = argc > 2 ? atoi(argv[2]) : 4096;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * wg_per_cu;
    printf("# device %s, %d CUs, clock %d kHz, blocks=%d (x256 thr), iters=%d\n", prop.name, cus, prop.clockRate, blocks, iters);
    u32* d_out; CK(hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(u32)));

    struct Item { const char* name; kern_t k; };
    std::vector<Item> items = {
#define R(N) {#N, raw_##N}
        R(v_add_u32), R(v_add3_u32), R(v_xor_b32), R(v_min_u32), R(v_cndmask_b32), R(v_alignbit_b32),
        R(v_mul_lo_u32), R(v_mul_hi_u32), R(v_mad_u64_u32),
        R(v_mul_u32_u24), R(v_mul_hi_u32_u24), R(v_mad_u32_u24), R(v_mad_i32_i24),
        R(v_mad_u32_u16), R(v_pk_mul_lo_u16), R(v_pk_mad_u16), R(v_dot4_u32_u8),
        R(v_fma_f32), R(v_mul_f32), R(v_rndne_f32), R(v_cvt_f32_u32), R(v_cvt_u32_f32),
        R(v_pk_fma_f32), R(v_pk_mul_f32), R(v_pk_add_f32),
        R(v_fma_f64), R(v_mul_f64), R(v_add_f64), R(v_rndne_f64),
        R(v_lshlrev_b64), R(v_lshl_add_u64), R(add64_pair),
#undef R
        {"bfly_mont64", bfly1<Mont64, u64>},
        {"bfly_gold64", bfly1<Gold64, u64>},
        {"bfly_shoup32", bfly2<Shoup32, u32>},
        {"bfly_u24", bfly2<U24, u32>},
        {"bfly_f32c", bfly2<F32C, float>},
        {"bfly_f64c", bfly2<F64C, double>},
    };
    double base = 0;
    for (auto& it : items) {
        double g;
        if (time_kernel(it.name, it.k, d_out, blocks, iters, &g)) return 1;
        if (base == 0) base = g;  // v_add_u32 = 1 slot
        printf("%-20s %10.1f Gop/s  %6.2f slots\n", it.name, g, base / g);
        fflush(stdout);
    }
    CK(hipFree(d_out));
    return 0;
}
