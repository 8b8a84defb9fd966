// LWE key switch for large batches as an integer matrix product on the matrix cores.
//
//   out[ct] = (0, ..., 0, b[ct]) - sum_k d[ct][k] * KSK[k]        k = (coefficient i, level lv)
//
// is, for a batch, the product  D (count x K, signed digits)  x  KSK (K x (n+1), 64-bit words) mod 2^64.
// Every key word is cut into eight balanced signed bytes (w = sum_b L_b 256^b mod 2^64, L_b in
// [-128, 127]), which makes it eight int8 x int8 -> int32 products
//   C[ct][b * cols_pad + col] = sum_k d[ct][k] * L_b[k][col]      (|C| <= K * 64 * 128 < 2^31, exact)
// run on v_mfma_i32_32x32x32_i8, and a recombination  sum_b C_b * 256^b  in wrapping 64-bit
// arithmetic.  The result is bit-identical to the VALU kernel's (k_keyswitch): both are exact.
#include <hip/hip_runtime.h>

#include "engine.h"

namespace lfhe {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int KSM_TILE = 128;     // C tile per workgroup, both ways
constexpr int KSM_KB = 64;        // bytes of k per LDS stage
constexpr int KSM_ROW = KSM_KB + 16;   // LDS row stride: 16-byte reads of 8 consecutive rows cover all 32 banks once

// ---- key preparation: limbs[n = b * cols_pad + col][k], k contiguous
__global__ void __launch_bounds__(256) k_ksm_key_limbs(const u64* __restrict__ ksk, int8_t* __restrict__ limbs, int K,
                                                       int cols, int cols_pad) {
    // one thread per (col, 16 consecutive k): sixteen strided key reads, eight 16-byte limb stores
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kchunks = K / 16;
    if (id >= (size_t)cols_pad * kchunks) return;
    const int col = (int)(id % cols_pad), k0 = (int)(id / cols_pad) * 16;
    int8_t l[8][16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        u64 w = col < cols ? ksk[(size_t)(k0 + j) * cols + col] : 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int8_t v = (int8_t)(w & 0xff);
            l[b][j] = v;
            w = (w - (u64)(i64)v) >> 8;   // exact: the low byte is now zero; the carry of a negative limb moves up
        }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b)
        *reinterpret_cast<v4i*>(limbs + ((size_t)b * cols_pad + col) * K + k0) = *reinterpret_cast<const v4i*>(l[b]);
}

// ---- digits of a batch: D[ct][i * levels + lv] (signed), rows count..m_pad-1 zero
// (`scale` multiplies every input word first: 1 for the key switch of the bootstrap)
__global__ void __launch_bounds__(256) k_ksm_digits(const u64* __restrict__ in, int8_t* __restrict__ D, int count,
                                                    int m_pad, int big, int base_log, int levels, BlockMap imap,
                                                    u64 scale) {
    const size_t total = (size_t)m_pad * big;
    const int rep = base_log * levels;
    const u64 mask = (1ull << base_log) - 1;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(id % big), ct = (int)(id / big);
        int8_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ct < count) {
            const u64 x = in[imap.block(ct) * (size_t)(big + 1) + i] * scale;
            u64 state = (((x >> (64 - rep - 1)) + 1) >> 1) & ((1ull << rep) - 1);
            for (int lv = levels - 1; lv >= 0; --lv) {   // lv = 0 most significant
                const u64 o = state & mask;
                state >>= base_log;
                const u64 carry = (((o - 1) | state) & o) >> (base_log - 1);
                state += carry;
                d[lv] = (int8_t)((int)o - (int)(carry << base_log));
            }
        }
        int8_t* dst = D + (size_t)ct * big * levels + (size_t)i * levels;
        for (int lv = 0; lv < levels; ++lv) dst[lv] = d[lv];
    }
}

// ---- C[m][n] = sum_k D[m][k] * limbs[n][k]; grid (N / 128, M / 128), 256 threads = 2 x 2 waves of 64 x 64
__global__ void __launch_bounds__(256, 2) k_ksm_gemm(const int8_t* __restrict__ D, const int8_t* __restrict__ L,
                                                     int* __restrict__ C, int K, int N) {
    __shared__ __attribute__((aligned(16))) int8_t As[KSM_TILE * KSM_ROW];
    __shared__ __attribute__((aligned(16))) int8_t Bs[KSM_TILE * KSM_ROW];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const size_t m0 = (size_t)blockIdx.y * KSM_TILE, n0 = (size_t)blockIdx.x * KSM_TILE;

    // staging: two 16-byte pieces of each tile per thread and stage
    const int srow0 = t >> 2, sseg = t & 3;          // rows srow0 and srow0 + 64, bytes 16 sseg..
    const int8_t* ga0 = D + (m0 + srow0) * (size_t)K + sseg * 16;
    const int8_t* ga1 = ga0 + (size_t)64 * K;
    const int8_t* gb0 = L + (n0 + srow0) * (size_t)K + sseg * 16;
    const int8_t* gb1 = gb0 + (size_t)64 * K;
    const int so0 = srow0 * KSM_ROW + sseg * 16, so1 = so0 + 64 * KSM_ROW;

    v16i acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    v4i ra0 = *reinterpret_cast<const v4i*>(ga0), ra1 = *reinterpret_cast<const v4i*>(ga1);
    v4i rb0 = *reinterpret_cast<const v4i*>(gb0), rb1 = *reinterpret_cast<const v4i*>(gb1);
    for (int k0 = 0; k0 < K; k0 += KSM_KB) {
        __syncthreads();   // the previous stage has been read
        *reinterpret_cast<v4i*>(As + so0) = ra0; *reinterpret_cast<v4i*>(As + so1) = ra1;
        *reinterpret_cast<v4i*>(Bs + so0) = rb0; *reinterpret_cast<v4i*>(Bs + so1) = rb1;
        __syncthreads();
        if (k0 + KSM_KB < K) {   // next stage in flight during the products
            ra0 = *reinterpret_cast<const v4i*>(ga0 + k0 + KSM_KB); ra1 = *reinterpret_cast<const v4i*>(ga1 + k0 + KSM_KB);
            rb0 = *reinterpret_cast<const v4i*>(gb0 + k0 + KSM_KB); rb1 = *reinterpret_cast<const v4i*>(gb1 + k0 + KSM_KB);
        }
#pragma unroll
        for (int s = 0; s < KSM_KB / 32; ++s) {
            // lane (r, h) holds bytes k = 32 s + 16 h .. + 15 of row r of its tile, for both operands:
            // whatever order the instruction gives the 32 k positions, both sides agree on it
            v4i a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const v4i*>(As + (wm * 64 + i * 32 + r) * KSM_ROW + s * 32 + h * 16);
                b[i] = *reinterpret_cast<const v4i*>(Bs + (wn * 64 + i * 32 + r) * KSM_ROW + s * 32 + h * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // result element e of a lane: column lane & 31, row (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const size_t row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                C[row * (size_t)N + col] = acc[i][j][e];
            }
        }
}

// The same product for large batches with 256 x 256 tiles: 512 threads = 2 x 4 waves of 128 x 64.
// With 128 x 128 tiles the kernel is bound by what the tiles re-read through L2 (3.9 GB per 4096
// ciphertexts at 8.4 TB/s); tiles of twice the side read half of it.  One workgroup per CU (128
// accumulator registers per lane), two waves per SIMD.
constexpr int KSM_BIG = 256;
__global__ void __launch_bounds__(512, 1) k_ksm_gemm256(const int8_t* __restrict__ D, const int8_t* __restrict__ L,
                                                        int* __restrict__ C, int K, int N) {
    __shared__ __attribute__((aligned(16))) int8_t As[KSM_BIG * KSM_ROW];
    __shared__ __attribute__((aligned(16))) int8_t Bs[KSM_BIG * KSM_ROW];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const size_t m0 = (size_t)blockIdx.y * KSM_BIG, n0 = (size_t)blockIdx.x * KSM_BIG;

    // staging: two 16-byte pieces of each tile per thread and stage (rows srow0 and srow0 + 128)
    const int srow0 = t >> 2, sseg = t & 3;
    const int8_t* ga0 = D + (m0 + srow0) * (size_t)K + sseg * 16;
    const int8_t* ga1 = ga0 + (size_t)128 * K;
    const int8_t* gb0 = L + (n0 + srow0) * (size_t)K + sseg * 16;
    const int8_t* gb1 = gb0 + (size_t)128 * K;
    const int so0 = srow0 * KSM_ROW + sseg * 16, so1 = so0 + 128 * KSM_ROW;

    v16i acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    v4i ra0 = *reinterpret_cast<const v4i*>(ga0), ra1 = *reinterpret_cast<const v4i*>(ga1);
    v4i rb0 = *reinterpret_cast<const v4i*>(gb0), rb1 = *reinterpret_cast<const v4i*>(gb1);
    for (int k0 = 0; k0 < K; k0 += KSM_KB) {
        __syncthreads();   // the previous stage has been read
        *reinterpret_cast<v4i*>(As + so0) = ra0; *reinterpret_cast<v4i*>(As + so1) = ra1;
        *reinterpret_cast<v4i*>(Bs + so0) = rb0; *reinterpret_cast<v4i*>(Bs + so1) = rb1;
        __syncthreads();
        if (k0 + KSM_KB < K) {   // next stage in flight during the products
            ra0 = *reinterpret_cast<const v4i*>(ga0 + k0 + KSM_KB); ra1 = *reinterpret_cast<const v4i*>(ga1 + k0 + KSM_KB);
            rb0 = *reinterpret_cast<const v4i*>(gb0 + k0 + KSM_KB); rb1 = *reinterpret_cast<const v4i*>(gb1 + k0 + KSM_KB);
        }
#pragma unroll
        for (int s = 0; s < KSM_KB / 32; ++s) {
            v4i a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const v4i*>(As + (wm * 128 + i * 32 + r) * KSM_ROW + s * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j] = *reinterpret_cast<const v4i*>(Bs + (wn * 64 + j * 32 + r) * KSM_ROW + s * 32 + h * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const size_t row = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                C[row * (size_t)N + col] = acc[i][j][e];
            }
        }
}

// picks the tile size: the large tiles need whole tiles both ways and enough of them to fill the GPU
static void launch_ksm_gemm(const int8_t* dig, const int8_t* limbs, int* prod, int M, int K, int N, hipStream_t st) {
    if (M % KSM_BIG == 0 && N % KSM_BIG == 0 && (size_t)(M / KSM_BIG) * (N / KSM_BIG) >= 256)
        hipLaunchKernelGGL(k_ksm_gemm256, dim3(N / KSM_BIG, M / KSM_BIG), dim3(512), 0, st, dig, limbs, prod, K, N);
    else
        hipLaunchKernelGGL(k_ksm_gemm, dim3(N / KSM_TILE, M / KSM_TILE), dim3(256), 0, st, dig, limbs, prod, K, N);
}

// ---- out[ct][col] = (col == n ? b : 0) - sum_b C[ct][b * cols_pad + col] * 256^b
__global__ void __launch_bounds__(256) k_ksm_combine(const int* __restrict__ C, const u64* __restrict__ in,
                                                     u64* __restrict__ out, int count, int big, int n, int N, int cols_pad,
                                                     BlockMap imap) {
    const int cols = n + 1;
    const size_t total = (size_t)count * cols;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(id % cols), ct = (int)(id / cols);
        const int* c = C + (size_t)ct * N + col;
        u64 s = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) s += (u64)(i64)c[(size_t)b * cols_pad] << (8 * b);
        const u64 body = col == n ? in[imap.block(ct) * (size_t)(big + 1) + big] : 0;
        out[id] = body - s;
    }
}

// ---- host side
size_t ksm_cols_pad(int n) { return (size_t)((n + 1 + 15) / 16) * 16; }
size_t ksm_limb_bytes(int big, int levels, int n) { return 8 * ksm_cols_pad(n) * (size_t)big * levels; }
size_t ksm_rows_pad(size_t count) { return (count + KSM_TILE - 1) / KSM_TILE * KSM_TILE; }
size_t ksm_digit_bytes(size_t count, int big, int levels) { return ksm_rows_pad(count) * (size_t)big * levels; }
size_t ksm_product_bytes(size_t count, int n) { return ksm_rows_pad(count) * 8 * ksm_cols_pad(n) * sizeof(int); }
bool ksm_supported(int big, int levels, int n) {
    // the product kernel walks K in stages of 64 and N in tiles of 128
    return ((size_t)big * levels) % KSM_KB == 0 && (8 * ksm_cols_pad(n)) % KSM_TILE == 0;
}

void launch_ksm_key_limbs(const u64* ksk, int8_t* limbs, int big, int levels, int n, hipStream_t st) {
    const int K = big * levels;
    const size_t threads = ksm_cols_pad(n) * (size_t)(K / 16);
    hipLaunchKernelGGL(k_ksm_key_limbs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, ksk, limbs, K, n + 1,
                       (int)ksm_cols_pad(n));
}

void launch_keyswitch_mfma(const int8_t* limbs, const u64* in, int8_t* dig, int* prod, u64* out, size_t count, int big,
                           int n, int base_log, int levels, const BlockMap& imap, hipStream_t st) {
    const int K = big * levels, N = (int)(8 * ksm_cols_pad(n)), M = (int)ksm_rows_pad(count);
    size_t dthreads = (size_t)M * big;
    unsigned dblocks = (unsigned)((dthreads + 255) / 256);
    if (dblocks > 16384) dblocks = 16384;
    hipLaunchKernelGGL(k_ksm_digits, dim3(dblocks), dim3(256), 0, st, in, dig, (int)count, M, big, base_log, levels, imap,
                       (u64)1);
    launch_ksm_gemm(dig, limbs, prod, M, K, N, st);
    const size_t total = count * (size_t)(n + 1);
    unsigned cblocks = (unsigned)((total + 255) / 256);
    if (cblocks > 16384) cblocks = 16384;
    hipLaunchKernelGGL(k_ksm_combine, dim3(cblocks), dim3(256), 0, st, prod, in, out, (int)count, big, n, N,
                       (int)ksm_cols_pad(n), imap);
}

// ------------------------------------------------------------------------------------------------
// Ciphertext compression: packing key switch of up to `per` = N_c blocks into one GLWE ciphertext
// of dimension k_c and polynomial size N_c, then modulus switch to 2^storage_log.
//   G = sum_j X^j * ( (0, .., 0, s b_j) - sum_k d[j][k] * PKSK[k] ),   d = digits of s a_j,
// s = `scale` (the message modulus: moves the message up into the carry bits, which the storage
// modulus keeps).  The inner sums are the same int8 matrix product as the key switch above with the
// (k_c + 1) N_c words of a GLWE ciphertext as columns; k_pack_combine rotates, adds and rounds.
// Decompression: k_unpack_extract turns coefficient j of a stored GLWE into an LWE ciphertext of
// dimension k_c N_c (scaled back to 2^64); a bootstrap under the decompression key does the rest.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_combine(const int* __restrict__ C, const u64* __restrict__ in,
                                                      uint16_t* __restrict__ out, int count, int big, int kc, int nc,
                                                      int N, int cols_pad, int storage_log, u64 scale, BlockMap imap) {
    // one workgroup per (GLWE, polynomial p); thread = coefficient c
    const int g = blockIdx.x, p = blockIdx.y, c = threadIdx.x;
    if (c >= nc) return;
    const int first = g * nc;
    const int cnt = min(nc, count - first);
    u64 acc = 0;
    for (int j = 0; j < cnt; ++j) {
        // coefficient c of X^j * P is P[c - j], negated when it wrapped
        const int src = c - j;
        const bool wrap = src < 0;
        const int* row = C + (size_t)(first + j) * N + p * nc + (wrap ? src + nc : src);
        u64 v = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) v += (u64)(i64)row[(size_t)b * cols_pad] << (8 * b);
        acc += wrap ? v : (u64)0 - v;      // the key switch subtracts
    }
    if (p == kc && c < cnt) acc += in[imap.block(first + c) * (size_t)(big + 1) + big] * scale;
    const u64 r = ((acc >> (64 - storage_log - 1)) + 1) >> 1;
    out[((size_t)g * (kc + 1) + p) * nc + c] = (uint16_t)(r & ((1u << storage_log) - 1));
}

// small[ct][kc nc + 1]: sample extraction of coefficient ct % nc of GLWE ct / nc, words scaled to 2^64
__global__ void __launch_bounds__(256) k_unpack_extract(const uint16_t* __restrict__ glwe, u64* __restrict__ small, int count,
                                                        int kc, int nc, int storage_log) {
    const int ct = blockIdx.x;
    const int g = ct / nc, j = ct % nc;
    const uint16_t* G = glwe + (size_t)g * (kc + 1) * nc;
    u64* o = small + (size_t)ct * ((size_t)kc * nc + 1);
    const int sh = 64 - storage_log;
    for (int w = threadIdx.x; w < kc * nc; w += blockDim.x) {
        const int p = w / nc, i = w % nc;
        const int src = j - i;
        const u64 v = (u64)G[p * nc + (src < 0 ? src + nc : src)] << sh;
        o[w] = src < 0 ? (u64)0 - v : v;
    }
    if (threadIdx.x == 0) o[(size_t)kc * nc] = (u64)G[kc * nc + j] << sh;
}

void launch_pack(const int8_t* limbs, const u64* in, int8_t* dig, int* prod, uint16_t* out, size_t count, int big, int kc,
                 int nc, int base_log, int levels, int storage_log, u64 scale, const BlockMap& imap, hipStream_t st) {
    const int cols = (kc + 1) * nc;                    // a multiple of 16 for every supported shape
    const int K = big * levels, N = 8 * cols, M = (int)ksm_rows_pad(count);
    size_t dthreads = (size_t)M * big;
    unsigned dblocks = (unsigned)((dthreads + 255) / 256);
    if (dblocks > 16384) dblocks = 16384;
    hipLaunchKernelGGL(k_ksm_digits, dim3(dblocks), dim3(256), 0, st, in, dig, (int)count, M, big, base_log, levels, imap,
                       scale);
    launch_ksm_gemm(dig, limbs, prod, M, K, N, st);
    const unsigned groups = (unsigned)((count + nc - 1) / nc);
    hipLaunchKernelGGL(k_pack_combine, dim3(groups, kc + 1), dim3(256), 0, st, prod, in, out, (int)count, big, kc, nc, N,
                       cols, storage_log, scale, imap);
}

void launch_pack_key_limbs(const u64* pksk, int8_t* limbs, int big, int levels, int kc, int nc, hipStream_t st) {
    const int K = big * levels, cols = (kc + 1) * nc;
    const size_t threads = (size_t)cols * (K / 16);
    hipLaunchKernelGGL(k_ksm_key_limbs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, pksk, limbs, K, cols, cols);
}

void launch_unpack_extract(const uint16_t* glwe, u64* small, size_t count, int kc, int nc, int storage_log, hipStream_t st) {
    if (!count) return;
    hipLaunchKernelGGL(k_unpack_extract, dim3((unsigned)count), dim3(256), 0, st, glwe, small, (int)count, kc, nc, storage_log);
}

}  // namespace lfhe
