#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include "tables.h"

namespace lfhe {

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
#define HC(x) hip_check((x), #x)

Engine::Engine(int device, const EngineParams& p) : device_(device), prm_(p) {
    if (p.poly_size != (uint32_t)N || p.glwe_dim != 1 || p.pbs_level != 1 || p.pbs_base_log != 23)
        throw std::runtime_error(
            "unsupported parameters: the kernels are built for N=2048, k=1, one PBS level of base 2^23");
    if (p.ks_level < 1 || p.ks_level > 4 || p.ks_base_log < 1 || p.ks_base_log > 7 ||
        p.ks_base_log * p.ks_level > 32)
        throw std::runtime_error("unsupported key-switch decomposition");
    if (p.lwe_dim < 1 || p.lwe_dim > 4096) throw std::runtime_error("unsupported LWE dimension");
    if (p.bsk_modulus != 0 && p.bsk_modulus != QP_U)
        throw std::runtime_error("unsupported bootstrapping-key modulus: 0 (2^64) or 703687441727489");
    if (p.bsk_grouping < 1 || p.bsk_grouping > 3) throw std::runtime_error("unsupported key grouping: 1, 2 or 3");
    if (p.ms_zeros_count && (p.ms_bound_log2 < 27 || p.ms_bound_log2 > 63 || !(p.ms_r_sigma_factor > 0) ||
                             !(p.ms_r_sigma_factor < 1024) || !(p.ms_input_variance >= 0) ||
                             !(p.ms_input_variance < 1e-3)))
        throw std::runtime_error("unsupported modulus-switch noise reduction parameters");
    int ndev = 0;
    HC(hipGetDeviceCount(&ndev));
    if (ndev <= 0) throw std::runtime_error("no HIP device: the HIP path is the only path");
    if (device < 0 || device >= ndev) throw std::runtime_error("device index out of range");
    HC(hipSetDevice(device_));
    blind_rotate_setup(device_);
    HC(hipSetDevice(device_));
    HC(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    HC(hipGetDeviceProperties(&prop, device_));
    cu_count_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // the 512-thread kernels hold one ciphertext per CU at a time (grouped key: 207 registers) or two
    // (one key bit per step); measured with the grouped key: 4.2 vs 4.5 ms at 64 ciphertexts, 4.6 vs 4.8 at
    // 256, but 8.8 vs 7.4 ms at 512
    wide_below_ = (p.bsk_grouping >= 2 ? 1 : 2) * cu_count_;
    if (const char* e = std::getenv("LFHE_WIDE_BELOW")) wide_below_ = std::atoi(e);   // tuning / A-B runs
    pair_above_ = -2;   // automatic (pair_threshold)
    lockstep_above_ = 4 * cu_count_;
    if (const char* e = std::getenv("LFHE_LOCKSTEP_ABOVE")) lockstep_above_ = std::atoi(e);
    if (const char* e = std::getenv("LFHE_PAIR_ABOVE")) pair_above_ = std::atoi(e);   // -1: never
    if (const char* e = std::getenv("LFHE_BR_CHUNK")) br_chunk_ = std::atoi(e);       // 0: automatic
    if (const char* e = std::getenv("LFHE_KS_MFMA_MIN")) ksm_min_ = std::atoi(e);     // 0: VALU key switch only
    if (!ksm_supported(N, (int)p.ks_level, (int)p.lwe_dim)) ksm_min_ = 0;

    if (!single_modulus() && p.bsk_grouping >= 2) {
        // lift guard of the grouped key over 2^64 (kernels.hip: k_blind_rotate_g64_pair): marks of one launch and
        // the count of ciphertexts that took the slow path, made here so that no bootstrap call allocates
        g64_guard_.capacity = 65536;
        HC(hipMalloc(&g64_guard_.fault, g64_guard_.capacity * sizeof(int)));
        HC(hipMalloc(&g64_guard_.fallbacks, sizeof(unsigned long long)));
        HC(hipMemset(g64_guard_.fallbacks, 0, sizeof(unsigned long long)));
        double frac = 0.875;      // of the half-window p1 / 2
        if (const char* e = std::getenv("LFHE_G64_GUARD_FRAC")) frac = std::atof(e);   // tests force the slow path with a small one
        g64_guard_.k_guard = frac * 0.5 * P1;
        // launches of at most one ciphertext per two CUs spread each ciphertext's two primes over two CUs
        size_t split = (size_t)cu_count_ / 2;
        if (const char* e = std::getenv("LFHE_G64_SPLIT_BELOW")) split = (size_t)std::max(0, std::atoi(e));   // 0: never
        if (const char* e = std::getenv("LFHE_G64_SPLIT_POLL")) g64_guard_.poll_limit = (unsigned)std::max(0, std::atoi(e));   // tests: 0 makes every workgroup give up
        if (split) {
            HC(hipMalloc(&g64_guard_.xchg, g64_split_payload_bytes(split)));
            HC(hipMalloc(&g64_guard_.xchg_flags, g64_split_flag_bytes(split)));
            g64_guard_.split_below = split;
        }
    }

    HostTables H;
    build_host_tables(H);
    HC(hipMalloc(&d_tw_, sizeof(double) * (4 * N + 2 * (MONO_WORDS + 8))));
    for (int r = 0; r < 2; ++r) {
        HC(hipMemcpy(d_tw_ + (2 * r) * N, H.tw_fwd[r].data(), sizeof(double) * N, hipMemcpyHostToDevice));
        HC(hipMemcpy(d_tw_ + (2 * r + 1) * N, H.tw_inv[r].data(), sizeof(double) * N, hipMemcpyHostToDevice));
        T_.tw_fwd[r] = d_tw_ + (2 * r) * N;
        T_.tw_inv[r] = d_tw_ + (2 * r + 1) * N;
        T_.pr[r].p = (double)H.p[r];
        T_.pr[r].pinv = 1.0 / (double)H.p[r];
        T_.two32[r] = H.two32[r];
        T_.ninv[r] = H.ninv[r];
    }
    T_.p0inv_mod_p1 = H.p0inv_mod_p1;
    for (int r = 0; r < 2; ++r) {   // psi^k over prime r for every exponent k of X^k (k < 2N), then the 8th roots of unity
        std::vector<double> mono(MONO_WORDS + 8);
        u64 w = 1;
        for (int k = 0; k < MONO_WORDS; ++k) {
            mono[k] = h_centred(w, H.p[r]);
            if (k % (N / 4) == 0) mono[MONO_WORDS + k / (N / 4)] = h_centred(w, H.p[r]);
            w = h_mulmod(w, H.psi[r], H.p[r]);
        }
        double* dst = d_tw_ + 4 * N + r * (MONO_WORDS + 8);
        HC(hipMemcpy(dst, mono.data(), sizeof(double) * mono.size(), hipMemcpyHostToDevice));
        (r == 0 ? T_.mono : T_.mono1) = dst;
    }
}

Engine::~Engine() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    noalloc_release(device_, stream_);
    if (own_keys_) { (void)hipFree(d_bsk_); (void)hipFree(d_ksk_); }
    (void)hipFree(d_cast_ksk_);
    (void)hipFree(d_tw_);
    (void)hipFree(d_small_);
    (void)hipFree(d_ksdig_);
    (void)hipFree(d_ksk_limbs_);
    (void)hipFree(d_ksm_dig_);
    (void)hipFree(d_ksm_prod_);
    (void)hipFree(d_pack_limbs_);
    (void)hipFree(d_decomp_bsk_);
    (void)hipFree(g64_guard_.fault);
    (void)hipFree(g64_guard_.fallbacks);
    (void)hipFree(g64_guard_.xchg);
    (void)hipFree(g64_guard_.xchg_flags);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void Engine::sync() {
    HC(hipSetDevice(device_));
    HC(hipStreamSynchronize(stream_));
}

uint64_t Engine::lift_fallbacks() {
    if (!g64_guard_.fallbacks) return 0;
    sync();
    unsigned long long v = 0;
    HC(hipMemcpy(&v, g64_guard_.fallbacks, sizeof v, hipMemcpyDeviceToHost));
    return v;
}

void Engine::load_bsk(const u64* host_bsk) {
    HC(hipSetDevice(device_));
    if (!own_keys_ || !d_bsk_) {
        if (d_bsk_ && !own_keys_) throw std::runtime_error("keys are attached, not owned");
        HC(hipMalloc(&d_bsk_, bsk_dev_bytes()));
        own_keys_ = true;
    }
    try {
        bsk_factored_ = build_bsk_image(host_bsk, bsk_ggsw_count(), d_bsk_);
    } catch (...) {
        // a refused key leaves a half-written image behind: the engine is without keys until the next load
        (void)hipFree(d_bsk_);
        d_bsk_ = nullptr;
        throw;
    }
    publish_key_flags();
}

// The count word of the key-switching key image (how many encryptions of zero follow) carries in its
// top bit whether the bootstrapping-key image is in factored form, so that an engine which attaches
// both images (received by a broadcast) knows without being told.
static constexpr u64 FACTORED_FLAG = 1ull << 63;
void Engine::publish_key_flags() {
    if (!d_ksk_ || !own_keys_) return;
    u64 word = 0;
    HC(hipMemcpy(&word, ms_zeros_dev() - 1, sizeof word, hipMemcpyDeviceToHost));
    word = (word & ~FACTORED_FLAG) | (bsk_factored_ ? FACTORED_FLAG : 0);
    HC(hipMemcpy(ms_zeros_dev() - 1, &word, sizeof word, hipMemcpyHostToDevice));
}

bool Engine::build_bsk_image(const u64* host_bsk, size_t total, double* dst) {
    // upload and transform in slabs of 64 GGSWs to bound the staging buffer
    const size_t slab = 64;
    const size_t row = single_modulus() ? BSKQ_ROW : BSK_ROW;
    u64* d_raw = nullptr;
    HC(hipMalloc(&d_raw, slab * 4 * N * sizeof(u64)));
    if (prm_.bsk_grouping == 3) {
        // three key bits per step: every slab of eight groups is transformed into a staging buffer and
        // written out in the factored layout its kernel streams (kernels.hip: k_bsk_factor3).  A key with a
        // non-invertible word (one in 10^8) has no such image: the caller has to come with another key.
        // With both primes (2^64) the staging rows hold both transforms and each prime is factored by itself.
        double* d_tmp = nullptr;
        int* d_flag = nullptr;
        HC(hipMalloc(&d_tmp, slab * row * sizeof(double)));
        HC(hipMalloc(&d_flag, sizeof(int)));
        HC(hipMemsetAsync(d_flag, 0, sizeof(int), stream_));
        for (size_t i0 = 0; i0 < total; i0 += slab) {
            const size_t cnt = (total - i0) < slab ? (total - i0) : slab;
            HC(hipMemcpyAsync(d_raw, host_bsk + i0 * 4 * N, cnt * 4 * N * sizeof(u64), hipMemcpyHostToDevice, stream_));
            if (single_modulus()) launch_bskq_prepare(T_, d_raw, d_tmp, cnt * 4, stream_);
            else launch_ntt_forward(T_, d_raw, d_tmp, cnt * 4, 1, stream_);
            launch_bsk_factor3(T_, d_tmp, dst + i0 * row, cnt / 8, d_flag, single_modulus() ? 1 : 2, stream_);
            HC(hipGetLastError());
            HC(hipStreamSynchronize(stream_));
        }
        int flag = 0;
        HC(hipMemcpy(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost));
        HC(hipFree(d_tmp));
        HC(hipFree(d_flag));
        HC(hipFree(d_raw));
        if (flag) throw std::runtime_error("unsupported key: a word of the bootstrapping key is not invertible modulo the prime "
                                           "(one key in 10^8, one in 5 * 10^7 over 2^64); generate another key");
        return true;
    }
    auto transform_all = [&]() {
        for (size_t i0 = 0; i0 < total; i0 += slab) {
            const size_t cnt = (total - i0) < slab ? (total - i0) : slab;
            HC(hipMemcpyAsync(d_raw, host_bsk + i0 * 4 * N, cnt * 4 * N * sizeof(u64), hipMemcpyHostToDevice, stream_));
            if (single_modulus()) launch_bskq_prepare(T_, d_raw, dst + i0 * row, cnt * 4, stream_);
            else launch_ntt_forward(T_, d_raw, dst + i0 * row, cnt * 4, 1, stream_);
            HC(hipGetLastError());
            HC(hipStreamSynchronize(stream_));
        }
    };
    transform_all();
    bool factored = false;
    const char* env = std::getenv("LFHE_BSK_FACTORED");   // 0: keep the plain image (tests of that path, A/B runs)
    const bool allow = !(env && std::atoi(env) == 0);
    if (prm_.bsk_grouping == 2 && allow) {
        // the factored form needs every K3 word invertible; a key with a zero word (one in 10^8 keys) keeps
        // the plain image and the kernels that go with it
        int* d_flag = reinterpret_cast<int*>(d_raw);
        int flag = 0;
        HC(hipMemsetAsync(d_flag, 0, sizeof(int), stream_));
        launch_bsk_factor(T_, dst, total / 4, row, d_flag, stream_);
        HC(hipGetLastError());
        HC(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, stream_));
        HC(hipStreamSynchronize(stream_));
        factored = flag == 0;
        if (!factored) transform_all();
    }
    HC(hipFree(d_raw));
    return factored;
}

void Engine::load_compression_keys(const CompShape& shape, const u64* host_pksk, const u64* host_decomp_bsk) {
    HC(hipSetDevice(device_));
    if (shape.glwe_words() % 16 || ((size_t)N * shape.ks_level) % 64 || shape.nc > 256 || shape.small_dim() > 4096 ||
        shape.storage_log != LOGN + 1)
        throw std::runtime_error("unsupported compression shape");
    comp_ = shape;
    const size_t K = (size_t)N * shape.ks_level, cols = (size_t)shape.glwe_words();
    if (!d_pack_limbs_) HC(hipMalloc(&d_pack_limbs_, 8 * cols * K));
    u64* d_raw = nullptr;
    HC(hipMalloc(&d_raw, shape.pksk_words() * sizeof(u64)));
    HC(hipMemcpyAsync(d_raw, host_pksk, shape.pksk_words() * sizeof(u64), hipMemcpyHostToDevice, stream_));
    launch_pack_key_limbs(d_raw, d_pack_limbs_, N, shape.ks_level, shape.kc, shape.nc, stream_);
    HC(hipGetLastError());
    HC(hipStreamSynchronize(stream_));
    HC(hipFree(d_raw));
    const size_t row = single_modulus() ? BSKQ_ROW : BSK_ROW;
    if (!d_decomp_bsk_) HC(hipMalloc(&d_decomp_bsk_, decomp_bsk_ggsw_count() * row * sizeof(double)));
    try {
        decomp_factored_ = build_bsk_image(host_decomp_bsk, decomp_bsk_ggsw_count(), d_decomp_bsk_);
    } catch (...) {
        (void)hipFree(d_decomp_bsk_);      // as in load_bsk: no compression keys until the next load
        d_decomp_bsk_ = nullptr;
        throw;
    }
}

void Engine::compress(const u64* blocks, size_t count, uint16_t* out, u64 scale) {
    if (!has_compression_keys()) throw std::runtime_error("no compression keys loaded");
    if (!count) return;
    HC(hipSetDevice(device_));
    // launches of at most 4096 blocks (a multiple of the GLWE capacity), one scratch pair
    const size_t chunk = 4096;
    const size_t cap = count < chunk ? count : chunk;
    int8_t* dig = nullptr;
    int* prod = nullptr;
    HC(hipMalloc(&dig, ksm_rows_pad(cap) * (size_t)N * comp_.ks_level));
    HC(hipMalloc(&prod, ksm_rows_pad(cap) * 8 * (size_t)comp_.glwe_words() * sizeof(int)));
    for (size_t c0 = 0; c0 < count; c0 += chunk) {
        const size_t cnt = count - c0 < chunk ? count - c0 : chunk;
        launch_pack(d_pack_limbs_, blocks + c0 * (size_t)(N + 1), dig, prod, out + (c0 / comp_.nc) * (size_t)comp_.glwe_words(),
                    cnt, N, comp_.kc, comp_.nc, comp_.ks_base_log, comp_.ks_level, comp_.storage_log, scale, BlockMap(), stream_);
        HC(hipGetLastError());
    }
    HC(hipStreamSynchronize(stream_));
    HC(hipFree(dig));
    HC(hipFree(prod));
}

void Engine::decompress(const uint16_t* glwe, size_t count, const u64* lut, u64* blocks_out) {
    if (!has_compression_keys()) throw std::runtime_error("no compression keys loaded");
    if (!count) return;
    HC(hipSetDevice(device_));
    u64* small = nullptr;
    HC(hipMalloc(&small, count * ((size_t)comp_.small_dim() + 1) * sizeof(u64)));
    launch_unpack_extract(glwe, small, count, comp_.kc, comp_.nc, comp_.storage_log, stream_);
    HC(hipGetLastError());
    blind_rotate_with(d_decomp_bsk_, (uint32_t)comp_.small_dim(), small, lut, nullptr, nullptr, blocks_out, count, BlockMap(),
                      decomp_factored_ ? 1 : 0);
    HC(hipStreamSynchronize(stream_));
    HC(hipFree(small));
}

void Engine::set_ms_params(u32 zcount) {
    ms_ = MsParams{};
    ms_.zcount = zcount;
    if (!zcount) return;
    ms_.var_in_fx = (u64)std::llround(std::ldexp(prm_.ms_input_variance, 76));
    ms_.r_fx = (u64)std::llround(std::ldexp(prm_.ms_r_sigma_factor, 16));
    ms_.bound_fx = 1ull << (prm_.ms_bound_log2 - 26);
    ms_.log2_2n = LOGN + 1;
}

void Engine::load_ksk(const u64* host_ksk, const u64* ms_zeros) {
    HC(hipSetDevice(device_));
    if (!d_ksk_) {
        if (d_bsk_ && !own_keys_) throw std::runtime_error("keys are attached, not owned");
        HC(hipMalloc(&d_ksk_, ksk_dev_bytes()));
        own_keys_ = true;
    }
    HC(hipMemcpyAsync(d_ksk_, host_ksk, ksk_words() * sizeof(u64), hipMemcpyHostToDevice, stream_));
    launch_ksk_colsum(d_ksk_, N * (int)prm_.ks_level, (int)prm_.lwe_dim + 1, 1 << (prm_.ks_base_log - 1),
                      d_ksk_ + ksk_words(), stream_);
    HC(hipGetLastError());
    const u64 zc = (ms_zeros && prm_.ms_zeros_count) ? prm_.ms_zeros_count : 0;
    HC(hipMemcpyAsync(ms_zeros_dev() - 1, &zc, sizeof zc, hipMemcpyHostToDevice, stream_));
    if (zc) HC(hipMemcpyAsync(ms_zeros_dev(), ms_zeros, ms_zeros_words() * sizeof(u64), hipMemcpyHostToDevice, stream_));
    else if (ms_zeros_words()) HC(hipMemsetAsync(ms_zeros_dev(), 0, ms_zeros_words() * sizeof(u64), stream_));
    build_ksk_limbs();
    HC(hipStreamSynchronize(stream_));
    set_ms_params((u32)zc);
    publish_key_flags();
}

void Engine::build_ksk_limbs() {
    if (!ksm_min_) return;
    if (!d_ksk_limbs_) HC(hipMalloc(&d_ksk_limbs_, ksm_limb_bytes(N, (int)prm_.ks_level, (int)prm_.lwe_dim)));
    launch_ksm_key_limbs(d_ksk_, d_ksk_limbs_, N, (int)prm_.ks_level, (int)prm_.lwe_dim, stream_);
    HC(hipGetLastError());
}

void Engine::attach_device_keys(double* bsk_dev, u64* ksk_dev_image) {
    HC(hipSetDevice(device_));
    if (own_keys_) {
        (void)hipFree(d_bsk_);
        (void)hipFree(d_ksk_);
        own_keys_ = false;
    }
    d_bsk_ = bsk_dev;
    d_ksk_ = ksk_dev_image;
    u64 zc = 0;
    HC(hipMemcpy(&zc, ms_zeros_dev() - 1, sizeof zc, hipMemcpyDeviceToHost));
    bsk_factored_ = (zc & FACTORED_FLAG) != 0;
    zc &= ~FACTORED_FLAG;
    if (zc != 0 && zc != prm_.ms_zeros_count) throw std::runtime_error("attached key image does not match the parameter set");
    if (bsk_factored_ && prm_.bsk_grouping < 2)
        throw std::runtime_error("attached key image does not match the parameter set");
    set_ms_params((u32)zc);
    build_ksk_limbs();
    HC(hipStreamSynchronize(stream_));
}

void Engine::ms_noise_reduce(u64* lwe_small, size_t count, int* chosen) {
    if (!count || !ms_.zcount) return;
    HC(hipSetDevice(device_));
    launch_ms_noise_reduce(lwe_small, (int)prm_.lwe_dim, ms_zeros_dev(), ms_, chosen, count, stream_);
    HC(hipGetLastError());
}

u64* Engine::small_scratch(size_t count) {
    if (count > small_cap_) {
        HC(hipSetDevice(device_));
        HC(hipStreamSynchronize(stream_));
        (void)hipFree(d_small_);
        d_small_ = nullptr;
        HC(hipMalloc(&d_small_, count * (prm_.lwe_dim + 1) * sizeof(u64)));
        small_cap_ = count;
    }
    return d_small_;
}

void Engine::load_cast_ksk(const u64* host_ksk) {
    HC(hipSetDevice(device_));
    if (!d_cast_ksk_) HC(hipMalloc(&d_cast_ksk_, (ksk_words() + prm_.lwe_dim + 1) * sizeof(u64)));
    HC(hipMemcpyAsync(d_cast_ksk_, host_ksk, ksk_words() * sizeof(u64), hipMemcpyHostToDevice, stream_));
    launch_ksk_colsum(d_cast_ksk_, N * (int)prm_.ks_level, (int)prm_.lwe_dim + 1, 1 << (prm_.ks_base_log - 1),
                      d_cast_ksk_ + ksk_words(), stream_);
    HC(hipGetLastError());
    HC(hipStreamSynchronize(stream_));
}

void Engine::cast_batch(const u64* lwe_in, const u64* identity_lut, u64* lwe_out, size_t count) {
    if (!count) return;
    if (!d_cast_ksk_) throw std::runtime_error("no cast key loaded");
    u64* small = small_scratch(count);
    keyswitch(lwe_in, small, count, BlockMap(), true);
    ms_noise_reduce(small, count);
    blind_rotate(small, identity_lut, nullptr, nullptr, lwe_out, count);
}

void Engine::keyswitch(const u64* lwe_in, u64* lwe_small_out, size_t count, const BlockMap& imap, bool with_cast_key,
                       const PbsScratch* ext) {
    const u64* key = with_cast_key ? d_cast_ksk_ : d_ksk_;
    if (!key) throw std::runtime_error("no key-switching key loaded");
    if (!count) return;
    HC(hipSetDevice(device_));
    if (imap.idx && !imap.ops) throw std::logic_error("block map without a period");
    // a launch takes a whole number of program instances, so that the block map shifts by pointer arithmetic
    auto whole_instances = [&](size_t chunk) {
        if (imap.idx) chunk = imap.ops >= chunk ? imap.ops : chunk / imap.ops * imap.ops;
        return chunk > count ? count : chunk;
    };
    if (!with_cast_key && ksm_min_ > 0 && d_ksk_limbs_ && count >= (size_t)ksm_min_) {
        // large batch: int8 matrix product on the matrix cores, in launches of about 4096 ciphertexts
        size_t chunk = whole_instances(4096);
        int8_t* dig = d_ksm_dig_;
        int* prod = d_ksm_prod_;
        if (ext) {
            // rows (multiples of the product tile) the caller's two regions hold
            const size_t rows = std::min(ext->ksm_dig_bytes / ksm_digit_bytes(1, N, (int)prm_.ks_level),
                                         ext->ksm_prod_bytes / ksm_product_bytes(1, (int)prm_.lwe_dim)) * ksm_rows_pad(1);
            if (chunk > rows) chunk = whole_instances(rows);
            if (!ext->ksm_dig || !ext->ksm_prod || !rows || ksm_rows_pad(chunk) > rows)
                throw std::invalid_argument("scratch too small for one key-switch launch");
            dig = ext->ksm_dig;
            prod = ext->ksm_prod;
        } else if (chunk > ksm_cap_) {
            HC(hipStreamSynchronize(stream_));
            (void)hipFree(d_ksm_dig_); (void)hipFree(d_ksm_prod_);
            d_ksm_dig_ = nullptr; d_ksm_prod_ = nullptr; ksm_cap_ = 0;
            HC(hipMalloc(&d_ksm_dig_, ksm_digit_bytes(chunk, N, (int)prm_.ks_level)));
            HC(hipMalloc(&d_ksm_prod_, ksm_product_bytes(chunk, (int)prm_.lwe_dim)));
            ksm_cap_ = chunk;
            dig = d_ksm_dig_;
            prod = d_ksm_prod_;
        }
        for (size_t c0 = 0; c0 < count; c0 += chunk) {
            const size_t cnt = count - c0 < chunk ? count - c0 : chunk;
            const u64* in = lwe_in + (imap.idx ? (c0 / imap.ops) * imap.inst_stride : c0) * (size_t)(N + 1);
            launch_keyswitch_mfma(d_ksk_limbs_, in, dig, prod, lwe_small_out + c0 * (prm_.lwe_dim + 1), cnt, N,
                                  (int)prm_.lwe_dim, (int)prm_.ks_base_log, (int)prm_.ks_level, imap, stream_);
            HC(hipGetLastError());
        }
        return;
    }
    // the grid's second dimension counts groups of sixteen ciphertexts and ends at 65535: longer
    // batches go in several launches
    size_t chunk = whole_instances(ks_valu_max_count());
    u32* digits = d_ksdig_;
    if (ext) {
        while (chunk > 1 && keyswitch_scratch_words(chunk, N) > ext->ks_dig_words) chunk = whole_instances(chunk / 2);
        if (!ext->ks_dig || keyswitch_scratch_words(chunk, N) > ext->ks_dig_words)
            throw std::invalid_argument("scratch too small for one key-switch launch");
        digits = ext->ks_dig;
    } else if (const size_t need = keyswitch_scratch_words(chunk, N); need > ksdig_cap_) {
        HC(hipStreamSynchronize(stream_));
        (void)hipFree(d_ksdig_);
        d_ksdig_ = nullptr;
        HC(hipMalloc(&d_ksdig_, need * sizeof(u32)));
        ksdig_cap_ = need;
        digits = d_ksdig_;
    }
    for (size_t c0 = 0; c0 < count; c0 += chunk) {
        const size_t cnt = count - c0 < chunk ? count - c0 : chunk;
        const u64* in = lwe_in + (imap.idx ? (c0 / imap.ops) * imap.inst_stride : c0) * (size_t)(N + 1);
        launch_keyswitch(key, key + ksk_words(), in, digits, lwe_small_out + c0 * (prm_.lwe_dim + 1), cnt, N,
                         (int)prm_.lwe_dim, (int)prm_.ks_base_log, (int)prm_.ks_level, imap, stream_);
        HC(hipGetLastError());
    }
}

void Engine::pbs_layer(u64* arena, size_t instances, size_t inst_stride_blocks, const u32* in_idx, const u32* out_idx,
                       const u32* lut_of_op, u32 ops, const u64* luts) {
    const size_t count = instances * ops;
    if (!count) return;
    u64* small = small_scratch(count);
    BlockMap im{in_idx, ops, inst_stride_blocks}, om{out_idx, ops, inst_stride_blocks};
    keyswitch(arena, small, count, im);
    ms_noise_reduce(small, count);
    blind_rotate(small, luts, lut_of_op, nullptr, arena, count, om);
}

void Engine::lin_layer(u64* arena, size_t instances, size_t inst_stride_blocks, const LinOp* ops, int nops, u64 delta) {
    if (!instances || !nops) return;
    HC(hipSetDevice(device_));
    launch_lin_batch(arena, ops, nops, instances, inst_stride_blocks, delta, stream_);
    HC(hipGetLastError());
}

// Ciphertexts per blind-rotation launch.  The workgroups of a launch walk the bootstrapping key in
// near lock step, which is what keeps the key row of the current step in L2; over a long launch they
// drift apart, the resident workgroups end up spread over the whole key and the key reads fall
// through to HBM (measured: 86.7 ciphertexts/ms in launches of 4096, 70.7 in one launch of 65536).
// Long batches are therefore cut into launches of about four rounds of resident workgroups, a
// whole number of program instances each (so that block maps and table indices shift by pointer
// arithmetic alone), picked to fill its last round as well as possible.
static size_t launch_chunk(size_t count, size_t ops, size_t slots) {
    const size_t target = 4 * slots;
    if (count <= target + target / 2) return count;
    if (ops == 0) return target;
    if (ops >= target) return ops;
    size_t best = 0;
    double best_fill = -1;
    for (size_t m = (target / 2 + ops - 1) / ops; m * ops <= 2 * target; ++m) {
        const size_t c = m * ops, rounds = (c + slots - 1) / slots;
        const double fill = (double)c / (double)(rounds * slots);
        if (fill > best_fill + 1e-9) { best_fill = fill; best = c; }
    }
    return best ? best : ops;
}

// Grouped key: which launches put two ciphertexts in a workgroup (returns the count above which they
// do, for this launch).  Measured per launch size (ms, one ciphertext per workgroup / two): 384: 6.6 /
// 6.4, 512: 7.0 / 6.6, 640: 9.9 / 11.4, 768: 10.2 / 11.4, 1024: 13.0 / 11.9, 2048: 24.5 / 22.2, 4096:
// 46.8 / 42.0 -- two per workgroup win except between two and three and a half ciphertexts per CU,
// where paired workgroups would double up on a quarter to a half of the CUs.
int Engine::pair_threshold(size_t count) const {
    if (pair_above_ != -2) return pair_above_;   // forced by LFHE_PAIR_ABOVE (-1: never)
    const size_t cu = (size_t)cu_count_;
    // three key bits per step (ms, one / two per workgroup): 512: 5.45 / 5.24, 640: 7.96 / 9.04, 768: 8.18 / 9.06,
    // 896: 10.51 / 9.33, 1024: 10.69 / 9.68 -- one per workgroup only between two and three and a quarter per CU
    if (prm_.bsk_grouping == 3) return (count <= 2 * cu || 4 * count > 13 * cu) ? 0 : -1;
    return (count <= 2 * cu || 2 * count > 7 * cu) ? 0 : -1;
}

void Engine::blind_rotate(const u64* lwe_small, const u64* luts, const u32* lut_index, u64* glwe_out, u64* lwe_out,
                          size_t count, const BlockMap& omap) {
    if (!d_bsk_) throw std::runtime_error("no bootstrapping key loaded");
    blind_rotate_with(d_bsk_, prm_.lwe_dim, lwe_small, luts, lut_index, glwe_out, lwe_out, count, omap);
}

void Engine::blind_rotate_with(const double* bsk, uint32_t n, const u64* lwe_small, const u64* luts, const u32* lut_index,
                               u64* glwe_out, u64* lwe_out, size_t count, const BlockMap& omap, int factored) {
    if (!count) return;
    const bool fact = factored < 0 ? bsk_factored_ : factored != 0;
    HC(hipSetDevice(device_));
    const size_t n1 = (size_t)n + 1;
    // ciphertexts in flight: four per CU, two with the one-bit-per-step key over 2^64
    const size_t slots = (size_t)(single_modulus() || prm_.bsk_grouping >= 2 ? 4 : 2) * (size_t)cu_count_;
    if (omap.idx && !omap.ops) throw std::logic_error("block map without a period");
    size_t chunk = br_chunk_ > 0 ? (omap.ops ? ((size_t)br_chunk_ + omap.ops - 1) / omap.ops * omap.ops : (size_t)br_chunk_)
                                 : launch_chunk(count, omap.ops, slots);
    if (g64_guard_.capacity && chunk > g64_guard_.capacity)   // one mark per ciphertext of a launch
        chunk = omap.ops ? std::max<size_t>(1, g64_guard_.capacity / omap.ops) * omap.ops : g64_guard_.capacity;
    for (size_t c0 = 0; c0 < count; c0 += chunk) {
        const size_t cnt = count - c0 < chunk ? count - c0 : chunk;
        // the launch sees ciphertext c0 as its ciphertext 0: shift every per-ciphertext pointer
        const u64* in = lwe_small + c0 * n1;
        u64* gout = glwe_out ? glwe_out + c0 * 2 * N : nullptr;
        u64* lout = lwe_out;
        const u32* li = lut_index;
        if (omap.idx) { if (lout) lout += (c0 / omap.ops) * omap.inst_stride * (size_t)(N + 1); }
        else {
            if (lout) lout += c0 * (size_t)(N + 1);
            if (li && !omap.ops) li += c0;   // with a period the launch starts on a multiple of it
        }
        if (single_modulus())
            launch_blind_rotate_q(T_, bsk, in, (int)n, luts, li, gout, lout, cnt, omap,
                                  1.0 / (double)qp_gadget(), wide_below_, lockstep_above_,
                                  prm_.bsk_grouping == 3 ? 4 : prm_.bsk_grouping == 2 && fact ? 3 : (int)prm_.bsk_grouping,
                                  pair_threshold(cnt),
                                  stream_);
        else
            launch_blind_rotate(T_, bsk, in, (int)n, luts, li, gout, lout, cnt, omap,
                                prm_.bsk_grouping == 3 ? 4 : prm_.bsk_grouping == 2 && fact ? 3 : (int)prm_.bsk_grouping,
                                pair_above_ != -2 ? pair_above_ : (cnt > 2 * (size_t)cu_count_ ? 0 : -1),
                                prm_.bsk_grouping >= 2 ? wide_below_ : 0, g64_guard_, stream_);
        HC(hipGetLastError());
    }
}

void Engine::pbs_batch(const u64* lwe_in, const u64* luts, const u32* lut_index, u64* lwe_out, size_t count) {
    if (!count) return;
    u64* small = small_scratch(count);
    keyswitch(lwe_in, small, count);
    ms_noise_reduce(small, count);
    blind_rotate(small, luts, lut_index, nullptr, lwe_out, count);
}

void Engine::pbs_batch_on(const PbsScratch& scratch, const u64* lwe_in, const u64* luts, const u32* lut_index,
                          u64* lwe_out, size_t count) {
    if (!count) return;
    if (!scratch.small || scratch.small_words < count * ((size_t)prm_.lwe_dim + 1))
        throw std::invalid_argument("scratch too small for the key-switched ciphertexts");
    keyswitch(lwe_in, scratch.small, count, BlockMap(), false, &scratch);
    ms_noise_reduce(scratch.small, count);
    blind_rotate(scratch.small, luts, lut_index, nullptr, lwe_out, count);
}

void Engine::external_product(const u32* ggsw_index, const u64* glwe, u64* out, size_t count) {
    if (!d_bsk_) throw std::runtime_error("no bootstrapping key loaded");
    if (!count) return;
    HC(hipSetDevice(device_));
    if (bsk_factored_) throw std::runtime_error("unsupported on a grouped key held in factored form: its rows are no longer GGSWs");
    if (single_modulus()) launch_external_product_q(T_, d_bsk_, ggsw_index, glwe, out, count, 1.0 / (double)qp_gadget(), stream_);
    else launch_external_product(T_, d_bsk_, ggsw_index, glwe, out, count, stream_);
    HC(hipGetLastError());
}

void Engine::ntt_forward(const u64* polys, double* out, size_t count) {
    if (!count) return;
    HC(hipSetDevice(device_));
    launch_ntt_forward(T_, polys, out, count, 0, stream_);
    HC(hipGetLastError());
}

void Engine::ntt_inverse(const double* in, u64* out, size_t count) {
    if (!count) return;
    HC(hipSetDevice(device_));
    launch_ntt_inverse(T_, in, out, count, stream_);
    HC(hipGetLastError());
}

}  // namespace lfhe
