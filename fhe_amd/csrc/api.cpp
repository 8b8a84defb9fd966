// C ABI (include/lfhe.h): context, keys, the bootstrap path and its stages.  The integer operators
// (lfhe_encrypt ... lfhe_evaluate_batch) live in radix.cpp.
#include <sys/random.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

#include "ctx.h"
#include "tables.h"

using namespace lfhe;

#define HC(x) hip_check((x), #x)

namespace {

// RAII device buffer on the context's device
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    void alloc(size_t count) {
        n = count;
        if (count) HC(hipMalloc(&p, count * sizeof(T)));
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    void upload(const T* h, size_t count, hipStream_t st) {
        HC(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, st));
    }
    void download(T* h, size_t count, hipStream_t st) const {
        HC(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, st));
    }
};

int host_threads() {
    unsigned hc = std::thread::hardware_concurrency();
    if (hc == 0) hc = 4;
    return (int)(hc > 16 ? 16 : hc);
}

// 32 bytes from the operating system's generator
void os_entropy(uint8_t out[32]) {
    size_t got = 0;
    while (got < 32) {
        const ssize_t r = getrandom(out + got, 32 - got, 0);
        if (r <= 0) throw std::runtime_error("getrandom failed: no entropy for keys and encryptions");
        got += (size_t)r;
    }
}

void require_server(lfhe_ctx* c) {
    if (!c->eng->has_keys()) throw std::logic_error("server keys not loaded: call lfhe_keygen or an import first");
}

}  // namespace

extern "C" {

const char* lfhe_version(void) { return "lfhe-mi355x 0.1 (gfx950)"; }

static void params_out(const EngineParams& d, lfhe_params* p) {
    std::memset(p, 0, sizeof *p);
    p->lwe_dim = d.lwe_dim; p->glwe_dim = d.glwe_dim; p->poly_size = d.poly_size;
    p->pbs_base_log = d.pbs_base_log; p->pbs_level = d.pbs_level;
    p->ks_base_log = d.ks_base_log; p->ks_level = d.ks_level;
    p->message_modulus = d.message_modulus; p->carry_modulus = d.carry_modulus;
    p->lwe_noise_log2 = d.lwe_noise_log2; p->glwe_noise_log2 = d.glwe_noise_log2;
    p->ms_zeros_count = d.ms_zeros_count; p->ms_bound_log2 = d.ms_bound_log2;
    p->ms_r_sigma_factor = d.ms_r_sigma_factor; p->ms_input_variance = d.ms_input_variance;
    p->bsk_modulus = d.bsk_modulus;
    p->bsk_grouping = d.bsk_grouping;
}

void lfhe_default_params(lfhe_params* p) {
    if (!p) return;
    params_out(EngineParams(), p);
}

int32_t lfhe_ctx_create_custom(lfhe_ctx** out, const lfhe_params* p, int32_t device, const uint8_t seed[32]) {
    if (!out || !p) return LFHE_ERR_ARG;
    *out = nullptr;
    lfhe_ctx* c = new (std::nothrow) lfhe_ctx();
    if (!c) return LFHE_ERR_DEVICE;
    c->prm.lwe_dim = p->lwe_dim; c->prm.glwe_dim = p->glwe_dim; c->prm.poly_size = p->poly_size;
    c->prm.pbs_base_log = p->pbs_base_log; c->prm.pbs_level = p->pbs_level;
    c->prm.ks_base_log = p->ks_base_log; c->prm.ks_level = p->ks_level;
    c->prm.message_modulus = p->message_modulus; c->prm.carry_modulus = p->carry_modulus;
    c->prm.lwe_noise_log2 = p->lwe_noise_log2; c->prm.glwe_noise_log2 = p->glwe_noise_log2;
    c->prm.ms_zeros_count = p->ms_zeros_count; c->prm.ms_bound_log2 = p->ms_bound_log2;
    c->prm.ms_r_sigma_factor = p->ms_r_sigma_factor; c->prm.ms_input_variance = p->ms_input_variance;
    c->prm.bsk_modulus = p->bsk_modulus;
    c->prm.bsk_grouping = p->bsk_grouping ? p->bsk_grouping : 1;
    int32_t rc = LFHE_OK;
    try {
        // Keys come from `seed` (null: from the operating system).  Encryption randomness never
        // does: a restarted server with the same key seed must not hand out the same masks and
        // noise again, so its generator starts from fresh entropy unless lfhe_encryption_reseed
        // asks for a reproducible stream.
        if (seed) std::memcpy(c->seed, seed, 32);
        else os_entropy(c->seed);
        uint8_t fresh[32];
        os_entropy(fresh);
        c->enc_rng.init(fresh, 16);
        if (p->message_modulus != 4 || p->carry_modulus != 4)
            throw std::invalid_argument("unsupported parameters: message_modulus and carry_modulus must be 4");
        c->eng.reset(new Engine(device, c->prm));
    } catch (const std::invalid_argument& e) {
        c->err = e.what(); rc = LFHE_ERR_UNSUPPORTED;
    } catch (const std::exception& e) {
        c->err = e.what();
        rc = std::strstr(e.what(), "unsupported") ? LFHE_ERR_UNSUPPORTED : LFHE_ERR_DEVICE;
    }
    if (rc != LFHE_OK) {
        std::fprintf(stderr, "lfhe_ctx_create: %s\n", c->err.c_str());
        delete c;
        return rc;
    }
    *out = c;
    return LFHE_OK;
}

int32_t lfhe_ctx_create(lfhe_ctx** out, uint32_t params_id, int32_t device, const uint8_t seed[32]) {
    if (params_id != LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 &&
        params_id != LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK &&
        params_id != LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK_GROUPED &&
        params_id != LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_GROUPED &&
        params_id != LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK_GROUPED3 &&
        params_id != LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_GROUPED3)
        return LFHE_ERR_UNSUPPORTED;
    lfhe_params p;
    lfhe_default_params(&p);
    if (params_id == LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK ||
        params_id == LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK_GROUPED ||
        params_id == LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK_GROUPED3)
        p.bsk_modulus = LFHE_BSK_MODULUS_PRIME;
    if (params_id == LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK_GROUPED3 ||
        params_id == LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_GROUPED3)
        p.bsk_grouping = 3;
    if (params_id == LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK_GROUPED ||
        params_id == LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_GROUPED)
        p.bsk_grouping = 2;
    return lfhe_ctx_create_custom(out, &p, device, seed);
}

void lfhe_ctx_destroy(lfhe_ctx* ctx) { delete ctx; }

int32_t lfhe_encryption_reseed(lfhe_ctx* ctx, const uint8_t seed[32]) {
    return guarded(ctx, [&]() -> int32_t {
        uint8_t fresh[32];
        if (!seed) os_entropy(fresh);
        ctx->enc_rng.init(seed ? seed : fresh, 16);
        return LFHE_OK;
    });
}

uint64_t lfhe_arena_chunk(uint64_t budget_bytes, uint64_t instance_bytes) {
    // the scatter / gather launches index instances with the grid's second dimension (65535 at most)
    const uint64_t grid_limit = 65535;
    uint64_t cap = budget_bytes / (instance_bytes ? instance_bytes : 1);
    if (cap > grid_limit) cap = grid_limit;
    return cap ? cap : 1;
}

const char* lfhe_last_error(const lfhe_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int32_t lfhe_get_params(const lfhe_ctx* ctx, lfhe_params* p) {
    if (!ctx || !p) return LFHE_ERR_ARG;
    params_out(ctx->prm, p);
    return LFHE_OK;
}

int32_t lfhe_health(lfhe_ctx* ctx, char* out, uint64_t* out_len) {
    if (!ctx || !out_len) return LFHE_ERR_ARG;
    char buf[256];
    int n = std::snprintf(buf, sizeof buf,
                          "{\"status\":\"ok\",\"threshold\":false,\"parties\":1,\"backend\":\"hip-gfx950\","
                          "\"device\":%d,\"keys\":%s,\"pbs\":%llu}",
                          ctx->eng->device(), ctx->eng->has_keys() ? "true" : "false",
                          (unsigned long long)ctx->pbs_count);
    const uint64_t need = (uint64_t)n + 1;
    if (!out) { *out_len = need; return LFHE_OK; }
    if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
    std::memcpy(out, buf, need);
    *out_len = need;
    return LFHE_OK;
}

// ---- keys --------------------------------------------------------------------------------------
static void upload_server_keys(lfhe_ctx* c) {
    c->have_server = false;
    c->eng->load_bsk(c->bsk.data());
    c->eng->load_ksk(c->ksk.data(), c->ms_zeros.empty() ? nullptr : c->ms_zeros.data());
    c->have_server = true;
}

int32_t lfhe_keygen(lfhe_ctx* ctx) {
    return guarded(ctx, [&]() -> int32_t {
        const EngineParams& p = ctx->prm;
        ctx->lwe_sk.assign(p.lwe_dim, 0);
        ctx->glwe_sk.assign(N, 0);
        ChaCha r1(ctx->seed, 1), r2(ctx->seed, 2);
        gen_binary_key(r1, ctx->lwe_sk.data(), p.lwe_dim);
        gen_binary_key(r2, ctx->glwe_sk.data(), N);
        ctx->have_client = true;
        ctx->bsk.resize(ctx->eng->bsk_raw_words());
        ctx->ksk.resize(ctx->eng->ksk_words());
        if (p.bsk_modulus) gen_bsk_qp(p, ctx->lwe_sk.data(), ctx->glwe_sk.data(), ctx->seed, ctx->bsk.data(), host_threads());
        else gen_bsk(p, ctx->lwe_sk.data(), ctx->glwe_sk.data(), ctx->seed, ctx->bsk.data(), host_threads());
        gen_ksk(p, ctx->glwe_sk.data(), ctx->lwe_sk.data(), ctx->seed, ctx->ksk.data(), host_threads());
        ctx->ms_zeros.resize(ctx->eng->ms_zeros_words());
        gen_ms_zeros(p, ctx->lwe_sk.data(), ctx->seed, ctx->ms_zeros.data(), host_threads());
        upload_server_keys(ctx);
        return LFHE_OK;
    });
}

int32_t lfhe_keygen_server(lfhe_ctx* ctx) {
    return guarded(ctx, [&]() -> int32_t {
        if (!ctx->have_client) throw std::logic_error("no client keys: import them first");
        const EngineParams& p = ctx->prm;
        ctx->bsk.resize(ctx->eng->bsk_raw_words());
        ctx->ksk.resize(ctx->eng->ksk_words());
        if (p.bsk_modulus) gen_bsk_qp(p, ctx->lwe_sk.data(), ctx->glwe_sk.data(), ctx->seed, ctx->bsk.data(), host_threads());
        else gen_bsk(p, ctx->lwe_sk.data(), ctx->glwe_sk.data(), ctx->seed, ctx->bsk.data(), host_threads());
        gen_ksk(p, ctx->glwe_sk.data(), ctx->lwe_sk.data(), ctx->seed, ctx->ksk.data(), host_threads());
        ctx->ms_zeros.resize(ctx->eng->ms_zeros_words());
        gen_ms_zeros(p, ctx->lwe_sk.data(), ctx->seed, ctx->ms_zeros.data(), host_threads());
        upload_server_keys(ctx);
        return LFHE_OK;
    });
}

int32_t lfhe_server_keys_import_raw(lfhe_ctx* ctx, const uint64_t* bsk, uint64_t bsk_words, const uint64_t* ksk,
                                    uint64_t ksk_words, const uint64_t* ms_zeros, uint64_t ms_words) {
    return guarded(ctx, [&]() -> int32_t {
        if (!bsk || !ksk) throw std::invalid_argument("null key pointer");
        if (bsk_words != ctx->eng->bsk_raw_words() || ksk_words != ctx->eng->ksk_words() ||
            (ms_zeros && ms_words && ms_words != ctx->eng->ms_zeros_words()))
            throw std::invalid_argument("server key size does not match the parameter set");
        if (ctx->prm.bsk_modulus)
            for (uint64_t i = 0; i < bsk_words; ++i)
                if (bsk[i] >= ctx->prm.bsk_modulus) throw std::invalid_argument("bootstrapping key word not reduced");
        ctx->bsk.assign(bsk, bsk + bsk_words);
        ctx->ksk.assign(ksk, ksk + ksk_words);
        if (ms_zeros && ms_words) ctx->ms_zeros.assign(ms_zeros, ms_zeros + ms_words);
        else ctx->ms_zeros.clear();
        upload_server_keys(ctx);
        return LFHE_OK;
    });
}

int32_t lfhe_client_keys_import_raw(lfhe_ctx* ctx, const uint64_t* lwe_sk, const uint64_t* glwe_sk) {
    return guarded(ctx, [&]() -> int32_t {
        if (!lwe_sk || !glwe_sk) throw std::invalid_argument("null key pointer");
        for (uint32_t i = 0; i < ctx->prm.lwe_dim; ++i)
            if (lwe_sk[i] > 1) throw std::invalid_argument("secret key words must be 0 or 1");
        for (int i = 0; i < N; ++i)
            if (glwe_sk[i] > 1) throw std::invalid_argument("secret key words must be 0 or 1");
        ctx->lwe_sk.assign(lwe_sk, lwe_sk + ctx->prm.lwe_dim);
        ctx->glwe_sk.assign(glwe_sk, glwe_sk + N);
        ctx->have_client = true;
        return LFHE_OK;
    });
}

namespace {
constexpr uint32_t KEY_MAGIC = 0x4b46484cu;  // "LHFK"
struct KeyHeader {
    uint32_t magic, which;
    lfhe_params prm;
    uint64_t words_a, words_b, words_c;
};
}  // namespace

int32_t lfhe_keys_export(lfhe_ctx* ctx, uint32_t which, uint8_t* out, uint64_t* out_len) {
    return guarded(ctx, [&]() -> int32_t {
        if (!out_len) throw std::invalid_argument("null out_len");
        static const std::vector<uint64_t> none;
        const std::vector<uint64_t>*a, *b, *c = &none;
        if (which == LFHE_KEY_CLIENT) {
            if (!ctx->have_client) throw std::logic_error("no client keys");
            a = &ctx->lwe_sk; b = &ctx->glwe_sk;
        } else if (which == LFHE_KEY_SERVER) {
            if (!ctx->have_server) throw std::logic_error("no server keys");
            a = &ctx->bsk; b = &ctx->ksk; c = &ctx->ms_zeros;
        } else {
            throw std::invalid_argument("unknown key kind");
        }
        const uint64_t need = sizeof(KeyHeader) + (a->size() + b->size() + c->size()) * 8;
        if (!out) { *out_len = need; return LFHE_OK; }
        if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
        KeyHeader h{};
        h.magic = KEY_MAGIC; h.which = which;
        lfhe_get_params(ctx, &h.prm);
        h.words_a = a->size(); h.words_b = b->size(); h.words_c = c->size();
        std::memcpy(out, &h, sizeof h);
        std::memcpy(out + sizeof h, a->data(), a->size() * 8);
        std::memcpy(out + sizeof h + a->size() * 8, b->data(), b->size() * 8);
        if (!c->empty()) std::memcpy(out + sizeof h + (a->size() + b->size()) * 8, c->data(), c->size() * 8);
        *out_len = need;
        return LFHE_OK;
    });
}

int32_t lfhe_keys_import(lfhe_ctx* ctx, uint32_t which, const uint8_t* in, uint64_t len) {
    if (!ctx) return LFHE_ERR_ARG;
    if (!in || len < sizeof(KeyHeader)) { ctx->err = "key blob too short"; return LFHE_ERR_FORMAT; }
    KeyHeader h;
    std::memcpy(&h, in, sizeof h);
    lfhe_params mine;
    lfhe_get_params(ctx, &mine);
    // The word counts are caller data: each is compared with what this context's parameter set
    // prescribes for `which` before anything is summed, multiplied or allocated.
    uint64_t want_a, want_b, want_c;
    if (which == LFHE_KEY_CLIENT) {
        want_a = ctx->prm.lwe_dim; want_b = (uint64_t)N; want_c = 0;
    } else if (which == LFHE_KEY_SERVER) {
        want_a = ctx->eng->bsk_raw_words(); want_b = ctx->eng->ksk_words(); want_c = ctx->eng->ms_zeros_words();
    } else {
        ctx->err = "unknown key kind";
        return LFHE_ERR_ARG;
    }
    const bool counts_ok = h.words_a == want_a && h.words_b == want_b && (h.words_c == want_c || h.words_c == 0);
    if (h.magic != KEY_MAGIC || h.which != which || std::memcmp(&h.prm, &mine, sizeof mine) != 0 || !counts_ok ||
        (len - sizeof h) % 8 != 0 || (len - sizeof h) / 8 != h.words_a + h.words_b + h.words_c) {
        ctx->err = "key blob does not match this context";
        return LFHE_ERR_FORMAT;
    }
    return guarded(ctx, [&]() -> int32_t {
        std::vector<uint64_t> va(h.words_a), vb(h.words_b), vc(h.words_c);
        std::memcpy(va.data(), in + sizeof h, h.words_a * 8);
        std::memcpy(vb.data(), in + sizeof h + h.words_a * 8, h.words_b * 8);
        if (h.words_c) std::memcpy(vc.data(), in + sizeof h + (h.words_a + h.words_b) * 8, h.words_c * 8);
        if (which == LFHE_KEY_CLIENT) return lfhe_client_keys_import_raw(ctx, va.data(), vb.data());
        return lfhe_server_keys_import_raw(ctx, va.data(), va.size(), vb.data(), vb.size(),
                                           vc.empty() ? nullptr : vc.data(), vc.size());
    });
}

int32_t lfhe_device_keys_size(const lfhe_ctx* ctx, uint64_t* bsk_bytes, uint64_t* ksk_bytes) {
    if (!ctx || !bsk_bytes || !ksk_bytes) return LFHE_ERR_ARG;
    *bsk_bytes = ctx->eng->bsk_dev_bytes();
    *ksk_bytes = ctx->eng->ksk_dev_bytes();
    return LFHE_OK;
}

int32_t lfhe_device_keys_ptr(lfhe_ctx* ctx, void** bsk_dev, void** ksk_dev) {
    return guarded(ctx, [&]() -> int32_t {
        if (!bsk_dev || !ksk_dev) throw std::invalid_argument("null pointer");
        require_server(ctx);
        *bsk_dev = ctx->eng->bsk_dev();
        *ksk_dev = ctx->eng->ksk_dev();
        return LFHE_OK;
    });
}

int32_t lfhe_device_keys_attach(lfhe_ctx* ctx, void* bsk_dev, void* ksk_dev) {
    return guarded(ctx, [&]() -> int32_t {
        if (!bsk_dev || !ksk_dev) throw std::invalid_argument("null pointer");
        ctx->eng->attach_device_keys(static_cast<double*>(bsk_dev), static_cast<u64*>(ksk_dev));
        return LFHE_OK;
    });
}

// ---- the bootstrap path ------------------------------------------------------------------------
int32_t lfhe_make_lut(lfhe_ctx* ctx, const uint64_t* table, uint64_t* lut_body) {
    if (!ctx || !table || !lut_body) return LFHE_ERR_ARG;
    make_lut_body(ctx->space(), ctx->delta(), table, lut_body);
    return LFHE_OK;
}

int32_t lfhe_pbs_batch_device(lfhe_ctx* ctx, uint64_t count, const void* d_lwe_in, const void* d_lut_bodies,
                              const void* d_lut_index, void* d_lwe_out) {
    return guarded(ctx, [&]() -> int32_t {
        if (count && (!d_lwe_in || !d_lut_bodies || !d_lwe_out)) throw std::invalid_argument("null device pointer");
        require_server(ctx);
        ctx->eng->pbs_batch(static_cast<const u64*>(d_lwe_in), static_cast<const u64*>(d_lut_bodies),
                            static_cast<const u32*>(d_lut_index), static_cast<u64*>(d_lwe_out), count);
        ctx->pbs_count += count;
        return LFHE_OK;
    });
}

int32_t lfhe_sync(lfhe_ctx* ctx) {
    return guarded(ctx, [&]() -> int32_t { ctx->eng->sync(); return LFHE_OK; });
}

int32_t lfhe_pbs_batch_device_timed(lfhe_ctx* ctx, uint64_t count, const void* d_lwe_in, const void* d_lut_bodies,
                                    const void* d_lut_index, void* d_lwe_out, uint32_t iters, double* ks_ms,
                                    double* br_ms) {
    return guarded(ctx, [&]() -> int32_t {
        if (!count || !iters || !d_lwe_in || !d_lut_bodies || !d_lwe_out || !ks_ms || !br_ms)
            throw std::invalid_argument("bad argument");
        require_server(ctx);
        Engine& E = *ctx->eng;
        HC(hipSetDevice(E.device()));
        u64* small = E.small_scratch(count);
        hipEvent_t e0, e1, e2;
        HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1)); HC(hipEventCreate(&e2));
        double ks = 0, br = 0;
        for (uint32_t it = 0; it < iters; ++it) {
            HC(hipEventRecord(e0, E.stream()));
            E.keyswitch(static_cast<const u64*>(d_lwe_in), small, count);
            E.ms_noise_reduce(small, count);
            HC(hipEventRecord(e1, E.stream()));
            E.blind_rotate(small, static_cast<const u64*>(d_lut_bodies), static_cast<const u32*>(d_lut_index),
                           nullptr, static_cast<u64*>(d_lwe_out), count);
            HC(hipEventRecord(e2, E.stream()));
            HC(hipEventSynchronize(e2));
            float a = 0, b = 0;
            HC(hipEventElapsedTime(&a, e0, e1));
            HC(hipEventElapsedTime(&b, e1, e2));
            ks += a; br += b;
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
        *ks_ms = ks / iters;
        *br_ms = br / iters;
        ctx->pbs_count += count * iters;
        return LFHE_OK;
    });
}

int32_t lfhe_pbs_batch(lfhe_ctx* ctx, uint64_t count, const uint64_t* lwe_in, const uint64_t* lut_bodies,
                       uint32_t nluts, const uint32_t* lut_index, uint64_t* lwe_out) {
    return guarded(ctx, [&]() -> int32_t {
        if (!count) return LFHE_OK;
        if (!lwe_in || !lut_bodies || !lwe_out || !nluts) throw std::invalid_argument("null pointer");
        if (lut_index)
            for (uint64_t i = 0; i < count; ++i)
                if (lut_index[i] >= nluts) throw std::invalid_argument("lut_index out of range");
        require_server(ctx);
        Engine& E = *ctx->eng;
        HC(hipSetDevice(E.device()));
        const size_t w = (size_t)N + 1;
        DevBuf<u64> din(count * w), dout(count * w), dl((size_t)nluts * N);
        DevBuf<u32> di(lut_index ? count : 0);
        din.upload(lwe_in, count * w, E.stream());
        dl.upload(lut_bodies, (size_t)nluts * N, E.stream());
        if (lut_index) di.upload(lut_index, count, E.stream());
        E.pbs_batch(din.p, dl.p, di.p, dout.p, count);
        dout.download(lwe_out, count * w, E.stream());
        E.sync();
        ctx->pbs_count += count;
        return LFHE_OK;
    });
}

// ---- stages ------------------------------------------------------------------------------------
int32_t lfhe_ntt_info(const lfhe_ctx* ctx, uint64_t primes[2], uint64_t psi[2]) {
    if (!ctx || !primes || !psi) return LFHE_ERR_ARG;
    HostTables H;
    build_host_tables(H);
    for (int r = 0; r < 2; ++r) { primes[r] = H.p[r]; psi[r] = H.psi[r]; }
    return LFHE_OK;
}

int32_t lfhe_ntt_forward(lfhe_ctx* ctx, uint64_t count, const uint64_t* polys, uint64_t* residues) {
    return guarded(ctx, [&]() -> int32_t {
        if (!count) return LFHE_OK;
        if (!polys || !residues) throw std::invalid_argument("null pointer");
        Engine& E = *ctx->eng;
        HC(hipSetDevice(E.device()));
        DevBuf<u64> din(count * N);
        DevBuf<double> dout(count * 2 * N);
        din.upload(polys, count * N, E.stream());
        E.ntt_forward(din.p, dout.p, count);
        std::vector<double> h(count * 2 * N);
        dout.download(h.data(), h.size(), E.stream());
        E.sync();
        const double pr[2] = {P0, P1};
        for (uint64_t c = 0; c < count; ++c)
            for (int r = 0; r < 2; ++r)
                for (int e = 0; e < N; ++e) {
                    double v = h[(c * 2 + r) * N + e];
                    if (v < 0) v += pr[r];
                    residues[(c * 2 + r) * N + h_bitrev((u32)e, LOGN)] = (uint64_t)v;
                }
        return LFHE_OK;
    });
}

int32_t lfhe_ntt_inverse(lfhe_ctx* ctx, uint64_t count, const uint64_t* residues, uint64_t* polys) {
    return guarded(ctx, [&]() -> int32_t {
        if (!count) return LFHE_OK;
        if (!residues || !polys) throw std::invalid_argument("null pointer");
        Engine& E = *ctx->eng;
        HC(hipSetDevice(E.device()));
        const u64 pr[2] = {P0_U, P1_U};
        std::vector<double> h(count * 2 * N);
        for (uint64_t c = 0; c < count; ++c)
            for (int r = 0; r < 2; ++r)
                for (int e = 0; e < N; ++e) {
                    const u64 v = residues[(c * 2 + r) * N + h_bitrev((u32)e, LOGN)];
                    if (v >= pr[r]) throw std::invalid_argument("residue not reduced");
                    h[(c * 2 + r) * N + e] = h_centred(v, pr[r]);
                }
        DevBuf<double> din(h.size());
        DevBuf<u64> dout(count * N);
        din.upload(h.data(), h.size(), E.stream());
        E.ntt_inverse(din.p, dout.p, count);
        dout.download(polys, count * N, E.stream());
        E.sync();
        return LFHE_OK;
    });
}

int32_t lfhe_external_product(lfhe_ctx* ctx, uint64_t count, const uint32_t* ggsw_index, const uint64_t* glwe,
                              uint64_t* out) {
    return guarded(ctx, [&]() -> int32_t {
        if (!count) return LFHE_OK;
        if (!ggsw_index || !glwe || !out) throw std::invalid_argument("null pointer");
        for (uint64_t i = 0; i < count; ++i)
            if (ggsw_index[i] >= ctx->eng->bsk_ggsw_count()) throw std::invalid_argument("ggsw_index out of range");
        if (ctx->prm.bsk_modulus)
            for (uint64_t i = 0; i < count * 2 * N; ++i)
                if (glwe[i] >= ctx->prm.bsk_modulus) throw std::invalid_argument("residue not reduced");
        require_server(ctx);
        Engine& E = *ctx->eng;
        HC(hipSetDevice(E.device()));
        DevBuf<u64> din(count * 2 * N), dout(count * 2 * N);
        DevBuf<u32> di(count);
        din.upload(glwe, count * 2 * N, E.stream());
        di.upload(ggsw_index, count, E.stream());
        E.external_product(di.p, din.p, dout.p, count);
        dout.download(out, count * 2 * N, E.stream());
        E.sync();
        return LFHE_OK;
    });
}

int32_t lfhe_keyswitch(lfhe_ctx* ctx, uint64_t count, const uint64_t* lwe_in, uint64_t* lwe_small_out) {
    return guarded(ctx, [&]() -> int32_t {
        if (!count) return LFHE_OK;
        if (!lwe_in || !lwe_small_out) throw std::invalid_argument("null pointer");
        require_server(ctx);
        Engine& E = *ctx->eng;
        HC(hipSetDevice(E.device()));
        const size_t w = (size_t)N + 1, ws = (size_t)ctx->prm.lwe_dim + 1;
        DevBuf<u64> din(count * w), dout(count * ws);
        din.upload(lwe_in, count * w, E.stream());
        E.keyswitch(din.p, dout.p, count);
        dout.download(lwe_small_out, count * ws, E.stream());
        E.sync();
        return LFHE_OK;
    });
}

int32_t lfhe_ms_noise_reduce(lfhe_ctx* ctx, uint64_t count, uint64_t* lwe_small, int32_t* chosen) {
    return guarded(ctx, [&]() -> int32_t {
        if (!count) return LFHE_OK;
        if (!lwe_small) throw std::invalid_argument("null pointer");
        require_server(ctx);
        Engine& E = *ctx->eng;
        if (!E.has_ms_zeros()) throw std::logic_error("the key set holds no encryptions of zero");
        HC(hipSetDevice(E.device()));
        const size_t ws = (size_t)ctx->prm.lwe_dim + 1;
        DevBuf<u64> d(count * ws);
        DevBuf<int> dc(count);
        d.upload(lwe_small, count * ws, E.stream());
        E.ms_noise_reduce(d.p, count, dc.p);
        d.download(lwe_small, count * ws, E.stream());
        std::vector<int> hc(count);
        dc.download(hc.data(), count, E.stream());
        E.sync();
        if (chosen) for (uint64_t i = 0; i < count; ++i) chosen[i] = hc[i];
        return LFHE_OK;
    });
}

int32_t lfhe_blind_rotate(lfhe_ctx* ctx, uint64_t count, const uint64_t* lwe_small, const uint64_t* lut_bodies,
                          uint32_t nluts, const uint32_t* lut_index, uint64_t* glwe_out) {
    return guarded(ctx, [&]() -> int32_t {
        if (!count) return LFHE_OK;
        if (!lwe_small || !lut_bodies || !glwe_out || !nluts) throw std::invalid_argument("null pointer");
        if (lut_index)
            for (uint64_t i = 0; i < count; ++i)
                if (lut_index[i] >= nluts) throw std::invalid_argument("lut_index out of range");
        require_server(ctx);
        Engine& E = *ctx->eng;
        HC(hipSetDevice(E.device()));
        const size_t ws = (size_t)ctx->prm.lwe_dim + 1;
        DevBuf<u64> din(count * ws), dout(count * 2 * N), dl((size_t)nluts * N);
        DevBuf<u32> di(lut_index ? count : 0);
        din.upload(lwe_small, count * ws, E.stream());
        dl.upload(lut_bodies, (size_t)nluts * N, E.stream());
        if (lut_index) di.upload(lut_index, count, E.stream());
        E.blind_rotate(din.p, dl.p, di.p, dout.p, nullptr, count);
        dout.download(glwe_out, count * 2 * N, E.stream());
        E.sync();
        return LFHE_OK;
    });
}

}  // extern "C"
