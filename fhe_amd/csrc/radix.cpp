// Integer operators on radix ciphertexts (euint8 ... euint256, ebool): serialisation, encryption,
// decryption and the operator scheduler that turns each operation over a batch into layers of
// independent programmable bootstraps.
#include <cstring>
#include <stdexcept>

#include "ctx.h"

using namespace lfhe;

struct lfhe_radix {
    int placeholder = 0;
};

lfhe_ctx::lfhe_ctx() = default;
lfhe_ctx::~lfhe_ctx() = default;

namespace {

constexpr uint32_t CT_MAGIC = 0x5443464cu;  // "LFCT"
struct CtHeader {
    uint32_t magic;
    uint16_t version;
    uint16_t bit_width;
    uint32_t nblocks;
    uint8_t noise_level;
    uint8_t degree;
    uint16_t reserved;
};
static_assert(sizeof(CtHeader) == 16, "header layout");

bool width_ok(uint32_t w) { return w == 1 || w == 8 || w == 16 || w == 32 || w == 64 || w == 128 || w == 160 || w == 256; }
uint32_t blocks_of(uint32_t w) { return w == 1 ? 1 : w / 2; }

uint64_t ct_bytes(uint32_t w) { return sizeof(CtHeader) + (uint64_t)blocks_of(w) * (N + 1) * 8; }

uint32_t block_value(const uint8_t v[32], uint32_t i) {  // i-th base-4 digit
    const uint32_t bit = 2 * i;
    return (v[bit >> 3] >> (bit & 7)) & 3u;
}

}  // namespace

extern "C" {

int32_t lfhe_ciphertext_size(const lfhe_ctx* ctx, uint32_t bit_width, uint64_t* out_len) {
    if (!ctx || !out_len || !width_ok(bit_width)) return LFHE_ERR_ARG;
    *out_len = ct_bytes(bit_width);
    return LFHE_OK;
}

static int32_t encrypt_impl(lfhe_ctx* ctx, const uint8_t value_le[32], uint32_t bit_width, uint8_t* out,
                            uint64_t* out_len, bool trivial) {
    return guarded(ctx, [&]() -> int32_t {
        if (!value_le || !out_len) throw std::invalid_argument("null pointer");
        if (!width_ok(bit_width)) throw std::invalid_argument("unsupported bit width");
        const uint64_t need = ct_bytes(bit_width);
        if (!out) { *out_len = need; return LFHE_OK; }
        if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
        if (!trivial && !ctx->have_client) throw std::logic_error("no client key: call lfhe_keygen or import one");
        const uint32_t nb = blocks_of(bit_width);
        CtHeader h{CT_MAGIC, 1, (uint16_t)bit_width, nb, 1, 3, 0};
        if (bit_width == 1) h.degree = 1;
        std::memcpy(out, &h, sizeof h);
        std::vector<uint64_t> blk(N + 1);
        for (uint32_t i = 0; i < nb; ++i) {
            uint64_t m = bit_width == 1 ? (value_le[0] & 1u) : block_value(value_le, i);
            if (trivial) {
                std::fill(blk.begin(), blk.end(), 0);
                blk[N] = m * ctx->delta();
            } else {
                lwe_encrypt(ctx->glwe_sk.data(), N, m * ctx->delta(), ctx->prm.glwe_noise_log2, ctx->enc_rng, blk.data());
            }
            std::memcpy(out + sizeof h + (size_t)i * (N + 1) * 8, blk.data(), (N + 1) * 8);
        }
        *out_len = need;
        return LFHE_OK;
    });
}

int32_t lfhe_encrypt(lfhe_ctx* ctx, const uint8_t value_le[32], uint32_t bit_width, uint8_t* out, uint64_t* out_len) {
    return encrypt_impl(ctx, value_le, bit_width, out, out_len, false);
}

int32_t lfhe_trivial_encrypt(lfhe_ctx* ctx, const uint8_t value_le[32], uint32_t bit_width, uint8_t* out,
                             uint64_t* out_len) {
    return encrypt_impl(ctx, value_le, bit_width, out, out_len, true);
}

int32_t lfhe_decrypt(lfhe_ctx* ctx, const uint8_t* ct, uint64_t ct_len, uint8_t value_le[32], uint32_t* bit_width) {
    if (!ctx) return LFHE_ERR_ARG;
    if (!ct || !value_le) { ctx->err = "null pointer"; return LFHE_ERR_ARG; }
    if (!ctx->have_client) { ctx->err = "no client key"; return LFHE_ERR_STATE; }
    CtHeader h;
    if (ct_len < sizeof h) { ctx->err = "ciphertext too short"; return LFHE_ERR_FORMAT; }
    std::memcpy(&h, ct, sizeof h);
    if (h.magic != CT_MAGIC || h.version != 1 || !width_ok(h.bit_width) || h.nblocks != blocks_of(h.bit_width) ||
        ct_len != ct_bytes(h.bit_width)) {
        ctx->err = "malformed ciphertext";
        return LFHE_ERR_FORMAT;
    }
    std::memset(value_le, 0, 32);
    std::vector<uint64_t> blk(N + 1);
    // recombine base 4 with carries: blocks may hold values up to 15
    uint64_t carry = 0;
    const uint32_t bits = h.bit_width;
    for (uint32_t i = 0; i < h.nblocks; ++i) {
        std::memcpy(blk.data(), ct + sizeof h + (size_t)i * (N + 1) * 8, (N + 1) * 8);
        const uint64_t phase = lwe_phase(ctx->glwe_sk.data(), N, blk.data());
        const uint64_t delta = ctx->delta();
        uint64_t m = ((phase + delta / 2) / delta) % ctx->space();
        m += carry;
        const uint32_t digit = (uint32_t)(m & 3);
        carry = m >> 2;
        const uint32_t bit = 2 * i;
        if (bits == 1) value_le[0] = (uint8_t)(m & 1);
        else value_le[bit >> 3] |= (uint8_t)(digit << (bit & 7));
    }
    if (bit_width) *bit_width = h.bit_width;
    return LFHE_OK;
}

int32_t lfhe_public_key(lfhe_ctx* ctx, uint8_t*, uint64_t*) {
    if (!ctx) return LFHE_ERR_ARG;
    ctx->err = "public-key encryption is not built yet";
    return LFHE_ERR_UNSUPPORTED;
}

int32_t lfhe_op_from_name(const char* name) {
    if (!name) return -1;
    static const struct { const char* n; int op; } tab[] = {
        {"add", LFHE_OP_ADD}, {"sub", LFHE_OP_SUB}, {"mul", LFHE_OP_MUL}, {"eq", LFHE_OP_EQ}, {"ne", LFHE_OP_NE},
        {"ge", LFHE_OP_GE}, {"gt", LFHE_OP_GT}, {"le", LFHE_OP_LE}, {"lt", LFHE_OP_LT}, {"min", LFHE_OP_MIN},
        {"max", LFHE_OP_MAX}, {"and", LFHE_OP_AND}, {"or", LFHE_OP_OR}, {"xor", LFHE_OP_XOR}, {"not", LFHE_OP_NOT},
        {"neg", LFHE_OP_NEG}, {"select", LFHE_OP_SELECT}, {"shl", LFHE_OP_SHL}, {"shr", LFHE_OP_SHR},
        {"rotl", LFHE_OP_ROTL}, {"rotr", LFHE_OP_ROTR}, {"div", LFHE_OP_DIV}, {"rem", LFHE_OP_REM},
        {"cast", LFHE_OP_CAST}};
    for (const auto& e : tab)
        if (!std::strcmp(e.n, name)) return e.op;
    return -1;
}

int32_t lfhe_evaluate(lfhe_ctx* ctx, uint32_t, uint32_t, const uint8_t*, uint64_t, const uint8_t*, uint64_t, int32_t,
                      const uint8_t*, uint64_t, uint8_t*, uint64_t*) {
    if (!ctx) return LFHE_ERR_ARG;
    ctx->err = "operator layer not built yet";
    return LFHE_ERR_UNSUPPORTED;
}

int32_t lfhe_evaluate_batch(lfhe_ctx* ctx, uint32_t, uint32_t, uint64_t, const uint8_t*, const uint8_t*, int32_t,
                            const uint8_t*, uint8_t*) {
    if (!ctx) return LFHE_ERR_ARG;
    ctx->err = "operator layer not built yet";
    return LFHE_ERR_UNSUPPORTED;
}

int32_t lfhe_pbs_count(const lfhe_ctx* ctx, uint64_t* out) {
    if (!ctx || !out) return LFHE_ERR_ARG;
    *out = ctx->pbs_count;
    return LFHE_OK;
}

}  // extern "C"
