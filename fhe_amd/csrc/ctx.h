// The context behind the C ABI: engine + host-side key material + generator state.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/lfhe.h"
#include "engine.h"
#include "host_tfhe.h"

struct lfhe_ctx {
    lfhe::EngineParams prm;
    std::unique_ptr<lfhe::Engine> eng;
    std::string err;
    uint8_t seed[32];
    lfhe::ChaCha enc_rng;          // encryption randomness (stream 16)
    std::vector<uint64_t> lwe_sk;  // n words of 0/1
    std::vector<uint64_t> glwe_sk; // N words of 0/1
    bool have_client = false;
    std::vector<uint64_t> bsk, ksk;  // host copies of the server keys (coefficient domain)
    std::vector<uint64_t> ms_zeros;  // [ms_zeros_count][n+1], empty when the key set has none
    bool have_server = false;
    std::vector<uint64_t> pke_sk;    // secret key of public-key encryptions made elsewhere (N words), optional
    uint64_t pbs_count = 0;
    std::unique_ptr<struct lfhe_radix> radix;  // operator layer state (radix.cpp)

    lfhe_ctx();
    ~lfhe_ctx();  // defined in radix.cpp, where lfhe_radix is complete

    uint64_t delta() const { return (1ull << 63) / ((uint64_t)prm.message_modulus * prm.carry_modulus); }
    uint32_t space() const { return prm.message_modulus * prm.carry_modulus; }
    uint32_t big1() const { return (uint32_t)lfhe::N + 1; }
};

namespace lfhe {
// runs fn, mapping exceptions to status codes and recording the message
template <class F>
int32_t guarded(lfhe_ctx* ctx, F&& fn) {
    if (!ctx) return LFHE_ERR_ARG;
    try {
        return fn();
    } catch (const std::invalid_argument& e) {
        ctx->err = e.what();
        return LFHE_ERR_ARG;
    } catch (const std::logic_error& e) {
        ctx->err = e.what();
        return LFHE_ERR_STATE;
    } catch (const std::exception& e) {
        ctx->err = e.what();
        // "unsupported ...": something this library does not do (a key without a device image), not a device fault
        return ctx->err.compare(0, 11, "unsupported") == 0 ? LFHE_ERR_UNSUPPORTED : LFHE_ERR_DEVICE;
    }
}
}  // namespace lfhe
