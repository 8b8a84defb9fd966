// Host-side TFHE pieces that are not on the throughput path: the deterministic generator, key
// generation, LWE encryption / decryption of radix blocks, accumulator (LUT) construction.
#pragma once
#include <stdint.h>

#include <vector>

#include "engine.h"

namespace lfhe {

// ChaCha20 keystream generator (RFC 8439 block function, 64-bit block counter, 64-bit stream id)
class ChaCha {
public:
    ChaCha() = default;
    ChaCha(const uint8_t seed[32], uint64_t stream) { init(seed, stream); }
    void init(const uint8_t seed[32], uint64_t stream);
    uint64_t next();
    // TUniform(b): integer of [-2^b, 2^b] (end points at half weight) as a torus element
    uint64_t tuniform(uint32_t b);
    // the same shape on [-bound, bound], returned as an offset in [0, 2 bound]
    uint64_t tuniform_bound(uint64_t bound);

private:
    void refill();
    uint32_t key_[8] = {0};
    uint64_t counter_ = 0, stream_ = 0;
    uint32_t buf_[16] = {0};
    uint32_t pos_ = 16;
};

void gen_binary_key(ChaCha& r, uint64_t* key, size_t len);
// b = <a, s> + plaintext + e
void lwe_encrypt(const uint64_t* sk, size_t dim, uint64_t plaintext, uint32_t noise_log2, ChaCha& r, uint64_t* ct);
uint64_t lwe_phase(const uint64_t* sk, size_t dim, const uint64_t* ct);

// bootstrapping key bsk[n][row][poly][N] (GGSW(s_i) = Z + s_i G, one level of base 2^base_log) and
// key-switching key ksk[N][level][n+1]; `threads` host threads
void gen_bsk(const EngineParams& p, const uint64_t* lwe_sk, const uint64_t* glwe_sk, const uint8_t seed[32],
             uint64_t* bsk, int threads);
// the bootstrapping key over the prime ring modulus (EngineParams::bsk_modulus), same layout; with
// EngineParams::bsk_grouping == 2 four GGSWs per pair of key bits: bsk[group][4][row][poly][N]
void gen_bsk_qp(const EngineParams& p, const uint64_t* lwe_sk, const uint64_t* glwe_sk, const uint8_t seed[32],
                uint64_t* bsk, int threads);
// stream_base separates the generator streams of different keys made from one seed
void gen_ksk(const EngineParams& p, const uint64_t* glwe_sk, const uint64_t* lwe_sk, const uint8_t seed[32],
             uint64_t* ksk, int threads, uint64_t stream_base = 0x200000000ull);

// packing key-switching key of the compression scheme: for every bit i of the big (GLWE-flattened)
// key and level lv a GLWE encryption, under the compression secret comp_sk[kc][nc], of the constant
// polynomial s_i * 2^(64 - base_log (lv + 1)); out[N][level][(kc + 1) nc] (mask polynomials, then body)
void gen_packing_ksk(const CompShape& c, const uint64_t* glwe_sk, const uint64_t* comp_sk, const uint8_t seed[32],
                     uint64_t* out, int threads);

// encryptions of zero under the small key for the modulus-switch noise reduction:
// zeros[ms_zeros_count][n+1]
void gen_ms_zeros(const EngineParams& p, const uint64_t* lwe_sk, const uint8_t seed[32], uint64_t* zeros, int threads);

// Compact (RLWE) public key under the GLWE secret key: pk = (a, b = a (*) s + e), 2N words.
void gen_public_key(const EngineParams& p, const uint64_t* glwe_sk, const uint8_t seed[32], uint64_t* pk);
// Public-key encryption of up to N plaintexts (already scaled by delta) into LWE blocks under the
// flattened GLWE key: one RLWE encryption (u binary, c0 = a u + e1, c1 = b u + e2 + M), then the
// sample extraction of coefficient i for block i.  out: [count][N+1].
void pk_encrypt_blocks(const EngineParams& p, const uint64_t* pk, const uint64_t* plaintexts, size_t count,
                       ChaCha& r, uint64_t* out);

// body polynomial of the accumulator of f given as a table over [0, space): box-encoded, rotated
// left by half a box with the wrapped part negated
void make_lut_body(uint32_t space, uint64_t delta, const uint64_t* table, uint64_t* body /* [N] */);

}  // namespace lfhe
