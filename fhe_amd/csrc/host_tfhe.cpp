#include "host_tfhe.h"

#include <cstring>
#include <thread>
#include <vector>

namespace lfhe {

static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

void ChaCha::init(const uint8_t seed[32], uint64_t stream) {
    for (int i = 0; i < 8; ++i)
        key_[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) |
                  ((uint32_t)seed[4 * i + 3] << 24);
    counter_ = 0;
    stream_ = stream;
    pos_ = 16;
}

void ChaCha::refill() {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; ++i) s[4 + i] = key_[i];
    s[12] = (uint32_t)counter_;
    s[13] = (uint32_t)(counter_ >> 32);
    s[14] = (uint32_t)stream_;
    s[15] = (uint32_t)(stream_ >> 32);
    uint32_t x[16];
    std::memcpy(x, s, sizeof x);
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    };
    for (int i = 0; i < 10; ++i) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) buf_[i] = x[i] + s[i];
    ++counter_;
    pos_ = 0;
}

uint64_t ChaCha::next() {
    if (pos_ >= 16) refill();
    uint64_t lo = buf_[pos_], hi = buf_[pos_ + 1];
    pos_ += 2;
    return lo | (hi << 32);
}

// TUniform(b) as tfhe-rs defines it: the integers of [-2^b, 2^b], the two end points with probability
// 2^-(b+2) each, every inner value with 2^-(b+1): b + 2 random bits v give floor((v + 1) / 2) - 2^b.
uint64_t ChaCha::tuniform(uint32_t b) {
    const uint64_t v = next() & ((1ull << (b + 2)) - 1);
    return ((v + 1) >> 1) - (1ull << b);
}

// The same shape for a bound that is not a power of two: offset in [0, 2 bound] of a value of
// [-bound, bound] whose end points have half the weight of the inner values (4 bound <= 2^20, so the
// bias of the modulo is below 2^-44).
uint64_t ChaCha::tuniform_bound(uint64_t bound) { return ((next() % (4 * bound)) + 1) >> 1; }

void gen_binary_key(ChaCha& r, uint64_t* key, size_t len) {
    for (size_t i = 0; i < len; i += 64) {
        uint64_t bits = r.next();
        for (size_t j = 0; j < 64 && i + j < len; ++j) key[i + j] = (bits >> j) & 1;
    }
}

void lwe_encrypt(const uint64_t* sk, size_t dim, uint64_t plaintext, uint32_t noise_log2, ChaCha& r, uint64_t* ct) {
    uint64_t b = plaintext + r.tuniform(noise_log2);
    for (size_t i = 0; i < dim; ++i) {
        ct[i] = r.next();
        b += ct[i] & (0 - sk[i]);
    }
    ct[dim] = b;
}

uint64_t lwe_phase(const uint64_t* sk, size_t dim, const uint64_t* ct) {
    uint64_t b = ct[dim];
    for (size_t i = 0; i < dim; ++i) b -= ct[i] & (0 - sk[i]);
    return b;
}

// body += a (*) s for a binary s: sum of negacyclic rotations of a
static void add_mask_times_binary_key(const uint64_t* a, const uint64_t* s, uint64_t* body) {
    for (int i = 0; i < N; ++i) {
        if (!s[i]) continue;
        // X^i * a: coefficient j of a lands on j + i, negated past N
        for (int j = 0; j < N - i; ++j) body[j + i] += a[j];
        for (int j = N - i; j < N; ++j) body[j + i - N] -= a[j];
    }
}

template <class F>
static void parallel_for(size_t n, int threads, F&& fn) {
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    for (int w = 0; w < threads; ++w)
        pool.emplace_back([&, w]() {
            for (size_t i = w; i < n; i += threads) fn(i);
        });
    for (auto& th : pool) th.join();
}

// the bit each GGSW of the bootstrapping key encrypts: the key bits themselves, or per group of b = 2 or 3
// key bits the 2^b products over the group of s_i (bit i of the GGSW's number set) or 1 - s_i (bits past
// the end of the key count as 0); for pairs: (1-s)(1-s'), s(1-s'), (1-s)s', s s'
static std::vector<uint64_t> bsk_messages(const EngineParams& p, const uint64_t* lwe_sk) {
    std::vector<uint64_t> msg;
    const uint32_t b = p.bsk_grouping;
    if (b >= 2) {
        for (uint32_t g = 0; b * g < p.lwe_dim; ++g)
            for (uint32_t m = 0; m < (1u << b); ++m) {
                uint64_t v = 1;
                for (uint32_t i = 0; i < b; ++i) {
                    const uint64_t s = b * g + i < p.lwe_dim ? lwe_sk[b * g + i] : 0;
                    v &= (m >> i & 1) ? s : 1 - s;
                }
                msg.push_back(v);
            }
    } else {
        msg.assign(lwe_sk, lwe_sk + p.lwe_dim);
    }
    return msg;
}

void gen_bsk(const EngineParams& p, const uint64_t* lwe_sk, const uint64_t* glwe_sk, const uint8_t seed[32],
             uint64_t* bsk, int threads) {
    const std::vector<uint64_t> msg = bsk_messages(p, lwe_sk);
    const size_t rows = msg.size() * 2;  // one level, k + 1 = 2 rows per GGSW
    parallel_for(rows, threads, [&](size_t row) {
        ChaCha r(seed, 0x100000000ull + row);  // one stream per GLWE row: order-independent
        uint64_t* mask = bsk + row * 2 * N;
        uint64_t* body = mask + N;
        for (int j = 0; j < N; ++j) mask[j] = r.next();
        for (int j = 0; j < N; ++j) body[j] = r.tuniform(p.glwe_noise_log2);
        add_mask_times_binary_key(mask, glwe_sk, body);
        const size_t i = row >> 1;
        const int rr = (int)(row & 1);
        (rr == 0 ? mask : body)[0] += msg[i] << (64 - p.pbs_base_log);
    });
}

// the same key over the ring modulus q': residues in [0, q'), noise of the set's own shape scaled to
// q' -- the integers of [-E, E] with the end points at half weight, E = round(2^glwe_noise_log2 q' /
// 2^64) (variance 8.5 for E = 5, where the set's TUniform(17) scaled by q' / 2^64 has 8.31) --, gadget
// round(q' / 2^base_log)
void gen_bsk_qp(const EngineParams& p, const uint64_t* lwe_sk, const uint64_t* glwe_sk, const uint8_t seed[32],
                uint64_t* bsk, int threads) {
    const uint64_t Q = QP_U;
    const uint64_t G = (Q + (1ull << (p.pbs_base_log - 1))) >> p.pbs_base_log;
    const uint64_t E = (uint64_t)((((unsigned __int128)1 << p.glwe_noise_log2) * Q + ((unsigned __int128)1 << 63)) >> 64);
    const std::vector<uint64_t> msg = bsk_messages(p, lwe_sk);
    const size_t rows = msg.size() * 2;
    parallel_for(rows, threads, [&](size_t row) {
        ChaCha r(seed, 0x400000000ull + row);
        uint64_t* mask = bsk + row * 2 * N;
        uint64_t* body = mask + N;
        for (int j = 0; j < N; ++j) mask[j] = (uint64_t)(((unsigned __int128)r.next() * Q) >> 64);
        for (int j = 0; j < N; ++j) {
            const uint64_t e = r.tuniform_bound(E);
            body[j] = e >= E ? e - E : e + Q - E;
        }
        // body += mask (*) s over Z_q' for the binary key s: sum of negacyclic rotations, kept below 2^63
        // by reducing every 4096 additions (each term is below q' < 2^50)
        std::vector<uint64_t> pos(N, 0), negs(N, 0);
        int pending = 0;
        for (int i = 0; i < N; ++i) {
            if (!glwe_sk[i]) continue;
            for (int j = 0; j < N - i; ++j) pos[j + i] += mask[j];
            for (int j = N - i; j < N; ++j) negs[j + i - N] += mask[j];
            if (++pending == 4096) { for (int j = 0; j < N; ++j) { pos[j] %= Q; negs[j] %= Q; } pending = 0; }
        }
        for (int j = 0; j < N; ++j) body[j] = (body[j] + pos[j] % Q + Q - negs[j] % Q) % Q;
        const size_t i = row >> 1;
        const int rr = (int)(row & 1);
        if (msg[i]) { uint64_t& w = (rr == 0 ? mask : body)[0]; w = (w + G) % Q; }
    });
}

void gen_ksk(const EngineParams& p, const uint64_t* glwe_sk, const uint64_t* lwe_sk, const uint8_t seed[32],
             uint64_t* ksk, int threads, uint64_t stream_base) {
    const size_t n1 = (size_t)p.lwe_dim + 1;
    parallel_for((size_t)N, threads, [&](size_t i) {
        ChaCha r(seed, stream_base + i);
        for (uint32_t lv = 0; lv < p.ks_level; ++lv)
            lwe_encrypt(lwe_sk, p.lwe_dim, glwe_sk[i] << (64 - p.ks_base_log * (lv + 1)), p.lwe_noise_log2, r,
                        ksk + (i * p.ks_level + lv) * n1);
    });
}

void gen_packing_ksk(const CompShape& c, const uint64_t* glwe_sk, const uint64_t* comp_sk, const uint8_t seed[32],
                     uint64_t* out, int threads) {
    const size_t nc = (size_t)c.nc, words = (size_t)c.glwe_words();
    parallel_for((size_t)N * c.ks_level, threads, [&](size_t row) {
        ChaCha r(seed, 0x700000000ull + row);
        uint64_t* g = out + row * words;
        uint64_t* body = g + (size_t)c.kc * nc;
        for (size_t j = 0; j < (size_t)c.kc * nc; ++j) g[j] = r.next();
        for (size_t j = 0; j < nc; ++j) body[j] = r.tuniform(c.noise_log2);
        for (int p = 0; p < c.kc; ++p) {        // body += a_p (*) S_p, negacyclic, S_p binary
            const uint64_t* a = g + (size_t)p * nc;
            const uint64_t* s = comp_sk + (size_t)p * nc;
            for (size_t i = 0; i < nc; ++i) {
                if (!s[i]) continue;
                for (size_t j = 0; j < nc - i; ++j) body[j + i] += a[j];
                for (size_t j = nc - i; j < nc; ++j) body[j + i - nc] -= a[j];
            }
        }
        const size_t i = row / c.ks_level, lv = row % c.ks_level;
        body[0] += glwe_sk[i] << (64 - c.ks_base_log * (lv + 1));
    });
}

void gen_ms_zeros(const EngineParams& p, const uint64_t* lwe_sk, const uint8_t seed[32], uint64_t* zeros, int threads) {
    const size_t n1 = (size_t)p.lwe_dim + 1;
    parallel_for((size_t)p.ms_zeros_count, threads, [&](size_t i) {
        ChaCha r(seed, 0x300000000ull + i);
        lwe_encrypt(lwe_sk, p.lwe_dim, 0, p.lwe_noise_log2, r, zeros + i * n1);
    });
}

void gen_public_key(const EngineParams& p, const uint64_t* glwe_sk, const uint8_t seed[32], uint64_t* pk) {
    ChaCha r(seed, 5);
    uint64_t* a = pk;
    uint64_t* b = pk + N;
    for (int j = 0; j < N; ++j) a[j] = r.next();
    for (int j = 0; j < N; ++j) b[j] = r.tuniform(p.glwe_noise_log2);
    add_mask_times_binary_key(a, glwe_sk, b);
}

void pk_encrypt_blocks(const EngineParams& p, const uint64_t* pk, const uint64_t* plaintexts, size_t count,
                       ChaCha& r, uint64_t* out) {
    std::vector<uint64_t> u(N), c0(N), c1(N);
    gen_binary_key(r, u.data(), N);
    for (int j = 0; j < N; ++j) c0[j] = r.tuniform(p.glwe_noise_log2);
    for (int j = 0; j < N; ++j) c1[j] = r.tuniform(p.glwe_noise_log2) + ((size_t)j < count ? plaintexts[j] : 0);
    add_mask_times_binary_key(pk, u.data(), c0.data());       // c0 = a (*) u + e1
    add_mask_times_binary_key(pk + N, u.data(), c1.data());   // c1 = b (*) u + e2 + M
    for (size_t i = 0; i < count; ++i) {
        uint64_t* o = out + i * (size_t)(N + 1);
        for (size_t j = 0; j <= i; ++j) o[j] = c0[i - j];
        for (size_t j = i + 1; j < (size_t)N; ++j) o[j] = 0 - c0[N + i - j];
        o[N] = c1[i];
    }
}

void make_lut_body(uint32_t space, uint64_t delta, const uint64_t* table, uint64_t* body) {
    const uint32_t box = N / space, half = box / 2;
    std::vector<uint64_t> tmp(N);
    for (uint32_t i = 0; i < space; ++i)
        for (uint32_t j = 0; j < box; ++j) tmp[i * box + j] = table[i] * delta;
    for (uint32_t j = 0; j < half; ++j) tmp[j] = 0 - tmp[j];
    for (uint32_t j = 0; j < (uint32_t)N; ++j) body[j] = tmp[(j + half) % N];
}

}  // namespace lfhe
