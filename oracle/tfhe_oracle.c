/*
 * tfhe_oracle.c -- CPU restatement of the TFHE programmable-bootstrap path (see tfhe_oracle.h for
 * what it restates, what pins it and what stays unpinned).  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C11, integers only.  Polynomial products are exact in Z[X]/(X^N+1) and then wrapped to
 * 2^64, which is the semantics of tfhe-rs polynomial_wrapping_mul used by the reference at
 * ml/extensions/rust/src/computations.rs:50-54.  Two routes are kept: schoolbook (the definition)
 * and a two-prime NTT + CRT route (the fast one); tests check them against each other.
 */
#include "tfhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef __int128 i128;
typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;

/* ------------------------------------------------------------------------------------------- */
/* ChaCha20 keystream (RFC 8439 quarter round / block function), 64-bit block counter.          */
/* ------------------------------------------------------------------------------------------- */
static inline u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d)                                                                             \
    do {                                                                                           \
        a += b; d ^= a; d = rotl32(d, 16);                                                         \
        c += d; b ^= c; b = rotl32(b, 12);                                                         \
        a += b; d ^= a; d = rotl32(d, 8);                                                          \
        c += d; b ^= c; b = rotl32(b, 7);                                                          \
    } while (0)

static void chacha_block(orc_rng* r) {
    u32 s[16], x[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
    for (int i = 0; i < 8; ++i) s[4 + i] = r->key[i];
    s[12] = (u32)r->counter; s[13] = (u32)(r->counter >> 32);
    s[14] = (u32)r->stream;  s[15] = (u32)(r->stream >> 32);
    memcpy(x, s, sizeof x);
    for (int i = 0; i < 10; ++i) {
        QR(x[0], x[4], x[8], x[12]); QR(x[1], x[5], x[9], x[13]);
        QR(x[2], x[6], x[10], x[14]); QR(x[3], x[7], x[11], x[15]);
        QR(x[0], x[5], x[10], x[15]); QR(x[1], x[6], x[11], x[12]);
        QR(x[2], x[7], x[8], x[13]); QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i) r->buf[i] = x[i] + s[i];
    r->counter++;
    r->pos = 0;
}

void orc_rng_init(orc_rng* r, const uint8_t seed[32], uint64_t stream) {
    for (int i = 0; i < 8; ++i)
        r->key[i] = (u32)seed[4 * i] | ((u32)seed[4 * i + 1] << 8) | ((u32)seed[4 * i + 2] << 16) |
                    ((u32)seed[4 * i + 3] << 24);
    r->counter = 0;
    r->stream = stream;
    r->pos = 16;
}

uint64_t orc_rng_u64(orc_rng* r) {
    if (r->pos >= 16) chacha_block(r);
    u64 lo = r->buf[r->pos], hi = r->buf[r->pos + 1];
    r->pos += 2;
    return lo | (hi << 32);
}

/* TUniform(b) as tfhe-rs defines it: the integers of [-2^b, 2^b], the two end points with
 * probability 2^-(b+2) each, every inner value with 2^-(b+1).  b + 2 random bits v give
 * floor((v + 1) / 2) - 2^b: v = 0 and v = 2^(b+2) - 1 are the end points, every other value is hit
 * by two v.  Variance (2^(2b+1) + 1) / 6. */
uint64_t orc_rng_tuniform(orc_rng* r, uint32_t b) {
    const u64 v = orc_rng_u64(r) & (((u64)1 << (b + 2)) - 1);
    return ((v + 1) >> 1) - ((u64)1 << b);
}

/* The same shape for a bound E that is not a power of two (keys over the prime ring modulus, whose
 * noise bound is 2^b scaled to that modulus): integers of [-E, E], end points at half weight.
 * Returned as an offset in [0, 2E]. */
uint64_t orc_rng_tuniform_bound(orc_rng* r, uint64_t E) {
    const u64 v = orc_rng_u64(r) % (4 * E);   /* 4E <= 2^20: the modulo bias is below 2^-44 */
    return (v + 1) >> 1;
}

/* ------------------------------------------------------------------------------------------- */
/* modular arithmetic                                                                          */
/* ------------------------------------------------------------------------------------------- */
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t p) { return (u64)(((u128)a * b) % p); }

uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t p) {
    u64 r = 1 % p;
    a %= p;
    while (e) {
        if (e & 1) r = orc_mulmod(r, a, p);
        a = orc_mulmod(a, a, p);
        e >>= 1;
    }
    return r;
}

static int is_prime_u64(u64 n) {
    if (n < 2) return 0;
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (size_t i = 0; i < sizeof small / sizeof small[0]; ++i) {
        if (n == small[i]) return 1;
        if (n % small[i] == 0) return 0;
    }
    u64 d = n - 1;
    int s = 0;
    while ((d & 1) == 0) { d >>= 1; ++s; }
    /* these 12 bases are a deterministic Miller-Rabin certificate below 2^64 */
    for (size_t i = 0; i < sizeof small / sizeof small[0]; ++i) {
        u64 x = orc_powmod(small[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int k = 1; k < s; ++k) {
            x = orc_mulmod(x, x, n);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

/* Montgomery arithmetic mod an odd p < 2^63, R = 2^64 */
typedef struct {
    u64 p, pinv_neg, r2; /* -p^-1 mod 2^64, R^2 mod p */
} mont_t;

static inline u64 mont_redc(const mont_t* m, u128 t) {
    u64 q = (u64)t * m->pinv_neg;
    u128 u = t + (u128)q * m->p;
    u64 r = (u64)(u >> 64);
    return r >= m->p ? r - m->p : r;
}
static inline u64 mont_mul(const mont_t* m, u64 a, u64 b) { return mont_redc(m, (u128)a * b); }
static inline u64 to_mont(const mont_t* m, u64 a) { return mont_mul(m, a, m->r2); }
static void mont_init(mont_t* m, u64 p) {
    m->p = p;
    u64 inv = p; /* Newton: inv = p^-1 mod 2^64 */
    for (int i = 0; i < 6; ++i) inv *= 2 - p * inv;
    m->pinv_neg = (u64)0 - inv;
    u64 r = (u64)(((u128)1 << 64) % p);
    m->r2 = orc_mulmod(r, r, p);
}

/* ------------------------------------------------------------------------------------------- */
/* negacyclic NTT over one prime (tables in Montgomery form, data in plain form)               */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    mont_t m;
    u64 psi;          /* primitive 2N-th root of unity (plain) */
    u64* w_fwd;       /* psi^bitrev(i), Montgomery form, i < N */
    u64* w_inv;       /* psi^-bitrev(i), Montgomery form */
    u64 ninv_mont;    /* N^-1, Montgomery form */
} ntt_prime;

static u32 bitrev(u32 x, u32 bits) {
    u32 r = 0;
    for (u32 i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

static u64 find_psi(u64 p, u32 N) {
    /* smallest g whose (p-1)/2N power has order exactly 2N (its N-th power is -1) */
    for (u64 g = 2;; ++g) {
        u64 c = orc_powmod(g, (p - 1) / (2 * (u64)N), p);
        if (orc_powmod(c, N, p) == p - 1) return c;
    }
}

static void ntt_prime_init(ntt_prime* t, u64 p, u32 N) {
    u32 logn = 0;
    while ((1u << logn) < N) ++logn;
    mont_init(&t->m, p);
    t->psi = find_psi(p, N);
    u64 psi_inv = orc_powmod(t->psi, p - 2, p);
    t->w_fwd = (u64*)malloc(sizeof(u64) * N);
    t->w_inv = (u64*)malloc(sizeof(u64) * N);
    for (u32 i = 0; i < N; ++i) {
        u32 e = bitrev(i, logn);
        t->w_fwd[i] = to_mont(&t->m, orc_powmod(t->psi, e, p));
        t->w_inv[i] = to_mont(&t->m, orc_powmod(psi_inv, e, p));
    }
    t->ninv_mont = to_mont(&t->m, orc_powmod(N, p - 2, p));
}

/* forward: natural order in, bit-reversed order out (Cooley-Tukey, twist merged) */
static void ntt_fwd(const ntt_prime* t, u64* a, u32 N) {
    const u64 p = t->m.p;
    u32 len = N;
    for (u32 m = 1; m < N; m <<= 1) {
        len >>= 1;
        for (u32 i = 0; i < m; ++i) {
            const u64 w = t->w_fwd[m + i];
            u64* x = a + 2 * i * len;
            for (u32 j = 0; j < len; ++j) {
                u64 u = x[j], v = mont_mul(&t->m, x[j + len], w);
                u64 s = u + v; if (s >= p) s -= p;
                u64 d = u >= v ? u - v : u + p - v;
                x[j] = s; x[j + len] = d;
            }
        }
    }
}

/* inverse: bit-reversed order in, natural order out (Gentleman-Sande), scaled by 1/N */
static void ntt_inv(const ntt_prime* t, u64* a, u32 N) {
    const u64 p = t->m.p;
    u32 len = 1;
    for (u32 m = N >> 1; m >= 1; m >>= 1) {
        for (u32 i = 0; i < m; ++i) {
            const u64 w = t->w_inv[m + i];
            u64* x = a + 2 * i * len;
            for (u32 j = 0; j < len; ++j) {
                u64 u = x[j], v = x[j + len];
                u64 s = u + v; if (s >= p) s -= p;
                u64 d = u >= v ? u - v : u + p - v;
                x[j] = s; x[j + len] = mont_mul(&t->m, d, w);
            }
        }
        len <<= 1;
    }
    for (u32 j = 0; j < N; ++j) a[j] = mont_mul(&t->m, a[j], t->ninv_mont);
}

void orc_ntt_definition(uint64_t p, uint64_t psi, const uint64_t* in, uint64_t* out, uint32_t N) {
    for (u32 j = 0; j < N; ++j) {
        const u64 step = orc_powmod(psi, 2 * (u64)j + 1, p);
        u64 acc = 0, w = 1;
        for (u32 i = 0; i < N; ++i) {
            acc = (u64)(((u128)acc + (u128)orc_mulmod(in[i] % p, w, p)) % p);
            w = orc_mulmod(w, step, p);
        }
        out[j] = acc;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* context                                                                                     */
/* ------------------------------------------------------------------------------------------- */
#define ORC_QP ((u64)703687441727489ull)

struct orc_ctx {
    orc_params prm;
    u32 logN;
    ntt_prime pr[2];
    u64 p0_inv_mod_p1; /* plain */
    ntt_prime qp;        /* the prime ring modulus of the single-modulus bootstrap */
    orc_ms_params ms;    /* zeros_count == 0: the noise-reduction step is off */
    const u64* ms_zeros; /* borrowed */
};

void orc_default_params(orc_params* p) {
    /* decoded from the reference fixture sdk/relayer/src/test/keys/privateKey.bin (offsets
     * 23867-24007): PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 */
    p->lwe_dim = 918; p->glwe_dim = 1; p->poly_size = 2048;
    p->pbs_base_log = 23; p->pbs_level = 1;
    p->ks_base_log = 4; p->ks_level = 4;
    p->message_modulus = 4; p->carry_modulus = 4;
    p->lwe_noise_log2 = 45; p->glwe_noise_log2 = 17;
}

orc_ctx* orc_ctx_new(const orc_params* p) {
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof *c);
    c->prm = *p;
    while ((1u << c->logN) < p->poly_size) ++c->logN;
    /* the two largest primes below 2^62 that are 1 mod 2^16 */
    u64 cand = ((u64)1 << 62) + 1;
    int found = 0;
    while (found < 2) {
        cand -= (u64)1 << 16;
        if (is_prime_u64(cand)) ntt_prime_init(&c->pr[found++], cand, p->poly_size);
    }
    c->p0_inv_mod_p1 = orc_powmod(c->pr[0].m.p % c->pr[1].m.p, c->pr[1].m.p - 2, c->pr[1].m.p);
    ntt_prime_init(&c->qp, ORC_QP, p->poly_size);
    return c;
}

void orc_ctx_free(orc_ctx* c) {
    if (!c) return;
    for (int i = 0; i < 2; ++i) { free(c->pr[i].w_fwd); free(c->pr[i].w_inv); }
    free(c->qp.w_fwd); free(c->qp.w_inv);
    free(c);
}

/* ------------------------------------------------------------------------------------------- */
/* polynomial products                                                                         */
/* ------------------------------------------------------------------------------------------- */
void orc_negacyclic_mul_schoolbook(uint64_t* out, const int64_t* small, const uint64_t* b, uint32_t N) {
    memset(out, 0, sizeof(u64) * N);
    for (u32 i = 0; i < N; ++i) {
        const u64 s = (u64)small[i];
        if (!s) continue;
        for (u32 j = 0; j < N; ++j) {
            u32 k = i + j;
            u64 t = s * b[j];
            if (k < N) out[k] += t; else out[k - N] -= t;
        }
    }
}

static inline u64 reduce_signed(i64 v, u64 p) {
    i64 r = v % (i64)p;
    return r < 0 ? (u64)(r + (i64)p) : (u64)r;
}

/* residues (mod p0, mod p1) of an integer known to lie in (-P/2, P/2), P = p0*p1 -> that integer
 * wrapped to 2^64 */
static inline u64 crt_wrap(const orc_ctx* c, u64 y0, u64 y1) {
    const u64 p0 = c->pr[0].m.p, p1 = c->pr[1].m.p;
    u64 d = (y1 + p1 - y0 % p1) % p1;
    u64 k = orc_mulmod(d, c->p0_inv_mod_p1, p1);
    u128 x = (u128)y0 + (u128)p0 * k;          /* in [0, P) */
    const u128 P = (u128)p0 * p1;
    if (x > P / 2) return (u64)x - (u64)P;     /* x - P, wrapped */
    return (u64)x;
}

typedef struct {
    u64* v[2]; /* NTT-domain, Montgomery form, one array per prime */
} poly_hat;

static void torus_poly_to_hat(const orc_ctx* c, const u64* b, u64* h0, u64* h1) {
    const u32 N = c->prm.poly_size;
    for (u32 i = 0; i < N; ++i) {
        i64 s = (i64)b[i]; /* centred representative in [-2^63, 2^63) */
        h0[i] = reduce_signed(s, c->pr[0].m.p);
        h1[i] = reduce_signed(s, c->pr[1].m.p);
    }
    ntt_fwd(&c->pr[0], h0, N);
    ntt_fwd(&c->pr[1], h1, N);
    for (u32 i = 0; i < N; ++i) { h0[i] = to_mont(&c->pr[0].m, h0[i]); h1[i] = to_mont(&c->pr[1].m, h1[i]); }
}

void orc_negacyclic_mul_ntt(const orc_ctx* c, uint64_t* out, const int64_t* small, const uint64_t* b) {
    const u32 N = c->prm.poly_size;
    u64* buf = (u64*)malloc(sizeof(u64) * N * 4);
    u64 *a0 = buf, *a1 = buf + N, *b0 = buf + 2 * N, *b1 = buf + 3 * N;
    for (u32 i = 0; i < N; ++i) {
        a0[i] = reduce_signed(small[i], c->pr[0].m.p);
        a1[i] = reduce_signed(small[i], c->pr[1].m.p);
    }
    ntt_fwd(&c->pr[0], a0, N); ntt_fwd(&c->pr[1], a1, N);
    torus_poly_to_hat(c, b, b0, b1);
    for (u32 i = 0; i < N; ++i) {
        a0[i] = mont_mul(&c->pr[0].m, a0[i], b0[i]);
        a1[i] = mont_mul(&c->pr[1].m, a1[i], b1[i]);
    }
    ntt_inv(&c->pr[0], a0, N); ntt_inv(&c->pr[1], a1, N);
    for (u32 i = 0; i < N; ++i) out[i] = crt_wrap(c, a0[i], a1[i]);
    free(buf);
}

void orc_monomial_mul(uint64_t* out, const uint64_t* in, uint32_t rot, uint32_t N) {
    rot %= 2 * N;
    for (u32 j = 0; j < N; ++j) {
        u32 k = j + rot;
        if (k >= 2 * N) k -= 2 * N;
        if (k < N) out[k] = in[j]; else out[k - N] = (u64)0 - in[j];
    }
}

/* ------------------------------------------------------------------------------------------- */
/* decomposition (tfhe-rs SignedDecomposer: closest representable, balanced digits)             */
/* ------------------------------------------------------------------------------------------- */
void orc_decompose(uint64_t x, uint32_t base_log, uint32_t levels, int64_t* digits) {
    const u32 rep_bits = base_log * levels;
    const u32 shift = 64 - rep_bits;
    /* closest representable: keep the top rep_bits bits, rounding half up */
    u64 state = (x >> (shift - 1));
    state = (state + 1) >> 1;
    state &= (rep_bits == 64) ? ~(u64)0 : (((u64)1 << rep_bits) - 1);
    const u64 mask = ((u64)1 << base_log) - 1;
    for (u32 j = levels; j-- > 0;) { /* least significant digit (highest level) first */
        u64 outp = state & mask;
        state >>= base_log;
        u64 carry = ((outp - 1) | state) & outp;
        carry >>= base_log - 1;
        state += carry;
        digits[j] = (i64)outp - (i64)(carry << base_log);
    }
}

/* ------------------------------------------------------------------------------------------- */
/* keys and encryption                                                                         */
/* ------------------------------------------------------------------------------------------- */
void orc_gen_binary_key(orc_rng* r, uint64_t* key, size_t len) {
    for (size_t i = 0; i < len; i += 64) {
        u64 bits = orc_rng_u64(r);
        for (size_t j = 0; j < 64 && i + j < len; ++j) key[i + j] = (bits >> j) & 1;
    }
}

void orc_lwe_encrypt(const uint64_t* sk, size_t dim, uint64_t plaintext, uint32_t noise_log2,
                     orc_rng* r, uint64_t* ct) {
    u64 b = plaintext + orc_rng_tuniform(r, noise_log2);
    for (size_t i = 0; i < dim; ++i) {
        ct[i] = orc_rng_u64(r);
        b += ct[i] * sk[i];
    }
    ct[dim] = b;
}

uint64_t orc_lwe_phase(const uint64_t* sk, size_t dim, const uint64_t* ct) {
    u64 b = ct[dim];
    for (size_t i = 0; i < dim; ++i) b -= ct[i] * sk[i];
    return b;
}

void orc_lwe_trivial(size_t dim, uint64_t plaintext, uint64_t* ct) {
    memset(ct, 0, sizeof(u64) * dim);
    ct[dim] = plaintext;
}

void orc_glwe_encrypt(const orc_ctx* c, const uint64_t* glwe_sk, const uint64_t* pt, orc_rng* r,
                      uint64_t* ct) {
    const u32 N = c->prm.poly_size, k = c->prm.glwe_dim;
    u64* body = ct + (size_t)k * N;
    u64* tmp = (u64*)malloc(sizeof(u64) * N);
    i64* s = (i64*)malloc(sizeof(i64) * N);
    for (u32 j = 0; j < N; ++j) body[j] = (pt ? pt[j] : 0) + orc_rng_tuniform(r, c->prm.glwe_noise_log2);
    for (u32 i = 0; i < k; ++i) {
        u64* a = ct + (size_t)i * N;
        for (u32 j = 0; j < N; ++j) { a[j] = orc_rng_u64(r); s[j] = (i64)glwe_sk[(size_t)i * N + j]; }
        orc_negacyclic_mul_ntt(c, tmp, s, a);
        for (u32 j = 0; j < N; ++j) body[j] += tmp[j];
    }
    free(tmp); free(s);
}

void orc_glwe_phase(const orc_ctx* c, const uint64_t* glwe_sk, const uint64_t* ct, uint64_t* phase) {
    const u32 N = c->prm.poly_size, k = c->prm.glwe_dim;
    u64* tmp = (u64*)malloc(sizeof(u64) * N);
    i64* s = (i64*)malloc(sizeof(i64) * N);
    memcpy(phase, ct + (size_t)k * N, sizeof(u64) * N);
    for (u32 i = 0; i < k; ++i) {
        for (u32 j = 0; j < N; ++j) s[j] = (i64)glwe_sk[(size_t)i * N + j];
        orc_negacyclic_mul_ntt(c, tmp, s, ct + (size_t)i * N);
        for (u32 j = 0; j < N; ++j) phase[j] -= tmp[j];
    }
    free(tmp); free(s);
}

size_t orc_bsk_len(const orc_params* p) {
    return (size_t)p->lwe_dim * p->pbs_level * (p->glwe_dim + 1) * (p->glwe_dim + 1) * p->poly_size;
}

void orc_gen_bsk(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r,
                 uint64_t* bsk) {
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k = P->glwe_dim;
    const size_t row_len = (size_t)(k + 1) * N;
    u64* row = bsk;
    for (u32 i = 0; i < P->lwe_dim; ++i)
        for (u32 lv = 0; lv < P->pbs_level; ++lv)
            for (u32 rr = 0; rr <= k; ++rr, row += row_len) {
                orc_glwe_encrypt(c, glwe_sk, NULL, r, row);
                /* Z + s_i * G: gadget value on the constant coefficient of polynomial rr */
                row[(size_t)rr * N] += lwe_sk[i] << (64 - P->pbs_base_log * (lv + 1));
            }
}

size_t orc_ksk_len(const orc_params* p) {
    return (size_t)p->glwe_dim * p->poly_size * p->ks_level * (p->lwe_dim + 1);
}

void orc_gen_ksk(const orc_ctx* c, const uint64_t* big_sk, const uint64_t* lwe_sk, orc_rng* r,
                 uint64_t* ksk) {
    const orc_params* P = &c->prm;
    const size_t big = (size_t)P->glwe_dim * P->poly_size;
    u64* ct = ksk;
    for (size_t i = 0; i < big; ++i)
        for (u32 lv = 0; lv < P->ks_level; ++lv, ct += P->lwe_dim + 1)
            orc_lwe_encrypt(lwe_sk, P->lwe_dim, big_sk[i] << (64 - P->ks_base_log * (lv + 1)),
                            P->lwe_noise_log2, r, ct);
}

/* ------------------------------------------------------------------------------------------- */
/* external product / cmux                                                                     */
/* ------------------------------------------------------------------------------------------- */
/* NTT-domain image of one GGSW: hat[level][row][poly][prime][N] (Montgomery form) */
static void ggsw_to_hat(const orc_ctx* c, const u64* ggsw, u64* hat) {
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k1 = P->glwe_dim + 1;
    const size_t polys = (size_t)P->pbs_level * k1 * k1;
    for (size_t q = 0; q < polys; ++q)
        torus_poly_to_hat(c, ggsw + q * N, hat + q * 2 * N, hat + q * 2 * N + N);
}

static void external_product_hat(const orc_ctx* c, const u64* hat, const u64* glwe, u64* out) {
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k1 = P->glwe_dim + 1, L = P->pbs_level;
    u64* acc = (u64*)calloc((size_t)k1 * 2 * N, sizeof(u64)); /* [poly][prime][N] */
    u64* d0 = (u64*)malloc(sizeof(u64) * 2 * N);
    u64* d1 = d0 + N;
    i64* digs = (i64*)malloc(sizeof(i64) * L);
    i64* dpoly = (i64*)malloc(sizeof(i64) * (size_t)L * N);
    for (u32 rr = 0; rr < k1; ++rr) {
        for (u32 j = 0; j < N; ++j) {
            orc_decompose(glwe[(size_t)rr * N + j], P->pbs_base_log, L, digs);
            for (u32 lv = 0; lv < L; ++lv) dpoly[(size_t)lv * N + j] = digs[lv];
        }
        for (u32 lv = 0; lv < L; ++lv) {
            for (u32 j = 0; j < N; ++j) {
                d0[j] = reduce_signed(dpoly[(size_t)lv * N + j], c->pr[0].m.p);
                d1[j] = reduce_signed(dpoly[(size_t)lv * N + j], c->pr[1].m.p);
            }
            ntt_fwd(&c->pr[0], d0, N); ntt_fwd(&c->pr[1], d1, N);
            for (u32 cc = 0; cc < k1; ++cc) {
                const u64* h = hat + (((size_t)lv * k1 + rr) * k1 + cc) * 2 * N;
                u64* a0 = acc + (size_t)cc * 2 * N;
                u64* a1 = a0 + N;
                const u64 p0 = c->pr[0].m.p, p1 = c->pr[1].m.p;
                for (u32 j = 0; j < N; ++j) {
                    u64 t0 = a0[j] + mont_mul(&c->pr[0].m, d0[j], h[j]);
                    a0[j] = t0 >= p0 ? t0 - p0 : t0;
                    u64 t1 = a1[j] + mont_mul(&c->pr[1].m, d1[j], h[N + j]);
                    a1[j] = t1 >= p1 ? t1 - p1 : t1;
                }
            }
        }
    }
    for (u32 cc = 0; cc < k1; ++cc) {
        u64* a0 = acc + (size_t)cc * 2 * N;
        u64* a1 = a0 + N;
        ntt_inv(&c->pr[0], a0, N); ntt_inv(&c->pr[1], a1, N);
        for (u32 j = 0; j < N; ++j) out[(size_t)cc * N + j] = crt_wrap(c, a0[j], a1[j]);
    }
    free(acc); free(d0); free(digs); free(dpoly);
}

void orc_external_product(const orc_ctx* c, const uint64_t* ggsw, const uint64_t* glwe, uint64_t* out) {
    const orc_params* P = &c->prm;
    const size_t polys = (size_t)P->pbs_level * (P->glwe_dim + 1) * (P->glwe_dim + 1);
    u64* hat = (u64*)malloc(sizeof(u64) * polys * 2 * P->poly_size);
    ggsw_to_hat(c, ggsw, hat);
    external_product_hat(c, hat, glwe, out);
    free(hat);
}

static void cmux_rotate_hat(const orc_ctx* c, const u64* hat, u32 rot, u64* acc) {
    const u32 N = c->prm.poly_size, k1 = c->prm.glwe_dim + 1;
    u64* diff = (u64*)malloc(sizeof(u64) * (size_t)k1 * N * 2);
    u64* prod = diff + (size_t)k1 * N;
    for (u32 q = 0; q < k1; ++q) {
        orc_monomial_mul(diff + (size_t)q * N, acc + (size_t)q * N, rot, N);
        for (u32 j = 0; j < N; ++j) diff[(size_t)q * N + j] -= acc[(size_t)q * N + j];
    }
    external_product_hat(c, hat, diff, prod);
    for (size_t j = 0; j < (size_t)k1 * N; ++j) acc[j] += prod[j];
    free(diff);
}

void orc_cmux_rotate(const orc_ctx* c, const uint64_t* ggsw, uint32_t rot, uint64_t* acc) {
    const orc_params* P = &c->prm;
    const size_t polys = (size_t)P->pbs_level * (P->glwe_dim + 1) * (P->glwe_dim + 1);
    u64* hat = (u64*)malloc(sizeof(u64) * polys * 2 * P->poly_size);
    ggsw_to_hat(c, ggsw, hat);
    cmux_rotate_hat(c, hat, rot, acc);
    free(hat);
}

/* ------------------------------------------------------------------------------------------- */
/* mod switch, key switch, blind rotation, sample extraction, PBS                              */
/* ------------------------------------------------------------------------------------------- */
uint32_t orc_modswitch(uint64_t x, uint32_t log2_2N) {
    u64 t = x >> (64 - log2_2N - 1);
    t = (t + 1) >> 1;
    return (u32)(t & (((u64)1 << log2_2N) - 1));
}

/* ---- modulus-switch noise reduction ---- */
void orc_default_ms_params(orc_ms_params* m) {
    /* reference fixture sdk/relayer/src/test/keys/privateKey.bin, offsets 24068-24110 */
    m->zeros_count = 1449;
    m->bound_log2 = 58;
    m->r_sigma_factor = 13.179852282053789;
    m->input_variance = 2.63039184094559e-07;
}

void orc_gen_ms_zeros(const orc_ctx* c, const uint64_t* lwe_sk, orc_rng* r, uint32_t count, uint64_t* zeros) {
    const u32 n = c->prm.lwe_dim;
    for (u32 i = 0; i < count; ++i)
        orc_lwe_encrypt(lwe_sk, n, 0, c->prm.lwe_noise_log2, r, zeros + (size_t)i * (n + 1));
}

static u64 floor_sqrt_u64(u64 v) {
    u64 lo = 0, hi = (u64)1 << 32; /* invariant: lo^2 <= v < hi^2 */
    while (hi - lo > 1) {
        const u64 mid = lo + (hi - lo) / 2;
        if (mid * mid <= v) lo = mid; else hi = mid;
    }
    return lo;
}

/* estimate of ct + add (add may be NULL), units of 2^-38 of the torus:
 *   e = rounded - exact per word (units 2^-64), s1x2 = 2 e_b - sum e_i, s2 = sum (e_i >> 26)^2,
 *   v = round(input_variance 2^76) + s2 / 4,
 *   estimate = |s1x2| / 2^27 + round(r_sigma 2^16) floor(sqrt(v)) / 2^16                      */
static u64 ms_estimate_sum(const orc_ctx* c, const orc_ms_params* m, const u64* ct, const u64* add) {
    const u32 n = c->prm.lwe_dim, log2n = c->logN + 1;
    i64 s1x2 = 0;
    u64 s2 = 0;
    for (u32 j = 0; j <= n; ++j) {
        const u64 a = ct[j] + (add ? add[j] : 0);
        const u64 rounded = (u64)orc_modswitch(a, log2n) << (64 - log2n);
        u64 diff = rounded - a;
        /* orc_modswitch reduces mod 2N: a value that rounds up to 2N came back as 0, which is the
         * same torus point, so the wrapped difference is the signed rounding error */
        const i64 e = (i64)diff;
        if (j == n) s1x2 += 2 * e;
        else {
            const i64 h = e >> 26;
            s1x2 -= e;
            s2 += (u64)(h * h);
        }
    }
    const u64 var_fx = (u64)llround(ldexp(m->input_variance, 76)) + (s2 >> 2);
    const u64 r_fx = (u64)llround(ldexp(m->r_sigma_factor, 16));
    const u64 mean_fx = (u64)(s1x2 < 0 ? -s1x2 : s1x2) >> 27;
    return mean_fx + ((r_fx * floor_sqrt_u64(var_fx)) >> 16);
}

uint64_t orc_ms_estimate(const orc_ctx* c, const orc_ms_params* m, const uint64_t* lwe_small) {
    return ms_estimate_sum(c, m, lwe_small, NULL);
}

int32_t orc_ms_noise_reduce(const orc_ctx* c, const orc_ms_params* m, const uint64_t* zeros, uint64_t* lwe_small) {
    const u32 n = c->prm.lwe_dim;
    const u64 bound = (u64)1 << (m->bound_log2 - 26);
    u64 best = ms_estimate_sum(c, m, lwe_small, NULL);
    int32_t pick = -1;
    if (best > bound) {
        for (u32 k = 0; k < m->zeros_count; ++k) {
            const u64 e = ms_estimate_sum(c, m, lwe_small, zeros + (size_t)k * (n + 1));
            if (e <= bound) { pick = (int32_t)k; break; }
            if (e < best) { best = e; pick = (int32_t)k; }
        }
    }
    if (pick >= 0)
        for (u32 j = 0; j <= n; ++j) lwe_small[j] += zeros[(size_t)pick * (n + 1) + j];
    return pick;
}

void orc_ctx_set_ms(orc_ctx* c, const orc_ms_params* m, const uint64_t* zeros) {
    if (m && zeros && m->zeros_count) { c->ms = *m; c->ms_zeros = zeros; }
    else { memset(&c->ms, 0, sizeof c->ms); c->ms_zeros = NULL; }
}

void orc_keyswitch(const orc_ctx* c, const uint64_t* ksk, const uint64_t* in, uint64_t* out) {
    const orc_params* P = &c->prm;
    const size_t big = (size_t)P->glwe_dim * P->poly_size;
    const u32 n = P->lwe_dim, L = P->ks_level;
    i64 digs[64];
    memset(out, 0, sizeof(u64) * n);
    out[n] = in[big];
    for (size_t i = 0; i < big; ++i) {
        orc_decompose(in[i], P->ks_base_log, L, digs);
        for (u32 lv = 0; lv < L; ++lv) {
            const u64 d = (u64)digs[lv];
            if (!d) continue;
            const u64* row = ksk + (i * L + lv) * (n + 1);
            for (u32 j = 0; j <= n; ++j) out[j] -= d * row[j];
        }
    }
}

void orc_sample_extract(const orc_ctx* c, const uint64_t* glwe, uint64_t* lwe_big) {
    const u32 N = c->prm.poly_size, k = c->prm.glwe_dim;
    for (u32 i = 0; i < k; ++i) {
        const u64* a = glwe + (size_t)i * N;
        u64* o = lwe_big + (size_t)i * N;
        o[0] = a[0];
        for (u32 j = 1; j < N; ++j) o[j] = (u64)0 - a[N - j];
    }
    lwe_big[(size_t)k * N] = glwe[(size_t)k * N];
}

/* prepared (NTT-domain) bootstrapping key */
struct orc_prepared_bsk {
    u64* hat;        /* [i][level][row][poly][prime][N] */
    size_t per_ggsw; /* words per GGSW image */
    u32 grouping;    /* key bits per blind-rotation step (1, or 2 for a grouped key) */
    u32 raw;         /* 1: hat holds the GGSWs as they are (coefficient domain, 2^64): grouped key over 2^64 */
};

orc_prepared_bsk* orc_bsk_prepare(const orc_ctx* c, const uint64_t* bsk) {
    const orc_params* P = &c->prm;
    const size_t polys = (size_t)P->pbs_level * (P->glwe_dim + 1) * (P->glwe_dim + 1);
    orc_prepared_bsk* b = (orc_prepared_bsk*)malloc(sizeof *b);
    b->per_ggsw = polys * 2 * P->poly_size;
    b->grouping = 1;
    b->raw = 0;
    b->hat = (u64*)malloc(sizeof(u64) * b->per_ggsw * P->lwe_dim);
    for (u32 i = 0; i < P->lwe_dim; ++i)
        ggsw_to_hat(c, bsk + (size_t)i * polys * P->poly_size, b->hat + (size_t)i * b->per_ggsw);
    return b;
}

void orc_bsk_prepared_free(orc_prepared_bsk* b) {
    if (!b) return;
    free(b->hat);
    free(b);
}

/* Grouped key over 2^64: the LWE key bits `bits` at a time (2 or 3; bits past the end of the key count
 * as 0); group g holds 2^bits GGSWs over the torus, GGSW m encrypting the product over the group's
 * bits i of s_i (bit i of m set) or 1 - s_i: for pairs (1-s)(1-s'), s(1-s'), (1-s)s' and s s'. */
size_t orc_bsk_grouped_count_n(const orc_params* p, uint32_t bits) {
    return ((size_t)1 << bits) * (((size_t)p->lwe_dim + bits - 1) / bits);
}
size_t orc_bsk_grouped_count(const orc_params* p) { return orc_bsk_grouped_count_n(p, 2); }

void orc_gen_bsk_grouped_n(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r,
                           uint32_t bits, uint64_t* bsk) {
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k = P->glwe_dim;
    const size_t row_len = (size_t)(k + 1) * N;
    const u32 groups = (P->lwe_dim + bits - 1) / bits, terms = 1u << bits;
    u64* row = bsk;
    for (u32 g = 0; g < groups; ++g)
        for (u32 m = 0; m < terms; ++m) {
            u64 msg = 1;
            for (u32 i = 0; i < bits; ++i) {
                const u64 s = bits * g + i < P->lwe_dim ? lwe_sk[bits * g + i] : 0;
                msg &= (m >> i & 1) ? s : 1 - s;
            }
            for (u32 lv = 0; lv < P->pbs_level; ++lv)
                for (u32 rr = 0; rr <= k; ++rr, row += row_len) {
                    orc_glwe_encrypt(c, glwe_sk, NULL, r, row);
                    row[(size_t)rr * N] += msg << (64 - P->pbs_base_log * (lv + 1));
                }
        }
}

void orc_gen_bsk_grouped(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r,
                         uint64_t* bsk) {
    orc_gen_bsk_grouped_n(c, lwe_sk, glwe_sk, r, 2, bsk);
}

orc_prepared_bsk* orc_bsk_grouped_prepare_n(const orc_ctx* c, const uint64_t* bsk, uint32_t bits) {
    const orc_params* P = &c->prm;
    const size_t polys = (size_t)P->pbs_level * (P->glwe_dim + 1) * (P->glwe_dim + 1);
    const size_t count = orc_bsk_grouped_count_n(P, bits);
    orc_prepared_bsk* b = (orc_prepared_bsk*)malloc(sizeof *b);
    b->per_ggsw = polys * P->poly_size;
    b->grouping = bits;
    b->raw = 1;
    b->hat = (u64*)malloc(sizeof(u64) * b->per_ggsw * count);
    memcpy(b->hat, bsk, sizeof(u64) * b->per_ggsw * count);
    return b;
}

orc_prepared_bsk* orc_bsk_grouped_prepare(const orc_ctx* c, const uint64_t* bsk) {
    return orc_bsk_grouped_prepare_n(c, bsk, 2);
}

/* One step with `bits` mask elements over 2^64:  acc <- B [.] acc  with
 *     B = sum over m of X^(sum of the exponents a_i whose bit i is set in m) G_m
 * (for pairs G0 + X^a G1 + X^a' G2 + X^(a+a') G3), an encryption of X^(sum a_i s_i).  B is formed here in
 * the coefficient domain with wrapping words (only its value modulo 2^64 enters the result), then the
 * ordinary external product runs over the two 62-bit primes, whose product leaves no doubt about the lift. */
void orc_group_step_n(const orc_ctx* c, const uint64_t* ggsws, uint32_t bits, const uint32_t* a, uint64_t* acc) {
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k1 = P->glwe_dim + 1, terms = 1u << bits;
    const size_t polys = (size_t)P->pbs_level * k1 * k1, per = polys * N;
    u64* B = (u64*)calloc(per + N + (size_t)k1 * N, sizeof(u64));
    u64 *rot = B + per, *out = rot + N;
    for (u32 m = 0; m < terms; ++m) {
        u32 e = 0;
        for (u32 i = 0; i < bits; ++i)
            if (m >> i & 1) e = (e + a[i]) % (2 * N);
        for (size_t q = 0; q < polys; ++q) {
            orc_monomial_mul(rot, ggsws + (size_t)m * per + q * N, e, N);
            for (u32 j = 0; j < N; ++j) B[q * N + j] += rot[j];
        }
    }
    orc_external_product(c, B, acc, out);
    memcpy(acc, out, sizeof(u64) * (size_t)k1 * N);
    free(B);
}

void orc_group_step(const orc_ctx* c, const uint64_t* ggsw4, uint32_t a0, uint32_t a1, uint64_t* acc) {
    const u32 a[2] = {a0, a1};
    orc_group_step_n(c, ggsw4, 2, a, acc);
}

static void blind_rotate_grouped(const orc_ctx* c, const orc_prepared_bsk* pb, const u64* lwe_small,
                                 const u64* lut, u64* acc) {
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k1 = P->glwe_dim + 1, n = P->lwe_dim;
    const u32 bits = pb->grouping, log2n = c->logN + 1, groups = (n + bits - 1) / bits;
    const u32 bt = orc_modswitch(lwe_small[n], log2n);
    for (u32 q = 0; q < k1; ++q)
        orc_monomial_mul(acc + (size_t)q * N, lut + (size_t)q * N, (2 * N - bt) % (2 * N), N);
    for (u32 g = 0; g < groups; ++g) {
        u32 a[3] = {0, 0, 0};
        for (u32 i = 0; i < bits; ++i)
            if (bits * g + i < n) a[i] = orc_modswitch(lwe_small[bits * g + i], log2n);
        /* every step runs, also the one whose exponents are all zero */
        orc_group_step_n(c, pb->hat + ((size_t)g << bits) * pb->per_ggsw, bits, a, acc);
    }
}

static void blind_rotate_hat(const orc_ctx* c, const orc_prepared_bsk* pb, const u64* lwe_small,
                             const u64* lut, u64* acc) {
    if (pb->raw) { blind_rotate_grouped(c, pb, lwe_small, lut, acc); return; }
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k1 = P->glwe_dim + 1, n = P->lwe_dim;
    const u32 log2n = c->logN + 1;
    const u32 bt = orc_modswitch(lwe_small[n], log2n);
    /* acc = X^{-b~} * lut */
    for (u32 q = 0; q < k1; ++q)
        orc_monomial_mul(acc + (size_t)q * N, lut + (size_t)q * N, (2 * N - bt) % (2 * N), N);
    for (u32 i = 0; i < n; ++i) {
        const u32 at = orc_modswitch(lwe_small[i], log2n);
        if (at == 0) continue; /* X^0 - 1 = 0: the external product of zero is zero */
        cmux_rotate_hat(c, pb->hat + (size_t)i * pb->per_ggsw, at, acc);
    }
}

void orc_blind_rotate(const orc_ctx* c, const uint64_t* bsk, const uint64_t* lwe_small,
                      const uint64_t* lut, uint64_t* acc_out) {
    orc_prepared_bsk* pb = orc_bsk_prepare(c, bsk);
    blind_rotate_hat(c, pb, lwe_small, lut, acc_out);
    orc_bsk_prepared_free(pb);
}

void orc_blind_rotate_prepared(const orc_ctx* c, const orc_prepared_bsk* pb, const uint64_t* lwe_small,
                               const uint64_t* lut, uint64_t* acc_out) {
    blind_rotate_hat(c, pb, lwe_small, lut, acc_out);
}

void orc_pbs_prepared(const orc_ctx* c, const uint64_t* ksk, const orc_prepared_bsk* pb,
                      const uint64_t* lwe_big, const uint64_t* lut, uint64_t* out_big) {
    const orc_params* P = &c->prm;
    u64* small = (u64*)malloc(sizeof(u64) * (P->lwe_dim + 1));
    u64* acc = (u64*)malloc(sizeof(u64) * (size_t)(P->glwe_dim + 1) * P->poly_size);
    orc_keyswitch(c, ksk, lwe_big, small);
    if (c->ms_zeros) orc_ms_noise_reduce(c, &c->ms, c->ms_zeros, small);
    blind_rotate_hat(c, pb, small, lut, acc);
    orc_sample_extract(c, acc, out_big);
    free(small); free(acc);
}

void orc_pbs(const orc_ctx* c, const uint64_t* ksk, const uint64_t* bsk, const uint64_t* lwe_big,
             const uint64_t* lut, uint64_t* out_big) {
    orc_prepared_bsk* pb = orc_bsk_prepare(c, bsk);
    orc_pbs_prepared(c, ksk, pb, lwe_big, lut, out_big);
    orc_bsk_prepared_free(pb);
}

void orc_pbs_batch_prepared(const orc_ctx* c, const uint64_t* ksk, const orc_prepared_bsk* pb,
                            const uint64_t* in, const uint64_t* luts, const uint32_t* lut_index,
                            uint64_t* out, size_t count, int nthreads) {
    const orc_params* P = &c->prm;
    const size_t big1 = (size_t)P->glwe_dim * P->poly_size + 1;
    const size_t lut_len = (size_t)(P->glwe_dim + 1) * P->poly_size;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (long i = 0; i < (long)count; ++i)
        orc_pbs_prepared(c, ksk, pb, in + (size_t)i * big1,
                         luts + (size_t)(lut_index ? lut_index[i] : 0) * lut_len, out + (size_t)i * big1);
}

void orc_pbs_batch(const orc_ctx* c, const uint64_t* ksk, const uint64_t* bsk, const uint64_t* in,
                   const uint64_t* luts, const uint32_t* lut_index, uint64_t* out, size_t count,
                   int nthreads) {
    orc_prepared_bsk* pb = orc_bsk_prepare(c, bsk);
    orc_pbs_batch_prepared(c, ksk, pb, in, luts, lut_index, out, count, nthreads);
    orc_bsk_prepared_free(pb);
}

/* ------------------------------------------------------------------------------------------- */
/* bootstrap over the prime ring modulus q'                                                    */
/* ------------------------------------------------------------------------------------------- */
uint64_t orc_qp(void) { return ORC_QP; }

uint64_t orc_qp_gadget(const orc_ctx* c) {
    const u32 b = c->prm.pbs_base_log;
    return (ORC_QP + ((u64)1 << (b - 1))) >> b;
}

uint32_t orc_qp_noise_bound(const orc_ctx* c) {
    /* round(2^noise_log2 * q' / 2^64) */
    const u128 t = ((u128)1 << c->prm.glwe_noise_log2) * ORC_QP + ((u128)1 << 63);
    return (u32)(t >> 64);
}

uint64_t orc_to_qp(uint64_t torus) {
    const u64 r = (u64)(((u128)torus * ORC_QP + ((u128)1 << 63)) >> 64);
    return r == ORC_QP ? 0 : r;
}

uint64_t orc_from_qp(uint64_t res) {
    return (u64)((((u128)res << 64) + ORC_QP / 2) / ORC_QP);
}

static inline i64 qp_centre(u64 r) { return r > ORC_QP / 2 ? (i64)r - (i64)ORC_QP : (i64)r; }

int64_t orc_qp_digit(const orc_ctx* c, uint64_t res) {
    const double inv_g = 1.0 / (double)orc_qp_gadget(c);
    return (i64)rint((double)qp_centre(res) * inv_g);
}

static inline u64 qp_add(u64 a, u64 b) { u64 s = a + b; return s >= ORC_QP ? s - ORC_QP : s; }
static inline u64 qp_sub(u64 a, u64 b) { return a >= b ? a - b : a + ORC_QP - b; }

/* out = a (*) s over Z_q' for a binary polynomial s, through the transform */
static void qp_mul_binary(const orc_ctx* c, const u64* a, const u64* s, u64* out) {
    const u32 N = c->prm.poly_size;
    u64* t = (u64*)malloc(sizeof(u64) * 2 * N);
    u64* u = t + N;
    memcpy(t, a, sizeof(u64) * N);
    memcpy(u, s, sizeof(u64) * N);
    ntt_fwd(&c->qp, t, N); ntt_fwd(&c->qp, u, N);
    for (u32 j = 0; j < N; ++j) t[j] = mont_mul(&c->qp.m, t[j], to_mont(&c->qp.m, u[j]));
    ntt_inv(&c->qp, t, N);
    memcpy(out, t, sizeof(u64) * N);
    free(t);
}

/* one GGSW over q' of the bit `msg`: rows (Z + msg G), written at `row` ((k+1)^2 polynomials) */
static void gen_ggsw_qp(const orc_ctx* c, u64 msg, const u64* glwe_sk, orc_rng* r, u64* row, u64* tmp) {
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k = P->glwe_dim;
    const u64 G = orc_qp_gadget(c);
    const u64 E = orc_qp_noise_bound(c);
    for (u32 rr = 0; rr <= k; ++rr, row += (size_t)(k + 1) * N) {
        u64* body = row + (size_t)k * N;
        for (u32 j = 0; j < N; ++j) {
            const u64 e = orc_rng_tuniform_bound(r, E);          /* [-E, E], end points at half weight */
            body[j] = e >= E ? e - E : e + ORC_QP - E;
        }
        for (u32 q = 0; q < k; ++q) {
            u64* a = row + (size_t)q * N;
            for (u32 j = 0; j < N; ++j) a[j] = (u64)(((u128)orc_rng_u64(r) * ORC_QP) >> 64);
            qp_mul_binary(c, a, glwe_sk + (size_t)q * N, tmp);
            for (u32 j = 0; j < N; ++j) body[j] = qp_add(body[j], tmp[j]);
        }
        if (msg) row[(size_t)rr * N] = qp_add(row[(size_t)rr * N], G % ORC_QP);
    }
}

void orc_gen_bsk_qp(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r, uint64_t* bsk) {
    const orc_params* P = &c->prm;
    const size_t per = (size_t)(P->glwe_dim + 1) * (P->glwe_dim + 1) * P->poly_size;
    u64* tmp = (u64*)malloc(sizeof(u64) * P->poly_size);
    for (u32 i = 0; i < P->lwe_dim; ++i) gen_ggsw_qp(c, lwe_sk[i], glwe_sk, r, bsk + (size_t)i * per, tmp);
    free(tmp);
}

/* Grouped key: the LWE key bits are taken `bits` at a time (2 or 3; bits past the end of the key count
 * as 0); group g holds 2^bits GGSWs, GGSW m encrypting the product over the group's bits i of
 * s_i (bit i of m set) or 1 - s_i (clear): exactly one of them encrypts 1.  For pairs that is
 * (1-s)(1-s'), s(1-s'), (1-s)s', s s'. */
size_t orc_bsk_qp_grouped_count_n(const orc_params* p, uint32_t bits) {
    return ((size_t)1 << bits) * (((size_t)p->lwe_dim + bits - 1) / bits);
}
size_t orc_bsk_qp_grouped_count(const orc_params* p) { return orc_bsk_qp_grouped_count_n(p, 2); }

void orc_gen_bsk_qp_grouped_n(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r,
                              uint32_t bits, uint64_t* bsk) {
    const orc_params* P = &c->prm;
    const size_t per = (size_t)(P->glwe_dim + 1) * (P->glwe_dim + 1) * P->poly_size;
    const u32 groups = (P->lwe_dim + bits - 1) / bits, terms = 1u << bits;
    u64* tmp = (u64*)malloc(sizeof(u64) * P->poly_size);
    for (u32 g = 0; g < groups; ++g)
        for (u32 m = 0; m < terms; ++m) {
            u64 msg = 1;
            for (u32 i = 0; i < bits; ++i) {
                const u64 s = bits * g + i < P->lwe_dim ? lwe_sk[bits * g + i] : 0;
                msg &= (m >> i & 1) ? s : 1 - s;
            }
            gen_ggsw_qp(c, msg, glwe_sk, r, bsk + ((size_t)terms * g + m) * per, tmp);
        }
    free(tmp);
}

void orc_gen_bsk_qp_grouped(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r,
                            uint64_t* bsk) {
    orc_gen_bsk_qp_grouped_n(c, lwe_sk, glwe_sk, r, 2, bsk);
}

/* NTT-domain image of one GGSW over q': hat[row][poly][N], Montgomery form */
static void ggsw_qp_to_hat(const orc_ctx* c, const u64* ggsw, u64* hat) {
    const u32 N = c->prm.poly_size, k1 = c->prm.glwe_dim + 1;
    for (size_t q = 0; q < (size_t)k1 * k1; ++q) {
        u64* h = hat + q * N;
        memcpy(h, ggsw + q * N, sizeof(u64) * N);
        ntt_fwd(&c->qp, h, N);
        for (u32 j = 0; j < N; ++j) h[j] = to_mont(&c->qp.m, h[j]);
    }
}

static void external_product_qp_hat(const orc_ctx* c, const u64* hat, const u64* glwe, u64* out) {
    const u32 N = c->prm.poly_size, k1 = c->prm.glwe_dim + 1;
    u64* acc = (u64*)calloc((size_t)k1 * N, sizeof(u64));
    u64* d = (u64*)malloc(sizeof(u64) * N);
    for (u32 rr = 0; rr < k1; ++rr) {
        for (u32 j = 0; j < N; ++j) {
            const i64 dg = orc_qp_digit(c, glwe[(size_t)rr * N + j]);
            d[j] = dg < 0 ? (u64)(dg + (i64)ORC_QP) : (u64)dg;
        }
        ntt_fwd(&c->qp, d, N);
        for (u32 cc = 0; cc < k1; ++cc) {
            const u64* h = hat + ((size_t)rr * k1 + cc) * N;
            u64* a = acc + (size_t)cc * N;
            for (u32 j = 0; j < N; ++j) a[j] = qp_add(a[j], mont_mul(&c->qp.m, d[j], h[j]));
        }
    }
    for (u32 cc = 0; cc < k1; ++cc) {
        ntt_inv(&c->qp, acc + (size_t)cc * N, N);
        memcpy(out + (size_t)cc * N, acc + (size_t)cc * N, sizeof(u64) * N);
    }
    free(acc); free(d);
}

void orc_external_product_qp(const orc_ctx* c, const uint64_t* ggsw, const uint64_t* glwe, uint64_t* out) {
    const u32 N = c->prm.poly_size, k1 = c->prm.glwe_dim + 1;
    u64* hat = (u64*)malloc(sizeof(u64) * (size_t)k1 * k1 * N);
    ggsw_qp_to_hat(c, ggsw, hat);
    external_product_qp_hat(c, hat, glwe, out);
    free(hat);
}

static orc_prepared_bsk* bsk_qp_prepare_n(const orc_ctx* c, const u64* bsk, size_t count, u32 grouping) {
    const orc_params* P = &c->prm;
    const size_t per = (size_t)(P->glwe_dim + 1) * (P->glwe_dim + 1) * P->poly_size;
    orc_prepared_bsk* b = (orc_prepared_bsk*)malloc(sizeof *b);
    b->per_ggsw = per;
    b->grouping = grouping;
    b->raw = 0;
    b->hat = (u64*)malloc(sizeof(u64) * per * count);
    for (size_t i = 0; i < count; ++i) ggsw_qp_to_hat(c, bsk + i * per, b->hat + i * per);
    return b;
}

orc_prepared_bsk* orc_bsk_qp_prepare(const orc_ctx* c, const uint64_t* bsk) {
    return bsk_qp_prepare_n(c, bsk, c->prm.lwe_dim, 1);
}

orc_prepared_bsk* orc_bsk_qp_grouped_prepare(const orc_ctx* c, const uint64_t* bsk) {
    return bsk_qp_prepare_n(c, bsk, orc_bsk_qp_grouped_count(&c->prm), 2);
}

orc_prepared_bsk* orc_bsk_qp_grouped_prepare_n(const orc_ctx* c, const uint64_t* bsk, uint32_t bits) {
    return bsk_qp_prepare_n(c, bsk, orc_bsk_qp_grouped_count_n(&c->prm, bits), bits);
}

/* X^rot * in over Z_q' */
static void qp_monomial_mul(u64* out, const u64* in, u32 rot, u32 N) {
    for (u32 j = 0; j < N; ++j) {
        u32 k = j + rot;
        if (k >= 2 * N) k -= 2 * N;
        if (k < N) out[k] = in[j]; else out[k - N] = in[j] ? ORC_QP - in[j] : 0;
    }
}

/* Two mask elements per step:  acc <- B [.] acc  with the GGSW
 *     B = GGSW((1-s)(1-s')) + X^a GGSW(s(1-s')) + X^a' GGSW((1-s)s') + X^(a+a') GGSW(s s'),
 * an encryption of X^(a s + a' s') (exactly one of the four bits is 1).  B is formed in the
 * transform domain, where a monomial is a point-wise factor, so a step costs one external product.
 * The accumulator is replaced, not added to: what the gadget decomposition rounds away
 * (|.| <= G/2 per coefficient and step) stays in the error instead of being carried. */
/* one step: acc <- B [.] acc for the 2^bits NTT-domain GGSWs at hat and the exponents a[0 .. bits-1]: term m
 * carries X^(sum of the a_i whose bit is set in m) */
static void group_step_hat_n(const orc_ctx* c, const u64* hat, size_t per, u32 bits, const u32* ai, u64* acc) {
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k1 = P->glwe_dim + 1, terms = 1u << bits;
    const mont_t* M = &c->qp.m;
    u64* tmp = (u64*)malloc(sizeof(u64) * ((size_t)k1 * N * 2 + (size_t)terms * N));
    u64 *dig = tmp, *out = tmp + (size_t)k1 * N, *mono = tmp + (size_t)2 * k1 * N;
    /* transform of X^a, Montgomery form */
    for (u32 m = 0; m < terms; ++m) {
        u32 a = 0;
        for (u32 i = 0; i < bits; ++i)
            if (m >> i & 1) a = (a + ai[i]) % (2 * N);
        u64* mm = mono + (size_t)m * N;
        memset(mm, 0, sizeof(u64) * N);
        if (a < N) mm[a] = 1; else mm[a - N] = ORC_QP - 1;
        ntt_fwd(&c->qp, mm, N);
        for (u32 j = 0; j < N; ++j) mm[j] = to_mont(M, mm[j]);
    }
    for (u32 rr = 0; rr < k1; ++rr) {
        u64* d = dig + (size_t)rr * N;
        for (u32 j = 0; j < N; ++j) {
            const i64 dg = orc_qp_digit(c, acc[(size_t)rr * N + j]);
            d[j] = dg < 0 ? (u64)(dg + (i64)ORC_QP) : (u64)dg;
        }
        ntt_fwd(&c->qp, d, N);
    }
    for (u32 cc = 0; cc < k1; ++cc) {
        u64* o = out + (size_t)cc * N;
        for (u32 j = 0; j < N; ++j) {
            u64 sum = 0;
            for (u32 rr = 0; rr < k1; ++rr) {
                u64 b = 0;
                for (u32 m = 0; m < terms; ++m) {
                    /* both factors are Montgomery forms: two reductions give the plain product */
                    const u64 h = hat[(size_t)m * per + ((size_t)rr * k1 + cc) * N + j];
                    b = qp_add(b, mont_mul(M, mont_mul(M, mono[(size_t)m * N + j], h), 1));
                }
                sum = qp_add(sum, mont_mul(M, dig[(size_t)rr * N + j], to_mont(M, b)));
            }
            o[j] = sum;
        }
        ntt_inv(&c->qp, o, N);
    }
    memcpy(acc, out, sizeof(u64) * (size_t)k1 * N);
    free(tmp);
}

static void group_step_hat(const orc_ctx* c, const u64* hat4, size_t per, u32 a0, u32 a1, u64* acc) {
    const u32 a[2] = {a0, a1};
    group_step_hat_n(c, hat4, per, 2, a, acc);
}

void orc_group_step_qp(const orc_ctx* c, const uint64_t* ggsw4, uint32_t a0, uint32_t a1, uint64_t* acc) {
    const u32 N = c->prm.poly_size, k1 = c->prm.glwe_dim + 1;
    const size_t per = (size_t)k1 * k1 * N;
    u64* hat = (u64*)malloc(sizeof(u64) * 4 * per);
    for (u32 m = 0; m < 4; ++m) ggsw_qp_to_hat(c, ggsw4 + (size_t)m * per, hat + (size_t)m * per);
    group_step_hat(c, hat, per, a0, a1, acc);
    free(hat);
}

static void blind_rotate_qp_grouped(const orc_ctx* c, const orc_prepared_bsk* pb, const u64* lwe_small,
                                    const u64* lut, u64* acc) {
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k1 = P->glwe_dim + 1, n = P->lwe_dim;
    const u32 bits = pb->grouping, log2n = c->logN + 1, groups = (n + bits - 1) / bits;
    u64* tmp = (u64*)malloc(sizeof(u64) * (size_t)k1 * N);
    for (size_t j = 0; j < (size_t)k1 * N; ++j) tmp[j] = orc_to_qp(lut[j]);
    const u32 bt = orc_modswitch(lwe_small[n], log2n);
    for (u32 q = 0; q < k1; ++q)
        qp_monomial_mul(acc + (size_t)q * N, tmp + (size_t)q * N, (2 * N - bt) % (2 * N), N);
    free(tmp);
    for (u32 g = 0; g < groups; ++g) {
        /* every step runs, also one whose exponents are all zero (B then encrypts 1): the sequence of
         * operations does not depend on the ciphertext */
        u32 a[3] = {0, 0, 0};
        for (u32 i = 0; i < bits; ++i)
            if (bits * g + i < n) a[i] = orc_modswitch(lwe_small[bits * g + i], log2n);
        group_step_hat_n(c, pb->hat + ((size_t)g << bits) * pb->per_ggsw, pb->per_ggsw, bits, a, acc);
    }
}

void orc_blind_rotate_qp(const orc_ctx* c, const orc_prepared_bsk* pb, const uint64_t* lwe_small,
                         const uint64_t* lut, uint64_t* acc) {
    if (pb->grouping >= 2) { blind_rotate_qp_grouped(c, pb, lwe_small, lut, acc); return; }
    const orc_params* P = &c->prm;
    const u32 N = P->poly_size, k1 = P->glwe_dim + 1, n = P->lwe_dim;
    const u32 log2n = c->logN + 1;
    u64* tmp = (u64*)malloc(sizeof(u64) * (size_t)k1 * N * 3);
    u64 *diff = tmp + (size_t)k1 * N, *prod = tmp + (size_t)2 * k1 * N;
    for (size_t j = 0; j < (size_t)k1 * N; ++j) tmp[j] = orc_to_qp(lut[j]);
    const u32 bt = orc_modswitch(lwe_small[n], log2n);
    for (u32 q = 0; q < k1; ++q)
        qp_monomial_mul(acc + (size_t)q * N, tmp + (size_t)q * N, (2 * N - bt) % (2 * N), N);
    for (u32 i = 0; i < n; ++i) {
        const u32 at = orc_modswitch(lwe_small[i], log2n);
        if (at == 0) continue;
        for (u32 q = 0; q < k1; ++q) {
            qp_monomial_mul(diff + (size_t)q * N, acc + (size_t)q * N, at, N);
            for (u32 j = 0; j < N; ++j)
                diff[(size_t)q * N + j] = qp_sub(diff[(size_t)q * N + j], acc[(size_t)q * N + j]);
        }
        external_product_qp_hat(c, pb->hat + (size_t)i * pb->per_ggsw, diff, prod);
        for (size_t j = 0; j < (size_t)k1 * N; ++j) acc[j] = qp_add(acc[j], prod[j]);
    }
    free(tmp);
}

void orc_pbs_qp(const orc_ctx* c, const uint64_t* ksk, const orc_prepared_bsk* pb, const uint64_t* lwe_big,
                const uint64_t* lut, uint64_t* out_big) {
    const orc_params* P = &c->prm;
    const size_t big = (size_t)P->glwe_dim * P->poly_size;
    u64* small = (u64*)malloc(sizeof(u64) * (P->lwe_dim + 1));
    u64* acc = (u64*)malloc(sizeof(u64) * (size_t)(P->glwe_dim + 1) * P->poly_size);
    orc_keyswitch(c, ksk, lwe_big, small);
    if (c->ms_zeros) orc_ms_noise_reduce(c, &c->ms, c->ms_zeros, small);
    orc_blind_rotate_qp(c, pb, small, lut, acc);
    /* sample extraction over q' (negation is modular), then every word back to 2^64 */
    for (u32 i = 0; i < P->glwe_dim; ++i) {
        const u64* a = acc + (size_t)i * P->poly_size;
        u64* o = out_big + (size_t)i * P->poly_size;
        o[0] = orc_from_qp(a[0]);
        for (u32 j = 1; j < P->poly_size; ++j) o[j] = orc_from_qp(a[P->poly_size - j] ? ORC_QP - a[P->poly_size - j] : 0);
    }
    out_big[big] = orc_from_qp(acc[big]);
    free(small); free(acc);
}

void orc_pbs_batch_qp(const orc_ctx* c, const uint64_t* ksk, const orc_prepared_bsk* pb, const uint64_t* in,
                      const uint64_t* luts, const uint32_t* lut_index, uint64_t* out, size_t count, int nthreads) {
    const orc_params* P = &c->prm;
    const size_t big1 = (size_t)P->glwe_dim * P->poly_size + 1;
    const size_t lut_len = (size_t)(P->glwe_dim + 1) * P->poly_size;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (long i = 0; i < (long)count; ++i)
        orc_pbs_qp(c, ksk, pb, in + (size_t)i * big1, luts + (size_t)(lut_index ? lut_index[i] : 0) * lut_len,
                   out + (size_t)i * big1);
}

/* ------------------------------------------------------------------------------------------- */
/* shortint encoding                                                                           */
/* ------------------------------------------------------------------------------------------- */
static u64 delta_of(const orc_ctx* c) {
    /* one padding bit above message and carry (the pattern of encryption_delta,
     * ml/extensions/rust/src/encryption.rs:5-22) */
    return ((u64)1 << 63) / ((u64)c->prm.message_modulus * c->prm.carry_modulus);
}

uint64_t orc_encode(const orc_ctx* c, uint64_t m) { return m * delta_of(c); }

uint64_t orc_decode(const orc_ctx* c, uint64_t phase) {
    const u64 delta = delta_of(c);
    const u64 space = (u64)c->prm.message_modulus * c->prm.carry_modulus;
    u64 r = (phase + delta / 2) / delta; /* includes the padding bit */
    return r % space;
}

void orc_make_lut(const orc_ctx* c, const uint64_t* table, uint64_t* lut) {
    const u32 N = c->prm.poly_size, k = c->prm.glwe_dim;
    const u32 space = c->prm.message_modulus * c->prm.carry_modulus;
    const u32 box = N / space, half = box / 2;
    const u64 delta = delta_of(c);
    u64* body = lut + (size_t)k * N;
    u64* tmp = (u64*)malloc(sizeof(u64) * N);
    memset(lut, 0, sizeof(u64) * (size_t)k * N);
    for (u32 i = 0; i < space; ++i)
        for (u32 j = 0; j < box; ++j) tmp[i * box + j] = table[i] * delta;
    for (u32 j = 0; j < half; ++j) tmp[j] = (u64)0 - tmp[j];
    for (u32 j = 0; j < N; ++j) body[j] = tmp[(j + half) % N]; /* rotate left by half a box */
    free(tmp);
}
