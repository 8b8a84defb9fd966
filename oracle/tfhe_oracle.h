/*
 * tfhe_oracle.h -- CPU restatement of the TFHE programmable-bootstrap path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under the product package may include, link or call this
 * file: it exists so that tests/, the smoke check and the benchmark's cpu_baseline leg have an
 * independent, plain-C, integer-only statement of the algorithms to compare the HIP kernels with.
 *
 * What it restates.  The reference tree (luxfi/fhe) holds no FHE arithmetic of its own: the engine
 * is an absent dependency (Go github.com/luxfi/tfhe on luxfi/lattice, unpinned; or tfhe-rs, pinned
 * as npm node-tfhe 1.4.0-alpha.3 in sdk/relayer/package.json and as cargo git rev ec78318a in
 * ml/extensions/rust/Cargo.toml:40).  This file therefore follows the published TFHE algorithms as
 * tfhe-rs implements them, at the parameter set the reference names
 * (sdk/relayer/src/tfhe.ts:14-19, PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128) and whose values
 * are decoded from sdk/relayer/src/test/keys/privateKey.bin, and it anchors on the reference's own
 * call sites:
 *   - exact wrapping negacyclic product: ml/extensions/rust/src/computations.rs:29-56
 *     (polynomial_wrapping_mul / polynomial_wrapping_add_mul_assign);
 *   - sample extraction: ml/extensions/rust/src/computations.rs:125-129;
 *   - delta encoding: ml/extensions/rust/src/encryption.rs:5-40;
 *   - the KS -> PBS atomic pattern and accumulator generation:
 *     ml/biometrics/notebooks/main.rs:68-71 (generate_accumulator, keyswitch_programmable_bootstrap).
 *
 * Parity pins.  (1) The negacyclic product is pinned by the reference's key fixtures: with s the
 * compact-public-key secret (privateKey.bin @24133) and (a,b) = publicKey.bin @98,
 * b - a (*) rev(s) has |e| <= 2^17 (tests/test_oracle_golden.py).  (2) Decrypted operator results
 * are pinned by the reference's known-answer tests
 * (tests/fhevm-suite/e2e/test/fhevmOperations/ (all .ts files), extracted to tests/golden/).  (3) Ciphertext
 * bits (NTT-domain values, key-switched and bootstrapped ciphertexts) are PARITY UNPINNED: no
 * engine, and no stored ciphertext, exists in the reference tree to compare with.
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t lwe_dim;         /* n   (918)  */
    uint32_t glwe_dim;        /* k   (1)    */
    uint32_t poly_size;       /* N   (2048) */
    uint32_t pbs_base_log;    /* 23 */
    uint32_t pbs_level;       /* 1  */
    uint32_t ks_base_log;     /* 4  */
    uint32_t ks_level;        /* 4  */
    uint32_t message_modulus; /* 4  */
    uint32_t carry_modulus;   /* 4  */
    uint32_t lwe_noise_log2;  /* TUniform bound for the small (LWE) key: 45 */
    uint32_t glwe_noise_log2; /* TUniform bound for the GLWE / big key: 17 */
} orc_params;

/* ---- deterministic generator: ChaCha20 keystream (RFC 8439 block function), 64-bit counter ---- */
typedef struct {
    uint32_t key[8];
    uint64_t counter;
    uint64_t stream;
    uint32_t buf[16];
    uint32_t pos;
} orc_rng;

void orc_rng_init(orc_rng* r, const uint8_t seed[32], uint64_t stream);
uint64_t orc_rng_u64(orc_rng* r);
/* uniform integer in [-2^log2_bound, 2^log2_bound], returned as a torus element (mod 2^64) */
uint64_t orc_rng_tuniform(orc_rng* r, uint32_t log2_bound);
uint64_t orc_rng_tuniform_bound(orc_rng* r, uint64_t bound);   /* offset in [0, 2 bound] */

/* ---- context: parameters + NTT tables for the two oracle primes ---- */
typedef struct orc_ctx orc_ctx;
orc_ctx* orc_ctx_new(const orc_params* p);
void orc_ctx_free(orc_ctx* c);
void orc_default_params(orc_params* p);

/* ---- polynomial arithmetic in Z_{2^64}[X]/(X^N+1) ---- */
/* out = small (*) b, schoolbook, wrapping.  `small` is a signed polynomial. */
void orc_negacyclic_mul_schoolbook(uint64_t* out, const int64_t* small, const uint64_t* b, uint32_t N);
/* same product through the two-prime NTT + CRT (requires |small| < 2^23 so the integer result
 * stays below the CRT modulus) */
void orc_negacyclic_mul_ntt(const orc_ctx* c, uint64_t* out, const int64_t* small, const uint64_t* b);
/* out = X^rot * in for rot in [0, 2N) */
void orc_monomial_mul(uint64_t* out, const uint64_t* in, uint32_t rot, uint32_t N);

/* direct O(N^2) definition of the negacyclic transform over an arbitrary prime:
 * out[j] = sum_i in[i] * psi^((2j+1) i) mod p, natural order.  For checking device transforms. */
void orc_ntt_definition(uint64_t p, uint64_t psi, const uint64_t* in, uint64_t* out, uint32_t N);
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t p);
uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t p);

/* ---- gadget decomposition (signed, closest representative) ---- */
/* digits[level j = 0..levels-1] such that sum_j digits[j] * 2^(64 - base_log*(j+1)) is the closest
 * representable value to x; every digit lies in [-B/2, B/2]. */
void orc_decompose(uint64_t x, uint32_t base_log, uint32_t levels, int64_t* digits);

/* ---- keys and encryption ---- */
void orc_gen_binary_key(orc_rng* r, uint64_t* key, size_t len);
void orc_lwe_encrypt(const uint64_t* sk, size_t dim, uint64_t plaintext, uint32_t noise_log2,
                     orc_rng* r, uint64_t* ct);
uint64_t orc_lwe_phase(const uint64_t* sk, size_t dim, const uint64_t* ct);
void orc_lwe_trivial(size_t dim, uint64_t plaintext, uint64_t* ct);
/* GLWE encryption of the plaintext polynomial pt (NULL = zero) under glwe_sk (k polynomials) */
void orc_glwe_encrypt(const orc_ctx* c, const uint64_t* glwe_sk, const uint64_t* pt, orc_rng* r,
                      uint64_t* ct);
void orc_glwe_phase(const orc_ctx* c, const uint64_t* glwe_sk, const uint64_t* ct, uint64_t* phase);

/* bootstrapping key: bsk[i][level][row][poly][N], i < n; GGSW(s_i) as Z + s_i * G */
size_t orc_bsk_len(const orc_params* p);
void orc_gen_bsk(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r,
                 uint64_t* bsk);
/* key-switching key big (kN) -> small (n): ksk[i][level][n+1] */
size_t orc_ksk_len(const orc_params* p);
void orc_gen_ksk(const orc_ctx* c, const uint64_t* big_sk, const uint64_t* lwe_sk, orc_rng* r,
                 uint64_t* ksk);

/* ---- the bootstrap path ---- */
/* out = ggsw [.] glwe  (external product, exact arithmetic).  ggsw: [level][row][poly][N] */
void orc_external_product(const orc_ctx* c, const uint64_t* ggsw, const uint64_t* glwe, uint64_t* out);
/* acc <- acc + ggsw [.] (X^rot * acc - acc) */
void orc_cmux_rotate(const orc_ctx* c, const uint64_t* ggsw, uint32_t rot, uint64_t* acc);
/* round a torus element to Z_{2N} */
uint32_t orc_modswitch(uint64_t x, uint32_t log2_2N);
/* ---- modulus-switch noise reduction ----
 * tfhe-rs improve_lwe_ciphertext_modulus_switch_noise_for_binary_key (absent dependency; the
 * parameters are the modulus_switch_noise_reduction fields of the reference fixture
 * sdk/relayer/src/test/keys/privateKey.bin, offsets 24068-24110: 1449 zeros, bound 2^58,
 * r_sigma 13.179852282053789, input variance 2.63039184094559e-07).  Rounding a small ciphertext to
 * Z_{2N} adds e_b - sum e_i s_i; over a uniform binary key that has mean e_b - sum e_i / 2 and
 * variance sum e_i^2 / 4.  The candidates ct, ct + zero_0, ct + zero_1, ... are tried in order; the
 * first whose estimate |mean| + r_sigma sqrt(input_variance + variance) is within the bound is
 * taken, else the one with the smallest estimate.  The estimate is computed in fixed point (units
 * of 2^-38 of the torus; see orc_ms_estimate) so that the choice does not depend on the order of
 * floating-point sums.  PARITY UNPINNED: the reference holds no vector for this step. */
typedef struct {
    uint32_t zeros_count;
    uint32_t bound_log2;
    double r_sigma_factor;
    double input_variance;
} orc_ms_params;
void orc_default_ms_params(orc_ms_params* m);
/* zeros[count][n+1]: encryptions of zero under the small key, noise TUniform(lwe_noise_log2) */
void orc_gen_ms_zeros(const orc_ctx* c, const uint64_t* lwe_sk, orc_rng* r, uint32_t count, uint64_t* zeros);
uint64_t orc_ms_estimate(const orc_ctx* c, const orc_ms_params* m, const uint64_t* lwe_small);
/* in place; returns the index of the encryption of zero that was added, -1 if none */
int32_t orc_ms_noise_reduce(const orc_ctx* c, const orc_ms_params* m, const uint64_t* zeros, uint64_t* lwe_small);
/* make orc_pbs* run the step between key switch and modulus switch (zeros is borrowed; NULL: off) */
void orc_ctx_set_ms(orc_ctx* c, const orc_ms_params* m, const uint64_t* zeros);

/* key switch LWE(kN) -> LWE(n) */
void orc_keyswitch(const orc_ctx* c, const uint64_t* ksk, const uint64_t* in, uint64_t* out);
/* blind rotation of the accumulator `lut` ((k+1) polys) by the small ciphertext; out = GLWE */
void orc_blind_rotate(const orc_ctx* c, const uint64_t* bsk, const uint64_t* lwe_small,
                      const uint64_t* lut, uint64_t* acc_out);
void orc_sample_extract(const orc_ctx* c, const uint64_t* glwe, uint64_t* lwe_big);
/* KS -> MS -> BR -> SE */
void orc_pbs(const orc_ctx* c, const uint64_t* ksk, const uint64_t* bsk, const uint64_t* lwe_big,
             const uint64_t* lut, uint64_t* out_big);
/* count independent bootstraps, lut_index[i] selects one of the LUTs (each (k+1)N words);
 * nthreads <= 0 means one thread */
void orc_pbs_batch(const orc_ctx* c, const uint64_t* ksk, const uint64_t* bsk, const uint64_t* in,
                   const uint64_t* luts, const uint32_t* lut_index, uint64_t* out, size_t count,
                   int nthreads);

/* NTT-domain image of a bootstrapping key, built once and reused across bootstraps */
typedef struct orc_prepared_bsk orc_prepared_bsk;
orc_prepared_bsk* orc_bsk_prepare(const orc_ctx* c, const uint64_t* bsk);
void orc_bsk_prepared_free(orc_prepared_bsk* b);
void orc_blind_rotate_prepared(const orc_ctx* c, const orc_prepared_bsk* pb, const uint64_t* lwe_small,
                               const uint64_t* lut, uint64_t* acc_out);
void orc_pbs_prepared(const orc_ctx* c, const uint64_t* ksk, const orc_prepared_bsk* pb,
                      const uint64_t* lwe_big, const uint64_t* lut, uint64_t* out_big);
void orc_pbs_batch_prepared(const orc_ctx* c, const uint64_t* ksk, const orc_prepared_bsk* pb,
                            const uint64_t* in, const uint64_t* luts, const uint32_t* lut_index,
                            uint64_t* out, size_t count, int nthreads);

/* ---- bootstrap over a prime ring modulus ----
 * Variant of the blind rotation in which the bootstrapping key and the accumulator live in
 * Z_q'[X]/(X^N+1) for the NTT-friendly prime q' = 703687441727489 (5*2^47 - 49151, 1 mod 8192)
 * instead of q = 2^64: one exact transform per polynomial, no residue system, no lift.  The LWE
 * ciphertexts on both sides keep q = 2^64 (the construction of tfhe-rs' NTT bootstrap,
 * programmable_bootstrap_ntt64_lwe_ciphertext: modulus switch in, switch back out).  Relative to
 * its modulus everything keeps the parameter set's proportions:
 *   gadget   G  = round(q' / 2^pbs_base_log)                    (83886080 = 5*2^24 for base 2^23)
 *   digit    d  = rint(c * fl(1/G)) in IEEE double, c the centred residue in (-q'/2, q'/2)
 *   noise    e  in [-E, E], end points at half weight (the set's TUniform shape), E = round(2^glwe_noise_log2 * q' / 2^64)   (5 for 2^17)
 *   LUT      round(v * q' / 2^64) of the 2^64-scale accumulator body
 *   output   round(r * 2^64 / q') of every word of the sample-extracted ciphertext
 * pbs_level must be 1.  PARITY UNPINNED against the reference (no engine in the tree). */
uint64_t orc_qp(void);
uint64_t orc_qp_gadget(const orc_ctx* c);
uint32_t orc_qp_noise_bound(const orc_ctx* c);
uint64_t orc_to_qp(uint64_t torus);   /* Z_{2^64} -> Z_q', rounding */
uint64_t orc_from_qp(uint64_t res);   /* Z_q' -> Z_{2^64}, rounding */
int64_t orc_qp_digit(const orc_ctx* c, uint64_t res);
/* bsk[i][row][poly][N], residues in [0, q'): GGSW(s_i) = Z + s_i G */
void orc_gen_bsk_qp(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r, uint64_t* bsk);
/* out = ggsw [.] glwe over Z_q' (both residues in [0, q')) */
void orc_external_product_qp(const orc_ctx* c, const uint64_t* ggsw, const uint64_t* glwe, uint64_t* out);
orc_prepared_bsk* orc_bsk_qp_prepare(const orc_ctx* c, const uint64_t* bsk);
/* Grouped key over q': the key bits in pairs (s, s') = (s_2g, s_2g+1), s' = 0 past the end of an
 * odd-length key; bsk[g][m][row][poly][N] with GGSW m = 0..3 of (1-s)(1-s'), s(1-s'), (1-s)s', s s'.
 * A blind rotation with a prepared grouped key takes one external product per PAIR of mask
 * elements: acc <- (G0 + X^a G1 + X^a' G2 + X^(a+a') G3) [.] acc, the sum formed in the transform
 * domain (the multi-bit blind rotation of tfhe-rs, grouping factor 2; a pair with a = a' = 0 is
 * skipped).  orc_blind_rotate_qp / orc_pbs_qp / orc_pbs_batch_qp take either kind of prepared key. */
/* Grouped key over 2^64: four GGSWs over the torus per pair of key bits (of (1-s)(1-s'), s(1-s'),
 * (1-s)s', s s'); the prepared key goes to orc_blind_rotate_prepared / orc_pbs_prepared /
 * orc_pbs_batch_prepared like the one-bit key's.  orc_group_step: acc <- (G0 + X^a0 G1 + X^a1 G2 +
 * X^(a0+a1) G3) [.] acc for the four GGSWs at ggsw4. */
size_t orc_bsk_grouped_count(const orc_params* p);
void orc_gen_bsk_grouped(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r,
                         uint64_t* bsk);
orc_prepared_bsk* orc_bsk_grouped_prepare(const orc_ctx* c, const uint64_t* bsk);
void orc_group_step(const orc_ctx* c, const uint64_t* ggsw4, uint32_t a0, uint32_t a1, uint64_t* acc);
/* the same with `bits` (2 or 3) key bits per step: 2^bits GGSWs over the torus per group, ordered as in
 * the grouped key over q' below; orc_group_step_n takes the `bits` exponents at a */
size_t orc_bsk_grouped_count_n(const orc_params* p, uint32_t bits);
void orc_gen_bsk_grouped_n(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r,
                           uint32_t bits, uint64_t* bsk);
orc_prepared_bsk* orc_bsk_grouped_prepare_n(const orc_ctx* c, const uint64_t* bsk, uint32_t bits);
void orc_group_step_n(const orc_ctx* c, const uint64_t* ggsws, uint32_t bits, const uint32_t* a, uint64_t* acc);

size_t orc_bsk_qp_grouped_count(const orc_params* p);   /* GGSWs in the grouped key */
void orc_gen_bsk_qp_grouped(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r,
                            uint64_t* bsk);
orc_prepared_bsk* orc_bsk_qp_grouped_prepare(const orc_ctx* c, const uint64_t* bsk);
/* the same with `bits` (2 or 3) key bits per blind-rotation step: 2^bits GGSWs per group, GGSW m of the
 * product over the group's bits of s_i (bit i of m set) or 1 - s_i; a step multiplies the accumulator
 * by the sum over m of X^(sum of the a_i with bit i of m set) GGSW_m */
size_t orc_bsk_qp_grouped_count_n(const orc_params* p, uint32_t bits);
void orc_gen_bsk_qp_grouped_n(const orc_ctx* c, const uint64_t* lwe_sk, const uint64_t* glwe_sk, orc_rng* r,
                              uint32_t bits, uint64_t* bsk);
orc_prepared_bsk* orc_bsk_qp_grouped_prepare_n(const orc_ctx* c, const uint64_t* bsk, uint32_t bits);
/* one step on its own: ggsw4 = four GGSWs over q' in the coefficient domain, acc (k+1 polys of
 * residues) is replaced by B [.] acc for the exponents a0, a1 in [0, 2N) */
void orc_group_step_qp(const orc_ctx* c, const uint64_t* ggsw4, uint32_t a0, uint32_t a1, uint64_t* acc);
/* lut: the 2^64-scale accumulator ((k+1) polys); acc_out: (k+1) polys of residues in [0, q') */
void orc_blind_rotate_qp(const orc_ctx* c, const orc_prepared_bsk* pb, const uint64_t* lwe_small,
                         const uint64_t* lut, uint64_t* acc_out);
/* KS -> (noise reduction) -> MS -> BR over q' -> SE -> switch to 2^64 */
void orc_pbs_qp(const orc_ctx* c, const uint64_t* ksk, const orc_prepared_bsk* pb, const uint64_t* lwe_big,
                const uint64_t* lut, uint64_t* out_big);
void orc_pbs_batch_qp(const orc_ctx* c, const uint64_t* ksk, const orc_prepared_bsk* pb, const uint64_t* in,
                      const uint64_t* luts, const uint32_t* lut_index, uint64_t* out, size_t count, int nthreads);

/* ---- shortint encoding ---- */
/* accumulator for f given as a table over [0, message_modulus*carry_modulus) */
void orc_make_lut(const orc_ctx* c, const uint64_t* table, uint64_t* lut);
uint64_t orc_encode(const orc_ctx* c, uint64_t m);          /* m * delta */
uint64_t orc_decode(const orc_ctx* c, uint64_t phase);      /* round(phase / delta) mod (msg*carry) */

#ifdef __cplusplus
}
#endif
#endif
