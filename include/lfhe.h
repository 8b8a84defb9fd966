/*
 * lfhe.h -- C ABI of liblfhe, the MI355X (gfx950) TFHE compute backend.
 *
 * This is the boundary a host SDK binds (N-API / FFI / cgo / ctypes).  It replaces the engine the
 * reference reaches over HTTP on :8448 and through its WASM / native bindings; each entry point
 * names the reference interface it stands in for.  Plain pointers and sizes only.
 *
 * Conventions
 *  - every function returns an int32_t status: LFHE_OK (0) or a negative LFHE_ERR_* code;
 *    lfhe_last_error(ctx) gives the text of the last failure on that context;
 *  - byte outputs use a size query: pass out == NULL to get the required length in *out_len;
 *    a too-small buffer returns LFHE_ERR_SIZE with the required length in *out_len;
 *  - a context is bound to one GPU and owns one HIP stream; calls on one context must not overlap,
 *    distinct contexts are independent;
 *  - there is no CPU fallback: creating a context without a usable GPU fails.
 */
#ifndef LFHE_H
#define LFHE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFHE_OK 0
#define LFHE_ERR_ARG (-1)         /* invalid argument */
#define LFHE_ERR_STATE (-2)       /* missing keys, wrong call order */
#define LFHE_ERR_DEVICE (-3)      /* HIP failure / no GPU */
#define LFHE_ERR_SIZE (-4)        /* output buffer too small */
#define LFHE_ERR_UNSUPPORTED (-5) /* parameter set / operator not supported */
#define LFHE_ERR_FORMAT (-6)      /* malformed ciphertext or key blob */

/* parameter set ids */
#define LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 1u
/* the same set with the bootstrapping key and the blind-rotation accumulator over the NTT prime
 * LFHE_BSK_MODULUS_PRIME instead of 2^64 (LWE ciphertexts, key switch and every interface stay at
 * 2^64; noise and gadget keep their proportion to the modulus) */
#define LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK 2u
#define LFHE_BSK_MODULUS_PRIME 703687441727489ull
/* the prime-modulus set with the LWE key bits taken two at a time in the blind rotation: the
 * bootstrapping key holds four GGSWs per pair of bits (of (1-s)(1-s'), s(1-s'), (1-s)s', s s'), a
 * step multiplies the accumulator by G0 + X^a G1 + X^a' G2 + X^(a+a') G3 -- one external product
 * for two mask elements (the multi-bit blind rotation of tfhe-rs with grouping factor 2).
 * Ciphertexts, secret keys, key switch, encodings and every interface are those of the named set;
 * only the server's bootstrapping key differs (twice the words; 1.38 times the variance of a
 * bootstrap's output, which leaves log2 p_fail at noise level 5 at -129.2 or below -- measured and
 * computed, tools/noise_probe.py and tools/pfail.py -- where the set's own fixture states -129.15). */
#define LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK_GROUPED 3u
/* the named set itself -- every modulus 2^64, as published -- with the key bits taken two at a time:
 * four GGSWs over 2^64 per pair of bits, half the transforms of set 1 (two-prime transform and lift
 * per step).  The lift of a step is exact as long as a sum of 4096 products of a 23-bit digit and a sum
 * of four uniform 64-bit key words (standard deviation 2^90.4) stays inside the +-2^97.6 the two primes
 * tell apart, 149 standard deviations out; every lift is checked against a guard band at the end of
 * that window and a ciphertext that reaches it is redone on a path that is exact without condition
 * (lfhe_lift_fallback_count).  Set 1 is exact without condition. */
#define LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_GROUPED 4u
/* the prime-modulus set with the LWE key bits taken THREE at a time: eight GGSWs per triple of bits
 * (one per product of s_i or 1 - s_i over the triple), 306 blind-rotation steps instead of 459, a
 * server key of 2448 GGSWs (160 MB on the device), 1.46 times the variance of set 1's bootstrap output,
 * log2 p_fail at noise level 5 at -128.9 or below (measured and computed as for set 3).  Everything a
 * client sees is that of the named set.  The device image exists only in factored form: a key with a
 * word that has no inverse modulo the prime (one key in 10^8) is refused at load, LFHE_ERR_UNSUPPORTED. */
#define LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_PRIME_BSK_GROUPED3 5u
/* the named set itself (every modulus 2^64) with the key bits taken three at a time: eight GGSWs over 2^64
 * per triple of bits, a third of the transforms of set 1, a server key of 2448 GGSWs (321 MB on the device
 * for both primes).  The lift window of set 4 applies with eight key words per sum (standard deviation
 * 2^90.9, 107 standard deviations to the end of the window), behind the same guard and exact slow path. */
#define LFHE_PARAMS_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128_GROUPED3 6u

typedef struct lfhe_ctx lfhe_ctx;

/* Numeric parameter set.  The default is the set the reference names at
 * sdk/relayer/src/tfhe.ts:14-19 (values as stored in sdk/relayer/src/test/keys/privateKey.bin). */
typedef struct {
    uint32_t lwe_dim;         /* 918  */
    uint32_t glwe_dim;        /* 1    */
    uint32_t poly_size;       /* 2048 */
    uint32_t pbs_base_log;    /* 23   */
    uint32_t pbs_level;       /* 1    */
    uint32_t ks_base_log;     /* 4    */
    uint32_t ks_level;        /* 4    */
    uint32_t message_modulus; /* 4    */
    uint32_t carry_modulus;   /* 4    */
    uint32_t lwe_noise_log2;  /* 45   */
    uint32_t glwe_noise_log2; /* 17   */
    /* modulus-switch noise reduction (privateKey.bin 24068-24110): encryptions of zero in the
     * server key, bound on the estimate as a power of two of the 2^64 torus, sigma factor, variance
     * of the key-switched input as a fraction of the torus squared.  ms_zeros_count = 0: off. */
    uint32_t ms_zeros_count;  /* 1449 */
    uint32_t ms_bound_log2;   /* 58   */
    double ms_r_sigma_factor; /* 13.179852282053789 */
    double ms_input_variance; /* 2.63039184094559e-07 */
    /* ring modulus of the bootstrapping key: 0 = 2^64 (exact product through two NTT primes and a
     * CRT lift), LFHE_BSK_MODULUS_PRIME = that prime (one transform, no lift; the construction of
     * tfhe-rs programmable_bootstrap_ntt64_lwe_ciphertext) */
    uint64_t bsk_modulus;
    /* LWE key bits per blind-rotation step: 1, 2 (the bootstrapping key then has 4 * ceil(lwe_dim / 2)
     * GGSWs) or 3 (with LFHE_BSK_MODULUS_PRIME only; 8 * ceil(lwe_dim / 3) GGSWs) */
    uint32_t bsk_grouping;
    uint32_t reserved;        /* 0 */
} lfhe_params;

const char* lfhe_version(void);
void lfhe_default_params(lfhe_params* p);

/* ---- context ------------------------------------------------------------------------------------
 * Replaces: engine process start-up (`fhe-server -addr :8448`, e2e/scripts/start-server.sh:17;
 * local-dev/docker-compose.yml:29).  `seed` drives key generation (lfhe_keygen and the keys derived
 * from it): the same seed gives the same keys, so it is as secret as they are; NULL takes 32 bytes
 * from the operating system.  Encryption masks and noise are NOT derived from it: their generator
 * starts from operating-system entropy in every context (see lfhe_encryption_reseed). */
int32_t lfhe_ctx_create(lfhe_ctx** out, uint32_t params_id, int32_t device, const uint8_t seed[32]);
/* same with explicit numbers (the kernels need N = 2048, k = 1, one PBS level of base 2^23; the
 * LWE dimension and the key-switch decomposition are free) */
int32_t lfhe_ctx_create_custom(lfhe_ctx** out, const lfhe_params* p, int32_t device, const uint8_t seed[32]);
void lfhe_ctx_destroy(lfhe_ctx* ctx);
/* Restart the generator behind lfhe_encrypt / lfhe_encrypt_public: from `seed` (reproducible
 * ciphertexts, for tests and known-answer vectors only -- two ciphertexts made from one stream
 * position reveal the difference of their plaintexts), or from fresh operating-system entropy when
 * seed is NULL.  Replaces: the seeder a tfhe-rs client installs before encrypting
 * (ml/extensions/rust/src/lib_wasm.rs:77-136 JsSeeder; ml/extensions/tests/test_random_seed.py). */
int32_t lfhe_encryption_reseed(lfhe_ctx* ctx, const uint8_t seed[32]);
const char* lfhe_last_error(const lfhe_ctx* ctx);
int32_t lfhe_get_params(const lfhe_ctx* ctx, lfhe_params* out);
/* Replaces: GET /health (e2e/test/fhe.test.ts:31-37).  Writes a NUL-terminated JSON object
 * {"status":"ok","threshold":false,"parties":1,...}. */
int32_t lfhe_health(lfhe_ctx* ctx, char* out, uint64_t* out_len);

/* ---- keys ---------------------------------------------------------------------------------------
 * Replaces: server-side key generation at engine start and GET /publickey
 * (packages/luxfhejs/src/index.ts:63-72, packages/hardhat-plugin/src/index.ts:80-87);
 * client-side generateKeys (sdk/relayer/generateKeys.js:20-66). */
int32_t lfhe_keygen(lfhe_ctx* ctx);
/* server keys (bootstrapping key, key-switching key, noise-reduction zeros) for client keys that
 * were imported, e.g. the secret keys of a ClientKey made by the reference's generateKeys.js */
int32_t lfhe_keygen_server(lfhe_ctx* ctx);
#define LFHE_KEY_CLIENT 1u /* secret keys */
#define LFHE_KEY_SERVER 2u /* bootstrapping + key-switching keys (coefficient domain) */
int32_t lfhe_keys_export(lfhe_ctx* ctx, uint32_t which, uint8_t* out, uint64_t* out_len);
int32_t lfhe_keys_import(lfhe_ctx* ctx, uint32_t which, const uint8_t* in, uint64_t len);
/* raw import of server keys: bsk[n][row][poly][N], ksk[N][level][n+1], and the encryptions of
 * zero ms_zeros[ms_zeros_count][n+1] of the modulus-switch noise reduction (NULL / 0: the key set
 * has none and the step is skipped), 64-bit words */
int32_t lfhe_server_keys_import_raw(lfhe_ctx* ctx, const uint64_t* bsk, uint64_t bsk_words, const uint64_t* ksk,
                                    uint64_t ksk_words, const uint64_t* ms_zeros, uint64_t ms_words);
/* raw import of the secret keys: lwe_sk[n], glwe_sk[N], words holding 0 or 1 */
int32_t lfhe_client_keys_import_raw(lfhe_ctx* ctx, const uint64_t* lwe_sk, const uint64_t* glwe_sk);
/* compact public key (an RLWE encryption of zero under the GLWE key, 2N words + header) */
int32_t lfhe_public_key(lfhe_ctx* ctx, uint8_t* out, uint64_t* out_len);
/* device images of the server keys (the second holds the key-switching key and the encryptions of
 * zero), for replication to other GPUs (one broadcast per key set, no
 * per-operation collective): sizes, the pointers owned by this context, and attaching images that
 * were received into caller-owned device memory */
int32_t lfhe_device_keys_size(const lfhe_ctx* ctx, uint64_t* bsk_bytes, uint64_t* ksk_bytes);
int32_t lfhe_device_keys_ptr(lfhe_ctx* ctx, void** bsk_dev, void** ksk_dev);
int32_t lfhe_device_keys_attach(lfhe_ctx* ctx, void* bsk_dev, void* ksk_dev);

/* ---- encrypt / decrypt --------------------------------------------------------------------------
 * Replaces: POST /encrypt {value,bitWidth} (packages/luxfhejs/src/index.ts:127-141,
 * packages/hardhat-plugin/src/index.ts:57-67), TrivialEncrypt
 * (tests/fhevm-suite/e2e/test/utils.ts:209), POST /decrypt
 * (packages/hardhat-plugin/src/index.ts:69-78).  Values are little-endian, 32 bytes.
 * bit_width: 1 (ebool), 8, 16, 32, 64, 128, 160, 256. */
int32_t lfhe_encrypt(lfhe_ctx* ctx, const uint8_t value_le[32], uint32_t bit_width, uint8_t* out, uint64_t* out_len);
int32_t lfhe_trivial_encrypt(lfhe_ctx* ctx, const uint8_t value_le[32], uint32_t bit_width, uint8_t* out,
                             uint64_t* out_len);
/* client-side encryption with the public key alone (no secret key, no GPU work).  Replaces: the
 * compact-list encryption of sdk/relayer/src/sdk/encrypt.ts:54-196 (without its ZK proof). */
int32_t lfhe_encrypt_public(lfhe_ctx* ctx, const uint8_t* pk, uint64_t pk_len, const uint8_t value_le[32],
                            uint32_t bit_width, uint8_t* out, uint64_t* out_len);
int32_t lfhe_decrypt(lfhe_ctx* ctx, const uint8_t* ct, uint64_t ct_len, uint8_t value_le[32], uint32_t* bit_width);
/* Inputs encrypted under a dedicated public-key secret (the reference encrypts user inputs with
 * V1_0_PARAM_PKE_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 keys and casts them to the compute
 * parameters, sdk/relayer/src/tfhe.ts:14-19, sdk/relayer/src/sdk/encrypt.ts:54-196).
 * lfhe_pke_key_set: that secret (N words of 0/1); builds the cast key towards the small key with the
 * set's key-switch decomposition.  lfhe_cast_from_pke: a ciphertext under the public-key secret
 * -> the same values under the compute key, one key switch + bootstrap per block. */
int32_t lfhe_pke_key_set(lfhe_ctx* ctx, const uint64_t* pke_sk);
int32_t lfhe_cast_from_pke(lfhe_ctx* ctx, const uint8_t* ct, uint64_t ct_len, uint8_t* out, uint64_t* out_len);
/* A client's compact ciphertext list (one mask polynomial of N words + one body word per LWE
 * ciphertext, encrypted under the public-key secret set with lfhe_pke_key_set) -> n_values
 * ciphertexts under the compute key, back to back, of the given bit widths in list order.
 * packed != 0: every LWE ciphertext holds two radix blocks (low + 4 * high), as
 * build_with_proof_packed makes them; the bootstrap of the cast splits them.  At most N = 2048 LWE
 * ciphertexts (the reference caps an input at 2048 bits, sdk/relayer/src/constants.ts:1-3).
 * Replaces: the coprocessor's expand-and-cast of the inputs built by
 * sdk/relayer/src/sdk/encrypt.ts:54-196 (CompactCiphertextList.builder ... push_u64 ...
 * build_with_proof_packed); the zero-knowledge proof is not checked here. */
int32_t lfhe_expand_compact_list(lfhe_ctx* ctx, const uint64_t* mask, const uint64_t* bodies, uint64_t n_lwe,
                                 int32_t packed, const uint32_t* bit_widths, uint32_t n_values, uint8_t* out,
                                 uint64_t* out_len);
int32_t lfhe_ciphertext_size(const lfhe_ctx* ctx, uint32_t bit_width, uint64_t* out_len);

/* ---- compression for storage --------------------------------------------------------------------
 * Replaces: CompressionKey / cpu_compress_ciphertexts_into_list / cuda_compress_ciphertexts_into_
 * single_glwe and CompressedModulusSwitched...::extract of the reference's extension
 * (ml/extensions/rust/src/compression.rs:132-291), in the form tfhe-rs gives it for radix
 * ciphertexts (CompressedCiphertextList): up to 256 blocks are packed by a key switch into one GLWE
 * ciphertext of dimension 4 x 256 and stored modulo 2^12 (1.5 bytes per coefficient instead of
 * 16 392 bytes per block); decompression extracts each block and bootstraps it under a key for the
 * compression secret.  Shape and noise: tfhe-rs COMP_PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128
 * (packing base 2^4 x 3 levels, TUniform(43)).
 * lfhe_compression_keygen: makes the compression secret, the packing key and the decompression
 * bootstrapping key from the context's client keys and loads them (server keys must be loaded).
 * lfhe_compress: `cts` = n ciphertexts back to back (any widths; carries must be clear, which every
 * ciphertext this library returns satisfies); writes the blob, or its size when out is NULL.
 * lfhe_decompress: the blob -> the same ciphertexts back to back, one bootstrap per block.
 * lfhe_compression_keys_import_raw: the same two keys made elsewhere -- packing key
 * [2048][3][5 * 256] words (mask polynomials, then body, of the GLWE encryption of
 * s_i 2^(64 - 4 (lv + 1))), bootstrapping key for the 1024 bits of the compression secret in the
 * layout and modulus of the context's bootstrapping key. */
int32_t lfhe_compression_keygen(lfhe_ctx* ctx);
int32_t lfhe_compression_keys_import_raw(lfhe_ctx* ctx, const uint64_t* packing_key, uint64_t packing_words,
                                         const uint64_t* decompression_bsk, uint64_t bsk_words);
int32_t lfhe_compress(lfhe_ctx* ctx, const uint8_t* cts, uint64_t cts_len, uint8_t* out, uint64_t* out_len);
int32_t lfhe_decompress(lfhe_ctx* ctx, const uint8_t* blob, uint64_t blob_len, uint8_t* out, uint64_t* out_len);

/* ---- operators ----------------------------------------------------------------------------------
 * Replaces: POST /evaluate {op,left,right,bitWidth} (e2e/test/fhe.test.ts:105-175,
 * local-dev/demo.ts:57-81) and the executor events FheAdd ... FheIfThenElse
 * (tests/fhevm-suite/e2e/test/utils.ts:182-214). */
enum {
    LFHE_OP_ADD = 0, LFHE_OP_SUB = 1, LFHE_OP_MUL = 2,
    LFHE_OP_EQ = 3, LFHE_OP_NE = 4, LFHE_OP_GE = 5, LFHE_OP_GT = 6, LFHE_OP_LE = 7, LFHE_OP_LT = 8,
    LFHE_OP_MIN = 9, LFHE_OP_MAX = 10,
    LFHE_OP_AND = 11, LFHE_OP_OR = 12, LFHE_OP_XOR = 13,
    LFHE_OP_NOT = 14, LFHE_OP_NEG = 15,
    LFHE_OP_SELECT = 16, /* cond ? lhs : rhs */
    LFHE_OP_SHL = 17, LFHE_OP_SHR = 18, LFHE_OP_ROTL = 19, LFHE_OP_ROTR = 20,
    /* shift / rotation amounts are taken modulo the width; an encrypted amount is a euint8
     * (tests/fhevm-suite/e2e/test/fhevmOperations: shl(euintN, euint8) and shl(euintN, uint8)) */
    LFHE_OP_DIV = 21, LFHE_OP_REM = 22, /* clear divisor only, as in the reference's operator set */
    LFHE_OP_CAST = 23
};
int32_t lfhe_op_from_name(const char* name); /* "add", "sub", "lt", ... ; -1 if unknown */

/* One operation.  rhs may be a ciphertext, or (rhs_is_scalar != 0) a 32-byte little-endian clear
 * value; cond is only read by LFHE_OP_SELECT; unary operators ignore rhs.  The result width is
 * bit_width, or 1 (ebool) for comparisons.  Shifts and rotations take a euint8 ciphertext or a
 * clear amount; division and remainder take a clear divisor (zero is LFHE_ERR_ARG).  Operators
 * with a clear amount or divisor run a schedule specialised on that value, and so does a product
 * with a clear factor (the factor's digits pick the partial products: a 64-bit product costs
 * 200 .. 900 bootstraps instead of 1814, none for 0, 1 and powers of four) and and / or / xor with a
 * clear operand (no bootstrap where its digit is 0 or 3); in a batch, when every instance has the
 * same clear operand. */
int32_t lfhe_evaluate(lfhe_ctx* ctx, uint32_t op, uint32_t bit_width, const uint8_t* lhs, uint64_t lhs_len,
                      const uint8_t* rhs, uint64_t rhs_len, int32_t rhs_is_scalar, const uint8_t* cond,
                      uint64_t cond_len, uint8_t* out, uint64_t* out_len);
/* n independent instances of one operation, bootstrapped together layer by layer.  Every instance
 * has the same widths; ciphertext i of an operand array starts at operand + i * stride where stride
 * is lfhe_ciphertext_size of its width (cond: width 1); scalars are 32 bytes each. */
int32_t lfhe_evaluate_batch(lfhe_ctx* ctx, uint32_t op, uint32_t bit_width, uint64_t n, const uint8_t* lhs,
                            const uint8_t* rhs, int32_t rhs_is_scalar, const uint8_t* cond, uint8_t* out);
/* Instances per pass of lfhe_evaluate_batch / lfhe_evaluate_fused: a batch larger than this runs in
 * passes.  budget_bytes: device memory the arena may take; instance_bytes: arena bytes of one
 * instance.  The result never exceeds the 65535 instances one scatter / gather launch can index.
 * Host logic only (no context, no GPU). */
uint64_t lfhe_arena_chunk(uint64_t budget_bytes, uint64_t instance_bytes);
/* Several operations as ONE schedule.  Values are numbered: 0 .. n_inputs-1 are the inputs,
 * n_inputs + k is the result of step k; a step names its operands by number (-1: unused) or carries
 * a clear right operand.  Intermediate results never leave the GPU and independent steps share
 * bootstrap layers (the subtraction and the addition of a token transfer run side by side).
 * inputs[i] / outs[o]: arrays of n ciphertexts, as in lfhe_evaluate_batch; the width of output o
 * is that of the step it names (1 for comparisons).  Replaces: one coprocessor round trip per
 * executor event (tests/fhevm-suite/e2e/test/utils.ts:182-214) for transactions of one shape, e.g.
 * EncryptedERC20.sol:208-220 (le, select, sub, add). */
typedef struct {
    uint32_t op, bit_width;
    int32_t lhs, rhs, cond;
    uint32_t rhs_is_scalar;
    uint8_t scalar[32];
} lfhe_step;
int32_t lfhe_evaluate_fused(lfhe_ctx* ctx, const lfhe_step* steps, uint32_t nsteps, uint32_t n_inputs,
                            const uint32_t* input_widths, uint64_t n, const uint8_t* const* inputs,
                            const int32_t* outputs, uint32_t n_outputs, uint8_t* const* outs);
/* ---- ciphertexts resident on the GPU, addressed by handle ----------------------------------------
 * Replaces: the coprocessor's ciphertext store behind the executor events -- every event names its
 * operands and its result by bytes32 handle (tests/fhevm-suite/e2e/test/utils.ts:182-214: lhs, rhs,
 * result; test/instance.ts:137-139 reads a result back by handle) -- so that a chain of events keeps
 * its ciphertexts in device memory and only inputs and decryption requests cross PCIe.
 * lfhe_store_put: uploads a ciphertext under `handle` (an existing entry is replaced).
 * lfhe_store_get: downloads it (size query as everywhere).  lfhe_store_drop: frees the entry.
 * lfhe_store_stat: number of entries and device bytes held by the store (either may be NULL); with a
 * handle, also whether it exists (LFHE_ERR_ARG if not) and its bit width.
 * lfhe_evaluate_batch_handles: lfhe_evaluate_batch with every ciphertext operand given as a handle
 * (arrays of n x 32 bytes; a clear right operand stays n x 32 bytes of values) and the results
 * entered under out_handles; an unknown handle or a stored width that does not match is LFHE_ERR_ARG
 * and nothing is stored (a batch too large for one arena of device memory runs in parts, lfhe_arena_chunk
 * instances at a time, and a failure in a later part leaves the results of the parts before it stored).  lfhe_evaluate_fused_handles: the same for lfhe_evaluate_fused
 * (input_handles[i] / out_handles[o]: n handles each).  Results are bit-identical to the host-buffer
 * entry points. */
int32_t lfhe_store_put(lfhe_ctx* ctx, const uint8_t handle[32], const uint8_t* ct, uint64_t ct_len);
int32_t lfhe_store_get(lfhe_ctx* ctx, const uint8_t handle[32], uint8_t* out, uint64_t* out_len);
int32_t lfhe_store_drop(lfhe_ctx* ctx, const uint8_t handle[32]);
int32_t lfhe_store_stat(const lfhe_ctx* ctx, const uint8_t handle[32], uint64_t* entries, uint64_t* device_bytes,
                        uint32_t* bit_width);
int32_t lfhe_evaluate_batch_handles(lfhe_ctx* ctx, uint32_t op, uint32_t bit_width, uint64_t n, const uint8_t* lhs_handles,
                                    const uint8_t* rhs /* n handles, or n clear values of 32 bytes */, int32_t rhs_is_scalar,
                                    const uint8_t* cond_handles, const uint8_t* out_handles);
int32_t lfhe_evaluate_fused_handles(lfhe_ctx* ctx, const lfhe_step* steps, uint32_t nsteps, uint32_t n_inputs,
                                    const uint32_t* input_widths, uint64_t n, const uint8_t* const* input_handles,
                                    const int32_t* outputs, uint32_t n_outputs, const uint8_t* const* out_handles);
/* its schedule as JSON (no context, no GPU); input_noise may be NULL (all 1) */
int32_t lfhe_fused_dump(const lfhe_step* steps, uint32_t nsteps, uint32_t n_inputs, const uint32_t* input_widths,
                        const int32_t* input_noise, const int32_t* outputs, uint32_t n_outputs, char* out,
                        uint64_t* out_len);
/* The bootstrap schedule of an operator as JSON (layers of linear combinations and bootstraps over
 * block ids, the lookup tables, input/output block ids).  Needs no context and no GPU: it is the
 * host-side plan the executor replays, exposed for inspection and for checking the plan against a
 * plaintext model.  "lut_wraps"[i] = 1 marks a table whose bootstraps take inputs up to 31 (the
 * padding bit in use: the result for x >= 16 is minus the entry of x - 16); its entries are values
 * modulo 32 (t >= 16 stands for t - 32), and the constant of the next linear combination brings the
 * block back into [0, 16). */
int32_t lfhe_program_dump(uint32_t op, uint32_t bit_width, int32_t lhs_noise, int32_t rhs_noise, int32_t cond_noise,
                          const uint8_t* scalar /* 32 bytes or NULL: clear shift amount / divisor */, char* out,
                          uint64_t* out_len);
/* Oblivious pseudo-random integer: the low random_bits bits uniform (1 <= random_bits <= bit_width),
 * the rest zero, derived from `seed` with one bootstrap per two bits; the same seed gives the same
 * plaintext under the same keys.  Replaces: the executor events FheRand(randType, seed) and
 * FheRandBounded(upperBound = 2^random_bits, randType, seed)
 * (tests/fhevm-suite/e2e/test/utils.ts:212-213, contracts/Rand.sol). */
int32_t lfhe_rand(lfhe_ctx* ctx, uint32_t bit_width, const uint8_t seed[16], uint32_t random_bits, uint8_t* out,
                  uint64_t* out_len);
/* cumulative count of programmable bootstraps issued by this context */
int32_t lfhe_pbs_count(const lfhe_ctx* ctx, uint64_t* out);
/* How the device holds a grouped bootstrapping key (sets 3 and 4): *out = 1 when its four GGSWs per
 * pair of key bits are held in factored form -- the step's GGSW entry K0 + m1 K1 + m2 K2 + m1 m2 K3
 * as K3 (m1 + K2/K3)(m2 + K1/K3) + (K0 - K1 K2/K3), two modular products per entry instead of
 * three, same results --, 0 for the rows as generated (one-bit keys; a grouped key with a
 * non-invertible word).  No counterpart in the reference: a measurement aid (the benchmark prices
 * the work that is actually done). */
int32_t lfhe_bsk_image_form(const lfhe_ctx* ctx, uint32_t* out);
/* Parameter set 4 (grouped key over 2^64) watches every lift of its blind rotation: a ciphertext with a
 * coefficient in the guard band at the end of the lift's window (the last eighth of it, 130 standard
 * deviations out) is bootstrapped again, in the same call and on the device, with the step taken apart
 * into four products that lift exactly without condition -- the same bits as an exact evaluation, at four
 * times the transforms.  *out = how many ciphertexts of this context took that slow path so far (waits
 * for the context's stream; always 0 for the other sets).  No counterpart in the reference: the guard
 * of a construction the reference does not have. */
int32_t lfhe_lift_fallback_count(lfhe_ctx* ctx, uint64_t* out);

/* ---- the bootstrap path itself ------------------------------------------------------------------
 * Replaces: tfhe-rs keyswitch_programmable_bootstrap + generate_accumulator as the reference calls
 * them at ml/biometrics/notebooks/main.rs:68-71.  A ciphertext block is an LWE under the big key:
 * N + 1 = 2049 words.  A LUT is given as its table over [0, message_modulus*carry_modulus); entries
 * are taken modulo twice that (an entry with the padding bit set is the negative value a negacyclic
 * table returns for it). */
int32_t lfhe_make_lut(lfhe_ctx* ctx, const uint64_t* table, uint64_t* lut_body /* [N] */);
/* host buffers: count bootstraps, LUT bodies [nluts][N], lut_index [count] or NULL (all LUT 0) */
int32_t lfhe_pbs_batch(lfhe_ctx* ctx, uint64_t count, const uint64_t* lwe_in, const uint64_t* lut_bodies,
                       uint32_t nluts, const uint32_t* lut_index, uint64_t* lwe_out);
/* device buffers, asynchronous on the context's stream; lfhe_sync waits */
int32_t lfhe_pbs_batch_device(lfhe_ctx* ctx, uint64_t count, const void* d_lwe_in, const void* d_lut_bodies,
                              const void* d_lut_index, void* d_lwe_out);
int32_t lfhe_sync(lfhe_ctx* ctx);
/* lfhe_pbs_batch_device for callers that cannot allocate device memory: a stream under HIP graph
 * capture, a tight serving loop, a harness that forbids hipMalloc.  No counterpart in the
 * reference.  Same kernels and bit-identical results; the intermediates of the batch live in a
 * static device array of the library (about 25 MiB per GPU, held until the process ends) instead of
 * the context's grow-only buffers, and the call makes no hipMalloc / hipFree.  At most
 * LFHE_NOALLOC_MAX_COUNT ciphertexts per call (more is LFHE_ERR_ARG; so is a custom parameter set
 * with an LWE dimension above 1024).  All calls on one GPU share that GPU's array; the library
 * orders them on the device (a call from another context waits for the previous user's work through
 * an event made at set-up, host threads are serialised for the length of the enqueue), so contexts
 * may call back to back without a wait in between.  The first call on a GPU is the set-up (array
 * address, event): make it outside a graph capture, e.g. with count = 0. */
#define LFHE_NOALLOC_MAX_COUNT 1024u
int32_t lfhe_pbs_batch_device_noalloc(lfhe_ctx* ctx, uint64_t count, const void* d_lwe_in, const void* d_lut_bodies,
                                      const void* d_lut_index, void* d_lwe_out);
/* the same pipeline timed with HIP events on the context's stream: `iters` back-to-back runs;
 * returns the average duration of the key-switch kernel and of the blind-rotation kernel (ms) */
int32_t lfhe_pbs_batch_device_timed(lfhe_ctx* ctx, uint64_t count, const void* d_lwe_in, const void* d_lut_bodies,
                                    const void* d_lut_index, void* d_lwe_out, uint32_t iters, double* ks_ms,
                                    double* br_ms);

/* ---- stages of the path, for parity checks (host buffers) ---------------------------------------
 * Replaces: tfhe-rs keyswitch_lwe_ciphertext, blind_rotate_assign, add_external_product_assign and
 * the polynomial transform inside them (absent dependency; algorithms as published).  The
 * transform is an exact negacyclic NTT over two primes; lfhe_ntt_info reports them with the
 * 2N-th roots so that a checker can evaluate the definition out[j] = sum_i a_i psi^((2j+1) i). */
int32_t lfhe_ntt_info(const lfhe_ctx* ctx, uint64_t primes[2], uint64_t psi[2]);
/* polys [count][N] torus words -> residues [count][2][N] in [0,p), natural index j */
int32_t lfhe_ntt_forward(lfhe_ctx* ctx, uint64_t count, const uint64_t* polys, uint64_t* residues);
/* residues [count][2][N] of a polynomial with coefficients in (-P/2, P/2) -> its coefficients mod 2^64 */
int32_t lfhe_ntt_inverse(lfhe_ctx* ctx, uint64_t count, const uint64_t* residues, uint64_t* polys);
/* out[c] = BSK[ggsw_index[c]] [.] glwe[c]; glwe and out are [count][2][N] (with a prime
 * bsk_modulus: residues in [0, modulus), and lfhe_blind_rotate returns residues too) */
int32_t lfhe_external_product(lfhe_ctx* ctx, uint64_t count, const uint32_t* ggsw_index, const uint64_t* glwe,
                              uint64_t* out);
/* [count][N+1] -> [count][n+1] */
int32_t lfhe_keyswitch(lfhe_ctx* ctx, uint64_t count, const uint64_t* lwe_in, uint64_t* lwe_small_out);
/* modulus-switch noise reduction, in place on [count][n+1] small ciphertexts: the first candidate
 * ct + zero_k within the bound (or the best) replaces ct; chosen[count] (may be NULL) receives k,
 * -1 when nothing was added.  Replaces: tfhe-rs improve_lwe_ciphertext_modulus_switch_noise_for_binary_key
 * (absent dependency; parameters at privateKey.bin 24068-24110). */
int32_t lfhe_ms_noise_reduce(lfhe_ctx* ctx, uint64_t count, uint64_t* lwe_small, int32_t* chosen);
/* [count][n+1] small ciphertexts -> rotated accumulators [count][2][N] */
int32_t lfhe_blind_rotate(lfhe_ctx* ctx, uint64_t count, const uint64_t* lwe_small, const uint64_t* lut_bodies,
                          uint32_t nluts, const uint32_t* lut_index, uint64_t* glwe_out);

#ifdef __cplusplus
}
#endif
#endif
