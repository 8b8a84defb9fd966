/*
 * lfhe_napi.c -- Node N-API addon over the C ABI of liblfhe (include/lfhe.h).
 *
 * This is the binding the reference's TypeScript host layer needs to call the GPU backend in
 * process instead of over HTTP: packages/luxfhejs/src/index.ts:63-141 (getPublicKey, encrypt_*)
 * and packages/hardhat-plugin/src/index.ts:57-87 (encrypt, decrypt, getPublicKey) map one to one
 * onto the functions exported here; index.js next to this file wraps them in a class with the
 * SDK's method names.
 *
 * The library is opened with dlopen when the addon is first used, so loading the addon itself
 * needs neither a GPU nor the library; creating a context without a GPU throws.
 * The plain calls are synchronous.  evaluateAsync returns a Promise and runs on a libuv worker
 * thread, so a 40 ms multiplication does not block the event loop; requests that are pending when
 * a worker gets the library and share context, operator, width and operand kind go out as ONE
 * lfhe_evaluate_batch call (a bootstrap layer of a single operation leaves most of the GPU idle, so
 * sixteen concurrent promises cost little more than one); index.js gathers the calls of one event-
 * loop turn before they get here (evaluateBatchAsync).  The library is entered by one thread at
 * a time (a context is single-threaded).
 * Values cross the boundary as 32-byte little-endian buffers (index.js converts BigInt).
 */
#define NAPI_VERSION 6
#include <dlfcn.h>
#include <node_api.h>
#include <pthread.h>
#include <time.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lfhe.h"

/* ---- the C ABI, resolved at run time ---- */
static void* g_lib;
#define FN(ret, name, args) static ret(*p_##name) args;
FN(int32_t, lfhe_ctx_create, (lfhe_ctx**, uint32_t, int32_t, const uint8_t*))
FN(void, lfhe_ctx_destroy, (lfhe_ctx*))
FN(const char*, lfhe_last_error, (const lfhe_ctx*))
FN(const char*, lfhe_version, (void))
FN(int32_t, lfhe_health, (lfhe_ctx*, char*, uint64_t*))
FN(int32_t, lfhe_keygen, (lfhe_ctx*))
FN(int32_t, lfhe_keys_export, (lfhe_ctx*, uint32_t, uint8_t*, uint64_t*))
FN(int32_t, lfhe_keys_import, (lfhe_ctx*, uint32_t, const uint8_t*, uint64_t))
FN(int32_t, lfhe_public_key, (lfhe_ctx*, uint8_t*, uint64_t*))
FN(int32_t, lfhe_encrypt, (lfhe_ctx*, const uint8_t*, uint32_t, uint8_t*, uint64_t*))
FN(int32_t, lfhe_trivial_encrypt, (lfhe_ctx*, const uint8_t*, uint32_t, uint8_t*, uint64_t*))
FN(int32_t, lfhe_encrypt_public, (lfhe_ctx*, const uint8_t*, uint64_t, const uint8_t*, uint32_t, uint8_t*, uint64_t*))
FN(int32_t, lfhe_decrypt, (lfhe_ctx*, const uint8_t*, uint64_t, uint8_t*, uint32_t*))
FN(int32_t, lfhe_ciphertext_size, (const lfhe_ctx*, uint32_t, uint64_t*))
FN(int32_t, lfhe_op_from_name, (const char*))
FN(int32_t, lfhe_evaluate, (lfhe_ctx*, uint32_t, uint32_t, const uint8_t*, uint64_t, const uint8_t*, uint64_t, int32_t,
                            const uint8_t*, uint64_t, uint8_t*, uint64_t*))
FN(int32_t, lfhe_evaluate_batch, (lfhe_ctx*, uint32_t, uint32_t, uint64_t, const uint8_t*, const uint8_t*, int32_t,
                                  const uint8_t*, uint8_t*))
FN(int32_t, lfhe_pbs_count, (const lfhe_ctx*, uint64_t*))
FN(int32_t, lfhe_rand, (lfhe_ctx*, uint32_t, const uint8_t*, uint32_t, uint8_t*, uint64_t*))
FN(int32_t, lfhe_keygen_server, (lfhe_ctx*))
FN(int32_t, lfhe_client_keys_import_raw, (lfhe_ctx*, const uint64_t*, const uint64_t*))
FN(int32_t, lfhe_pke_key_set, (lfhe_ctx*, const uint64_t*))
FN(int32_t, lfhe_cast_from_pke, (lfhe_ctx*, const uint8_t*, uint64_t, uint8_t*, uint64_t*))
FN(int32_t, lfhe_expand_compact_list, (lfhe_ctx*, const uint64_t*, const uint64_t*, uint64_t, int32_t, const uint32_t*,
                                       uint32_t, uint8_t*, uint64_t*))
FN(int32_t, lfhe_compression_keygen, (lfhe_ctx*))
FN(int32_t, lfhe_compress, (lfhe_ctx*, const uint8_t*, uint64_t, uint8_t*, uint64_t*))
FN(int32_t, lfhe_decompress, (lfhe_ctx*, const uint8_t*, uint64_t, uint8_t*, uint64_t*))
FN(int32_t, lfhe_store_put, (lfhe_ctx*, const uint8_t*, const uint8_t*, uint64_t))
FN(int32_t, lfhe_store_get, (lfhe_ctx*, const uint8_t*, uint8_t*, uint64_t*))
FN(int32_t, lfhe_store_drop, (lfhe_ctx*, const uint8_t*))
FN(int32_t, lfhe_store_stat, (const lfhe_ctx*, const uint8_t*, uint64_t*, uint64_t*, uint32_t*))
FN(int32_t, lfhe_evaluate_batch_handles, (lfhe_ctx*, uint32_t, uint32_t, uint64_t, const uint8_t*, const uint8_t*, int32_t,
                                          const uint8_t*, const uint8_t*))
#undef FN

static napi_value throw_msg(napi_env env, const char* msg) {
    napi_throw_error(env, NULL, msg);
    return NULL;
}

static int resolve_all(napi_env env, const char* path) {
    if (g_lib) return 1;
    g_lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) {
        char buf[768];
        snprintf(buf, sizeof buf, "cannot load the lfhe library (%s): %s", path, dlerror());
        throw_msg(env, buf);
        return 0;
    }
#define GET(name)                                               \
    *(void**)(&p_##name) = dlsym(g_lib, #name);                 \
    if (!p_##name) {                                            \
        throw_msg(env, "symbol missing from the library: " #name); \
        dlclose(g_lib); g_lib = NULL;                           \
        return 0;                                               \
    }
    GET(lfhe_ctx_create) GET(lfhe_ctx_destroy) GET(lfhe_last_error) GET(lfhe_version) GET(lfhe_health)
    GET(lfhe_keygen) GET(lfhe_keys_export) GET(lfhe_keys_import) GET(lfhe_public_key) GET(lfhe_encrypt)
    GET(lfhe_trivial_encrypt) GET(lfhe_encrypt_public) GET(lfhe_decrypt) GET(lfhe_ciphertext_size)
    GET(lfhe_op_from_name) GET(lfhe_evaluate) GET(lfhe_evaluate_batch) GET(lfhe_pbs_count)
    GET(lfhe_rand) GET(lfhe_keygen_server) GET(lfhe_client_keys_import_raw) GET(lfhe_pke_key_set) GET(lfhe_cast_from_pke)
    GET(lfhe_compression_keygen) GET(lfhe_compress) GET(lfhe_decompress) GET(lfhe_expand_compact_list)
    GET(lfhe_store_put) GET(lfhe_store_get) GET(lfhe_store_drop) GET(lfhe_store_stat) GET(lfhe_evaluate_batch_handles)
#undef GET
    return 1;
}

#define NEED_LIB() if (!g_lib) return throw_msg(env, "call open(path) first")

/* ---- argument helpers ---- */
static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        throw_msg(env, "wrong number of arguments");
        return 0;
    }
    return 1;
}

static lfhe_ctx* get_ctx(napi_env env, napi_value v) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        throw_msg(env, "not a context handle");
        return NULL;
    }
    lfhe_ctx** slot = (lfhe_ctx**)p;
    if (!*slot) throw_msg(env, "context was destroyed");
    return *slot;
}

/* Buffer / Uint8Array -> pointer + length; null/undefined -> (NULL, 0) */
static int get_bytes(napi_env env, napi_value v, const uint8_t** data, size_t* len) {
    napi_valuetype t;
    napi_typeof(env, v, &t);
    if (t == napi_null || t == napi_undefined) { *data = NULL; *len = 0; return 1; }
    bool is_buf = false, is_ta = false;
    napi_is_buffer(env, v, &is_buf);
    if (is_buf) {
        void* p;
        napi_get_buffer_info(env, v, &p, len);
        *data = (const uint8_t*)p;
        return 1;
    }
    napi_is_typedarray(env, v, &is_ta);
    if (is_ta) {
        napi_typedarray_type tt;
        void* p;
        napi_value ab;
        size_t off;
        napi_get_typedarray_info(env, v, &tt, len, &p, &ab, &off);
        if (tt == napi_uint8_array) { *data = (const uint8_t*)p; return 1; }
    }
    throw_msg(env, "expected a Buffer or Uint8Array");
    return 0;
}

static napi_value fail(napi_env env, lfhe_ctx* c, int32_t rc, const char* what) {
    char buf[640];
    snprintf(buf, sizeof buf, "%s failed (%d): %s", what, rc, c ? p_lfhe_last_error(c) : "no context");
    return throw_msg(env, buf);
}

static napi_value make_buffer(napi_env env, const uint8_t* src, size_t len) {
    napi_value out;
    void* dst;
    if (napi_create_buffer_copy(env, len, src, &dst, &out) != napi_ok) return throw_msg(env, "out of memory");
    return out;
}

static void finalize_ctx(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    lfhe_ctx** slot = (lfhe_ctx**)data;
    if (*slot && p_lfhe_ctx_destroy) p_lfhe_ctx_destroy(*slot);
    free(slot);
}

/* ---- exported functions ---- */
static napi_value js_open(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    char path[512];
    size_t n;
    if (napi_get_value_string_utf8(env, argv[0], path, sizeof path, &n) != napi_ok) return throw_msg(env, "path expected");
    if (!resolve_all(env, path)) return NULL;
    napi_value v;
    napi_create_string_utf8(env, p_lfhe_version(), NAPI_AUTO_LENGTH, &v);
    return v;
}

static napi_value js_create(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    const uint8_t* seed; size_t slen;
    if (!get_bytes(env, argv[0], &seed, &slen)) return NULL;
    if (slen != 32) return throw_msg(env, "seed must be 32 bytes");
    int32_t device; uint32_t params;
    napi_get_value_int32(env, argv[1], &device);
    napi_get_value_uint32(env, argv[2], &params);
    lfhe_ctx* c = NULL;
    const int32_t rc = p_lfhe_ctx_create(&c, params, device, seed);
    if (rc != LFHE_OK) {
        char buf[160];
        snprintf(buf, sizeof buf, "lfhe_ctx_create failed (%d): no usable GPU or unsupported parameters", rc);
        return throw_msg(env, buf);
    }
    lfhe_ctx** slot = (lfhe_ctx**)malloc(sizeof *slot);
    if (!slot) { p_lfhe_ctx_destroy(c); return throw_msg(env, "out of memory"); }
    *slot = c;
    napi_value ext;
    napi_create_external(env, slot, finalize_ctx, NULL, &ext);
    return ext;
}

static napi_value js_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    void* p = NULL;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
        lfhe_ctx** slot = (lfhe_ctx**)p;
        if (*slot) { p_lfhe_ctx_destroy(*slot); *slot = NULL; }
    }
    return NULL;
}

static napi_value js_keygen(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const int32_t rc = p_lfhe_keygen(c);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_keygen");
    return NULL;
}

static napi_value js_health(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    char buf[512];
    uint64_t n = sizeof buf;
    const int32_t rc = p_lfhe_health(c, buf, &n);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_health");
    napi_value v;
    napi_create_string_utf8(env, buf, NAPI_AUTO_LENGTH, &v);
    return v;
}

/* size-query pattern shared by the blob getters */
typedef int32_t (*blob_fn)(lfhe_ctx*, uint32_t, uint8_t*, uint64_t*);
static napi_value blob_out(napi_env env, lfhe_ctx* c, blob_fn fn, uint32_t which, const char* what) {
    uint64_t n = 0;
    int32_t rc = fn(c, which, NULL, &n);
    if (rc != LFHE_OK) return fail(env, c, rc, what);
    uint8_t* tmp = (uint8_t*)malloc(n ? n : 1);
    if (!tmp) return throw_msg(env, "out of memory");
    rc = fn(c, which, tmp, &n);
    napi_value out = rc == LFHE_OK ? make_buffer(env, tmp, n) : fail(env, c, rc, what);
    free(tmp);
    return out;
}
static int32_t pk_adapter(lfhe_ctx* c, uint32_t which, uint8_t* out, uint64_t* n) { (void)which; return p_lfhe_public_key(c, out, n); }

static napi_value js_public_key(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    return c ? blob_out(env, c, pk_adapter, 0, "lfhe_public_key") : NULL;
}

static napi_value js_keys_export(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    uint32_t which;
    napi_get_value_uint32(env, argv[1], &which);
    return blob_out(env, c, p_lfhe_keys_export, which, "lfhe_keys_export");
}

static napi_value js_keys_import(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    uint32_t which;
    napi_get_value_uint32(env, argv[1], &which);
    const uint8_t* d; size_t n;
    if (!get_bytes(env, argv[2], &d, &n)) return NULL;
    const int32_t rc = p_lfhe_keys_import(c, which, d, n);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_keys_import");
    return NULL;
}

/* encrypt(ctx, valueLE32, bits, mode) mode: 0 secret key, 1 trivial */
static napi_value js_encrypt(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t* v; size_t vl;
    if (!get_bytes(env, argv[1], &v, &vl)) return NULL;
    if (vl != 32) return throw_msg(env, "value must be 32 little-endian bytes");
    uint32_t bits, mode;
    napi_get_value_uint32(env, argv[2], &bits);
    napi_get_value_uint32(env, argv[3], &mode);
    uint64_t n = 0;
    int32_t rc = p_lfhe_ciphertext_size(c, bits, &n);
    if (rc != LFHE_OK) return throw_msg(env, "Encryption failed: unsupported bit width");
    uint8_t* tmp = (uint8_t*)malloc(n);
    if (!tmp) return throw_msg(env, "out of memory");
    rc = mode ? p_lfhe_trivial_encrypt(c, v, bits, tmp, &n) : p_lfhe_encrypt(c, v, bits, tmp, &n);
    napi_value out = rc == LFHE_OK ? make_buffer(env, tmp, n) : fail(env, c, rc, "Encryption");
    free(tmp);
    return out;
}

static napi_value js_encrypt_public(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t *pk, *v; size_t pl, vl;
    if (!get_bytes(env, argv[1], &pk, &pl) || !get_bytes(env, argv[2], &v, &vl)) return NULL;
    if (vl != 32) return throw_msg(env, "value must be 32 little-endian bytes");
    uint32_t bits;
    napi_get_value_uint32(env, argv[3], &bits);
    uint64_t n = 0;
    if (p_lfhe_ciphertext_size(c, bits, &n) != LFHE_OK) return throw_msg(env, "Encryption failed: unsupported bit width");
    uint8_t* tmp = (uint8_t*)malloc(n);
    if (!tmp) return throw_msg(env, "out of memory");
    const int32_t rc = p_lfhe_encrypt_public(c, pk, pl, v, bits, tmp, &n);
    napi_value out = rc == LFHE_OK ? make_buffer(env, tmp, n) : fail(env, c, rc, "Encryption");
    free(tmp);
    return out;
}

/* decrypt(ctx, ct) -> {value: Buffer(32) little-endian, bitWidth} */
static napi_value js_decrypt(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t* ct; size_t n;
    if (!get_bytes(env, argv[1], &ct, &n)) return NULL;
    uint8_t val[32];
    uint32_t bits = 0;
    const int32_t rc = p_lfhe_decrypt(c, ct, n, val, &bits);
    if (rc != LFHE_OK) return fail(env, c, rc, "Decryption");
    napi_value obj, b, w;
    napi_create_object(env, &obj);
    b = make_buffer(env, val, 32);
    napi_create_uint32(env, bits, &w);
    napi_set_named_property(env, obj, "value", b);
    napi_set_named_property(env, obj, "bitWidth", w);
    return obj;
}

static napi_value js_op_from_name(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    char name[32];
    size_t n;
    if (napi_get_value_string_utf8(env, argv[0], name, sizeof name, &n) != napi_ok) return throw_msg(env, "name expected");
    napi_value v;
    napi_create_int32(env, p_lfhe_op_from_name(name), &v);
    return v;
}

/* evaluate(ctx, op, bits, lhs, rhs|null, rhsIsScalar, cond|null) -> Buffer */
static napi_value js_evaluate(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[7];
    if (!get_args(env, info, 7, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    uint32_t op, bits;
    napi_get_value_uint32(env, argv[1], &op);
    napi_get_value_uint32(env, argv[2], &bits);
    const uint8_t *l, *r, *cd; size_t ll, rl, cl;
    if (!get_bytes(env, argv[3], &l, &ll) || !get_bytes(env, argv[4], &r, &rl) || !get_bytes(env, argv[6], &cd, &cl)) return NULL;
    bool scalar = false;
    napi_get_value_bool(env, argv[5], &scalar);
    uint64_t n = 0;
    int32_t rc = p_lfhe_evaluate(c, op, bits, l, ll, r, rl, scalar, cd, cl, NULL, &n);
    if (rc != LFHE_OK) return fail(env, c, rc, "Evaluation");
    uint8_t* tmp = (uint8_t*)malloc(n);
    if (!tmp) return throw_msg(env, "out of memory");
    rc = p_lfhe_evaluate(c, op, bits, l, ll, r, rl, scalar, cd, cl, tmp, &n);
    napi_value out = rc == LFHE_OK ? make_buffer(env, tmp, n) : fail(env, c, rc, "Evaluation");
    free(tmp);
    return out;
}

/* evaluateBatch(ctx, op, bits, count, lhsAll, rhsAll|null, rhsIsScalar, condAll|null, outBits) -> Buffer (count results) */
static napi_value js_evaluate_batch(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[9];
    if (!get_args(env, info, 9, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    uint32_t op, bits, count, out_bits;
    napi_get_value_uint32(env, argv[1], &op);
    napi_get_value_uint32(env, argv[2], &bits);
    napi_get_value_uint32(env, argv[3], &count);
    napi_get_value_uint32(env, argv[8], &out_bits);
    const uint8_t *l, *r, *cd; size_t ll, rl, cl;
    if (!get_bytes(env, argv[4], &l, &ll) || !get_bytes(env, argv[5], &r, &rl) || !get_bytes(env, argv[7], &cd, &cl)) return NULL;
    bool scalar = false;
    napi_get_value_bool(env, argv[6], &scalar);
    uint64_t in_sz = 0, out_sz = 0, rhs_sz = 0, cond_sz = 0;
    if (p_lfhe_ciphertext_size(c, bits, &in_sz) != LFHE_OK || p_lfhe_ciphertext_size(c, out_bits, &out_sz) != LFHE_OK)
        return throw_msg(env, "unsupported bit width");
    p_lfhe_ciphertext_size(c, 1, &cond_sz);
    const int shift = op >= LFHE_OP_SHL && op <= LFHE_OP_ROTR;
    if (shift) p_lfhe_ciphertext_size(c, 8, &rhs_sz); else rhs_sz = in_sz;
    if (ll != (size_t)count * in_sz || (r && rl != (size_t)count * (scalar ? 32 : rhs_sz)) || (cd && cl != (size_t)count * cond_sz))
        return throw_msg(env, "operand arrays do not hold `count` ciphertexts of the stated width");
    uint8_t* tmp = (uint8_t*)malloc((size_t)count * out_sz + 1);
    if (!tmp) return throw_msg(env, "out of memory");
    const int32_t rc = p_lfhe_evaluate_batch(c, op, bits, count, l, r, scalar, cd, tmp);
    napi_value out = rc == LFHE_OK ? make_buffer(env, tmp, (size_t)count * out_sz) : fail(env, c, rc, "Evaluation");
    free(tmp);
    return out;
}

/* ---- evaluateAsync: Promise-returning evaluation on the libuv pool, with coalescing ---- */
typedef struct eval_req {
    lfhe_ctx* c;
    uint32_t op, bits;
    int scalar;
    uint8_t *l, *r, *cd;          /* copies of the operands: the JS buffers may be gone when the worker runs */
    size_t ll, rl, cl;
    uint8_t* out;
    uint64_t out_len;
    int32_t rc;
    char err[320];
    int state;                    /* 0 queued, 1 taken by a batch, 2 done */
    struct eval_req* next;
    napi_deferred deferred;
    napi_async_work work;
} eval_req;

static pthread_mutex_t g_q_mutex = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_q_cond = PTHREAD_COND_INITIALIZER;
static pthread_mutex_t g_lib_mutex = PTHREAD_MUTEX_INITIALIZER;   /* one thread inside the library at a time */
static eval_req* g_q_head;

static void req_free(eval_req* q) {
    free(q->l); free(q->r); free(q->cd); free(q->out); free(q);
}

static uint8_t* dup_bytes(const uint8_t* p, size_t n) {
    if (!p) return NULL;
    uint8_t* d = (uint8_t*)malloc(n ? n : 1);
    if (d) memcpy(d, p, n);
    return d;
}

static void run_one(eval_req* q) {
    uint64_t n = 0;
    q->rc = p_lfhe_evaluate(q->c, q->op, q->bits, q->l, q->ll, q->r, q->rl, q->scalar, q->cd, q->cl, NULL, &n);
    if (q->rc == LFHE_OK) {
        q->out = (uint8_t*)malloc(n ? n : 1);
        if (!q->out) { q->rc = LFHE_ERR_DEVICE; snprintf(q->err, sizeof q->err, "out of memory"); return; }
        q->rc = p_lfhe_evaluate(q->c, q->op, q->bits, q->l, q->ll, q->r, q->rl, q->scalar, q->cd, q->cl, q->out, &n);
        q->out_len = n;
    }
    if (q->rc != LFHE_OK) snprintf(q->err, sizeof q->err, "Evaluation failed (%d): %s", q->rc, p_lfhe_last_error(q->c));
}

static int same_shape(const eval_req* a, const eval_req* b) {
    return a->c == b->c && a->op == b->op && a->bits == b->bits && a->scalar == b->scalar && a->ll == b->ll &&
           a->rl == b->rl && a->cl == b->cl && (a->r != NULL) == (b->r != NULL) && (a->cd != NULL) == (b->cd != NULL);
}

/* worker thread: serve this request and every queued request of the same shape in one batch */
static void eval_execute(napi_env env, void* data) {
    (void)env;
    eval_req* me = (eval_req*)data;
    /* Whoever holds the library gathers what is pending at that moment: a worker that started while
     * the JS thread was still queueing a burst (Promise.all over many calls) would otherwise take the
     * first request alone, and so would each of its siblings.  Waiting here also means that everything
     * that arrives while one batch runs goes out together in the next. */
    pthread_mutex_lock(&g_lib_mutex);
    pthread_mutex_lock(&g_q_mutex);
    if (me->state != 0) {                       /* another worker took it into its batch */
        pthread_mutex_unlock(&g_lib_mutex);
        while (me->state != 2) pthread_cond_wait(&g_q_cond, &g_q_mutex);
        pthread_mutex_unlock(&g_q_mutex);
        return;
    }
    eval_req* batch[256];
    size_t count = 0;
    for (eval_req** pp = &g_q_head; *pp;) {
        eval_req* q = *pp;
        if (q->state == 0 && count < 256 && (q == me || (me->op != LFHE_OP_CAST && same_shape(q, me)))) {
            q->state = 1;
            batch[count++] = q;
            *pp = q->next;
        } else {
            pp = &q->next;
        }
    }
    pthread_mutex_unlock(&g_q_mutex);

    int done = 0;
    if (count > 1) {
        uint64_t out_sz = 0;
        const int cmp = me->op >= LFHE_OP_EQ && me->op <= LFHE_OP_LT;
        uint8_t *L = (uint8_t*)malloc(count * me->ll), *R = me->r ? (uint8_t*)malloc(count * me->rl) : NULL,
                *Cd = me->cd ? (uint8_t*)malloc(count * me->cl) : NULL, *O = NULL;
        if (p_lfhe_ciphertext_size(me->c, cmp ? 1 : me->bits, &out_sz) == LFHE_OK) O = (uint8_t*)malloc(count * out_sz);
        if (L && O && (!me->r || R) && (!me->cd || Cd)) {
            for (size_t i = 0; i < count; ++i) {
                memcpy(L + i * me->ll, batch[i]->l, me->ll);
                if (R) memcpy(R + i * me->rl, batch[i]->r, me->rl);
                if (Cd) memcpy(Cd + i * me->cl, batch[i]->cd, me->cl);
            }
            if (p_lfhe_evaluate_batch(me->c, me->op, me->bits, count, L, R, me->scalar, Cd, O) == LFHE_OK) {
                done = 1;
                for (size_t i = 0; i < count; ++i) {
                    batch[i]->out = dup_bytes(O + i * out_sz, out_sz);
                    batch[i]->out_len = out_sz;
                    batch[i]->rc = batch[i]->out ? LFHE_OK : LFHE_ERR_DEVICE;
                    if (!batch[i]->out) snprintf(batch[i]->err, sizeof batch[i]->err, "out of memory");
                }
            }
        }
        free(L); free(R); free(Cd); free(O);
    }
    if (!done)                                   /* a single request, or a batch that failed: one by one, each with its own error */
        for (size_t i = 0; i < count; ++i) run_one(batch[i]);
    pthread_mutex_unlock(&g_lib_mutex);

    pthread_mutex_lock(&g_q_mutex);
    for (size_t i = 0; i < count; ++i) batch[i]->state = 2;
    pthread_cond_broadcast(&g_q_cond);
    pthread_mutex_unlock(&g_q_mutex);
}

/* main thread: settle the promise */
static void eval_complete(napi_env env, napi_status status, void* data) {
    eval_req* q = (eval_req*)data;
    if (status == napi_ok && q->rc == LFHE_OK) {
        napi_value buf;
        void* dst;
        if (napi_create_buffer_copy(env, q->out_len, q->out, &dst, &buf) == napi_ok) napi_resolve_deferred(env, q->deferred, buf);
        else { napi_value e, m; napi_create_string_utf8(env, "out of memory", NAPI_AUTO_LENGTH, &m); napi_create_error(env, NULL, m, &e); napi_reject_deferred(env, q->deferred, e); }
    } else {
        napi_value e, m;
        napi_create_string_utf8(env, status == napi_ok ? q->err : "evaluation was cancelled", NAPI_AUTO_LENGTH, &m);
        napi_create_error(env, NULL, m, &e);
        napi_reject_deferred(env, q->deferred, e);
    }
    napi_delete_async_work(env, q->work);
    req_free(q);
}

/* evaluateAsync(ctx, op, bits, lhs, rhs|null, rhsIsScalar, cond|null) -> Promise<Buffer> */
static napi_value js_evaluate_async(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[7];
    if (!get_args(env, info, 7, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t *l, *r, *cd; size_t ll, rl, cl;
    if (!get_bytes(env, argv[3], &l, &ll) || !get_bytes(env, argv[4], &r, &rl) || !get_bytes(env, argv[6], &cd, &cl)) return NULL;
    eval_req* q = (eval_req*)calloc(1, sizeof *q);
    if (!q) return throw_msg(env, "out of memory");
    q->c = c;
    napi_get_value_uint32(env, argv[1], &q->op);
    napi_get_value_uint32(env, argv[2], &q->bits);
    bool scalar = false;
    napi_get_value_bool(env, argv[5], &scalar);
    q->scalar = scalar ? 1 : 0;
    q->l = dup_bytes(l, ll); q->r = dup_bytes(r, rl); q->cd = dup_bytes(cd, cl);
    q->ll = ll; q->rl = rl; q->cl = cl;
    if (!l || !q->l || (r && !q->r) || (cd && !q->cd)) { req_free(q); return throw_msg(env, l ? "out of memory" : "missing left operand"); }
    napi_value promise, name;
    if (napi_create_promise(env, &q->deferred, &promise) != napi_ok) { req_free(q); return throw_msg(env, "cannot create a promise"); }
    napi_create_string_utf8(env, "lfhe.evaluate", NAPI_AUTO_LENGTH, &name);
    if (napi_create_async_work(env, NULL, name, eval_execute, eval_complete, q, &q->work) != napi_ok) {
        req_free(q);
        return throw_msg(env, "cannot create async work");
    }
    pthread_mutex_lock(&g_q_mutex);               /* queued before the work can start: a running worker may already take it */
    eval_req** pp = &g_q_head;
    while (*pp) pp = &(*pp)->next;
    *pp = q;
    pthread_mutex_unlock(&g_q_mutex);
    if (napi_queue_async_work(env, q->work) != napi_ok) {
        pthread_mutex_lock(&g_q_mutex);
        for (pp = &g_q_head; *pp; pp = &(*pp)->next)
            if (*pp == q) { *pp = q->next; break; }
        const int taken = q->state != 0;
        pthread_mutex_unlock(&g_q_mutex);
        if (!taken) { napi_delete_async_work(env, q->work); req_free(q); }
        return throw_msg(env, "cannot queue async work");
    }
    return promise;
}

/* ---- evaluateBatchAsync: one lfhe_evaluate_batch call on the libuv pool ---- */
typedef struct {
    lfhe_ctx* c;
    uint32_t op, bits, count;
    int scalar;
    uint8_t *l, *r, *cd, *out;
    size_t out_len;
    int32_t rc;
    char err[320];
    napi_deferred deferred;
    napi_async_work work;
} batch_req;

static void batch_free(batch_req* q) { free(q->l); free(q->r); free(q->cd); free(q->out); free(q); }

static void batch_execute(napi_env env, void* data) {
    (void)env;
    batch_req* q = (batch_req*)data;
    pthread_mutex_lock(&g_lib_mutex);
    q->rc = p_lfhe_evaluate_batch(q->c, q->op, q->bits, q->count, q->l, q->r, q->scalar, q->cd, q->out);
    if (q->rc != LFHE_OK) snprintf(q->err, sizeof q->err, "Evaluation failed (%d): %s", q->rc, p_lfhe_last_error(q->c));
    pthread_mutex_unlock(&g_lib_mutex);
}

static void batch_complete(napi_env env, napi_status status, void* data) {
    batch_req* q = (batch_req*)data;
    napi_value buf, e, m;
    void* dst;
    if (status == napi_ok && q->rc == LFHE_OK && napi_create_buffer_copy(env, q->out_len, q->out, &dst, &buf) == napi_ok) {
        napi_resolve_deferred(env, q->deferred, buf);
    } else {
        napi_create_string_utf8(env, status != napi_ok ? "evaluation was cancelled" : q->rc != LFHE_OK ? q->err : "out of memory",
                                NAPI_AUTO_LENGTH, &m);
        napi_create_error(env, NULL, m, &e);
        napi_reject_deferred(env, q->deferred, e);
    }
    napi_delete_async_work(env, q->work);
    batch_free(q);
}

/* evaluateBatchAsync(ctx, op, bits, count, lhsAll, rhsAll|null, rhsIsScalar, condAll|null, outBits) -> Promise<Buffer> */
static napi_value js_evaluate_batch_async(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[9];
    if (!get_args(env, info, 9, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    uint32_t op, bits, count, out_bits;
    napi_get_value_uint32(env, argv[1], &op);
    napi_get_value_uint32(env, argv[2], &bits);
    napi_get_value_uint32(env, argv[3], &count);
    napi_get_value_uint32(env, argv[8], &out_bits);
    const uint8_t *l, *r, *cd; size_t ll, rl, cl;
    if (!get_bytes(env, argv[4], &l, &ll) || !get_bytes(env, argv[5], &r, &rl) || !get_bytes(env, argv[7], &cd, &cl)) return NULL;
    bool scalar = false;
    napi_get_value_bool(env, argv[6], &scalar);
    uint64_t in_sz = 0, out_sz = 0, rhs_sz = 0, cond_sz = 0;
    if (p_lfhe_ciphertext_size(c, bits, &in_sz) != LFHE_OK || p_lfhe_ciphertext_size(c, out_bits, &out_sz) != LFHE_OK)
        return throw_msg(env, "unsupported bit width");
    p_lfhe_ciphertext_size(c, 1, &cond_sz);
    const int shift = op >= LFHE_OP_SHL && op <= LFHE_OP_ROTR;
    if (shift) p_lfhe_ciphertext_size(c, 8, &rhs_sz); else rhs_sz = in_sz;
    if (!count || !l || ll != (size_t)count * in_sz || (r && rl != (size_t)count * (scalar ? 32 : rhs_sz)) ||
        (cd && cl != (size_t)count * cond_sz))
        return throw_msg(env, "operand arrays do not hold `count` ciphertexts of the stated width");
    batch_req* q = (batch_req*)calloc(1, sizeof *q);
    if (!q) return throw_msg(env, "out of memory");
    q->c = c; q->op = op; q->bits = bits; q->count = count; q->scalar = scalar ? 1 : 0;
    q->l = dup_bytes(l, ll); q->r = dup_bytes(r, rl); q->cd = dup_bytes(cd, cl);
    q->out_len = (size_t)count * out_sz;
    q->out = (uint8_t*)malloc(q->out_len + 1);
    if (!q->l || (r && !q->r) || (cd && !q->cd) || !q->out) { batch_free(q); return throw_msg(env, "out of memory"); }
    napi_value promise, name;
    if (napi_create_promise(env, &q->deferred, &promise) != napi_ok) { batch_free(q); return throw_msg(env, "cannot create a promise"); }
    napi_create_string_utf8(env, "lfhe.evaluateBatch", NAPI_AUTO_LENGTH, &name);
    if (napi_create_async_work(env, NULL, name, batch_execute, batch_complete, q, &q->work) != napi_ok ||
        napi_queue_async_work(env, q->work) != napi_ok) {
        batch_free(q);
        return throw_msg(env, "cannot queue async work");
    }
    return promise;
}

/* rand(ctx, bits, seed16, randomBits) -> Buffer */
static napi_value js_rand(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    uint32_t bits, rbits;
    napi_get_value_uint32(env, argv[1], &bits);
    napi_get_value_uint32(env, argv[3], &rbits);
    const uint8_t* seed; size_t sl;
    if (!get_bytes(env, argv[2], &seed, &sl)) return NULL;
    if (sl != 16) return throw_msg(env, "seed must be 16 bytes");
    uint64_t n = 0;
    if (p_lfhe_ciphertext_size(c, bits, &n) != LFHE_OK) return throw_msg(env, "unsupported bit width");
    uint8_t* tmp = (uint8_t*)malloc(n);
    if (!tmp) return throw_msg(env, "out of memory");
    const int32_t rc = p_lfhe_rand(c, bits, seed, rbits, tmp, &n);
    napi_value out = rc == LFHE_OK ? make_buffer(env, tmp, n) : fail(env, c, rc, "lfhe_rand");
    free(tmp);
    return out;
}

/* importClientKeys(ctx, lweSk: BigUint64Array-backed Buffer, glweSk) ; keygenServer(ctx) ; setPkeKey(ctx, sk) */
static napi_value js_import_client(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t *a, *b; size_t al, bl;
    if (!get_bytes(env, argv[1], &a, &al) || !get_bytes(env, argv[2], &b, &bl)) return NULL;
    if (al % 8 || bl != 2048 * 8 || ((uintptr_t)a & 7) || ((uintptr_t)b & 7)) return throw_msg(env, "keys are arrays of 64-bit words");
    const int32_t rc = p_lfhe_client_keys_import_raw(c, (const uint64_t*)a, (const uint64_t*)b);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_client_keys_import_raw");
    return NULL;
}

static napi_value js_keygen_server(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const int32_t rc = p_lfhe_keygen_server(c);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_keygen_server");
    return NULL;
}

static napi_value js_set_pke_key(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t* a; size_t al;
    if (!get_bytes(env, argv[1], &a, &al)) return NULL;
    if (al != 2048 * 8 || ((uintptr_t)a & 7)) return throw_msg(env, "the key is 2048 64-bit words");
    const int32_t rc = p_lfhe_pke_key_set(c, (const uint64_t*)a);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_pke_key_set");
    return NULL;
}

static napi_value js_cast_from_pke(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t* ct; size_t n;
    if (!get_bytes(env, argv[1], &ct, &n)) return NULL;
    uint64_t on = n;
    uint8_t* tmp = (uint8_t*)malloc(n ? n : 1);
    if (!tmp) return throw_msg(env, "out of memory");
    const int32_t rc = p_lfhe_cast_from_pke(c, ct, n, tmp, &on);
    napi_value out = rc == LFHE_OK ? make_buffer(env, tmp, on) : fail(env, c, rc, "lfhe_cast_from_pke");
    free(tmp);
    return out;
}

/* expandCompactList(ctx, words (mask[2048] then bodies, 64-bit LE), packed, widths (Uint32Array bytes)) -> Buffer */
static napi_value js_expand_compact_list(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t *words, *widths; size_t wl, tl;
    if (!get_bytes(env, argv[1], &words, &wl) || !get_bytes(env, argv[3], &widths, &tl)) return NULL;
    bool packed = false;
    if (napi_get_value_bool(env, argv[2], &packed) != napi_ok) return throw_msg(env, "packed must be a boolean");
    if (wl <= 2048 * 8 || wl % 8 || ((uintptr_t)words & 7)) return throw_msg(env, "the list is 2048 mask words and one body word per ciphertext");
    if (!tl || tl % 4 || ((uintptr_t)widths & 3)) return throw_msg(env, "widths is an array of 32-bit integers");
    const uint64_t* w64 = (const uint64_t*)words;
    const uint64_t n_lwe = wl / 8 - 2048;
    uint64_t on = 0;
    int32_t rc = p_lfhe_expand_compact_list(c, w64, w64 + 2048, n_lwe, packed ? 1 : 0, (const uint32_t*)widths,
                                            (uint32_t)(tl / 4), NULL, &on);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_expand_compact_list");
    uint8_t* tmp = (uint8_t*)malloc(on ? on : 1);
    if (!tmp) return throw_msg(env, "out of memory");
    rc = p_lfhe_expand_compact_list(c, w64, w64 + 2048, n_lwe, packed ? 1 : 0, (const uint32_t*)widths, (uint32_t)(tl / 4),
                                    tmp, &on);
    napi_value out = rc == LFHE_OK ? make_buffer(env, tmp, on) : fail(env, c, rc, "lfhe_expand_compact_list");
    free(tmp);
    return out;
}

static napi_value js_compression_keygen(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const int32_t rc = p_lfhe_compression_keygen(c);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_compression_keygen");
    return NULL;
}

/* compress / decompress: size query first (out = NULL), then the call proper */
typedef int32_t (*recode_fn)(lfhe_ctx*, const uint8_t*, uint64_t, uint8_t*, uint64_t*);
static napi_value blob_call(napi_env env, napi_callback_info info, recode_fn fn, const char* what) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t* in; size_t n;
    if (!get_bytes(env, argv[1], &in, &n)) return NULL;
    uint64_t on = 0;
    int32_t rc = fn(c, in, n, NULL, &on);
    if (rc != LFHE_OK) return fail(env, c, rc, what);
    uint8_t* tmp = (uint8_t*)malloc(on ? on : 1);
    if (!tmp) return throw_msg(env, "out of memory");
    rc = fn(c, in, n, tmp, &on);
    napi_value out = rc == LFHE_OK ? make_buffer(env, tmp, on) : fail(env, c, rc, what);
    free(tmp);
    return out;
}
static napi_value js_compress(napi_env env, napi_callback_info info) {
    NEED_LIB();
    return blob_call(env, info, p_lfhe_compress, "lfhe_compress");
}
static napi_value js_decompress(napi_env env, napi_callback_info info) {
    NEED_LIB();
    return blob_call(env, info, p_lfhe_decompress, "lfhe_decompress");
}

static napi_value js_pbs_count(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    uint64_t n = 0;
    p_lfhe_pbs_count(c, &n);
    napi_value v;
    napi_create_double(env, (double)n, &v);
    return v;
}

/* ---- ciphertexts resident on the GPU, addressed by 32-byte handles (the executor events' bytes32) ---- */
static int get_handle(napi_env env, napi_value v, const uint8_t** h) {
    size_t n = 0;
    if (!get_bytes(env, v, h, &n)) return 0;
    if (!*h || n != 32) { throw_msg(env, "a handle is 32 bytes"); return 0; }
    return 1;
}

/* storePut(ctx, handle, ciphertext) */
static napi_value js_store_put(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t *h, *ct; size_t n;
    if (!get_handle(env, argv[1], &h) || !get_bytes(env, argv[2], &ct, &n)) return NULL;
    const int32_t rc = p_lfhe_store_put(c, h, ct, n);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_store_put");
    napi_value u;
    napi_get_undefined(env, &u);
    return u;
}

/* storeGet(ctx, handle) -> Buffer */
static napi_value js_store_get(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t* h;
    if (!get_handle(env, argv[1], &h)) return NULL;
    uint64_t n = 0;
    int32_t rc = p_lfhe_store_get(c, h, NULL, &n);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_store_get");
    uint8_t* tmp = (uint8_t*)malloc(n ? n : 1);
    if (!tmp) return throw_msg(env, "out of memory");
    rc = p_lfhe_store_get(c, h, tmp, &n);
    napi_value out = rc == LFHE_OK ? make_buffer(env, tmp, n) : fail(env, c, rc, "lfhe_store_get");
    free(tmp);
    return out;
}

/* storeDrop(ctx, handle) */
static napi_value js_store_drop(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t* h;
    if (!get_handle(env, argv[1], &h)) return NULL;
    const int32_t rc = p_lfhe_store_drop(c, h);
    if (rc != LFHE_OK) return fail(env, c, rc, "lfhe_store_drop");
    napi_value u;
    napi_get_undefined(env, &u);
    return u;
}

/* storeStat(ctx, handle|null) -> { entries, deviceBytes, bitWidth (0: no such handle) } */
static napi_value js_store_stat(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    const uint8_t* h = NULL; size_t hn = 0;
    if (!get_bytes(env, argv[1], &h, &hn)) return NULL;
    if (h && hn != 32) return throw_msg(env, "a handle is 32 bytes");
    uint64_t entries = 0, bytes = 0;
    uint32_t width = 0;
    const int32_t rc = p_lfhe_store_stat(c, h, &entries, &bytes, &width);
    if (rc != LFHE_OK && !(h && rc == LFHE_ERR_ARG)) return fail(env, c, rc, "lfhe_store_stat");
    if (rc != LFHE_OK) width = 0;
    napi_value o, v;
    napi_create_object(env, &o);
    napi_create_double(env, (double)entries, &v); napi_set_named_property(env, o, "entries", v);
    napi_create_double(env, (double)bytes, &v); napi_set_named_property(env, o, "deviceBytes", v);
    napi_create_uint32(env, width, &v); napi_set_named_property(env, o, "bitWidth", v);
    return o;
}

/* evaluateBatchHandles(ctx, op, bits, count, lhsHandles, rhs|null, rhsIsScalar, condHandles|null, outHandles):
 * count x 32 bytes each (a clear right operand: count x 32 bytes of little-endian values); results stay on the GPU */
static napi_value js_evaluate_batch_handles(napi_env env, napi_callback_info info) {
    NEED_LIB();
    napi_value argv[9];
    if (!get_args(env, info, 9, argv)) return NULL;
    lfhe_ctx* c = get_ctx(env, argv[0]);
    if (!c) return NULL;
    uint32_t op, bits, count;
    napi_get_value_uint32(env, argv[1], &op);
    napi_get_value_uint32(env, argv[2], &bits);
    napi_get_value_uint32(env, argv[3], &count);
    const uint8_t *l, *r, *cd, *o; size_t ll, rl, cl, ol;
    if (!get_bytes(env, argv[4], &l, &ll) || !get_bytes(env, argv[5], &r, &rl) || !get_bytes(env, argv[7], &cd, &cl) ||
        !get_bytes(env, argv[8], &o, &ol))
        return NULL;
    bool scalar = false;
    napi_get_value_bool(env, argv[6], &scalar);
    const size_t want = (size_t)count * 32;
    if (ll != want || ol != want || (r && rl != want) || (cd && cl != want))
        return throw_msg(env, "handle arrays hold `count` entries of 32 bytes each");
    const int32_t rc = p_lfhe_evaluate_batch_handles(c, op, bits, count, l, r, scalar, cd, o);
    if (rc != LFHE_OK) return fail(env, c, rc, "Evaluation");
    napi_value u;
    napi_get_undefined(env, &u);
    return u;
}

/* Every synchronous export enters the library under the same mutex as the async workers. */
typedef struct { const char* name; napi_callback fn; int locked; } export_entry;

static napi_value locked_call(napi_env env, napi_callback_info info) {
    void* data = NULL;
    size_t argc = 0;
    if (napi_get_cb_info(env, info, &argc, NULL, NULL, &data) != napi_ok || !data) return throw_msg(env, "bad call");
    const export_entry* e = (const export_entry*)data;
    pthread_mutex_lock(&g_lib_mutex);
    napi_value v = e->fn(env, info);
    pthread_mutex_unlock(&g_lib_mutex);
    return v;
}

static napi_value init(napi_env env, napi_value exports) {
    static const export_entry tab[] = {
        {"open", js_open, 1}, {"createContext", js_create, 1}, {"destroy", js_destroy, 1}, {"keygen", js_keygen, 1},
        {"health", js_health, 1}, {"publicKey", js_public_key, 1}, {"keysExport", js_keys_export, 1},
        {"keysImport", js_keys_import, 1}, {"encrypt", js_encrypt, 1}, {"encryptPublic", js_encrypt_public, 1},
        {"decrypt", js_decrypt, 1}, {"opFromName", js_op_from_name, 0}, {"evaluate", js_evaluate, 1},
        {"evaluateBatch", js_evaluate_batch, 1}, {"evaluateAsync", js_evaluate_async, 0},
        {"evaluateBatchAsync", js_evaluate_batch_async, 0}, {"pbsCount", js_pbs_count, 1},
        {"rand", js_rand, 1}, {"importClientKeys", js_import_client, 1}, {"keygenServer", js_keygen_server, 1},
        {"setPkeKey", js_set_pke_key, 1}, {"castFromPke", js_cast_from_pke, 1},
        {"compressionKeygen", js_compression_keygen, 1}, {"compress", js_compress, 1}, {"decompress", js_decompress, 1},
        {"expandCompactList", js_expand_compact_list, 1}, {"storePut", js_store_put, 1}, {"storeGet", js_store_get, 1},
        {"storeDrop", js_store_drop, 1}, {"storeStat", js_store_stat, 1},
        {"evaluateBatchHandles", js_evaluate_batch_handles, 1}};
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; ++i) {
        napi_value f;
        if (tab[i].locked) napi_create_function(env, tab[i].name, NAPI_AUTO_LENGTH, locked_call, (void*)&tab[i], &f);
        else napi_create_function(env, tab[i].name, NAPI_AUTO_LENGTH, tab[i].fn, NULL, &f);
        napi_set_named_property(env, exports, tab[i].name, f);
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
