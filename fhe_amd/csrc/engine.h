// Device-side engine: transform tables, key material in HBM, kernel launchers and the batched
// bootstrap pipeline (key switch -> modulus switch + blind rotation + sample extraction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "ms_noise.h"
#include "pbs_phases.h"
#include "radix_program.h"

namespace lfhe {

// Where ciphertext number ct of a launch lives, in blocks of N + 1 words: contiguous (idx == null)
// or, for a radix program, block idx[ct % ops] of instance ct / ops in an arena.
struct BlockMap {
    const u32* idx = nullptr;
    u32 ops = 0;
    size_t inst_stride = 0;
    __host__ __device__ size_t block(size_t ct) const {
        return idx ? (ct / ops) * inst_stride + idx[ct % ops] : ct;
    }
};

struct EngineParams {
    uint32_t lwe_dim = 918;
    uint32_t glwe_dim = 1;
    uint32_t poly_size = 2048;
    uint32_t pbs_base_log = 23, pbs_level = 1;
    uint32_t ks_base_log = 4, ks_level = 4;
    uint32_t message_modulus = 4, carry_modulus = 4;
    uint32_t lwe_noise_log2 = 45, glwe_noise_log2 = 17;
    // modulus-switch noise reduction (ms_noise.h); ms_zeros_count = 0 switches it off
    uint32_t ms_zeros_count = 1449;
    uint32_t ms_bound_log2 = 58;
    double ms_r_sigma_factor = 13.179852282053789;
    double ms_input_variance = 2.63039184094559e-07;
    // ring modulus of the bootstrapping key and the accumulator: 0 = 2^64 (two-prime transform and
    // lift), QP_U = the NTT prime itself (single transform, modulus switch on the way in and out)
    uint64_t bsk_modulus = 0;
    // key bits per blind-rotation step: 1, or 2: the key then holds four GGSWs per pair of bits and a
    // step is one external product for two mask elements (over either ring modulus)
    uint32_t bsk_grouping = 1;
};

// kernel launchers (kernels.hip)
void launch_ntt_forward(const Tables& T, const u64* in, double* out, size_t polys, int mode, hipStream_t st);
void launch_ntt_inverse(const Tables& T, const double* in, u64* out, size_t polys, hipStream_t st);
void launch_external_product(const Tables& T, const double* bsk, const u32* idx, const u64* glwe, u64* out,
                             size_t count, hipStream_t st);
// Lift guard of the grouped key over 2^64: per-ciphertext marks of one launch (`capacity` of them), the
// bound on the lift's quotient from which a coefficient counts as near the window's edge, and the
// running count of ciphertexts that were bootstrapped again on the slow path.
struct G64Guard {
    int* fault = nullptr;
    size_t capacity = 0;
    double k_guard = 0;
    unsigned long long* fallbacks = nullptr;
    // two CUs per ciphertext for launches of at most split_below ciphertexts (kernels.hip: k_blind_rotate_g64_split):
    // hand-over buffers and step counters in device memory, how often a workgroup polls before it gives up
    double* xchg = nullptr;
    unsigned* xchg_flags = nullptr;
    size_t split_below = 0;
    unsigned poll_limit = 1u << 18;
};
size_t g64_split_payload_bytes(size_t cts);
size_t g64_split_flag_bytes(size_t cts);
void launch_blind_rotate(const Tables& T, const double* bsk, const u64* lwe_small, int n, const u64* luts,
                         const u32* lut_index, u64* glwe_out, u64* lwe_out, size_t count,
                         const BlockMap& omap, int grouping, int pair_above, int wide_below, const G64Guard& guard, hipStream_t st);
void launch_bskq_prepare(const Tables& T, const u64* in, double* out, size_t polys, hipStream_t st);
// three key bits per step: groups of eight transformed GGSWs (src) -> the factored image the kernel streams (dst)
void launch_bsk_factor3(const Tables& T, const double* src, double* dst, size_t groups, int* degenerate, int primes, hipStream_t st);
// grouped key image -> factored form, in place (row_words: BSKQ_ROW or BSK_ROW); *degenerate (device) is set
// when a word has none
void launch_bsk_factor(const Tables& T, double* img, size_t groups, size_t row_words, int* degenerate, hipStream_t st);
void launch_external_product_q(const Tables& T, const double* bsk, const u32* idx, const u64* glwe, u64* out,
                               size_t count, double inv_g, hipStream_t st);
void launch_blind_rotate_q(const Tables& T, const double* bsk, const u64* lwe_small, int n, const u64* luts,
                           const u32* lut_index, u64* glwe_out, u64* lwe_out, size_t count,
                           const BlockMap& omap, double inv_g, int wide_below, int lockstep_above, int grouping,
                           int pair_above, hipStream_t st);
void launch_lin_batch(u64* arena, const LinOp* ops, int nops, size_t instances, size_t inst_stride_blocks, u64 delta,
                      hipStream_t st);
void launch_blocks_move(u64* arena, size_t instances, size_t inst_stride_blocks, const u32* idx, u32 first, u32 nblocks,
                        u64* flat, size_t ct_words, u32 header_words, bool to_arena, hipStream_t st);
void launch_blocks_move_ptrs(u64* arena, size_t instances, size_t inst_stride_blocks, const u32* idx, u32 first,
                             u32 nblocks, u64* const* ptrs, bool to_arena, hipStream_t st);
void launch_ksk_colsum(const u64* ksk, int rows, int cols, int half_base, u64* colsum, hipStream_t st);
// dig_scratch: keyswitch_scratch_words(count, big) 32-bit words of device memory
void launch_keyswitch(const u64* ksk, const u64* colsum, const u64* in, u32* dig_scratch, u64* out, size_t count,
                      int big, int n, int base_log, int levels, const BlockMap& imap, hipStream_t st);
size_t keyswitch_scratch_words(size_t count, int big);
size_t ks_valu_max_count();   // ciphertexts one launch_keyswitch call can take (grid limit)
// the same key switch as an int8 matrix product on the matrix cores (ks_mfma.hip), for large batches
bool ksm_supported(int big, int levels, int n);
size_t ksm_limb_bytes(int big, int levels, int n);            // key image: eight signed bytes per key word
size_t ksm_digit_bytes(size_t count, int big, int levels);    // scratch: signed digits of a batch
size_t ksm_product_bytes(size_t count, int n);                // scratch: the int32 partial products
void launch_ksm_key_limbs(const u64* ksk, int8_t* limbs, int big, int levels, int n, hipStream_t st);
void launch_keyswitch_mfma(const int8_t* limbs, const u64* in, int8_t* dig, int* prod, u64* out, size_t count, int big,
                           int n, int base_log, int levels, const BlockMap& imap, hipStream_t st);
void launch_ms_noise_reduce(u64* lwe_small, int n, const u64* zeros, const MsParams& m, int* chosen, size_t count,
                            hipStream_t st);
void launch_lwe_linear(const u64* a, const u64* b, u64* out, size_t count, int dim1, u64 ca, u64 cb, u64 add_body,
                       hipStream_t st);

// ciphertext compression (ks_mfma.hip)
void launch_pack(const int8_t* limbs, const u64* in, int8_t* dig, int* prod, uint16_t* out, size_t count, int big, int kc,
                 int nc, int base_log, int levels, int storage_log, u64 scale, const BlockMap& imap, hipStream_t st);
void launch_pack_key_limbs(const u64* pksk, int8_t* limbs, int big, int levels, int kc, int nc, hipStream_t st);
void launch_unpack_extract(const uint16_t* glwe, u64* small, size_t count, int kc, int nc, int storage_log, hipStream_t st);
size_t ksm_rows_pad(size_t count);

// Shape of the compression scheme: packing key switch into GLWE ciphertexts of dimension kc and
// polynomial size nc (nc blocks per GLWE), stored modulo 2^storage_log; decompression bootstraps
// under a key for the flattened GLWE secret (kc * nc bits).  The numbers are those of tfhe-rs'
// COMP_PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128.
struct CompShape {
    int kc = 4, nc = 256;
    int ks_base_log = 4, ks_level = 3;
    int storage_log = 12;
    uint32_t noise_log2 = 43;
    int small_dim() const { return kc * nc; }
    int glwe_words() const { return (kc + 1) * nc; }
    size_t pksk_words() const { return (size_t)N * ks_level * glwe_words(); }
};

// Caller-provided scratch of one bootstrap batch (all device memory): the key-switched small
// ciphertexts, the packed digits of the VALU key switch, the digits and partial products of the
// matrix-core key switch.  A region that is too small for the whole batch makes the stage run in
// several launches; one that cannot hold a single launch is an error.  Nothing is allocated.
struct PbsScratch {
    u64* small = nullptr;      size_t small_words = 0;
    u32* ks_dig = nullptr;     size_t ks_dig_words = 0;
    int8_t* ksm_dig = nullptr; size_t ksm_dig_bytes = 0;
    int* ksm_prod = nullptr;   size_t ksm_prod_bytes = 0;
};

class Engine {
public:
    // throws std::runtime_error on unsupported parameters or HIP failure
    Engine(int device, const EngineParams& p);
    ~Engine();
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    const EngineParams& params() const { return prm_; }
    int device() const { return device_; }
    hipStream_t stream() const { return stream_; }
    const Tables& tables() const { return T_; }
    bool has_keys() const { return d_bsk_ && d_ksk_; }

    // GGSWs in the bootstrapping key
    size_t ggsw_count_for(uint32_t n) const {
        const size_t b = prm_.bsk_grouping;
        return b >= 2 ? ((size_t)1 << b) * (((size_t)n + b - 1) / b) : n;
    }
    size_t bsk_ggsw_count() const { return ggsw_count_for(prm_.lwe_dim); }
    size_t bsk_raw_words() const { return bsk_ggsw_count() * 4 * N; }
    size_t ksk_words() const { return (size_t)N * prm_.ks_level * (prm_.lwe_dim + 1); }
    bool single_modulus() const { return prm_.bsk_modulus != 0; }
    // gadget of the single-modulus key: round(q' / 2^base_log); digits are rint(c / gadget)
    u64 qp_gadget() const { return (QP_U + (1ull << (prm_.pbs_base_log - 1))) >> prm_.pbs_base_log; }
    size_t bsk_dev_bytes() const { return bsk_ggsw_count() * (single_modulus() ? BSKQ_ROW : BSK_ROW) * sizeof(double); }
    size_t ms_zeros_words() const { return (size_t)prm_.ms_zeros_count * (prm_.lwe_dim + 1); }
    // device image of the key-switching key: key words, n+1 column sums, one word holding the
    // number of encryptions of zero that follow (0 when the key set came without them), the zeros
    size_t ksk_dev_bytes() const { return (ksk_words() + prm_.lwe_dim + 1 + 1 + ms_zeros_words()) * sizeof(u64); }

    // key material: host coefficient-domain keys in, device images built here
    void load_bsk(const u64* host_bsk);   // [ggsw][row][poly][N]
    // [N][level][n+1]; ms_zeros [ms_zeros_count][n+1] or null
    void load_ksk(const u64* host_ksk, const u64* ms_zeros = nullptr);
    // or attach images that already sit in device memory (e.g. received by a broadcast);
    // the engine does not take ownership
    void attach_device_keys(double* bsk_dev, u64* ksk_dev_image);
    // a second key-switching key, from the secret key of public-key encryptions to the small key
    // (same decomposition); used only by cast_batch, local to this engine
    void load_cast_ksk(const u64* host_ksk);
    bool has_cast_key() const { return d_cast_ksk_ != nullptr; }
    // count blocks under the public-key secret -> blocks under the compute key, values kept
    void cast_batch(const u64* lwe_in, const u64* identity_lut, u64* lwe_out, size_t count);
    double* bsk_dev() const { return d_bsk_; }
    u64* ksk_dev() const { return d_ksk_; }
    bool has_ms_zeros() const { return ms_.zcount != 0; }

    // ---- device-pointer operations, asynchronous on stream()
    // count bootstraps: lwe_in [count][N+1], luts [nluts][N] body polynomials, lut_index [count] or null
    void pbs_batch(const u64* lwe_in, const u64* luts, const u32* lut_index, u64* lwe_out, size_t count);
    // the same pipeline (same kernels, same bits) on the caller's scratch; allocates nothing
    void pbs_batch_on(const PbsScratch& scratch, const u64* lwe_in, const u64* luts, const u32* lut_index, u64* lwe_out,
                      size_t count);
    // ext: scratch to use instead of the engine's own grow-only buffers
    void keyswitch(const u64* lwe_in, u64* lwe_small_out, size_t count, const BlockMap& imap = BlockMap(),
                   bool with_cast_key = false, const PbsScratch* ext = nullptr);
    // in place on [count][n+1] small ciphertexts; a no-op without encryptions of zero;
    // chosen [count] (device, may be null) receives the candidate taken (-1: none added)
    void ms_noise_reduce(u64* lwe_small, size_t count, int* chosen = nullptr);
    void blind_rotate(const u64* lwe_small, const u64* luts, const u32* lut_index, u64* glwe_out, u64* lwe_out,
                      size_t count, const BlockMap& omap = BlockMap());
    // one bootstrap layer of a radix program: ops bootstraps per instance, inputs and outputs are
    // blocks of the arena, lut_of_op[ops] selects each bootstrap's table
    void pbs_layer(u64* arena, size_t instances, size_t inst_stride_blocks, const u32* in_idx, const u32* out_idx,
                   const u32* lut_of_op, u32 ops, const u64* luts);
    void lin_layer(u64* arena, size_t instances, size_t inst_stride_blocks, const LinOp* ops, int nops, u64 delta);
    // the same blind rotation under another bootstrapping key image of this engine's flavour, for
    // small ciphertexts of dimension n
    // factored: whether that image holds the grouped key in factored form (-1: as the main key's)
    void blind_rotate_with(const double* bsk, uint32_t n, const u64* lwe_small, const u64* luts, const u32* lut_index,
                           u64* glwe_out, u64* lwe_out, size_t count, const BlockMap& omap = BlockMap(), int factored = -1);
    // a grouped key is held in factored form (two products per GGSW entry
    // instead of three, pbs_phases.h) unless one of its words has none
    bool bsk_factored() const { return bsk_factored_; }
    // ---- compression: host keys in (packing key [N][level][(kc+1) nc]; bootstrapping key for the
    // flattened compression secret, same layout and flavour as load_bsk takes)
    void load_compression_keys(const CompShape& shape, const u64* host_pksk, const u64* host_decomp_bsk);
    bool has_compression_keys() const { return d_pack_limbs_ && d_decomp_bsk_; }
    const CompShape& comp_shape() const { return comp_; }
    size_t decomp_bsk_ggsw_count() const { return ggsw_count_for((uint32_t)comp_.small_dim()); }
    // count blocks (device, contiguous) -> ceil(count / nc) GLWEs of (kc+1) nc 16-bit words (device)
    void compress(const u64* blocks, size_t count, uint16_t* out, u64 scale);
    // stored GLWEs (device) -> count blocks under the compute key, table applied by the bootstrap
    void decompress(const uint16_t* glwe, size_t count, const u64* lut, u64* blocks_out);
    void external_product(const u32* ggsw_index, const u64* glwe, u64* out, size_t count);
    void ntt_forward(const u64* polys, double* out, size_t count);
    void ntt_inverse(const double* in, u64* out, size_t count);
    void sync();
    // ciphertexts whose bootstrap was redone on the slow, unconditionally exact path because a lift came
    // near the end of its window (grouped key over 2^64 only; 0 for every other key)
    uint64_t lift_fallbacks();

    // scratch owned by the engine (grow-only): the small-LWE intermediate of pbs_batch
    u64* small_scratch(size_t count);

private:
    int device_;
    EngineParams prm_;
    hipStream_t stream_ = nullptr;
    Tables T_{};
    double* d_tw_ = nullptr;
    double* d_bsk_ = nullptr;
    u64* d_ksk_ = nullptr;
    u64* d_cast_ksk_ = nullptr;   // key words followed by the n+1 column sums
    bool own_keys_ = false;
    u64* d_small_ = nullptr;
    size_t small_cap_ = 0;
    u32* d_ksdig_ = nullptr;   // packed key-switch digits of the launch in flight
    size_t ksdig_cap_ = 0;
    // matrix-core key switch: byte limbs of the key (built from d_ksk_, always owned), scratch for
    // one launch of at most ksm_cap_ ciphertexts, batch size from which it is used (0: never)
    int8_t* d_ksk_limbs_ = nullptr;
    int8_t* d_ksm_dig_ = nullptr;
    int* d_ksm_prod_ = nullptr;
    size_t ksm_cap_ = 0;
    int ksm_min_ = 128;
    void build_ksk_limbs();
    // compression: byte limbs of the packing key, bootstrapping key image for the compression secret
    CompShape comp_{};
    int8_t* d_pack_limbs_ = nullptr;
    double* d_decomp_bsk_ = nullptr;
    bool build_bsk_image(const u64* host_bsk, size_t ggsw_count, double* dst);   // true: factored form
    bool bsk_factored_ = false, decomp_factored_ = false;
    void publish_key_flags();   // the count word of the key-switching key image also says how the other image is held
    MsParams ms_{};
    G64Guard g64_guard_{};
    int cu_count_ = 256;
    int br_chunk_ = 0;      // ciphertexts per blind-rotation launch; 0: about four rounds of resident workgroups
    int wide_below_ = -1;   // launches of at most this many ciphertexts use 512 threads each
    int lockstep_above_ = 1024;   // grouped key: larger launches keep a workgroup barrier after every phase
    int pair_above_ = -2;   // grouped key: launches of more ciphertexts put two in a workgroup (-1: never, -2: automatic)
    int pair_threshold(size_t count) const;
    void set_ms_params(u32 zcount);
    u64* ms_zeros_dev() const { return d_ksk_ + ksk_words() + prm_.lwe_dim + 1 + 1; }
};

void hip_check(hipError_t e, const char* what);
// once per device, with any device current: resolves the device address of the scratch array behind
// lfhe_pbs_batch_device_noalloc (noalloc.hip)
void noalloc_setup(int device);
void noalloc_release(int device, hipStream_t st);
// once per device: the 512-thread blind-rotation kernels ask for more than 64 KiB of dynamic LDS
void blind_rotate_setup(int device);

}  // namespace lfhe
