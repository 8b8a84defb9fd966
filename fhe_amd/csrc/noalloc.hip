// lfhe_pbs_batch_device_noalloc: the bootstrap batch of lfhe_pbs_batch_device for callers that
// cannot allocate (a stream under HIP graph capture, a tight serving loop, a harness that forbids
// hipMalloc).  Same kernels, same launches, same bits; the intermediates -- key-switched small
// ciphertexts, key-switch digits, partial products of the matrix-core key switch -- live in one
// file-scope device array instead of the engine's grow-only hipMalloc buffers.
//
// The array exists once per GPU the library runs on (about 25 MiB each, held until the process
// ends).  In host code its NAME is only a stand-in: the device address is asked for once per device
// with hipGetSymbolAddress and everything below uses that pointer.  Every call on a GPU uses that
// GPU's one array.  Calls from different contexts are ordered on the device by the library: the
// entry point remembers the stream that used the array last and, when another stream comes, makes
// it wait for an event recorded behind the last user's work (one event per device, made at set-up;
// nothing is allocated per call); host threads are serialised by a mutex for the length of the
// enqueue.  What still races is a call that overlaps, on the device, with work the CALLER queued
// on the same buffers -- the usual rule for in/out pointers.  Nothing in the array has to start at
// zero: the digit kernels write every word of the rows they hand on (padding rows included) and the
// split key switch clears its output itself.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <stdexcept>

#include "ctx.h"

namespace lfhe {

constexpr size_t NA_MAX_CT = LFHE_NOALLOC_MAX_COUNT;   // ciphertexts per call
constexpr size_t NA_MAX_LWE = 1024;                    // small LWE dimension the regions are sized for
constexpr size_t NA_KS_ROWS = 256;                     // ciphertexts per matrix-core key-switch launch
constexpr size_t NA_MAX_KS_LEVELS = 4;

constexpr size_t na_align(size_t b) { return (b + 255) / 256 * 256; }
// region sizes in bytes
constexpr size_t NA_SMALL = na_align(NA_MAX_CT * (NA_MAX_LWE + 1) * sizeof(u64));
constexpr size_t NA_KS_DIG = na_align(NA_MAX_CT * (size_t)N * sizeof(u32));                 // VALU key switch, whole batch
constexpr size_t NA_KSM_DIG = na_align(NA_KS_ROWS * (size_t)N * NA_MAX_KS_LEVELS);          // int8 digits of one launch
constexpr size_t NA_KSM_PROD = na_align(NA_KS_ROWS * 8 * ((NA_MAX_LWE + 1 + 15) / 16 * 16) * sizeof(int));
constexpr size_t NA_TOTAL = NA_SMALL + NA_KS_DIG + NA_KSM_DIG + NA_KSM_PROD;

static __device__ __align__(256) unsigned char scratch[NA_TOTAL];

// device address of `scratch` on each device, asked for once per device with that device current
static unsigned char* g_scratch_ptr[64];
static std::atomic<uint64_t> g_scratch_known{0};
static std::mutex g_scratch_mutex;
// who used the array last on each device, and the event that orders the next user behind it
static hipStream_t g_last_stream[64];
static hipEvent_t g_order_event[64];
static std::mutex g_use_mutex[64];

void noalloc_setup(int device) {
    if (device < 0 || device >= 64) throw std::runtime_error("device index out of range for the scratch table");
    if (g_scratch_known.load(std::memory_order_acquire) >> device & 1) return;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_known.load(std::memory_order_relaxed) >> device & 1) return;
    hip_check(hipSetDevice(device), "hipSetDevice");
    void* p = nullptr;
    hip_check(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)");
    size_t bytes = 0;
    hip_check(hipGetSymbolSize(&bytes, HIP_SYMBOL(scratch)), "hipGetSymbolSize(scratch)");
    if (!p || bytes != NA_TOTAL) throw std::runtime_error("scratch symbol not resolved");
    hip_check(hipEventCreateWithFlags(&g_order_event[device], hipEventDisableTiming), "hipEventCreateWithFlags");
    g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    g_scratch_known.fetch_or(1ull << device, std::memory_order_release);
}

// An engine that goes away takes its stream with it: forget it as the array's last user (the engine has
// waited for the stream by then, so nothing is left to order against).
void noalloc_release(int device, hipStream_t st) {
    if (device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lock(g_use_mutex[device]);
    if (g_last_stream[device] == st) g_last_stream[device] = nullptr;
}

static PbsScratch noalloc_regions(int device) {
    if (!(g_scratch_known.load(std::memory_order_acquire) >> device & 1))
        throw std::logic_error("scratch not set up on this device");
    unsigned char* base = g_scratch_ptr[device];
    PbsScratch s;
    s.small = reinterpret_cast<u64*>(base);
    s.small_words = NA_SMALL / sizeof(u64);
    s.ks_dig = reinterpret_cast<u32*>(base + NA_SMALL);
    s.ks_dig_words = NA_KS_DIG / sizeof(u32);
    s.ksm_dig = reinterpret_cast<int8_t*>(base + NA_SMALL + NA_KS_DIG);
    s.ksm_dig_bytes = NA_KSM_DIG;
    s.ksm_prod = reinterpret_cast<int*>(base + NA_SMALL + NA_KS_DIG + NA_KSM_DIG);
    s.ksm_prod_bytes = NA_KSM_PROD;
    return s;
}

}  // namespace lfhe

using namespace lfhe;

extern "C" int32_t lfhe_pbs_batch_device_noalloc(lfhe_ctx* ctx, uint64_t count, const void* d_lwe_in,
                                                 const void* d_lut_bodies, const void* d_lut_index, void* d_lwe_out) {
    return guarded(ctx, [&]() -> int32_t {
        if (count && (!d_lwe_in || !d_lut_bodies || !d_lwe_out)) throw std::invalid_argument("null device pointer");
        if (count > NA_MAX_CT)
            throw std::invalid_argument("lfhe_pbs_batch_device_noalloc takes at most 1024 ciphertexts per call "
                                        "(LFHE_NOALLOC_MAX_COUNT): split the batch or use lfhe_pbs_batch_device");
        if (ctx->prm.lwe_dim > NA_MAX_LWE || ctx->prm.ks_level > NA_MAX_KS_LEVELS)
            throw std::invalid_argument("lfhe_pbs_batch_device_noalloc: parameter set beyond the scratch array "
                                        "(LWE dimension <= 1024, key-switch levels <= 4)");
        if (!ctx->eng->has_keys()) throw std::logic_error("server keys not loaded: call lfhe_keygen or an import first");
        const int dev = ctx->eng->device();
        noalloc_setup(dev);   // first call on this device: resolves the symbol, makes the event; later calls: one atomic load
        const PbsScratch s = noalloc_regions(dev);
        std::lock_guard<std::mutex> use(g_use_mutex[dev]);
        const hipStream_t st = ctx->eng->stream(), prev = g_last_stream[dev];
        if (prev && prev != st) {
            // another context's stream used the array last: everything queued there so far comes first
            hip_check(hipSetDevice(dev), "hipSetDevice");
            hip_check(hipEventRecord(g_order_event[dev], prev), "hipEventRecord");
            hip_check(hipStreamWaitEvent(st, g_order_event[dev], 0), "hipStreamWaitEvent");
        }
        g_last_stream[dev] = st;
        ctx->eng->pbs_batch_on(s, static_cast<const u64*>(d_lwe_in), static_cast<const u64*>(d_lut_bodies),
                               static_cast<const u32*>(d_lut_index), static_cast<u64*>(d_lwe_out), count);
        ctx->pbs_count += count;
        return LFHE_OK;
    });
}
