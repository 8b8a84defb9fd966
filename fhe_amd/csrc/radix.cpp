// Integer operators on radix ciphertexts (euint8 ... euint256, ebool): serialisation, encryption,
// decryption and the executor that replays an operator's Program (radix_program.h) for a batch of
// independent instances, layer by layer, on the GPU.
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>

#include "ctx.h"
#include "radix_program.h"

using namespace lfhe;

#define HC(x) hip_check((x), #x)

namespace lfhe {
void launch_fill_trivial(u64* arena, size_t instances, size_t inst_stride_blocks, u32 first_block, u32 m,
                         const u64* bodies, hipStream_t st);
}

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
constexpr size_t BLK_WORDS = (size_t)N + 1;
constexpr size_t BLK_BYTES = BLK_WORDS * 8;

bool width_ok(uint32_t w) { return w == 1 || w == 8 || w == 16 || w == 32 || w == 64 || w == 128 || w == 160 || w == 256; }
uint32_t blocks_of(uint32_t w) { return w == 1 ? 1 : w / 2; }
uint64_t ct_bytes(uint32_t w) { return sizeof(CtHeader) + (uint64_t)blocks_of(w) * BLK_BYTES; }

uint32_t block_value(const uint8_t v[32], uint32_t i) {  // i-th base-4 digit
    const uint32_t bit = 2 * i;
    return (v[bit >> 3] >> (bit & 7)) & 3u;
}

CtHeader parse_header(const uint8_t* ct, uint32_t expect_width) {
    CtHeader h;
    std::memcpy(&h, ct, sizeof h);
    if (h.magic != CT_MAGIC || h.version != 1 || h.bit_width != expect_width || h.nblocks != blocks_of(expect_width))
        throw std::invalid_argument("malformed ciphertext or width mismatch");
    if (h.noise_level < 1 || h.noise_level > 2 || h.degree > 3) throw std::invalid_argument("ciphertext metadata out of range");
    return h;
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    DevBuf() = default;
    explicit DevBuf(size_t count) { if (count) HC(hipMalloc(&p, count * sizeof(T))); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p) { o.p = nullptr; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { if (p) (void)hipFree(p); p = o.p; o.p = nullptr; }
        return *this;
    }
};

// a Program with its descriptors and tables resident on the device
struct Compiled {
    Program prog;
    uint64_t last_use = 0;                 // for the bound on programs specialised on a clear operand
    std::vector<LinOp*> d_lin;            // per layer (null when empty)
    std::vector<u32*> d_in, d_out, d_lut;  // per layer
    u64* d_luts = nullptr;                 // [nluts][N]
    ~Compiled() {
        for (auto p : d_lin) (void)hipFree(p);
        for (auto p : d_in) (void)hipFree(p);
        for (auto p : d_out) (void)hipFree(p);
        for (auto p : d_lut) (void)hipFree(p);
        (void)hipFree(d_luts);
    }
};

}  // namespace

typedef std::array<uint8_t, 32> Scalar;

// ---- the context's ciphertext store: ciphertexts resident in device memory under 32-byte handles.
// Memory comes in slabs per ciphertext size (one hipMalloc serves many entries) and goes back to a
// free list when an entry is dropped or replaced; the slabs stay with the context.
typedef std::array<uint8_t, 32> Handle;
struct HandleHash {
    size_t operator()(const Handle& h) const {
        uint64_t a, b;
        std::memcpy(&a, h.data(), 8);
        std::memcpy(&b, h.data() + 13, 8);
        return (size_t)(a ^ (b * 0x9E3779B97F4A7C15ull));
    }
};
struct StoreEntry {
    u64* dev = nullptr;       // nblocks blocks of N + 1 words
    uint16_t width = 0;
    uint32_t nblocks = 0;
    uint8_t noise = 1, degree = 3;
};
struct lfhe_store {
    std::unordered_map<Handle, StoreEntry, HandleHash> map;
    std::map<uint32_t, std::vector<u64*>> free_list;   // by block count
    std::vector<u64*> slabs;
    uint64_t reserved = 0;                              // bytes of device memory held by the slabs
    u64* alloc(uint32_t nblocks) {
        auto& fl = free_list[nblocks];
        if (fl.empty()) {
            const size_t words = (size_t)nblocks * BLK_WORDS;
            const size_t per = std::max<size_t>(1, std::min<size_t>(256, ((size_t)64 << 20) / (words * 8)));
            u64* slab = nullptr;
            HC(hipMalloc(&slab, per * words * 8));
            slabs.push_back(slab);
            reserved += per * words * 8;
            for (size_t k = per; k-- > 0;) fl.push_back(slab + k * words);
        }
        u64* p = fl.back();
        fl.pop_back();
        return p;
    }
    void release(const StoreEntry& e) { if (e.dev) free_list[e.nblocks].push_back(e.dev); }
    ~lfhe_store() { for (u64* s : slabs) (void)hipFree(s); }
};
static Handle handle_at(const uint8_t* handles, uint64_t i) {
    Handle h;
    std::memcpy(h.data(), handles + i * 32, 32);
    return h;
}

struct lfhe_radix {
    lfhe_store store;
    std::map<std::tuple<uint32_t, uint32_t, int, int, int, bool, Scalar>, std::unique_ptr<Compiled>> cache;
    uint64_t clock = 0;          // use counter
    size_t specialised = 0;      // entries whose key holds a clear operand
    std::map<std::string, std::unique_ptr<Compiled>> fused;   // key: the bytes of the request's shape
};

lfhe_ctx::lfhe_ctx() = default;
lfhe_ctx::~lfhe_ctx() {
    if (eng) (void)hipSetDevice(eng->device());
    radix.reset();
}

namespace {

// How many instances of a program fit one arena: half of the free device memory (the bootstrap
// scratch of the largest layer needs room too), LFHE_ARENA_BYTES overrides the budget.  Larger
// batches are run in chunks of this many instances.
uint64_t arena_capacity(size_t inst_bytes) {
    size_t free_b = 0, total_b = 0;
    HC(hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = free_b / 2;
    if (const char* e = std::getenv("LFHE_ARENA_BYTES")) budget = std::strtoull(e, nullptr, 10);
    return lfhe_arena_chunk(budget, inst_bytes);
}

// descriptors and tables of a program to the device
void upload_program(lfhe_ctx* ctx, Compiled& C) {
    Engine& E = *ctx->eng;
    HC(hipSetDevice(E.device()));
    const Program& P = C.prog;
    std::vector<u64> bodies(P.luts.size() * (size_t)N);
    for (size_t i = 0; i < P.luts.size(); ++i)
        make_lut_body(ctx->space(), ctx->delta(), P.luts[i].data(), bodies.data() + i * N);
    if (!bodies.empty()) {
        HC(hipMalloc(&C.d_luts, bodies.size() * 8));
        HC(hipMemcpy(C.d_luts, bodies.data(), bodies.size() * 8, hipMemcpyHostToDevice));
    }
    for (const Layer& L : P.layers) {
        LinOp* dl = nullptr;
        if (!L.lin.empty()) {
            HC(hipMalloc(&dl, L.lin.size() * sizeof(LinOp)));
            HC(hipMemcpy(dl, L.lin.data(), L.lin.size() * sizeof(LinOp), hipMemcpyHostToDevice));
        }
        C.d_lin.push_back(dl);
        u32 *di = nullptr, *dout = nullptr, *dlut = nullptr;
        if (!L.pbs.empty()) {
            std::vector<u32> a, b, l;
            for (const PbsOp& q : L.pbs) { a.push_back(q.in); b.push_back(q.out); l.push_back(q.lut); }
            HC(hipMalloc(&di, a.size() * 4)); HC(hipMalloc(&dout, a.size() * 4)); HC(hipMalloc(&dlut, a.size() * 4));
            HC(hipMemcpy(di, a.data(), a.size() * 4, hipMemcpyHostToDevice));
            HC(hipMemcpy(dout, b.data(), a.size() * 4, hipMemcpyHostToDevice));
            HC(hipMemcpy(dlut, l.data(), a.size() * 4, hipMemcpyHostToDevice));
        }
        C.d_in.push_back(di); C.d_out.push_back(dout); C.d_lut.push_back(dlut);
    }
}

// programs specialised on a clear right operand: shift and rotation amounts, divisors
bool specialised_op(uint32_t op) {
    return op == LFHE_OP_SHL || op == LFHE_OP_SHR || op == LFHE_OP_ROTL || op == LFHE_OP_ROTR || op == LFHE_OP_DIV ||
           op == LFHE_OP_REM;
}
// operators whose schedule is specialised on a clear right operand only when the whole batch shares it
bool shared_scalar_op(uint32_t op) { return op == LFHE_OP_MUL || op == LFHE_OP_AND || op == LFHE_OP_OR || op == LFHE_OP_XOR; }
bool shift_op(uint32_t op) { return op == LFHE_OP_SHL || op == LFHE_OP_SHR || op == LFHE_OP_ROTL || op == LFHE_OP_ROTR; }

size_t program_cache_limit() {
    static const size_t limit = [] {
        const char* e = std::getenv("LFHE_PROGRAM_CACHE");
        const long v = e ? std::atol(e) : 0;
        return (size_t)(v > 0 ? v : 256);
    }();
    return limit;
}

Compiled& compiled_for(lfhe_ctx* ctx, uint32_t op, uint32_t width, int ln, int rn, int cn, const uint8_t* scalar) {
    if (!ctx->radix) ctx->radix.reset(new lfhe_radix());
    Scalar sc{};
    if (scalar) std::memcpy(sc.data(), scalar, 32);
    auto key = std::make_tuple(op, width, ln, rn, cn, scalar != nullptr, sc);
    lfhe_radix& R = *ctx->radix;
    auto it = R.cache.find(key);
    if (it != R.cache.end()) {
        it->second->last_use = ++R.clock;
        return *it->second;
    }
    // Programs built for a clear operand are kept per value (shift amounts, divisors, factors, masks):
    // a service that sees ever new values must not grow without bound, so the least recently used
    // one goes when there are more than program_cache_limit() of them.
    if (scalar && R.specialised >= program_cache_limit()) {
        auto oldest = R.cache.end();
        for (auto e = R.cache.begin(); e != R.cache.end(); ++e)
            if (std::get<5>(e->first) && (oldest == R.cache.end() || e->second->last_use < oldest->second->last_use)) oldest = e;
        if (oldest != R.cache.end()) {
            HC(hipSetDevice(ctx->eng->device()));
            HC(hipStreamSynchronize(ctx->eng->stream()));   // nothing in flight reads its tables
            R.cache.erase(oldest);
            --R.specialised;
        }
    }
    std::unique_ptr<Compiled> c(new Compiled());
    c->prog = build_program(op, width, ln, rn, cn, scalar);
    upload_program(ctx, *c);
    c->last_use = ++R.clock;
    Compiled& ref = *c;
    R.cache[key] = std::move(c);
    if (scalar) ++R.specialised;
    return ref;
}

// An operand array of n ciphertexts: the caller's host array (per ciphertext a header and its blocks), or n
// handles of 32 bytes into the context's device store.  Results likewise.
struct Src {
    const uint8_t* host = nullptr;
    const uint8_t* handles = nullptr;
    explicit operator bool() const { return host || handles; }
    Src at(uint64_t off, uint64_t stride) const { return Src{host ? host + off * stride : nullptr, handles ? handles + off * 32 : nullptr}; }
};
struct Dst {
    uint8_t* host = nullptr;
    const uint8_t* handles = nullptr;
    explicit operator bool() const { return host || handles; }
    Dst at(uint64_t off, uint64_t stride) const { return Dst{host ? host + off * stride : nullptr, handles ? handles + off * 32 : nullptr}; }
};

const StoreEntry& stored(lfhe_ctx* ctx, const uint8_t* handles, uint64_t i, uint32_t width) {
    if (!ctx->radix) throw std::invalid_argument("unknown handle: the store is empty");
    const auto it = ctx->radix->store.map.find(handle_at(handles, i));
    if (it == ctx->radix->store.map.end()) throw std::invalid_argument("unknown handle");
    if (it->second.width != width) throw std::invalid_argument("stored ciphertext has another width than the operation");
    return it->second;
}

// header (or store entry) check of every ciphertext of an operand array; returns the largest noise level
int check_operand(lfhe_ctx* ctx, const Src& a, uint64_t n, uint32_t width) {
    int noise = 1;
    if (a.handles) {
        for (uint64_t i = 0; i < n; ++i) noise = std::max<int>(noise, stored(ctx, a.handles, i, width).noise);
        return noise;
    }
    const uint64_t stride = ct_bytes(width);
    for (uint64_t i = 0; i < n; ++i) noise = std::max<int>(noise, parse_header(a.host + i * stride, width).noise_level);
    return noise;
}

void replay(lfhe_ctx* ctx, Compiled& C, u64* arena, uint64_t n);

void cast_one(lfhe_ctx* ctx, const uint8_t* in, uint64_t in_len, uint32_t to_width, uint8_t* out) {
    if (in_len < sizeof(CtHeader)) throw std::invalid_argument("ciphertext too short");
    CtHeader h;
    std::memcpy(&h, in, sizeof h);
    // the header is caller data: the block count that is copied comes from the width the length was
    // checked against, never from the header's own field
    if (h.magic != CT_MAGIC || h.version != 1 || !width_ok(h.bit_width) || h.nblocks != blocks_of(h.bit_width) ||
        in_len != ct_bytes(h.bit_width))
        throw std::invalid_argument("malformed ciphertext");
    if (h.noise_level < 1 || h.noise_level > 2 || h.degree > 3) throw std::invalid_argument("ciphertext metadata out of range");
    if (to_width == 1 && h.bit_width != 1) throw std::invalid_argument("cast to ebool: use ne(x, 0)");
    const uint32_t nb = blocks_of(to_width), have = std::min(nb, blocks_of(h.bit_width));
    CtHeader o{CT_MAGIC, 1, (uint16_t)to_width, nb, h.noise_level, h.degree, 0};
    std::memcpy(out, &o, sizeof o);
    std::memcpy(out + sizeof o, in + sizeof h, (size_t)have * BLK_BYTES);
    std::memset(out + sizeof o + (size_t)have * BLK_BYTES, 0, (size_t)(nb - have) * BLK_BYTES);  // trivial zeros
    (void)ctx;
}

// Operand and result arrays cross PCIe whole: one copy between the caller's array and a device
// staging buffer of the same layout, one kernel between the staging buffer and the arena blocks.
constexpr uint32_t CT_HEADER_WORDS = 2;

void upload_array(Engine& E, u64* staging, u64* arena, uint32_t nblocks_inst, const uint8_t* src, uint64_t ct_stride,
                  uint32_t first, uint32_t nb, uint64_t n) {
    HC(hipMemcpyAsync(staging, src, n * ct_stride, hipMemcpyHostToDevice, E.stream()));
    launch_blocks_move(arena, n, nblocks_inst, nullptr, first, nb, staging, ct_stride / 8, CT_HEADER_WORDS, true, E.stream());
    HC(hipGetLastError());
}

// blocks: device copy of the arena block numbers of the result's blocks
void download_array(Engine& E, u64* staging, u64* arena, uint32_t nblocks_inst, const u32* blocks, uint32_t nb,
                    uint8_t* dst, uint64_t ct_stride, uint64_t n) {
    launch_blocks_move(arena, n, nblocks_inst, blocks, 0, nb, staging, ct_stride / 8, CT_HEADER_WORDS, false, E.stream());
    HC(hipGetLastError());
    HC(hipMemcpyAsync(dst, staging, n * ct_stride, hipMemcpyDeviceToHost, E.stream()));
}

// device tables of per-instance pointers, alive until the call has synchronised
typedef std::vector<DevBuf<u64*>> PtrTables;

// an operand array into arena blocks first .. first + nb - 1 of every instance
void load_operand(lfhe_ctx* ctx, u64* staging, u64* arena, uint32_t nblocks_inst, const Src& a, uint32_t width,
                  uint32_t first, uint64_t n, PtrTables& keep) {
    Engine& E = *ctx->eng;
    const uint32_t nb = blocks_of(width);
    if (!a.handles) {
        upload_array(E, staging, arena, nblocks_inst, a.host, ct_bytes(width), first, nb, n);
        return;
    }
    std::vector<u64*> ptrs(n);
    for (uint64_t i = 0; i < n; ++i) ptrs[i] = stored(ctx, a.handles, i, width).dev;
    keep.emplace_back(n);
    HC(hipMemcpy(keep.back().p, ptrs.data(), n * sizeof(u64*), hipMemcpyHostToDevice));
    launch_blocks_move_ptrs(arena, n, nblocks_inst, nullptr, first, nb, keep.back().p, true, E.stream());
    HC(hipGetLastError());
}

// Result entries of a call that writes into the store: allocated up front, entered under their handles
// once the call has finished (an entry that already has the handle is replaced), given back if it fails.
struct FreshEntries {
    lfhe_store& S;
    std::vector<std::pair<Handle, StoreEntry>> items;
    bool committed = false;
    explicit FreshEntries(lfhe_store& s) : S(s) {}
    ~FreshEntries() { if (!committed) for (auto& it : items) S.release(it.second); }
    void commit() {
        for (auto& it : items) {
            auto f = S.map.find(it.first);
            if (f != S.map.end()) { S.release(f->second); f->second = it.second; }
            else S.map.emplace(it.first, it.second);
        }
        committed = true;
    }
};

// arena blocks `blocks` (device list of nb block numbers) of every instance out to a result array
void store_result(lfhe_ctx* ctx, u64* staging, u64* arena, uint32_t nblocks_inst, const u32* blocks, uint32_t nb,
                  const Dst& out, const CtHeader& h, uint64_t n, PtrTables& keep, FreshEntries& fresh) {
    Engine& E = *ctx->eng;
    if (!out.handles) {
        download_array(E, staging, arena, nblocks_inst, blocks, nb, out.host, ct_bytes(h.bit_width), n);
        return;
    }
    std::vector<u64*> ptrs(n);
    for (uint64_t i = 0; i < n; ++i) {
        StoreEntry e;
        e.dev = fresh.S.alloc(nb);
        e.width = h.bit_width; e.nblocks = nb; e.noise = h.noise_level; e.degree = h.degree;
        fresh.items.emplace_back(handle_at(out.handles, i), e);
        ptrs[i] = e.dev;
    }
    keep.emplace_back(n);
    HC(hipMemcpy(keep.back().p, ptrs.data(), n * sizeof(u64*), hipMemcpyHostToDevice));
    launch_blocks_move_ptrs(arena, n, nblocks_inst, blocks, 0, nb, keep.back().p, false, E.stream());
    HC(hipGetLastError());
}

void run_batch(lfhe_ctx* ctx, uint32_t op, uint32_t width, uint64_t n, const Src& lhs, const Src& rhs_src,
               bool rhs_scalar, const Src& cond, const Dst& out) {
    const uint8_t* rhs = rhs_src.host;    // clear operands are always the caller's (32 bytes each)
    if (rhs_scalar && rhs_src.handles) throw std::invalid_argument("a clear operand is not a handle");
    if (!width_ok(width)) throw std::invalid_argument("unsupported bit width");
    if (!n) return;
    if (!lhs || !out) throw std::invalid_argument("null operand");
    if (!ctx->eng->has_keys()) throw std::logic_error("server keys not loaded");
    const bool unary = op == LFHE_OP_NOT || op == LFHE_OP_NEG;
    if (!unary && !rhs_src) throw std::invalid_argument("missing right operand");
    if (op == LFHE_OP_SELECT && !cond) throw std::invalid_argument("select needs a condition");
    if (!ctx->radix) ctx->radix.reset(new lfhe_radix());
    const int ln = check_operand(ctx, lhs, n, width);
    const uint32_t rhs_width = shift_op(op) ? 8 : width;  // shift amounts are euint8
    const int rn = (unary || rhs_scalar) ? 1 : check_operand(ctx, rhs_src, n, rhs_width);
    const int cn = op == LFHE_OP_SELECT ? check_operand(ctx, cond, n, 1) : 1;
    // a specialised program serves instances that share the clear operand: the caller groups them
    bool special = rhs_scalar && specialised_op(op);
    if (special)
        for (uint64_t i = 1; i < n; ++i)
            if (std::memcmp(rhs, rhs + i * 32, 32)) throw std::logic_error("instances of one specialised run differ");
    // a product with a clear factor has a schedule of its own (the factor's digits pick the partial
    // products: about a third of the bootstraps), and so have and / or / xor with a clear operand (no
    // bootstrap where its digit is 0 or 3), when every instance has the same clear operand; instances
    // with different ones run the general schedule on constant blocks, in one batch
    if (rhs_scalar && shared_scalar_op(op) && n) {
        special = true;
        for (uint64_t i = 1; i < n && special; ++i) special = !std::memcmp(rhs, rhs + i * 32, 32);
    }
    Compiled& C = compiled_for(ctx, op, width, ln, rn, cn, special ? rhs : nullptr);
    const Program& P = C.prog;
    Engine& E = *ctx->eng;
    HC(hipSetDevice(E.device()));
    hipStream_t st = E.stream();
    const uint32_t m = blocks_of(width);
    const size_t inst_bytes = (size_t)P.nblocks * BLK_BYTES;
    const uint64_t in_stride = ct_bytes(width);
    if (const uint64_t cap = arena_capacity(inst_bytes); n > cap) {   // too many instances for one arena
        const uint64_t rs = rhs_scalar ? 32 : ct_bytes(rhs_width), os = ct_bytes(P.out_width);
        for (uint64_t off = 0; off < n; off += cap)
            run_batch(ctx, op, width, std::min(cap, n - off), lhs.at(off, in_stride), rhs_src.at(off, rs),
                      rhs_scalar, cond.at(off, ct_bytes(1)), out.at(off, os));
        return;
    }
    DevBuf<u64> arena(n * (size_t)P.nblocks * BLK_WORDS);
    const uint64_t out_stride = ct_bytes(P.out_width);
    DevBuf<u64> staging(n * std::max<uint64_t>({in_stride, ct_bytes(rhs_width), out_stride}) / 8);
    PtrTables tables;
    FreshEntries fresh(ctx->radix->store);
    auto upload = [&](const Src& src, uint32_t w, uint32_t first) {
        load_operand(ctx, staging.p, arena.p, P.nblocks, src, w, first, n, tables);
    };
    upload(lhs, width, P.lhs[0]);
    DevBuf<u64> scal;
    if (!unary && !special) {
        if (rhs_scalar) {
            std::vector<u64> bodies(n * m);
            for (uint64_t i = 0; i < n; ++i)
                for (uint32_t j = 0; j < m; ++j) {
                    const uint8_t* v = rhs + i * 32;
                    const u64 d = width == 1 ? (v[0] & 1u) : block_value(v, j);
                    bodies[i * m + j] = d * ctx->delta();
                }
            DevBuf<u64> tmp(bodies.size());
            scal = std::move(tmp);
            HC(hipMemcpyAsync(scal.p, bodies.data(), bodies.size() * 8, hipMemcpyHostToDevice, st));
            HC(hipStreamSynchronize(st));  // bodies goes out of scope
            launch_fill_trivial(arena.p, n, P.nblocks, P.rhs[0], m, scal.p, st);
            HC(hipGetLastError());
        } else {
            upload(rhs_src, rhs_width, P.rhs[0]);
        }
    }
    if (op == LFHE_OP_SELECT) upload(cond, 1, P.cond[0]);

    replay(ctx, C, arena.p, n);
    DevBuf<u32> out_blocks(P.out.size());
    HC(hipMemcpyAsync(out_blocks.p, P.out.data(), P.out.size() * sizeof(u32), hipMemcpyHostToDevice, st));
    const CtHeader h{CT_MAGIC, 1, (uint16_t)P.out_width, (uint32_t)P.out.size(), P.out_noise, P.out_degree, 0};
    store_result(ctx, staging.p, arena.p, P.nblocks, out_blocks.p, (uint32_t)P.out.size(), out, h, n, tables, fresh);
    E.sync();
    fresh.commit();
    if (out.host)
        for (uint64_t i = 0; i < n; ++i) std::memcpy(out.host + i * out_stride, &h, sizeof h);
}

// replay of a compiled program over an arena holding `n` instances
void replay(lfhe_ctx* ctx, Compiled& C, u64* arena, uint64_t n) {
    Engine& E = *ctx->eng;
    const Program& P = C.prog;
    for (size_t l = 0; l < P.layers.size(); ++l) {
        const Layer& L = P.layers[l];
        E.lin_layer(arena, n, P.nblocks, C.d_lin[l], (int)L.lin.size(), ctx->delta());
        if (!L.pbs.empty()) {
            E.pbs_layer(arena, n, P.nblocks, C.d_in[l], C.d_out[l], C.d_lut[l], (u32)L.pbs.size(), C.d_luts);
            ctx->pbs_count += n * L.pbs.size();
        }
    }
}

void run_fused(lfhe_ctx* ctx, const lfhe_step* steps, uint32_t nsteps, uint32_t n_inputs, const uint32_t* in_widths,
               uint64_t n, const Src* inputs, const int32_t* outputs, uint32_t n_outputs, const Dst* outs) {
    if (!steps || !nsteps || !in_widths || !inputs || !outputs || !n_outputs || !outs)
        throw std::invalid_argument("null or empty argument");
    if (!ctx->radix) ctx->radix.reset(new lfhe_radix());
    if (!n) return;
    if (!ctx->eng->has_keys()) throw std::logic_error("server keys not loaded");
    std::vector<uint32_t> widths(in_widths, in_widths + n_inputs);
    std::vector<int> noises(n_inputs);
    for (uint32_t i = 0; i < n_inputs; ++i) {
        if (!width_ok(widths[i]) || !inputs[i]) throw std::invalid_argument("bad input");
        noises[i] = check_operand(ctx, inputs[i], n, widths[i]);
    }
    std::string key((const char*)steps, nsteps * sizeof(lfhe_step));
    key.append((const char*)widths.data(), widths.size() * 4);
    key.append((const char*)noises.data(), noises.size() * sizeof(int));
    key.append((const char*)outputs, n_outputs * 4);
    if (!ctx->radix) ctx->radix.reset(new lfhe_radix());
    auto it = ctx->radix->fused.find(key);
    if (it == ctx->radix->fused.end()) {
        std::vector<FusedStep> fs(nsteps);
        for (uint32_t k = 0; k < nsteps; ++k) {
            fs[k].op = steps[k].op; fs[k].width = steps[k].bit_width;
            fs[k].lhs = steps[k].lhs; fs[k].rhs = steps[k].rhs; fs[k].cond = steps[k].cond;
            fs[k].rhs_is_scalar = steps[k].rhs_is_scalar;
            std::memcpy(fs[k].scalar, steps[k].scalar, 32);
            if (!width_ok(fs[k].width)) throw std::invalid_argument("unsupported bit width");
        }
        // fused schedules are kept per request shape (clear operands included): bounded like the
        // programs specialised on a clear operand, least recently used first
        lfhe_radix& R = *ctx->radix;
        if (R.fused.size() >= program_cache_limit()) {
            auto oldest = R.fused.begin();
            for (auto e = R.fused.begin(); e != R.fused.end(); ++e)
                if (e->second->last_use < oldest->second->last_use) oldest = e;
            HC(hipSetDevice(ctx->eng->device()));
            HC(hipStreamSynchronize(ctx->eng->stream()));
            R.fused.erase(oldest);
        }
        std::unique_ptr<Compiled> c(new Compiled());
        c->prog = build_fused(fs, widths, noises, std::vector<int32_t>(outputs, outputs + n_outputs));
        upload_program(ctx, *c);
        it = R.fused.emplace(key, std::move(c)).first;
    }
    it->second->last_use = ++ctx->radix->clock;
    Compiled& C = *it->second;
    const Program& P = C.prog;
    Engine& E = *ctx->eng;
    HC(hipSetDevice(E.device()));
    hipStream_t st = E.stream();
    const size_t inst_bytes = (size_t)P.nblocks * BLK_BYTES;
    if (const uint64_t cap = arena_capacity(inst_bytes); n > cap) {   // too many instances for one arena
        std::vector<Src> in_c(n_inputs);
        std::vector<Dst> out_c(n_outputs);
        for (uint64_t off = 0; off < n; off += cap) {
            for (uint32_t i = 0; i < n_inputs; ++i) in_c[i] = inputs[i].at(off, ct_bytes(widths[i]));
            for (uint32_t o = 0; o < n_outputs; ++o) out_c[o] = outs[o].at(off, ct_bytes(P.out_widths[o]));
            run_fused(ctx, steps, nsteps, n_inputs, in_widths, std::min(cap, n - off), in_c.data(), outputs, n_outputs,
                      out_c.data());
        }
        return;
    }
    DevBuf<u64> arena(n * (size_t)P.nblocks * BLK_WORDS);
    uint64_t widest = 0;
    for (uint32_t i = 0; i < n_inputs; ++i) widest = std::max(widest, ct_bytes(widths[i]));
    for (uint32_t o = 0; o < n_outputs; ++o) widest = std::max(widest, ct_bytes(P.out_widths[o]));
    DevBuf<u64> staging(n * widest / 8);
    PtrTables tables;
    FreshEntries fresh(ctx->radix->store);
    for (uint32_t i = 0; i < n_inputs; ++i)
        load_operand(ctx, staging.p, arena.p, P.nblocks, inputs[i], widths[i], P.ins[i][0], n, tables);
    replay(ctx, C, arena.p, n);
    std::vector<DevBuf<u32>> out_blocks;
    auto header = [&](uint32_t o) {
        return CtHeader{CT_MAGIC, 1, (uint16_t)P.out_widths[o], (uint32_t)P.outs[o].size(), P.out_noises[o], P.out_degrees[o], 0};
    };
    for (uint32_t o = 0; o < n_outputs; ++o) {
        if (!outs[o]) throw std::invalid_argument("null output");
        out_blocks.emplace_back(P.outs[o].size());
        HC(hipMemcpyAsync(out_blocks.back().p, P.outs[o].data(), P.outs[o].size() * sizeof(u32), hipMemcpyHostToDevice, st));
        store_result(ctx, staging.p, arena.p, P.nblocks, out_blocks.back().p, (uint32_t)P.outs[o].size(), outs[o], header(o), n,
                     tables, fresh);
    }
    E.sync();
    fresh.commit();
    for (uint32_t o = 0; o < n_outputs; ++o) {
        if (!outs[o].host) continue;
        const uint64_t stride = ct_bytes(P.out_widths[o]);
        const CtHeader h = header(o);
        for (uint64_t i = 0; i < n; ++i) std::memcpy(outs[o].host + i * stride, &h, sizeof h);
    }
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
        std::vector<uint64_t> blk(BLK_WORDS);
        for (uint32_t i = 0; i < nb; ++i) {
            uint64_t m = bit_width == 1 ? (value_le[0] & 1u) : block_value(value_le, i);
            if (trivial) {
                std::fill(blk.begin(), blk.end(), 0);
                blk[N] = m * ctx->delta();
            } else {
                lwe_encrypt(ctx->glwe_sk.data(), N, m * ctx->delta(), ctx->prm.glwe_noise_log2, ctx->enc_rng, blk.data());
            }
            std::memcpy(out + sizeof h + (size_t)i * BLK_BYTES, blk.data(), BLK_BYTES);
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
    std::vector<uint64_t> blk(BLK_WORDS);
    // recombine base 4 with carries: blocks may hold values up to 15
    uint64_t carry = 0;
    const uint32_t bits = h.bit_width;
    for (uint32_t i = 0; i < h.nblocks; ++i) {
        std::memcpy(blk.data(), ct + sizeof h + (size_t)i * BLK_BYTES, BLK_BYTES);
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

namespace {
constexpr uint32_t PK_MAGIC = 0x4b50464cu;  // "LFPK"
// flags bit 0: the key polynomial is the secret key in reversed coefficient order (body = mask (*) rev(s)
// + e), the convention of tfhe-rs compact public keys; blocks are then emitted with reversed masks
struct PkHeader { uint32_t magic, poly_size, noise_log2, flags; };
constexpr uint64_t PK_BYTES = sizeof(PkHeader) + 2ull * N * 8;
}  // namespace

int32_t lfhe_public_key(lfhe_ctx* ctx, uint8_t* out, uint64_t* out_len) {
    return guarded(ctx, [&]() -> int32_t {
        if (!out_len) throw std::invalid_argument("null out_len");
        if (!out) { *out_len = PK_BYTES; return LFHE_OK; }
        if (*out_len < PK_BYTES) { *out_len = PK_BYTES; return LFHE_ERR_SIZE; }
        if (!ctx->have_client) throw std::logic_error("no client key: call lfhe_keygen or import one");
        PkHeader h{PK_MAGIC, (uint32_t)N, ctx->prm.glwe_noise_log2, 0};
        std::vector<uint64_t> pk(2 * N);
        gen_public_key(ctx->prm, ctx->glwe_sk.data(), ctx->seed, pk.data());
        std::memcpy(out, &h, sizeof h);
        std::memcpy(out + sizeof h, pk.data(), pk.size() * 8);
        *out_len = PK_BYTES;
        return LFHE_OK;
    });
}

int32_t lfhe_encrypt_public(lfhe_ctx* ctx, const uint8_t* pk, uint64_t pk_len, const uint8_t value_le[32],
                            uint32_t bit_width, uint8_t* out, uint64_t* out_len) {
    return guarded(ctx, [&]() -> int32_t {
        if (!value_le || !out_len) throw std::invalid_argument("null pointer");
        if (!width_ok(bit_width)) throw std::invalid_argument("unsupported bit width");
        const uint64_t need = ct_bytes(bit_width);
        if (!out) { *out_len = need; return LFHE_OK; }
        if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
        PkHeader h;
        if (!pk || pk_len != PK_BYTES) throw std::invalid_argument("malformed public key");
        std::memcpy(&h, pk, sizeof h);
        if (h.magic != PK_MAGIC || h.poly_size != (uint32_t)N) throw std::invalid_argument("malformed public key");
        const uint32_t nb = blocks_of(bit_width);
        std::vector<uint64_t> key(2 * N), pt(nb);
        std::memcpy(key.data(), pk + sizeof h, key.size() * 8);
        for (uint32_t i = 0; i < nb; ++i)
            pt[i] = (uint64_t)(bit_width == 1 ? (value_le[0] & 1u) : block_value(value_le, i)) * ctx->delta();
        CtHeader ch{CT_MAGIC, 1, (uint16_t)bit_width, nb, 1, (uint8_t)(bit_width == 1 ? 1 : 3), 0};
        std::memcpy(out, &ch, sizeof ch);
        std::vector<uint64_t> blocks((size_t)nb * BLK_WORDS);
        pk_encrypt_blocks(ctx->prm, key.data(), pt.data(), nb, ctx->enc_rng, blocks.data());
        if (h.flags & 1u)   // LWE under rev(s) -> LWE under s
            for (uint32_t i = 0; i < nb; ++i) std::reverse(blocks.data() + (size_t)i * BLK_WORDS, blocks.data() + (size_t)i * BLK_WORDS + N);
        std::memcpy(out + sizeof ch, blocks.data(), blocks.size() * 8);
        *out_len = need;
        return LFHE_OK;
    });
}

// Oblivious pseudo-random integers: every block starts as a ciphertext whose 2049 words come
// straight from the seeded generator, so its phase is uniform and known to nobody; one bootstrap
// maps the phase to its low message bits.  The table holds the odd multiples (2r - (2^b - 1)) of
// half a message step, a set that the negacyclic sign flip of the padding-bit half maps onto
// itself, and the constant (2^b - 1) / 2 steps is added afterwards: uniform over [0, 2^b).
int32_t lfhe_rand(lfhe_ctx* ctx, uint32_t bit_width, const uint8_t seed[16], uint32_t random_bits, uint8_t* out,
                  uint64_t* out_len) {
    return guarded(ctx, [&]() -> int32_t {
        if (!seed || !out_len) throw std::invalid_argument("null pointer");
        if (!width_ok(bit_width)) throw std::invalid_argument("unsupported bit width");
        if (random_bits == 0 || random_bits > bit_width) throw std::invalid_argument("random_bits outside 1..bit_width");
        const uint64_t need = ct_bytes(bit_width);
        if (!out) { *out_len = need; return LFHE_OK; }
        if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
        if (!ctx->eng->has_keys()) throw std::logic_error("server keys not loaded");
        Engine& E = *ctx->eng;
        HC(hipSetDevice(E.device()));
        const uint32_t nb = blocks_of(bit_width);
        const uint32_t full = random_bits / 2, odd = random_bits & 1, live = full + odd;
        uint8_t key[32] = {0};
        std::memcpy(key, seed, 16);
        ChaCha rng(key, 0x500000000ull);
        std::vector<u64> in((size_t)live * BLK_WORDS), res((size_t)live * BLK_WORDS);
        for (auto& w : in) w = rng.next();
        // two tables: two random bits, one random bit (entries in units of half a message step)
        const u64 half = ctx->delta() / 2;
        std::vector<u64> bodies(2 * (size_t)N);
        uint64_t tab[16];
        for (int x = 0; x < 16; ++x) tab[x] = (u64)(i64)(2 * (x & 3) - 3);
        make_lut_body(ctx->space(), half, tab, bodies.data());
        for (int x = 0; x < 16; ++x) tab[x] = (u64)(i64)(2 * (x & 1) - 1);
        make_lut_body(ctx->space(), half, tab, bodies.data() + N);
        std::vector<u32> idx(live, 0);
        if (odd) idx[live - 1] = 1;
        DevBuf<u64> din(in.size()), dout(in.size()), dl(bodies.size());
        DevBuf<u32> di(live);
        hipStream_t st = E.stream();
        HC(hipMemcpyAsync(din.p, in.data(), in.size() * 8, hipMemcpyHostToDevice, st));
        HC(hipMemcpyAsync(dl.p, bodies.data(), bodies.size() * 8, hipMemcpyHostToDevice, st));
        HC(hipMemcpyAsync(di.p, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, st));
        E.pbs_batch(din.p, dl.p, di.p, dout.p, live);
        HC(hipMemcpyAsync(res.data(), dout.p, res.size() * 8, hipMemcpyDeviceToHost, st));
        E.sync();
        ctx->pbs_count += live;
        CtHeader h{CT_MAGIC, 1, (uint16_t)bit_width, nb, 1, (uint8_t)(bit_width == 1 ? 1 : 3), 0};
        std::memcpy(out, &h, sizeof h);
        std::memset(out + sizeof h, 0, (size_t)nb * BLK_BYTES);   // blocks above the random bits: trivial zeros
        for (uint32_t b = 0; b < live; ++b) {
            u64* blk = res.data() + (size_t)b * BLK_WORDS;
            blk[N] += (odd && b == live - 1) ? half : 3 * half;
            std::memcpy(out + sizeof h + (size_t)b * BLK_BYTES, blk, BLK_BYTES);
        }
        *out_len = need;
        return LFHE_OK;
    });
}

// ---- ciphertexts made under a separate public-key secret (tfhe-rs: dedicated compact public key
// parameters + a key-switching key towards the compute parameters) ---------------------------------
int32_t lfhe_pke_key_set(lfhe_ctx* ctx, const uint64_t* pke_sk) {
    return guarded(ctx, [&]() -> int32_t {
        if (!pke_sk) throw std::invalid_argument("null key pointer");
        if (!ctx->have_client) throw std::logic_error("no client keys: the cast key encrypts under the small key");
        for (int i = 0; i < N; ++i)
            if (pke_sk[i] > 1) throw std::invalid_argument("secret key words must be 0 or 1");
        ctx->pke_sk.assign(pke_sk, pke_sk + N);
        std::vector<uint64_t> ksk(ctx->eng->ksk_words());
        unsigned hc = std::thread::hardware_concurrency();
        gen_ksk(ctx->prm, ctx->pke_sk.data(), ctx->lwe_sk.data(), ctx->seed, ksk.data(), (int)(hc ? (hc > 16 ? 16 : hc) : 4),
                0x600000000ull);
        ctx->eng->load_cast_ksk(ksk.data());
        return LFHE_OK;
    });
}

int32_t lfhe_cast_from_pke(lfhe_ctx* ctx, const uint8_t* ct, uint64_t ct_len, uint8_t* out, uint64_t* out_len) {
    return guarded(ctx, [&]() -> int32_t {
        if (!ct || !out_len) throw std::invalid_argument("null pointer");
        CtHeader h;
        if (ct_len < sizeof h) throw std::invalid_argument("ciphertext too short");
        std::memcpy(&h, ct, sizeof h);
        if (h.magic != CT_MAGIC || h.version != 1 || !width_ok(h.bit_width) || h.nblocks != blocks_of(h.bit_width) ||
            ct_len != ct_bytes(h.bit_width))
            throw std::invalid_argument("malformed ciphertext");
        if (!out) { *out_len = ct_len; return LFHE_OK; }
        if (*out_len < ct_len) { *out_len = ct_len; return LFHE_ERR_SIZE; }
        Engine& E = *ctx->eng;
        if (!E.has_keys() || !E.has_cast_key()) throw std::logic_error("server keys or cast key not loaded");
        HC(hipSetDevice(E.device()));
        const size_t words = (size_t)h.nblocks * BLK_WORDS;
        std::vector<u64> lut(N);
        uint64_t tab[16];
        for (int x = 0; x < 16; ++x) tab[x] = (uint64_t)x;
        make_lut_body(ctx->space(), ctx->delta(), tab, lut.data());
        DevBuf<u64> din(words), dout(words), dl(N);
        hipStream_t st = E.stream();
        HC(hipMemcpyAsync(din.p, ct + sizeof h, words * 8, hipMemcpyHostToDevice, st));
        HC(hipMemcpyAsync(dl.p, lut.data(), (size_t)N * 8, hipMemcpyHostToDevice, st));
        E.cast_batch(din.p, dl.p, dout.p, h.nblocks);
        HC(hipMemcpyAsync(out + sizeof h, dout.p, words * 8, hipMemcpyDeviceToHost, st));
        E.sync();
        ctx->pbs_count += h.nblocks;
        h.noise_level = 1;
        std::memcpy(out, &h, sizeof h);
        *out_len = ct_len;
        return LFHE_OK;
    });
}

// A client's compact ciphertext list -> radix ciphertexts under the compute key.  The list is one
// RLWE-style encryption under the public-key secret: mask polynomial c0 and one body word per LWE
// (coefficient i of c1).  Expansion is the sample extraction of coefficient i in the convention of
// tfhe-rs compact public keys (body = mask (*) rev(s) + e, so the extracted mask is reversed to give
// an LWE under s itself), then the cast of lfhe_cast_from_pke.  A packed list holds two radix blocks
// per LWE (low block + 4 * high block); the cast's bootstrap splits them with two tables.
int32_t lfhe_expand_compact_list(lfhe_ctx* ctx, const uint64_t* mask, const uint64_t* bodies, uint64_t n_lwe,
                                 int32_t packed, const uint32_t* bit_widths, uint32_t n_values, uint8_t* out,
                                 uint64_t* out_len) {
    return guarded(ctx, [&]() -> int32_t {
        if (!mask || !bodies || !bit_widths || !out_len || !n_values) throw std::invalid_argument("null or empty argument");
        uint64_t total = 0, need = 0;
        for (uint32_t v = 0; v < n_values; ++v) {
            if (!width_ok(bit_widths[v])) throw std::invalid_argument("unsupported bit width");
            total += blocks_of(bit_widths[v]);
            need += ct_bytes(bit_widths[v]);
        }
        if (n_lwe == 0 || n_lwe > (uint64_t)N) throw std::invalid_argument("a compact list holds 1 .. 2048 ciphertexts per mask");
        if (n_lwe != (packed ? (total + 1) / 2 : total)) throw std::invalid_argument("list length does not match the bit widths");
        if (!out) { *out_len = need; return LFHE_OK; }
        if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
        Engine& E = *ctx->eng;
        if (!E.has_keys() || !E.has_cast_key()) throw std::logic_error("server keys or cast key not loaded");
        HC(hipSetDevice(E.device()));
        std::vector<u64> lwe((size_t)n_lwe * BLK_WORDS);
        for (uint64_t i = 0; i < n_lwe; ++i) {
            u64* o = lwe.data() + i * BLK_WORDS;
            for (uint64_t j = 0; j <= i; ++j) o[j] = mask[i - j];
            for (uint64_t j = i + 1; j < (uint64_t)N; ++j) o[j] = (u64)0 - mask[N + i - j];
            std::reverse(o, o + N);      // LWE under rev(s) -> LWE under s
            o[N] = bodies[i];
        }
        const uint32_t mm = ctx->prm.message_modulus;
        std::vector<u64> luts(2 * (size_t)N);
        uint64_t tab[16];
        for (uint32_t x = 0; x < 16; ++x) tab[x] = packed ? x % mm : x;
        make_lut_body(ctx->space(), ctx->delta(), tab, luts.data());
        for (uint32_t x = 0; x < 16; ++x) tab[x] = x / mm;
        make_lut_body(ctx->space(), ctx->delta(), tab, luts.data() + N);
        DevBuf<u64> din(lwe.size()), dlo(lwe.size()), dhi(packed ? lwe.size() : 0), dl(luts.size());
        hipStream_t st = E.stream();
        HC(hipMemcpyAsync(din.p, lwe.data(), lwe.size() * 8, hipMemcpyHostToDevice, st));
        HC(hipMemcpyAsync(dl.p, luts.data(), luts.size() * 8, hipMemcpyHostToDevice, st));
        E.cast_batch(din.p, dl.p, dlo.p, n_lwe);
        if (packed) E.cast_batch(din.p, dl.p + N, dhi.p, n_lwe);
        std::vector<u64> lo(lwe.size()), hi(packed ? lwe.size() : 0);
        HC(hipMemcpyAsync(lo.data(), dlo.p, lo.size() * 8, hipMemcpyDeviceToHost, st));
        if (packed) HC(hipMemcpyAsync(hi.data(), dhi.p, hi.size() * 8, hipMemcpyDeviceToHost, st));
        E.sync();
        ctx->pbs_count += n_lwe * (packed ? 2 : 1);
        uint8_t* w = out;
        uint64_t b = 0;   // running radix block number over the whole list
        for (uint32_t v = 0; v < n_values; ++v) {
            const uint32_t nb = blocks_of(bit_widths[v]);
            CtHeader h{CT_MAGIC, 1, (uint16_t)bit_widths[v], nb, 1, (uint8_t)(bit_widths[v] == 1 ? 1 : 3), 0};
            std::memcpy(w, &h, sizeof h);
            for (uint32_t k = 0; k < nb; ++k, ++b) {
                const u64* src = packed ? ((b & 1) ? hi.data() : lo.data()) + (b / 2) * BLK_WORDS : lo.data() + b * BLK_WORDS;
                std::memcpy(w + sizeof h + (size_t)k * BLK_BYTES, src, BLK_BYTES);
            }
            w += ct_bytes(bit_widths[v]);
        }
        *out_len = need;
        return LFHE_OK;
    });
}

// ---- compression for storage ----------------------------------------------------------------------
namespace {
constexpr uint32_t CZ_MAGIC = 0x5a43464cu;  // "LFCZ"
struct CzHeader {
    uint32_t magic;
    uint16_t version, storage_log;
    uint32_t n_cts, total_blocks;
    uint32_t kc, nc;
};
struct CzEntry { uint16_t bit_width, nblocks; };
static_assert(sizeof(CzHeader) == 24 && sizeof(CzEntry) == 4, "blob layout");

// stored words of GLWE g: the kc mask polynomials and the first `cnt` body coefficients
size_t cz_group_bytes(const CompShape& c, size_t cnt) { return (((size_t)c.small_dim() + cnt) * c.storage_log + 7) / 8; }

void pack_bits(const uint16_t* v, size_t n, int bits, uint8_t* out) {
    uint64_t acc = 0;
    int have = 0;
    for (size_t i = 0; i < n; ++i) {
        acc |= (uint64_t)v[i] << have;
        have += bits;
        while (have >= 8) { *out++ = (uint8_t)acc; acc >>= 8; have -= 8; }
    }
    if (have) *out = (uint8_t)acc;
}
void unpack_bits(const uint8_t* in, size_t n, int bits, uint16_t* v) {
    uint64_t acc = 0;
    int have = 0;
    for (size_t i = 0; i < n; ++i) {
        while (have < bits) { acc |= (uint64_t)*in++ << have; have += 8; }
        v[i] = (uint16_t)(acc & ((1u << bits) - 1));
        acc >>= bits;
        have -= bits;
    }
}
}  // namespace

int32_t lfhe_compression_keygen(lfhe_ctx* ctx) {
    return guarded(ctx, [&]() -> int32_t {
        if (!ctx->have_client) throw std::logic_error("no client keys: the compression keys are made from them");
        if (!ctx->eng->has_keys()) throw std::logic_error("server keys not loaded");
        const CompShape c;
        unsigned hc = std::thread::hardware_concurrency();
        const int threads = (int)(hc ? (hc > 16 ? 16 : hc) : 4);
        std::vector<uint64_t> comp_sk((size_t)c.small_dim());
        ChaCha r(ctx->seed, 7);
        gen_binary_key(r, comp_sk.data(), comp_sk.size());
        std::vector<uint64_t> pksk(c.pksk_words());
        gen_packing_ksk(c, ctx->glwe_sk.data(), comp_sk.data(), ctx->seed, pksk.data(), threads);
        // bootstrapping key for the compression secret, of the context's own flavour, from its own
        // generator streams (a seed of its own: the masks must not repeat the main key's)
        EngineParams p = ctx->prm;
        p.lwe_dim = (uint32_t)c.small_dim();
        uint8_t seed2[32];
        std::memcpy(seed2, ctx->seed, 32);
        seed2[0] ^= 0xC5; seed2[31] ^= 0x5C;
        std::vector<uint64_t> bsk(ctx->eng->ggsw_count_for(p.lwe_dim) * 4 * (size_t)N);
        if (p.bsk_modulus) gen_bsk_qp(p, comp_sk.data(), ctx->glwe_sk.data(), seed2, bsk.data(), threads);
        else gen_bsk(p, comp_sk.data(), ctx->glwe_sk.data(), seed2, bsk.data(), threads);
        ctx->eng->load_compression_keys(c, pksk.data(), bsk.data());
        return LFHE_OK;
    });
}

int32_t lfhe_compression_keys_import_raw(lfhe_ctx* ctx, const uint64_t* packing_key, uint64_t packing_words,
                                         const uint64_t* decompression_bsk, uint64_t bsk_words) {
    return guarded(ctx, [&]() -> int32_t {
        if (!packing_key || !decompression_bsk) throw std::invalid_argument("null key pointer");
        const CompShape c;
        if (packing_words != c.pksk_words() ||
            bsk_words != ctx->eng->ggsw_count_for((uint32_t)c.small_dim()) * 4 * (uint64_t)N)
            throw std::invalid_argument("compression key size does not match the parameter set");
        if (ctx->prm.bsk_modulus)
            for (uint64_t i = 0; i < bsk_words; ++i)
                if (decompression_bsk[i] >= ctx->prm.bsk_modulus) throw std::invalid_argument("bootstrapping key word not reduced");
        ctx->eng->load_compression_keys(c, packing_key, decompression_bsk);
        return LFHE_OK;
    });
}

int32_t lfhe_compress(lfhe_ctx* ctx, const uint8_t* cts, uint64_t cts_len, uint8_t* out, uint64_t* out_len) {
    return guarded(ctx, [&]() -> int32_t {
        if (!cts || !out_len) throw std::invalid_argument("null pointer");
        Engine& E = *ctx->eng;
        if (!E.has_compression_keys()) throw std::logic_error("no compression keys: call lfhe_compression_keygen");
        const CompShape& c = E.comp_shape();
        // walk the ciphertexts
        std::vector<CzEntry> dir;
        std::vector<const uint8_t*> starts;
        size_t total = 0;
        for (uint64_t off = 0; off < cts_len;) {
            if (cts_len - off < sizeof(CtHeader)) throw std::invalid_argument("truncated ciphertext");
            CtHeader h;
            std::memcpy(&h, cts + off, sizeof h);
            if (h.magic != CT_MAGIC || h.version != 1 || !width_ok(h.bit_width) || h.nblocks != blocks_of(h.bit_width) ||
                cts_len - off < ct_bytes(h.bit_width))
                throw std::invalid_argument("malformed ciphertext");
            if (h.degree > 3 || h.noise_level < 1 || h.noise_level > 2) throw std::invalid_argument("ciphertext metadata out of range");
            dir.push_back(CzEntry{h.bit_width, (uint16_t)h.nblocks});
            starts.push_back(cts + off);
            total += h.nblocks;
            off += ct_bytes(h.bit_width);
        }
        if (dir.empty()) throw std::invalid_argument("nothing to compress");
        const size_t groups = (total + c.nc - 1) / c.nc;
        size_t need = sizeof(CzHeader) + dir.size() * sizeof(CzEntry);
        for (size_t g = 0; g < groups; ++g) need += cz_group_bytes(c, std::min<size_t>(c.nc, total - g * c.nc));
        if (!out) { *out_len = need; return LFHE_OK; }
        if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
        HC(hipSetDevice(E.device()));
        hipStream_t st = E.stream();
        DevBuf<u64> din(total * BLK_WORDS);
        size_t b0 = 0;
        for (size_t i = 0; i < dir.size(); ++i) {
            HC(hipMemcpyAsync(din.p + b0 * BLK_WORDS, starts[i] + sizeof(CtHeader), (size_t)dir[i].nblocks * BLK_BYTES,
                              hipMemcpyHostToDevice, st));
            b0 += dir[i].nblocks;
        }
        DevBuf<uint16_t> dg(groups * (size_t)c.glwe_words());
        // the message moves up by the message modulus into the (clear) carry bits: the stored modulus keeps it
        E.compress(din.p, total, dg.p, ctx->prm.message_modulus);
        std::vector<uint16_t> hg(groups * (size_t)c.glwe_words());
        HC(hipMemcpy(hg.data(), dg.p, hg.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
        CzHeader zh{CZ_MAGIC, 1, (uint16_t)c.storage_log, (uint32_t)dir.size(), (uint32_t)total, (uint32_t)c.kc, (uint32_t)c.nc};
        uint8_t* w = out;
        std::memcpy(w, &zh, sizeof zh); w += sizeof zh;
        std::memcpy(w, dir.data(), dir.size() * sizeof(CzEntry)); w += dir.size() * sizeof(CzEntry);
        for (size_t g = 0; g < groups; ++g) {
            const size_t cnt = std::min<size_t>(c.nc, total - g * c.nc);
            pack_bits(hg.data() + g * c.glwe_words(), (size_t)c.small_dim() + cnt, c.storage_log, w);
            w += cz_group_bytes(c, cnt);
        }
        *out_len = need;
        return LFHE_OK;
    });
}

int32_t lfhe_decompress(lfhe_ctx* ctx, const uint8_t* blob, uint64_t blob_len, uint8_t* out, uint64_t* out_len) {
    return guarded(ctx, [&]() -> int32_t {
        if (!blob || !out_len) throw std::invalid_argument("null pointer");
        Engine& E = *ctx->eng;
        if (!E.has_compression_keys()) throw std::logic_error("no compression keys: call lfhe_compression_keygen");
        const CompShape& c = E.comp_shape();
        CzHeader zh;
        if (blob_len < sizeof zh) throw std::invalid_argument("blob too short");
        std::memcpy(&zh, blob, sizeof zh);
        if (zh.magic != CZ_MAGIC || zh.version != 1 || zh.storage_log != (uint32_t)c.storage_log || zh.kc != (uint32_t)c.kc ||
            zh.nc != (uint32_t)c.nc || zh.n_cts == 0 || zh.n_cts > (1u << 24) ||
            blob_len < sizeof zh + (uint64_t)zh.n_cts * sizeof(CzEntry))
            throw std::invalid_argument("malformed compressed list");
        std::vector<CzEntry> dir(zh.n_cts);
        std::memcpy(dir.data(), blob + sizeof zh, dir.size() * sizeof(CzEntry));
        size_t total = 0;
        uint64_t need = 0;
        for (const CzEntry& e : dir) {
            if (!width_ok(e.bit_width) || e.nblocks != blocks_of(e.bit_width)) throw std::invalid_argument("malformed compressed list");
            total += e.nblocks;
            need += ct_bytes(e.bit_width);
        }
        if (total != zh.total_blocks) throw std::invalid_argument("malformed compressed list");
        const size_t groups = (total + c.nc - 1) / c.nc;
        uint64_t packed = 0;
        for (size_t g = 0; g < groups; ++g) packed += cz_group_bytes(c, std::min<size_t>(c.nc, total - g * c.nc));
        if (blob_len != sizeof zh + dir.size() * sizeof(CzEntry) + packed) throw std::invalid_argument("malformed compressed list");
        if (!out) { *out_len = need; return LFHE_OK; }
        if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
        std::vector<uint16_t> hg(groups * (size_t)c.glwe_words(), 0);
        const uint8_t* r = blob + sizeof zh + dir.size() * sizeof(CzEntry);
        for (size_t g = 0; g < groups; ++g) {
            const size_t cnt = std::min<size_t>(c.nc, total - g * c.nc);
            unpack_bits(r, (size_t)c.small_dim() + cnt, c.storage_log, hg.data() + g * c.glwe_words());
            r += cz_group_bytes(c, cnt);
        }
        HC(hipSetDevice(E.device()));
        hipStream_t st = E.stream();
        DevBuf<uint16_t> dg(hg.size());
        HC(hipMemcpyAsync(dg.p, hg.data(), hg.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
        // the table undoes the shift: value 4 m (+ up to two boxes of noise either way) -> m
        std::vector<u64> lut(N);
        uint64_t tab[16];
        const uint32_t mm = ctx->prm.message_modulus;
        for (uint32_t x = 0; x < 16; ++x) tab[x] = ((x + mm / 2) / mm) % mm;
        make_lut_body(ctx->space(), ctx->delta(), tab, lut.data());
        DevBuf<u64> dl(N), dout(total * BLK_WORDS);
        HC(hipMemcpyAsync(dl.p, lut.data(), (size_t)N * 8, hipMemcpyHostToDevice, st));
        E.decompress(dg.p, total, dl.p, dout.p);
        ctx->pbs_count += total;
        uint8_t* w = out;
        size_t b0 = 0;
        for (const CzEntry& e : dir) {
            CtHeader h{CT_MAGIC, 1, e.bit_width, e.nblocks, 1, (uint8_t)(e.bit_width == 1 ? 1 : 3), 0};
            std::memcpy(w, &h, sizeof h);
            HC(hipMemcpy(w + sizeof h, dout.p + b0 * BLK_WORDS, (size_t)e.nblocks * BLK_BYTES, hipMemcpyDeviceToHost));
            w += ct_bytes(e.bit_width);
            b0 += e.nblocks;
        }
        *out_len = need;
        return LFHE_OK;
    });
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

int32_t lfhe_program_dump(uint32_t op, uint32_t bit_width, int32_t lhs_noise, int32_t rhs_noise, int32_t cond_noise,
                          const uint8_t* scalar, char* out, uint64_t* out_len) {
    if (!out_len) return LFHE_ERR_ARG;
    std::string js;
    try {
        if (!width_ok(bit_width)) return LFHE_ERR_ARG;
        js = build_program(op, bit_width, lhs_noise, rhs_noise, cond_noise,
                           specialised_op(op) || shared_scalar_op(op) ? scalar : nullptr).to_json();
    } catch (const std::domain_error&) {
        return LFHE_ERR_ARG;
    } catch (const std::invalid_argument&) {
        return LFHE_ERR_UNSUPPORTED;
    } catch (const std::exception&) {
        return LFHE_ERR_STATE;
    }
    const uint64_t need = js.size() + 1;
    if (!out) { *out_len = need; return LFHE_OK; }
    if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
    std::memcpy(out, js.c_str(), need);
    *out_len = need;
    return LFHE_OK;
}

// n instances of one operator; a specialised operator (shift amount, divisor) with clear operands that
// differ between instances runs once per distinct value, on the instances that share it
static int32_t evaluate_batch_impl(lfhe_ctx* ctx, uint32_t op, uint32_t bit_width, uint64_t n, const Src& lhs,
                                   const Src& rhs_src, int32_t rhs_is_scalar, const Src& cond, const Dst& out) {
    if (!ctx) return LFHE_ERR_ARG;
    if (op == LFHE_OP_CAST) {
        ctx->err = "cast has no batch form: it moves no data through the GPU";
        return LFHE_ERR_UNSUPPORTED;
    }
    int32_t rc = guarded(ctx, [&]() -> int32_t {
        const uint8_t* rhs = rhs_src.host;
        const bool special = rhs_is_scalar && specialised_op(op) && rhs && n > 1;
        bool uniform = true;
        if (special)
            for (uint64_t i = 1; i < n && uniform; ++i) uniform = !std::memcmp(rhs, rhs + i * 32, 32);
        if (uniform) {
            run_batch(ctx, op, bit_width, n, lhs, rhs_src, rhs_is_scalar != 0, cond, out);
            return LFHE_OK;
        }
        // instances grouped by their clear operand; each group is one specialised run
        if (!width_ok(bit_width) || !lhs || !out) throw std::invalid_argument("bad argument");
        std::map<Scalar, std::vector<uint64_t>> groups;
        for (uint64_t i = 0; i < n; ++i) {
            Scalar sc;
            std::memcpy(sc.data(), rhs + i * 32, 32);
            groups[sc].push_back(i);
        }
        const uint64_t stride = ct_bytes(bit_width);
        for (const auto& [sc, idx] : groups) {
            std::vector<uint8_t> gr(idx.size() * 32);
            for (size_t k = 0; k < idx.size(); ++k) std::memcpy(gr.data() + k * 32, sc.data(), 32);
            if (lhs.handles) {   // handles in, handles out: the group's handles side by side
                if (!out.handles) throw std::invalid_argument("operands from the store need results into the store");
                std::vector<uint8_t> gl(idx.size() * 32), go(idx.size() * 32);
                for (size_t k = 0; k < idx.size(); ++k) {
                    std::memcpy(gl.data() + k * 32, lhs.handles + idx[k] * 32, 32);
                    std::memcpy(go.data() + k * 32, out.handles + idx[k] * 32, 32);
                }
                run_batch(ctx, op, bit_width, idx.size(), Src{nullptr, gl.data()}, Src{gr.data(), nullptr}, true, Src{},
                          Dst{nullptr, go.data()});
                continue;
            }
            std::vector<uint8_t> gl(idx.size() * stride), go(idx.size() * stride);
            for (size_t k = 0; k < idx.size(); ++k) std::memcpy(gl.data() + k * stride, lhs.host + idx[k] * stride, stride);
            run_batch(ctx, op, bit_width, idx.size(), Src{gl.data(), nullptr}, Src{gr.data(), nullptr}, true, Src{},
                      Dst{go.data(), nullptr});
            for (size_t k = 0; k < idx.size(); ++k) std::memcpy(out.host + idx[k] * stride, go.data() + k * stride, stride);
        }
        return LFHE_OK;
    });
    if (rc == LFHE_ERR_ARG && ctx->err == "operator not supported") rc = LFHE_ERR_UNSUPPORTED;
    return rc;
}

int32_t lfhe_evaluate_batch(lfhe_ctx* ctx, uint32_t op, uint32_t bit_width, uint64_t n, const uint8_t* lhs,
                            const uint8_t* rhs, int32_t rhs_is_scalar, const uint8_t* cond, uint8_t* out) {
    return evaluate_batch_impl(ctx, op, bit_width, n, Src{lhs, nullptr}, Src{rhs, nullptr}, rhs_is_scalar, Src{cond, nullptr},
                               Dst{out, nullptr});
}

int32_t lfhe_evaluate_batch_handles(lfhe_ctx* ctx, uint32_t op, uint32_t bit_width, uint64_t n, const uint8_t* lhs_handles,
                                    const uint8_t* rhs, int32_t rhs_is_scalar, const uint8_t* cond_handles,
                                    const uint8_t* out_handles) {
    return evaluate_batch_impl(ctx, op, bit_width, n, Src{nullptr, lhs_handles},
                               rhs_is_scalar ? Src{rhs, nullptr} : Src{nullptr, rhs}, rhs_is_scalar, Src{nullptr, cond_handles},
                               Dst{nullptr, out_handles});
}

int32_t lfhe_evaluate_fused(lfhe_ctx* ctx, const lfhe_step* steps, uint32_t nsteps, uint32_t n_inputs,
                            const uint32_t* input_widths, uint64_t n, const uint8_t* const* inputs,
                            const int32_t* outputs, uint32_t n_outputs, uint8_t* const* outs) {
    int32_t rc = guarded(ctx, [&]() -> int32_t {
        if (!inputs || !outs) throw std::invalid_argument("null or empty argument");
        std::vector<Src> in(n_inputs);
        std::vector<Dst> out(n_outputs);
        for (uint32_t i = 0; i < n_inputs; ++i) in[i].host = inputs[i];
        for (uint32_t o = 0; o < n_outputs; ++o) out[o].host = outs[o];
        run_fused(ctx, steps, nsteps, n_inputs, input_widths, n, in.data(), outputs, n_outputs, out.data());
        return LFHE_OK;
    });
    if (rc == LFHE_ERR_ARG && ctx->err == "operator not supported") rc = LFHE_ERR_UNSUPPORTED;
    return rc;
}

int32_t lfhe_evaluate_fused_handles(lfhe_ctx* ctx, const lfhe_step* steps, uint32_t nsteps, uint32_t n_inputs,
                                    const uint32_t* input_widths, uint64_t n, const uint8_t* const* input_handles,
                                    const int32_t* outputs, uint32_t n_outputs, const uint8_t* const* out_handles) {
    int32_t rc = guarded(ctx, [&]() -> int32_t {
        if (!input_handles || !out_handles) throw std::invalid_argument("null or empty argument");
        std::vector<Src> in(n_inputs);
        std::vector<Dst> out(n_outputs);
        for (uint32_t i = 0; i < n_inputs; ++i) in[i].handles = input_handles[i];
        for (uint32_t o = 0; o < n_outputs; ++o) out[o].handles = out_handles[o];
        run_fused(ctx, steps, nsteps, n_inputs, input_widths, n, in.data(), outputs, n_outputs, out.data());
        return LFHE_OK;
    });
    if (rc == LFHE_ERR_ARG && ctx->err == "operator not supported") rc = LFHE_ERR_UNSUPPORTED;
    return rc;
}

// ---- the ciphertext store itself
int32_t lfhe_store_put(lfhe_ctx* ctx, const uint8_t handle[32], const uint8_t* ct, uint64_t ct_len) {
    return guarded(ctx, [&]() -> int32_t {
        if (!handle || !ct || ct_len < sizeof(CtHeader)) throw std::invalid_argument("null or short argument");
        CtHeader h;
        std::memcpy(&h, ct, sizeof h);
        if (h.magic != CT_MAGIC || h.version != 1 || !width_ok(h.bit_width) || h.nblocks != blocks_of(h.bit_width) ||
            ct_len != ct_bytes(h.bit_width))
            throw std::invalid_argument("malformed ciphertext");
        if (h.noise_level < 1 || h.noise_level > 2 || h.degree > 3) throw std::invalid_argument("ciphertext metadata out of range");
        if (!ctx->radix) ctx->radix.reset(new lfhe_radix());
        HC(hipSetDevice(ctx->eng->device()));
        lfhe_store& S = ctx->radix->store;
        FreshEntries fresh(S);
        StoreEntry e;
        e.dev = S.alloc(h.nblocks);
        e.width = h.bit_width; e.nblocks = h.nblocks; e.noise = h.noise_level; e.degree = h.degree;
        fresh.items.emplace_back(handle_at(handle, 0), e);
        HC(hipMemcpy(e.dev, ct + sizeof h, (size_t)h.nblocks * BLK_BYTES, hipMemcpyHostToDevice));
        fresh.commit();
        return LFHE_OK;
    });
}

int32_t lfhe_store_get(lfhe_ctx* ctx, const uint8_t handle[32], uint8_t* out, uint64_t* out_len) {
    return guarded(ctx, [&]() -> int32_t {
        if (!handle || !out_len) throw std::invalid_argument("null argument");
        if (!ctx->radix) throw std::invalid_argument("unknown handle");
        const auto it = ctx->radix->store.map.find(handle_at(handle, 0));
        if (it == ctx->radix->store.map.end()) throw std::invalid_argument("unknown handle");
        const StoreEntry& e = it->second;
        const uint64_t need = ct_bytes(e.width);
        if (!out) { *out_len = need; return LFHE_OK; }
        if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
        HC(hipSetDevice(ctx->eng->device()));
        ctx->eng->sync();
        const CtHeader h{CT_MAGIC, 1, e.width, e.nblocks, e.noise, e.degree, 0};
        std::memcpy(out, &h, sizeof h);
        HC(hipMemcpy(out + sizeof h, e.dev, (size_t)e.nblocks * BLK_BYTES, hipMemcpyDeviceToHost));
        *out_len = need;
        return LFHE_OK;
    });
}

int32_t lfhe_store_drop(lfhe_ctx* ctx, const uint8_t handle[32]) {
    return guarded(ctx, [&]() -> int32_t {
        if (!handle) throw std::invalid_argument("null argument");
        if (!ctx->radix) throw std::invalid_argument("unknown handle");
        lfhe_store& S = ctx->radix->store;
        const auto it = S.map.find(handle_at(handle, 0));
        if (it == S.map.end()) throw std::invalid_argument("unknown handle");
        ctx->eng->sync();          // nothing in flight reads the blocks that go back to the free list
        S.release(it->second);
        S.map.erase(it);
        return LFHE_OK;
    });
}

int32_t lfhe_store_stat(const lfhe_ctx* ctx, const uint8_t handle[32], uint64_t* entries, uint64_t* device_bytes,
                        uint32_t* bit_width) {
    if (!ctx) return LFHE_ERR_ARG;
    const lfhe_store* S = ctx->radix ? &ctx->radix->store : nullptr;
    if (entries) *entries = S ? S->map.size() : 0;
    if (device_bytes) *device_bytes = S ? S->reserved : 0;
    if (handle) {
        if (!S) return LFHE_ERR_ARG;
        const auto it = S->map.find(handle_at(handle, 0));
        if (it == S->map.end()) return LFHE_ERR_ARG;
        if (bit_width) *bit_width = it->second.width;
    }
    return LFHE_OK;
}

int32_t lfhe_fused_dump(const lfhe_step* steps, uint32_t nsteps, uint32_t n_inputs, const uint32_t* input_widths,
                        const int32_t* input_noise, const int32_t* outputs, uint32_t n_outputs, char* out,
                        uint64_t* out_len) {
    if (!steps || !input_widths || !outputs || !out_len) return LFHE_ERR_ARG;
    std::string js;
    try {
        std::vector<FusedStep> fs(nsteps);
        for (uint32_t k = 0; k < nsteps; ++k) {
            fs[k].op = steps[k].op; fs[k].width = steps[k].bit_width;
            fs[k].lhs = steps[k].lhs; fs[k].rhs = steps[k].rhs; fs[k].cond = steps[k].cond;
            fs[k].rhs_is_scalar = steps[k].rhs_is_scalar;
            std::memcpy(fs[k].scalar, steps[k].scalar, 32);
            if (!width_ok(fs[k].width)) return LFHE_ERR_ARG;
        }
        std::vector<int> noise(n_inputs, 1);
        if (input_noise) noise.assign(input_noise, input_noise + n_inputs);
        js = build_fused(fs, std::vector<uint32_t>(input_widths, input_widths + n_inputs), noise,
                         std::vector<int32_t>(outputs, outputs + n_outputs)).to_json();
    } catch (const std::domain_error&) {
        return LFHE_ERR_ARG;
    } catch (const std::invalid_argument& e) {
        return std::string(e.what()) == "operator not supported" ? LFHE_ERR_UNSUPPORTED : LFHE_ERR_ARG;
    } catch (const std::exception&) {
        return LFHE_ERR_STATE;
    }
    const uint64_t need = js.size() + 1;
    if (!out) { *out_len = need; return LFHE_OK; }
    if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
    std::memcpy(out, js.c_str(), need);
    *out_len = need;
    return LFHE_OK;
}

int32_t lfhe_evaluate(lfhe_ctx* ctx, uint32_t op, uint32_t bit_width, const uint8_t* lhs, uint64_t lhs_len,
                      const uint8_t* rhs, uint64_t rhs_len, int32_t rhs_is_scalar, const uint8_t* cond,
                      uint64_t cond_len, uint8_t* out, uint64_t* out_len) {
    if (!ctx) return LFHE_ERR_ARG;
    if (!out_len || !width_ok(bit_width)) { ctx->err = "bad argument"; return LFHE_ERR_ARG; }
    const bool cmp = op >= LFHE_OP_EQ && op <= LFHE_OP_LT;
    const uint64_t need = ct_bytes(cmp ? 1 : bit_width);
    if (!out) { *out_len = need; return LFHE_OK; }
    if (*out_len < need) { *out_len = need; return LFHE_ERR_SIZE; }
    if (op == LFHE_OP_CAST) {
        const int32_t rc = guarded(ctx, [&]() -> int32_t {
            if (!lhs) throw std::invalid_argument("null operand");
            cast_one(ctx, lhs, lhs_len, bit_width, out);
            return LFHE_OK;
        });
        if (rc == LFHE_OK) *out_len = need;
        return rc;
    }
    const bool unary = op == LFHE_OP_NOT || op == LFHE_OP_NEG;
    if (!lhs || lhs_len != ct_bytes(bit_width) || (!unary && !rhs) ||
        (!unary && rhs_len != (rhs_is_scalar ? 32u : ct_bytes(shift_op(op) ? 8 : bit_width))) ||
        (op == LFHE_OP_SELECT && (!cond || cond_len != ct_bytes(1)))) {
        ctx->err = "operand length does not match the bit width";
        return LFHE_ERR_ARG;
    }
    const int32_t rc = lfhe_evaluate_batch(ctx, op, bit_width, 1, lhs, rhs, rhs_is_scalar, cond, out);
    if (rc == LFHE_OK) *out_len = need;
    return rc;
}

int32_t lfhe_pbs_count(const lfhe_ctx* ctx, uint64_t* out) {
    if (!ctx || !out) return LFHE_ERR_ARG;
    *out = ctx->pbs_count;
    return LFHE_OK;
}

int32_t lfhe_bsk_image_form(const lfhe_ctx* ctx, uint32_t* out) {
    if (!ctx || !out) return LFHE_ERR_ARG;
    *out = ctx->eng->bsk_factored() ? 1u : 0u;
    return LFHE_OK;
}

int32_t lfhe_lift_fallback_count(lfhe_ctx* ctx, uint64_t* out) {
    return guarded(ctx, [&]() -> int32_t {
        if (!out) throw std::invalid_argument("null argument");
        *out = ctx->eng->lift_fallbacks();
        return LFHE_OK;
    });
}

}  // extern "C"
