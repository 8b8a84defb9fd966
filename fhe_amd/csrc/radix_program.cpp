#include "radix_program.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <functional>
#include <sstream>

#include "../../include/lfhe.h"

namespace lfhe {

// ------------------------------------------------------------------------------------------------
// builder
// ------------------------------------------------------------------------------------------------
Builder::Info Builder::eval(const Expr& e) const {
    Info r{0, e.cst, e.cst, 0};
    for (const auto& [c, b] : e.t) {
        const Info& i = info_[b];
        r.level = std::max(r.level, i.level);
        if (c >= 0) { r.lo += c * i.lo; r.hi += c * i.hi; }
        else { r.lo += c * i.hi; r.hi += c * i.lo; }
        r.noise += std::abs(c) * i.noise;
    }
    return r;
}

uint32_t Builder::lut_id(const std::function<int(int)>& f, bool wraps) {
    std::array<uint64_t, 16> tab;
    for (int x = 0; x < 16; ++x) {
        const int v = f(x);
        // a value of 16 and more has the padding bit set: for the input of a wrapped bootstrap only
        if (wraps ? (v < -BLOCK_MAX || v > BLOCK_MAX) : (v < 0 || v > 2 * BLOCK_MAX + 1))
            throw std::logic_error("lookup table value outside the block");
        tab[x] = (uint64_t)((v + 2 * (BLOCK_MAX + 1)) % (2 * (BLOCK_MAX + 1)));
    }
    auto it = lut_ids_.find({tab, wraps});
    if (it != lut_ids_.end()) return it->second;
    const uint32_t id = (uint32_t)P.luts.size();
    P.luts.push_back(tab);
    P.lut_wraps.push_back(wraps ? 1 : 0);
    lut_ids_[{tab, wraps}] = id;
    return id;
}

uint32_t Builder::lin(const Expr& e0, const Info& i, int limit) {
    // the linear operations of one layer run side by side in one launch: a term that is itself the
    // output of a linear operation of this layer is replaced by the combination that defines it
    Expr e;
    e.cst = e0.cst;
    for (const auto& [c, b] : e0.t) {
        auto it = lin_expr_.find(b);
        if (it == lin_expr_.end() || info_[b].level != i.level) { e.t.push_back({c, b}); continue; }
        for (const auto& [c2, b2] : it->second.t) e.t.push_back({c * c2, b2});
        e.cst += c * it->second.cst;
    }
    if (e.t.size() > (size_t)LIN_MAX_TERMS) throw std::logic_error("linear combination too long");
    if (i.lo < 0 || i.hi > limit) throw std::logic_error("block value range overflows the carry space");
    if (i.noise > MAX_NOISE) throw std::logic_error("noise level above the parameter set's budget");
    const uint32_t out = new_block(i.level, i.lo, i.hi, i.noise);
    LinOp op{};
    op.out = out;
    op.n = (uint32_t)e.t.size();
    for (size_t k = 0; k < e.t.size(); ++k) { op.coef[k] = e.t[k].first; op.src[k] = e.t[k].second; }
    op.cst = e.cst;
    layer(i.level).lin.push_back(op);
    lin_expr_[out] = e;
    return out;
}

uint32_t Builder::mat(const Expr& e, int hi_known, int lo_known, int limit) {
    if (e.t.size() == 1 && e.t[0].first == 1 && e.cst == 0 && info_[e.t[0].second].lo >= 0) return e.t[0].second;
    Info i = eval(e);
    if (hi_known >= 0) i.hi = std::min(i.hi, hi_known);
    if (lo_known >= 0) i.lo = std::max(i.lo, lo_known);
    return lin(e, i, limit);
}

uint32_t Builder::pbs_wrapped(const Expr& e, const std::function<int(int)>& f) {
    const int space = BLOCK_MAX + 1;
    const Info i = eval(e);
    if (i.lo < 0 || i.hi >= 2 * space) throw std::logic_error("bootstrap input range overflows the padding bit");
    if (i.noise > MAX_NOISE) throw std::logic_error("bootstrap input noise above the parameter set's budget");
    const uint32_t in = mat(e, -1, -1, 2 * space - 1);
    int lo = space, hi = -space;
    for (int x = i.lo; x <= i.hi; ++x) {
        const int v = x < space ? f(x) : -f(x - space);
        lo = std::min(lo, v);
        hi = std::max(hi, v);
    }
    const uint32_t out = new_block(i.level + 1, lo, hi, 1);
    layer(i.level).pbs.push_back({in, out, lut_id(f, true)});
    return out;
}

uint32_t Builder::pbs(const Expr& e, const std::function<int(int)>& f) {
    const Info i = eval(e);
    if (i.lo < 0 || i.hi > BLOCK_MAX) throw std::logic_error("bootstrap input range overflows the carry space");
    if (i.noise > MAX_NOISE) throw std::logic_error("bootstrap input noise above the parameter set's budget");
    const uint32_t in = mat(e);
    int lo = BLOCK_MAX, hi = 0;
    for (int x = i.lo; x <= i.hi; ++x) { lo = std::min(lo, f(x)); hi = std::max(hi, f(x)); }
    const uint32_t out = new_block(i.level + 1, lo, hi, 1);
    layer(i.level).pbs.push_back({in, out, lut_id(f)});
    made_by_[out] = {in, f};
    return out;
}

// A table of a block that is itself the plain output of a bootstrap is a table of that bootstrap's
// input: one bootstrap, one layer earlier (the first one is pruned if nothing else reads it).  Not
// done behind the caller's back in pbs(): moving a bootstrap one layer up can push a layer over the
// number of ciphertexts the GPU holds at once (1024), which costs more than the layer saved.
uint32_t Builder::pbs_of(uint32_t b, const std::function<int(int)>& g) {
    auto it = made_by_.find(b);
    if (it == made_by_.end()) return pbs(E(b), g);
    const uint32_t in = it->second.first;
    const auto f = it->second.second;
    return pbs(E(in), [f, g](int x) { return g(f(x)); });
}

// Operations nothing reads are dropped: an operator may ask for more than its result needs (the low
// digits of a product whose upper half is the result, the carry states of blocks a shift discards).
// Block ids stay as they are.
void Builder::prune() {
    std::vector<char> live(P.nblocks, 0);
    for (uint32_t b : P.out) live[b] = 1;
    for (const auto& o : P.outs)
        for (uint32_t b : o) live[b] = 1;
    for (size_t l = P.layers.size(); l-- > 0;) {
        Layer& L = P.layers[l];
        std::vector<PbsOp> pbs;
        for (const PbsOp& q : L.pbs)
            if (live[q.out]) { pbs.push_back(q); live[q.in] = 1; }
        L.pbs.swap(pbs);
        // a linear operation of this layer may read another one of this layer only through the
        // combination it was flattened to, so one backward pass over the list is enough
        std::vector<LinOp> lin;
        for (size_t i = L.lin.size(); i-- > 0;) {
            const LinOp& q = L.lin[i];
            if (!live[q.out]) continue;
            for (uint32_t k = 0; k < q.n; ++k) live[q.src[k]] = 1;
            lin.push_back(q);
        }
        std::reverse(lin.begin(), lin.end());
        L.lin.swap(lin);
    }
    while (!P.layers.empty() && P.layers.back().pbs.empty() && P.layers.back().lin.empty()) P.layers.pop_back();
}

// ------------------------------------------------------------------------------------------------
// Layer balancing.  The builder places every bootstrap in the earliest layer its inputs allow.  For
// ONE instance of an operator a layer costs by the number of ciphertexts the fullest compute unit
// gets: up to `cap` (one per unit) a layer costs the same whatever it holds, and the cost rises in
// steps at every further multiple.  Many bootstraps of a large early layer are not needed by the
// next layer at all (the partial products of the low columns of a multiplication, the groups of a
// column sum that only the last rounds read): every bootstrap gets a deadline -- the last layer
// that still lets its consumers meet theirs, the operator's depth staying what it is -- and a list
// scheduler fills each layer with what is due, rounded up to a multiple of `cap`, most urgent first.
// Two fillings are tried (the first layer as built, or cut like the others); the cheaper one under
// the step model replaces the program if it beats it, and only if the rebuilt program passes a
// dependency check.  The bootstraps and their tables are untouched: for large batches, where a
// layer's cost is proportional to its size, nothing changes.
// ------------------------------------------------------------------------------------------------
void Builder::balance(size_t cap) {
    if (!cap || P.layers.size() < 3) return;
    const size_t L = P.layers.size();
    struct Pb { PbsOp op; uint32_t layer; };
    struct Ln { LinOp op; uint32_t layer; };
    std::vector<Pb> pbs;
    std::vector<Ln> lin;
    std::vector<int> made(P.nblocks, -1);   // -1 input, 2 i: bootstrap i, 2 i + 1: linear operation i
    for (size_t l = 0; l < L; ++l) {
        for (const LinOp& q : P.layers[l].lin) { made[q.out] = 2 * (int)lin.size() + 1; lin.push_back({q, (uint32_t)l}); }
        for (const PbsOp& q : P.layers[l].pbs) { made[q.out] = 2 * (int)pbs.size(); pbs.push_back({q, (uint32_t)l}); }
    }
    const size_t n = pbs.size();
    if (!n) return;
    // the bootstraps a block's value depends on directly (through linear operations)
    std::vector<std::vector<uint32_t>> block_preds(P.nblocks);
    std::vector<char> known(P.nblocks, 0);
    std::function<const std::vector<uint32_t>&(uint32_t)> preds_of = [&](uint32_t b) -> const std::vector<uint32_t>& {
        if (known[b]) return block_preds[b];
        known[b] = 1;
        const int m = made[b];
        if (m >= 0 && !(m & 1)) block_preds[b] = {(uint32_t)(m / 2)};
        else if (m >= 0) {
            std::vector<uint32_t> acc;
            const LinOp& q = lin[(size_t)(m / 2)].op;
            for (uint32_t k = 0; k < q.n; ++k) { const auto& r = preds_of(q.src[k]); acc.insert(acc.end(), r.begin(), r.end()); }
            std::sort(acc.begin(), acc.end());
            acc.erase(std::unique(acc.begin(), acc.end()), acc.end());
            block_preds[b] = std::move(acc);
        }
        return block_preds[b];
    };
    std::vector<std::vector<uint32_t>> preds(n), succ(n);
    for (size_t i = 0; i < n; ++i) {
        preds[i] = preds_of(pbs[i].op.in);
        for (uint32_t j : preds[i]) succ[j].push_back((uint32_t)i);
    }
    const uint32_t never = 1u << 30;
    std::vector<uint32_t> due(n, never);
    // an output made by a linear operation needs its bootstraps one layer before the last
    auto final_block = [&](uint32_t b) {
        const uint32_t last = (made[b] >= 0 && (made[b] & 1)) ? (uint32_t)L - 2 : (uint32_t)L - 1;
        for (uint32_t j : preds_of(b)) due[j] = std::min(due[j], last);
    };
    for (uint32_t b : P.out) final_block(b);
    for (const auto& o : P.outs) for (uint32_t b : o) final_block(b);
    for (size_t i = n; i-- > 0;) {   // consumers sit in later layers, hence later in the list
        for (uint32_t s : succ[i]) due[i] = std::min(due[i], due[s] - 1);
        if (due[i] == never) due[i] = (uint32_t)L - 1;
        if (due[i] < pbs[i].layer) return;   // cannot be: the program as built meets its own depth
    }
    auto cost = [cap](size_t x) {
        const double c = (double)cap;
        if (!x) return 0.0;
        if (x <= cap) return 1.0;
        if (x <= 2 * cap) return 1.9;
        if (x <= 3 * cap) return 2.85;
        if (x <= 4 * cap) return 3.35;
        return 3.35 * (double)x / (4 * c);
    };
    auto total = [&](const std::vector<uint32_t>& where) {
        std::vector<size_t> sizes(L, 0);
        for (uint32_t l : where) sizes[l] += 1;
        double t = 0;
        for (size_t x : sizes) t += cost(x);
        return t;
    };
    auto schedule = [&](bool first_as_built, std::vector<uint32_t>& where) {
        where.assign(n, never);
        std::vector<uint32_t> waiting(n);
        for (size_t i = 0; i < n; ++i) waiting[i] = (uint32_t)preds[i].size();
        std::vector<uint32_t> ready, next;
        for (size_t i = 0; i < n; ++i) if (!waiting[i]) ready.push_back((uint32_t)i);
        size_t placed = 0;
        for (uint32_t l = 0; l < L && placed < n; ++l) {
            std::sort(ready.begin(), ready.end(), [&](uint32_t a, uint32_t b) {
                return due[a] != due[b] ? due[a] < due[b] : (pbs[a].layer != pbs[b].layer ? pbs[a].layer < pbs[b].layer : a < b);
            });
            size_t must = 0;
            while (must < ready.size() && due[ready[must]] <= l) ++must;
            size_t take = std::max(cap, (must + cap - 1) / cap * cap);
            if (l == 0 && first_as_built) take = ready.size();
            take = std::min(take, ready.size());
            next.assign(ready.begin() + (long)take, ready.end());
            for (size_t k = 0; k < take; ++k) {
                const uint32_t i = ready[k];
                where[i] = l;
                ++placed;
            }
            for (size_t k = 0; k < take; ++k)
                for (uint32_t s : succ[ready[k]])
                    if (--waiting[s] == 0) next.push_back(s);
            ready.swap(next);
        }
        return placed == n;
    };
    std::vector<uint32_t> as_built(n), best, trial;
    for (size_t i = 0; i < n; ++i) as_built[i] = pbs[i].layer;
    double best_cost = total(as_built);
    for (int variant = 0; variant < 2; ++variant)
        if (schedule(variant == 1, trial) && total(trial) < best_cost - 1e-9) { best_cost = total(trial); best = trial; }
    if (best.empty()) return;

    // rebuild: linear operations move to the first layer that has all their sources; one that now
    // shares its layer with a linear operation it reads takes that one's combination in its place
    std::vector<uint32_t> lin_layer(lin.size(), 0);
    std::vector<std::vector<std::pair<int64_t, uint32_t>>> terms(lin.size());
    std::vector<int64_t> consts(lin.size());
    for (size_t a = 0; a < lin.size(); ++a) {
        const LinOp& q = lin[a].op;
        uint32_t at = 0;
        for (uint32_t k = 0; k < q.n; ++k) {
            const int m = made[q.src[k]];
            if (m >= 0 && !(m & 1)) at = std::max(at, best[(size_t)(m / 2)] + 1);
            else if (m >= 0) { if ((size_t)(m / 2) >= a) return; at = std::max(at, lin_layer[(size_t)(m / 2)]); }
        }
        std::map<uint32_t, int64_t> acc;
        int64_t c = q.cst;
        for (uint32_t k = 0; k < q.n; ++k) {
            const int m = made[q.src[k]];
            if (m >= 0 && (m & 1) && lin_layer[(size_t)(m / 2)] == at) {
                const size_t b = (size_t)(m / 2);
                for (const auto& t : terms[b]) acc[t.second] += (int64_t)q.coef[k] * t.first;
                c += (int64_t)q.coef[k] * consts[b];
            } else {
                acc[q.src[k]] += q.coef[k];
            }
        }
        for (const auto& t : acc) if (t.second) terms[a].push_back({t.second, t.first});
        if (terms[a].size() > (size_t)LIN_MAX_TERMS) return;
        consts[a] = c;
        lin_layer[a] = at;
    }
    uint32_t depth = 0;
    for (size_t i = 0; i < n; ++i) depth = std::max(depth, best[i] + 1);
    for (size_t a = 0; a < lin.size(); ++a) depth = std::max(depth, lin_layer[a] + 1);
    if (depth > L) return;
    // dependency check of what is about to be written
    for (size_t i = 0; i < n; ++i) {
        const int m = made[pbs[i].op.in];
        if (m >= 0 && !(m & 1) && best[(size_t)(m / 2)] + 1 > best[i]) return;
        if (m >= 0 && (m & 1) && lin_layer[(size_t)(m / 2)] > best[i]) return;
    }
    for (size_t a = 0; a < lin.size(); ++a)
        for (const auto& t : terms[a]) {
            const int m = made[t.second];
            if (m >= 0 && !(m & 1) && best[(size_t)(m / 2)] + 1 > lin_layer[a]) return;
            if (m >= 0 && (m & 1) && lin_layer[(size_t)(m / 2)] >= lin_layer[a]) return;
            if (t.first > INT32_MAX || t.first < INT32_MIN) return;
        }
    std::vector<Layer> layers(L);
    for (size_t a = 0; a < lin.size(); ++a) {
        LinOp q{};
        q.out = lin[a].op.out;
        q.n = (uint32_t)terms[a].size();
        for (size_t k = 0; k < terms[a].size(); ++k) { q.coef[k] = (int32_t)terms[a][k].first; q.src[k] = terms[a][k].second; }
        if (consts[a] > INT32_MAX || consts[a] < INT32_MIN) return;
        q.cst = (int32_t)consts[a];
        layers[lin_layer[a]].lin.push_back(q);
    }
    for (size_t i = 0; i < n; ++i) layers[best[i]].pbs.push_back(pbs[i].op);
    P.layers.swap(layers);
    prune();   // linear operations whose combination was taken over may have no reader left
}

static size_t layer_cap() {
    static const size_t cap = [] {
        const char* e = std::getenv("LFHE_LAYER_CAP");   // 0: programs stay as built
        return e ? (size_t)std::atol(e) : (size_t)256;   // one ciphertext per compute unit of an MI355X
    }();
    return cap;
}

void Builder::finish() {
    prune();
    balance(layer_cap());
    int noise = 0, deg = 0;
    for (uint32_t b : P.out) { noise = std::max(noise, info_[b].noise); deg = std::max(deg, info_[b].hi); }
    P.out_noise = (uint8_t)std::max(noise, 1);
    P.out_degree = (uint8_t)deg;
    if (deg > 3) throw std::logic_error("operator output is not carry-free");
}

std::string Program::to_json() const {
    std::ostringstream o;
    auto list = [&](const std::vector<uint32_t>& v) {
        o << "[";
        for (size_t i = 0; i < v.size(); ++i) o << (i ? "," : "") << v[i];
        o << "]";
    };
    o << "{\"op\":" << op << ",\"width\":" << width << ",\"out_width\":" << out_width << ",\"rhs_width\":" << rhs_width << ",\"nblocks\":" << nblocks
      << ",\"out_noise\":" << (int)out_noise << ",\"pbs\":" << pbs_per_instance() << ",\"lhs\":";
    list(lhs); o << ",\"rhs\":"; list(rhs); o << ",\"cond\":"; list(cond); o << ",\"out\":"; list(out);
    auto lists = [&](const char* key, const std::vector<std::vector<uint32_t>>& vv) {
        o << ",\"" << key << "\":[";
        for (size_t i = 0; i < vv.size(); ++i) { o << (i ? "," : ""); list(vv[i]); }
        o << "]";
    };
    lists("ins", ins); lists("outs", outs);
    o << ",\"in_widths\":"; list(in_widths); o << ",\"out_widths\":"; list(out_widths);
    o << ",\"luts\":[";
    for (size_t i = 0; i < luts.size(); ++i) {
        o << (i ? "," : "") << "[";
        for (int x = 0; x < 16; ++x) o << (x ? "," : "") << luts[i][x];
        o << "]";
    }
    o << "],\"lut_wraps\":[";
    for (size_t i = 0; i < lut_wraps.size(); ++i) o << (i ? "," : "") << (int)lut_wraps[i];
    o << "],\"layers\":[";
    for (size_t l = 0; l < layers.size(); ++l) {
        o << (l ? "," : "") << "{\"lin\":[";
        for (size_t i = 0; i < layers[l].lin.size(); ++i) {
            const LinOp& q = layers[l].lin[i];
            o << (i ? "," : "") << "{\"out\":" << q.out << ",\"cst\":" << q.cst << ",\"terms\":[";
            for (uint32_t k = 0; k < q.n; ++k) o << (k ? "," : "") << "[" << q.coef[k] << "," << q.src[k] << "]";
            o << "]}";
        }
        o << "],\"pbs\":[";
        for (size_t i = 0; i < layers[l].pbs.size(); ++i) {
            const PbsOp& q = layers[l].pbs[i];
            o << (i ? "," : "") << "[" << q.in << "," << q.out << "," << q.lut << "]";
        }
        o << "]}";
    }
    o << "]}";
    return o.str();
}

// ------------------------------------------------------------------------------------------------
// operator algorithms
// ------------------------------------------------------------------------------------------------
namespace {

using Expr = Builder::Expr;
typedef std::vector<uint32_t> Blocks;

int hi2(int x) { return x >> 2; }
int lo2(int x) { return x & 3; }

// block values are made carry-free and brought back to one bootstrap's noise when an operator
// packs two blocks into one bootstrap input (4*x + y needs x at noise 1)
Blocks clean(Builder& B, const Blocks& v) {
    Blocks r;
    for (uint32_t b : v) r.push_back((B.noise(b) > 1 || B.hi(b) > 3) ? B.pbs(B.E(b), lo2) : b);
    return r;
}

// only the blocks that hold carries are bootstrapped (for operators that take the difference of two
// blocks: a - b + 3 costs the sum of the noise levels, nothing is scaled)
Blocks carry_free(Builder& B, const Blocks& v) {
    Blocks r;
    for (uint32_t b : v) r.push_back(B.hi(b) > 3 ? B.pbs(B.E(b), lo2) : b);
    return r;
}
Expr diff3(Builder& B, uint32_t a, uint32_t b) { return Builder::add(Builder::sub(B.E(a), B.E(b)), Builder::C(3)); }

Expr pack(Builder& B, uint32_t hi_blk, uint32_t lo_blk) { return Builder::add(Builder::scale(B.E(hi_blk), 4), B.E(lo_blk)); }

// Three-valued states (0 kill, 1 pass, 2 set) and their composition "the most significant state
// that is not `pass` decides" -- carry states of an addition (none / propagate / generate), signs of a
// comparison (less / equal / greater).  With the state at position k (0 = least significant) held
// as state * 2^k, the plain sum of the blocks is the addition the states describe -- `pass` is a
// one bit that hands a carry on, `set` a carry into the next position -- so g composed states are
// set when the sum reaches 2^g, pass when it is 2^g - 1 (all pass), kill otherwise.
//   * Three states: the largest sum, 2 + 4 + 8, fits a block.
//   * FOUR states: the sum reaches 30 and runs into the padding bit, where a blind rotation returns
//     minus the table entry of x - 16.  The table holds -1 below 15 and 0 at 15, so the rotation
//     returns +1 from 16 on: state - 1, and the 1 is a constant added afterwards (Builder::
//     pbs_wrapped).  A node made this way is `wrapped`.
//   * A state that is never `pass` (a resolved carry) can join either sum below position 0 as the
//     single bit "carry in": 1 + 2 + 4 + 8 and 1 + 2 + 4 + 8 + 16 still fit, and the result is again
//     resolved.  The bit is half a state: only a node that is not wrapped can hand it out; a
//     wrapped resolved carry takes the least significant state position instead.
// A state is a function of one bootstrap input, so the table that produces it can hand it out at
// whatever position its consumer needs; a second consumer at another position costs a second
// bootstrap of the same input in the same layer, never an extra layer.
class States {
public:
    explicit States(Builder& b) : B(b) {}
    int leaf(const Expr& in, std::function<int(int)> fn) {
        nodes.push_back({in, std::move(fn), false, {-1, -1, -1, -1, -1}});
        return (int)nodes.size() - 1;
    }
    bool wrapped(int node) const { return nodes[node].wrapped; }
    // how many state positions combine(states, carry) needs
    size_t positions(size_t nstates, int carry) const { return nstates + (carry >= 0 && nodes[carry].wrapped ? 1 : 0); }
    // states[0] is the most significant; carry (optional): a resolved state entering below states.back()
    int combine(std::vector<int> states, int carry = -1) {
        Expr in;
        if (carry >= 0) {
            if (nodes[carry].wrapped) states.push_back(carry);
            else in = at(carry, -1);
        }
        const int g = (int)states.size();
        if (g > 4) throw std::logic_error("more than four states in one bootstrap");
        for (int j = 0; j < g; ++j) in = Builder::add(in, at(states[j], g - 1 - j));
        const bool resolved = carry >= 0;
        if (g <= 3) return leaf(in, [g, resolved](int x) { return x >= (1 << g) ? 2 : (x == (1 << g) - 1 && !resolved ? 1 : 0); });
        nodes.push_back({in, [resolved](int x) { return x == 15 && !resolved ? 0 : -1; }, true, {-1, -1, -1, -1, -1}});
        return (int)nodes.size() - 1;
    }
    // state * 2^k as an expression; k = -1 (resolved states that are not wrapped): the bit "set"
    Expr at(int node, int k) {
        Node& n = nodes[node];
        if (n.wrapped && k < 0) throw std::logic_error("a wrapped state has no carry bit");
        if (n.blk[k + 1] < 0) {
            const auto fn = n.fn;
            if (n.wrapped) n.blk[k + 1] = (int)B.pbs_wrapped(n.in, [fn, k](int x) { return fn(x) * (1 << k); });
            else n.blk[k + 1] = (int)B.pbs(n.in, [fn, k](int x) { return k < 0 ? fn(x) >> 1 : fn(x) << k; });
        }
        const Expr e = B.E((uint32_t)nodes[node].blk[k + 1]);
        return nodes[node].wrapped ? Builder::add(e, Builder::C(1 << k)) : e;
    }
    // a bootstrap of the state's own input through g(state)
    uint32_t map(int node, std::function<int(int)> g) {
        if (nodes[node].wrapped) throw std::logic_error("map of a wrapped state");
        const auto fn = nodes[node].fn;
        return B.pbs(nodes[node].in, [fn, g](int x) { return g(fn(x)); });
    }

private:
    struct Node {
        Expr in;
        std::function<int(int)> fn;  // the state on [0,16); wrapped: state - 1
        bool wrapped;
        int blk[5];  // state * 2^k for k = -1 .. 3
    };
    Builder& B;
    std::vector<Node> nodes;
};

// Carry propagation over per-block sums v[i] in [0,7] (v[i] + carry_in[i] <= 7): digit i =
// (v[i] + carry_in[i]) mod 4, carry_out[i] = (v[i] + carry_in[i]) >= 4.  One layer extracts digit and
// carry state (none, propagate, generate); position 0 has no carry in, so its state is resolved from
// the start.  A prefix network in base four then grows two things at once.  With q = position - 1
// and F the number of q whose carry out is resolved, level L (w = 4^L) holds for every q >= F the
// state of the run from the start of q's group of width w (groups counted from F) up to q.  The next
// level resolves q in [F + w, F + R w) from the resolved carry into F and the states of up to R
// groups (R = 4 when the carry can enter as a bit, else 3), and composes the states beyond into
// groups of width 4 w.  The last bootstrap of a position adds its incoming carry to the digit; it
// has room for a resolved carry AND one unresolved state on top (digit + 4 (carry + state), carry in
// = the sum reaches 2), so q in [F, F + w) never need a resolved carry of their own.  Two prefix
// levels serve 32 blocks, three serve 128, and none of them waits for the others to finish: blocks
// whose sums arrive early (the low columns of a product) resolve early.
// prefix levels still needed when q < F are resolved and states span w; bit: the carry into F can be
// handed out as a bit.  *R: the number of groups to resolve at this level.
int prefix_levels(size_t F, size_t w, bool bit, size_t nq, int* R) {
    if (F + w >= nq) return 0;
    int best = 1 << 20;
    for (int r = bit ? 4 : 3; r >= 2; --r) {
        // the carry into the new F composes r states and the carry: wrapped when that needs four positions
        const bool next_bit = r + (bit ? 0 : 1) <= 3;
        const int l = 1 + prefix_levels(F + r * w, 4 * w, next_bit, nq, nullptr);
        if (l < best) { best = l; if (R) *R = r; }
    }
    return best;
}

Blocks propagate(Builder& B, const std::vector<Expr>& v) {
    const size_t m = v.size();
    Blocks res(m), msg(m);
    for (size_t i = 0; i < m; ++i) msg[i] = B.pbs(v[i], lo2);
    if (m) res[0] = msg[0];
    if (m < 2) return res;
    States S(B);
    const int c0 = S.leaf(v[0], [](int x) { return x >= 4 ? 2 : 0; });
    const size_t nq = m - 2;  // q = 0 .. nq-1: positions 1 .. m-2, the ones whose carry out is used
    std::vector<int> st(nq);
    for (size_t q = 0; q < nq; ++q) st[q] = S.leaf(v[q + 1], [](int x) { return x >= 4 ? 2 : (x == 3 ? 1 : 0); });
    // what the last bootstrap of a position adds to its digit: a resolved carry, and a state above it
    struct CarryIn { int carry, state; };
    std::vector<CarryIn> cin(m, CarryIn{-1, -1});
    cin[1] = {c0, -1};
    size_t F = 0;
    int carry = c0;  // the resolved carry into q = F
    for (size_t w = 1; F < nq; w *= 4) {
        for (size_t q = F; q < std::min(F + w, nq); ++q) cin[q + 2] = {carry, st[q]};
        int R = 0;
        if (prefix_levels(F, w, !S.wrapped(carry), nq, &R) == 0) break;
        std::vector<int> nx = st;
        for (size_t q = F + w; q < nq; ++q) {
            const bool resolve = q < F + R * w;
            const size_t start = resolve ? F : F + R * w + (q - F - R * w) / (4 * w) * (4 * w);
            const size_t part = (q - start) / w;
            std::vector<int> run{st[q]};
            for (size_t k = part; k >= 1; --k) run.push_back(st[start + k * w - 1]);
            if (resolve) {
                nx[q] = S.combine(run, carry);
                cin[q + 2] = {nx[q], -1};
            } else if (part >= 1) {
                nx[q] = S.combine(run);
            }
        }
        st.swap(nx);
        F += R * w;
        if (F - 1 < nq) carry = st[F - 1];
    }
    for (size_t i = 1; i < m; ++i) {
        // above the digit: carry bit + state (0 .. 3), or the carry alone as the state `set` (0 / 2)
        Expr in = B.E(msg[i]);
        if (cin[i].state >= 0) in = Builder::add(Builder::add(in, S.at(cin[i].carry, 1)), S.at(cin[i].state, 2));
        else in = Builder::add(in, S.at(cin[i].carry, 2));
        res[i] = B.pbs(in, [](int x) { return (lo2(x) + (hi2(x) >= 2 ? 1 : 0)) & 3; });
    }
    return res;
}

Blocks op_add(Builder& B, const Blocks& a, const Blocks& b) {
    std::vector<Expr> v;
    for (size_t i = 0; i < a.size(); ++i) v.push_back(Builder::add(B.E(a[i]), B.E(b[i])));
    return propagate(B, v);
}

// a - b = a + ~b + 1 blockwise: v[i] = a[i] + (3 - b[i]) (+1 in block 0)
Blocks op_sub(Builder& B, const Blocks& a, const Blocks& b) {
    std::vector<Expr> v;
    for (size_t i = 0; i < a.size(); ++i)
        v.push_back(Builder::add(Builder::sub(B.E(a[i]), B.E(b[i])), Builder::C(i == 0 ? 4 : 3)));
    return propagate(B, v);
}

Blocks op_neg(Builder& B, const Blocks& a) {
    std::vector<Expr> v;
    for (size_t i = 0; i < a.size(); ++i) v.push_back(Builder::sub(Builder::C(i == 0 ? 4 : 3), B.E(a[i])));
    return propagate(B, v);
}

Blocks op_bitwise(Builder& B, const Blocks& a0, const Blocks& b0, uint32_t op) {
    const Blocks a = clean(B, a0), b = clean(B, b0);
    Blocks r;
    for (size_t i = 0; i < a.size(); ++i)
        r.push_back(B.pbs(pack(B, a[i], b[i]), [op](int x) {
            const int p = hi2(x), q = lo2(x);
            return op == LFHE_OP_AND ? (p & q) : op == LFHE_OP_OR ? (p | q) : (p ^ q);
        }));
    return r;
}

// Column sums of a schoolbook product.  A group of up to five blocks of one column (sum at most 15)
// is replaced by its digit (same column) and its carry (next column): g blocks become one here and
// one there.  The heights come down in rounds to the targets 2 < 5 < 11 < 26 < 65 < ... (a column of
// height T + 3 floor(T / 2) can be brought to T when its neighbour sends it floor(T / 2) carries),
// as in Dadda's reduction: a round looks at the columns from the least significant one up, counts
// the carries the column below sends in this very round, and forms only the groups the target asks
// for.  Counting them in the same round is what keeps a column of two blocks that receives a carry
// from pushing the problem one column up per round.  At the end every column holds at most two
// blocks; columns beyond the last are dropped (arithmetic modulo 4^columns).
// (the digits of the sum: the reduction and one carry propagation; `minuend`, if given, turns the
// result into minuend - sum: see subtract_columns)
Blocks subtract_columns(Builder& B, const Blocks& minuend, const std::vector<Blocks>& cols);
Blocks sum_columns(Builder& B, const std::vector<Blocks>& cols0, const Blocks* minuend = nullptr) {
    const size_t m = cols0.size();
    // one round on heights alone: the sizes of the groups each column forms
    auto plan_round = [m](std::vector<size_t>& h, size_t target, std::vector<std::vector<size_t>>& groups) {
        std::vector<size_t> nh(m, 0);
        groups.assign(m, {});
        for (size_t c = 0; c < m; ++c) {
            const size_t own = h[c], tall = own + nh[c];   // nh[c] holds the carries of column c - 1 by now
            size_t need = tall > target ? tall - target : 0, used = 0;
            while (need > 0 && own - used >= 2) {
                const size_t g = std::min<size_t>({5, own - used, need + 1});
                groups[c].push_back(g);
                used += g;
                need -= g - 1;
                nh[c] += 1;
                if (c + 1 < m) nh[c + 1] += 1;
            }
            nh[c] += own - used;
        }
        h.swap(nh);
    };
    auto tallest = [](const std::vector<size_t>& h) { return h.empty() ? (size_t)0 : *std::max_element(h.begin(), h.end()); };
    auto group = [m](Builder& T, const std::vector<Blocks>& cols, std::vector<Blocks>& nx, size_t c, size_t k, size_t g) {
        Expr s;
        for (size_t u = 0; u < g; ++u) s = Builder::add(s, T.E(cols[c][k + u]));
        nx[c].push_back(T.pbs(s, lo2));
        if (c + 1 < m) nx[c + 1].push_back(T.pbs(s, hi2));
    };
    // the whole reduction under one list of targets
    auto reduce_with = [&](Builder& T, const std::vector<size_t>& targets, std::vector<Blocks>& cols) {
        std::vector<size_t> h(m);
        for (size_t c = 0; c < m; ++c) h[c] = cols[c].size();
        size_t ti = 0;
        for (size_t rounds = 0; tallest(h) > 2; ++rounds) {
            if (rounds > 4 * m + 16) return false;
            while (ti + 1 < targets.size() && tallest(h) <= targets[ti]) ++ti;
            std::vector<std::vector<size_t>> groups;
            plan_round(h, tallest(h) > targets[ti] ? targets[ti] : 2, groups);
            std::vector<Blocks> nx(m);
            for (size_t c = 0; c < m; ++c) {
                size_t k = 0;
                for (size_t g : groups[c]) { group(T, cols, nx, c, k, g); k += g; }
                nx[c].insert(nx[c].end(), cols[c].begin() + (long)k, cols[c].end());
            }
            cols.swap(nx);
        }
        return true;
    };
    // every column of more than two blocks grouped whole, five at a time, round after round (carries
    // of a round are not counted in that round: columns of two that receive one ripple)
    auto reduce_whole = [&](Builder& T, std::vector<Blocks>& cols) {
        for (size_t rounds = 0;; ++rounds) {
            size_t longest = 0;
            for (const auto& c : cols) longest = std::max(longest, c.size());
            if (longest <= 2) return true;
            if (rounds > 4 * m + 16) return false;
            std::vector<Blocks> nx(m);
            for (size_t c = 0; c < m; ++c) {
                const Blocks& t = cols[c];
                if (t.size() <= 2) { nx[c].insert(nx[c].end(), t.begin(), t.end()); continue; }
                for (size_t k = 0; k < t.size(); k += 5) {
                    const size_t g = std::min<size_t>(5, t.size() - k);
                    if (g == 1) nx[c].push_back(t[k]);
                    else group(T, cols, nx, c, k, g);
                }
            }
            cols.swap(nx);
        }
    };
    // The targets that fit uniform columns are not always the best for a triangle of columns whose
    // low end is ready early, and what counts is the depth after the carry propagation: every tail
    // of the sequence (down to "as far as possible in every round") and the whole-column grouping are
    // built on copies of the builder, carry propagation included; the shallowest result wins, then
    // the one with the fewest bootstraps.
    size_t top = 0;
    for (const auto& c : cols0) top = std::max(top, c.size());
    std::vector<size_t> seq{2};
    while (seq.back() < top) seq.push_back(seq.back() + 3 * (seq.back() / 2));
    bool have = false;
    Builder best_builder;
    Blocks best;
    std::array<uint64_t, 2> best_cost{};
    for (size_t len = 0; len <= seq.size(); ++len) {
        Builder trial = B;
        std::vector<Blocks> cols = cols0;
        if (len == 0 ? !reduce_whole(trial, cols)
                     : !reduce_with(trial, std::vector<size_t>(seq.rend() - (long)len, seq.rend()), cols)) continue;
        Blocks r;
        if (minuend) {
            r = subtract_columns(trial, *minuend, cols);
        } else {
            std::vector<Expr> v;
            for (size_t c = 0; c < m; ++c) {
                Expr e;
                for (uint32_t blk : cols[c]) e = Builder::add(e, trial.E(blk));
                v.push_back(e);
            }
            r = propagate(trial, v);
        }
        uint64_t deepest = 0;
        for (uint32_t blk : r) deepest = std::max<uint64_t>(deepest, (uint64_t)trial.level(blk));
        const std::array<uint64_t, 2> cost{deepest, trial.P.pbs_per_instance()};
        if (!have || cost < best_cost) { have = true; best_cost = cost; best_builder = std::move(trial); best = std::move(r); }
    }
    if (!have) throw std::logic_error("column reduction did not converge");
    B = std::move(best_builder);
    return best;
}

// a - P modulo 4^m, where P is given as columns of at most two carry-free blocks (the state a column
// reduction ends in) and a as carry-free blocks: with t_c the sum of column c (at most 6),
//   a - P = sum 4^c (a_c + 6 - t_c) - 6 (4^m - 1) / 3 = sum 4^c (a_c + 6 - t_c) + 2   (mod 4^m),
// so one round of digit + carry over x_c = a_c + 6 - t_c (+ 2 in column 0; at most 11) leaves columns of
// two blocks again, and ONE carry propagation finishes: P never has to be propagated on its own.
Blocks subtract_columns(Builder& B, const Blocks& a, const std::vector<Blocks>& cols) {
    const size_t m = cols.size();
    std::vector<Expr> v(m);
    for (size_t c = 0; c < m; ++c) {
        Expr x = Builder::add(B.E(a[c]), Builder::C(c == 0 ? 8 : 6));
        for (uint32_t blk : cols[c]) x = Builder::sub(x, B.E(blk));
        v[c] = Builder::add(v[c], B.E(B.pbs(x, lo2)));
        if (c + 1 < m) v[c + 1] = Builder::add(v[c + 1], B.E(B.pbs(x, hi2)));
    }
    return propagate(B, v);
}

// schoolbook product: low and high digit of every block pair, column sums of up to five terms
// reduced to digit + carry until every column has at most two terms, then one carry propagation
Blocks op_mul(Builder& B, const Blocks& a0, const Blocks& b0) {
    const Blocks a = clean(B, a0), b = clean(B, b0);
    const size_t m = a.size();
    std::vector<Blocks> cols(m);
    for (size_t i = 0; i < m; ++i)
        for (size_t j = 0; i + j < m; ++j) {
            const Expr in = pack(B, a[i], b[j]);
            cols[i + j].push_back(B.pbs(in, [](int x) { return (hi2(x) * lo2(x)) & 3; }));
            if (i + j + 1 < m) cols[i + j + 1].push_back(B.pbs(in, [](int x) { return (hi2(x) * lo2(x)) >> 2; }));
        }
    return sum_columns(B, cols);
}

// equality: per-block equality bits (of the difference of the blocks), AND-reduced five at a time
uint32_t op_eq(Builder& B, const Blocks& a0, const Blocks& b0, bool negate) {
    // per block a table of a - b + 3: operands at noise level 2 (results of a select) need no clean-up
    const Blocks a = carry_free(B, a0), b = carry_free(B, b0);
    const size_t m = a.size();
    Blocks cur;
    for (size_t i = 0; i < m; ++i)
        cur.push_back(B.pbs(diff3(B, a[i], b[i]), [=](int x) {
            const int e = x == 3;
            return (m == 1 && negate) ? 1 - e : e;
        }));
    while (cur.size() > 1) {
        Blocks nx;
        const bool last = cur.size() <= 5;
        for (size_t k = 0; k < cur.size(); k += 5) {
            const size_t g = std::min<size_t>(5, cur.size() - k);
            if (g == 1 && !last) { nx.push_back(cur[k]); continue; }
            Expr s;
            for (size_t u = 0; u < g; ++u) s = Builder::add(s, B.E(cur[k + u]));
            nx.push_back(B.pbs(s, [=](int x) {
                const int e = x == (int)g;
                return (last && negate) ? 1 - e : e;
            }));
        }
        cur.swap(nx);
    }
    return cur[0];
}

// order: per-block sign (0 less, 1 equal, 2 greater) read off a - b + 3 (no scaling: operands at noise
// level 2 are taken as they are), most significant non-equal block wins; the signs compose four at
// a time (States)
uint32_t op_cmp(Builder& B, const Blocks& a0, const Blocks& b0, uint32_t op) {
    const Blocks a = carry_free(B, a0), b = carry_free(B, b0);
    const size_t m = a.size();
    auto verdict = [op](int s) {
        switch (op) {
            case LFHE_OP_GT: return s == 2 ? 1 : 0;
            case LFHE_OP_GE: return s != 0 ? 1 : 0;
            case LFHE_OP_LT: return s == 0 ? 1 : 0;
            default: return s != 2 ? 1 : 0;  // LE
        }
    };
    States S(B);
    std::vector<int> cur;  // most significant first
    for (size_t i = m; i-- > 0;)
        cur.push_back(S.leaf(diff3(B, a[i], b[i]), [](int x) { return x > 3 ? 2 : (x < 3 ? 0 : 1); }));   // sign of a - b
    // four at a time (wrapped nodes) or three; the last node must not be wrapped (its verdict is a
    // table of the state itself), so the last round composes at most three
    std::function<int(size_t)> rounds = [&](size_t n) -> int {
        if (n <= 1) return 0;
        if (n <= 3) return 1;
        return 1 + std::min(rounds((n + 2) / 3), n > 4 ? rounds((n + 3) / 4) : 1 << 20);
    };
    while (cur.size() > 1) {
        const size_t n = cur.size();
        const size_t fan = n > 4 && rounds((n + 3) / 4) <= rounds((n + 2) / 3) ? 4 : 3;
        std::vector<int> nx;
        for (size_t k = 0; k < n; k += fan) {
            const size_t g = std::min<size_t>(fan, n - k);
            if (g == 1) nx.push_back(cur[k]);
            else nx.push_back(S.combine(std::vector<int>(cur.begin() + k, cur.begin() + k + g)));
        }
        cur.swap(nx);
    }
    return S.map(cur[0], verdict);
}

// cond ? a : b per block in ONE bootstrap: x = 2 (a - b + 3) + cond fits the block (at most 13) and
// costs exactly the noise budget (2 + 2 + 1); the table returns a - b + 3 when the low bit is set
// and 3 otherwise, and b + table - 3 is the selected digit.
Blocks op_select(Builder& B, uint32_t c0, const Blocks& a0, const Blocks& b0) {
    bool fresh = true;
    for (uint32_t blk : a0) fresh = fresh && B.noise(blk) <= 1 && B.hi(blk) <= 3;
    for (uint32_t blk : b0) fresh = fresh && B.noise(blk) <= 1 && B.hi(blk) <= 3;
    Blocks r;
    if (!fresh) {
        // Operands above noise level 1 (results of an earlier select) cannot be scaled by two.  Two
        // bootstraps per block instead of a clean-up of both operands and one: the condition comes
        // ready-scaled as 4 c from the table that made it (or from a table of its block), a + 4 c is looked up through "a if c else
        // 0", b + 4 c through "0 if c else b", and the sum of the two is the selected digit.
        const Blocks a = carry_free(B, a0), b = carry_free(B, b0);
        const Expr c4 = B.E(B.pbs_of(c0, [](int x) { return 4 * (x & 1); }));
        for (size_t i = 0; i < a.size(); ++i) {
            const uint32_t u = B.pbs(Builder::add(B.E(a[i]), c4), [](int x) { return (x & 4) ? (x & 3) : 0; });
            const uint32_t w = B.pbs(Builder::add(B.E(b[i]), c4), [](int x) { return (x & 4) ? 0 : (x & 3); });
            r.push_back(B.mat(Builder::add(B.E(u), B.E(w)), std::max(B.hi(a[i]), B.hi(b[i])), 0));
        }
        return r;
    }
    const uint32_t c = clean(B, {c0})[0];
    const Blocks &a = a0, &b = b0;
    for (size_t i = 0; i < a.size(); ++i) {
        Expr in = Builder::add(Builder::scale(Builder::sub(B.E(a[i]), B.E(b[i])), 2), Builder::add(B.E(c), Builder::C(6)));
        const uint32_t t = B.pbs(in, [](int x) { return (x & 1) ? (x >> 1) : 3; });
        // between 0 and 3 whatever the interval arithmetic says: it is a[i] or b[i]
        r.push_back(B.mat(Builder::add(Builder::add(B.E(b[i]), B.E(t)), Builder::C(-3)),
                          std::max(B.hi(a[i]), B.hi(b[i])), 0));
    }
    return r;
}

// ---- shifts and rotations -----------------------------------------------------------------------
// An integer of W bits is the bit string X[0..W); every variant maps output bit p to one source bit
// (or to a zero fill) for a given amount t.
long source_bit(uint32_t op, long W, long p, long t) {
    switch (op) {
        case LFHE_OP_SHL: return p - t >= 0 ? p - t : -1;
        case LFHE_OP_SHR: return p + t < W ? p + t : -1;
        case LFHE_OP_ROTL: return ((p - t) % W + W) % W;
        default: return (p + t) % W;  // ROTR
    }
}

// the part of output block i (under an amount of t bits) that comes from source block u holding xu
int shifted_part(uint32_t op, long W, long i, long u, long t, int xu) {
    int r = 0;
    for (int ob = 0; ob < 2; ++ob) {
        const long q = source_bit(op, W, 2 * i + ob, t);
        if (q >= 0 && q / 2 == u) r |= ((xu >> (q & 1)) & 1) << ob;
    }
    return r;
}

// clear amount: re-indexing of blocks when t is even, one bootstrap per block when it is odd
Blocks op_shift_clear(Builder& B, uint32_t op, const Blocks& x, long t) {
    const long m = (long)x.size(), W = 2 * m;
    t %= W;
    Blocks r;
    for (long i = 0; i < m; ++i) {
        std::vector<long> srcs;
        for (int ob = 0; ob < 2; ++ob) {
            const long q = source_bit(op, W, 2 * i + ob, t);
            if (q >= 0 && (srcs.empty() || srcs.back() != q / 2)) srcs.push_back(q / 2);
        }
        if (srcs.empty()) { r.push_back(B.mat(Builder::C(0))); continue; }
        if (t % 2 == 0) { r.push_back(x[srcs[0]]); continue; }
        if (srcs.size() == 2 && B.noise(x[srcs[0]]) <= 1 && B.noise(x[srcs[1]]) <= 1 && B.hi(x[srcs[0]]) <= 3 &&
            B.hi(x[srcs[1]]) <= 3) {
            const long u0 = srcs[0], u1 = srcs[1];
            r.push_back(B.pbs(pack(B, x[u0], x[u1]), [=](int in) {
                return shifted_part(op, W, i, u0, t, hi2(in)) | shifted_part(op, W, i, u1, t, lo2(in));
            }));
            continue;
        }
        Expr sum;
        for (long u : srcs)
            sum = Builder::add(sum, B.E(B.pbs(B.E(x[u]), [=](int in) { return shifted_part(op, W, i, u, t, in & 3); })));
        r.push_back(B.mat(sum, 3));
    }
    return r;
}

// encrypted amount (the blocks of a euint8, of which the low log2 W bits count): one stage per
// base-4 digit s of the amount, out[i] = sum over source blocks u of f_u(s, x[u]); the parts are
// disjoint bit fields, so the sum stays below 4
Blocks op_shift_encrypted(Builder& B, uint32_t op, const Blocks& x0, const Blocks& amount0) {
    const long m = (long)x0.size(), W = 2 * m;
    int log2w = 0;
    while ((1L << log2w) < W) ++log2w;
    if ((1L << log2w) != W) throw std::invalid_argument("operator not supported");  // width not a power of two
    const int ndigits = (log2w + 1) / 2;
    Blocks x = x0;
    for (int j = 0; j < ndigits; ++j) {
        const long unit = 1L << (2 * j);
        const int vmask = (j == ndigits - 1 && (log2w & 1)) ? 1 : 3;
        // A stage looks up 4 s + x[u] (s: this stage's digit of the amount).  Its outputs are sums of up
        // to four bootstrap outputs (noise level up to 4), so the next stage cannot scale anything by
        // four: it takes 4 s ready-made from a table of the amount block (noise level 1; these tables
        // run beside the first stage and cost no layer) and adds x[u] as it is.  The first stage scales
        // the amount block itself when both operands are fresh.
        int xnoise = 0;
        for (uint32_t blk : x) xnoise = std::max(xnoise, B.noise(blk));
        const bool direct = j == 0 && B.hi(amount0[j]) <= 3 && 4 * B.noise(amount0[j]) + xnoise <= MAX_NOISE;
        const Expr s4 = direct ? Builder::scale(B.E(amount0[j]), 4) : B.E(B.pbs(B.E(amount0[j]), [](int v) { return 4 * lo2(v); }));
        Blocks nx;
        for (long i = 0; i < m; ++i) {
            std::vector<long> srcs;
            for (int v = 0; v <= vmask; ++v)
                for (int ob = 0; ob < 2; ++ob) {
                    const long q = source_bit(op, W, 2 * i + ob, v * unit);
                    if (q >= 0 && std::find(srcs.begin(), srcs.end(), q / 2) == srcs.end()) srcs.push_back(q / 2);
                }
            Expr sum;
            for (long u : srcs)
                sum = Builder::add(sum, B.E(B.pbs(Builder::add(s4, B.E(x[u])), [=](int in) {
                    return shifted_part(op, W, i, u, (hi2(in) & vmask) * unit, lo2(in));
                })));
            nx.push_back(B.mat(sum, 3));
        }
        x = nx;
    }
    return clean(B, x);
}

// ---- division by a clear divisor ----------------------------------------------------------------
// little-endian multi-word naturals, just enough for the reciprocal of a divisor of up to 256 bits
typedef std::vector<uint64_t> Nat;
int nat_bit(const Nat& a, size_t i) { return i / 64 < a.size() ? (int)((a[i / 64] >> (i % 64)) & 1) : 0; }
size_t nat_bits(const Nat& a) {
    for (size_t i = a.size() * 64; i-- > 0;)
        if (nat_bit(a, i)) return i + 1;
    return 0;
}
int nat_cmp(const Nat& a, const Nat& b) {
    for (size_t i = std::max(a.size(), b.size()); i-- > 0;) {
        const uint64_t x = i < a.size() ? a[i] : 0, y = i < b.size() ? b[i] : 0;
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}
void nat_sub(Nat& a, const Nat& b) {  // a -= b, a >= b
    uint64_t borrow = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        const uint64_t y = i < b.size() ? b[i] : 0, x = a[i];
        a[i] = x - y - borrow;
        borrow = (x < y || (x == y && borrow)) ? 1 : 0;
    }
}
void nat_shl1(Nat& a, int bit) {
    uint64_t c = (uint64_t)bit;
    for (auto& w : a) { const uint64_t n = w >> 63; w = (w << 1) | c; c = n; }
}
// floor(num / den) by shift and subtract
Nat nat_div(const Nat& num, const Nat& den) {
    Nat q(num.size(), 0), r(num.size() + 1, 0);
    for (size_t i = num.size() * 64; i-- > 0;) {
        nat_shl1(r, nat_bit(num, i));
        if (nat_cmp(r, den) >= 0) { nat_sub(r, den); q[i / 64] |= 1ull << (i % 64); }
    }
    return q;
}
int nat_digit(const Nat& a, size_t i) { return nat_bit(a, 2 * i) | (nat_bit(a, 2 * i + 1) << 1); }

// product of the blocks a (carry-free) with a clear natural c, `columns` base-4 digits of it; plus the
// carry-free blocks `addend`, if given, from column `addend_col` on (they join the column sums)
Blocks op_mul_clear(Builder& B, const Blocks& a, const Nat& c, size_t columns, const Blocks* addend = nullptr,
                    size_t addend_col = 0, const Blocks* minuend = nullptr) {
    const size_t m = a.size();
    std::vector<Blocks> cols(columns);
    for (size_t i = 0; i < m; ++i) {
        uint32_t lo[4] = {0, a[i], 0, 0}, hi[4] = {0, 0, 0, 0};
        bool have[4] = {false, true, false, false};
        for (size_t j = 0; i + j < columns && 2 * j < 64 * c.size(); ++j) {
            const int d = nat_digit(c, j);
            if (!d) continue;
            if (!have[d]) {
                lo[d] = B.pbs_of(a[i], [d](int x) { return ((x & 3) * d) & 3; });
                hi[d] = B.pbs_of(a[i], [d](int x) { return ((x & 3) * d) >> 2; });
                have[d] = true;
            }
            cols[i + j].push_back(lo[d]);
            if (d > 1 && i + j + 1 < columns) cols[i + j + 1].push_back(hi[d]);
        }
    }
    if (addend)
        for (size_t i = 0; i < addend->size() && addend_col + i < columns; ++i) cols[addend_col + i].push_back((*addend)[i]);
    // at most one carry-free block per column (c = 0, 1, a power of four): nothing to add up
    if (minuend) return sum_columns(B, cols, minuend);   // minuend - a c
    bool single = true;
    for (const auto& col : cols) single = single && (col.empty() || (col.size() == 1 && B.hi(col[0]) <= 3));
    if (single) {
        Blocks r;
        for (const auto& col : cols) r.push_back(col.empty() ? B.mat(Builder::C(0)) : col[0]);
        return r;
    }
    return sum_columns(B, cols);
}

// Division of a W-bit integer by a clear d that is not a power of two, by multiplication with a
// reciprocal (Granlund and Montgomery, "Division by invariant integers using multiplication",
// unsigned case): l = ceil(log2 d), r = floor(2^W (2^l - d) / d) + 1, t = high half of r * a,
// quotient = (t + ((a - t) >> 1)) >> (l - 1) = (a + t) >> l = (a r + a 2^W) >> (W + l).
void op_divrem_clear(Builder& B, const Blocks& a0, const Nat& d, Blocks* quot, Blocks* rem) {
    const Blocks a = clean(B, a0);
    const size_t m = a.size(), W = 2 * m;
    const size_t dbits = nat_bits(d);
    if (!dbits) throw std::domain_error("division by zero");
    if (dbits > W) throw std::invalid_argument("divisor wider than the operand");
    bool pow2 = true;
    for (size_t i = 0; i + 1 < dbits; ++i) pow2 = pow2 && !nat_bit(d, i);
    if (pow2) {
        const long k = (long)dbits - 1;
        if (quot) *quot = op_shift_clear(B, LFHE_OP_SHR, a, k);
        if (rem) {
            Blocks r;
            for (long i = 0; i < (long)m; ++i) {
                if (2 * i + 1 < k) r.push_back(a[i]);
                else if (2 * i < k) r.push_back(B.pbs(B.E(a[i]), [](int x) { return x & 1; }));
                else r.push_back(B.mat(Builder::C(0)));
            }
            *rem = r;
        }
        return;
    }
    const size_t l = dbits;  // 2^(l-1) < d < 2^l
    Nat num((2 * W + 63) / 64 + 1, 0), two_l((W + 63) / 64 + 1, 0);
    two_l[l / 64] |= 1ull << (l % 64);
    nat_sub(two_l, d);  // 2^l - d
    for (size_t i = 0; i < W; ++i)
        if (nat_bit(two_l, i)) num[(i + W) / 64] |= 1ull << ((i + W) % 64);
    Nat r = nat_div(num, d);
    for (size_t i = 0; i < r.size(); ++i)  // + 1
        if (++r[i]) break;
    // t + ((a - t) >> 1) = (a + t) >> 1 (the detour keeps the sum inside W bits); with one block more
    // the sum itself fits, and a + t is the upper part of a r + a 2^W: the blocks of a join the column
    // sums of the product from column m on, and the quotient is one shift away
    const Blocks full = op_mul_clear(B, a, r, 2 * m + 1, &a, m);
    Blocks q = op_shift_clear(B, LFHE_OP_SHR, Blocks(full.begin() + (long)m, full.end()), (long)l);
    q.pop_back();   // the quotient is below 2^W
    if (quot) *quot = q;
    if (rem) {
        // a - q d = a + q (2^W - d) modulo 2^W: one product with a in its column sums saves the
        // subtraction's layers, but 2^W - d has a non-zero digit wherever d runs out (every one of them
        // costs a pair of bootstraps per block it meets): taken only when it is no longer than d
        Nat neg_d((W + 63) / 64 + 1, 0);
        neg_d[W / 64] |= 1ull << (W % 64);
        nat_sub(neg_d, d);
        size_t nz_d = 0, nz_neg = 0;
        for (size_t j = 0; j < m; ++j) { nz_d += nat_digit(d, j) != 0; nz_neg += nat_digit(neg_d, j) != 0; }
        if (nz_neg <= nz_d + 2) *rem = op_mul_clear(B, clean(B, q), neg_d, m, &a, 0);
        else *rem = op_mul_clear(B, clean(B, q), d, m, nullptr, 0, &a);
    }
}

}  // namespace

// One operator on blocks that already exist in the builder.  b: right operand blocks (encrypted, or
// the constant blocks of a clear operand); scalar: the clear right operand, for the operators whose
// schedule is specialised on it (shift / rotation amounts, divisors, clear factors), else null.
static Blocks apply_op(Builder& B, uint32_t op, uint32_t width, const Blocks& a, const Blocks& b, uint32_t cond,
                       const uint8_t* scalar, uint32_t& out_width) {
    const int deg = width == 1 ? 1 : 3;
    const bool shift = op == LFHE_OP_SHL || op == LFHE_OP_SHR || op == LFHE_OP_ROTL || op == LFHE_OP_ROTR;
    const bool divrem = op == LFHE_OP_DIV || op == LFHE_OP_REM;
    if ((shift || divrem) && width == 1) throw std::invalid_argument("operator not supported");
    if (divrem && !scalar) throw std::invalid_argument("operator not supported");  // clear divisor only
    out_width = width;
    Blocks out;
    switch (op) {
        case LFHE_OP_ADD: return op_add(B, a, b);
        case LFHE_OP_SUB: return op_sub(B, a, b);
        case LFHE_OP_NEG: return op_neg(B, a);
        case LFHE_OP_MUL:
            if (scalar) {   // clear factor: its digits select the partial products, one bootstrap pair per block and digit value
                Nat c(4);
                for (int i = 0; i < 32; ++i) c[i / 8] |= (uint64_t)scalar[i] << (8 * (i % 8));
                return op_mul_clear(B, clean(B, a), c, a.size());
            }
            return op_mul(B, a, b);
        case LFHE_OP_AND: case LFHE_OP_OR: case LFHE_OP_XOR:
            if (scalar) {   // clear operand: a table of the block alone, and no bootstrap where the digit is 0 or 3
                for (size_t i = 0; i < a.size(); ++i) {
                    const int c = width == 1 ? (scalar[0] & 1) : ((scalar[(2 * i) >> 3] >> ((2 * i) & 7)) & 3);
                    const bool keep = op == LFHE_OP_AND ? c == deg : c == 0;            // x & 3, x | 0, x ^ 0
                    const bool fixed = op == LFHE_OP_AND ? c == 0 : (op == LFHE_OP_OR && c == deg);
                    if (keep && B.hi(a[i]) <= deg) out.push_back(a[i]);
                    else if (fixed) out.push_back(B.mat(Builder::C(op == LFHE_OP_AND ? 0 : deg)));
                    else if (op == LFHE_OP_XOR && c == deg && B.hi(a[i]) <= deg) out.push_back(B.mat(Builder::sub(Builder::C(deg), B.E(a[i]))));
                    else out.push_back(B.pbs(B.E(a[i]), [op, c, deg](int x) {
                        const int p = x & deg;
                        return op == LFHE_OP_AND ? (p & c) : op == LFHE_OP_OR ? (p | c) : (p ^ c);
                    }));
                }
                return out;
            }
            return op_bitwise(B, a, b, op);
        case LFHE_OP_NOT:
            for (uint32_t blk : a) out.push_back(B.mat(Builder::sub(Builder::C(deg), B.E(blk))));
            return out;
        case LFHE_OP_EQ: out_width = 1; return {op_eq(B, a, b, false)};
        case LFHE_OP_NE: out_width = 1; return {op_eq(B, a, b, true)};
        case LFHE_OP_GE: case LFHE_OP_GT: case LFHE_OP_LE: case LFHE_OP_LT:
            out_width = 1;
            return {op_cmp(B, a, b, op)};
        case LFHE_OP_MIN: return op_select(B, op_cmp(B, a, b, LFHE_OP_LT), a, b);
        case LFHE_OP_MAX: return op_select(B, op_cmp(B, a, b, LFHE_OP_GT), a, b);
        case LFHE_OP_SELECT: return op_select(B, cond, a, b);
        case LFHE_OP_SHL: case LFHE_OP_SHR: case LFHE_OP_ROTL: case LFHE_OP_ROTR:
            if (scalar) {
                long t = 0;  // amount modulo the width (a divisor of 2^16 * 5 for every supported width)
                for (int i = 31; i >= 0; --i) t = (t * 256 + scalar[i]) % (long)width;
                return op_shift_clear(B, op, a, t);
            }
            return op_shift_encrypted(B, op, a, b);
        case LFHE_OP_DIV: case LFHE_OP_REM: {
            Nat d(4);
            for (int i = 0; i < 32; ++i) d[i / 8] |= (uint64_t)scalar[i] << (8 * (i % 8));
            op_divrem_clear(B, a, d, op == LFHE_OP_DIV ? &out : nullptr, op == LFHE_OP_REM ? &out : nullptr);
            return out;
        }
        default: throw std::invalid_argument("operator not supported");
    }
}

static bool specialised_on_scalar(uint32_t op) {
    return op == LFHE_OP_SHL || op == LFHE_OP_SHR || op == LFHE_OP_ROTL || op == LFHE_OP_ROTR || op == LFHE_OP_DIV ||
           op == LFHE_OP_REM || op == LFHE_OP_MUL || op == LFHE_OP_AND || op == LFHE_OP_OR || op == LFHE_OP_XOR;
}

Program build_program(uint32_t op, uint32_t width, int lhs_noise, int rhs_noise, int cond_noise, const uint8_t* scalar) {
    Builder B;
    B.P.op = op;
    B.P.width = width;
    const uint32_t m = width == 1 ? 1 : width / 2;
    const int deg = width == 1 ? 1 : 3;
    Blocks a, b;
    for (uint32_t i = 0; i < m; ++i) a.push_back(B.input(deg, lhs_noise));
    B.P.lhs = a;
    const bool unary = op == LFHE_OP_NOT || op == LFHE_OP_NEG;
    const bool shift = op == LFHE_OP_SHL || op == LFHE_OP_SHR || op == LFHE_OP_ROTL || op == LFHE_OP_ROTR;
    const bool specialised = scalar && specialised_on_scalar(op);
    B.P.rhs_width = (unary || specialised) ? 0 : (shift ? 8 : width);
    if (B.P.rhs_width) {
        const uint32_t mr = B.P.rhs_width == 1 ? 1 : B.P.rhs_width / 2;
        for (uint32_t i = 0; i < mr; ++i) b.push_back(B.input(deg, rhs_noise));
        B.P.rhs = b;
    }
    uint32_t c = 0;
    if (op == LFHE_OP_SELECT) {
        c = B.input(1, cond_noise);
        B.P.cond = {c};
    }
    B.P.out = apply_op(B, op, width, a, b, c, specialised ? scalar : nullptr, B.P.out_width);
    B.finish();
    return B.P;
}

// Several operators chained into one program: intermediate results stay in the arena, independent
// steps share bootstrap layers.  Value references: 0 .. n_inputs-1 are the inputs, n_inputs + k is
// the result of step k.
Program build_fused(const std::vector<FusedStep>& steps, const std::vector<uint32_t>& in_widths,
                    const std::vector<int>& in_noise, const std::vector<int32_t>& outputs) {
    Builder B;
    B.P.op = 0xffffffffu;
    std::vector<Blocks> val;
    std::vector<uint32_t> wid;
    for (size_t i = 0; i < in_widths.size(); ++i) {
        const uint32_t w = in_widths[i], m = w == 1 ? 1 : w / 2;
        Blocks v;
        for (uint32_t k = 0; k < m; ++k) v.push_back(B.input(w == 1 ? 1 : 3, in_noise[i]));
        B.P.ins.push_back(v);
        val.push_back(v);
        wid.push_back(w);
    }
    B.P.in_widths = in_widths;
    auto ref = [&](int32_t r, uint32_t want_width, const char* what) -> const Blocks& {
        if (r < 0 || (size_t)r >= val.size()) throw std::invalid_argument(std::string("bad value reference for ") + what);
        if (wid[r] != want_width) throw std::invalid_argument(std::string("width mismatch at ") + what);
        return val[r];
    };
    for (const FusedStep& st : steps) {
        const uint32_t w = st.width;
        const bool unary = st.op == LFHE_OP_NOT || st.op == LFHE_OP_NEG;
        const bool shift = st.op == LFHE_OP_SHL || st.op == LFHE_OP_SHR || st.op == LFHE_OP_ROTL || st.op == LFHE_OP_ROTR;
        const Blocks& a = ref(st.lhs, w, "lhs");
        Blocks b;
        const uint8_t* scalar = nullptr;
        if (!unary) {
            if (st.rhs_is_scalar) {
                if (specialised_on_scalar(st.op)) {
                    scalar = st.scalar;
                } else {  // a clear operand becomes constant blocks
                    const uint32_t m = w == 1 ? 1 : w / 2;
                    for (uint32_t k = 0; k < m; ++k) {
                        const uint32_t bit = 2 * k;
                        const int d = w == 1 ? (st.scalar[0] & 1) : ((st.scalar[bit >> 3] >> (bit & 7)) & 3);
                        b.push_back(B.mat(Builder::C(d)));
                    }
                }
            } else {
                b = ref(st.rhs, shift ? 8 : w, "rhs");
            }
        }
        uint32_t c = 0;
        if (st.op == LFHE_OP_SELECT) c = ref(st.cond, 1, "cond")[0];
        uint32_t ow = w;
        val.push_back(apply_op(B, st.op, w, a, b, c, scalar, ow));
        wid.push_back(ow);
    }
    for (int32_t r : outputs) {
        if (r < 0 || (size_t)r >= val.size()) throw std::invalid_argument("bad output reference");
        int noise = 0, deg = 0;
        for (uint32_t blk : val[r]) { noise = std::max(noise, B.noise(blk)); deg = std::max(deg, B.hi(blk)); }
        if (deg > 3) throw std::logic_error("operator output is not carry-free");
        B.P.outs.push_back(val[r]);
        B.P.out_widths.push_back(wid[r]);
        B.P.out_noises.push_back((uint8_t)std::max(noise, 1));
        B.P.out_degrees.push_back((uint8_t)deg);
    }
    B.prune();
    B.balance(layer_cap());
    return B.P;
}

}  // namespace lfhe
