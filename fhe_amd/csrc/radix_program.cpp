#include "radix_program.h"

#include <algorithm>
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

uint32_t Builder::lut_id(const std::function<int(int)>& f) {
    std::array<uint64_t, 16> tab;
    for (int x = 0; x < 16; ++x) {
        const int v = f(x);
        if (v < 0 || v > BLOCK_MAX) throw std::logic_error("lookup table value outside the block");
        tab[x] = (uint64_t)v;
    }
    auto it = lut_ids_.find(tab);
    if (it != lut_ids_.end()) return it->second;
    const uint32_t id = (uint32_t)P.luts.size();
    P.luts.push_back(tab);
    lut_ids_[tab] = id;
    return id;
}

uint32_t Builder::lin(const Expr& e, const Info& i) {
    if (e.t.size() > (size_t)LIN_MAX_TERMS) throw std::logic_error("linear combination too long");
    if (i.lo < 0 || i.hi > BLOCK_MAX) throw std::logic_error("block value range overflows the carry space");
    if (i.noise > MAX_NOISE) throw std::logic_error("noise level above the parameter set's budget");
    const uint32_t out = new_block(i.level, i.lo, i.hi, i.noise);
    LinOp op{};
    op.out = out;
    op.n = (uint32_t)e.t.size();
    for (size_t k = 0; k < e.t.size(); ++k) { op.coef[k] = e.t[k].first; op.src[k] = e.t[k].second; }
    op.cst = e.cst;
    layer(i.level).lin.push_back(op);
    return out;
}

uint32_t Builder::mat(const Expr& e, int hi_known) {
    if (e.t.size() == 1 && e.t[0].first == 1 && e.cst == 0) return e.t[0].second;
    Info i = eval(e);
    if (hi_known >= 0) i.hi = std::min(i.hi, hi_known);
    return lin(e, i);
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
    return out;
}

void Builder::finish() {
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
    o << "{\"op\":" << op << ",\"width\":" << width << ",\"out_width\":" << out_width << ",\"nblocks\":" << nblocks
      << ",\"out_noise\":" << (int)out_noise << ",\"pbs\":" << pbs_per_instance() << ",\"lhs\":";
    list(lhs); o << ",\"rhs\":"; list(rhs); o << ",\"cond\":"; list(cond); o << ",\"out\":"; list(out);
    o << ",\"luts\":[";
    for (size_t i = 0; i < luts.size(); ++i) {
        o << (i ? "," : "") << "[";
        for (int x = 0; x < 16; ++x) o << (x ? "," : "") << luts[i][x];
        o << "]";
    }
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

Expr pack(Builder& B, uint32_t hi_blk, uint32_t lo_blk) { return Builder::add(Builder::scale(B.E(hi_blk), 4), B.E(lo_blk)); }

// Carry propagation over per-block sums v[i] in [0,7]: digit i = (v[i] + carry_in[i]) mod 4,
// carry_out[i] = (v[i] + carry_in[i]) >= 4.  One layer extracts digit and carry state (0 none,
// 1 propagate, 2 generate), a Hillis-Steele scan resolves the states in ceil(log2 m) layers, a
// last layer adds the incoming carry.
Blocks propagate(Builder& B, const std::vector<Expr>& v) {
    const size_t m = v.size();
    Blocks msg(m), st(m ? m - 1 : 0);
    for (size_t i = 0; i < m; ++i) {
        msg[i] = B.pbs(v[i], lo2);
        if (i + 1 < m) {
            if (i == 0) st[i] = B.pbs(v[i], [](int x) { return x >= 4 ? 2 : 0; });
            else st[i] = B.pbs(v[i], [](int x) { return x >= 4 ? 2 : (x == 3 ? 1 : 0); });
        }
    }
    for (size_t d = 1; d < st.size(); d *= 2) {
        Blocks nx = st;
        for (size_t i = d; i < st.size(); ++i)
            nx[i] = B.pbs(pack(B, st[i], st[i - d]), [](int x) { return hi2(x) == 1 ? lo2(x) : std::min(hi2(x), 2); });
        st = nx;
    }
    Blocks res(m);
    for (size_t i = 0; i < m; ++i) {
        if (i == 0) res[i] = msg[i];
        else res[i] = B.pbs(Builder::add(B.E(msg[i]), Builder::scale(B.E(st[i - 1]), 4)),
                            [](int x) { return (lo2(x) + (hi2(x) == 2 ? 1 : 0)) & 3; });
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
    for (;;) {
        size_t longest = 0;
        for (const auto& c : cols) longest = std::max(longest, c.size());
        if (longest <= 2) break;
        std::vector<Blocks> nx(m);
        for (size_t c = 0; c < m; ++c) {
            const Blocks& t = cols[c];
            if (t.size() <= 2) { nx[c].insert(nx[c].end(), t.begin(), t.end()); continue; }
            for (size_t k = 0; k < t.size(); k += 5) {
                const size_t g = std::min<size_t>(5, t.size() - k);
                if (g == 1) { nx[c].push_back(t[k]); continue; }
                Expr s;
                for (size_t u = 0; u < g; ++u) s = Builder::add(s, B.E(t[k + u]));
                nx[c].push_back(B.pbs(s, lo2));
                if (c + 1 < m) nx[c + 1].push_back(B.pbs(s, hi2));
            }
        }
        cols.swap(nx);
    }
    std::vector<Expr> v;
    for (size_t c = 0; c < m; ++c) {
        Expr s;
        for (uint32_t blk : cols[c]) s = Builder::add(s, B.E(blk));
        v.push_back(s);
    }
    return propagate(B, v);
}

// equality: per-block equality bits, AND-reduced five at a time
uint32_t op_eq(Builder& B, const Blocks& a0, const Blocks& b0, bool negate) {
    const Blocks a = clean(B, a0), b = clean(B, b0);
    const size_t m = a.size();
    Blocks cur;
    for (size_t i = 0; i < m; ++i)
        cur.push_back(B.pbs(pack(B, a[i], b[i]), [=](int x) {
            const int e = hi2(x) == lo2(x);
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

// order: per-block sign (0 equal, 1 less, 2 greater), most significant non-equal block wins
uint32_t op_cmp(Builder& B, const Blocks& a0, const Blocks& b0, uint32_t op) {
    const Blocks a = clean(B, a0), b = clean(B, b0);
    const size_t m = a.size();
    auto verdict = [op](int s) {
        switch (op) {
            case LFHE_OP_GT: return s == 2 ? 1 : 0;
            case LFHE_OP_GE: return s != 1 ? 1 : 0;
            case LFHE_OP_LT: return s == 1 ? 1 : 0;
            default: return s != 2 ? 1 : 0;  // LE
        }
    };
    auto sign = [](int x) { return hi2(x) > lo2(x) ? 2 : (hi2(x) < lo2(x) ? 1 : 0); };
    Blocks cur;  // most significant first
    for (size_t i = m; i-- > 0;)
        cur.push_back(B.pbs(pack(B, a[i], b[i]), [=](int x) { return m == 1 ? verdict(sign(x)) : sign(x); }));
    while (cur.size() > 1) {
        Blocks nx;
        const bool last = cur.size() == 2;
        for (size_t k = 0; k < cur.size(); k += 2) {
            if (k + 1 == cur.size()) { nx.push_back(cur[k]); continue; }
            nx.push_back(B.pbs(pack(B, cur[k], cur[k + 1]), [=](int x) {
                const int s = hi2(x) != 0 ? std::min(hi2(x), 2) : std::min(lo2(x), 2);
                return last ? verdict(s) : s;
            }));
        }
        cur.swap(nx);
    }
    return cur[0];
}

Blocks op_select(Builder& B, uint32_t c0, const Blocks& a0, const Blocks& b0) {
    const uint32_t c = clean(B, {c0})[0];
    const Blocks a = clean(B, a0), b = clean(B, b0);
    Blocks r;
    for (size_t i = 0; i < a.size(); ++i) {
        const uint32_t p = B.pbs(pack(B, c, a[i]), [](int x) { return hi2(x) ? lo2(x) : 0; });
        const uint32_t q = B.pbs(pack(B, c, b[i]), [](int x) { return hi2(x) ? 0 : lo2(x); });
        r.push_back(B.mat(Builder::add(B.E(p), B.E(q)), std::max(B.hi(a[i]), B.hi(b[i]))));  // one of p, q is zero
    }
    return r;
}

}  // namespace

Program build_program(uint32_t op, uint32_t width, int lhs_noise, int rhs_noise, int cond_noise) {
    Builder B;
    B.P.op = op;
    B.P.width = width;
    const uint32_t m = width == 1 ? 1 : width / 2;
    const int deg = width == 1 ? 1 : 3;
    Blocks a, b;
    for (uint32_t i = 0; i < m; ++i) a.push_back(B.input(deg, lhs_noise));
    B.P.lhs = a;
    const bool unary = op == LFHE_OP_NOT || op == LFHE_OP_NEG;
    if (!unary) {
        for (uint32_t i = 0; i < m; ++i) b.push_back(B.input(deg, rhs_noise));
        B.P.rhs = b;
    }
    uint32_t c = 0;
    if (op == LFHE_OP_SELECT) {
        c = B.input(1, cond_noise);
        B.P.cond = {c};
    }
    B.P.out_width = width;
    switch (op) {
        case LFHE_OP_ADD: B.P.out = op_add(B, a, b); break;
        case LFHE_OP_SUB: B.P.out = op_sub(B, a, b); break;
        case LFHE_OP_NEG: B.P.out = op_neg(B, a); break;
        case LFHE_OP_MUL: B.P.out = op_mul(B, a, b); break;
        case LFHE_OP_AND: case LFHE_OP_OR: case LFHE_OP_XOR: B.P.out = op_bitwise(B, a, b, op); break;
        case LFHE_OP_NOT:
            for (uint32_t i = 0; i < m; ++i) B.P.out.push_back(B.mat(Builder::sub(Builder::C(deg), B.E(a[i]))));
            break;
        case LFHE_OP_EQ: B.P.out = {op_eq(B, a, b, false)}; B.P.out_width = 1; break;
        case LFHE_OP_NE: B.P.out = {op_eq(B, a, b, true)}; B.P.out_width = 1; break;
        case LFHE_OP_GE: case LFHE_OP_GT: case LFHE_OP_LE: case LFHE_OP_LT:
            B.P.out = {op_cmp(B, a, b, op)}; B.P.out_width = 1; break;
        case LFHE_OP_MIN: B.P.out = op_select(B, op_cmp(B, a, b, LFHE_OP_LT), a, b); break;
        case LFHE_OP_MAX: B.P.out = op_select(B, op_cmp(B, a, b, LFHE_OP_GT), a, b); break;
        case LFHE_OP_SELECT: B.P.out = op_select(B, c, a, b); break;
        default: throw std::invalid_argument("operator not supported");
    }
    B.finish();
    return B.P;
}

}  // namespace lfhe
