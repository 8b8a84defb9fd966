// Radix integer operators as programs over ciphertext blocks.
//
// An encrypted integer is a little-endian vector of blocks, each an LWE ciphertext holding a value
// in [0,16): two message bits, two carry bits, one padding bit above them.  Every operator is
// compiled once per (operator, width, input noise) into a Program: a list of layers, each made of
// linear combinations of blocks (free of bootstraps) followed by independent programmable
// bootstraps.  The executor (radix.cpp) replays the program for a whole batch of independent
// instances at once, so every layer becomes one key-switch launch and one blind-rotation launch
// over instances x bootstraps-per-layer ciphertexts.
//
// The builder tracks, for every block, the range of values it can hold and its noise level in
// units of one bootstrap output (the parameter set allows 5), and refuses to build a program that
// would overflow the 4-bit block or exceed the noise budget.  A few bootstraps take sums that run
// into the padding bit on purpose (inputs up to 31, tables flagged in lut_wraps: pbs_wrapped); the
// block a consumer sees is back in [0,16) after the constant of its linear combination.
// Finished programs are pruned of operations no output depends on.
#pragma once
#include <stdint.h>

#include <array>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace lfhe {

constexpr int LIN_MAX_TERMS = 6;

struct LinOp {  // out = sum coef[k] * block[src[k]] + cst * delta   (plain data: copied to the device)
    uint32_t out, n;
    uint32_t src[LIN_MAX_TERMS];
    int32_t coef[LIN_MAX_TERMS];
    int32_t cst;
};
struct PbsOp {
    uint32_t in, out, lut;
};
struct Layer {
    std::vector<LinOp> lin;
    std::vector<PbsOp> pbs;
};

struct Program {
    uint32_t op = 0, width = 0, out_width = 0;
    uint32_t rhs_width = 0;                // 0: no encrypted right operand
    uint32_t nblocks = 0;                  // arena blocks per instance
    std::vector<uint32_t> lhs, rhs, cond;  // input block ids
    std::vector<uint32_t> out;             // output block ids, little-endian
    std::vector<std::array<uint64_t, 16>> luts;
    // per table: 1 when its bootstraps take inputs up to 31 (padding bit in use).  A blind rotation
    // maps an input x + 16 to minus the entry of x, so such a table holds values modulo 32 (an entry
    // t >= 16 stands for t - 32) and its output may be negative until a constant is added.
    std::vector<uint8_t> lut_wraps;
    std::vector<Layer> layers;
    uint8_t out_noise = 1, out_degree = 3;
    // fused programs (build_fused): any number of inputs and outputs instead of lhs / rhs / cond / out
    std::vector<std::vector<uint32_t>> ins, outs;
    std::vector<uint32_t> in_widths, out_widths;
    std::vector<uint8_t> out_noises, out_degrees;
    uint64_t pbs_per_instance() const {
        uint64_t s = 0;
        for (const auto& l : layers) s += l.pbs.size();
        return s;
    }
    std::string to_json() const;
};

constexpr int MAX_NOISE = 5;   // max_noise_level of the parameter set
constexpr int BLOCK_MAX = 15;  // message + carry bits

class Builder {
public:
    struct Expr {
        std::vector<std::pair<int, uint32_t>> t;  // (coefficient, block)
        int cst = 0;
    };
    Program P;

    uint32_t input(int hi, int noise) { return new_block(0, 0, hi, noise); }
    Expr E(uint32_t b) const { Expr e; e.t.push_back({1, b}); return e; }
    static Expr C(int c) { Expr e; e.cst = c; return e; }
    static Expr add(Expr a, const Expr& b) {
        for (auto& x : b.t) a.t.push_back(x);
        a.cst += b.cst;
        return a;
    }
    static Expr scale(Expr a, int c) {
        for (auto& x : a.t) x.first *= c;
        a.cst *= c;
        return a;
    }
    static Expr sub(const Expr& a, const Expr& b) { return add(a, scale(b, -1)); }

    // bootstrap of e through the table f : [0,16) -> [0,16)
    uint32_t pbs(const Expr& e, const std::function<int(int)>& f);
    // g of the value of block b in one bootstrap: when b is itself the plain output of a bootstrap, a
    // second bootstrap of that one's input through g o f (in b's own layer, no layer added); else of b
    uint32_t pbs_of(uint32_t b, const std::function<int(int)>& g);
    // bootstrap of an input in [0,32) through f : [0,16) -> (-16,16), continued as f(x + 16) = -f(x)
    uint32_t pbs_wrapped(const Expr& e, const std::function<int(int)>& f);
    // e as a block (a linear op unless e already is a plain block); hi_known / lo_known >= 0
    // override the interval bounds when the caller knows tighter ones (e.g. a sum whose terms
    // exclude each other)
    uint32_t mat(const Expr& e, int hi_known = -1, int lo_known = -1, int limit = BLOCK_MAX);
    int hi(uint32_t b) const { return info_[b].hi; }
    int noise(uint32_t b) const { return info_[b].noise; }
    int level(uint32_t b) const { return info_[b].level; }   // the layer whose bootstraps may read the block
    void finish();
    void prune();   // drops the operations no output depends on
    // Moves bootstraps that have slack out of layers larger than `cap` ciphertexts into later ones,
    // without adding a layer (radix_program.cpp).  cap 0: leaves the program as built.
    void balance(size_t cap);

private:
    struct Info { int level, lo, hi, noise; };
    std::vector<Info> info_;
    std::map<std::pair<std::array<uint64_t, 16>, bool>, uint32_t> lut_ids_;
    std::map<uint32_t, Expr> lin_expr_;  // blocks defined by a linear operation
    std::map<uint32_t, std::pair<uint32_t, std::function<int(int)>>> made_by_;  // bootstrap outputs: input block, table
    uint32_t new_block(int level, int lo, int hi, int noise) {
        info_.push_back({level, lo, hi, noise});
        return P.nblocks++;
    }
    Info eval(const Expr& e) const;
    Layer& layer(int l) {
        if ((int)P.layers.size() <= l) P.layers.resize(l + 1);
        return P.layers[l];
    }
    uint32_t lut_id(const std::function<int(int)>& f, bool wraps = false);
    uint32_t lin(const Expr& e, const Info& i, int limit);
};

// one step of a fused program; lhs / rhs / cond are value references (-1: unused)
struct FusedStep {
    uint32_t op = 0, width = 0;
    int32_t lhs = -1, rhs = -1, cond = -1;
    uint32_t rhs_is_scalar = 0;
    uint8_t scalar[32] = {0};
};

// operator ids are the LFHE_OP_* values of include/lfhe.h
// scalar: a clear right operand (32 bytes, little-endian) the program is specialised on -- shift and
// rotation amounts, divisors; null for an encrypted (or trivially encrypted) right operand
Program build_program(uint32_t op, uint32_t width, int lhs_noise, int rhs_noise, int cond_noise,
                      const uint8_t* scalar = nullptr);

Program build_fused(const std::vector<FusedStep>& steps, const std::vector<uint32_t>& in_widths,
                    const std::vector<int>& in_noise, const std::vector<int32_t>& outputs);

}  // namespace lfhe
