// Host-side emulation of the device external product: runs the very same per-thread phase code
// (fhe_amd/csrc/pbs_phases.h) for 256 emulated threads with a plain array as LDS, and compares the
// result bit for bit with the CPU oracle.  Test code: links oracle/liboracle.so.
#define LFHE_TRACK_MAX 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <type_traits>
#include <vector>

#include "../../fhe_amd/csrc/pbs_phases.h"
#include "../../fhe_amd/csrc/tables.h"
#include "../../oracle/tfhe_oracle.h"

using namespace lfhe;

static u64 rng_state = 0x9e3779b97f4a7c15ull;
static u64 rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state * 0x2545F4914F6CDD1Dull;
}

// The emulated schedule is the most hostile one the synchronisation table (sync_after) allows: a
// workgroup barrier joins all wavefronts, anything weaker does not, so between two barriers one
// wavefront runs all its phases before the next one starts, in a scrambled wave order.  A hand-over
// marked wave-private that in truth crossed wavefronts would read stale words here.
constexpr int WAVE_ORDER[NT / 64] = {2, 0, 3, 1};

template <int PH, int END, typename F>
static void run_wave(int wave, F&& phase) {   // phases PH.. of one wavefront up to its next barrier
    if constexpr (PH < END) {
        for (int l = 0; l < 64; ++l) phase(std::integral_constant<int, PH>{}, wave * 64 + l);
        if constexpr (sync_after(PH) != Sync::Block) run_wave<PH + 1, END>(wave, phase);
    }
}
template <int PH, int END>
constexpr int segment_end() {   // first phase after PH's barrier-free run
    if constexpr (PH >= END) return END;
    else if constexpr (sync_after(PH) == Sync::Block) return PH + 1;
    else return segment_end<PH + 1, END>();
}
template <int PH, int END, typename F>
static void run_schedule(F&& phase) {
    if constexpr (PH < END) {
        for (int w : WAVE_ORDER) run_wave<PH, END>(w, phase);
        run_schedule<segment_end<PH, END>(), END>(phase);
    }
}

template <int PH>
static void run_phases(std::vector<EpState>& st, double* lds, const Tables& T, const double* bsk_i) {
    run_schedule<PH, EP_PHASES>([&](auto ph, int t) { ep_phase<decltype(ph)::value>(st[t], lds, t, T, bsk_i); });
}

static void make_bsk_row(const HostTables& H, const u64* ggsw, std::vector<double>& row) {
    row.assign(BSK_ROW, 0.0);
    std::vector<u64> in(N), out(N);
    for (int r = 0; r < 2; ++r) {
        const u64 p = H.p[r];
        const u64 ninv = h_powmod(N, p - 2, p);
        for (int q = 0; q < 4; ++q) {
            for (int i = 0; i < N; ++i) {
                i64 s = (i64)ggsw[q * N + i];
                i64 m = s % (i64)p;
                in[i] = m < 0 ? (u64)(m + (i64)p) : (u64)m;
            }
            orc_ntt_definition(p, H.psi[r], in.data(), out.data(), N);
            for (int e = 0; e < N; ++e) {
                u64 v = h_mulmod(out[h_bitrev((u32)e, LOGN)], ninv, p);
                int t = e >> 3, c4 = (e & 7) >> 1, h = e & 1;
                row[bsk_off(r, q, c4, t) + h] = h_centred(v, p);
            }
        }
    }
}

static int run_case(const orc_ctx* oc, const HostTables& H, const Tables& T, const u64* ggsw,
                    const u64* glwe, const char* name) {
    std::vector<double> row;
    make_bsk_row(H, ggsw, row);
    std::vector<EpState> st(NT);
    std::vector<double> lds(LDS_WORDS, 0.0);
    for (int t = 0; t < NT; ++t)
        for (int q = 0; q < 2; ++q)
            for (int j = 0; j < E; ++j) { st[t].dig[q][j] = decomp23(glwe[q * N + own1(t, j)]); st[t].out[q][j] = 0; }
    run_phases<0>(st, lds.data(), T, row.data());
    std::vector<u64> want(2 * N);
    orc_external_product(oc, ggsw, glwe, want.data());
    int bad = 0;
    for (int t = 0; t < NT; ++t)
        for (int q = 0; q < 2; ++q)
            for (int j = 0; j < E; ++j)
                if (st[t].out[q][j] != want[q * N + own1(t, j)]) ++bad;
    printf("%-28s mismatches: %d   max|x| = 2^%.2f = %.2f p\n", name, bad, log2(track_max_ref()), track_max_ref() / P0);
    if (track_max_ref() >= 9007199254740992.0) { printf("lazy-reduction bound exceeded\n"); ++bad; }
    track_max_ref() = 0;
    return bad;
}

// ---- one step of the grouped blind rotation (single ring modulus): phases 0-4, the grouped
// point-wise stage, phases 6-11, against orc_group_step_qp
// phases 0-11 of the grouped step, phase 5 being the grouped point-wise stage
static void run_group_phases(std::vector<EpState>& st, double* lds, const Tables& T, const double* rows,
                             unsigned a0, unsigned a1) {
    run_schedule<0, EPQ_PHASES>([&](auto ph, int t) {
        constexpr int PH = decltype(ph)::value;
        if constexpr (PH == 5) ep_group2_phase5(st[t], lds, t, T, rows, a0, a1);
        else ep_phase<PH, false, true>(st[t], lds, t, T, nullptr);
    });
}

static void make_bskq_rows(const HostTables& H, const u64* ggsw4, std::vector<double>& rows) {
    rows.assign(4 * BSKQ_ROW, 0.0);
    const u64 p = H.p[0];
    const u64 ninv = h_powmod(N, p - 2, p);
    std::vector<u64> out(N);
    for (int m = 0; m < 4; ++m)
        for (int q = 0; q < 4; ++q) {
            orc_ntt_definition(p, H.psi[0], ggsw4 + ((size_t)m * 4 + q) * N, out.data(), N);
            for (int e = 0; e < N; ++e) {
                const u64 v = h_mulmod(out[h_bitrev((u32)e, LOGN)], ninv, p);
                rows[(size_t)m * BSKQ_ROW + bsk_off(0, q, (e & 7) >> 1, e >> 3) + (e & 1)] = h_centred(v, p);
            }
        }
}

static int run_group_case(const orc_ctx* oc, const HostTables& H, const Tables& T,
                          const u64* ggsw4, const u64* acc_in, unsigned a0, unsigned a1, const char* name) {
    std::vector<double> rows;
    make_bskq_rows(H, ggsw4, rows);
    const u64 p = H.p[0];
    const u64 G = orc_qp_gadget(oc);
    const double inv_g = 1.0 / (double)G;
    std::vector<EpState> st(NT);
    std::vector<double> lds(LDS_WORDS, 0.0);
    for (int t = 0; t < NT; ++t)
        for (int q = 0; q < 2; ++q)
            for (int j = 0; j < E; ++j)
                st[t].x[q][j] = __builtin_rint(h_centred(acc_in[q * N + own1(t, j)], p) * inv_g);
    run_group_phases(st, lds.data(), T, rows.data(), a0, a1);
    std::vector<u64> want(acc_in, acc_in + 2 * N);
    orc_group_step_qp(oc, ggsw4, a0, a1, want.data());
    int bad = 0;
    for (int t = 0; t < NT; ++t)
        for (int q = 0; q < 2; ++q)
            for (int j = 0; j < E; ++j) {
                const double c = mnorm(st[t].x[q][j], T.pr[0]);
                const u64 got = c < 0 ? p - (u64)(-c) : (u64)c;
                if (got != want[q * N + own1(t, j)]) ++bad;
            }
    printf("%-28s mismatches: %d   max|x| = 2^%.2f = %.2f p\n", name, bad, log2(track_max_ref()), track_max_ref() / P0);
    if (track_max_ref() >= 9007199254740992.0) { printf("lazy-reduction bound exceeded\n"); ++bad; }
    track_max_ref() = 0;
    return bad;
}

int main() {
    HostTables H;
    build_host_tables(H);
    Tables T;
    for (int r = 0; r < 2; ++r) {
        T.tw_fwd[r] = H.tw_fwd[r].data(); T.tw_inv[r] = H.tw_inv[r].data();
        T.pr[r].p = (double)H.p[r]; T.pr[r].pinv = 1.0 / (double)H.p[r];
        T.two32[r] = H.two32[r]; T.ninv[r] = H.ninv[r];
    }
    T.p0inv_mod_p1 = H.p0inv_mod_p1;

    orc_params prm;
    orc_default_params(&prm);
    orc_ctx* oc = orc_ctx_new(&prm);

    // digit rule agrees with the oracle's decomposer on edge values
    {
        const u64 edge[] = {0, 1ull << 40, (1ull << 40) - 1, 1ull << 63, (1ull << 63) + (1ull << 41),
                            ~0ull, (1ull << 63) - (1ull << 40), (1ull << 63) - (1ull << 40) - 1};
        for (u64 v : edge) {
            int64_t d;
            orc_decompose(v, 23, 1, &d);
            if (d != decomp23(v)) { printf("decomp mismatch at %llx\n", (unsigned long long)v); return 1; }
        }
        for (int i = 0; i < 100000; ++i) {
            u64 v = rnd(); int64_t d;
            orc_decompose(v, 23, 1, &d);
            if (d != decomp23(v)) { printf("decomp mismatch at %llx\n", (unsigned long long)v); return 1; }
        }
    }

    int bad = 0;
    std::vector<u64> ggsw(4 * N), glwe(2 * N);
    for (auto& v : ggsw) v = rnd();
    for (auto& v : glwe) v = rnd();
    bad += run_case(oc, H, T, ggsw.data(), glwe.data(), "random");
    // worst-case magnitudes: every digit +2^22, every key coefficient -2^63 (|sum| = 2^97)
    for (auto& v : ggsw) v = 1ull << 63;
    for (auto& v : glwe) v = 1ull << 63;
    bad += run_case(oc, H, T, ggsw.data(), glwe.data(), "all-extreme");
    for (int i = 0; i < 4 * N; ++i) ggsw[i] = (i & 1) ? (1ull << 63) : (1ull << 63) - 1;
    for (int i = 0; i < 2 * N; ++i) glwe[i] = (i % 3) ? (1ull << 63) : (1ull << 63) + (1ull << 41);
    bad += run_case(oc, H, T, ggsw.data(), glwe.data(), "alternating-extreme");
    for (int i = 0; i < 4 * N; ++i) ggsw[i] = rnd();
    for (int i = 0; i < 2 * N; ++i) glwe[i] = (rnd() & 1) ? (1ull << 63) : (1ull << 63) + (1ull << 41);
    bad += run_case(oc, H, T, ggsw.data(), glwe.data(), "extreme-digits-random-key");
    // grouped step over the prime ring modulus
    {
        const u64 p = H.p[0];
        // the table as the engine lays it out: psi^k (k < 2N), then the 8th roots of unity
        std::vector<double> table(MONO_WORDS + 8);
        for (int k = 0; k < MONO_WORDS; ++k) {
            table[k] = h_centred(h_powmod(H.psi[0], k, p), p);
            if (k % (N / 4) == 0) table[MONO_WORDS + k / (N / 4)] = table[k];
        }
        T.mono = table.data();
        std::vector<u64> g3(4 * 4 * N), acc(2 * N);
        for (auto& v : g3) v = rnd() % p;
        for (auto& v : acc) v = rnd() % p;
        bad += run_group_case(oc, H, T, g3.data(), acc.data(), 1234, 3071, "grouped random");
        bad += run_group_case(oc, H, T, g3.data(), acc.data(), 0, 4095, "grouped a0 = 0");
        bad += run_group_case(oc, H, T, g3.data(), acc.data(), 2048, 2048, "grouped a0 + a1 = 2N");
        for (auto& v : g3) v = p / 2;
        for (int i = 0; i < 2 * N; ++i) acc[i] = (i & 1) ? p / 2 : p / 2 + 1;
        bad += run_group_case(oc, H, T, g3.data(), acc.data(), 4095, 1, "grouped extreme");
    }
    orc_ctx_free(oc);
    if (bad) { printf("FAIL\n"); return 1; }
    printf("OK\n");
    return 0;
}
