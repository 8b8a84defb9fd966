// Host check of the lift guard of the grouped key over 2^64 (fhe_amd/csrc/pbs_device.h:
// crt_wrap_checked), the same code the kernel runs: integers inside the guard-free part of the lift's
// window come back exact and unmarked, integers in the guard band exact and marked, and an integer
// that LEFT the window by up to a band's width -- the forced overflow -- comes back wrong (it cannot
// be told from its neighbour P away) but marked, which is what sends its ciphertext to the slow path.
#include <cstdio>
#include <cstdlib>

#include "../../fhe_amd/csrc/pbs_device.h"

using namespace lfhe;
typedef __int128 i128;

static double centred(i128 v, u64 p) {
    i128 r = v % (i128)p;
    if (r < 0) r += p;
    if (r > (i128)(p / 2)) r -= p;
    return (double)(long long)r;
}

static u64 h_powmod(u64 a, u64 e, u64 p) {
    unsigned __int128 r = 1, b = a % p;
    while (e) { if (e & 1) r = r * b % p; b = b * b % p; e >>= 1; }
    return (u64)r;
}

int main() {
    const Prime Q1{P1, 1.0 / P1};
    const double p0inv = centred((i128)h_powmod(P0_U % P1_U, P1_U - 2, P1_U), P1_U);
    const i128 Pfull = (i128)P0_U * (i128)P1_U;
    const double k_guard = 0.875 * 0.5 * P1;
    u64 seed = 88172645463325252ull;
    auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    int bad = 0;
    long marked_in = 0, marked_band = 0, marked_over = 0;
    const int T = 200000;
    for (int it = 0; it < T; ++it) {
        // a random integer as a fraction of the half window
        const i128 half = Pfull / 2;
        const i128 r = ((i128)(rnd() >> 1) << 34) | (rnd() >> 30);          // 97 random bits
        const int zone = it % 3;                                              // 0: |V| < 0.87 half, 1: band, 2: beyond the window
        i128 mag;
        if (zone == 0) mag = r % (half / 1000 * 870);
        else if (zone == 1) mag = half / 1000 * 876 + r % (half / 1000 * 123);
        else mag = half + 1 + r % (half / 1000 * 120);
        const i128 V = (rnd() & 1) ? mag : -mag;
        bool near_edge = false;
        const u64 got = crt_wrap_checked(centred(V, P0_U), centred(V, P1_U), p0inv, Q1, k_guard, near_edge);
        const u64 want = (u64)V;                                              // the integer wrapped to 2^64
        if (zone == 0) { if (got != want || near_edge) ++bad; marked_in += near_edge; }
        else if (zone == 1) { if (got != want || !near_edge) ++bad; marked_band += near_edge; }
        else { if (got == want || !near_edge) ++bad; marked_over += near_edge; }   // wrong by P, and seen
    }
    printf("lift guard: %d trials, marked %ld / %ld / %ld of %d inside / in the band / beyond the window; mismatches: %d\n", T,
           marked_in, marked_band, marked_over, T / 3, bad);
    puts(bad ? "FAILED" : "OK");
    return bad ? 1 : 0;
}
