// Host-side construction of the transform tables (exact integer arithmetic, then converted to
// centred f64 residues).  Header-only; used by the engine and by the host-side phase emulation.
#pragma once
#include <stdint.h>
#include <vector>

#include "pbs_device.h"

namespace lfhe {

typedef unsigned __int128 u128;

inline u64 h_mulmod(u64 a, u64 b, u64 p) { return (u64)(((u128)a * b) % p); }
inline u64 h_powmod(u64 a, u64 e, u64 p) {
    u64 r = 1;
    a %= p;
    while (e) {
        if (e & 1) r = h_mulmod(r, a, p);
        a = h_mulmod(a, a, p);
        e >>= 1;
    }
    return r;
}
inline u32 h_bitrev(u32 x, int bits) {
    u32 r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
inline double h_centred(u64 x, u64 p) { return x > p / 2 ? -(double)(p - x) : (double)x; }

struct HostTables {
    u64 p[2];
    u64 psi[2];                      // primitive 2N-th roots
    std::vector<double> tw_fwd[2];   // [N]
    std::vector<double> tw_inv[2];   // [N]
    double two32[2], ninv[2], p0inv_mod_p1;
};

inline void build_host_tables(HostTables& H) {
    H.p[0] = P0_U;
    H.p[1] = P1_U;
    for (int r = 0; r < 2; ++r) {
        const u64 p = H.p[r];
        u64 psi = 0;
        for (u64 g = 2; !psi; ++g) {
            u64 c = h_powmod(g, (p - 1) / (2 * (u64)N), p);
            if (h_powmod(c, N, p) == p - 1) psi = c;
        }
        H.psi[r] = psi;
        const u64 psi_inv = h_powmod(psi, p - 2, p);
        H.tw_fwd[r].resize(N);
        H.tw_inv[r].resize(N);
        for (int i = 0; i < N; ++i) {
            const u32 e = h_bitrev((u32)i, LOGN);
            H.tw_fwd[r][i] = h_centred(h_powmod(psi, e, p), p);
            H.tw_inv[r][i] = h_centred(h_powmod(psi_inv, e, p), p);
        }
        H.two32[r] = h_centred((u64)(((u128)1 << 32) % p), p);
        H.ninv[r] = h_centred(h_powmod(N, p - 2, p), p);
    }
    H.p0inv_mod_p1 = h_centred(h_powmod(H.p[0] % H.p[1], H.p[1] - 2, H.p[1]), H.p[1]);
}

}  // namespace lfhe
