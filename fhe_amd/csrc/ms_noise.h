// Modulus-switch noise reduction (the "drift" technique of the TUNIFORM_2M128 parameter sets).
//
// Rounding the 919 words of a key-switched ciphertext to Z_{2N} adds the error
//     e_b - sum_i e_i s_i          (e = rounded - exact, s the binary LWE key),
// whose mean over the key is  e_b - sum e_i / 2  and whose variance is  sum e_i^2 / 4.  The server
// key carries encryptions of zero under the small key; before the modulus switch the first
// candidate  ct + zero_k  (k = none, 0, 1, ...) whose estimate
//     |mean| + r_sigma * sqrt(input_variance + variance)
// is within the bound is taken, or the best one if none is.  The estimate is evaluated in fixed
// point on integers so that every implementation picks the same candidate:
//     e       signed, units of 2^-64 of the torus
//     s1x2  = 2 e_b - sum e_i                      (units 2^-65)
//     s2    = sum (e_i >> 26)^2                    (units 2^-76)
//     v     = round(input_variance * 2^76) + (s2 >> 2)
//     est   = (|s1x2| >> 27) + ((round(r_sigma * 2^16) * isqrt(v)) >> 16)      (units 2^-38)
//     bound = 2^(bound_log2 - 26)
#pragma once
#include "pbs_device.h"

namespace lfhe {

struct MsParams {
    u64 var_in_fx = 0;   // round(input_variance * 2^76)
    u64 r_fx = 0;        // round(r_sigma * 2^16)
    u64 bound_fx = 0;    // 2^(bound_log2 - 26)
    u32 log2_2n = LOGN + 1;
    u32 zcount = 0;      // number of encryptions of zero (0: the step is skipped)
};

LFHE_HD i64 ms_round_error(u64 a, u32 log2_2n) {
    u64 t = a >> (64 - log2_2n - 1);
    t = (t + 1) >> 1;
    return (i64)((t << (64 - log2_2n)) - a);
}

LFHE_HD u64 isqrt_u64(u64 v) {  // floor(sqrt(v)), v < 2^62
    u64 s = (u64)__builtin_sqrt((double)v);
    while (s * s > v) --s;
    while ((s + 1) * (s + 1) <= v) ++s;
    return s;
}

LFHE_HD u64 ms_estimate(i64 s1x2, u64 s2, const MsParams& m) {
    const u64 sd = isqrt_u64(m.var_in_fx + (s2 >> 2));
    const u64 ex = (u64)(s1x2 < 0 ? -s1x2 : s1x2) >> 27;
    return ex + ((m.r_fx * sd) >> 16);
}

}  // namespace lfhe
