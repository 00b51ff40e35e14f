// rgbm_numerics.h -- the numerics spec of librepairgbm (DESIGN.md "Numerics"), host + device.
//
// Everything that decides a bit of the result lives here and is compiled with
// -ffp-contract=off for both the host (clang) and the gfx950 device pass, so the same IEEE
// double operations run in the same order on the CPU and on the GPU:
//   * rg_exp      own exp(): 2^k * P13(r), plain mul/add Horner (no libm/ocml dependence)
//   * fx_from_f32 float32 gradient / hessian -> the model's fixed-point grid (exact integer histogram sums, numerics v2)
//   * leaf_gain / leaf_output / threshold_l1   LightGBM feature_histogram.hpp formulas
//     (GetLeafGain, CalculateSplittedLeafOutput, ThresholdL1) as reached from
//     python/repair/train.py:102-115 (no max_delta_step, no path smoothing, no monotone).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define RG_HD __host__ __device__ __forceinline__
#else
#define RG_HD inline
#endif

namespace rg {

constexpr int TILE_ROWS = 2048;         // rows per tile of the leaf-wise histogram kernel
RG_HD double k_eps() { return (double)1e-15f; }   // LightGBM kEpsilon (a float literal)

RG_HD double rg_exp(double x) {
    if (x != x) return x;
    if (x > 709.0) return INFINITY;
    if (x < -745.0) return 0.0;
    const double INV_LN2 = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kd = rint(x * INV_LN2);
    double r = (x - kd * LN2_HI) - kd * LN2_LO;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int k = (int)kd;
    int k1 = k / 2, k2 = k - k1;
    union { uint64_t u; double d; } a, b;
    a.u = (uint64_t)(1023 + k1) << 52;
    b.u = (uint64_t)(1023 + k2) << 52;
    return (p * a.d) * b.d;
}

RG_HD double threshold_l1(double s, double l1) {
    double reg = fabs(s) - l1;
    if (reg < 0.0) reg = 0.0;
    return (s > 0.0 ? 1.0 : (s < 0.0 ? -1.0 : 0.0)) * reg;
}
RG_HD double leaf_output(double G, double H, double l1, double l2) {
    double sg = (l1 > 0.0) ? threshold_l1(G, l1) : G;
    return -sg / (H + l2);
}
RG_HD double leaf_gain(double G, double H, double l1, double l2) {
    double sg = (l1 > 0.0) ? threshold_l1(G, l1) : G;
    return (sg * sg) / (H + l2);
}

RG_HD long long round_int(double x) { return (long long)(x + 0.5); }

// Numerics v2.1.  The gradient and hessian of a (row, class tree) are LightGBM's own float32 values (score_t; *_objective.hpp
// GetGradients: the double expression rounded once to float).  A histogram sum is the EXACT integer sum of those floats on a per-model
// fixed-point grid: fx = rint(v * 2^e).  The host picks e (fx_exponent) from two bounds: every converted value stays at or below 2^50
// in magnitude (the range of the rint trick below), and the int64 sum over every training row stays below 2^62 -- with
// |v_i| <= (bound / w_max) * w_i that sum is at most bound * (sum of the row weights) / w_max.  A float32 whose magnitude is at least
// 2^-27 of the bound is on the grid exactly (v2.0 capped the grid at 2^40: exact down to 2^-17 only, which cost hospital's many-class
// attributes 4e-3 in a probability against LightGBM's double sums).  Integer sums are associative, so a histogram does not depend on
// lanes, workgroups, launch geometry or the number of GPUs; and because a double sum of float32 values is itself exact until it outgrows
// 53 bits, the sums equal LightGBM's own wherever those did not have to round (tests/test_numerics_bound.py).
//   rint by the 1.5 * 2^52 trick: one IEEE add rounds x to the nearest integer (ties to even) for |x| < 2^51, and the low mantissa bits
//   of the sum are that integer in two's complement.  The oracle calls rint(): same value.
inline int fx_ceil_log2(double v) { int ex; double m = frexp(v, &ex); return (m == 0.5) ? ex - 1 : ex; }
inline int fx_exponent(double bound /* of one value, at the heaviest row */, double weight_ratio /* sum of the row weights / largest row weight */) {
    const int by_value = 50 - fx_ceil_log2(bound);
    const int by_sum = 62 - fx_ceil_log2(bound * (weight_ratio > 2.0 ? weight_ratio : 2.0));
    return by_value < by_sum ? by_value : by_sum;
}
RG_HD long long fx_from_f32(float v, double scale /* 2^e */) {
    double x = (double)v * scale;
    const double lim = 1125899906842624.0;           // 2^50 (|x| <= 2^50 by construction of e)
    if (x > lim) x = lim;
    if (x < -lim) x = -lim;
    const double magic = 6755399441055744.0;         // 1.5 * 2^52
    union { double d; long long i; } u, m;
    u.d = x + magic; m.d = magic;
    return u.i - m.i;
}

}  // namespace rg
