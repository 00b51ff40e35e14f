// rgbm_numerics.h -- the numerics spec of librepairgbm (DESIGN.md "Numerics"), host + device.
//
// Everything that decides a bit of the result lives here and is compiled with
// -ffp-contract=off for both the host (clang) and the gfx950 device pass, so the same IEEE
// double operations run in the same order on the CPU and on the GPU:
//   * rg_exp      own exp(): 2^k * P13(r), plain mul/add Horner (no libm/ocml dependence)
//   * fx_from_f32 float32 gradient / hessian -> the model's fixed-point grid (exact integer histogram sums, numerics v2.2)
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

// Numerics v2.2.  The gradient and hessian of a (row, class tree) are LightGBM's own float32 values (score_t; *_objective.hpp
// GetGradients: the double expression rounded once to float).  A histogram sum is the EXACT integer sum of those floats on a fixed-point
// grid: fx = rint(v * 2^e).  Integer sums are associative, so a histogram does not depend on lanes, workgroups, launch geometry or the
// number of GPUs; and because a double sum of float32 values is itself exact until it outgrows 53 bits, the sums equal LightGBM's own
// wherever those did not have to round (tests/test_numerics_bound.py).
//   v2.1 chose ONE e per model from worst-case bounds: every converted value stays at or below 2^50 in magnitude (the range of the rint
//   trick below), and the int64 sum over every training row stays below 2^62 -- with |v_i| <= (bound / w_max) * w_i that sum is at most
//   bound * (sum of the row weights) / w_max (fx_exponent).  At 10M (100M) equally weighted rows that leaves 38 (35) bits below the bound,
//   and a skewed many-class attribute -- class weights spread 100x, off-class probabilities of 1e-5 -- lost up to 4e-3 in a probability and,
//   on the 100M-row grid, 2 of 91 arg-max labels against LightGBM's double sums (profiles/r5_numerics_at_scale.txt).
//   v2.2 chooses e PER CLASS TREE AND BOOSTING ITERATION from what the gradients of that tree actually add up to.  Every (g, h) of the
//   iteration's bag has a coarse magnitude q = ceil(|v| * 2^c), c = 24 - ceil_log2(bound) (fx_coarse: an integer <= 2^24), and
//   Q = the exact integer sum of q over the bag (of all ranks) bounds the sum of the magnitudes: sum |v| <= Q * 2^-c.  With
//   e = c + 62 - ceil_log2(Q) every row's |rint(v 2^e)| <= |v| 2^e + 1/2 <= 1.5 q 2^(e - c)  (q >= 1 wherever v != 0, and e > c), so the
//   magnitudes of ANY set of rows add up to less than 1.5 * 2^62 < 2^63: no int64 sum, partial or total, can overflow.  e is clamped to
//   [the v2.1 exponent of the model (whose bound holds whatever Q is), 50 - ceil_log2(bound)] (fx_tree_exponent).  A one-against-the-rest
//   class tree of a K-class target holds ~2/K of the worst case at the first iteration and less as the fit improves: +4 bits for the
//   K = 64 synthetic target at once, up to +12 (10M rows) / +15 (100M rows) later -- hospital `Score` (55 classes) and `Sample` (303) are
//   LightGBM's arithmetic tree for tree on the 10M-row grid, and within 1e-10 with every label equal on the 100M-row grid.
//   rint by the 1.5 * 2^52 trick: one IEEE add rounds x to the nearest integer (ties to even) for |x| < 2^51, and the low mantissa bits
//   of the sum are that integer in two's complement.  The oracle calls rint(): same value.
inline int fx_ceil_log2(double v) { int ex; double m = frexp(v, &ex); return (m == 0.5) ? ex - 1 : ex; }
inline int fx_exponent(double bound /* of one value, at the heaviest row */, double weight_ratio /* sum of the row weights / largest row weight */) {
    const int by_value = 50 - fx_ceil_log2(bound);
    const int by_sum = 62 - fx_ceil_log2(bound * (weight_ratio > 2.0 ? weight_ratio : 2.0));
    return by_value < by_sum ? by_value : by_sum;
}
// what fit_setup derives once per model (host) and every kernel that measures or picks a grid reads
struct FxGrid {
    int32_t c_g, c_h;             // coarse exponents: |g| * 2^c_g <= 2^24, h * 2^c_h <= 2^24
    int32_t e_g_min, e_h_min;     // the v2.1 exponents (worst-case sum bound): the floor of every class tree's
    int32_t e_g_max, e_h_max;     // 50 - ceil_log2(bound): one value stays inside the rint trick
    long long q_mult;             // 1; test hook RGBM_FX_ROWS: ceil(R / training rows) -- Q as if the table held R rows
};
// the grid of one class tree in one boosting iteration (device table [K], rewritten by k_fx_scale after the gradients)
struct FxScale { double sg, sh, inv_sg, inv_sh; };      // 2^e_g, 2^e_h, 2^-e_g, 2^-e_h

RG_HD unsigned int fx_coarse(float v, int c) {           // ceil(|v| * 2^c): exact (a power-of-two scaling and a ceil of a double)
    double x = ceil(ldexp(fabs((double)v), c));
    if (x > 2147483648.0) x = 2147483648.0;              // (never reached: |v| <= bound by construction of c)
    return (unsigned int)x;
}
RG_HD int fx_ceil_log2_u64(unsigned long long q) {       // smallest n with q <= 2^n
    int n = 0;
    if (q <= 1ull) return 0;
    --q;
    while (q) { ++n; q >>= 1; }
    return n;
}
RG_HD int fx_tree_exponent(unsigned long long Q, long long q_mult, int c, int e_min, int e_max) {
    const unsigned long long q = Q * (unsigned long long)q_mult;
    int e = (q == 0ull) ? e_max : c + 62 - fx_ceil_log2_u64(q);
    if (e > e_max) e = e_max;
    if (e < e_min) e = e_min;
    return e;
}
RG_HD double fx_pow2(int e) {                            // 2^e for -1022 <= e <= 1023, from its bits
    union { unsigned long long u; double d; } x;
    x.u = (unsigned long long)(1023 + e) << 52;
    return x.d;
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
