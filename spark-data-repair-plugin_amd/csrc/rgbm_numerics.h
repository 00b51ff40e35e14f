// rgbm_numerics.h -- the numerics spec of librepairgbm (DESIGN.md "Numerics"), host + device.
//
// Everything that decides a bit of the result lives here and is compiled with
// -ffp-contract=off for both the host (clang) and the gfx950 device pass, so the same IEEE
// double operations run in the same order on the CPU and on the GPU:
//   * rg_exp      own exp(): 2^k * P13(r), plain mul/add Horner (no libm/ocml dependence)
//   * quantise    gradients/hessians -> fixed point (|gq| <= 2^20-1, 0 <= hq <= 2^21-1)
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

constexpr int GQ_MAX = (1 << 20) - 1;   // |quantised gradient|
constexpr int HQ_MAX = (1 << 21) - 1;   // quantised hessian
constexpr int TILE_ROWS = 2048;         // rows between two drains of the packed LDS histogram:
                                        // 2048 * GQ_MAX < 2^31 and 2048 * HQ_MAX < 2^32
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

RG_HD int quant_g(double g, double sg) {
    double a = rint(g * sg);
    if (a > (double)GQ_MAX) a = (double)GQ_MAX;
    if (a < -(double)GQ_MAX) a = -(double)GQ_MAX;
    return (int)a;
}
RG_HD int quant_h(double h, double sh) {
    double b = rint(h * sh);
    if (b > (double)HQ_MAX) b = (double)HQ_MAX;
    return (int)b;
}
RG_HD long long round_int(double x) { return (long long)(x + 0.5); }

// Numerics v1.02: the quantised hessian is a FUNCTION of the quantised gradient (and of the row's label and weight), not of
// the unrounded probability.  For the binary / softmax objectives g = (p - [y is this class]) * w and h = factor * p * (1 - p) * w, so
// p -- and with it h -- is recovered from g to within the gradient's own resolution (2^-20 of its bound, the same order as the
// rounding of h itself).  This is what lets the level passes stream 4 bytes per (row, class tree) instead of 8: they carry g only
// and recompute h for the rows they accumulate.  Regression: h = w, independent of g.
//   obj 0: g = response * w, h = |response| (1 - |response|) w;  obj 1: as above;  obj 2: h = w.
RG_HD int h_from_g(int gq, bool is_label_class, double w, double inv_w /* 1.0 / w, or 0 for w = 0 */, int obj, double inv_sg, double sh, double factor) {
    if (obj == 2) return quant_h(w, sh);
    const double a = ((double)gq * inv_sg) * inv_w;      // the reciprocal is taken once per label / row: no division per (row, class tree)
    double h;
    if (obj == 0) { const double r = fabs(a); h = r * (1.0 - r) * w; }
    else { const double p = is_label_class ? a + 1.0 : a; h = factor * p * (1.0 - p) * w; }
    if (!(h > 0.0)) h = 0.0;       // p rounded a hair outside [0, 1]; w = 0
    return quant_h(h, sh);
}
RG_HD double rg_inv_weight(double w) { return w > 0.0 ? 1.0 / w : 0.0; }

}  // namespace rg
