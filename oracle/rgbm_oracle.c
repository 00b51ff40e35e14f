/*
 * rgbm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C, single-threaded restatement of the per-attribute repair-model hot path of
 * maropu/spark-data-repair-plugin:
 *   python/repair/train.py:97-131   objective pick + fixed LightGBM parameters
 *   python/repair/train.py:215-216  model.fit(X, y)
 *   python/repair/model.py:1118-1133 predict_proba / predict / fill-only-NULL chain
 * whose arithmetic lives in the un-vendored third-party dependency lightgbm==3.3.1
 * (bin/requirements.txt:6).  LightGBM's source is NOT under /root/reference and is not
 * installable offline, so this file restates its published algorithm (bin.cpp GreedyFindBin /
 * FindBinWithZeroAsOneBin, {binary,multiclass,regression}_objective.hpp GetGradients /
 * BoostFromScore, feature_histogram.hpp FindBestThresholdSequentially, serial_tree_learner.cpp
 * leaf-wise growth, gbdt.cpp TrainOneIter, tree.h NumericalDecision, sklearn.py predict_proba)
 * from memory.
 *
 * PARITY STATUS: "parity unpinned" against real LightGBM bits (cannot be run here; tests/test_true_reference_optional.py is the
 * hook for the day the wheel exists).  What IS pinned: the reference's own golden labels (bin/testdata/adult_repair.csv,
 * test_model.py inline goldens) -- tests/test_oracle_golden.py --, tree growth node by node against a hand-computed tree and
 * scikit-learn's histogram GBDT for all three objectives (tests/test_oracle_split_pin.py), and the `spec` mode against the
 * `lightgbm_f32` mode of this file (tests/test_numerics_bound.py).  The HIP product is held bit-exact against THIS file.
 *
 * Deliberate, documented deviations from LightGBM 3.3.1:
 *  D1. (numerics v2.2) gradients/hessians are LightGBM's float32 values; a histogram sum is their EXACT integer sum on a fixed-point
 *      grid chosen per class tree and boosting iteration from the coarse sum of that tree's gradient magnitudes (see "Numerics v2.2"
 *      below) instead of a double accumulated in row order => order independent => bit-reproducible on a GPU, and equal to LightGBM's
 *      double sums wherever those did not round (rgbm_oracle_train.inc builds both; tests/test_numerics_bound.py compares them).
 *  D2. exp() is an own polynomial (rg_exp) so CPU and GPU produce identical bits.
 *  D3. features are the int32 label codes (ordinal, order preserving); NULL/unknown = -1 is
 *      LightGBM's NaN.  Bin boundaries are found on ALL training rows (no 200k sub-sample).
 *  D4. bagging / feature sampling use LightGBM's LCG as remembered (unverified).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>

#define ORC_API __attribute__((visibility("default")))

typedef struct {
    int32_t objective;        /* 0 binary, 1 multiclass, 2 regression (L2) */
    int32_t num_class;        /* classes (multiclass); 2 for binary; ignored for regression */
    int32_t n_estimators, num_leaves, max_depth, max_bin;
    int32_t min_data_in_leaf, min_data_in_bin, bagging_freq, seed;
    int32_t device_id, reserved;
    double learning_rate, lambda_l1, lambda_l2, min_gain_to_split;
    double min_sum_hessian_in_leaf, bagging_fraction, feature_fraction;
} orc_params;

typedef struct {
    int32_t n_codes;   /* dictionary size of the column */
    int32_t V;         /* number of value bins */
    int32_t has_nan;   /* 1 => an extra bin (index V) holds NULL rows (LightGBM MissingType::NaN) */
    int32_t* ub;       /* [V] last code that falls into value bin b (last entry INT32_MAX) */
    uint32_t* unseen;  /* CATEGORICAL features only, or NULL: bit c set => no training row held code c; such a category is MISSING for
                        * this model at prediction time (it has no place in the order of the codes), like a value outside the dictionary */
    int32_t n_unseen_words;
} orc_feat;

typedef struct {
    int32_t L;                 /* leaves */
    int32_t *feat, *theta, *dleft, *left, *right;   /* [L-1] internal nodes; child <0 => ~leaf */
    double* gain;              /* [L-1] */
    double* leaf_value;        /* [L] */
    int32_t* leaf_count;       /* [L] */
} orc_tree;

typedef struct orc_model {
    int32_t objective, num_class, K /* trees per iteration */, n_iter, F;
    orc_feat* feats;
    orc_tree* trees;           /* [n_iter*K] */
} orc_model;

static const double kEps = (double)1e-15f;      /* LightGBM kEpsilon is a float literal */

/* ------------------------------------------------------------------ numerics: exp (D2) */
/* exp(x) = 2^k * exp(r), k = rint(x/ln2), r = x - k*ln2 (two-part), exp(r) by a degree-13
 * Taylor polynomial in Horner form; every operation is a plain IEEE mul/add (no fma). */
static double rg_exp(double x) {
    if (x != x) return x;
    if (x > 709.0) return INFINITY;
    if (x < -745.0) return 0.0;
    const double INV_LN2 = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kd = rint(x * INV_LN2);
    double r = (x - kd * LN2_HI) - kd * LN2_LO;
    double p = 1.0 / 6227020800.0;                 /* 1/13! */
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
    /* scale by 2^k in two exact steps so that subnormal results round once */
    int k1 = k / 2, k2 = k - k1;
    union { uint64_t u; double d; } a, b;
    a.u = (uint64_t)(1023 + k1) << 52;
    b.u = (uint64_t)(1023 + k2) << 52;
    return (p * a.d) * b.d;
}

static double threshold_l1(double s, double l1) {
    double reg = fabs(s) - l1;
    if (reg < 0.0) reg = 0.0;
    return (s > 0.0 ? 1.0 : (s < 0.0 ? -1.0 : 0.0)) * reg;
}
static double leaf_output(double G, double H, double l1, double l2) {
    double sg = (l1 > 0.0) ? threshold_l1(G, l1) : G;
    return -sg / (H + l2);
}
static double leaf_gain(double G, double H, double l1, double l2) {
    double sg = (l1 > 0.0) ? threshold_l1(G, l1) : G;
    return (sg * sg) / (H + l2);
}

/* ------------------------------------------------------------------ binning (bin.cpp) */
/* GreedyFindBin restated in code space: distinct values are the codes with cnt>0, a bin
 * boundary between consecutive seen codes a<b is floor((a+b)/2) (the integer codes that are
 * <= the LightGBM midpoint).  Returns number of bins, fills ub (last code per bin). */
/* Bin upper bound between two neighbouring training codes a < b.  Codes of a NUMERIC column are ranks of its distinct values;
 * LightGBM puts the bound at the midpoint of the VALUES, so a code that lies between a and b (a value that only rows outside the
 * training set hold) goes left iff its value is <= (val[a] + val[b]) / 2: the bound in code space is the largest such code.
 * Without a value dictionary (categorical columns, host-array calls) the codes themselves are the scale. */
static int32_t mid_code(int32_t a, int32_t b, const double* vals) {
    if (!vals) return (int32_t)(((int64_t)a + (int64_t)b) / 2);
    const double m = (vals[a] + vals[b]) / 2.0;
    int32_t lo = a, hi = b - 1;                 /* vals[a] <= m always */
    while (lo < hi) { int32_t c = lo + (hi - lo + 1) / 2; if (vals[c] <= m) lo = c; else hi = c - 1; }
    return lo;
}

static int greedy_find_bin(const int32_t* dv, const int64_t* cnt, int nd, int max_bin,
                           int64_t total_cnt, int min_data_in_bin, int32_t* ub, const double* vals) {
    int nb = 0;
    if (nd <= 0) return 0;
    if (nd <= max_bin) {
        int64_t cur = 0;
        for (int i = 0; i < nd - 1; ++i) {
            cur += cnt[i];
            if (cur >= min_data_in_bin) {
                ub[nb++] = mid_code(dv[i], dv[i + 1], vals);
                cur = 0;
            }
        }
        ub[nb++] = INT32_MAX;
        return nb;
    }
    if (min_data_in_bin > 0) {
        int64_t m = total_cnt / min_data_in_bin;
        if (m < max_bin) max_bin = (int)m;
        if (max_bin < 1) max_bin = 1;
    }
    double mean_bin_size = (double)total_cnt / max_bin;
    int rest_bin_cnt = max_bin;
    int64_t rest_sample_cnt = total_cnt;
    char* big = (char*)calloc(nd, 1);
    for (int i = 0; i < nd; ++i) {
        if ((double)cnt[i] >= mean_bin_size) { big[i] = 1; --rest_bin_cnt; rest_sample_cnt -= cnt[i]; }
    }
    mean_bin_size = (double)rest_sample_cnt / rest_bin_cnt;
    int32_t* upper = (int32_t*)malloc(sizeof(int32_t) * max_bin);
    int32_t* lower = (int32_t*)malloc(sizeof(int32_t) * max_bin);
    int bin_cnt = 0;
    lower[0] = dv[0];
    int64_t cur = 0;
    for (int i = 0; i < nd - 1; ++i) {
        if (!big[i]) rest_sample_cnt -= cnt[i];
        cur += cnt[i];
        double half = mean_bin_size * 0.5f;
        if (half < 1.0) half = 1.0;
        if (big[i] || (double)cur >= mean_bin_size || (big[i + 1] && (double)cur >= half)) {
            upper[bin_cnt] = dv[i];
            ++bin_cnt;
            lower[bin_cnt] = dv[i + 1];
            if (bin_cnt >= max_bin - 1) break;
            cur = 0;
            if (!big[i]) { --rest_bin_cnt; mean_bin_size = (double)rest_sample_cnt / (double)rest_bin_cnt; }
        }
    }
    ++bin_cnt;
    for (int i = 0; i < bin_cnt - 1; ++i) {
        int32_t v = mid_code(upper[i], lower[i + 1], vals);
        if (nb == 0 || ub[nb - 1] != v) ub[nb++] = v;
    }
    ub[nb++] = INT32_MAX;
    free(big); free(upper); free(lower);
    return nb;
}

/* FindBin for one feature over the training rows.  All codes are shifted to positive values
 * (OrdinalEncoder emits 1..n), so FindBinWithZeroAsOneBin gives an (always empty) zero bin and
 * max_bin-1 bins to the values; one more bin is reserved when NULLs are present. */
static void find_bin(const int32_t* col, int64_t n, int32_t n_codes, const orc_params* p, orc_feat* f, const double* vals, int categorical) {
    int64_t* cnt = (int64_t*)calloc((size_t)(n_codes > 0 ? n_codes : 1), sizeof(int64_t));
    int64_t na = 0;
    for (int64_t i = 0; i < n; ++i) { int32_t c = col[i]; if (c < 0 || c >= n_codes) ++na; else ++cnt[c]; }
    int nd = 0;
    for (int32_t c = 0; c < n_codes; ++c) if (cnt[c] > 0) ++nd;
    int32_t* dv = (int32_t*)malloc(sizeof(int32_t) * (nd > 0 ? nd : 1));
    int64_t* dc = (int64_t*)malloc(sizeof(int64_t) * (nd > 0 ? nd : 1));
    nd = 0;
    for (int32_t c = 0; c < n_codes; ++c) if (cnt[c] > 0) { dv[nd] = c; dc[nd] = cnt[c]; ++nd; }
    int mb = p->max_bin - (na > 0 ? 1 : 0);   /* NaN bin */
    mb -= 1;                                  /* zero bin of FindBinWithZeroAsOneBin */
    if (mb < 1) mb = 1;
    f->n_codes = n_codes;
    f->has_nan = na > 0 ? 1 : 0;
    f->ub = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nd > 0 ? (nd < mb + 1 ? nd + 1 : mb + 1) : 1));
    f->V = greedy_find_bin(dv, dc, nd, mb, n - na, p->min_data_in_bin, f->ub, vals);
    f->unseen = NULL; f->n_unseen_words = 0;
    if (categorical && nd < n_codes) {
        f->n_unseen_words = (n_codes + 31) / 32;
        f->unseen = (uint32_t*)calloc((size_t)f->n_unseen_words, sizeof(uint32_t));
        for (int32_t c = 0; c < n_codes; ++c) if (cnt[c] == 0) f->unseen[c >> 5] |= 1u << (c & 31);
    }
    free(cnt); free(dv); free(dc);
}

static inline int code_to_bin(const orc_feat* f, int32_t c) {
    if (c < 0 || c >= f->n_codes) return -1;   /* missing */
    if (f->unseen && ((f->unseen[c >> 5] >> (c & 31)) & 1u)) return -1;
    int lo = 0, hi = f->V - 1;                 /* first b with c <= ub[b] */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (c <= f->ub[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}

/* ------------------------------------------------------------------ LightGBM's Random (D4) */
typedef struct { uint32_t x; } lgb_rand;
static inline int rnd16(lgb_rand* r) { r->x = 214013u * r->x + 2531011u; return (int)((r->x >> 16) & 0x7FFF); }
static inline int rnd32(lgb_rand* r) { r->x = 214013u * r->x + 2531011u; return (int)(r->x & 0x7FFFFFFF); }
static inline float rnd_float(lgb_rand* r) { return (float)rnd16(r) / 32768.0f; }
static inline int rnd_int(lgb_rand* r, int lo, int hi) { return rnd32(r) % (hi - lo) + lo; }

/* Random::Sample(N, K) -> ascending indices */
static int rnd_sample(lgb_rand* r, int N, int K, int* out) {
    int m = 0;
    if (K > N || K <= 0) return 0;
    if (K == N) { for (int i = 0; i < N; ++i) out[m++] = i; return m; }
    if (K > 1 && (double)K > ((double)N / log2((double)K))) {
        for (int i = 0; i < N; ++i) {
            double prob = (double)(K - m) / (double)(N - i);
            if (rnd_float(r) < prob) out[m++] = i;
        }
        return m;
    }
    char* in = (char*)calloc(N, 1);
    for (int rr = N - K; rr < N; ++rr) {
        int v = rnd_int(r, 0, rr);
        if (in[v]) in[rr] = 1; else in[v] = 1;
    }
    for (int i = 0; i < N; ++i) if (in[i]) out[m++] = i;
    free(in);
    return m;
}

/* ------------------------------------------------------------------ pieces shared by both numerics modes */
static inline int64_t round_int(double x) { return (int64_t)(x + 0.5); }

/* Numerics v2 (csrc/rgbm_numerics.h fx_from_f32): a float32 gradient / hessian on the model's fixed-point grid -- v * 2^e rounded to
 * the nearest integer (ties to even), clamped to +-2^50 (never reached: |v * 2^e| <= 2^E <= 2^40 by construction of e). */
static inline int64_t fx_from_f32(float v, double scale) {
    double x = (double)v * scale;
    if (x > 1125899906842624.0) x = 1125899906842624.0;
    if (x < -1125899906842624.0) x = -1125899906842624.0;
    return (int64_t)rint(x);
}

/* Numerics v2.2 (csrc/rgbm_numerics.h): the fixed-point grid is chosen PER CLASS TREE AND BOOSTING ITERATION.  Every in-bag (g, h) has a coarse
 * magnitude q = ceil(|v| * 2^c) with c = 24 - ceil_log2(bound) (an integer <= 2^24); Q = the exact integer sum of q over the rows of the
 * iteration's bag bounds the sum of |v|: sum |v| <= Q * 2^-c.  With e = c + 62 - ceil_log2(Q) every row's |rint(v 2^e)| <= |v| 2^e + 1/2 <=
 * 1.5 q 2^(e-c) (q >= 1 wherever v != 0 and e > c), so the sum of the magnitudes over ANY set of rows stays below 1.5 * 2^62 < 2^63: an int64
 * sum cannot overflow.  e never drops below the v2.1 exponent of the model (whose own bound holds whatever Q is) and never exceeds
 * 50 - ceil_log2(bound) (one value stays within the range of the device's rint trick).
 * RGBM_FX_ROWS = R (with RGBM_TEST_HOOKS=1; the product reads the same pair): the grid a table of R rows with this table's gradient
 * distribution gets -- Q is multiplied by ceil(R / N) and the v2.1 floor is the one of R rows. */
static inline int64_t f32_park(float v) { union { float f; uint32_t u; } x; x.f = v; return (int64_t)x.u; }
static inline float f32_unpark(int64_t b) { union { float f; uint32_t u; } x; x.u = (uint32_t)b; return x.f; }
static inline int64_t fx_coarse(float v, int c) { double x = ceil(ldexp(fabs((double)v), c)); if (x > 2147483648.0) x = 2147483648.0; return (int64_t)x; }
static inline int ceil_log2_u64(uint64_t q) { int n = 0; if (q <= 1) return 0; --q; while (q) { ++n; q >>= 1; } return n; }
typedef struct { int c_g, c_h, e_g_max, e_h_max; int64_t q_mult; } fx_grid;
static inline int fx_tree_exponent(int64_t q, int64_t q_mult, int c, int e_min, int e_max) {
    q *= q_mult;
    int e = (q <= 0) ? e_max : c + 62 - ceil_log2_u64((uint64_t)q);
    if (e > e_max) e = e_max;
    if (e < e_min) e = e_min;
    return e;
}
static int fx_test_hooks(void) { const char* ev = getenv("RGBM_TEST_HOOKS"); return ev && atoi(ev) != 0; }

/* Threads of the timing harness (bench.py cpu_baseline): LightGBM's col-wise mode builds the per-feature histograms in
 * parallel; so does this (features are independent output ranges and the sums are integers: bit-identical for any thread
 * count).  Default 1: the tests never change it. */
static int g_threads = 1;
ORC_API void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }

static inline int tree_leaf_for(const orc_tree* tr, const orc_feat* feats, const uint8_t* const* bcols, int64_t r) {
    if (tr->L <= 1) return 0;
    int node = 0;
    for (;;) {
        const orc_feat* f = &feats[tr->feat[node]];
        int bin = bcols[tr->feat[node]][r];
        int go_left = (bin == 255 || (f->has_nan && bin == f->V)) ? tr->dleft[node] : (bin <= tr->theta[node]);
        int nx = go_left ? tr->left[node] : tr->right[node];
        if (nx < 0) return ~nx;
        node = nx;
    }
}

static double pow2(int e) { return ldexp(1.0, e); }
static int ceil_log2(double v) {   /* smallest e with 2^e >= v, v>0 */
    int ex; double m = frexp(v, &ex);   /* v = m*2^ex, m in [0.5,1) */
    return (m == 0.5) ? ex - 1 : ex;
}


ORC_API void orc_model_free(orc_model* m);

/* ------------------------------------------------------------------ the trainer, in both numerics modes (see the .inc) */
/* ------------------------------------------------------------------ GBDT::Train */
ORC_API int orc_train2(const int32_t* X, int64_t N, int32_t F, const int32_t* n_codes,
                       const int32_t* y_code, int32_t n_y_codes, const double* y_value,
                       const double* class_weight, const double* sample_weight,
                       const orc_params* p, const double* const* feat_values, const int32_t* feat_kinds, orc_model** out);
ORC_API int orc_train2_f32(const int32_t* X, int64_t N, int32_t F, const int32_t* n_codes,
                           const int32_t* y_code, int32_t n_y_codes, const double* y_value,
                           const double* class_weight, const double* sample_weight,
                           const orc_params* p, const double* const* feat_values, const int32_t* feat_kinds, orc_model** out);
#define ORC_F32 0
#include "rgbm_oracle_train.inc"
#undef ORC_F32
#define ORC_F32 1
#include "rgbm_oracle_train.inc"
#undef ORC_F32

ORC_API int orc_train(const int32_t* X, int64_t N, int32_t F, const int32_t* n_codes,
                      const int32_t* y_code, int32_t n_y_codes, const double* y_value,
                      const double* class_weight, const double* sample_weight,
                      const orc_params* p, orc_model** out) {
    return orc_train2(X, N, F, n_codes, y_code, n_y_codes, y_value, class_weight, sample_weight, p, NULL, NULL, out);
}


/* ------------------------------------------------------------------ prediction */
/* GBDT::PredictRaw + objective ConvertOutput (sklearn.py predict_proba layout).
 * out: binary [n][2] = {1-p, p}; multiclass [n][K]; regression [n]. */
ORC_API int orc_predict(const orc_model* m, const int32_t* X, int64_t n, int32_t F, double* out) {
    if (!m || F != m->F) return -1;
    const int K = m->K;
    uint8_t* bins = (uint8_t*)malloc((size_t)F * (n > 0 ? n : 1));
    const uint8_t** bcols = (const uint8_t**)malloc(sizeof(uint8_t*) * F);
    for (int f = 0; f < F; ++f) {
        const orc_feat* ft = &m->feats[f]; uint8_t* b = bins + (size_t)f * n; const int32_t* col = X + (size_t)f * n;
        for (int64_t i = 0; i < n; ++i) { int bin = code_to_bin(ft, col[i]); b[i] = (uint8_t)(bin < 0 ? 255 : bin); }
        bcols[f] = b;
    }
    double* raw = (double*)malloc(sizeof(double) * K);
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int it = 0; it < m->n_iter; ++it) {
                const orc_tree* tr = &m->trees[(size_t)it * K + k];
                s += tr->leaf_value[tree_leaf_for(tr, m->feats, bcols, i)];
            }
            raw[k] = s;
        }
        if (m->objective == 0) {
            double pr = 1.0 / (1.0 + rg_exp(-raw[0]));
            out[i * 2] = 1.0 - pr; out[i * 2 + 1] = pr;
        } else if (m->objective == 1) {
            double wmax = raw[0];
            for (int k = 1; k < K; ++k) if (raw[k] > wmax) wmax = raw[k];
            double wsum = 0.0;
            for (int k = 0; k < K; ++k) { raw[k] = rg_exp(raw[k] - wmax); wsum += raw[k]; }
            for (int k = 0; k < K; ++k) out[i * K + k] = raw[k] / wsum;
        } else {
            out[i] = raw[0];
        }
    }
    free(bins); free(bcols); free(raw);
    return 0;
}

/* ------------------------------------------------------------------ (de)serialisation
 * Little-endian, shared layout with the product's rgbm_model_save so tests can memcmp.
 * header: "RGBM" u32 version=1, objective, num_class, K, n_iter, F
 * per feature: n_codes, V, has_nan, ub[V]
 * per tree: L, then feat[L-1] theta[L-1] dleft[L-1] left[L-1] right[L-1] (i32), gain[L-1] (f64),
 *           leaf_value[L] (f64), leaf_count[L] (i32) */
static void put(uint8_t** p, const void* src, size_t n, int write) { if (write) memcpy(*p, src, n); *p += n; }
static size_t serialise(const orc_model* m, uint8_t* buf) {
    int write = buf != NULL; uint8_t* p = buf ? buf : (uint8_t*)0;
    uint8_t* start = p;
    /* version 2 = version 1 + the unseen-category bitmap of every feature; written only when a feature has one */
    int ver = 1;
    for (int f = 0; f < m->F; ++f) if (m->feats[f].unseen) ver = 2;
    int32_t hdr[7] = {0x4D424752, ver, m->objective, m->num_class, m->K, m->n_iter, m->F};
    put(&p, hdr, sizeof(hdr), write);
    for (int f = 0; f < m->F; ++f) {
        int32_t h3[3] = {m->feats[f].n_codes, m->feats[f].V, m->feats[f].has_nan};
        put(&p, h3, sizeof(h3), write);
        put(&p, m->feats[f].ub, sizeof(int32_t) * m->feats[f].V, write);
        if (ver == 2) {
            int32_t nw = m->feats[f].unseen ? m->feats[f].n_unseen_words : 0;
            put(&p, &nw, 4, write);
            if (nw) put(&p, m->feats[f].unseen, 4 * (size_t)nw, write);
        }
    }
    for (int i = 0; i < m->n_iter * m->K; ++i) {
        const orc_tree* t = &m->trees[i]; int n = t->L - 1;
        put(&p, &t->L, 4, write);
        put(&p, t->feat, 4 * n, write); put(&p, t->theta, 4 * n, write); put(&p, t->dleft, 4 * n, write);
        put(&p, t->left, 4 * n, write); put(&p, t->right, 4 * n, write); put(&p, t->gain, 8 * n, write);
        put(&p, t->leaf_value, 8 * t->L, write); put(&p, t->leaf_count, 4 * t->L, write);
    }
    return (size_t)(p - start);
}
ORC_API int orc_model_save(const orc_model* m, void* buf, size_t* len) {
    if (!m || !len) return -1;
    size_t need = serialise(m, NULL);
    if (!buf) { *len = need; return 0; }
    if (*len < need) { *len = need; return -2; }
    serialise(m, (uint8_t*)buf); *len = need; return 0;
}
ORC_API int orc_model_load(const void* buf, size_t len, orc_model** out) {
    const uint8_t* p = (const uint8_t*)buf; const uint8_t* end = p + len;
    if (len < 28) return -1;
    int32_t hdr[7]; memcpy(hdr, p, 28); p += 28;
    if (hdr[0] != 0x4D424752 || (hdr[1] != 1 && hdr[1] != 2)) return -1;
    orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
    m->objective = hdr[2]; m->num_class = hdr[3]; m->K = hdr[4]; m->n_iter = hdr[5]; m->F = hdr[6];
    m->feats = (orc_feat*)calloc(m->F > 0 ? m->F : 1, sizeof(orc_feat));
    for (int f = 0; f < m->F; ++f) {
        if (p + 12 > end) { orc_model_free(m); return -1; }
        int32_t h3[3]; memcpy(h3, p, 12); p += 12;
        m->feats[f].n_codes = h3[0]; m->feats[f].V = h3[1]; m->feats[f].has_nan = h3[2];
        if (h3[1] < 0 || p + 4 * (size_t)h3[1] > end) { orc_model_free(m); return -1; }
        m->feats[f].ub = (int32_t*)malloc(4 * (size_t)(h3[1] > 0 ? h3[1] : 1)); memcpy(m->feats[f].ub, p, 4 * (size_t)h3[1]); p += 4 * (size_t)h3[1];
        if (hdr[1] == 2) {
            int32_t nw; if (p + 4 > end) { orc_model_free(m); return -1; }
            memcpy(&nw, p, 4); p += 4;
            if (nw < 0 || (nw != 0 && nw != (h3[0] + 31) / 32) || p + 4 * (size_t)nw > end) { orc_model_free(m); return -1; }
            if (nw) { m->feats[f].unseen = (uint32_t*)malloc(4 * (size_t)nw); memcpy(m->feats[f].unseen, p, 4 * (size_t)nw); m->feats[f].n_unseen_words = nw; p += 4 * (size_t)nw; }
        }
    }
    int nt = m->n_iter * m->K;
    m->trees = (orc_tree*)calloc(nt > 0 ? nt : 1, sizeof(orc_tree));
    for (int i = 0; i < nt; ++i) {
        orc_tree* t = &m->trees[i];
        if (p + 4 > end) { orc_model_free(m); return -1; }
        memcpy(&t->L, p, 4); p += 4; int n = t->L - 1;
        if (t->L < 1 || p + (size_t)n * 28 + (size_t)t->L * 12 > end) { t->L = 0; orc_model_free(m); return -1; }
        size_t a = (size_t)(n > 0 ? n : 1);
        t->feat = (int32_t*)malloc(4 * a); t->theta = (int32_t*)malloc(4 * a); t->dleft = (int32_t*)malloc(4 * a);
        t->left = (int32_t*)malloc(4 * a); t->right = (int32_t*)malloc(4 * a); t->gain = (double*)malloc(8 * a);
        t->leaf_value = (double*)malloc(8 * (size_t)t->L); t->leaf_count = (int32_t*)malloc(4 * (size_t)t->L);
        memcpy(t->feat, p, 4 * n); p += 4 * n; memcpy(t->theta, p, 4 * n); p += 4 * n; memcpy(t->dleft, p, 4 * n); p += 4 * n;
        memcpy(t->left, p, 4 * n); p += 4 * n; memcpy(t->right, p, 4 * n); p += 4 * n; memcpy(t->gain, p, 8 * n); p += 8 * n;
        memcpy(t->leaf_value, p, 8 * t->L); p += 8 * t->L; memcpy(t->leaf_count, p, 4 * t->L); p += 4 * t->L;
    }
    *out = m; return 0;
}
ORC_API void orc_model_free(orc_model* m) {
    if (!m) return;
    if (m->feats) { for (int f = 0; f < m->F; ++f) { free(m->feats[f].ub); free(m->feats[f].unseen); } free(m->feats); }
    if (m->trees) {
        for (int i = 0; i < m->n_iter * m->K; ++i) {
            orc_tree* t = &m->trees[i];
            free(t->feat); free(t->theta); free(t->dleft); free(t->left); free(t->right); free(t->gain);
            free(t->leaf_value); free(t->leaf_count);
        }
        free(m->trees);
    }
    free(m);
}
ORC_API int orc_model_info(const orc_model* m, int32_t* info /*[5]: objective,num_class,K,n_iter,F*/) {
    if (!m) return -1;
    info[0] = m->objective; info[1] = m->num_class; info[2] = m->K; info[3] = m->n_iter; info[4] = m->F;
    return 0;
}

/* ------------------------------------------------------------------ chained repair
 * RepairModel._repair's inner UDF (python/repair/model.py:1107-1133): for each target model in
 * order, score EVERY row, then overwrite only the NULL cells so that later models see the repair.
 * table: [C][n] codes, modified in place.  feat_cols: concatenated feature column lists,
 * feat_off[T+1].  class_code[t]: class index -> code written into the target column
 * (class_off[T+1]); for a regression target the entry list is empty and the column is left NULL
 * (numeric write-back is done by the host, which owns the value dictionary).
 * out_label [T][n] (class index or -1), out_prob [T][n] (probability of the arg-max / raw value). */
ORC_API int orc_repair_chain(const orc_model* const* models, int32_t T, const int32_t* target_col,
                             const int32_t* feat_cols, const int32_t* feat_off,
                             const int32_t* class_code, const int32_t* class_off,
                             int32_t* table, int64_t n, int32_t C, int32_t* out_label, double* out_prob) {
    (void)C;
    for (int t = 0; t < T; ++t) {
        const orc_model* m = models[t];
        int F = feat_off[t + 1] - feat_off[t];
        if (F != m->F) return -1;
        int32_t* X = (int32_t*)malloc(sizeof(int32_t) * (size_t)F * (n > 0 ? n : 1));
        for (int f = 0; f < F; ++f) memcpy(X + (size_t)f * n, table + (size_t)feat_cols[feat_off[t] + f] * n, sizeof(int32_t) * n);
        int ncol = (m->objective == 2) ? 1 : m->num_class;
        double* pr = (double*)malloc(sizeof(double) * (size_t)ncol * (n > 0 ? n : 1));
        int rc = orc_predict(m, X, n, F, pr);
        if (rc) { free(X); free(pr); return rc; }
        int32_t* tc = table + (size_t)target_col[t] * n;
        for (int64_t i = 0; i < n; ++i) {
            if (m->objective == 2) { out_label[(size_t)t * n + i] = -1; out_prob[(size_t)t * n + i] = pr[i]; continue; }
            int best = 0;
            for (int k = 1; k < ncol; ++k) if (pr[i * ncol + k] > pr[i * ncol + best]) best = k;
            out_label[(size_t)t * n + i] = best; out_prob[(size_t)t * n + i] = pr[i * ncol + best];
            if (tc[i] < 0 && class_off[t + 1] - class_off[t] > best) tc[i] = class_code[class_off[t] + best];
        }
        free(X); free(pr);
    }
    return 0;
}

ORC_API double orc_exp(double x) { return rg_exp(x); }
ORC_API int orc_find_bin(const int32_t* col, int64_t n, int32_t n_codes, const orc_params* p, int32_t* V, int32_t* has_nan, int32_t* ub /*cap max_bin*/) {
    orc_feat f; find_bin(col, n, n_codes, p, &f, NULL, 0); *V = f.V; *has_nan = f.has_nan;
    memcpy(ub, f.ub, sizeof(int32_t) * f.V); free(f.ub); return 0;
}
