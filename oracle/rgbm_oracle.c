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
 * PARITY STATUS: "parity unpinned" against real LightGBM bits (cannot be run here).  What IS
 * pinned: the reference's own golden labels (bin/testdata/adult_repair.csv, test_model.py
 * inline goldens) -- see tests/test_oracle_golden.py -- and agreement in accuracy with
 * scikit-learn's HistGradientBoosting.  The HIP product is held bit-exact against THIS file.
 *
 * Deliberate, documented deviations from LightGBM 3.3.1 (DESIGN.md "Numerics"):
 *  D1. gradients/hessians are computed in double and quantised to fixed point
 *      (|gq| < 2^20, 0 <= hq < 2^21, power-of-two scales fixed per model) instead of float32;
 *      histogram sums are exact int64 => order independent => bit-reproducible on a GPU.
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
#define GQ_MAX ((1 << 20) - 1)
#define HQ_MAX ((1 << 21) - 1)

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

/* ------------------------------------------------------------------ split search */
typedef struct {
    double gain;           /* relative gain (best_gain - min_gain_shift); -inf if none */
    int32_t feature, theta, default_left;
    int64_t left_gq, left_hq;
    int64_t left_cnt_est;
    double left_out, right_out;
} split_info;

typedef struct {
    double inv_sg, inv_sh;    /* 2^-e_g, 2^-e_h */
    const orc_params* p;
} split_ctx;

static inline int64_t round_int(double x) { return (int64_t)(x + 0.5); }

/* Numerics v1.02 (csrc/rgbm_numerics.h h_from_g): the quantised hessian is a function of the QUANTISED gradient, the row's label
 * and weight -- the level passes of the product carry g only and recompute h.  obj 0: g = response * w, h = |response|(1 - |response|) w;
 * obj 1: g = (p - [y is this class]) w, h = factor p (1 - p) w;  obj 2: h = w. */
#define GQ_MAX_ ((1 << 20) - 1)
#define HQ_MAX_ ((1 << 21) - 1)
static int32_t h_from_g(int32_t gq, int is_label_class, double w, int obj, double inv_sg, double sh, double factor) {
    double h;
    if (obj == 2) h = w;
    else {
        const double inv_w = w > 0.0 ? 1.0 / w : 0.0;          /* the product takes this reciprocal once per label / row */
        const double a = ((double)gq * inv_sg) * inv_w;
        if (obj == 0) { const double r = fabs(a); h = r * (1.0 - r) * w; }
        else { const double p = is_label_class ? a + 1.0 : a; h = factor * p * (1.0 - p) * w; }
        if (!(h > 0.0)) h = 0.0;
    }
    double b = rint(h * sh);
    if (b > HQ_MAX_) b = HQ_MAX_;
    return (int32_t)b;
}

/* FeatureHistogram::FindBestThresholdSequentially, both instantiations used by
 * FuncForNumricalL3 for MissingType::NaN / None, restated over exact integer bin sums.
 * hg/hh: [V + has_nan] bin sums; value bins 0..V-1, NaN bin at V. theta=-1 => only NULLs left. */
static void find_best_threshold(const int64_t* hg, const int64_t* hh, const orc_feat* f, int32_t fidx,
                                int64_t Gq, int64_t Hq, int64_t num_data, const split_ctx* c,
                                split_info* out) {
    const orc_params* p = c->p;
    const int V = f->V;
    const double sum_gradient = (double)Gq * c->inv_sg;
    const double sum_hessian = (double)Hq * c->inv_sh + 2 * kEps;
    const double gain_shift = leaf_gain(sum_gradient, sum_hessian, p->lambda_l1, p->lambda_l2);
    const double min_gain_shift = gain_shift + p->min_gain_to_split;
    const double cnt_factor = (double)num_data / sum_hessian;
    const int two_way = f->has_nan && V >= 1;

    out->gain = -INFINITY; out->feature = fidx; out->theta = 0; out->default_left = 1;
    out->left_gq = out->left_hq = 0; out->left_cnt_est = 0; out->left_out = out->right_out = 0.0;
    int is_splittable = 0;

    /* ---- REVERSE scan (missing -> left) */
    {
        double best_gain = -INFINITY; int best_theta = V; int64_t best_lg = 0, best_lh = 0, best_lc = 0;
        int64_t rg = 0, rh = 0, right_count = 0;
        for (int b = V - 1; b >= 0; --b) {
            rg += hg[b]; rh += hh[b];
            right_count += round_int((double)hh[b] * c->inv_sh * cnt_factor);
            double sum_right_hessian = (double)rh * c->inv_sh + kEps;
            if (right_count < p->min_data_in_leaf || sum_right_hessian < p->min_sum_hessian_in_leaf) continue;
            int64_t left_count = num_data - right_count;
            if (left_count < p->min_data_in_leaf) break;
            int64_t lh = Hq - rh, lg = Gq - rg;
            double sum_left_hessian = (double)lh * c->inv_sh + kEps;
            if (sum_left_hessian < p->min_sum_hessian_in_leaf) break;
            double sum_right_gradient = (double)rg * c->inv_sg;
            double sum_left_gradient = (double)lg * c->inv_sg;
            double cur = leaf_gain(sum_left_gradient, sum_left_hessian, p->lambda_l1, p->lambda_l2) +
                         leaf_gain(sum_right_gradient, sum_right_hessian, p->lambda_l1, p->lambda_l2);
            if (cur <= min_gain_shift) continue;
            is_splittable = 1;
            if (cur > best_gain) { best_gain = cur; best_theta = b - 1; best_lg = lg; best_lh = lh; best_lc = left_count; }
        }
        if (is_splittable && best_gain > out->gain + min_gain_shift) {
            out->theta = best_theta; out->default_left = 1;
            out->left_gq = best_lg; out->left_hq = best_lh; out->left_cnt_est = best_lc;
            double lH = (double)best_lh * c->inv_sh + kEps;
            double rH = (double)(Hq - best_lh) * c->inv_sh + kEps;
            out->left_out = leaf_output((double)best_lg * c->inv_sg, lH, p->lambda_l1, p->lambda_l2);
            out->right_out = leaf_output((double)(Gq - best_lg) * c->inv_sg, rH, p->lambda_l1, p->lambda_l2);
            out->gain = best_gain - min_gain_shift;
        }
    }
    /* ---- FORWARD scan (missing -> right), only with a NaN bin and >2 LightGBM bins */
    if (two_way) {
        double best_gain = -INFINITY; int best_theta = V; int64_t best_lg = 0, best_lh = 0, best_lc = 0;
        int64_t lg = 0, lh = 0, left_count = 0;
        for (int b = 0; b <= V - 1; ++b) {
            lg += hg[b]; lh += hh[b];
            left_count += round_int((double)hh[b] * c->inv_sh * cnt_factor);
            double sum_left_hessian = (double)lh * c->inv_sh + kEps;
            if (left_count < p->min_data_in_leaf || sum_left_hessian < p->min_sum_hessian_in_leaf) continue;
            int64_t right_count = num_data - left_count;
            if (right_count < p->min_data_in_leaf) break;
            int64_t rh = Hq - lh, rg = Gq - lg;
            double sum_right_hessian = (double)rh * c->inv_sh + kEps;
            if (sum_right_hessian < p->min_sum_hessian_in_leaf) break;
            double sum_left_gradient = (double)lg * c->inv_sg;
            double sum_right_gradient = (double)rg * c->inv_sg;
            double cur = leaf_gain(sum_left_gradient, sum_left_hessian, p->lambda_l1, p->lambda_l2) +
                         leaf_gain(sum_right_gradient, sum_right_hessian, p->lambda_l1, p->lambda_l2);
            if (cur <= min_gain_shift) continue;
            is_splittable = 1;
            if (cur > best_gain) { best_gain = cur; best_theta = b; best_lg = lg; best_lh = lh; best_lc = left_count; }
        }
        if (is_splittable && best_gain > out->gain + min_gain_shift) {
            out->theta = best_theta; out->default_left = 0;
            out->left_gq = best_lg; out->left_hq = best_lh; out->left_cnt_est = best_lc;
            double lH = (double)best_lh * c->inv_sh + kEps;
            double rH = (double)(Hq - best_lh) * c->inv_sh + kEps;
            out->left_out = leaf_output((double)best_lg * c->inv_sg, lH, p->lambda_l1, p->lambda_l2);
            out->right_out = leaf_output((double)(Gq - best_lg) * c->inv_sg, rH, p->lambda_l1, p->lambda_l2);
            out->gain = best_gain - min_gain_shift;
        }
    }
}

/* SplitInfo::operator> : larger gain wins, equal gain -> smaller feature index */
static inline int split_better(const split_info* a, const split_info* b) {
    if (a->gain != b->gain) return a->gain > b->gain;
    return a->feature < b->feature;
}

/* ------------------------------------------------------------------ trainer state */
typedef struct {
    int64_t N; int32_t F;
    const orc_params* p;
    orc_feat* feats;
    uint8_t* bins;            /* [F][N] (NaN bin index = V) */
    int32_t* hoff;            /* [F+1] histogram offsets */
    int32_t totbins;
    char* trivial;            /* [F] */
    split_ctx ctx;
} trainer;

/* Threads of the timing harness (bench.py cpu_baseline): LightGBM's col-wise mode builds the per-feature histograms in
 * parallel; so does this (features are independent output ranges and the sums are integers: bit-identical for any thread
 * count).  Default 1: the tests never change it. */
static int g_threads = 1;
ORC_API void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }

static void build_hist(const trainer* t, const int32_t* rows, int64_t n, const int32_t* gq, const int32_t* hq,
                       const char* used, int64_t* hg, int64_t* hh) {
    memset(hg, 0, sizeof(int64_t) * t->totbins);
    memset(hh, 0, sizeof(int64_t) * t->totbins);
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads) if (g_threads > 1 && n > 4096)
    for (int f = 0; f < t->F; ++f) {
        if (!used[f]) continue;
        const uint8_t* b = t->bins + (size_t)f * t->N;
        /* a private copy per feature: neighbouring features' bins share cache lines (false sharing between threads) */
        int64_t lg[256], lh[256];
        memset(lg, 0, sizeof(lg)); memset(lh, 0, sizeof(lh));
        for (int64_t i = 0; i < n; ++i) { int32_t r = rows[i]; lg[b[r]] += gq[r]; lh[b[r]] += hq[r]; }
        const int nb = (f + 1 < t->F ? t->hoff[f + 1] : t->totbins) - t->hoff[f];
        memcpy(hg + t->hoff[f], lg, sizeof(int64_t) * (size_t)nb); memcpy(hh + t->hoff[f], lh, sizeof(int64_t) * (size_t)nb);
    }
}

static void best_split_for_leaf(const trainer* t, const int64_t* hg, const int64_t* hh, const char* used,
                                int64_t Gq, int64_t Hq, int64_t num_data, split_info* best) {
    best->gain = -INFINITY; best->feature = -1;
    for (int f = 0; f < t->F; ++f) {
        if (!used[f]) continue;
        split_info s;
        find_best_threshold(hg + t->hoff[f], hh + t->hoff[f], &t->feats[f], f, Gq, Hq, num_data, &t->ctx, &s);
        if (s.gain == -INFINITY) continue;
        split_info cur = *best; if (cur.feature < 0) cur.feature = INT32_MAX;
        if (split_better(&s, &cur)) *best = s;
    }
}

/* SerialTreeLearner::Train for one tree.  rows: in-bag training row ids (n_in).  Produces the
 * tree (leaf values un-shrunk) and leaf_of_row assignment through the partition arrays. */
static void grow_tree(const trainer* t, int32_t* idx, int64_t n_in, const int32_t* gq, const int32_t* hq,
                      const char* used, orc_tree* tr, int64_t* leaf_begin, int64_t* leaf_cnt) {
    const orc_params* p = t->p;
    const int NL = p->num_leaves;
    int64_t* pool_g = (int64_t*)malloc(sizeof(int64_t) * (size_t)NL * t->totbins);
    int64_t* pool_h = (int64_t*)malloc(sizeof(int64_t) * (size_t)NL * t->totbins);
    split_info* best = (split_info*)malloc(sizeof(split_info) * NL);
    int64_t* LG = (int64_t*)calloc(NL, sizeof(int64_t));
    int64_t* LH = (int64_t*)calloc(NL, sizeof(int64_t));
    int* depth = (int*)calloc(NL, sizeof(int));
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_in > 0 ? n_in : 1));
    int* leaf_parent_node = (int*)malloc(sizeof(int) * NL);   /* node whose child pointer refers to leaf */
    int* leaf_is_left = (int*)malloc(sizeof(int) * NL);

    tr->L = 1;
    tr->feat = (int32_t*)calloc(NL, sizeof(int32_t)); tr->theta = (int32_t*)calloc(NL, sizeof(int32_t));
    tr->dleft = (int32_t*)calloc(NL, sizeof(int32_t)); tr->left = (int32_t*)calloc(NL, sizeof(int32_t));
    tr->right = (int32_t*)calloc(NL, sizeof(int32_t)); tr->gain = (double*)calloc(NL, sizeof(double));
    tr->leaf_value = (double*)calloc(NL, sizeof(double)); tr->leaf_count = (int32_t*)calloc(NL, sizeof(int32_t));

    leaf_begin[0] = 0; leaf_cnt[0] = n_in; depth[0] = 0; leaf_parent_node[0] = -1; leaf_is_left[0] = 0;
    for (int l = 1; l < NL; ++l) { leaf_begin[l] = 0; leaf_cnt[l] = 0; }
    for (int l = 0; l < NL; ++l) { best[l].gain = -INFINITY; best[l].feature = -1; }
    for (int64_t i = 0; i < n_in; ++i) { LG[0] += gq[idx[i]]; LH[0] += hq[idx[i]]; }
    tr->leaf_count[0] = (int32_t)n_in;

    build_hist(t, idx, n_in, gq, hq, used, pool_g, pool_h);
    /* BeforeFindBestSplit at the root: depth ok; count check */
    if (!(n_in < (int64_t)p->min_data_in_leaf * 2))
        best_split_for_leaf(t, pool_g, pool_h, used, LG[0], LH[0], n_in, &best[0]);

    for (int s = 0; s < NL - 1; ++s) {
        /* ArrayArgs<SplitInfo>::ArgMax uses SplitInfo::operator> (gain, then smaller feature) */
        int bl = 0;
        for (int l = 1; l < tr->L; ++l) {
            split_info a = best[l], b = best[bl];
            if (a.feature < 0) a.feature = INT32_MAX;
            if (b.feature < 0) b.feature = INT32_MAX;
            if (split_better(&a, &b)) bl = l;
        }
        const split_info sp = best[bl];
        if (!(sp.gain > 0.0)) break;
        /* Tree::Split */
        int node = tr->L - 1, right_leaf = tr->L;
        tr->feat[node] = sp.feature; tr->theta[node] = sp.theta; tr->dleft[node] = sp.default_left;
        tr->gain[node] = sp.gain;
        tr->left[node] = ~bl; tr->right[node] = ~right_leaf;
        if (leaf_parent_node[bl] >= 0) {
            if (leaf_is_left[bl]) tr->left[leaf_parent_node[bl]] = node; else tr->right[leaf_parent_node[bl]] = node;
        }
        leaf_parent_node[bl] = node; leaf_is_left[bl] = 1;
        leaf_parent_node[right_leaf] = node; leaf_is_left[right_leaf] = 0;
        tr->leaf_value[bl] = sp.left_out; tr->leaf_value[right_leaf] = sp.right_out;
        /* DataPartition::Split (stable) */
        const orc_feat* f = &t->feats[sp.feature];
        const uint8_t* b = t->bins + (size_t)sp.feature * t->N;
        int64_t beg = leaf_begin[bl], cnt = leaf_cnt[bl], nl = 0, nr = 0;
        for (int64_t i = 0; i < cnt; ++i) {
            int32_t r = idx[beg + i]; int bin = b[r];
            int go_left = (f->has_nan && bin == f->V) ? sp.default_left : (bin <= sp.theta);
            if (go_left) idx[beg + nl++] = r; else tmp[nr++] = r;
        }
        memcpy(idx + beg + nl, tmp, sizeof(int32_t) * (size_t)nr);
        leaf_cnt[bl] = nl; leaf_begin[right_leaf] = beg + nl; leaf_cnt[right_leaf] = nr;
        tr->leaf_count[bl] = (int32_t)nl; tr->leaf_count[right_leaf] = (int32_t)nr;
        int64_t pG = LG[bl], pH = LH[bl];
        LG[bl] = sp.left_gq; LH[bl] = sp.left_hq; LG[right_leaf] = pG - sp.left_gq; LH[right_leaf] = pH - sp.left_hq;
        depth[right_leaf] = depth[bl] = depth[bl] + 1;
        tr->L += 1;
        best[bl].gain = -INFINITY; best[bl].feature = -1; best[right_leaf].gain = -INFINITY; best[right_leaf].feature = -1;
        if (tr->L >= NL) break;   /* no further split will be taken */
        /* BeforeFindBestSplit */
        if (p->max_depth > 0 && depth[bl] >= p->max_depth) continue;
        if (nr < (int64_t)p->min_data_in_leaf * 2 && nl < (int64_t)p->min_data_in_leaf * 2) continue;
        /* histograms: smaller child built, larger = parent - smaller (parent lives in slot bl) */
        int smaller = (nl < nr) ? bl : right_leaf, larger = (nl < nr) ? right_leaf : bl;
        int64_t* par_g = pool_g + (size_t)bl * t->totbins; int64_t* par_h = pool_h + (size_t)bl * t->totbins;
        int64_t* rg_ = pool_g + (size_t)right_leaf * t->totbins; int64_t* rh_ = pool_h + (size_t)right_leaf * t->totbins;
        if (smaller == right_leaf) {
            build_hist(t, idx + leaf_begin[right_leaf], nr, gq, hq, used, rg_, rh_);
            for (int i = 0; i < t->totbins; ++i) { par_g[i] -= rg_[i]; par_h[i] -= rh_[i]; }
        } else {
            /* build left into the right slot temporarily, then swap roles */
            build_hist(t, idx + leaf_begin[bl], nl, gq, hq, used, rg_, rh_);
            for (int i = 0; i < t->totbins; ++i) {
                int64_t lg = rg_[i], lh = rh_[i];
                rg_[i] = par_g[i] - lg; rh_[i] = par_h[i] - lh; par_g[i] = lg; par_h[i] = lh;
            }
        }
        (void)larger;
        best_split_for_leaf(t, par_g, par_h, used, LG[bl], LH[bl], nl, &best[bl]);
        best_split_for_leaf(t, rg_, rh_, used, LG[right_leaf], LH[right_leaf], nr, &best[right_leaf]);
    }
    free(pool_g); free(pool_h); free(best); free(LG); free(LH); free(depth); free(tmp);
    free(leaf_parent_node); free(leaf_is_left);
}

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

static int frexp_exp(double v) { int ex; (void)frexp(v, &ex); return ex; }   /* v < 2^ex strictly, v >= 2^(ex-1) */

ORC_API void orc_model_free(orc_model* m);

/* ------------------------------------------------------------------ GBDT::Train */
ORC_API int orc_train2(const int32_t* X, int64_t N, int32_t F, const int32_t* n_codes,
                       const int32_t* y_code, int32_t n_y_codes, const double* y_value,
                       const double* class_weight, const double* sample_weight,
                       const orc_params* p, const double* const* feat_values, const int32_t* feat_kinds, orc_model** out);

ORC_API int orc_train(const int32_t* X, int64_t N, int32_t F, const int32_t* n_codes,
                      const int32_t* y_code, int32_t n_y_codes, const double* y_value,
                      const double* class_weight, const double* sample_weight,
                      const orc_params* p, orc_model** out) {
    return orc_train2(X, N, F, n_codes, y_code, n_y_codes, y_value, class_weight, sample_weight, p, NULL, NULL, out);
}

/* feat_values[f] (may be NULL, as may the whole array): the ascending distinct values behind the codes of NUMERIC feature f
 * (rgbm_table_set_column_values on the product side) -- bin bounds are then value midpoints (mid_code).
 * feat_kinds[f] == 1 (array may be NULL): CATEGORICAL feature (rgbm_table_set_column_kind) -- codes no training row holds are
 * recorded in the model and are MISSING at prediction time. */
ORC_API int orc_train2(const int32_t* X, int64_t N, int32_t F, const int32_t* n_codes,
                       const int32_t* y_code, int32_t n_y_codes, const double* y_value,
                       const double* class_weight, const double* sample_weight,
                       const orc_params* p, const double* const* feat_values, const int32_t* feat_kinds, orc_model** out) {
    if (N <= 0 || F <= 0 || !X || !y_code || !p || !out) return -1;
    if (p->max_bin < 2 || p->max_bin > 255 || p->num_leaves < 2) return -2;
    const int obj = p->objective;
    const int K = (obj == 1) ? p->num_class : 1;
    if (obj == 1 && (p->num_class < 2 || n_y_codes > p->num_class)) return -3;
    if (obj == 0 && n_y_codes > 2) return -3;
    if (obj == 2 && !y_value) return -3;

    trainer t; memset(&t, 0, sizeof(t));
    t.N = N; t.F = F; t.p = p;
    t.feats = (orc_feat*)calloc(F, sizeof(orc_feat));
    t.bins = (uint8_t*)malloc((size_t)F * N);
    t.hoff = (int32_t*)malloc(sizeof(int32_t) * (F + 1));
    t.trivial = (char*)calloc(F, 1);
    t.hoff[0] = 0;
    for (int f = 0; f < F; ++f) {
        const int32_t* col = X + (size_t)f * N;
        find_bin(col, N, n_codes[f], p, &t.feats[f], feat_values ? feat_values[f] : NULL, feat_kinds ? feat_kinds[f] == 1 : 0);
        const orc_feat* ft = &t.feats[f];
        uint8_t* b = t.bins + (size_t)f * N;
        for (int64_t i = 0; i < N; ++i) { int bin = code_to_bin(ft, col[i]); b[i] = (uint8_t)(bin < 0 ? ft->V : bin); }
        t.hoff[f + 1] = t.hoff[f] + ft->V + ft->has_nan;
        t.trivial[f] = (ft->V + ft->has_nan <= 1) || ft->V == 0;
    }
    t.totbins = t.hoff[F] > 0 ? t.hoff[F] : 1;

    /* per-row weights, class totals */
    double* w = NULL;
    if (class_weight || sample_weight) {
        w = (double*)malloc(sizeof(double) * N);
        for (int64_t i = 0; i < N; ++i) {
            double v = class_weight ? class_weight[y_code[i]] : 1.0;
            if (sample_weight) v = v * sample_weight[i];
            w[i] = (double)(float)v;   /* LightGBM Metadata keeps weights as float32 (label_t) */
        }
    }
    double w_max = 0.0;
    if (w) { for (int64_t i = 0; i < N; ++i) if (w[i] > w_max) w_max = w[i]; } else w_max = 1.0;
    if (!(w_max > 0.0)) w_max = 1.0;

    /* LightGBM keeps labels as float32 (label_t): regression targets are rounded once here */
    double* yv32 = NULL;
    if (obj == 2) {
        yv32 = (double*)malloc(sizeof(double) * (n_y_codes > 0 ? n_y_codes : 1));
        for (int c = 0; c < n_y_codes; ++c) yv32[c] = (double)(float)y_value[c];
        y_value = yv32;
    }
    /* BoostFromScore.  Label totals are defined order-free: without per-row sample weights the
     * weight of label c is cnt[c] * class_weight[c]; sums run over labels in ascending order.
     * (With sample weights they are row-order sums -- host-array path only.) */
    double* init = (double*)calloc(K, sizeof(double));
    double ymin = 0.0, ymax = 0.0;
    {
        int nl = n_y_codes > 0 ? n_y_codes : 1;
        if (obj == 0 && nl < 2) nl = 2;
        if (obj == 1 && nl < K) nl = K;
        int64_t* cnt = (int64_t*)calloc(nl, sizeof(int64_t));
        double* tot = (double*)calloc(nl, sizeof(double));
        for (int64_t i = 0; i < N; ++i) ++cnt[y_code[i]];
        if (sample_weight) { for (int64_t i = 0; i < N; ++i) tot[y_code[i]] += w[i]; }
        else { for (int c = 0; c < nl; ++c) tot[c] = (double)cnt[c] * (class_weight ? (double)(float)class_weight[c] : 1.0); }
        double sumw = 0.0;
        for (int c = 0; c < nl; ++c) sumw += tot[c];
        if (obj == 2) {
            double suml = 0.0; int first = 1;
            for (int c = 0; c < nl; ++c) {
                if (cnt[c] == 0) continue;
                if (first) { ymin = ymax = y_value[c]; first = 0; }
                if (y_value[c] < ymin) ymin = y_value[c];
                if (y_value[c] > ymax) ymax = y_value[c];
            }
            if (sample_weight) { for (int64_t i = 0; i < N; ++i) suml += y_value[y_code[i]] * w[i]; }
            else { for (int c = 0; c < nl; ++c) suml += tot[c] * y_value[c]; }
            init[0] = suml / sumw;
        } else if (obj == 0) {
            double pavg = tot[1] / sumw;
            if (pavg > 1.0 - kEps) pavg = 1.0 - kEps;
            if (pavg < kEps) pavg = kEps;
            init[0] = log(pavg / (1.0 - pavg));
        } else {
            for (int k = 0; k < K; ++k) { double pr = tot[k] / sumw; init[k] = log(pr > kEps ? pr : kEps); }
        }
        free(cnt); free(tot);
    }

    /* quantisation scales (D1) */
    double bound_g, bound_h;
    const double factor = (obj == 1) ? (double)K / (double)(K - 1) : 1.0;
    if (obj == 2) { bound_g = (ymax - ymin) * w_max; if (!(bound_g > 0.0)) bound_g = 1.0; bound_h = w_max; }
    else if (obj == 0) { bound_g = w_max; bound_h = 0.25 * w_max; }
    else { bound_g = w_max; bound_h = factor * 0.25 * w_max; }
    /* hessians reach their bound exactly (regression: h = w, so the heaviest rows have h = bound): the scale is the largest
     * power of two with bound * 2^e_h STRICTLY below 2^21, i.e. 21 - (frexp exponent), one less than for the gradients
     * when the bound is an exact power of two -- otherwise h = 1 would be clamped to 2^21 - 1 (a 2^-21 bias in every leaf). */
    const int e_g = 20 - ceil_log2(bound_g), e_h = 21 - frexp_exp(bound_h);
    const double sg = pow2(e_g), sh = pow2(e_h);
    t.ctx.inv_sg = pow2(-e_g); t.ctx.inv_sh = pow2(-e_h); t.ctx.p = p;

    orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
    m->objective = obj; m->num_class = (obj == 1) ? K : (obj == 0 ? 2 : 1); m->K = K; m->F = F;
    m->feats = t.feats;
    m->trees = (orc_tree*)calloc((size_t)p->n_estimators * K, sizeof(orc_tree));

    double* score = (double*)malloc(sizeof(double) * (size_t)K * N);
    for (int k = 0; k < K; ++k) for (int64_t i = 0; i < N; ++i) score[(size_t)k * N + i] = init[k];
    int32_t* gq = (int32_t*)malloc(sizeof(int32_t) * (size_t)K * N);
    int32_t* hq = (int32_t*)malloc(sizeof(int32_t) * (size_t)K * N);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * N);
    int32_t* bag = (int32_t*)malloc(sizeof(int32_t) * N);
    int64_t bag_cnt = N;
    for (int64_t i = 0; i < N; ++i) bag[i] = (int32_t)i;
    int64_t* leaf_begin = (int64_t*)malloc(sizeof(int64_t) * p->num_leaves);
    int64_t* leaf_cnt = (int64_t*)malloc(sizeof(int64_t) * p->num_leaves);
    char* used = (char*)malloc(F);
    double* rec = (double*)malloc(sizeof(double) * K);
    const uint8_t** bcols = (const uint8_t**)malloc(sizeof(uint8_t*) * F);
    for (int f = 0; f < F; ++f) bcols[f] = t.bins + (size_t)f * N;
    char* in_bag = (char*)malloc(N);

    /* Config seeds derived from `seed` (config.cpp) */
    lgb_rand sr; sr.x = (uint32_t)p->seed;
    int data_random_seed = rnd16(&sr); (void)data_random_seed;
    int bagging_seed = rnd16(&sr);
    int drop_seed = rnd16(&sr); (void)drop_seed;
    int feature_fraction_seed = rnd16(&sr);
    lgb_rand ff_rand; ff_rand.x = (uint32_t)feature_fraction_seed;
    const int use_bagging = p->bagging_freq > 0 && p->bagging_fraction < 1.0;
    int64_t n_blocks = (N + 1023) / 1024;
    lgb_rand* bag_rands = NULL;
    if (use_bagging) {
        bag_rands = (lgb_rand*)malloc(sizeof(lgb_rand) * n_blocks);
        for (int64_t b = 0; b < n_blocks; ++b) bag_rands[b].x = (uint32_t)(bagging_seed + b);
    }
    int n_valid = 0; int* valid = (int*)malloc(sizeof(int) * F); int* samp = (int*)malloc(sizeof(int) * F);
    for (int f = 0; f < F; ++f) if (!t.trivial[f]) valid[n_valid++] = f;

    int n_iter = 0;
    for (int it = 0; it < p->n_estimators; ++it) {
        /* Bagging (gbdt.cpp BaggingHelper): every bagging_freq iterations */
        if (use_bagging && it % p->bagging_freq == 0) {
            int64_t l = 0, r = N;
            for (int64_t i = 0; i < N; ++i) {
                if (rnd_float(&bag_rands[i / 1024]) < p->bagging_fraction) bag[l++] = (int32_t)i; else bag[--r] = (int32_t)i;
            }
            bag_cnt = l;
        }
        /* Boosting(): gradients of all classes from the scores at iteration start */
#pragma omp parallel num_threads(g_threads) if (g_threads > 1 && N > 4096)
        {
        double* rec_t = (double*)malloc(sizeof(double) * (K > 0 ? K : 1));   /* per-thread softmax scratch */
#pragma omp for schedule(static)
        for (int64_t i = 0; i < N; ++i) {
            double* rec = rec_t;
            double wi = w ? w[i] : 1.0;
            if (obj == 0) {
                double label = (y_code[i] > 0) ? 1.0 : -1.0;
                double response = -label / (1.0 + rg_exp(label * score[i]));
                double abs_r = fabs(response);
                (void)abs_r;
                double g = response * wi;
                double a = rint(g * sg);
                if (a > GQ_MAX) a = GQ_MAX;
                if (a < -GQ_MAX) a = -GQ_MAX;
                gq[i] = (int32_t)a; hq[i] = h_from_g((int32_t)a, 0, wi, 0, t.ctx.inv_sg, sh, factor);
            } else if (obj == 1) {
                double wmax = score[i];
                for (int k = 1; k < K; ++k) { double s = score[(size_t)k * N + i]; if (s > wmax) wmax = s; }
                double wsum = 0.0;
                for (int k = 0; k < K; ++k) { rec[k] = rg_exp(score[(size_t)k * N + i] - wmax); wsum += rec[k]; }
                for (int k = 0; k < K; ++k) {
                    double pk = rec[k] / wsum;
                    double g = ((y_code[i] == k) ? (pk - 1.0) : pk) * wi;
                    double a = rint(g * sg);
                    if (a > GQ_MAX) a = GQ_MAX;
                    if (a < -GQ_MAX) a = -GQ_MAX;
                    gq[(size_t)k * N + i] = (int32_t)a;
                    hq[(size_t)k * N + i] = h_from_g((int32_t)a, y_code[i] == k, wi, 1, t.ctx.inv_sg, sh, factor);
                }
            } else {
                double g = (score[i] - y_value[y_code[i]]) * wi, h = wi;
                double a = rint(g * sg), b = rint(h * sh);
                if (a > GQ_MAX) a = GQ_MAX;
                if (a < -GQ_MAX) a = -GQ_MAX;
                if (b > HQ_MAX) b = HQ_MAX;
                gq[i] = (int32_t)a; hq[i] = (int32_t)b;
            }
        }
        free(rec_t);
        }
        int should_continue = 0;
        for (int k = 0; k < K; ++k) {
            orc_tree* tr = &m->trees[(size_t)it * K + k];
            /* ColSampler::ResetByTree */
            memset(used, 0, F);
            if (p->feature_fraction < 1.0) {
                int cnt = (int)floor((double)n_valid * p->feature_fraction + 0.5);
                if (cnt < 1) cnt = 1;
                int ns = rnd_sample(&ff_rand, n_valid, cnt, samp);
                for (int i = 0; i < ns; ++i) used[valid[samp[i]]] = 1;
            } else {
                for (int i = 0; i < n_valid; ++i) used[valid[i]] = 1;
            }
            memcpy(idx, bag, sizeof(int32_t) * (size_t)bag_cnt);
            grow_tree(&t, idx, bag_cnt, gq + (size_t)k * N, hq + (size_t)k * N, used, tr, leaf_begin, leaf_cnt);
            if (tr->L > 1) {
                should_continue = 1;
                /* Shrinkage + UpdateScore (in-bag by partition, out-of-bag by traversal) */
                for (int l = 0; l < tr->L; ++l) tr->leaf_value[l] = tr->leaf_value[l] * p->learning_rate;
                double* sc = score + (size_t)k * N;
                if (bag_cnt == N) {
                    for (int l = 0; l < tr->L; ++l)
                        for (int64_t i = 0; i < leaf_cnt[l]; ++i) sc[idx[leaf_begin[l] + i]] += tr->leaf_value[l];
                } else {
                    memset(in_bag, 0, N);
                    for (int l = 0; l < tr->L; ++l)
                        for (int64_t i = 0; i < leaf_cnt[l]; ++i) { int32_t r = idx[leaf_begin[l] + i]; sc[r] += tr->leaf_value[l]; in_bag[r] = 1; }
                    for (int64_t r = 0; r < N; ++r) if (!in_bag[r]) sc[r] += tr->leaf_value[tree_leaf_for(tr, t.feats, bcols, r)];
                }
                if (it == 0 && fabs(init[k]) > kEps)      /* AddBias: the model carries the init score */
                    for (int l = 0; l < tr->L; ++l) tr->leaf_value[l] += init[k];
            } else {
                /* constant tree: carries the init score in the first iteration only */
                tr->leaf_value[0] = (it == 0) ? init[k] : 0.0;
                tr->leaf_count[0] = (int32_t)bag_cnt;
            }
        }
        n_iter = it + 1;
        if (!should_continue) {
            /* "Stopped training because there are no more leaves that meet the split requirements":
             * the trees of this iteration are dropped unless they are the only ones. */
            if (it > 0) n_iter = it;
            break;
        }
    }
    m->n_iter = n_iter;
    *out = m;
    free(t.bins); free(t.hoff); free(t.trivial); free(w); free(init); free(score); free(gq); free(hq);
    free(idx); free(bag); free(leaf_begin); free(leaf_cnt); free(used); free(rec); free(bcols); free(in_bag);
    free(bag_rands); free(valid); free(samp); free(yv32);
    return 0;
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
