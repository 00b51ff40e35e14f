// rgbm_small.h -- the BATCHED small-table trainer: many independent fits (the folds x trials of a hyper-parameter search batch,
// python/repair/train.py:158-209, or the <= 10 000-row models of a reference-default job, python/repair/model.py:755-766) advance
// through their boosting iterations TOGETHER, three launches per iteration for the whole batch:
//
//   k_small_grad   (fit, row tile)      ObjectiveFunction::GetGradients of every fit
//   k_small_tree   (fit, class tree)    ONE workgroup grows ONE class tree of ONE fit, start to finish: the leaf-wise loop of
//                                       rgbm_kernels.h -- histogram of the smaller child, subtraction + threshold scans, best-leaf
//                                       pick, Tree::Split bookkeeping, row partition, shrinkage and AddScore -- with barriers where the
//                                       single-fit path has kernel boundaries
//   (+ k_small_bagging / k_small_bag_lists / k_small_oob for the fits that bag rows)
//
// Why: on ~10^4 rows no grower of this library is bound by data.  A single fit is a chain of ~35 dependent 5-40 us kernels per
// iteration (level grower) or 5 kernels per split (leaf-wise grower); a 48-fit search ran 48 such chains, 1.9-2.5 s whatever the
// number of host threads (profiles/r03af_hp_search_48_fits.log).  One workgroup per (fit, class tree) was built for a SINGLE fit in
// round 2 and was slower there (K workgroups on 256 CUs, ~24 us per split); with all fits of a batch in one grid the same kernel fills
// the chip -- hundreds of class trees at once -- and an iteration of the whole batch costs what one class tree costs.
//
// The arithmetic IS the leaf-wise grower's: the same device functions (scan_child, split_find_body, reduce_leaf_best, tree_step_pick,
// finish_split_body, grad_rows), the same exact integer histograms (numerics v2.2: the grid of a class tree is measured at the start of
// k_small_tree, from the same coarse sums the big-table trainer takes out of its gradient kernels), so every model is the one
// rgbm_table_train returns for that fit, bit for bit (tests/test_gpu_batch.py).
//
// Per-fit constants are read through a descriptor (SmallFit) instead of kernel arguments; every fit has its own bin records, because
// LightGBM bins a fit on ITS training rows (min_data_in_bin can merge a rare value in one fold and not in another).
// State of a class tree: leaves, per-feature candidates and the control block in LDS; row-index lists (ping-pong, as k_partition) and
// the per-leaf histogram pool in global memory (L2-resident at this size).
#pragma once
#include "rgbm_kernels.h"

namespace rg {

#ifndef SM_THREADS_N
#define SM_THREADS_N 512
#endif
constexpr int SM_THREADS = SM_THREADS_N;
constexpr int SM_WAVES = SM_THREADS / 64;
constexpr int SM_MAX_LEAVES = 256;
constexpr int SM_MAX_FEATS = 255;

// Packed threshold scan: on a small table one split is a chain of latencies, and the two longest links were (measured, -DSM_PROF:
// profiles/r04m_*) the flush of the LDS histogram feature by feature (33-38 % of the kernel) and FindBestThreshold with one wave per
// feature, two children one after the other (35-43 %).  A feature of the synthetic tables has 3-65 bins, so most of a wave's 256 bin
// slots sat idle.  Here the features of a child share waves: a lane owns 4 consecutive bins of ONE feature (ScanLane), the prefix
// sums and the arg-max become SEGMENTED wave scans, and both children are scanned at once by different waves.  Every candidate is
// evaluated by the same expressions, in the same order, as scan_child; a feature's winner is picked by the same total order
// (scan_better), so the Cand of every (child, feature) is the one scan_child writes -- bit for bit.
struct ScanLane { int32_t feat /* -1: idle lane */, bin0 /* first of the lane's 4 bins inside the feature */, seg_first /* 1: first lane of its feature */, seg_last /* last lane of its feature */; };

__device__ __forceinline__ long long seg_incl_scan(long long v, bool first) {
    const int lane = lane_id();
    int f = first ? 1 : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long o = __shfl_up(v, off); const int fo = __shfl_up(f, off);
        if (lane >= off && !f) { v += o; f = fo; }
    }
    return v;
}

template <bool REVERSE>
__device__ __forceinline__ ScanBest seg_best_scan(ScanBest v, bool first) {
    const int lane = lane_id();
    int f = first ? 1 : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double g2 = __shfl_up(v.gain, off); const int t2 = __shfl_up(v.theta, off);
        const long long a2 = __shfl_up(v.lg, off), b2 = __shfl_up(v.lh, off); const int fo = __shfl_up(f, off);
        if (lane >= off && !f) {
            if (scan_better<REVERSE>(g2, t2, v.gain, v.theta)) { v.gain = g2; v.theta = t2; v.lg = a2; v.lh = b2; }
            f = fo;
        }
    }
    return v;   // the LAST lane of a segment holds the segment's best
}

// one wave: the features of its lane map, for the child whose compact histogram is `hist` (indexed by FeatMeta::hoff + bin; LDS)
__device__ void scan_child_packed(const HistBin* hist, const FeatMeta* __restrict__ fmeta, const ScanLane sl, long long Gq, long long Hq, long long num_data,
                                  const TrainConst& c, const uint8_t* __restrict__ used_k, Cand* ck /* [F] of this child */) {
    const int lane = lane_id();
    const bool on = sl.feat >= 0;
    FeatMeta fm; fm.V = 0; fm.has_nan = 0; fm.nbins = 0; fm.hoff = 0; fm.fast_base = 0; fm.rep_shift = 0; fm.wide_off = 0; fm.pad = 0;
    if (on) fm = fmeta[sl.feat];
    const int V = fm.V;
    const double keps = k_eps();
    const double sum_gradient = (double)Gq * c.inv_sg;
    const double sum_hessian = (double)Hq * c.inv_sh + 2 * keps;
    const double gain_shift = leaf_gain(sum_gradient, sum_hessian, c.l1, c.l2);
    const double min_gain_shift = gain_shift + c.min_gain_to_split;
    const double cnt_factor = (double)num_data / sum_hessian;
    const bool two_way = fm.has_nan && V >= 1;
    long long lg[4], lh[4], lc[4];
    long long tg = 0, th = 0, tc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = sl.bin0 + j;
        const bool val = on && b < V;
        HistBin hb; hb.g = 0; hb.h = 0;
        if (val) hb = hist[fm.hoff + b];
        lg[j] = hb.g; lh[j] = hb.h;
        lc[j] = val ? round_int((double)hb.h * c.inv_sh * cnt_factor) : 0;
        tg += lg[j]; th += lh[j]; tc += lc[j];
    }
    const bool first = !on || sl.seg_first != 0;
    const long long ig = seg_incl_scan(tg, first), ih = seg_incl_scan(th, first), ic = seg_incl_scan(tc, first);
    const int last = on ? sl.seg_last : lane;
    const long long TG = __shfl(ig, last), TH = __shfl(ih, last), TC = __shfl(ic, last);     // totals over the VALUE bins of the lane's feature
    long long pg = ig - tg, ph = ih - th, pc = ic - tc;   // exclusive prefix at the lane's first bin
    ScanBest rv = {-INFINITY, 0, 0, 0}, fw = {-INFINITY, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = sl.bin0 + j;
        if (on && b < V) {
            {   // REVERSE candidate theta = b-1: right = value bins >= b (NaN rows stay left)
                const long long rg = TG - pg, rh = TH - ph, right_count = TC - pc;
                const double sum_right_hessian = (double)rh * c.inv_sh + keps;
                const long long left_count = num_data - right_count;
                const long long lhq = Hq - rh, lgq = Gq - rg;
                const double sum_left_hessian = (double)lhq * c.inv_sh + keps;
                const bool ok = !(right_count < c.min_data_in_leaf || sum_right_hessian < c.min_sum_hessian) &&
                                !(left_count < c.min_data_in_leaf) && !(sum_left_hessian < c.min_sum_hessian);
                if (ok) {
                    const double cur = leaf_gain((double)lgq * c.inv_sg, sum_left_hessian, c.l1, c.l2) +
                                       leaf_gain((double)rg * c.inv_sg, sum_right_hessian, c.l1, c.l2);
                    if (cur > min_gain_shift && scan_better<true>(cur, b - 1, rv.gain, rv.theta)) { rv.gain = cur; rv.theta = b - 1; rv.lg = lgq; rv.lh = lhq; }
                }
            }
            pg += lg[j]; ph += lh[j]; pc += lc[j];   // now inclusive through b
            if (two_way) {   // FORWARD candidate theta = b: left = value bins <= b (NaN rows go right)
                const long long lgq = pg, lhq = ph, left_count = pc;
                const double sum_left_hessian = (double)lhq * c.inv_sh + keps;
                const long long right_count = num_data - left_count;
                const long long rh = Hq - lhq, rg = Gq - lgq;
                const double sum_right_hessian = (double)rh * c.inv_sh + keps;
                const bool ok = !(left_count < c.min_data_in_leaf || sum_left_hessian < c.min_sum_hessian) &&
                                !(right_count < c.min_data_in_leaf) && !(sum_right_hessian < c.min_sum_hessian);
                if (ok) {
                    const double cur = leaf_gain((double)lgq * c.inv_sg, sum_left_hessian, c.l1, c.l2) +
                                       leaf_gain((double)rg * c.inv_sg, sum_right_hessian, c.l1, c.l2);
                    if (cur > min_gain_shift && scan_better<false>(cur, b, fw.gain, fw.theta)) { fw.gain = cur; fw.theta = b; fw.lg = lgq; fw.lh = lhq; }
                }
            }
        }
    }
    rv = seg_best_scan<true>(rv, first);
    fw = seg_best_scan<false>(fw, first);
    if (on && lane == sl.seg_last) {     // the same finish as scan_child's lane 0
        Cand o; o.gain = -INFINITY; o.theta = 0; o.dleft = 1; o.left_gq = 0; o.left_hq = 0; o.left_out = 0.0; o.right_out = 0.0;
        if (used_k[sl.feat]) {
            if (rv.gain > -INFINITY && rv.gain > o.gain + min_gain_shift) {
                o.theta = rv.theta; o.dleft = 1; o.left_gq = rv.lg; o.left_hq = rv.lh;
                const double lH = (double)rv.lh * c.inv_sh + keps, rH = (double)(Hq - rv.lh) * c.inv_sh + keps;
                o.left_out = leaf_output((double)rv.lg * c.inv_sg, lH, c.l1, c.l2);
                o.right_out = leaf_output((double)(Gq - rv.lg) * c.inv_sg, rH, c.l1, c.l2);
                o.gain = rv.gain - min_gain_shift;
            }
            if (fw.gain > -INFINITY && fw.gain > o.gain + min_gain_shift) {
                o.theta = fw.theta; o.dleft = 0; o.left_gq = fw.lg; o.left_hq = fw.lh;
                const double lH = (double)fw.lh * c.inv_sh + keps, rH = (double)(Hq - fw.lh) * c.inv_sh + keps;
                o.left_out = leaf_output((double)fw.lg * c.inv_sg, lH, c.l1, c.l2);
                o.right_out = leaf_output((double)(Gq - fw.lg) * c.inv_sg, rH, c.l1, c.l2);
                o.gain = fw.gain - min_gain_shift;
            }
        }
        ck[sl.feat] = o;
    }
}

struct SmallFit {   // one fit of a batch; lives in device memory, indexed by fit id
    TrainConst c;
    const uint4* rec;            // [nchunk][N] bin records of THIS fit
    const int32_t* ycol;         // [N] target codes (a column of the fit's table)
    float2* gh;                  // [K][N]
    double* score;               // [K][N]
    int32_t* idx0; int32_t* idx1;      // [K][n_train] ping-pong row lists
    int32_t* base_idx;           // [n_train] training rows (bagging: the rows in the bag)
    HistBin* pool;               // [K][num_leaves][totbins]
    const FeatMeta* fmeta; const ChunkMeta* cmeta;
    const uint8_t* used;         // [n_estimators][K][F] per-tree feature masks
    TreeOut out;
    const double* init; double* upd /* [K][num_leaves] */; int32_t* tree_L /* [K] leaves of the tree just grown */; int32_t* any_split /* [n_estimators] */;
    const double* class_w; const double* y_value;
    // bagging (bag_freq == 0: none)
    unsigned int* rand_state; const int32_t* sorted_rows; uint8_t* inbag; unsigned int* bagcnt /* [2] in bag, out of bag */; int32_t* oob;
    double bag_fraction; long long bag_nrb;
    int32_t bag_freq, tree0 /* first class tree of this fit in the batch's tree numbering */, n_estimators, pad;
    unsigned long long lds_hist;
    // validation rows scored while the fit trains (cross_val_score without a predictor: every new tree is added to their scores)
    const uint4* vrec;           // [nchunk][n_valid] bin records in the PREDICTOR's convention (NULL / unseen category = 255)
    double* vscore;              // [K][n_valid] raw scores, start at the initial scores
    long long n_valid;
    const ScanLane* scan_map;    // [scan_waves][64] lane -> (feature, first bin) of the packed threshold scan; scan_waves = 0: one wave per feature
    int32_t scan_waves, pad2;
    unsigned long long* prof;    // -DSM_PROF builds: [8] cycles per phase of k_small_tree, summed over this fit's class trees (else unused)
};

// the histogram area doubles as the leaf-delta table of the final score update: at least num_leaves doubles
__host__ __device__ inline size_t sm_hist_area(size_t lds_hist, int num_leaves) {
    const size_t a = lds_hist > (size_t)num_leaves * 8 ? lds_hist : (size_t)num_leaves * 8;
    return (a + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t sm_lds_bytes(size_t lds_hist, int num_leaves, int F, int packed_totbins = 0) {
    size_t b = sm_hist_area(lds_hist, num_leaves);
    b += (size_t)num_leaves * sizeof(Leaf);
    b = (b + 15) & ~(size_t)15;
    b += (size_t)2 * F * sizeof(Cand);
    b = (b + 15) & ~(size_t)15;
    b += (size_t)2 * packed_totbins * sizeof(HistBin);     // packed scan: compact histograms of the two children just built / derived
    return b + 64;
}

// A fit is FROZEN from the first boosting iteration in which no class tree could split: the model ends there (model_trees_begin: n_iter =
// the first iteration with any_split == 0; GBDT::TrainOneIter returns "finished"), so nothing later may reach the validation scores
// either -- with bagging or feature_fraction < 1 a later iteration could split again, and its trees would be in the CV scores but not
// in the model.  any_split[it - 1] is final when iteration it's kernels start (stream order); a frozen iteration leaves
// its own flag at zero, which freezes the next one.
__device__ __forceinline__ bool small_fit_frozen(const SmallFit& sf, int it) { return it > 0 && sf.any_split[it - 1] == 0; }

// ------------------------------------------------------------------------------------------------
// gradients of every fit: grid (row tiles, fits), block 256
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_small_grad(const SmallFit* __restrict__ fits, int it) {
    const SmallFit& sf = fits[blockIdx.y];
    if (it >= sf.n_estimators || small_fit_frozen(sf, it)) return;
    const TrainConst c = sf.c;
    const uint8_t* inbag = sf.bag_freq > 0 ? sf.inbag : nullptr;
    const long long first = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    if (c.objective == 0) grad_rows<0>(first, stride, sf.score, sf.ycol, sf.y_value, sf.class_w, nullptr, inbag, sf.gh, nullptr, 0, c);
    else if (c.objective == 1) grad_rows<1>(first, stride, sf.score, sf.ycol, sf.y_value, sf.class_w, nullptr, inbag, sf.gh, nullptr, 0, c);
    else grad_rows<2>(first, stride, sf.score, sf.ycol, sf.y_value, sf.class_w, nullptr, inbag, sf.gh, nullptr, 0, c);
}

// ------------------------------------------------------------------------------------------------
// GBDT::Bagging of the fits that bag at this iteration: grid (LCG blocks / 64, fits), block 64; then the bag / out-of-bag lists
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_small_bagging(const SmallFit* __restrict__ fits, int it) {
    const SmallFit& sf = fits[blockIdx.y];
    if (sf.bag_freq <= 0 || it >= sf.n_estimators || it % sf.bag_freq != 0) return;
    const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
    if (b == 0) { sf.bagcnt[0] = 0u; sf.bagcnt[1] = 0u; }
    const long long n = sf.c.n_train, g0 = b * 1024;
    if (b >= sf.bag_nrb || g0 >= n) return;
    unsigned int x = sf.rand_state[b];
    const long long g1 = g0 + 1024 < n ? g0 + 1024 : n;
    for (long long g = g0; g < g1; ++g) {   // one LCG per 1024 training-row positions (k_bagging)
        x = 214013u * x + 2531011u;
        const float f = (float)((x >> 16) & 0x7FFF) / 32768.0f;
        sf.inbag[sf.sorted_rows[g]] = ((double)f < sf.bag_fraction) ? 1 : 0;
    }
    sf.rand_state[b] = x;
}

__global__ __launch_bounds__(256) void k_small_bag_lists(const SmallFit* __restrict__ fits, int it) {
    const SmallFit& sf = fits[blockIdx.y];
    if (sf.bag_freq <= 0 || it >= sf.n_estimators || it % sf.bag_freq != 0) return;
    const long long n_train = sf.c.n_train;
    const int lane = threadIdx.x & 63;
    for (long long p0 = (long long)blockIdx.x * 256; p0 < n_train; p0 += (long long)gridDim.x * 256) {
        const long long p = p0 + threadIdx.x;
        int row = 0; bool in = false, out = false;
        if (p < n_train) { row = sf.sorted_rows[p]; in = sf.inbag[row] != 0; out = !in; }
        const unsigned long long mi = __ballot(in), mo = __ballot(out);
        unsigned int bi = 0, bo = 0;
        if (lane == 0) { if (mi) bi = atomicAdd(&sf.bagcnt[0], (unsigned)__popcll(mi)); if (mo) bo = atomicAdd(&sf.bagcnt[1], (unsigned)__popcll(mo)); }
        bi = __shfl(bi, 0); bo = __shfl(bo, 0);
        if (in) sf.base_idx[bi + __popcll(mi & ((1ull << lane) - 1))] = row;
        if (out) sf.oob[bo + __popcll(mo & ((1ull << lane) - 1))] = row;
    }
}

// out-of-bag rows take the new tree's output by traversal (ScoreUpdater::AddScore(tree, oob)): grid (row tiles, class trees of the batch)
__global__ __launch_bounds__(256) void k_small_oob(const SmallFit* __restrict__ fits, const int32_t* __restrict__ tree2fit, int it) {
    const SmallFit& sf = fits[tree2fit[blockIdx.y]];
    if (sf.bag_freq <= 0 || it >= sf.n_estimators || small_fit_frozen(sf, it)) return;
    const int k = (int)blockIdx.y - sf.tree0;
    if (sf.tree_L[k] <= 1) return;
    const TrainConst& c = sf.c;
    const long long n_oob = sf.bagcnt[1];
    const long long tbase = (long long)it * c.K + k;
    const long long nb = tbase * (c.num_leaves - 1);
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(sf.rec);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_oob; i += (long long)gridDim.x * 256) {
        const int row = sf.oob[i];
        int node = 0;
        for (;;) {
            const int f = sf.out.feat[nb + node];
            const int bin = rec8[((long long)(f >> 4) * c.N + row) * 16 + (f & 15)];
            const bool miss = sf.fmeta[f].has_nan && bin == sf.fmeta[f].V;
            const bool go_left = miss ? (sf.out.dleft[nb + node] != 0) : (bin <= sf.out.theta[nb + node]);
            const int nx = go_left ? sf.out.left[nb + node] : sf.out.right[nb + node];
            if (nx < 0) { sf.score[(long long)k * c.N + row] += sf.upd[(long long)k * c.num_leaves + (~nx)]; break; }
            node = nx;
        }
    }
}

// The tree of class k just grown by fit f is added to the scores of the fit's VALIDATION rows (LightGBM's valid ScoreUpdater; the
// reference scores a CV fold with model.predict after the fit, python/repair/train.py:171-172): the same leaf values in the same order
// as the predictor adds them, so the final scores are the predictor's bits.  grid (row tiles, class trees of the batch), block 256.
__global__ __launch_bounds__(256) void k_small_valid(const SmallFit* __restrict__ fits, const int32_t* __restrict__ tree2fit, int it) {
    const SmallFit& sf = fits[tree2fit[blockIdx.y]];
    if (sf.n_valid <= 0 || it >= sf.n_estimators || small_fit_frozen(sf, it)) return;
    const int k = (int)blockIdx.y - sf.tree0;
    if (sf.tree_L[k] <= 1) return;                         // no split: the tree adds nothing (its constant is the initial score)
    const TrainConst& c = sf.c;
    const long long nv = sf.n_valid;
    const long long tbase = (long long)it * c.K + k;
    const long long nb = tbase * (c.num_leaves - 1);
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(sf.vrec);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        int node = 0;
        for (;;) {
            const int f = sf.out.feat[nb + node];
            const int bin = rec8[((long long)(f >> 4) * nv + i) * 16 + (f & 15)];
            const bool go_left = (bin == 255) ? (sf.out.dleft[nb + node] != 0) : (bin <= sf.out.theta[nb + node]);     // k_predict_raw's rule
            const int nx = go_left ? sf.out.left[nb + node] : sf.out.right[nb + node];
            if (nx < 0) { sf.vscore[(long long)k * nv + i] += sf.upd[(long long)k * c.num_leaves + (~nx)]; break; }
            node = nx;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_small_tree: grid (class trees of the batch), block SM_THREADS; dynamic LDS = the largest sm_lds_bytes() of the batch
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SM_THREADS, 4 /* waves per SIMD: at most 128 VGPRs, so that two 512-thread workgroups share a CU */) void k_small_tree(const SmallFit* __restrict__ fits, const int32_t* __restrict__ tree2fit, int it) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ TreeState st;
    __shared__ int wl[SM_WAVES], wr[SM_WAVES];
    __shared__ unsigned long long fxq[SM_WAVES][2];
    __shared__ FxScale fx_tree;
    const SmallFit& sf = fits[tree2fit[blockIdx.x]];
    if (it >= sf.n_estimators || small_fit_frozen(sf, it)) return;
    TrainConst c = sf.c;
    const int k = (int)blockIdx.x - sf.tree0, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NL = c.num_leaves, F = c.F;
    size_t off = sm_hist_area((size_t)sf.lds_hist, NL);
    Leaf* lk = reinterpret_cast<Leaf*>(smem + off);
    off = (off + (size_t)NL * sizeof(Leaf) + 15) & ~(size_t)15;
    Cand* ck = reinterpret_cast<Cand*>(smem + off);
    off = (off + (size_t)2 * F * sizeof(Cand) + 15) & ~(size_t)15;
    const int W = sf.scan_waves;                                    // packed threshold scan: waves per child (0: one wave per feature)
    HistBin* histL = reinterpret_cast<HistBin*>(smem + off);        // [totbins] compact histogram of the left child / the root (packed scan only)
    HistBin* histR = histL + c.totbins;                             // [totbins] ... of the right child
    const long long N = c.N;
    const long long n_in = sf.bag_freq > 0 ? (long long)sf.bagcnt[0] : c.n_train;
    const long long tbase = (long long)it * c.K + k;
    const TreeOut out = sf.out;
    HistBin* pk = sf.pool + (long long)k * NL * c.totbins;
    int32_t* i0 = sf.idx0 + (long long)k * c.n_train;
    int32_t* i1 = sf.idx1 + (long long)k * c.n_train;
    const int32_t* base_idx = sf.base_idx;
    const float2* ghk = sf.gh + (long long)k * c.NG;
    const FeatMeta* fmeta = sf.fmeta; const ChunkMeta* cmeta = sf.cmeta;
    const uint8_t* used_k = sf.used + ((long long)it * c.K + k) * F;
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(sf.rec);
#if defined(SM_PROF)
    unsigned long long pt_ = __builtin_readcyclecounter();
#define SM_T(i) { if (tid == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); atomicAdd(sf.prof + (i), n_ - pt_); pt_ = n_; } }
#else
#define SM_T(i)
#endif

    // ---- numerics v2.2: this class tree's fixed-point grid, from the coarse sums of its (g, h) (k_fx_measure + k_fx_scale of the big-table
    // trainer; rows outside the training set / the bag carry (0, 0))
    {
        unsigned long long a0 = 0ull, a1 = 0ull;
        for (long long p = tid; p < N; p += SM_THREADS) { const float2 g = ghk[p]; a0 += fx_coarse(g.x, c.fx.c_g); a1 += fx_coarse(g.y, c.fx.c_h); }
        a0 = wave_sum_u64(a0); a1 = wave_sum_u64(a1);
        if (lane == 0) { fxq[wv][0] = a0; fxq[wv][1] = a1; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long q0 = 0ull, q1 = 0ull;
            for (int w = 0; w < SM_WAVES; ++w) { q0 += fxq[w][0]; q1 += fxq[w][1]; }
            const int e_g = fx_tree_exponent(q0, c.fx.q_mult, c.fx.c_g, c.fx.e_g_min, c.fx.e_g_max), e_h = fx_tree_exponent(q1, c.fx.q_mult, c.fx.c_h, c.fx.e_h_min, c.fx.e_h_max);
            FxScale f; f.sg = fx_pow2(e_g); f.sh = fx_pow2(e_h); f.inv_sg = fx_pow2(-e_g); f.inv_sh = fx_pow2(-e_h);
            fx_tree = f;
        }
        __syncthreads();
        c.sg = fx_tree.sg; c.sh = fx_tree.sh; c.inv_sg = fx_tree.inv_sg; c.inv_sh = fx_tree.inv_sh;
    }
    // ---- k_init_iter
    if (tid == 0) {
        TreeState s; memset(&s, 0, sizeof(s));
        s.L = 1; s.done = 0; s.hist_is_root = 1; s.hist_begin = 0; s.hist_count = (int)n_in; s.hist_buf = 2;
        s.do_hist = (n_in < (long long)c.min_data_in_leaf * 2) ? 0 : 1;
        st = s;
        Leaf r; memset(&r, 0, sizeof(r));
        r.begin = 0; r.count = (int)n_in; r.buf = 2; r.depth = 0; r.parent_node = -1; r.is_left = 0;
        r.best.gain = -INFINITY; r.best_feature = -1;
        lk[0] = r;
        out.L[tbase] = 1;
        out.leaf_count[tbase * NL] = (int)n_in;
    }
    __syncthreads();

    for (int step = 0; step < NL - 1; ++step) {
        // ---- k_hist: histogram of the root / of the smaller child of the last split, chunk by chunk, into its pool slot
        if (st.do_hist) {
            const bool is_root = st.hist_is_root != 0;
            const long long cnt = is_root ? N : (long long)st.hist_count;
            const int hb = st.hist_buf, hbeg = st.hist_begin;
            const int32_t* idx = hb == 0 ? i0 : (hb == 1 ? i1 : base_idx);
            HistBin* dst = pk + (long long)(is_root ? 0 : st.right_leaf) * c.totbins;
            for (int ch = 0; ch < c.nchunk; ++ch) {
                const ChunkMeta cm = cmeta[ch];
                const FeatMeta* fm = fmeta + cm.first_feat;
                unsigned long long* fast_g = reinterpret_cast<unsigned long long*>(smem);    // replicated gradient sums, then hessian sums (k_hist)
                unsigned long long* fast_h = fast_g + cm.fast_slots;
                for (int i = tid; i < 2 * cm.fast_slots; i += SM_THREADS) fast_g[i] = 0ull;
                int fbase[16], fshift[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j < cm.nfeat) { fbase[j] = fm[j].fast_base; fshift[j] = fm[j].rep_shift; } else { fbase[j] = 0; fshift[j] = 0; }
                }
                __syncthreads();
                SM_T(0)
                const uint4* recc = sf.rec + (long long)ch * N;
                for (long long p = tid; p < cnt; p += SM_THREADS) {
                    const long long row = is_root ? p : (long long)idx[hbeg + p];
                    const uint4 r = recc[row];
                    const float2 g = ghk[row];
                    if (g.x != 0.0f || g.y != 0.0f) {   // non-training / out-of-bag rows carry (0, 0)
                        const unsigned long long gq = (unsigned long long)fx_from_f32(g.x, c.sg), hq = (unsigned long long)fx_from_f32(g.y, c.sh);
                        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            if (j < cm.nfeat) {
                                const uint32_t bin = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                                const int slot = fbase[j] + (int)(bin << fshift[j]) + (lane & ((1 << fshift[j]) - 1));
                                atomicAdd(&fast_g[slot], gq);
                                atomicAdd(&fast_h[slot], hq);
                            }
                        }
                    }
                }
                __syncthreads();
                SM_T(1)
                // this workgroup owns the whole histogram: plain stores of every bin (no zeroing, no global atomics)
                if (W > 0) {
                    // ONE pass, a thread per bin of the chunk: replicas -> the built child's bin; the sibling = parent - built (exact
                    // integers); both children go to their pool slots (the parents of later splits) and, compact, into LDS for the scan
                    const bool sm_left = st.smaller_is_left != 0;
                    HistBin* pl = pk + (long long)st.split_leaf * c.totbins; HistBin* pr = pk + (long long)st.right_leaf * c.totbins;
                    for (int wbin = tid; wbin < cm.wide_bins; wbin += SM_THREADS) {
                        int j = 0;
                        for (int q = 1; q < cm.nfeat; ++q) j += (wbin >= fm[q].wide_off) ? 1 : 0;
                        const int b = wbin - fm[j].wide_off, sh = fm[j].rep_shift, s0 = fm[j].fast_base + (b << sh), gb = fm[j].hoff + b;
                        HistBin par; par.g = 0; par.h = 0;
                        if (!is_root) par = pl[gb];
                        long long tg = 0, th = 0;
                        for (int r2 = 0; r2 < (1 << sh); ++r2) { tg += (long long)fast_g[s0 + r2]; th += (long long)fast_h[s0 + r2]; }
                        HistBin built; built.g = tg; built.h = th;
                        if (is_root) { pk[gb] = built; histL[gb] = built; }
                        else {
                            HistBin sib; sib.g = par.g - built.g; sib.h = par.h - built.h;
                            const HistBin l = sm_left ? built : sib, r = sm_left ? sib : built;
                            pl[gb] = l; pr[gb] = r; histL[gb] = l; histR[gb] = r;
                        }
                    }
                } else
                for (int j = 0; j < cm.nfeat; ++j) {
                    const int sh = fm[j].rep_shift;
                    for (int b = tid; b < fm[j].nbins; b += SM_THREADS) {
                        long long tg = 0, th = 0;
                        const int s0 = fm[j].fast_base + (b << sh);
                        for (int r2 = 0; r2 < (1 << sh); ++r2) { tg += (long long)fast_g[s0 + r2]; th += (long long)fast_h[s0 + r2]; }
                        HistBin o; o.g = tg; o.h = th;
                        dst[fm[j].hoff + b] = o;
                    }
                }
                __syncthreads();
            }
            SM_T(2)
            if (W > 0) {
                // ---- FindBestThreshold, packed: the features of a child share waves; both children at once
                const int child = wv / W, slot = wv - child * W;
                if (wv < (is_root ? W : 2 * W)) {
                    const ScanLane sl = sf.scan_map[slot * 64 + lane];
                    if (is_root) {
                        // leaf totals = sum over all bins of any one feature (every row sits in exactly one bin)
                        const FeatMeta f0 = fmeta[0];
                        long long sg_ = 0, sh_ = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const int b = lane * 4 + j; if (b < f0.nbins) { const HistBin v = histL[f0.hoff + b]; sg_ += v.g; sh_ += v.h; } }
#pragma unroll
                        for (int o2 = 32; o2 >= 1; o2 >>= 1) { sg_ += __shfl_xor(sg_, o2); sh_ += __shfl_xor(sh_, o2); }
                        if (wv == 0 && lane == 0) { lk[0].Gq = sg_; lk[0].Hq = sh_; }
                        scan_child_packed(histL, fmeta, sl, sg_, sh_, (long long)lk[0].count, c, used_k, ck);
                    } else {
                        const Leaf lf = lk[child == 0 ? st.split_leaf : st.right_leaf];
                        scan_child_packed(child == 0 ? histL : histR, fmeta, sl, lf.Gq, lf.Hq, (long long)lf.count, c, used_k, ck + child * F);
                    }
                }
            } else
            // ---- k_split_find: one wave per feature
            for (int f = wv; f < F; f += SM_WAVES) split_find_body(pk, st, lk, fmeta, used_k, ck, f, c);
        }
        __syncthreads();
        SM_T(3)
        // ---- k_tree_step
        if (st.do_hist) {
            if (st.hist_is_root) { if (wv == 0) reduce_leaf_best(ck, F, &lk[0]); }
            else {
                if (wv == 0) reduce_leaf_best(ck, F, &lk[st.split_leaf]);
                if (wv == 1) reduce_leaf_best(ck + F, F, &lk[st.right_leaf]);
            }
        }
        __syncthreads();
        if (wv == 0) tree_step_pick<false>(&st, lk, pk, fmeta, out, tbase, c);
        __syncthreads();
        SM_T(4)
        if (st.done) break;
        // ---- k_partition: lefts fill the parent's range from the front, rights from the back, in the other buffer
        int nl_total = 0;
        if (st.do_partition) {
            const int cnt = st.part_count, beg = st.part_begin, pb = st.part_buf;
            const int32_t* src = pb == 0 ? i0 : (pb == 1 ? i1 : base_idx);
            int32_t* dst = pb == 0 ? i1 : i0;   // base list (2) -> buffer 0
            const int f = st.part_feature, theta = st.part_theta, dleft = st.part_dleft, nanbin = st.part_nanbin;
            const uint8_t* recf = rec8 + ((long long)(f >> 4) * N) * 16 + (f & 15);
            const int ntiles = (cnt + 4 * SM_THREADS - 1) / (4 * SM_THREADS);
            int curl = 0, curr = 0;   // uniform running totals
            for (int t = 0; t < ntiles; ++t) {
                int rows[4]; bool gl[4]; bool on[4];
                int nl = 0, nr = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = t * 4 * SM_THREADS + j * SM_THREADS + tid;
                    on[j] = p < cnt; gl[j] = false; rows[j] = 0;
                    if (on[j]) {
                        rows[j] = src[beg + p];
                        const int bin = recf[(long long)rows[j] * 16];
                        gl[j] = (bin == nanbin) ? (dleft != 0) : (bin <= theta);
                        if (gl[j]) ++nl; else ++nr;
                    }
                }
                int sl = nl, sr = nr;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int a = __shfl_up(sl, o), b = __shfl_up(sr, o); if (lane >= o) { sl += a; sr += b; } }
                if (lane == 63) { wl[wv] = sl; wr[wv] = sr; }
                __syncthreads();
                int ol = sl - nl, orr = sr - nr, totl = 0, totr = 0;
#pragma unroll
                for (int w = 0; w < SM_WAVES; ++w) { if (w < wv) { ol += wl[w]; orr += wr[w]; } totl += wl[w]; totr += wr[w]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (on[j]) {
                        if (gl[j]) dst[beg + curl + ol++] = rows[j];
                        else dst[beg + cnt - 1 - (curr + orr++)] = rows[j];
                    }
                }
                curl += totl; curr += totr;
                __syncthreads();
            }
            nl_total = curl;
        }
        __syncthreads();
        SM_T(5)
        // ---- k_finish_split
        if (tid == 0) finish_split_body(&st, lk, out, tbase, nl_total, c);
        __syncthreads();
        SM_T(6)
    }

    // ---- k_finalize_tree + k_score_update
    const int L = st.L;
    double* lv = out.leaf_value + tbase * NL;
    double* uv = sf.upd + (long long)k * NL;
    if (tid == 0) sf.tree_L[k] = L;
    if (L <= 1) {
        if (tid == 0) { lv[0] = (it == 0) ? sf.init[k] : 0.0; uv[0] = 0.0; }
        return;
    }
    if (tid == 0) atomicOr(sf.any_split + it, 1);
    double* suv = reinterpret_cast<double*>(smem);      // the histogram area is free now
    for (int l = tid; l < L; l += SM_THREADS) {
        double v = lv[l] * c.learning_rate;    // Tree::Shrinkage
        uv[l] = v; suv[l] = v;
        if (it == 0 && fabs(sf.init[k]) > k_eps()) v += sf.init[k];   // Tree::AddBias (model only; scores already hold init)
        lv[l] = v;
    }
    __syncthreads();
    double* sk = sf.score + (long long)k * N;
    for (int l = 0; l < L; ++l) {
        const Leaf lf = lk[l];
        const int32_t* src = lf.buf == 0 ? i0 : (lf.buf == 1 ? i1 : base_idx);
        const double d = suv[l];
        for (int p = tid; p < lf.count; p += SM_THREADS) { const int row = src[lf.begin + p]; sk[row] += d; }
    }
    SM_T(7)
#undef SM_T
}

}  // namespace rg
