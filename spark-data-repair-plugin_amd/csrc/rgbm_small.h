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
// finish_split_body, grad_rows), the same exact integer histograms (numerics v2.1), so every model is the one rgbm_table_train
// returns for that fit, bit for bit (tests/test_gpu_batch.py).
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
};

// the histogram area doubles as the leaf-delta table of the final score update: at least num_leaves doubles
__host__ __device__ inline size_t sm_hist_area(size_t lds_hist, int num_leaves) {
    const size_t a = lds_hist > (size_t)num_leaves * 8 ? lds_hist : (size_t)num_leaves * 8;
    return (a + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t sm_lds_bytes(size_t lds_hist, int num_leaves, int F) {
    size_t b = sm_hist_area(lds_hist, num_leaves);
    b += (size_t)num_leaves * sizeof(Leaf);
    b = (b + 15) & ~(size_t)15;
    b += (size_t)2 * F * sizeof(Cand);
    return b + 64;
}

// ------------------------------------------------------------------------------------------------
// gradients of every fit: grid (row tiles, fits), block 256
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_small_grad(const SmallFit* __restrict__ fits, int it) {
    const SmallFit& sf = fits[blockIdx.y];
    if (it >= sf.n_estimators) return;
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
    if (sf.bag_freq <= 0 || it >= sf.n_estimators) return;
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

// ------------------------------------------------------------------------------------------------
// k_small_tree: grid (class trees of the batch), block SM_THREADS; dynamic LDS = the largest sm_lds_bytes() of the batch
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SM_THREADS, 4 /* waves per SIMD: at most 128 VGPRs, so that two 512-thread workgroups share a CU */) void k_small_tree(const SmallFit* __restrict__ fits, const int32_t* __restrict__ tree2fit, int it) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ TreeState st;
    __shared__ int wl[SM_WAVES], wr[SM_WAVES];
    const SmallFit& sf = fits[tree2fit[blockIdx.x]];
    if (it >= sf.n_estimators) return;
    const TrainConst c = sf.c;
    const int k = (int)blockIdx.x - sf.tree0, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NL = c.num_leaves, F = c.F;
    size_t off = sm_hist_area((size_t)sf.lds_hist, NL);
    Leaf* lk = reinterpret_cast<Leaf*>(smem + off);
    off = (off + (size_t)NL * sizeof(Leaf) + 15) & ~(size_t)15;
    Cand* ck = reinterpret_cast<Cand*>(smem + off);
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
                // this workgroup owns the whole histogram: plain stores of every bin (no zeroing, no global atomics)
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
            // ---- k_split_find: one wave per feature
            for (int f = wv; f < F; f += SM_WAVES) split_find_body(pk, st, lk, fmeta, used_k, ck, f, c);
        }
        __syncthreads();
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
        // ---- k_finish_split
        if (tid == 0) finish_split_body(&st, lk, out, tbase, nl_total, c);
        __syncthreads();
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
}

}  // namespace rg
