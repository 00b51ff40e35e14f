// rgbm_small.h -- the fused small-table grower: ONE workgroup grows ONE class tree of a boosting iteration, start to finish,
// inside a single launch.
//
// Why it was built: on a table of ~10^4 rows (the reference trains every model on a 10 000-row sample by default, model.py:755-766,
// and the hyper-parameter search runs ~50 such fits per target) neither grower of this library is bound by data.  The level
// grower is a chain of ~35 dependent kernels per iteration whose own fixed costs add up to ~420 us; the leaf-wise
// grower launches 5 kernels per split.  Here the whole leaf-wise loop of rgbm_kernels.h -- histogram of the smaller child,
// subtraction + threshold scans, best-leaf pick, Tree::Split bookkeeping, row partition, shrinkage and AddScore -- runs in one
// workgroup with barriers where the other path has kernel boundaries: 2 launches per iteration (gradients + this).
// The arithmetic IS the leaf-wise grower's: the same device functions (scan_child, split_find_body, reduce_leaf_best,
// tree_step_pick, finish_split_body), the same integer histograms, so the trees are the same bits.
//
// What the measurement said (MI355X, round 2): bit-exact on every test case, but SLOWER than the level grower where it was
// supposed to win -- 0.72 vs 0.50 ms per boosting iteration on 10 000 rows (K = 8), the reference-default job 3.43 vs 1.88 s:
// ~30 sequential splits per tree, each a chain of L2 round trips (index list -> bin record -> histogram pool -> threshold
// scans on 8 waves) of ~24 us, against 7 levels whose scans spread over the whole chip.  It therefore stays an OPT-IN
// (RGBM_GROWER=small / RGBM_SMALL_ROWS), kept under test as the starting point for the batched variant of DESIGN 9.5.
//
// State: leaves, per-feature candidates and the control block live in LDS; row-index lists (ping-pong, as k_partition) and the
// per-leaf histogram pool stay in global memory (L2-resident at this size).  grid (K), block SM_THREADS.
#pragma once
#include "rgbm_kernels.h"

namespace rg {

constexpr int SM_THREADS = 512;
constexpr int SM_WAVES = SM_THREADS / 64;
constexpr int SM_MAX_LEAVES = 256;
constexpr int SM_MAX_FEATS = 128;

__host__ __device__ inline size_t sm_lds_bytes(size_t lds_hist, int num_leaves, int F) {
    size_t b = (lds_hist + 15) & ~(size_t)15;
    b += (size_t)num_leaves * sizeof(Leaf);
    b = (b + 15) & ~(size_t)15;
    b += (size_t)2 * F * sizeof(Cand);
    return b + 64;
}

__global__ __launch_bounds__(SM_THREADS) void k_small_tree(const uint4* __restrict__ rec, const int2* __restrict__ gh,
                                                           int32_t* __restrict__ idx0, int32_t* __restrict__ idx1, const int32_t* __restrict__ base_idx,
                                                           HistBin* __restrict__ pool, const FeatMeta* __restrict__ fmeta, const ChunkMeta* __restrict__ cmeta,
                                                           const uint8_t* __restrict__ used_it /* [K][F] of this iteration */, TreeOut out,
                                                           const double* __restrict__ init, double* __restrict__ upd_value /* [K][NL] (out-of-bag update) */,
                                                           TreeState* __restrict__ state_out /* [K]: L of the finished tree (out-of-bag update) */,
                                                           int32_t* __restrict__ any_split, double* __restrict__ score,
                                                           const unsigned int* __restrict__ n_in_ptr, int it, size_t lds_hist, TrainConst c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ TreeState st;
    __shared__ int wl[SM_WAVES], wr[SM_WAVES];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NL = c.num_leaves, F = c.F;
    size_t off = (lds_hist + 15) & ~(size_t)15;
    Leaf* lk = reinterpret_cast<Leaf*>(smem + off);
    off = (off + (size_t)NL * sizeof(Leaf) + 15) & ~(size_t)15;
    Cand* ck = reinterpret_cast<Cand*>(smem + off);
    const long long N = c.N;
    const long long n_in = n_in_ptr ? (long long)n_in_ptr[0] : c.n_train;
    const long long tbase = (long long)it * c.K + k;
    HistBin* pk = pool + (long long)k * NL * c.totbins;
    int32_t* i0 = idx0 + (long long)k * c.n_train;
    int32_t* i1 = idx1 + (long long)k * c.n_train;
    const int2* ghk = gh + (long long)k * N;
    const uint8_t* used_k = used_it + (long long)k * F;
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(rec);

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
                unsigned long long* fast = reinterpret_cast<unsigned long long*>(smem);
                HistBin* wide = reinterpret_cast<HistBin*>(smem + (size_t)cm.fast_slots * 8);
                for (int i = tid; i < cm.fast_slots; i += SM_THREADS) fast[i] = 0ull;
                for (int i = tid; i < cm.wide_bins; i += SM_THREADS) { wide[i].g = 0; wide[i].h = 0; }
                int fbase[16], fshift[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j < cm.nfeat) { fbase[j] = fm[j].fast_base; fshift[j] = fm[j].rep_shift; } else { fbase[j] = 0; fshift[j] = 0; }
                }
                __syncthreads();
                const uint4* recc = rec + (long long)ch * N;
                const long long ntiles = (cnt + TILE_ROWS - 1) / TILE_ROWS;
                for (long long t = 0; t < ntiles; ++t) {
                    const long long p0 = t * TILE_ROWS;
#pragma unroll 2
                    for (int s = 0; s < TILE_ROWS / SM_THREADS; ++s) {
                        const long long p = p0 + s * SM_THREADS + tid;
                        if (p < cnt) {
                            const long long row = is_root ? p : (long long)idx[hbeg + p];
                            const uint4 r = recc[row];
                            const int2 g = ghk[row];
                            const unsigned long long packed = ((unsigned long long)(long long)g.x << 32) + (unsigned long long)(unsigned int)g.y;
                            if (packed != 0ull) {
                                const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                                for (int j = 0; j < 16; ++j) {
                                    if (j < cm.nfeat) {
                                        const uint32_t bin = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                                        const int slot = fbase[j] + (int)(bin << fshift[j]) + (lane & ((1 << fshift[j]) - 1));
                                        atomicAdd(&fast[slot], packed);
                                    }
                                }
                            }
                        }
                    }
                    __syncthreads();
                    // drain packed -> wide (2048 rows per tile: no field can overflow)
                    for (int j = 0; j < cm.nfeat; ++j) {
                        const int nslots = fm[j].nbins << fm[j].rep_shift;
                        for (int s2 = tid; s2 < nslots; s2 += SM_THREADS) {
                            const unsigned long long v = fast[fm[j].fast_base + s2];
                            if (v) {
                                fast[fm[j].fast_base + s2] = 0ull;
                                const int bin = s2 >> fm[j].rep_shift;
                                const long long gq = (long long)(int)(v >> 32);
                                const long long hq = (long long)(unsigned int)(v & 0xFFFFFFFFull);
                                HistBin* wb = &wide[fm[j].wide_off + bin];
                                atomicAdd(reinterpret_cast<unsigned long long*>(&wb->g), (unsigned long long)gq);
                                atomicAdd(reinterpret_cast<unsigned long long*>(&wb->h), (unsigned long long)hq);
                            }
                        }
                    }
                    __syncthreads();
                }
                // this workgroup owns the whole histogram: plain stores of every bin (no zeroing, no global atomics)
                for (int j = 0; j < cm.nfeat; ++j)
                    for (int b = tid; b < fm[j].nbins; b += SM_THREADS) dst[fm[j].hoff + b] = wide[fm[j].wide_off + b];
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
                if (wv == (SM_WAVES > 1 ? 1 : 0)) reduce_leaf_best(ck + F, F, &lk[st.right_leaf]);
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
    double* uv = upd_value + (long long)k * NL;
    if (tid == 0) state_out[k].L = L;
    if (L <= 1) {
        if (tid == 0) { lv[0] = (it == 0) ? init[k] : 0.0; uv[0] = 0.0; }
        return;
    }
    if (tid == 0) atomicOr(any_split + it, 1);
    double* suv = reinterpret_cast<double*>(smem);      // the histogram area is free now
    for (int l = tid; l < L; l += SM_THREADS) {
        double v = lv[l] * c.learning_rate;    // Tree::Shrinkage
        uv[l] = v; suv[l] = v;
        if (it == 0 && fabs(init[k]) > k_eps()) v += init[k];   // Tree::AddBias (model only; scores already hold init)
        lv[l] = v;
    }
    __syncthreads();
    double* sk = score + (long long)k * N;
    for (int l = 0; l < L; ++l) {
        const Leaf lf = lk[l];
        const int32_t* src = lf.buf == 0 ? i0 : (lf.buf == 1 ? i1 : base_idx);
        const double d = suv[l];
        for (int p = tid; p < lf.count; p += SM_THREADS) { const int row = src[lf.begin + p]; sk[row] += d; }
    }
}

}  // namespace rg
