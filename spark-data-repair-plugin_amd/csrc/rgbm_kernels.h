// rgbm_kernels.h -- hand-written HIP kernels for gfx950 (wave64, 160 KiB LDS/CU).
//
// Kernel inventory (LightGBM concept each one re-creates):
//   k_count_codes    per-feature code frequencies of the training rows   (BinMapper::FindBin input)
//   k_pack_bins      int32 codes -> u8 bin records, 16 features / 16 B    (Dataset construction)
//   k_grad_*         scores -> float32 (g,h) per row and class            (ObjectiveFunction::GetGradients)
//   k_hist           LDS histogram scan of the leaf-wise grower             (ConstructHistograms)
//   k_split_find     parent-minus-child subtraction + per-feature threshold scan (FindBestThreshold)
//   k_tree_step      argmax over features/leaves, Tree::Split bookkeeping
//   k_partition      unstable two-cursor row partition                    (DataPartition::Split)
//   k_finish_split   child ranges, BeforeFindBestSplit checks, smaller-child choice
//   k_finalize_tree / k_score_update   shrinkage, AddBias, ScoreUpdater::AddScore
//   k_predict_raw / k_softmax_argmax / k_fill_cells   GBDT::PredictRaw + chain fill (model.py:1118-1133)
//
// No path here is a dense contraction, so MFMA is unused by design; every kernel is an HBM scan
// or an LDS/atomic-bound reduction.
#pragma once
#include <hip/hip_runtime.h>
#include "rgbm_device.h"
#include "rgbm_numerics.h"

namespace rg {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// (g, h) of one (row, class tree): float2 [K][N] -- LightGBM's score_t pair, the double expression rounded once to float32
__device__ __forceinline__ void store_gh(float2* gh, long long idx, double g, double h) { gh[idx] = make_float2((float)g, (float)h); }

// the constants of class tree k: the model's, with the fixed-point grid this tree has in this iteration (numerics v2.2, k_fx_scale)
__device__ __forceinline__ TrainConst tree_const(const TrainConst& c, const FxScale* __restrict__ fxs, int k) {
    TrainConst t = c;
    const FxScale f = fxs[k];
    t.sg = f.sg; t.sh = f.sh; t.inv_sg = f.inv_sg; t.inv_sh = f.inv_sh;
    return t;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct TreeOut {   // flat device arrays of every tree of the model, [(it*K+k)] major
    int32_t* L; int32_t* feat; int32_t* theta; int32_t* dleft; int32_t* left; int32_t* right; double* gain;
    double* leaf_value; int32_t* leaf_count;
};

// ------------------------------------------------------------------------------------------------
// K0: code frequencies.  grid (gx, F), block 256.  Small dictionaries are privatised in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int COUNT_LDS_CODES = 8192;
__global__ __launch_bounds__(256) void k_count_codes(const int32_t* __restrict__ codes, long long N,
                                                     const int32_t* __restrict__ ycol /* training rows: ycol[row] >= 0; may be null */,
                                                     const int32_t* __restrict__ feat_col, const int32_t* __restrict__ n_codes,
                                                     const long long* __restrict__ cnt_off, unsigned int* __restrict__ cnt,
                                                     const uint8_t* __restrict__ mult = nullptr /* rows with multiplicities (rgbm_table_set_row_multiplicity): a row counts mult[row] times */) {
    __shared__ unsigned int lc[COUNT_LDS_CODES];
    const int f = blockIdx.y;
    const int32_t* col = codes + (long long)feat_col[f] * N;
    const int nc = n_codes[f];
    unsigned int* out = cnt + cnt_off[f];
    const bool use_lds = nc <= COUNT_LDS_CODES;
    if (use_lds) { for (int i = threadIdx.x; i < nc; i += 256) lc[i] = 0; __syncthreads(); }
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < N; r += (long long)gridDim.x * 256) {
        if (ycol && ycol[r] < 0) continue;
        int c = col[r];
        if (c < 0 || c >= nc) continue;
        const unsigned int m = mult ? (unsigned int)mult[r] : 1u;
        if (use_lds) atomicAdd(&lc[c], m); else atomicAdd(&out[c], m);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < nc; i += 256) { unsigned v = lc[i]; if (v) atomicAdd(&out[i], v); }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: bin_pack.  thread per row; one 16-byte record per (chunk,row).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_bins(const int32_t* __restrict__ codes, long long Ntab, long long row0, long long n,
                                                   const int32_t* __restrict__ feat_col, const int32_t* __restrict__ n_codes,
                                                   const long long* __restrict__ lut_off, const uint8_t* __restrict__ lut,
                                                   const uint8_t* __restrict__ miss_bin, int F, int nchunk, uint4* __restrict__ rec,
                                                   const uint8_t* __restrict__ mult = nullptr /* [Ntab] row multiplicities: ride in byte 15 of the LAST chunk's record (the caller made sure that chunk holds <= 15 features) */) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int ch = 0; ch < nchunk; ++ch) {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int f = ch * 16 + j;
            uint32_t b = 0;
            if (f < F) {
                int c = codes[(long long)feat_col[f] * Ntab + row0 + i];
                b = (c < 0 || c >= n_codes[f]) ? miss_bin[f] : lut[lut_off[f] + c];
            }
            w[j >> 2] |= b << (8 * (j & 3));
        }
        if (mult && ch == nchunk - 1) w[3] = (w[3] & 0x00FFFFFFu) | ((uint32_t)mult[row0 + i] << 24);
        rec[(long long)ch * n + i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__global__ void k_iota_train(const int32_t* __restrict__ ycol, long long N, int32_t* __restrict__ out, unsigned int* __restrict__ counter) {
    // unstable compaction of training rows (order is irrelevant: every sum downstream is an exact integer)
    long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool on = r < N && ycol[r] >= 0;
    unsigned long long m = __ballot(on);
    int lane = lane_id();
    unsigned int base = 0;
    if (lane == 0 && m) base = atomicAdd(counter, (unsigned)__popcll(m));
    base = __shfl(base, 0);
    if (on) out[base + __popcll(m & ((1ull << lane) - 1))] = (int32_t)r;
}

// ------------------------------------------------------------------------------------------------
// K2: gradients.  thread per row (grid-stride); scores read coalesced per class.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_score(double* __restrict__ score, long long N, int K, const double* __restrict__ init) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    for (int k = 0; k < K; ++k) score[(long long)k * N + i] = init[k];
}

// Level grower: AddScore of the PREVIOUS iteration's trees, folded into this iteration's gradient kernel (round 6).  The kernel reads every score anyway; with the final
// node id of the row (1 B) and the tree's node -> delta table it adds the previous tree's output on the way, writes the score back and continues with the new
// value -- the same double addition k_level_final made, on the same operands, so the scores (and everything after them) keep their bits.  k_level_final used to
// read and write all K x N scores in a pass of its own (17 B per (row, class tree)); what is left of it is the last routing step + the deepest counts
// (k_level_last: the node ids only).  node == nullptr: nothing pending (first iteration, leaf-wise grower, batched small fits).
struct PendingScore {
    const uint8_t* node;      // [K][NS] the node every row ended in (LV_INACTIVE = 255: the row takes no part, its score never changes)
    const double* ndelta;     // [K][256] node -> Shrinkage * leaf output (k_level_replay)
    const int32_t* L;         // [K] leaves of the previous iteration's trees: <= 1 = no split, no score change
    long long NS;
};

// the rows first, first + stride, ... of one fit (k_grad: a grid-stride loop; k_small_grad: the same loop per fit of a batch)
template <int OBJ>
__device__ __forceinline__ void grad_rows(long long first, long long stride, const double* score /* (not restrict: the pending AddScore writes it) */, const int32_t* __restrict__ ycol,
                                          const double* __restrict__ y_value, const double* __restrict__ class_w,
                                          const double* __restrict__ sample_w, const uint8_t* __restrict__ row_in_bag /* null = no bagging */,
                                          float2* __restrict__ gh, uint8_t* __restrict__ node0 /* level grower: node ids to reset, or null */,
                                          long long NS, const TrainConst& c,
                                          unsigned long long* qacc = nullptr /* OBJ != 1: this thread's sums of the coarse magnitudes of its (g, h) (numerics v2.2), or null */,
                                          const uint8_t* __restrict__ mult = nullptr /* row multiplicities: a row's magnitudes count mult[row] times */,
                                          const PendingScore pd = PendingScore{nullptr, nullptr, nullptr, 0}) {
    const long long N = c.N;
    for (long long i = first; i < N; i += stride) {
        const int y = ycol[i];
        if (pd.node) {   // the previous iteration's AddScore (before the node ids are reset below)
            double* sc = const_cast<double*>(score);
            const int KK = (OBJ == 1) ? c.K : 1;
            for (int k = 0; k < KK; ++k) {
                const int n = pd.node[(long long)k * pd.NS + i];
                if (n != 255 && pd.L[k] > 1) sc[(long long)k * N + i] += pd.ndelta[k * 256 + n];
            }
        }
        if (node0) {   // every training row restarts in node 0 (the root); all other rows never take part
            const int KK = (OBJ == 1) ? c.K : 1;
            const uint8_t v = (y < 0) ? (uint8_t)255 : (uint8_t)0;
            for (int k = 0; k < KK; ++k) node0[(long long)k * NS + i] = v;
        }
        if (y < 0) continue;   // not a training row: its gh stays 0 for ever
        if (row_in_bag && !row_in_bag[i]) {   // out of bag this round: contributes nothing to any histogram
            const int K = (OBJ == 1) ? c.K : 1;
            for (int k = 0; k < K; ++k) store_gh(gh, (long long)k * c.NG + i, 0.0, 0.0);
            continue;
        }
        double wi = class_w ? class_w[y] : 1.0;
        if (sample_w) wi = wi * sample_w[i];
        wi = (double)(float)wi;   // LightGBM Metadata keeps weights as float32
        if (OBJ == 0) {   // binary_objective.hpp GetGradients (sigmoid = 1, label_weight = 1)
            double label = (y > 0) ? 1.0 : -1.0;
            double response = -label / (1.0 + rg_exp(label * score[i]));
            double abs_r = fabs(response);
            const float g32 = (float)(response * wi), h32 = (float)(abs_r * (1.0 - abs_r) * wi);
            gh[i] = make_float2(g32, h32);
            if (qacc) { const unsigned long long m = mult ? mult[i] : 1; qacc[0] += m * fx_coarse(g32, c.fx.c_g); qacc[1] += m * fx_coarse(h32, c.fx.c_h); }
        } else if (OBJ == 2) {   // RegressionL2loss::GetGradients
            const float g32 = (float)((score[i] - y_value[y]) * wi), h32 = (float)wi;
            gh[i] = make_float2(g32, h32);
            if (qacc) { const unsigned long long m = mult ? mult[i] : 1; qacc[0] += m * fx_coarse(g32, c.fx.c_g); qacc[1] += m * fx_coarse(h32, c.fx.c_h); }
        } else {
            const int K = c.K;
            double wmax = score[i];
            for (int k = 1; k < K; ++k) { double s = score[(long long)k * N + i]; if (s > wmax) wmax = s; }
            double wsum = 0.0;
            for (int k = 0; k < K; ++k) wsum += rg_exp(score[(long long)k * N + i] - wmax);
            for (int k = 0; k < K; ++k) {
                double pk = rg_exp(score[(long long)k * N + i] - wmax) / wsum;
                store_gh(gh, (long long)k * c.NG + i, ((y == k) ? (pk - 1.0) : pk) * wi, c.factor * pk * (1.0 - pk) * wi);   // MulticlassSoftmax::GetGradients
            }
        }
    }
}

template <int OBJ>
__global__ __launch_bounds__(256) void k_grad(double* score, const int32_t* __restrict__ ycol,
                                              const double* __restrict__ y_value, const double* __restrict__ class_w,
                                              const double* __restrict__ sample_w, const uint8_t* __restrict__ row_in_bag /* null = no bagging */,
                                              float2* __restrict__ gh, uint8_t* __restrict__ node0 /* level grower: node ids to reset, or null */,
                                              long long NS, unsigned long long* __restrict__ qpart /* OBJ != 1: [gridDim.x][2] coarse sums of this workgroup's (g, h), or null */,
                                              const uint8_t* __restrict__ mult, const PendingScore pd, TrainConst c) {
    unsigned long long acc[2] = {0ull, 0ull};
    const bool measure = OBJ != 1 && qpart != nullptr;
    grad_rows<OBJ>((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, score, ycol, y_value, class_w, sample_w, row_in_bag, gh, node0, NS, c,
                   measure ? acc : nullptr, mult, pd);
    if (measure) {
        __shared__ unsigned long long ws[4][2];
        const unsigned long long a0 = wave_sum_u64(acc[0]), a1 = wave_sum_u64(acc[1]);
        if (lane_id() == 0) { ws[threadIdx.x >> 6][0] = a0; ws[threadIdx.x >> 6][1] = a1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            qpart[(size_t)blockIdx.x * 2] = ws[0][0] + ws[1][0] + ws[2][0] + ws[3][0];
            qpart[(size_t)blockIdx.x * 2 + 1] = ws[0][1] + ws[1][1] + ws[2][1] + ws[3][1];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Numerics v2.2: the fixed-point grid of every class tree of the iteration (rgbm_numerics.h).
//   the gradient kernels leave, per workgroup (or wave), the coarse sums Q_g, Q_h of the (g, h) they wrote: qpart [parts][K][2]
//   k_fx_reduce   column sums of qpart -> Q [K][2]          (k_fx_measure: the same sums straight from the (g, h) array, for the gradient
//                                                            kernels that do not measure -- k_grad<1>, K > 112 -- and as the cross-check)
//   (row-sharded: one integer all-reduce of Q)
//   k_fx_scale    Q -> FxScale [K], and Q back to zero for the next iteration
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fx_reduce(const unsigned long long* __restrict__ qpart, long long nparts, int C2 /* 2 K */, unsigned long long* __restrict__ Q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);     // [C2]
    for (int i = threadIdx.x; i < C2; i += 256) acc[i] = 0ull;
    __syncthreads();
    const long long total = nparts * C2;
    const long long per = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < total ? lo + per : total;
    int col = (int)((lo + threadIdx.x) % C2);
    const int step = 256 % C2;
    for (long long e = lo + threadIdx.x; e < hi; e += 256) {
        const unsigned long long v = qpart[e];
        if (v) atomicAdd(&acc[col], v);                                        // (LDS; the lanes of a wave hold 64 consecutive columns)
        col += step; if (col >= C2) col -= C2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C2; i += 256) { const unsigned long long v = acc[i]; if (v) atomicAdd(&Q[i], v); }
}

// grid (gx, K), block 256
__global__ __launch_bounds__(256) void k_fx_measure(const float2* __restrict__ gh, long long N, long long NG, FxGrid fx, unsigned long long* __restrict__ Q,
                                                    const uint8_t* __restrict__ mult = nullptr) {
    const int k = blockIdx.y;
    const float2* ghk = gh + (long long)k * NG;
    unsigned long long a0 = 0ull, a1 = 0ull;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const float2 g = ghk[i];
        const unsigned long long m = mult ? mult[i] : 1;
        a0 += m * fx_coarse(g.x, fx.c_g); a1 += m * fx_coarse(g.y, fx.c_h);
    }
    __shared__ unsigned long long ws[4][2];
    a0 = wave_sum_u64(a0); a1 = wave_sum_u64(a1);
    if (lane_id() == 0) { ws[threadIdx.x >> 6][0] = a0; ws[threadIdx.x >> 6][1] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long s0 = ws[0][0] + ws[1][0] + ws[2][0] + ws[3][0], s1 = ws[0][1] + ws[1][1] + ws[2][1] + ws[3][1];
        if (s0) atomicAdd(&Q[2 * k], s0);
        if (s1) atomicAdd(&Q[2 * k + 1], s1);
    }
}

__global__ __launch_bounds__(256) void k_fx_scale(unsigned long long* __restrict__ Q, int K, FxGrid fx, FxScale* __restrict__ fxs) {
    for (int k = threadIdx.x; k < K; k += 256) {
        const int e_g = fx_tree_exponent(Q[2 * k], fx.q_mult, fx.c_g, fx.e_g_min, fx.e_g_max);
        const int e_h = fx_tree_exponent(Q[2 * k + 1], fx.q_mult, fx.c_h, fx.e_h_min, fx.e_h_max);
        FxScale f; f.sg = fx_pow2(e_g); f.sh = fx_pow2(e_h); f.inv_sg = fx_pow2(-e_g); f.inv_sh = fx_pow2(-e_h);
        fxs[k] = f;
        Q[2 * k] = 0ull; Q[2 * k + 1] = 0ull;
    }
}

// Multiclass gradients, FP64-VALU bound (one exp, one division and two quantisations per row and class).
// A workgroup of 256 threads owns 64 rows; wave c owns the classes k = c, c+4, c+8, ...  The K x 64 tile of scores sits
// in LDS (every score is read from HBM once, fully coalesced), the maximum is combined across the four waves
// (order-independent), exp(s_k - max) overwrites the tile, ONE wave adds the K terms of every row in class order (the
// summation order of the numerics spec), and every wave finishes its own classes.  Same arithmetic, in the same order,
// as k_grad<1>; 4x the waves per LDS byte of a thread-per-row layout, which is what an FP64-bound kernel needs.
__global__ __launch_bounds__(256) void k_grad_mc(double* score, const int32_t* __restrict__ ycol,
                                                 const double* __restrict__ class_w, const double* __restrict__ sample_w,
                                                 const uint8_t* __restrict__ row_in_bag, float2* __restrict__ gh,
                                                 uint8_t* __restrict__ node0, long long NS,
                                                 unsigned long long* __restrict__ qpart /* [gridDim.x][K][2] coarse sums of this workgroup's (g, h) (numerics v2.2), or null */,
                                                 const uint8_t* __restrict__ mult /* row multiplicities (<= 255: q * m stays below 2^32), or null */,
                                                 const PendingScore pd, TrainConst c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);          // [K][64]
    double* pmax = tile + (size_t)c.K * 64;                  // [4][64]
    double* psum = pmax + 256;                               // [64]
    const long long N = c.N;
    const int K = c.K, r = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + r;
    const bool valid = i < N;
    const long long ic = valid ? i : N - 1;
    double m = -INFINITY;
    const int y = ycol[ic];
    if (pd.node) {   // with the previous iteration's AddScore on the way (this thread also resets the node ids of its (row, class tree) pairs below)
        double* sc = score;
#pragma unroll 4
        for (int k = wv; k < K; k += 4) {
            double v = score[(long long)k * N + ic];
            const int n = pd.node[(long long)k * pd.NS + ic];
            if (n != 255 && pd.L[k] > 1) { v += pd.ndelta[k * 256 + n]; if (valid) sc[(long long)k * N + i] = v; }
            tile[k * 64 + r] = v; if (v > m) m = v;
        }
    } else {
#pragma unroll 4
        for (int k = wv; k < K; k += 4) { const double v = __builtin_nontemporal_load(score + (long long)k * N + ic); tile[k * 64 + r] = v; if (v > m) m = v; }      // (read once per iteration)
    }
    pmax[wv * 64 + r] = m;
    if (node0 && valid) {   // every training row restarts in node 0 (the root); all other rows never take part
        const uint8_t v = (y < 0) ? (uint8_t)255 : (uint8_t)0;
        for (int k = wv; k < K; k += 4) node0[(long long)k * NS + i] = v;
    }
    const bool out_of_bag = row_in_bag && !row_in_bag[ic];
    __syncthreads();
    double wmax = pmax[r];
    { const double b1 = pmax[64 + r], b2 = pmax[128 + r], b3 = pmax[192 + r]; if (b1 > wmax) wmax = b1; if (b2 > wmax) wmax = b2; if (b3 > wmax) wmax = b3; }
    for (int k = wv; k < K; k += 4) tile[k * 64 + r] = rg_exp(tile[k * 64 + r] - wmax);
    __syncthreads();
    if (wv == 0) { double wsum = 0.0; for (int k = 0; k < K; ++k) wsum += tile[k * 64 + r]; psum[r] = wsum; }
    __syncthreads();
    const bool train = valid && y >= 0;     // (not a training row: its gh stays 0 for ever)
    if (!qpart && !train) return;
    if (train && out_of_bag) { for (int k = wv; k < K; k += 4) store_gh(gh, (long long)k * c.NG + i, 0.0, 0.0); }
    const bool on = train && !out_of_bag;
    if (!qpart && !on) return;
    double wi = (on && class_w) ? class_w[y] : 1.0;
    if (on && sample_w) wi = wi * sample_w[i];
    wi = (double)(float)wi;   // LightGBM Metadata keeps weights as float32
    const double wsum = psum[r];
    const unsigned int mrow = (mult && on) ? (unsigned int)mult[i] : 1u;
    for (int k = wv; k < K; k += 4) {
        double packed = 0.0;                                  // bits 0..31: coarse |g|, bits 32..63: coarse h (a double only by type: the tile's)
        if (on) {
            const double pk = tile[k * 64 + r] / wsum;
            const float g32 = (float)(((y == k) ? (pk - 1.0) : pk) * wi), h32 = (float)(c.factor * pk * (1.0 - pk) * wi);
            { typedef float v2f __attribute__((ext_vector_type(2))); v2f gv; gv.x = g32; gv.y = h32; __builtin_nontemporal_store(gv, reinterpret_cast<v2f*>(gh + (long long)k * c.NG + i)); }
            if (qpart) packed = __hiloint2double((int)(mrow * fx_coarse(h32, c.fx.c_h)), (int)(mrow * fx_coarse(g32, c.fx.c_g)));
        }
        // numerics v2.2: the tile entry of (k, r) is dead once pk is known -- it takes the coarse magnitudes of this (row, class tree)
        if (qpart) tile[k * 64 + r] = packed;
    }
    if (!qpart) return;
    __syncthreads();
    // thread t adds up the 64 rows of class t (rotated start: the 64 lanes of a wave read 64 different LDS banks pairs)
    for (int t = threadIdx.x; t < K; t += 256) {
        unsigned long long s0 = 0ull, s1 = 0ull;
#pragma unroll 8
        for (int j = 0; j < 64; ++j) { const double p = tile[t * 64 + ((j + t) & 63)]; s0 += (unsigned int)__double2loint(p); s1 += (unsigned int)__double2hiint(p); }
        unsigned long long* dst = qpart + ((size_t)blockIdx.x * K + t) * 2;
        dst[0] = s0; dst[1] = s1;
    }
}

// Few classes (K < 16): one thread per row, the row's K scores parked in its own LDS column (no barrier); the
// four-waves-per-row-block layout above would leave most of its waves idle.
template <int R>
__global__ __launch_bounds__(R) void k_grad_mc_rows(double* score, const int32_t* __restrict__ ycol,
                                                    const double* __restrict__ class_w, const double* __restrict__ sample_w,
                                                    const uint8_t* __restrict__ row_in_bag, float2* __restrict__ gh,
                                                    uint8_t* __restrict__ node0, long long NS,
                                                    unsigned long long* __restrict__ qpart /* [gridDim.x * R / 64][K][2] coarse sums of every wave's (g, h) (numerics v2.2), or null */,
                                                    const uint8_t* __restrict__ mult, const PendingScore pd, TrainConst c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile0 = reinterpret_cast<double*>(smem);
    double* tile = tile0 + threadIdx.x;   // element k at tile[k * R]
    const long long N = c.N;
    const int K = c.K;
    const long long i = (long long)blockIdx.x * R + threadIdx.x;
    const bool valid = i < N;
    if (!qpart && !valid) return;
    const long long ic = valid ? i : N - 1;
    const double* sp = score + ic;
    double wmax = -INFINITY;
    if (pd.node) {   // with the previous iteration's AddScore on the way
        double* sc = score + ic;
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            double v = sp[(long long)k * N];
            const int n = pd.node[(long long)k * pd.NS + ic];
            if (n != 255 && pd.L[k] > 1) { v += pd.ndelta[k * 256 + n]; if (valid) sc[(long long)k * N] = v; }
            tile[k * R] = v; if (v > wmax) wmax = v;
        }
    } else {
#pragma unroll 4
        for (int k = 0; k < K; ++k) { const double v = __builtin_nontemporal_load(sp + (long long)k * N); tile[k * R] = v; if (v > wmax) wmax = v; }
    }
    const int y = ycol[ic];
    if (node0 && valid) {
        const uint8_t v = (y < 0) ? (uint8_t)255 : (uint8_t)0;
        for (int kk = 0; kk < K; ++kk) node0[(long long)kk * NS + i] = v;
    }
    const bool train = valid && y >= 0;
    if (!qpart && !train) return;
    const bool oob = train && row_in_bag && !row_in_bag[i];
    if (oob) { for (int kk = 0; kk < K; ++kk) store_gh(gh, (long long)kk * c.NG + i, 0.0, 0.0); }
    const bool on = train && !oob;
    if (!qpart && !on) return;
    if (on) {
        double wi = class_w ? class_w[y] : 1.0;
        if (sample_w) wi = wi * sample_w[i];
        wi = (double)(float)wi;
        double wsum = 0.0;
        const unsigned int mrow = mult ? (unsigned int)mult[i] : 1u;
        for (int kk = 0; kk < K; ++kk) { const double e = rg_exp(tile[kk * R] - wmax); tile[kk * R] = e; wsum += e; }
        for (int kk = 0; kk < K; ++kk) {
            const double pk = tile[kk * R] / wsum;
            const float g32 = (float)(((y == kk) ? (pk - 1.0) : pk) * wi), h32 = (float)(c.factor * pk * (1.0 - pk) * wi);
            { typedef float v2f __attribute__((ext_vector_type(2))); v2f gv; gv.x = g32; gv.y = h32; __builtin_nontemporal_store(gv, reinterpret_cast<v2f*>(gh + (long long)kk * c.NG + i)); }
            // numerics v2.2: the dead tile entry takes the coarse magnitudes (bits 0..31: |g|, bits 32..63: h)
            if (qpart) tile[kk * R] = __hiloint2double((int)(mrow * fx_coarse(h32, c.fx.c_h)), (int)(mrow * fx_coarse(g32, c.fx.c_g)));
        }
    } else if (qpart) { for (int kk = 0; kk < K; ++kk) tile[kk * R] = 0.0; }
    if (!qpart) return;
    __syncthreads();
    // lane t < K of every wave adds up the wave's 64 rows of class t (rotated start: different banks)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < K) {
        unsigned long long s0 = 0ull, s1 = 0ull;
#pragma unroll 8
        for (int j = 0; j < 64; ++j) { const double p = tile0[lane * R + wv * 64 + ((j + lane) & 63)]; s0 += (unsigned int)__double2loint(p); s1 += (unsigned int)__double2hiint(p); }
        unsigned long long* dst = qpart + (((size_t)blockIdx.x * (R / 64) + wv) * K + lane) * 2;
        dst[0] = s0; dst[1] = s1;
    }
}

// ------------------------------------------------------------------------------------------------
// K3: hist_build of the LEAF-WISE grower (the fallback path; the level grower's passes are in rgbm_level.h).
//   grid (gx, K, nchunk), block 256.  Each lane owns one row per step: one dwordx4 load brings its 16 bin codes, one dwordx2
//   load its float32 (g,h), which go onto the model's fixed-point grid (fx_from_f32) and into the per-workgroup LDS histogram
//   with two 64-bit atomics per feature.  Low-cardinality features are replicated `rep` times (slot = bin*rep + lane%rep) so
//   that lanes hitting the same bin do not serialise on one LDS address.  The workgroup's histogram is flushed to the leaf's
//   global histogram with 64-bit atomics once.  All sums are integers, so the result is independent of scheduling.
//   Algorithmic bytes per scanned row: 16 B/chunk of bins (F useful) + 8 B (g,h) (+4 B row index off the root).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hist(const uint4* __restrict__ rec, const float2* __restrict__ gh,
                                              const int32_t* __restrict__ idx0, const int32_t* __restrict__ idx1,
                                              const int32_t* __restrict__ base_idx, const TreeState* __restrict__ state,
                                              HistBin* __restrict__ pool, const FeatMeta* __restrict__ fmeta,
                                              const ChunkMeta* __restrict__ cmeta, const FxScale* __restrict__ fxs, TrainConst c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int k = blockIdx.y, ch = blockIdx.z;
    const TreeState st = state[k];
    if (!st.do_hist) return;
    const double sg_k = fxs[k].sg, sh_k = fxs[k].sh;          // this class tree's grid in this iteration (numerics v2.2)
    const ChunkMeta cm = cmeta[ch];
    unsigned long long* fast_g = reinterpret_cast<unsigned long long*>(smem);   // [fast_slots] replicated gradient sums, then [fast_slots] hessian sums
    unsigned long long* fast_h = fast_g + cm.fast_slots;                        // (separate arrays: a wave instruction of atomics spreads over all LDS banks)
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * cm.fast_slots; i += 256) fast_g[i] = 0ull;
    __syncthreads();

    // per-feature slot bases / replication shifts of this chunk (uniform -> scalar registers)
    int fbase[16], fshift[16];
    const FeatMeta* fm = fmeta + cm.first_feat;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < cm.nfeat) { fbase[j] = fm[j].fast_base; fshift[j] = fm[j].rep_shift; } else { fbase[j] = 0; fshift[j] = 0; }
    }
    const long long N = c.N;
    const long long cnt = st.hist_is_root ? N : (long long)st.hist_count;
    const int32_t* idx = st.hist_buf == 0 ? idx0 + (long long)k * c.n_train : (st.hist_buf == 1 ? idx1 + (long long)k * c.n_train : base_idx);
    const uint4* recc = rec + (long long)ch * N;
    const float2* ghk = gh + (long long)k * c.NG;
    const long long ntiles = (cnt + TILE_ROWS - 1) / TILE_ROWS;

    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long p0 = t * TILE_ROWS;
#pragma unroll 2
        for (int s = 0; s < TILE_ROWS / 256; ++s) {
            long long p = p0 + s * 256 + tid;
            if (p < cnt) {
                long long row = st.hist_is_root ? p : (long long)idx[st.hist_begin + p];
                uint4 r = recc[row];
                const float2 g = ghk[row];
                if (g.x != 0.0f || g.y != 0.0f) {   // non-training / out-of-bag rows carry (0, 0)
                    const unsigned long long gq = (unsigned long long)fx_from_f32(g.x, sg_k), hq = (unsigned long long)fx_from_f32(g.y, sh_k);
                    uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (j < cm.nfeat) {
                            uint32_t bin = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                            int slot = fbase[j] + (int)(bin << fshift[j]) + (lane & ((1 << fshift[j]) - 1));
                            atomicAdd(&fast_g[slot], gq);
                            atomicAdd(&fast_h[slot], hq);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // flush to the leaf histogram being built (always the slot of the newest leaf; root: slot 0)
    const int slot_leaf = st.hist_is_root ? 0 : st.right_leaf;
    HistBin* dst = pool + ((long long)k * c.num_leaves + slot_leaf) * c.totbins;
    for (int j = 0; j < cm.nfeat; ++j) {
        const int sh = fm[j].rep_shift;
        for (int b = tid; b < fm[j].nbins; b += 256) {
            long long tg = 0, th = 0;
            const int s0 = fm[j].fast_base + (b << sh);
            for (int r2 = 0; r2 < (1 << sh); ++r2) { tg += (long long)fast_g[s0 + r2]; th += (long long)fast_h[s0 + r2]; }
            HistBin* d = &dst[fm[j].hoff + b];
            if (tg) atomicAdd(reinterpret_cast<unsigned long long*>(&d->g), (unsigned long long)tg);
            if (th) atomicAdd(reinterpret_cast<unsigned long long*>(&d->h), (unsigned long long)th);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4+K5: histogram subtraction + per-feature threshold scan.  One wave per (class tree, feature):
// lane l owns bins 4l..4l+3, integer prefix sums by wave scan, gains in double (rgbm_numerics.h),
// wave arg-max with LightGBM's visiting-order tie-breaks.  grid (ceil(F/4), K), block 256.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long wave_incl_scan(long long v) {
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        long long o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

struct ScanBest { double gain; int theta; long long lg, lh; };

// a beats b?  reverse scan: larger theta first on ties; forward scan: smaller theta first
template <bool REVERSE>
__device__ __forceinline__ bool scan_better(double ga, int ta, double gb, int tb) {
    if (ga != gb) return ga > gb;
    return REVERSE ? (ta > tb) : (ta < tb);
}

__device__ void scan_child(const long long (&bg)[4], const long long (&bh)[4], const FeatMeta& fm, long long Gq, long long Hq,
                           long long num_data, const TrainConst& c, Cand* out) {
    const int lane = lane_id();
    const int V = fm.V;
    const double keps = k_eps();
    const double sum_gradient = (double)Gq * c.inv_sg;
    const double sum_hessian = (double)Hq * c.inv_sh + 2 * keps;
    const double gain_shift = leaf_gain(sum_gradient, sum_hessian, c.l1, c.l2);
    const double min_gain_shift = gain_shift + c.min_gain_to_split;
    const double cnt_factor = (double)num_data / sum_hessian;
    const bool two_way = fm.has_nan && V >= 1;

    // per-lane local values over VALUE bins only (NaN bin excluded from the scans)
    long long lg[4], lh[4], lc[4];
    long long tg = 0, th = 0, tc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int b = lane * 4 + j;
        bool val = b < V;
        lg[j] = val ? bg[j] : 0; lh[j] = val ? bh[j] : 0;
        lc[j] = val ? round_int((double)bh[j] * c.inv_sh * cnt_factor) : 0;
        tg += lg[j]; th += lh[j]; tc += lc[j];
    }
    long long ig = wave_incl_scan(tg), ih = wave_incl_scan(th), ic = wave_incl_scan(tc);
    const long long TG = __shfl(ig, 63), TH = __shfl(ih, 63), TC = __shfl(ic, 63);
    long long pg = ig - tg, ph = ih - th, pc = ic - tc;   // exclusive prefix at the lane's first bin

    ScanBest rv = {-INFINITY, 0, 0, 0}, fw = {-INFINITY, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int b = lane * 4 + j;
        if (b < V) {
            // REVERSE candidate theta = b-1: right = value bins >= b (NaN rows stay left)
            {
                long long rg = TG - pg, rh = TH - ph, right_count = TC - pc;
                double sum_right_hessian = (double)rh * c.inv_sh + keps;
                long long left_count = num_data - right_count;
                long long lhq = Hq - rh, lgq = Gq - rg;
                double sum_left_hessian = (double)lhq * c.inv_sh + keps;
                bool ok = !(right_count < c.min_data_in_leaf || sum_right_hessian < c.min_sum_hessian) &&
                          !(left_count < c.min_data_in_leaf) && !(sum_left_hessian < c.min_sum_hessian);
                if (ok) {
                    double cur = leaf_gain((double)lgq * c.inv_sg, sum_left_hessian, c.l1, c.l2) +
                                 leaf_gain((double)rg * c.inv_sg, sum_right_hessian, c.l1, c.l2);
                    if (cur > min_gain_shift && scan_better<true>(cur, b - 1, rv.gain, rv.theta)) { rv.gain = cur; rv.theta = b - 1; rv.lg = lgq; rv.lh = lhq; }
                }
            }
            pg += lg[j]; ph += lh[j]; pc += lc[j];   // now inclusive through b
            if (two_way) {
                // FORWARD candidate theta = b: left = value bins <= b (NaN rows go right)
                long long lgq = pg, lhq = ph, left_count = pc;
                double sum_left_hessian = (double)lhq * c.inv_sh + keps;
                long long right_count = num_data - left_count;
                long long rh = Hq - lhq, rg = Gq - lgq;
                double sum_right_hessian = (double)rh * c.inv_sh + keps;
                bool ok = !(left_count < c.min_data_in_leaf || sum_left_hessian < c.min_sum_hessian) &&
                          !(right_count < c.min_data_in_leaf) && !(sum_right_hessian < c.min_sum_hessian);
                if (ok) {
                    double cur = leaf_gain((double)lgq * c.inv_sg, sum_left_hessian, c.l1, c.l2) +
                                 leaf_gain((double)rg * c.inv_sg, sum_right_hessian, c.l1, c.l2);
                    if (cur > min_gain_shift && scan_better<false>(cur, b, fw.gain, fw.theta)) { fw.gain = cur; fw.theta = b; fw.lg = lgq; fw.lh = lhq; }
                }
            }
        }
    }
    // wave arg-max (butterfly: every lane ends with the winner)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double g2 = __shfl_xor(rv.gain, off); int t2 = __shfl_xor(rv.theta, off);
        long long a2 = __shfl_xor(rv.lg, off), b2 = __shfl_xor(rv.lh, off);
        if (scan_better<true>(g2, t2, rv.gain, rv.theta)) { rv.gain = g2; rv.theta = t2; rv.lg = a2; rv.lh = b2; }
        g2 = __shfl_xor(fw.gain, off); t2 = __shfl_xor(fw.theta, off);
        a2 = __shfl_xor(fw.lg, off); b2 = __shfl_xor(fw.lh, off);
        if (scan_better<false>(g2, t2, fw.gain, fw.theta)) { fw.gain = g2; fw.theta = t2; fw.lg = a2; fw.lh = b2; }
    }
    if (lane == 0) {
        Cand o; o.gain = -INFINITY; o.theta = 0; o.dleft = 1; o.left_gq = 0; o.left_hq = 0; o.left_out = 0.0; o.right_out = 0.0;
        if (rv.gain > -INFINITY && rv.gain > o.gain + min_gain_shift) {
            o.theta = rv.theta; o.dleft = 1; o.left_gq = rv.lg; o.left_hq = rv.lh;
            double lH = (double)rv.lh * c.inv_sh + keps, rH = (double)(Hq - rv.lh) * c.inv_sh + keps;
            o.left_out = leaf_output((double)rv.lg * c.inv_sg, lH, c.l1, c.l2);
            o.right_out = leaf_output((double)(Gq - rv.lg) * c.inv_sg, rH, c.l1, c.l2);
            o.gain = rv.gain - min_gain_shift;
        }
        if (fw.gain > -INFINITY && fw.gain > o.gain + min_gain_shift) {
            o.theta = fw.theta; o.dleft = 0; o.left_gq = fw.lg; o.left_hq = fw.lh;
            double lH = (double)fw.lh * c.inv_sh + keps, rH = (double)(Hq - fw.lh) * c.inv_sh + keps;
            o.left_out = leaf_output((double)fw.lg * c.inv_sg, lH, c.l1, c.l2);
            o.right_out = leaf_output((double)(Gq - fw.lg) * c.inv_sg, rH, c.l1, c.l2);
            o.gain = fw.gain - min_gain_shift;
        }
        *out = o;
    }
}

// one wave: feature f of the leaf / the two leaves searched in this step (pk = the class tree's histogram pool, lk its leaves,
// used_k its feature mask, ck its [2][F] candidates; lk / ck may live in LDS)
__device__ __forceinline__ void split_find_body(HistBin* __restrict__ pk, const TreeState& st, Leaf* lk, const FeatMeta* __restrict__ fmeta,
                                                const uint8_t* __restrict__ used_k, Cand* ck, int f, const TrainConst& c) {
    const int lane = lane_id();
    const FeatMeta fm = fmeta[f];
    long long ag[4], ah[4], bgv[4], bhv[4];
    if (st.hist_is_root) {
        const HistBin* h0 = pk + fm.hoff;
        long long sg_ = 0, sh_ = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int b = lane * 4 + j;
            if (b < fm.nbins) { HistBin v = h0[b]; ag[j] = v.g; ah[j] = v.h; } else { ag[j] = 0; ah[j] = 0; }
            sg_ += ag[j]; sh_ += ah[j];
        }
        // leaf totals = sum over all bins of any one feature (every row sits in exactly one bin)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { sg_ += __shfl_xor(sg_, off); sh_ += __shfl_xor(sh_, off); }
        if (f == 0 && lane == 0) { lk[0].Gq = sg_; lk[0].Hq = sh_; }
        if (!used_k[f]) { if (lane == 0) ck[f].gain = -INFINITY; return; }
        scan_child(ag, ah, fm, sg_, sh_, (long long)lk[0].count, c, &ck[f]);
        return;
    }
    // children of the last split: parent histogram lives in slot[left leaf], the freshly built
    // smaller child in slot[right leaf]; rewrite both slots to hold (left, right).
    HistBin* hl = pk + (long long)st.split_leaf * c.totbins + fm.hoff;
    HistBin* hr = pk + (long long)st.right_leaf * c.totbins + fm.hoff;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int b = lane * 4 + j;
        if (b < fm.nbins) {
            HistBin par = hl[b], built = hr[b];
            HistBin l, r;
            if (st.smaller_is_left) { l = built; r.g = par.g - built.g; r.h = par.h - built.h; }
            else { r = built; l.g = par.g - built.g; l.h = par.h - built.h; }
            hl[b] = l; hr[b] = r;
            ag[j] = l.g; ah[j] = l.h; bgv[j] = r.g; bhv[j] = r.h;
        } else { ag[j] = ah[j] = bgv[j] = bhv[j] = 0; }
    }
    if (!used_k[f]) { if (lane == 0) { ck[f].gain = -INFINITY; ck[c.F + f].gain = -INFINITY; } return; }
    const Leaf L = lk[st.split_leaf], R = lk[st.right_leaf];
    scan_child(ag, ah, fm, L.Gq, L.Hq, (long long)L.count, c, &ck[f]);
    scan_child(bgv, bhv, fm, R.Gq, R.Hq, (long long)R.count, c, &ck[c.F + f]);
}

__global__ __launch_bounds__(256) void k_split_find(HistBin* __restrict__ pool, const TreeState* __restrict__ state,
                                                    Leaf* __restrict__ leaves, const FeatMeta* __restrict__ fmeta,
                                                    const uint8_t* __restrict__ used /* [K][F] */, Cand* __restrict__ cand /* [K][2][F] */,
                                                    const FxScale* __restrict__ fxs, TrainConst c_model) {
    const int k = blockIdx.y;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= c_model.F) return;
    const TreeState st = state[k];
    if (!st.do_hist) return;
    const TrainConst c = tree_const(c_model, fxs, k);
    split_find_body(pool + (long long)k * c.num_leaves * c.totbins, st, leaves + (long long)k * c.num_leaves, fmeta, used + (long long)k * c.F,
                    cand + (long long)k * 2 * c.F, f, c);
}

// ------------------------------------------------------------------------------------------------
// k_tree_step: one wave per class tree.  (1) reduce the per-feature candidates of the freshly
// searched leaves (SplitInfo::operator>: gain, then smaller feature); (2) pick the best leaf
// (ArrayArgs::ArgMax); (3) Tree::Split bookkeeping and partition request.
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ bool leaf_better(double ga, int fa, int la, double gb, int fb, int lb) {
    if (ga != gb) return ga > gb;
    int a = fa < 0 ? 0x7FFFFFFF : fa, b = fb < 0 ? 0x7FFFFFFF : fb;
    if (a != b) return a < b;
    return la < lb;
}

__device__ void reduce_leaf_best(const Cand* cf, int F, Leaf* leaf) {
    const int lane = lane_id();
    double bg = -INFINITY; int bf = -1;
    for (int f = lane; f < F; f += 64) {
        double g = cf[f].gain;
        if (g > -INFINITY && leaf_better(g, f, 0, bg, bf, 0)) { bg = g; bf = f; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double g2 = __shfl_xor(bg, off); int f2 = __shfl_xor(bf, off);
        if (leaf_better(g2, f2, 0, bg, bf, 0)) { bg = g2; bf = f2; }
    }
    if (lane == 0) {
        if (bf >= 0) { leaf->best = cf[bf]; leaf->best_feature = bf; }
        else { leaf->best.gain = -INFINITY; leaf->best_feature = -1; }
    }
}

// Steps (2) and (3) for one class tree, run by ONE wave after the candidates of the freshly searched leaves are reduced.
// `st` / `lk` may live in global memory (k_tree_step) or in LDS (k_small_tree).  ZERO_SLOT: clear the histogram slot the next
// k_hist launch accumulates into with atomics (the fused small-table grower stores its histograms directly).
template <bool ZERO_SLOT>
__device__ __forceinline__ void tree_step_pick(TreeState* st, Leaf* lk, HistBin* pool_k, const FeatMeta* __restrict__ fmeta, TreeOut out, long long tbase, const TrainConst& c) {
    const int lane = lane_id();
    const int L = st->L;
    double bg = -INFINITY; int bf = -1, bl = 0x7FFFFFFF;
    for (int l = lane; l < L; l += 64) {
        double g = lk[l].best.gain; int f = lk[l].best_feature;
        if (bl == 0x7FFFFFFF || leaf_better(g, f, l, bg, bf, bl)) { bg = g; bf = f; bl = l; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double g2 = __shfl_xor(bg, off); int f2 = __shfl_xor(bf, off), l2 = __shfl_xor(bl, off);
        if (l2 != 0x7FFFFFFF && (bl == 0x7FFFFFFF || leaf_better(g2, f2, l2, bg, bf, bl))) { bg = g2; bf = f2; bl = l2; }
    }
    if (L >= c.num_leaves || !(bg > 0.0)) {
        if (lane == 0) { st->done = 1; st->do_partition = 0; st->do_hist = 0; out.L[tbase] = L; }
        return;
    }
    const int right_leaf = L, node = L - 1;
    if (ZERO_SLOT) {   // zero the histogram slot the next hist pass will accumulate into
        HistBin* zs = pool_k + (long long)right_leaf * c.totbins;
        for (int i = lane; i < c.totbins; i += 64) { zs[i].g = 0; zs[i].h = 0; }
    }
    if (lane == 0) {
        Leaf P = lk[bl];
        const Cand sp = P.best;
        const long long nb = tbase * (c.num_leaves - 1);
        out.feat[nb + node] = bf; out.theta[nb + node] = sp.theta; out.dleft[nb + node] = sp.dleft; out.gain[nb + node] = sp.gain;
        out.left[nb + node] = ~bl; out.right[nb + node] = ~right_leaf;
        if (P.parent_node >= 0) { if (P.is_left) out.left[nb + P.parent_node] = node; else out.right[nb + P.parent_node] = node; }
        out.leaf_value[tbase * c.num_leaves + bl] = sp.left_out;
        out.leaf_value[tbase * c.num_leaves + right_leaf] = sp.right_out;
        Leaf Lf = P, Rf = P;
        Lf.parent_node = node; Lf.is_left = 1; Lf.depth = P.depth + 1; Lf.Gq = sp.left_gq; Lf.Hq = sp.left_hq;
        Lf.best.gain = -INFINITY; Lf.best_feature = -1;
        Rf.parent_node = node; Rf.is_left = 0; Rf.depth = P.depth + 1; Rf.Gq = P.Gq - sp.left_gq; Rf.Hq = P.Hq - sp.left_hq;
        Rf.best.gain = -INFINITY; Rf.best_feature = -1;
        lk[bl] = Lf; lk[right_leaf] = Rf;   // begin/count/buf are completed by finish_split
        st->split_leaf = bl; st->right_leaf = right_leaf; st->L = L + 1;
        st->do_partition = 1; st->do_hist = 0;
        st->part_feature = bf; st->part_theta = sp.theta; st->part_dleft = sp.dleft;
        st->part_nanbin = fmeta[bf].has_nan ? fmeta[bf].V : 255;
        st->part_begin = P.begin; st->part_count = P.count; st->part_buf = P.buf;
        st->cursor_left = 0; st->cursor_right = 0;
        out.L[tbase] = L + 1;
    }
}

__global__ __launch_bounds__(64) void k_tree_step(TreeState* __restrict__ state, Leaf* __restrict__ leaves, const Cand* __restrict__ cand,
                                                  HistBin* __restrict__ pool, const FeatMeta* __restrict__ fmeta, TreeOut out, int it, TrainConst c) {
    const int k = blockIdx.x;
    TreeState* st = &state[k];
    if (st->done) return;
    Leaf* lk = leaves + (long long)k * c.num_leaves;
    const Cand* ck = cand + (long long)k * 2 * c.F;
    if (st->do_hist) {
        if (st->hist_is_root) reduce_leaf_best(ck, c.F, &lk[0]);
        else { reduce_leaf_best(ck, c.F, &lk[st->split_leaf]); reduce_leaf_best(ck + c.F, c.F, &lk[st->right_leaf]); }
    }
    __syncthreads();
    tree_step_pick<true>(st, lk, pool + (long long)k * c.num_leaves * c.totbins, fmeta, out, (long long)it * c.K + k, c);
}

// ------------------------------------------------------------------------------------------------
// K6: partition.  Unstable (order inside a leaf is irrelevant to integer sums): every workgroup
// reserves output ranges from two global cursors; lefts fill from the front, rights from the back
// of the leaf's range in the other ping-pong buffer.  grid (gx, K), block 256, 4 rows per thread.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_partition(const uint8_t* __restrict__ rec8, int32_t* __restrict__ idx0, int32_t* __restrict__ idx1,
                                                   const int32_t* __restrict__ base_idx, TreeState* __restrict__ state, TrainConst c) {
    __shared__ int wl[4], wr[4];
    __shared__ unsigned int bases[2];
    const int k = blockIdx.y;
    TreeState* st = &state[k];
    if (!st->do_partition) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cnt = st->part_count, beg = st->part_begin, pb = st->part_buf;
    const int32_t* src = pb == 0 ? idx0 + (long long)k * c.n_train : (pb == 1 ? idx1 + (long long)k * c.n_train : base_idx);
    int32_t* dst = (pb == 0 ? idx1 : idx0) + (long long)k * c.n_train;   // base list (2) -> buffer 0
    const int f = st->part_feature, theta = st->part_theta, dleft = st->part_dleft, nanbin = st->part_nanbin;
    const uint8_t* recf = rec8 + ((long long)(f >> 4) * c.N) * 16 + (f & 15);
    const int ntiles = (cnt + 1023) / 1024;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int rows[4]; bool gl[4]; bool on[4];
        int nl = 0, nr = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int p = t * 1024 + j * 256 + tid;
            on[j] = p < cnt; gl[j] = false; rows[j] = 0;
            if (on[j]) {
                rows[j] = src[beg + p];
                int bin = recf[(long long)rows[j] * 16];
                gl[j] = (bin == nanbin) ? (dleft != 0) : (bin <= theta);
                if (gl[j]) ++nl; else ++nr;
            }
        }
        // exclusive scans of (nl, nr) across the workgroup
        int sl = nl, sr = nr;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { int a = __shfl_up(sl, off), b = __shfl_up(sr, off); if (lane >= off) { sl += a; sr += b; } }
        if (lane == 63) { wl[wv] = sl; wr[wv] = sr; }
        __syncthreads();
        int ol = sl - nl, orr = sr - nr, totl = 0, totr = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wv) { ol += wl[w]; orr += wr[w]; } totl += wl[w]; totr += wr[w]; }
        if (tid == 0) { bases[0] = atomicAdd(&st->cursor_left, (unsigned)totl); bases[1] = atomicAdd(&st->cursor_right, (unsigned)totr); }
        __syncthreads();
        const int bl_ = (int)bases[0], br_ = (int)bases[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (on[j]) {
                if (gl[j]) dst[beg + bl_ + ol++] = rows[j];
                else dst[beg + cnt - 1 - (br_ + orr++)] = rows[j];
            }
        }
        __syncthreads();
    }
}

// complete the children's ranges and decide the next histogram pass (one thread; `nl` = rows that went left)
__device__ __forceinline__ void finish_split_body(TreeState* st, Leaf* lk, TreeOut out, long long tbase, int nl, const TrainConst& c) {
    if (!st->do_partition) { st->do_hist = 0; return; }
    const int nr = st->part_count - nl;
    const int nbuf = st->part_buf == 0 ? 1 : 0;
    Leaf& Lf = lk[st->split_leaf]; Leaf& Rf = lk[st->right_leaf];
    Lf.begin = st->part_begin; Lf.count = nl; Lf.buf = nbuf;
    Rf.begin = st->part_begin + nl; Rf.count = nr; Rf.buf = nbuf;
    out.leaf_count[tbase * c.num_leaves + st->split_leaf] = nl;
    out.leaf_count[tbase * c.num_leaves + st->right_leaf] = nr;
    st->do_partition = 0;
    bool go = st->L < c.num_leaves;
    if (c.max_depth > 0 && Lf.depth >= c.max_depth) go = false;
    if (nr < c.min_data_in_leaf * 2 && nl < c.min_data_in_leaf * 2) go = false;
    st->do_hist = go ? 1 : 0;
    st->hist_is_root = 0;
    st->smaller_is_left = nl < nr ? 1 : 0;
    st->hist_begin = st->smaller_is_left ? Lf.begin : Rf.begin;
    st->hist_count = st->smaller_is_left ? nl : nr;
    st->hist_buf = nbuf;
}

// one wave per class tree
__global__ __launch_bounds__(64) void k_finish_split(TreeState* __restrict__ state, Leaf* __restrict__ leaves, TreeOut out, int it, TrainConst c) {
    const int k = blockIdx.x;
    if (threadIdx.x != 0) return;
    TreeState* st = &state[k];
    finish_split_body(st, leaves + (long long)k * c.num_leaves, out, (long long)it * c.K + k, (int)st->cursor_left, c);
}

__global__ __launch_bounds__(64) void k_init_iter(TreeState* __restrict__ state, Leaf* __restrict__ leaves, HistBin* __restrict__ pool,
                                                  TreeOut out, const unsigned int* __restrict__ n_in_ptr, int it, TrainConst c) {
    const int k = blockIdx.x, lane = lane_id();
    const long long n_in = n_in_ptr ? (long long)n_in_ptr[0] : c.n_train;
    HistBin* z = pool + (long long)k * c.num_leaves * c.totbins;
    for (int i = lane; i < c.totbins; i += 64) { z[i].g = 0; z[i].h = 0; }
    if (lane == 0) {
        TreeState s; memset(&s, 0, sizeof(s));
        s.L = 1; s.done = 0; s.hist_is_root = 1; s.hist_begin = 0; s.hist_count = (int)n_in; s.hist_buf = 2;
        s.do_hist = (n_in < (long long)c.min_data_in_leaf * 2) ? 0 : 1;
        state[k] = s;
        Leaf r; memset(&r, 0, sizeof(r));
        r.begin = 0; r.count = (int)n_in; r.buf = 2; r.depth = 0; r.parent_node = -1; r.is_left = 0;
        r.best.gain = -INFINITY; r.best_feature = -1;
        leaves[(long long)k * c.num_leaves] = r;
        const long long tbase = (long long)it * c.K + k;
        out.L[tbase] = 1;
        out.leaf_count[tbase * c.num_leaves] = (int)n_in;
    }
}

// ------------------------------------------------------------------------------------------------
// K7: shrinkage + score update.  k_finalize_tree sorts the (<= num_leaves) leaf ranges; the update
// kernel finds each position's leaf by binary search in LDS.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_finalize_tree(const TreeState* __restrict__ state, const Leaf* __restrict__ leaves, TreeOut out,
                                                      const double* __restrict__ init, double* __restrict__ upd_value /* [K][NL] */,
                                                      int32_t* __restrict__ sorted /* [K][NL][3] begin,count,leaf */, int32_t* __restrict__ any_split,
                                                      int it, TrainConst c) {
    const int k = blockIdx.x, lane = lane_id();
    const int L = state[k].L;
    const Leaf* lk = leaves + (long long)k * c.num_leaves;
    const long long tbase = (long long)it * c.K + k;
    double* lv = out.leaf_value + tbase * c.num_leaves;
    double* uv = upd_value + (long long)k * c.num_leaves;
    int32_t* so = sorted + (long long)k * c.num_leaves * 3;
    if (L <= 1) {
        if (lane == 0) { lv[0] = (it == 0) ? init[k] : 0.0; uv[0] = 0.0; so[0] = 0; so[1] = 0; so[2] = 0; }
        return;
    }
    if (lane == 0) atomicOr(any_split + it, 1);
    for (int l = lane; l < L; l += 64) {
        double v = lv[l] * c.learning_rate;    // Tree::Shrinkage
        uv[l] = v;
        if (it == 0 && fabs(init[k]) > k_eps()) v += init[k];   // Tree::AddBias (model only; scores already hold init)
        lv[l] = v;
        int rank = 0;
        const int b = lk[l].begin, n = lk[l].count;
        for (int m = 0; m < L; ++m) {
            const int b2 = lk[m].begin, n2 = lk[m].count;
            if (b2 < b || (b2 == b && (n2 < n || (n2 == n && m < l)))) ++rank;
        }
        so[rank * 3] = b; so[rank * 3 + 1] = n; so[rank * 3 + 2] = l;
    }
}

__global__ __launch_bounds__(256) void k_score_update(const TreeState* __restrict__ state, const Leaf* __restrict__ leaves,
                                                      const int32_t* __restrict__ sorted, const double* __restrict__ upd_value,
                                                      const int32_t* __restrict__ idx0, const int32_t* __restrict__ idx1,
                                                      const int32_t* __restrict__ base_idx, double* __restrict__ score, const unsigned int* __restrict__ n_in_ptr, TrainConst c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int k = blockIdx.y;
    const int L = state[k].L;
    if (L <= 1) return;
    int32_t* sb = reinterpret_cast<int32_t*>(smem);        // [L] begin
    int32_t* sl = sb + c.num_leaves;                       // [L] leaf
    int32_t* sbuf = sl + c.num_leaves;                     // [L] buf
    double* sv = reinterpret_cast<double*>(sbuf + c.num_leaves + (c.num_leaves & 1));
    const int32_t* so = sorted + (long long)k * c.num_leaves * 3;
    const Leaf* lk = leaves + (long long)k * c.num_leaves;
    for (int i = threadIdx.x; i < L; i += 256) {
        int leaf = so[i * 3 + 2];
        sb[i] = so[i * 3]; sl[i] = leaf; sbuf[i] = lk[leaf].buf; sv[i] = upd_value[(long long)k * c.num_leaves + leaf];
    }
    __syncthreads();
    double* sk = score + (long long)k * c.N;
    const long long n_in = n_in_ptr ? (long long)n_in_ptr[0] : c.n_train;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n_in; p += (long long)gridDim.x * 256) {
        int lo = 0, hi = L - 1;    // last i with sb[i] <= p
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (sb[mid] <= (int)p) lo = mid; else hi = mid - 1; }
        const int bf = sbuf[lo];
        const int32_t* src = bf == 0 ? idx0 + (long long)k * c.n_train : (bf == 1 ? idx1 + (long long)k * c.n_train : base_idx);
        const int row = src[p];
        sk[row] += sv[lo];
    }
}


// ------------------------------------------------------------------------------------------------
// K8: bagging (GBDT::Bagging / BaggingHelper).  LightGBM draws one LCG float per training row in
// blocks of 1024 rows, each block owning its own generator seeded bagging_seed + block; the
// generator state survives between bagging events.  Positions are ranks among the training rows in
// ascending row order, so a stable compaction of the training rows comes first.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_block_count(const int32_t* __restrict__ ycol, long long N, unsigned int* __restrict__ blk_cnt) {
    // one workgroup per 1024 table rows
    __shared__ unsigned int ws[4];
    const long long base = (long long)blockIdx.x * 1024;
    unsigned int c = 0;
    for (int j = 0; j < 4; ++j) { long long r = base + j * 256 + threadIdx.x; if (r < N && ycol[r] >= 0) ++c; }
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(1024) void k_block_scan(unsigned int* __restrict__ blk_cnt, long long nblk) {
    // single workgroup: in-place exclusive scan of the per-block counts
    __shared__ unsigned int part[1024];
    __shared__ unsigned int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long b0 = 0; b0 < nblk; b0 += 1024) {
        long long i = b0 + threadIdx.x;
        unsigned int v = i < nblk ? blk_cnt[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            unsigned int t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblk) blk_cnt[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_stable_compact(const int32_t* __restrict__ ycol, long long N, const unsigned int* __restrict__ blk_off,
                                                        int32_t* __restrict__ sorted_rows) {
    __shared__ unsigned int wsum[16];
    const long long base = (long long)blockIdx.x * 1024;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned int pre[4]; bool on[4];
    for (int j = 0; j < 4; ++j) {
        long long r = base + j * 256 + threadIdx.x;
        on[j] = r < N && ycol[r] >= 0;
        unsigned long long m = __ballot(on[j]);
        pre[j] = __popcll(m & ((1ull << lane) - 1));
        if (lane == 0) wsum[j * 4 + wv] = __popcll(m);
    }
    __syncthreads();
    unsigned int off = blk_off[blockIdx.x];
    for (int j = 0; j < 4; ++j) {
        unsigned int o = off;
        for (int q = 0; q < j * 4 + wv; ++q) o += wsum[q];
        if (on[j]) sorted_rows[o + pre[j]] = (int32_t)(base + j * 256 + threadIdx.x);
    }
}

__global__ void k_bagging(unsigned int* __restrict__ rand_state, long long n_local, double fraction, const int32_t* __restrict__ sorted_rows,
                          uint8_t* __restrict__ row_in_bag /* [N], only training rows are written */, unsigned int* __restrict__ counters /* [2], reset for k_bag_lists */,
                          long long off /* global position of this rank's first training row (row-sharded; else 0) */, long long n_global) {
    // LightGBM draws one number per training-row POSITION, from one LCG per 1024 positions (seed = bagging_seed + block).  A row shard
    // holds the positions [off, off + n_local): it walks the blocks that overlap them from their first position on (a block that
    // straddles two ranks is advanced by both), so that every rank keeps the stream of every block it touches in step.
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) { counters[0] = 0u; counters[1] = 0u; }
    const long long g0 = (off / 1024 + b) * 1024;
    if (g0 >= off + n_local || g0 >= n_global) return;
    unsigned int x = rand_state[b];
    const long long g1 = g0 + 1024 < n_global ? g0 + 1024 : n_global;
    for (long long g = g0; g < g1; ++g) {
        x = 214013u * x + 2531011u;
        float f = (float)((x >> 16) & 0x7FFF) / 32768.0f;
        if (g >= off && g < off + n_local) row_in_bag[sorted_rows[g - off]] = ((double)f < fraction) ? 1 : 0;
    }
    rand_state[b] = x;
}

// d_count -> the all-reduce buffer of the row-sharded trainer (a kernel, like everything else in the boosting loop)
__global__ __launch_bounds__(256) void k_copy_i32(const int32_t* __restrict__ src, int32_t* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// unstable split of the training rows into the bag list and the out-of-bag list
__global__ __launch_bounds__(256) void k_bag_lists(const int32_t* __restrict__ sorted_rows, long long n_train, const uint8_t* __restrict__ row_in_bag,
                                                   int32_t* __restrict__ bag, int32_t* __restrict__ oob, unsigned int* __restrict__ counters /* [2] */) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int row = 0; bool in = false, out = false;
    if (p < n_train) { row = sorted_rows[p]; in = row_in_bag[row] != 0; out = !in; }
    unsigned long long mi = __ballot(in), mo = __ballot(out);
    unsigned int bi = 0, bo = 0;
    if (lane == 0) { if (mi) bi = atomicAdd(&counters[0], (unsigned)__popcll(mi)); if (mo) bo = atomicAdd(&counters[1], (unsigned)__popcll(mo)); }
    bi = __shfl(bi, 0); bo = __shfl(bo, 0);
    if (in) bag[bi + __popcll(mi & ((1ull << lane) - 1))] = row;
    if (out) oob[bo + __popcll(mo & ((1ull << lane) - 1))] = row;
}

// out-of-bag score update by tree traversal on the training bins (ScoreUpdater::AddScore(tree, oob))
__global__ __launch_bounds__(256) void k_score_update_oob(const uint8_t* __restrict__ rec8, const int32_t* __restrict__ oob, const unsigned int* __restrict__ counters,
                                                          const TreeState* __restrict__ state, TreeOut out, const FeatMeta* __restrict__ fmeta,
                                                          const double* __restrict__ upd_value, double* __restrict__ score, int it, TrainConst c) {
    const int k = blockIdx.y;
    if (state[k].L <= 1) return;
    const long long n_oob = counters[1];
    const long long tbase = (long long)it * c.K + k;
    const long long nb = tbase * (c.num_leaves - 1);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_oob; i += (long long)gridDim.x * 256) {
        const int row = oob[i];
        int node = 0;
        for (;;) {
            const int f = out.feat[nb + node];
            const int bin = rec8[((long long)(f >> 4) * c.N + row) * 16 + (f & 15)];
            const bool miss = fmeta[f].has_nan && bin == fmeta[f].V;
            const bool go_left = miss ? (out.dleft[nb + node] != 0) : (bin <= out.theta[nb + node]);
            const int nx = go_left ? out.left[nb + node] : out.right[nb + node];
            if (nx < 0) { score[(long long)k * c.N + row] += upd_value[(long long)k * c.num_leaves + (~nx)]; break; }
            node = nx;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K9: batched leaf walk.  grid (ceil(n/256), K), block 256: thread = (row, class); 8-byte nodes.
// ------------------------------------------------------------------------------------------------
template <bool ONE_CHUNK /* F <= 16: the row's whole bin record stays in registers, a node visit is ONE load */>
__global__ __launch_bounds__(256) void k_predict_raw(const uint8_t* __restrict__ rec8, long long n, const PNode* __restrict__ nodes,
                                                     const double* __restrict__ leaf_value, int n_iter, int K, int node_stride, int leaf_stride,
                                                     double* __restrict__ raw /* [K][n] */) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= n) return;
    uint4 r = make_uint4(0, 0, 0, 0);
    if (ONE_CHUNK) r = reinterpret_cast<const uint4*>(rec8)[i];
    double s = 0.0;
    for (int it = 0; it < n_iter; ++it) {
        const long long t = (long long)it * K + k;
        const PNode* nd = nodes + t * node_stride;
        int node = 0;
        for (;;) {
            const PNode p = nd[node];
            const int f = p.w0 & 0xFFFF, theta = (int)((p.w0 >> 16) & 0x1FF) - 1, dleft = (p.w0 >> 25) & 1;
            int bin;
            if (ONE_CHUNK) {
                const bool hi = (f & 8) != 0;
                const uint32_t lo32 = hi ? r.z : r.x, hi32 = hi ? r.w : r.y;
                bin = (int)__builtin_amdgcn_perm(hi32, lo32, (uint32_t)(f & 7) | 0x0C0C0C00u);
            } else bin = rec8[((long long)(f >> 4) * n + i) * 16 + (f & 15)];
            const bool go_left = (bin == 255) ? (dleft != 0) : (bin <= theta);
            const int nx = go_left ? (int)(short)(p.w1 & 0xFFFF) : (int)(short)(p.w1 >> 16);
            if (nx < 0) { s += leaf_value[t * leaf_stride + (~nx)]; break; }
            node = nx;
        }
    }
    raw[(long long)k * n + i] = s;
}

// ------------------------------------------------------------------------------------------------
// k_predict_qs: GBDT::PredictRaw without walking the trees (bit-vector scoring over BINNED features, after QuickScorer, Lucchese et
// al., SIGIR 2015).  Number the leaves of a tree from left to right.  Every internal node owns a bit mask with zeros for the leaves
// of its LEFT subtree; a row's exit leaf is the lowest set bit of the AND of the masks of all nodes whose test is FALSE for the row.
// Features are bin codes here, so per tree and feature the AND over "all false nodes on this feature" is a function of the bin alone
// and is tabulated on the host: M[tree][feature][bin] (one more entry per feature for a missing / unseen value, which goes the node's
// default direction).  Scoring a (row, tree) pair is then F independent LDS lookups, F ANDs, one find-first-bit and one leaf-value
// lookup: no data-dependent chain, no divergence, ~2 instructions per feature.  (The walk over index-linked nodes, k_predict_raw, is
// latency-bound at ~150 lane-cycles per node visit; a fixed-depth walk over complete-tree copies was built first in round 3 and was
// SLOWER -- 7 visits of ~22 instructions against ~5 data-dependent ones.)
//   A workgroup owns 256 x QS_ROWS rows of ONE class; a row's table offsets (feature base + bin) are computed once and stay in
//   registers; the tables of that class's trees are staged through LDS, `tb` at a time, double-buffered.  The leaf values are added
//   in iteration order, as the oracle does: raw scores are bit-identical.
//   grid (ceil(n / (256 QS_ROWS)), K), block 256.  MW = mask words (1: <= 32 leaves, 2: <= 64), FMAX = 16 | 32 features.
// ------------------------------------------------------------------------------------------------
constexpr int QS_ROWS = 2;
//   TW / TBN > 0 (round 4): the staged trees sit at COMPILE-TIME strides (TW mask words per tree, padded; TBN trees per stage), the loop
//   over the trees of a stage is unrolled, and a tree's byte offset becomes the immediate of the ds_read: the per-read address add
//   (one VALU instruction per feature, row and tree -- a quarter of the loop) is gone.  TW = TBN = 0: strides from the arguments.
template <int MW, int FMAX, int TW, int TBN>
__global__ __launch_bounds__(256) void k_predict_qs(const uint8_t* __restrict__ rec8, long long n, const uint32_t* __restrict__ masks /* [T][S * MW] */,
                                                    const double* __restrict__ leaves /* [T][32 * MW] */, const uint32_t* __restrict__ used /* [T] features a tree splits on */,
                                                    const int32_t* __restrict__ foff /* [F + 1] */,
                                                    int F, int S, int tb_n_arg /* trees per LDS stage */, int n_iter, int K, double* __restrict__ raw /* [K][n] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LP = 32 * MW;
    const int tree_words_src = S * MW;                               // mask words of a tree in global memory
    const int tree_words = TW > 0 ? TW : tree_words_src;             // ... and its stride in LDS
    const int tb_n = TBN > 0 ? TBN : tb_n_arg;
    uint32_t* sm = reinterpret_cast<uint32_t*>(smem);                // [2][tb_n][tree_words]
    double* sl = reinterpret_cast<double*>(sm + 2 * tb_n * tree_words + ((2 * tb_n * tree_words) & 1));   // [2][tb_n][LP], 8-byte aligned
    uint32_t* su = reinterpret_cast<uint32_t*>(sl + 2 * tb_n * LP);  // [2][tb_n] used-feature bits
    const int k = blockIdx.y, tid = threadIdx.x;
    const long long base = (long long)blockIdx.x * (256 * QS_ROWS);
    int off[QS_ROWS][FMAX]; long long row[QS_ROWS]; double s[QS_ROWS];
#pragma unroll
    for (int q = 0; q < QS_ROWS; ++q) {
        const long long r0 = base + q * 256 + tid;
        row[q] = r0 < n ? r0 : n - 1; s[q] = 0.0;
#pragma unroll
        for (int f = 0; f < FMAX; ++f) {
            int o = 0;
            if (f < F) {
                const int fb = foff[f], nb = foff[f + 1] - fb - 1;                                   // nb value bins, entry nb = missing
                const unsigned bin = rec8[((long long)(f >> 4) * n + row[q]) * 16 + (f & 15)];
                o = (fb + (int)(bin < (unsigned)nb ? bin : (unsigned)nb)) * MW * 4;                 // byte offset inside a tree's table
            } else if (TW > 0) o = S * MW * 4;                         // no such feature: the all-ones pad entry behind the tree's masks
            off[q][f] = o;
        }
    }
    auto stage = [&](int it0, int buf) {
        for (int tb = 0; tb < tb_n; ++tb) {
            const int it = it0 + tb;
            if (it >= n_iter) break;
            const long long t = (long long)it * K + k;
            for (int i = tid; i < tree_words_src; i += 256) sm[(buf * tb_n + tb) * tree_words + i] = masks[t * tree_words_src + i];
            if (TW > 0 && tid < MW) sm[(buf * tb_n + tb) * tree_words + tree_words_src + tid] = 0xFFFFFFFFu;
            if (tid < LP) sl[(buf * tb_n + tb) * LP + tid] = leaves[t * LP + tid];
            if (tid == LP) su[buf * tb_n + tb] = used[t];
        }
    };
    stage(0, 0);
    __syncthreads();
    // the trees of LDS buffer `buf` (a compile-time constant in the fixed-stride variant: with the tree's stride it folds into the
    // immediate of the ds_read -- a run-time buffer base was one v_add per read, 256 of the 711 VALU instructions of a stage)
    auto score_stage = [&](const int buf, const int it0) __attribute__((always_inline)) {
        const int nt = (n_iter - it0) < tb_n ? (n_iter - it0) : tb_n;
        auto one_tree = [&](int tb) __attribute__((always_inline)) {
            const unsigned char* tm = reinterpret_cast<const unsigned char*>(sm + (buf * tb_n + tb) * tree_words);
            const double* tl = sl + (buf * tb_n + tb) * LP;
            // the mask of a feature the tree never splits on is all ones for every bin
            const uint32_t um = TBN > 0 ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)su[buf * tb_n + tb]);
            uint32_t v0[QS_ROWS], v1[QS_ROWS];
#pragma unroll
            for (int q = 0; q < QS_ROWS; ++q) { v0[q] = 0xFFFFFFFFu; v1[q] = 0xFFFFFFFFu; }
#pragma unroll
            for (int f = 0; f < FMAX; ++f) {
                // fixed strides: every feature's mask is read, all reads of a tree in flight together (a branch per feature would put a wait
                // for the LDS behind every read: measured in the ISA, 45 s_waitcnt per tree); dynamic strides: unused features are skipped
                if (TBN > 0 || (f < F && ((um >> f) & 1u))) {          // uniform
#pragma unroll
                    for (int q = 0; q < QS_ROWS; ++q) {
                        if (MW == 1) v0[q] &= *reinterpret_cast<const uint32_t*>(tm + off[q][f]);
                        else { const uint2 m2 = *reinterpret_cast<const uint2*>(tm + off[q][f]); v0[q] &= m2.x; v1[q] &= m2.y; }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < QS_ROWS; ++q) {
                int leaf;
                // (the AND always keeps the exit leaf's bit: count-trailing-zeros needs no zero case)
                if (MW == 1) leaf = __builtin_ctz(v0[q]);
                else leaf = v0[q] ? __builtin_ctz(v0[q]) : 32 + __builtin_ctz(v1[q]);
                s[q] += tl[leaf];
            }
        };
        if (TBN > 0) {
#pragma unroll
            for (int tb = 0; tb < (TBN > 0 ? TBN : 1); ++tb) if (tb < nt) one_tree(tb);     // (uniform; the last stage may be partial)
        } else {
            for (int tb = 0; tb < nt; ++tb) one_tree(tb);
        }
    };
    if (TBN > 0) {
        for (int it0 = 0; it0 < n_iter; it0 += 2 * tb_n) {
            if (it0 + tb_n < n_iter) stage(it0 + tb_n, 1);
            score_stage(0, it0);
            __syncthreads();
            if (it0 + tb_n >= n_iter) break;
            if (it0 + 2 * tb_n < n_iter) stage(it0 + 2 * tb_n, 0);
            score_stage(1, it0 + tb_n);
            __syncthreads();
        }
    } else {
        int buf = 0;
        for (int it0 = 0; it0 < n_iter; it0 += tb_n, buf ^= 1) {
            if (it0 + tb_n < n_iter) stage(it0 + tb_n, buf ^ 1);
            score_stage(buf, it0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int q = 0; q < QS_ROWS; ++q) if (base + q * 256 + tid < n) raw[(long long)k * n + row[q]] = s[q];
}

// ConvertOutput + arg-max (first maximum).  proba may be null (labels only).
__global__ __launch_bounds__(256) void k_softmax_argmax(const double* __restrict__ raw, long long n, int objective, int K,
                                                        double* __restrict__ proba /* [n][ncol] or null */, int32_t* __restrict__ label, double* __restrict__ top) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (objective == 2) { double v = raw[i]; if (proba) proba[i] = v; if (label) label[i] = -1; if (top) top[i] = v; return; }
    if (objective == 0) {
        double pr = 1.0 / (1.0 + rg_exp(-raw[i]));
        double p0 = 1.0 - pr;
        if (proba) { proba[i * 2] = p0; proba[i * 2 + 1] = pr; }
        int best = (pr > p0) ? 1 : 0;
        if (label) label[i] = best;
        if (top) top[i] = best ? pr : p0;
        return;
    }
    double wmax = raw[i];
    for (int k = 1; k < K; ++k) { double s = raw[(long long)k * n + i]; if (s > wmax) wmax = s; }
    double wsum = 0.0;
    for (int k = 0; k < K; ++k) wsum += rg_exp(raw[(long long)k * n + i] - wmax);
    int best = 0; double bp = -1.0;
    for (int k = 0; k < K; ++k) {
        double pk = rg_exp(raw[(long long)k * n + i] - wmax) / wsum;
        if (proba) proba[i * K + k] = pk;
        if (pk > bp) { bp = pk; best = k; }
    }
    if (label) label[i] = best;
    if (top) top[i] = bp;
}

// K10: fill only NULL cells (pdf[y].where(pdf[y].notna(), predicted), model.py:1128,1133)
__global__ __launch_bounds__(256) void k_fill_cells(int32_t* __restrict__ col, long long n, const int32_t* __restrict__ label,
                                                    const int32_t* __restrict__ class_code, int n_class_codes) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (col[i] < 0) { int l = label[i]; if (l >= 0 && l < n_class_codes) col[i] = class_code[l]; }
}

}  // namespace rg
