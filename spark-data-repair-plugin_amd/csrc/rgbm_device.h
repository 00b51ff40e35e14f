// rgbm_device.h -- device-side data structures of the gfx950 trainer / predictor.
//
// Layout in HBM (DESIGN.md "Data layout"):
//   codes   int32 [C][N]            the label-encoded table, column-major, resident (rgbm_table)
//   rec     u8    [nchunk][N][16]   per-model bin records: 16 features per 16-byte record so one
//                                   lane loads one row with a single dwordx4; chunk-major
//   gh      f32x2 [K][N]            (gradient, hessian) of every class tree: LightGBM's float32 values (numerics v2.2)
//   score   f64   [K][N]            raw scores
//   idx     i32   2 x [K][n_train]  ping-pong row-index lists, partitioned per leaf
//   pool    i64x2 [K][NL][totbins]  per-leaf histograms (exact integer sums)
#pragma once
#include <stdint.h>
#include "rgbm_numerics.h"

namespace rg {

constexpr int MAX_CHUNK_FEATS = 16;

struct HistBin { long long g, h; };

// best split of a (leaf, feature) pair or of a leaf
struct Cand {
    double gain;          // relative gain, -inf if none
    int32_t theta, dleft;
    long long left_gq, left_hq;
    double left_out, right_out;
};

struct Leaf {
    int32_t begin, count, buf, depth;   // row range inside idx buffer `buf` (2 = shared base list)
    int32_t parent_node, is_left;
    long long Gq, Hq;
    Cand best;
    int32_t best_feature, pad;
};

// per class-tree control block, rewritten by the tree_step / finish_split kernels
struct TreeState {
    int32_t L, done;
    int32_t split_leaf, right_leaf;
    int32_t do_partition, do_hist, smaller_is_left, hist_is_root;
    int32_t part_feature, part_theta, part_dleft, part_nanbin;
    int32_t part_begin, part_count, part_buf, pad0;
    int32_t hist_begin, hist_count, hist_buf, pad1;
    uint32_t cursor_left, cursor_right;
};

// per-feature constants (uploaded once per model)
struct FeatMeta {
    int32_t V, has_nan, nbins /* V + has_nan */, hoff /* offset into a leaf histogram */;
    int32_t fast_base /* slot offset inside the chunk's packed LDS histogram */, rep_shift, wide_off /* offset inside chunk */, pad;
};

struct ChunkMeta {
    int32_t nfeat, fast_slots, wide_bins, first_feat;
};

struct TrainConst {
    // the grid of the class tree at hand (numerics v2.2: per class tree and iteration).  The big-table trainers read it from the FxScale
    // table k_fx_scale writes (tree_const); the batched small-table trainer measures it inside k_small_tree; the host leaves the model's
    // v2.1 floor here
    double inv_sg, inv_sh, sg, sh;
    FxGrid fx;
    double l1, l2, min_gain_to_split, min_sum_hessian, learning_rate, factor;
    int32_t min_data_in_leaf, max_depth, num_leaves, F, K, totbins, nchunk, objective;
    long long N, n_train;
    long long NG;   // row stride of the (g, h) arrays [K][NG]: N, or N rounded up to a whole wave tile (level grower)
};

// packed tree node for the predictor: one 8-byte load per visit
struct PNode {
    uint32_t w0;   // feature[15:0] | (theta+1)[24:16] | dleft[25]
    uint32_t w1;   // left[15:0] | right[31:16] (int16, negative = ~leaf)
};

}  // namespace rg
