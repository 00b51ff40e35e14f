// rgbm_level.h -- the level-synchronous ("streaming") tree grower for gfx950.
//
// LightGBM grows a tree leaf-wise (best-first).  Re-creating that literally on a GPU means one
// partition + one small gathered histogram pass per split: 150 dependent launches per boosting
// iteration, random 16-byte gathers and launch-latency-bound small leaves (measured round 1:
// k_partition 50 %, k_hist 24 % of the time, profiles/r01a_*).  This grower produces the SAME tree,
// bit for bit, from at most max_depth+1 fully coalesced streaming passes over the row block:
//
//   * every row carries the id of the speculative node it sits in (u8 [K][N], updated in place);
//   * pass L routes every row from its depth-(L-1) node to the depth-L child (one byte compare on
//     a record that sits in registers), and accumulates the histogram of ONE child per expanded
//     parent (the other is parent - child: sums are exact integers, so which child is built never
//     changes a bit of the result); rows of the built child are counted, the sibling's count is
//     parent - built;
//   * k_level_split scans both children of every expanded parent (same FindBestThreshold code as
//     the leaf-wise path);
//   * k_level_plan decides which nodes of the new level to expand.  A node is expanded unless it
//     PROVABLY cannot be split by best-first growth: with pm(X) = min gain on the path root..X,
//     every known node Y with pm(Y) > pm(X) is split before X (induction on the best-first
//     queue), so X is dead once num_leaves-1 such nodes exist.  Expansion is a superset of the
//     final tree, never a subset (how much of a superset: DESIGN.md section 5, "How much of a level pass is
//     needed at all"); it also publishes the rows of the expanded parents, and a class tree whose share
//     is below 1 in 16 -- the deep levels of many-class targets -- is swept through its node ids only
//     (k_level_mt, sparse sweep);
//   * k_level_replay runs LightGBM's best-first selection (ArrayArgs::ArgMax + Tree::Split
//     numbering) over the speculative nodes and emits the tree in exactly the leaf-wise order.
//
// Used when 1 <= max_depth <= 7 and F <= 255 (the reference fixes max_depth = 7, train.py:109);
// every other configuration takes the leaf-wise path of rgbm_kernels.h.
//
// Numerics v2.2 (rgbm_numerics.h): (g, h) are LightGBM's float32 values; a histogram slot is a pair of
// int64 sums on the fixed-point grid of ITS class tree in THIS iteration (FxScale table, k_fx_scale), updated
// with two 64-bit LDS atomics.  No packing, no carries, no drains: by construction of the grid no sum of any
// set of rows can overflow.
#pragma once
#include "rgbm_kernels.h"

namespace rg {

constexpr int LV_INACTIVE = 255;     // node id of rows that do not take part (target cell NULL)
constexpr int LV_MAX_EXP = 64;       // expanded parents per level (depth <= 6)
constexpr int LV_MAX_BUILT = 32;     // built children per level (parents of depth <= 5)
constexpr int LV_CNT_REP = 16;
constexpr int LV_THREADS = 1024;     // threads of a pass workgroup: one workgroup per CU, it owns the CU's LDS
constexpr int LV_TILE = 2 * LV_THREADS;      // root pass: two rows per lane and tile
constexpr int LV_LDS_TOTAL = 160 * 1024;
constexpr int LV_LDS_BYTES = LV_LDS_TOTAL;
constexpr int LV_MAX_DEPTH = 7;
constexpr int LV_MAX_LEAVES = 128;

// multi-tree level pass (k_level_mt)
constexpr int MT_MAX_T = 64;                 // class trees per workgroup
constexpr int MT_MAX_NODES = 128;            // built nodes per workgroup (all its class trees together)
constexpr int MT_RT_BUDGET = 256;            // table entries the host sizes T for: 2^L per class tree (the nodes so far / the children of the level)
constexpr int MT_MAX_RT = MT_RT_BUDGET + MT_MAX_T;      // ... + one dummy entry per class tree (ids past the end of a table are clamped to it)
constexpr int MT_WT_ROWS = 256;              // rows of one wave tile: 4 consecutive rows per lane
#ifndef MT_RING_N
#define MT_RING_N 128
#endif
constexpr int MT_RING_PLAIN = MT_RING_N;           // entries of a wave's built-row ring: <= 63 waiting + <= 64 appended per row step (64: the waiting ones leave first)
constexpr int MT_RING_SPEC = MT_RING_N < 128 ? 128 : MT_RING_N;    // wave-specialised pass: the consumer takes full batches only, so a ring holds two of them
__host__ __device__ constexpr int mt_ring(bool spec) { return spec ? MT_RING_SPEC : MT_RING_PLAIN; }
constexpr int MT_CNT_REP = 4;              // copies of a built node's row counter (one LDS atomic per ring entry)

constexpr int MT_THREADS_ACC2 = 768;
#ifndef MT_CONSUMERS_N
#define MT_CONSUMERS_N 4
#endif
constexpr int MT_LOCK_EVERY = 4;              // lock-step: tile rounds between two looks at the row block's progress words
constexpr int MT_CONSUMERS = MT_CONSUMERS_N;              // wave-specialised pass: consumer waves of a workgroup (one per SIMD)         // workgroup of a two-chunk pass: 12 waves with 168 VGPRs each (two records per row stay in registers), a third less ring

struct SNode {   // speculative node of one class tree
    long long Gq, Hq;
    double pmin;                       // min gain on the path root..this node
    int32_t count, depth, parent, is_left;
    int32_t left, right;               // child ids, -1 = not expanded
    int32_t best_feature, searched;
    int32_t hslot, pad;                // histogram pool slot (-1: depth == max_depth)
    Cand best;
};

struct LvPlan {   // per class tree; rewritten by k_level_init / k_level_plan
    int32_t n_nodes, lvl_first, lvl_end, n_exp, n_built, live_rows /* rows of the expanded parents (k_level_plan) */, pad1, pad2, pad3, done, error, child_first, n_hslots, pad4;
    long long n_in;
    uint8_t exp[LV_MAX_EXP];           // expanded parents (ascending node id)
    uint8_t built_is_left[LV_MAX_EXP];
    uint32_t route0[256];              // feature | (theta+1)<<8 | nanbin<<16 | expanded<<24 | dleft<<25
    uint32_t route1[256];              // left | right<<8 | left built slot<<16 | right built slot<<24  (0xFF = none)
};

struct LevelConst {
    int32_t gx;                  // row blocks per class tree of THIS launch (partials: [K][gx][max_built][totbins])
    int32_t max_built;           // built-node stride of the partials of THIS launch
    int32_t nchunk, K, F, totbins, num_leaves, max_depth, min_data_in_leaf, lds_bytes;
    int32_t sib_local;           // k_level_split also writes the derived sibling count into the LOCAL count array (row-sharded: rank 0 only)
    int32_t xcd_blocks;          // root pass: 1-D grid of K * gx blocks, contiguous row blocks, all class trees of a row block on one XCD
    // k_level_mt launch: class trees per workgroup, tree groups, chunk whose features are accumulated, built-slot window, routing?
    int32_t mt_T, mt_G, mt_ch, mt_slot0, mt_nslots, mt_route;
    int32_t mt_sparse, mt_window;   // mt_sparse 1: class trees with few live rows are swept through their node ids (k_level_mt, plain single-chunk pass);
                                    // mt_window > 0: the class-tree groups of a row block walk it in step (wave-specialised pass, see "lock-step")
    uint32_t mt_epoch, has_mult;    // lock-step: tag of this launch in the progress words; has_mult: rows carry multiplicities in byte 15 of their (last / joint) record
    long long N, NS, NG;         // rows; row stride of the node-id arrays and of the (g, h) arrays (both N rounded up to a whole wave tile of 256 rows)
};

__device__ __forceinline__ uint32_t rec_byte(const uint4& r, int j) {
    uint32_t w = (j < 8) ? ((j < 4) ? r.x : r.y) : ((j < 12) ? r.z : r.w);
    return (w >> (8 * (j & 3))) & 0xFFu;
}

// ------------------------------------------------------------------------------------------------
// LDS histogram layout of a chunk: a slot is a pair of 8-byte sums; feature j holds nbins << sh_j slots (bin-major, replica-minor) so
// that lanes hitting the same bin spread over 2^sh_j addresses.  The layout has ONE parameter q, the "slot exponent": every feature is
// replicated until it holds about 2^q slots, sh_j = clamp(q - ceil_log2(nbins_j), 0, 5) -- few-bin features get many copies, many-bin
// features few.  Same-address collisions inside one wave instruction of atomics serialise, and with 64 lanes on b bins x r copies the
// expected pile-up is ~ 64 / (b r): equal slots per feature minimise the sum over the features for a given number of LDS bytes
// (measured at K = 64, level 4: 20-30 cycles per atomic instruction with one uniform replication factor, 12.8 in the root pass).
// q = 0 is the plain layout (one slot per bin).
// ------------------------------------------------------------------------------------------------
constexpr int LV_MAX_Q = 11;
__host__ __device__ inline int lv_shift(int nbins, int q) {
    int l = 0; while ((1 << l) < nbins) ++l;          // ceil_log2(nbins)
    int s = q - l;
    return s < 0 ? 0 : (s > 5 ? 5 : s);
}

__host__ __device__ inline int lv_slots(const FeatMeta* fm, int nfeat, int q) {
    int t = 0;
    for (int j = 0; j < nfeat; ++j) t += fm[j].nbins << lv_shift(fm[j].nbins, q);
    return t;
}

// FEATURE ROTATION of the level pass's histogram updates (round 5).  A batch of 64 built rows is one wave of LDS atomics per feature, and at
// the deep levels the LDS has no room for replicas: the rows of a batch come from one class tree, mostly from one or two of its nodes and --
// on clustered data -- hold the SAME bin in most features, so the 64 lanes of an instruction pile up on a handful of addresses (the
// micro-benchmark: 32 lanes on one address cost 7x a conflict-free instruction; levels 4-5 of the K = 64 target took 3.4 / 4.4 ms where
// routing + conflict-free atomics are 2.4 / 2.0).  With rotation lane l works on feature (j + l) mod 16 in step j: the lanes of one instruction
// spread over all 16 feature histograms of their nodes, and only the four lanes l, l + 16, l + 32, l + 48 can meet on an address -- which
// at most four replicas (indexed by l / 16) resolve completely.  The record of an entry is rotated by (l mod 16) bytes once (12 VALU), after
// which step j reads byte j exactly as before; feature slots a lane does not have (beyond nfeat, or byte 15 = the slot id) are masked to
// bin 0 of a per-replica dummy slot.  Replication under rotation is one uniform shift <= 2 for every feature.
// Measured (profiles/r5c_*, r5o_*): as the ONLY form it loses 35-60 % at levels 1-4 of the K = 64 target -- where the LDS has room, the replicated layout is
// better than conflict-free: rows of a cluster hold the same bin, replica = lane index puts the lanes of an instruction on CONSECUTIVE 8-byte slots (no
// bank conflict at all), while the rotated lanes land on pseudo-random banks -- and it wins where the LDS holds one or two copies: level 5 of K = 64
// 3.93 -> 3.43 ms, of K = 32 2.41 -> 1.41, of K = 16 1.44 -> 0.82, and most levels of the two-chunk pass (a node's two histograms are 10 KB).  So rotation is a
// TEMPLATE PARAMETER (ROTP) and the host picks it per launch: fewer than EIGHT copies (round 5: three; the flat pipeline of round 6 moved the balance towards many class trees per workgroup) of the launch's worst-case histograms fit -> rotate, and give the
// workgroup as many class trees as one copy allows (rgbm.hip, RGBM_MT_ROT / RGBM_MT_ROT_COPIES2 / RGBM_MT_ROT_T).  Bench step 83.9 -> 80.5 ms (same box),
// one rank's 12.5M x 32 shard 72.1 -> 64.1 ms.  -DMT_ROT=1 still rotates everything (the experiment).
#ifndef MT_ROT
#define MT_ROT 0
#endif
constexpr bool MT_ROT_ALL = MT_ROT != 0;             // -DMT_ROT=1: every level pass rotates (the experiment); default: only the launches the host picks (template parameter ROTP)
__host__ __device__ constexpr int mt_rot_dummy(bool rot) { return rot ? 4 : 0; }         // dummy slots per node (one per replica index): where masked-off lanes add
__host__ __device__ inline int mt_shift(int nbins, int q, bool rot = MT_ROT_ALL) { return rot ? (q < 0 ? 0 : (q > 2 ? 2 : q)) : lv_shift(nbins, q); }
__host__ __device__ inline int mt_slots(const FeatMeta* fm, int nfeat, int q, bool rot = MT_ROT_ALL) {
    int t = 0;
    for (int j = 0; j < nfeat; ++j) t += fm[j].nbins << mt_shift(fm[j].nbins, q, rot);
    return t;
}
__host__ __device__ constexpr int mt_max_q(bool rot) { return rot ? 2 : LV_MAX_Q; }
constexpr int MT_ROT_DUMMY = mt_rot_dummy(MT_ROT_ALL);

// bytes the root pass needs besides the histogram: nothing but alignment slack
constexpr int LV_ROOT_FIXED = 256;
// bytes k_level_mt needs besides the histogram: tree table | node -> tree map | scalars | per-feature flush table (32 features) | ring
// heads / tails / done flags | packed
// tree entries | route entries | built-row counters | per-node grid (numerics v2.2: the high words of 2^e_g, 2^e_h of the node's class tree) |
// per-wave rings (record(s) 16 / 32 B + (g, h) 8 B + slot 2 B per entry) | slack
#if !defined(MT_SPARSE_DIV_N)
#define MT_SPARSE_DIV_N 16
#endif
// a class tree is swept sparsely when its live rows are fewer than 1 in MT_SPARSE_DIV: the sweep costs ~25 instructions per 256 rows for the
// filter + one dense step (~430 with the gathers) per 64 groups of 4 rows that hold a live row, against ~330 per 256 rows tile by tile:
// break-even near 18 % live rows
constexpr long long MT_SPARSE_DIV = MT_SPARSE_DIV_N;
__host__ __device__ constexpr long long mt_fixed_bytes(int threads, bool acc2, bool spec = false, bool li = true /* the ring carries the built slot of an entry in an array of its own (a chunk with 16 features, rows with multiplicities) instead of in byte 15 of the record */) {
    return ((!acc2 && !spec) ? (long long)MT_MAX_T * 8 /* live-node masks of the sparse sweep */ : 0) + (long long)MT_MAX_T * 32 + MT_MAX_NODES + 16 + 512 + 256 + (long long)(MT_MAX_T + 2) * 8 + (long long)MT_MAX_RT * 8 + (long long)MT_MAX_NODES * MT_CNT_REP * 4 + (long long)MT_MAX_NODES * 8 +
           (long long)(threads / 64 - (spec ? MT_CONSUMERS : 0)) * mt_ring(spec) * ((acc2 ? 32 : 16) + 8 + (li ? 2 : 0)) + 256;     // (consumer waves have no ring)
}
// distance in bytes between a slot's gradient sum and its hessian sum in the LDS of a level pass: a COMPILE-TIME constant per instantiation (the second atomic
// of a pair is the first one's address register + an immediate offset), half of what the smaller fixed part (no slot array) leaves of the CU's LDS
__host__ __device__ constexpr long long mt_hd(int threads, bool acc2, bool spec) { return ((LV_LDS_TOTAL - mt_fixed_bytes(threads, acc2, spec, false)) / 2) & ~15ll; }
// bytes a level pass has for its histograms (both halves): the gradient half starts behind the fixed part and must end before the hessian half, which must end
// inside the LDS; lds_bytes < LV_LDS_TOTAL is the test hook that forces several built-slot windows per level
__host__ __device__ constexpr long long mt_hist_bytes(long long lds_bytes, int threads, bool acc2, bool spec, bool li) {
    const long long fixed = mt_fixed_bytes(threads, acc2, spec, li), hd = mt_hd(threads, acc2, spec);
    const long long room_h = (long long)LV_LDS_TOTAL - fixed - hd;
    const long long x = hd < room_h ? hd : room_h;
    const long long a = lds_bytes - fixed;
    return a < 2 * x ? a : 2 * x;
}

// ------------------------------------------------------------------------------------------------
// k_level_init: per class tree, start of a boosting iteration
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_level_init(LvPlan* __restrict__ plan, SNode* __restrict__ nodes, int32_t* __restrict__ count,
                                                   const unsigned int* __restrict__ n_in_ptr, long long n_train,
                                                   unsigned long long* __restrict__ Q, FxGrid fx, FxScale* __restrict__ fxs, LevelConst c) {
    const int k = blockIdx.x, lane = lane_id();
    // numerics v2.2: this class tree's fixed-point grid of the iteration from the coarse sums of its gradients (k_fx_scale's work, here so that
    // the chain of small kernels of an iteration does not grow by a launch); the sums go back to zero for the next iteration
    if (lane == 0) {
        const int e_g = fx_tree_exponent(Q[2 * k], fx.q_mult, fx.c_g, fx.e_g_min, fx.e_g_max);
        const int e_h = fx_tree_exponent(Q[2 * k + 1], fx.q_mult, fx.c_h, fx.e_h_min, fx.e_h_max);
        FxScale f; f.sg = fx_pow2(e_g); f.sh = fx_pow2(e_h); f.inv_sg = fx_pow2(-e_g); f.inv_sh = fx_pow2(-e_h);
        fxs[k] = f;
        Q[2 * k] = 0ull; Q[2 * k + 1] = 0ull;
    }
    LvPlan* pp = &plan[k];
    const long long n_in = n_in_ptr ? (long long)n_in_ptr[0] : n_train;
    // the child row counts of the new tree start at zero (here rather than in a hipMemsetAsync: the boosting loop is kernels only)
    for (int i = lane; i < 256; i += 64) { pp->route0[i] = 0; pp->route1[i] = 0xFFFFFFFFu; count[(long long)k * 256 + i] = 0; }
    if (lane == 0) {
        pp->n_nodes = 1; pp->lvl_first = 0; pp->lvl_end = 1; pp->n_exp = 0; pp->n_built = 1;
        pp->error = 0; pp->child_first = 1; pp->n_hslots = 1; pp->n_in = n_in;
        pp->done = (n_in < (long long)c.min_data_in_leaf * 2) ? 1 : 0;   // BeforeFindBestSplit on the root
        SNode r; memset(&r, 0, sizeof(r));
        r.count = (int)n_in; r.depth = 0; r.parent = -1; r.left = -1; r.right = -1; r.best_feature = -1; r.searched = 0; r.hslot = 0;
        r.best.gain = -INFINITY; r.pmin = -INFINITY;
        nodes[(long long)k * 256] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// k_level_root: ConstructHistograms of the root.  Every training row of class tree k sits in node 0: one coalesced pass over the
// (joint) bin record and the float32 (g, h) of every row, two 64-bit LDS atomics per feature (group).  One 1024-thread workgroup
// per CU owns the CU's LDS; the next tile is in flight while this tile's atomics run.
//   grid: xcd_blocks ? K * gx (1-D; id -> xcd = id % 8, r = id / 8, class tree = r % K, row block = (r / K) * 8 + xcd: all class trees
//         of a row block on ONE XCD, which then shares its L2 copy of the block's records) : (gx, K), strided tiles;  grid.z = chunk.
// Algorithmic bytes per row: F bin bytes + 8 B (g, h).
// ------------------------------------------------------------------------------------------------
template <bool WIDE /* the record holds eight 16-bit joint codes (groups of up to JOINT_WIDE_CAP joint bins) instead of sixteen bytes */>
__global__ __launch_bounds__(LV_THREADS, 1) void k_level_root(const uint4* __restrict__ rec, const float2* __restrict__ gh, const uint8_t* __restrict__ node,
                                                              const LvPlan* __restrict__ plan, HistBin* __restrict__ part,
                                                              const FeatMeta* __restrict__ fmeta, const ChunkMeta* __restrict__ cmeta,
                                                              const FxScale* __restrict__ fxs /* [K] this iteration's grid per class tree */, LevelConst c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int k, bx, nbx;
    if (c.xcd_blocks) { const unsigned id = blockIdx.x, r = id >> 3; k = (int)(r % (unsigned)c.K); bx = (int)(r / (unsigned)c.K) * 8 + (int)(id & 7u); nbx = c.gx; }
    else { k = blockIdx.y; bx = blockIdx.x; nbx = gridDim.x; }
    const int ch = blockIdx.z;
    if (plan[k].done) return;
    const double sg_k = fxs[k].sg, sh_k = fxs[k].sh;          // (uniform: scalar loads)
    const ChunkMeta cm = cmeta[ch];
    const FeatMeta* fm = fmeta + cm.first_feat;
    const int tid = threadIdx.x, lane = tid & 63, nfeat = cm.nfeat;
    // layout: the largest uniform replication that fits
    int s = LV_MAX_Q;
    while (s > 0 && (long long)lv_slots(fm, nfeat, s) * 16 + LV_ROOT_FIXED > c.lds_bytes) --s;
    int sh[16], fbase[16], spn = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < nfeat) { sh[j] = lv_shift(fm[j].nbins, s); fbase[j] = spn; spn += fm[j].nbins << sh[j]; }
        else { sh[j] = 0; fbase[j] = 0; }
    }
    // gradient sums and hessian sums live in SEPARATE arrays of 8-byte slots: one wave instruction of 64-bit atomics then spreads over
    // all 64 LDS banks (interleaved (g, h) pairs put every g on 16 of the 32 bank pairs: SQ_LDS_BANK_CONFLICT was 70 % of the LDS cycles)
    unsigned long long* hist_g = reinterpret_cast<unsigned long long*>(smem);
    unsigned long long* hist_h = hist_g + spn;
    for (int i = tid; i < 2 * spn; i += LV_THREADS) hist_g[i] = 0ull;
    int cj[16], sh3[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { sh3[j] = sh[j] + 3; cj[j] = (fbase[j] + (lane & ((1 << sh[j]) - 1))) * 8; }
    const int hdelta = spn;      // elements between a slot's gradient sum and its hessian sum
    __syncthreads();

    const long long N = c.N;
    const uint8_t* node_in = node + (long long)k * c.NS;
    const uint4* recc = rec + (long long)ch * N;
    const float2* ghk = gh + (long long)k * c.NG;
    const long long ntiles_all = (N + LV_TILE - 1) / LV_TILE;
    const long long tbeg = c.xcd_blocks ? ntiles_all * bx / nbx : bx;
    const long long ntiles = c.xcd_blocks ? ntiles_all * (bx + 1) / nbx : ntiles_all;
    const long long tstep = c.xcd_blocks ? 1 : nbx;

    constexpr int RPT = LV_TILE / LV_THREADS;
    int cur_n[RPT], nxt_n[RPT]; uint4 cur_r[RPT], nxt_r[RPT]; float2 cur_g[RPT], nxt_g[RPT];
    // straight-line loads (clamped offsets, no branches) so that the in-order vmcnt bookkeeping stays exact
    auto fetch = [&](long long t, int (&fn)[RPT], uint4 (&fr)[RPT], float2 (&fg)[RPT]) __attribute__((always_inline)) {
        const bool tv = t < ntiles;                                   // uniform
        const long long pb = tv ? t * LV_TILE : 0;
        const long long left_rows = N - pb;
        const unsigned lim = (unsigned)(left_rows < LV_TILE ? left_rows : LV_TILE) - 1u;   // last valid offset in the tile
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const unsigned o = (unsigned)(q * LV_THREADS + tid);
            const unsigned oc = o < lim ? o : lim;
            typedef float v2f __attribute__((ext_vector_type(2)));
            const int nv = (int)__builtin_nontemporal_load(node_in + pb + oc);          // (node ids and (g, h): read once per iteration; the records are re-read by the other class trees)
            fr[q] = (recc + pb)[oc];
            const v2f gv = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(ghk + pb + oc));
            fg[q] = make_float2(gv.x, gv.y);
            fn[q] = (tv && o <= lim) ? nv : LV_INACTIVE;
        }
    };
    auto accumulate = [&](bool on, const uint4& r, const float2& g) __attribute__((always_inline)) {
        if (on && (g.x != 0.0f || g.y != 0.0f)) {   // out-of-bag rows carry (0, 0)
            unsigned long long gq = (unsigned long long)fx_from_f32(g.x, sg_k), hq = (unsigned long long)fx_from_f32(g.y, sh_k);
            if (c.has_mult) { const unsigned long long m = r.w >> 24; gq *= m; hq *= m; }     // (a row that stands for m identical rows: exact integers, so m times the value IS their sum)
            unsigned char* hb = reinterpret_cast<unsigned char*>(hist_g);
            const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#define LV_ATOM(j) { const uint32_t code_ = WIDE ? ((w[((j) & 7) >> 1] >> (16 * ((j) & 1))) & 0xFFFFu) : ((w[(j) >> 2] >> (8 * ((j) & 3))) & 0xFFu); \
                     unsigned long long* p_ = reinterpret_cast<unsigned long long*>(hb + cj[j] + (int)(code_ << sh3[j])); \
                     atomicAdd(p_, gq); atomicAdd(p_ + hdelta, hq); }
#pragma unroll
            for (int j = 0; j < (WIDE ? 8 : 16); ++j) if (j < nfeat) LV_ATOM(j);
#undef LV_ATOM
        }
    };
    long long t = tbeg;
    fetch(t, cur_n, cur_r, cur_g);
    while (t < ntiles) {
        fetch(t + tstep, nxt_n, nxt_r, nxt_g);
#pragma unroll
        for (int q = 0; q < RPT; ++q) accumulate(cur_n[q] != LV_INACTIVE, cur_r[q], cur_g[q]);
        t += tstep; if (t >= ntiles) break;
        fetch(t + tstep, cur_n, cur_r, cur_g);
#pragma unroll
        for (int q = 0; q < RPT; ++q) accumulate(nxt_n[q] != LV_INACTIVE, nxt_r[q], nxt_g[q]);
        t += tstep;
    }
    __syncthreads();
    // flush this workgroup's partial histogram (plain stores: no global atomics, no zeroing)
    HistBin* dst = part + (((long long)k * c.gx + bx) * c.max_built) * c.totbins;
    for (int j = 0; j < nfeat; ++j) {
        const int shj = lv_shift(fm[j].nbins, s);
        int fb = 0;
        for (int q = 0; q < j; ++q) fb += fm[q].nbins << lv_shift(fm[q].nbins, s);
        for (int b = tid; b < fm[j].nbins; b += LV_THREADS) {
            long long tg = 0, th = 0;
            const unsigned long long* sg_ = hist_g + fb + (b << shj); const unsigned long long* sh_ = hist_h + fb + (b << shj);
            for (int r2 = 0; r2 < (1 << shj); ++r2) { tg += (long long)sg_[r2]; th += (long long)sh_[r2]; }
            HistBin o; o.g = tg; o.h = th;
            dst[fm[j].hoff + b] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_level_mt: THE level pass -- DataPartition::Split of a level and ConstructHistograms of its built children, for T class trees
// of a row block in ONE workgroup.
//
// A lane owns 4 consecutive rows; their bin records are loaded ONCE and stay in registers while the lane walks the T class trees of
// its workgroup.  Per class tree -- one STEP of the wave's flat software pipeline, which runs over all its (wave tile, class tree) pairs with
// the loads of two further steps in flight (round 6: see "flat pipeline" below) -- it reads one dword of node ids and the rows' float32 (g, h)
// (two dwordx4), takes every row's entry out of an LDS route table indexed by the node id itself, moves the rows to their children IN PLACE (changed
// dwords only) and appends the rows that fall into a BUILT child -- record, (g, h), histogram slot -- to its wave's LDS ring.  Whenever
// 64 entries wait, they are taken out as one FULL wave of histogram updates: (g, h) go onto the fixed-point grid and into the built
// child's LDS histogram with two 64-bit atomics per feature.  A wave instruction of LDS atomics costs the same for 6 active lanes as
// for 64 (profiles/r01_lds_atomic_active_lanes.txt) and a built child holds ~12 % of the rows, hence the ring.
//
// What this replaces (rounds 1-2): a routing kernel + a streaming kernel per level that re-read the bin records of the built rows from
// L2 / HBM per class tree (2.2-4.3 GB of 128-byte lines for 16-byte records at K = 64, profiles/r02q_hbm_traffic_pmc.txt), ~80 VALU
// instructions per 64 (row, class tree) pairs between them.  Here the records cross the memory system once per T class trees and the
// routing + ring append is one pass over registers.
//
// T is what the CU's LDS holds: T * (built nodes per class tree) histograms of the chunk + the rings.  The histograms of all class
// trees of a workgroup share the LDS; batches mix the class trees a wave has walked, which spreads the atomics like replication does.
//   launch = (chunk, built-slot window): the first launch of a level routes (mt_route) and builds chunk 0 / the first window of built
//   slots; tables with more than 16 features and levels whose histograms exceed the LDS take further launches that find the rows of
//   their built children by the (already final) node ids.
//   grid: 1-D, mt_G * gx blocks; id -> xcd = id % 8, slot = id / 8, tree group = slot % G, row block = (slot / G) * 8 + xcd.
// Algorithmic bytes: rows of built children x (F + 8); the pass also streams 1 + 8 B per (row, class tree) and writes <= 1 B.
// ------------------------------------------------------------------------------------------------
struct MtTree { int32_t base, nlev, rt_off, slot0, nb, live, child_first, k; };   // 32 B, one per class tree of the workgroup

template <int NCHR /* records a row needs for ROUTING: 1, 2 (both in registers), 0 = any number of chunks, the split byte is gathered */, bool BAG,
          bool ROUTE /* the first launch of a level: moves the rows to their children; later launches find the built rows by the final ids */,
          int THREADS /* 1024, or MT_THREADS_ACC2 */, bool ACC2 /* NCHR == 2 only: the histograms of BOTH chunks are accumulated by this launch */,
          bool SPEC /* wave-specialised: the last MT_CONSUMERS waves only run the batches (LDS atomics) out of the other waves' rings */,
          bool ROTP = false /* feature rotation of the histogram updates (see MT_ROT): the host picks it for the launches whose LDS holds fewer than eight copies of their worst-case histograms (RGBM_MT_ROT_COPIES2 = 16) */>
__global__ __launch_bounds__(THREADS, 1) void k_level_mt(const uint4* __restrict__ rec, const float2* __restrict__ gh, uint8_t* __restrict__ node /* [K][NS], in place */,
                                                         const uint8_t* __restrict__ inbag, const LvPlan* __restrict__ plan, HistBin* __restrict__ part,
                                                         int32_t* __restrict__ count, const FeatMeta* __restrict__ fmeta, const ChunkMeta* __restrict__ cmeta,
                                                         int32_t* __restrict__ err_flag, uint32_t* __restrict__ prog /* [row blocks][tree groups] lock-step progress words (SPEC), or null */,
                                                         const FxScale* __restrict__ fxs /* [K] this iteration's grid per class tree */, LevelConst c) {
    static_assert(!ACC2 || NCHR == 2, "a two-chunk pass keeps both records in registers");
    constexpr int WAVES = THREADS / 64;
    constexpr int NCONS = SPEC ? MT_CONSUMERS : 0, NPROD = WAVES - NCONS;     // waves that walk the rows / waves that only run batches
    constexpr int NACC = ACC2 ? 2 : 1;                     // chunks accumulated by this launch
    constexpr int NRINGS = SPEC ? NPROD : WAVES;
    constexpr int MT_RING = mt_ring(SPEC);
    constexpr bool ROT = ROTP || MT_ROT_ALL;
    // SPARSE: class trees whose expanded parents hold a few per cent of the rows (the deep levels of many-class targets: 1.3 % at level 6
    // of the K = 64 target, profiles/r04z_*) are not walked tile by tile; see "sparse sweep" below
    constexpr bool SPARSE = (NCHR == 1) && !ACC2 && !SPEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned id = blockIdx.x;
    const int xl = (int)(id & 7u), bslot = (int)(id >> 3);
    const int grp = bslot % c.mt_G, rb = (bslot / c.mt_G) * 8 + xl;
    const int k0 = grp * c.mt_T;
    const int nk = (c.K - k0) < c.mt_T ? (c.K - k0) : c.mt_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = ACC2 ? 0 : c.mt_ch;
    constexpr bool route = ROUTE;
    const ChunkMeta cm = cmeta[ch];
    const ChunkMeta cm1 = cmeta[ACC2 ? 1 : ch];
    const FeatMeta* fm = fmeta + cm.first_feat;
    const FeatMeta* fm1 = fmeta + cm1.first_feat;
    const int nfeat = cm.nfeat, nfeat1 = ACC2 ? cm1.nfeat : 0;
    const int wb = cm.wide_bins + (ACC2 ? cm1.wide_bins : 0);
    // the histogram slot of a ring entry rides in byte 15 of its (last) record whenever that chunk holds at most 15 features: one LDS
    // write and one LDS read less per entry (the 10M x 16 and the 100M x 32 shapes have 15 features in their last chunk)
    const bool li_in_rec = (ACC2 ? nfeat1 : nfeat) <= 15 && !c.has_mult;      // (with row multiplicities byte 15 is taken)

    // ---- LDS carve-up
    MtTree* ti = reinterpret_cast<MtTree*>(smem);                                              // [MT_MAX_T]
    uint8_t* nd_tree = smem + MT_MAX_T * 32;                                                   // [MT_MAX_NODES] local node -> class tree of the workgroup
    int32_t* scal = reinterpret_cast<int32_t*>(nd_tree + MT_MAX_NODES);                        // [4] total built nodes, replication shift, slots per node, any live class tree
    int32_t* ftab = scal + 4;                                                                  // [4][32] per accumulated feature: first wide bin | first slot | replication shift | histogram offset
    uint32_t* rsync = reinterpret_cast<uint32_t*>(ftab + 128);                                 // [3][16] per ring: entries appended | entries consumed | producer finished
    // (read / written with relaxed workgroup-scope atomics: real LDS instructions every time, never hoisted or merged; a `volatile`
    // pointer loses the LDS address space and turns into FLAT accesses that wait for every outstanding global load)
#define RS_LOAD(i) __hip_atomic_load(rsync + (i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define RS_STORE(i, v) __hip_atomic_store(rsync + (i), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
    uint2* tpk = reinterpret_cast<uint2*>(ftab + 128 + 64);                                    // [MT_MAX_T + 2] what the row loop needs of a class tree: base | nlev << 8 | live << 31, rt_off | k << 16
    uint2* rt = tpk + MT_MAX_T + 2;                                                            // [MT_MAX_RT] route entries / child -> slot entries
    int32_t* cnt = reinterpret_cast<int32_t*>(rt + MT_MAX_RT);                                 // [MT_MAX_NODES][MT_CNT_REP]
    uint2* nd_sc = reinterpret_cast<uint2*>(cnt + MT_MAX_NODES * MT_CNT_REP);                  // [MT_MAX_NODES] local node -> high words of 2^e_g, 2^e_h of its class tree (the low words of a power of two are 0)
    uint4* ring_rec_all = reinterpret_cast<uint4*>(nd_sc + MT_MAX_NODES);                      // [waves][MT_RING]
    uint4* ring_rec1_all = ring_rec_all + (ACC2 ? NRINGS * MT_RING : 0);                        // [waves][MT_RING] (two-chunk pass)
    uint2* ring_gh_all = reinterpret_cast<uint2*>(ring_rec1_all + NRINGS * MT_RING);            // [waves][MT_RING]
    uint16_t* ring_li_all = reinterpret_cast<uint16_t*>(ring_gh_all + NRINGS * MT_RING);        // [waves][MT_RING]
    size_t off = reinterpret_cast<unsigned char*>(ring_li_all + (li_in_rec ? 0 : NRINGS * MT_RING)) - smem;      // (no slot array where the slot rides in the record: 4 KB more for histograms)
    off = (off + 15) & ~(size_t)15;
    unsigned long long* xmask = reinterpret_cast<unsigned long long*>(smem + off);            // [MT_MAX_T] sparse sweep: bit i = node base + i of the class tree is live
    if (!ACC2 && !SPEC) off += (size_t)MT_MAX_T * 8;                                            // (reserved for every plain pass: mt_fixed_bytes)
    unsigned long long* hist_g = reinterpret_cast<unsigned long long*>(smem + off);     // [total][spn] gradient sums; the hessian sums [total][spn] (see k_level_root) start
    // MT_HD bytes further on (mt_hd: a compile-time distance, so that the second atomic of a (g, h) pair is the first one's address register with an immediate offset)
    constexpr int MT_HD = (int)mt_hd(THREADS, ACC2, SPEC);
    const long long avail = mt_hist_bytes(c.lds_bytes, THREADS, ACC2, SPEC, !li_in_rec);
    if (tid == 0 && ((long long)off > mt_fixed_bytes(THREADS, ACC2, SPEC, !li_in_rec) || (long long)off + MT_HD + avail / 2 > (long long)LV_LDS_TOTAL)) atomicOr(err_flag, 2);     // (never: mt_fixed_bytes covers the carve-up above)

    // ---- the class trees of this workgroup and their built slots inside this launch's window
    if (tid < 64) {   // wave 0: lane kk reads the plan of class tree k0 + kk; exclusive prefix sums place its built slots and table entries
        MtTree t; t.base = 0; t.nlev = 0; t.nb = 0; t.live = 0; t.child_first = 0; t.k = k0 + lane; t.slot0 = 0; t.rt_off = 0;
        if (lane < nk) {
            const LvPlan* pp = &plan[k0 + lane];
            const bool live = !pp->done && pp->n_exp > 0;
            t.live = live ? 1 : 0;
            t.child_first = pp->child_first;
            // routing launch: the table covers the nodes of the OLD level from the first expanded parent on; later launches: the children
            t.base = live ? (route ? (int)pp->exp[0] : pp->child_first) : 0;
            t.nlev = live ? (route ? pp->child_first - (int)pp->exp[0] : 2 * pp->n_exp) : 0;
            int nb = live ? pp->n_built - c.mt_slot0 : 0;
            if (nb > c.mt_nslots) nb = c.mt_nslots;
            t.nb = nb < 0 ? 0 : nb;
        }
        // sparse: a live class tree whose expanded parents hold less than 1/MT_SPARSE_DIV of its rows (and whose table fits a 64-bit node mask)
        bool sparse_t = false;
        if (SPARSE && lane < nk && t.live && t.nlev <= 64) {
            const LvPlan* pp = &plan[k0 + lane];
            const long long lr = route ? (long long)pp->live_rows : 0ll;     // (later launches of a level: not sparse -- they look for the built children)
            sparse_t = route && lr * MT_SPARSE_DIV < pp->n_in && c.mt_sparse != 0;
        }
        // entries of the class tree in the LDS table.  A ROUTING launch indexes its table by the node id itself: entries [0, child_first) -- the
        // nodes of the older levels and the unexpanded ones route to themselves -- plus ONE dummy entry at child_first that every larger id
        // (only LV_INACTIVE occurs) is clamped to and that routes to LV_INACTIVE: the row loop needs no "is the row in a node of this level" test,
        // no subtraction and no select -- the entry alone says where the row goes.  <= 2^L entries per class tree = what the host sizes T for.
        // A LATER launch of a level (more chunks / built-slot windows) looks for the rows of ITS built children by the final node ids: entries
        // [0, 2 n_exp) for the children (indexed id - child_first) and the same dummy behind them (an id below child_first wraps around to a large index).
        const int ntab = lane < nk ? (route ? (t.live ? t.child_first : 0) : t.nlev) + 1 : 0;
        int inc_nb = t.nb, inc_rt = ntab;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int a = __shfl_up(inc_nb, o), b2 = __shfl_up(inc_rt, o); if (lane >= o) { inc_nb += a; inc_rt += b2; } }
        t.slot0 = inc_nb - t.nb; t.rt_off = inc_rt - ntab;
        const int total = __shfl(inc_nb, 63), rt_total = __shfl(inc_rt, 63);
        const unsigned long long livem = __ballot(t.live != 0);
        // the host sizes T for the worst case of the level (2^(L-1) expanded parents per class tree), so this always holds; if it ever did
        // not, the pass would silently drop rows: the training call fails instead (err_flag bit 1)
        const bool ok = total <= MT_MAX_NODES && rt_total <= MT_MAX_RT;
        if (!ok) { t.live = 0; t.nb = 0; t.nlev = 0; if (lane == 0) atomicOr(err_flag, 2); }
        // the workgroup's class trees in LDS: the live dense ones first, then the sparse ones, then the ones that are finished or have nothing to split at this
        // level (the row loop walks [0, nkd), the sparse sweep [nkd, nke); a finished class tree is not streamed at all: neither its node ids nor its (g, h))
        const bool dead_t = lane < nk && (!t.live || (!route && t.nb == 0));       // (a later launch of a level: also a class tree with no built child in this launch's window)
        const unsigned long long dmask = __ballot(lane < nk && !sparse_t && !dead_t), smask = __ballot(lane < nk && sparse_t && !dead_t), zmask = __ballot(dead_t);
        const int nkd_ = __popcll(dmask), nks_ = __popcll(smask);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int pos = lane >= nk ? lane : (dead_t ? nkd_ + nks_ + __popcll(zmask & below) : (sparse_t ? nkd_ + __popcll(smask & below) : __popcll(dmask & below)));
        if (lane < nk) ti[pos] = t;
        tpk[pos] = make_uint2((uint32_t)t.base | (uint32_t)t.nlev << 8 | (uint32_t)(ntab > 0 ? ntab - 1 : 0) << 17 | (t.live && lane < nk ? 1u << 31 : 0u), (uint32_t)t.rt_off | (uint32_t)t.k << 16);
        if (lane < 2) tpk[64 + lane] = make_uint2(0u, 0u);
        if (SPARSE) xmask[lane] = 0ull;
        if (lane == 0) {
            // replication: the largest uniform shift whose histograms fit
            int s = mt_max_q(ROT);
            const int tot = ok ? total : 0;
            while (s > 0 && (long long)tot * (mt_slots(fm, nfeat, s, ROT) + (ACC2 ? mt_slots(fm1, nfeat1, s, ROT) : 0) + mt_rot_dummy(ROT)) * 16 > avail) --s;
            const int spn0 = mt_slots(fm, nfeat, s, ROT) + (ACC2 ? mt_slots(fm1, nfeat1, s, ROT) : 0) + mt_rot_dummy(ROT);
            if ((long long)tot * spn0 * 16 > avail) atomicOr(err_flag, 2);          // (the host's window sizing guarantees the plain layout fits)
            scal[0] = tot; scal[1] = s; scal[2] = spn0; scal[3] = ((ok && livem != 0ull) ? 1 : 0) | nkd_ << 8 | (nkd_ + nks_) << 16;
        }
    }
    __syncthreads();
    if (!(scal[3] & 1)) return;
    // wave-uniform from here on: in SGPRs, so that everything derived from them (shifts, strides) is scalar as well
    const int total = __builtin_amdgcn_readfirstlane(scal[0]), s = __builtin_amdgcn_readfirstlane(scal[1]), spn = __builtin_amdgcn_readfirstlane(scal[2]);
    if (!route && total == 0) return;
    const int nkd = __builtin_amdgcn_readfirstlane((scal[3] >> 8) & 0xFF), nke = __builtin_amdgcn_readfirstlane((scal[3] >> 16) & 0xFF);       // live dense class trees: [0, nkd), sparse ones: [nkd, nke)
    for (int kk = 0; kk < nk; ++kk) {
        const MtTree t = ti[kk];
        const LvPlan* pp = &plan[t.k];
        const int ntab_k = (route ? (t.live ? t.child_first : 0) : t.nlev) + 1;
        for (int i = tid; i < ntab_k; i += THREADS) {
            const int n = route ? i : t.base + i;
            uint2 e;
            if (route) {
                // LDS copy of the route table, specialised: built slots become workgroup-local (0xFF = that child's histogram is not built
                // in this launch) and an unexpanded node routes to itself.  Entry layout (what the row loop's instructions want):
                //   x: byte 0 = the split byte's index inside its half record (f & 7; NCHR == 0: the feature) | thr << 8 | off << 16 | expanded << 24 |
                //      second record << 30 | upper half of the record << 31;   y: left child | left built slot << 8 | right child << 16 | right built slot << 24
                const bool cur = i < ntab_k - 1 && n >= t.base;         // a node of the level being split (older ids: finished leaves; the last entry: the dummy)
                const uint32_t w0 = cur ? pp->route0[n] : 0u; const uint32_t w1 = cur ? pp->route1[n] : 0u;
                if (w0 & (1u << 24)) {
                    const int ls = (int)((w1 >> 16) & 0xFFu) - c.mt_slot0, rs = (int)(w1 >> 24) - c.mt_slot0;
                    const bool lb = ((w1 >> 16) & 0xFFu) != 0xFFu && ls >= 0 && ls < t.nb, rbb = (w1 >> 24) != 0xFFu && rs >= 0 && rs < t.nb;
                    // The test  left = (bin == NaN bin) ? default-left : (bin <= theta)  as ONE byte compare in the row loop:
                    //   left = ((bin - off) & 0xFF) <= thr.
                    // A node that sends missing values right, or whose feature has no NaN bin: off = 0, thr = theta (the NaN bin is the
                    // last bin, above every threshold).  A node that sends them LEFT: off = NaN bin V, thr = (theta - V) & 0xFF -- the
                    // NaN bin wraps to 0 (always <= thr), a value bin b < V to 256 + b - V, which is <= 256 + theta - V iff b <= theta.
                    // theta = -1 without the default-left case cannot come out of the split search; it maps to "never left" (off 255).
                    const uint32_t theta1 = (w0 >> 8) & 0xFFu, nanbin = (w0 >> 16) & 0xFFu, dleft = (w0 >> 25) & 1u;
                    uint32_t offb, thr;
                    if (nanbin != 255u && dleft) { offb = nanbin; thr = (theta1 - 1u - nanbin) & 0xFFu; }
                    else if (theta1 == 0u) { offb = 255u; thr = 0u; }
                    else { offb = 0u; thr = theta1 - 1u; }
                    const uint32_t f = w0 & 0xFFu;
                    e = make_uint2((NCHR == 0 ? f : (f & 7u) | ((f >> 4) & 1u) << 30 | ((f >> 3) & 1u) << 31) | thr << 8 | offb << 16 | 1u << 24,
                                   (w1 & 0xFFu) | (lb ? (uint32_t)(t.slot0 + ls) : 0xFFu) << 8 | ((w1 >> 8) & 0xFFu) << 16 | (rbb ? (uint32_t)(t.slot0 + rs) : 0xFFu) << 24);
                    if (SPARSE && kk >= nkd && kk < nke && n - t.base < 64) atomicOr(&xmask[kk], 1ull << (n - t.base));
                } else {
                    const uint32_t self = i < ntab_k - 1 ? (uint32_t)n : (uint32_t)LV_INACTIVE;
                    e = make_uint2(0u, self | 0xFF00u | self << 16 | 0xFF000000u);
                }
            } else {
                // children are numbered child_first + 2 ei (left), + 1 (right); one of the two is built, in slot ei.  y = the workgroup-local slot of a child
                // built by THIS launch, else 0xFF (every other child, the dummy)
                const int ei = i >> 1, ls = ei - c.mt_slot0;
                const bool mine = i < ntab_k - 1 && (pp->built_is_left[ei] ? 0 : 1) == (i & 1) && ls >= 0 && ls < t.nb;
                e = make_uint2(mine ? 1u : 0u, mine ? (uint32_t)(t.slot0 + ls) : 0xFFu);
            }
            rt[t.rt_off + i] = e;
        }
        const uint2 sc_k = make_uint2((uint32_t)__double2hiint(fxs[t.k].sg), (uint32_t)__double2hiint(fxs[t.k].sh));
        for (int i = tid; i < t.nb; i += THREADS) { nd_tree[t.slot0 + i] = (uint8_t)kk; nd_sc[t.slot0 + i] = sc_k; }
    }
    for (int i = tid; i < total * MT_CNT_REP; i += THREADS) cnt[i] = 0;
    if (tid < 48) rsync[tid] = 0u;            // (before the barrier below)
    unsigned long long* hist_h = hist_g + MT_HD / 8;
    for (int i = tid; i < total * spn; i += THREADS) { hist_g[i] = 0ull; hist_h[i] = 0ull; }
    // per accumulated feature: replication shift (scalar) and this lane's byte offset inside a node's slots (first slot + replica)
    // the shifts, 4 bits each, packed into one 64-bit SCALAR per chunk (16 separate scalars cost the row loop ~80 spill reloads per step)
    unsigned long long sh3p[NACC];
    int cj[NACC][16];
    {
        int o = 0;
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const FeatMeta* fa = a == 0 ? fm : fm1;
            const int na = a == 0 ? nfeat : nfeat1;
            unsigned long long pk = 0ull;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int shj = 0, fb = 0;
                if (j < na) { shj = mt_shift(fa[j].nbins, s, ROT); fb = o; o += fa[j].nbins << shj; }
                pk |= (unsigned long long)(shj + 3) << (4 * j); cj[a][j] = (fb + (lane & ((1 << shj) - 1))) * 8;
                if (tid == 0) {
                    const int q = a * 16 + j;
                    const int wide = (a == 0 ? 0 : cm.wide_bins) + (j < na ? fa[j].wide_off : 0);
                    ftab[q] = j < na ? wide : 0x7FFFFFFF; ftab[32 + q] = fb; ftab[64 + q] = shj; ftab[96 + q] = j < na ? fa[j].hoff - wide : 0;
                }
            }
            sh3p[a] = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pk) |
                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pk >> 32)) << 32;
        }
    }
    constexpr int hdelta = MT_HD / 8;
    __syncthreads();
    // feature rotation (MT_ROT): lane l works on feature slot (j + l) mod 16 in step j.  cj[a][j] becomes the byte offset of THAT feature's
    // first slot (+ the lane's replica l / 16), or of the lane's dummy slot when the chunk has no such feature; rmask[a] zeroes the bytes of
    // the rotated record that are not features, so that a masked lane adds to bin 0 of its dummy slot.
    uint32_t rmask[NACC][4];
    const int rot_sh = mt_shift(1, s, ROT) + 3;            // uniform under rotation
    if (ROT) {
        const int r16 = lane & 15, c4 = lane >> 4;
        const int spn_valid = spn - mt_rot_dummy(ROT);
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int na = a == 0 ? nfeat : nfeat1;
#pragma unroll
            for (int w = 0; w < 4; ++w) rmask[a][w] = 0u;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int f = (j + r16) & 15;
                const bool fv = f < na;
                cj[a][j] = fv ? (ftab[32 + a * 16 + f] + (c4 & ((1 << (rot_sh - 3)) - 1))) * 8 : (spn_valid + c4) * 8;
                rmask[a][j >> 2] |= fv ? (0xFFu << (8 * (j & 3))) : 0u;
            }
        }
    }

    const int my_ring = __builtin_amdgcn_readfirstlane(wave < NRINGS ? wave : 0);               // (consumer waves never append; scalar: the ring bases live in SGPRs)
    uint4* ring_rec = ring_rec_all + my_ring * MT_RING;
    uint4* ring_rec1 = ring_rec1_all + my_ring * MT_RING;
    uint2* ring_gh = ring_gh_all + my_ring * MT_RING;
    uint16_t* ring_li = ring_li_all + my_ring * MT_RING;
    int r_head = 0, r_cnt = 0;                                   // wave-uniform
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const unsigned spn8 = (unsigned)spn * 8u;
    // lane kk keeps the packed entry of class tree kk: the row loop fetches it with a readlane instead of an LDS read per step
    const uint2 tpk_v = tpk[lane];
    auto tree_entry = [&](int kk) __attribute__((always_inline)) -> uint2 {       // (of the trees the row loop walks: the dense ones in a plain pass)
        const int kc = kk < 64 ? kk : 63;
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)tpk_v.x, kc), y = (uint32_t)__builtin_amdgcn_readlane((int)tpk_v.y, kc);
        return kk < nkd ? make_uint2(x, y) : make_uint2(0u, 0u);
    };

    // one FULL (or final, partial) wave of histogram updates from the ring
    // (ring `rid`, first entry `head`): the calling wave's own ring in the plain pass, a producer's ring in the wave-specialised one
    auto run_batch_of = [&](int rid, int head, int nb) __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // ring entries are read by other lanes of this wave
        const bool on = lane < nb;
        const int pos = rid * MT_RING + ((head + lane) & (MT_RING - 1));
        const uint4 r = ring_rec_all[pos]; const uint2 g = ring_gh_all[pos];
        uint4 r1 = make_uint4(0, 0, 0, 0);
        if (ACC2) r1 = ring_rec1_all[pos];
        unsigned li;
        if (li_in_rec) li = (ACC2 ? r1.w : r.w) >> 24; else li = ring_li_all[pos];
        // (the packed shifts pass through an empty asm so that the compiler extracts them here, with scalar bit-field ops next to their
        // use, instead of hoisting 16 unpacked scalars out of the row loop and spilling them)
        unsigned long long shp[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a) { shp[a] = sh3p[a]; asm volatile("" : "+s"(shp[a])); }
        if (on) {
            const uint2 sc = nd_sc[li];                  // the grid of the entry's class tree
            unsigned long long gq = (unsigned long long)fx_from_f32(__uint_as_float(g.x), __hiloint2double((int)sc.x, 0)),
                               hq = (unsigned long long)fx_from_f32(__uint_as_float(g.y), __hiloint2double((int)sc.y, 0));
            int mrow = 1;
            if (c.has_mult) { mrow = (int)((ACC2 ? r1.w : r.w) >> 24); gq *= (unsigned long long)mrow; hq *= (unsigned long long)mrow; }
            if (ch == 0) atomicAdd(&cnt[li * MT_CNT_REP + (lane & (MT_CNT_REP - 1))], mrow);
            unsigned char* hb = reinterpret_cast<unsigned char*>(hist_g) + li * spn8;
            uint32_t w[4] = {r.x, r.y, r.z, r.w};
            uint32_t w1[4] = {r1.x, r1.y, r1.z, r1.w};
            if (ROT) {   // rotate the 16-byte record right by (lane mod 16) bytes: byte j of the result = byte (j + lane) mod 16 of the record
                auto rot16 = [&](uint32_t (&x)[4], const uint32_t (&mk)[4]) __attribute__((always_inline)) {
                    const bool d1 = (lane & 4) != 0, d2 = (lane & 8) != 0;
                    const uint32_t a0 = d1 ? x[1] : x[0], a1 = d1 ? x[2] : x[1], a2 = d1 ? x[3] : x[2], a3 = d1 ? x[0] : x[3];
                    const uint32_t b0 = d2 ? a2 : a0, b1 = d2 ? a3 : a1, b2 = d2 ? a0 : a2, b3 = d2 ? a1 : a3;
                    const uint32_t sb = (uint32_t)(lane & 3);
                    x[0] = __builtin_amdgcn_alignbyte(b1, b0, sb) & mk[0]; x[1] = __builtin_amdgcn_alignbyte(b2, b1, sb) & mk[1];
                    x[2] = __builtin_amdgcn_alignbyte(b3, b2, sb) & mk[2]; x[3] = __builtin_amdgcn_alignbyte(b0, b3, sb) & mk[3];
                };
                rot16(w, rmask[0]);
                if (ACC2) rot16(w1, rmask[NACC - 1]);
            }
#if defined(MT_DBL_ATOM)   /* timing experiment: twice the LDS atomics, same sums */
#define MT_ATOMIC_ADD(p, v) { atomicAdd(p, (v) - 1ull); atomicAdd(p, 1ull); }
#else
#define MT_ATOMIC_ADD(p, v) atomicAdd(p, v)
#endif
#define MT_ATOM(A, W, j) { unsigned long long* p_ = reinterpret_cast<unsigned long long*>(hb + cj[A][j] + (int)(((W[(j) >> 2] >> (8 * ((j) & 3))) & 0xFFu) << (ROT ? rot_sh : (int)((shp[A] >> (4 * (j))) & 15ull)))); \
                           MT_ATOMIC_ADD(p_, gq); MT_ATOMIC_ADD(p_ + hdelta, hq); }
            if (ROT) {        // every lane has sixteen slots per chunk (absent ones go to its dummy slot)
                MT_ATOM(0, w, 0); MT_ATOM(0, w, 1); MT_ATOM(0, w, 2); MT_ATOM(0, w, 3); MT_ATOM(0, w, 4); MT_ATOM(0, w, 5); MT_ATOM(0, w, 6); MT_ATOM(0, w, 7);
                MT_ATOM(0, w, 8); MT_ATOM(0, w, 9); MT_ATOM(0, w, 10); MT_ATOM(0, w, 11); MT_ATOM(0, w, 12); MT_ATOM(0, w, 13); MT_ATOM(0, w, 14); MT_ATOM(0, w, 15);
                if (ACC2) {
                    MT_ATOM(NACC - 1, w1, 0); MT_ATOM(NACC - 1, w1, 1); MT_ATOM(NACC - 1, w1, 2); MT_ATOM(NACC - 1, w1, 3); MT_ATOM(NACC - 1, w1, 4); MT_ATOM(NACC - 1, w1, 5);
                    MT_ATOM(NACC - 1, w1, 6); MT_ATOM(NACC - 1, w1, 7); MT_ATOM(NACC - 1, w1, 8); MT_ATOM(NACC - 1, w1, 9); MT_ATOM(NACC - 1, w1, 10); MT_ATOM(NACC - 1, w1, 11);
                    MT_ATOM(NACC - 1, w1, 12); MT_ATOM(NACC - 1, w1, 13); MT_ATOM(NACC - 1, w1, 14); MT_ATOM(NACC - 1, w1, 15);
                }
            } else
            if (nfeat >= 15) {   // the common shapes (full chunk, or 15 features): no per-feature branches
                MT_ATOM(0, w, 0); MT_ATOM(0, w, 1); MT_ATOM(0, w, 2); MT_ATOM(0, w, 3); MT_ATOM(0, w, 4); MT_ATOM(0, w, 5); MT_ATOM(0, w, 6); MT_ATOM(0, w, 7);
                MT_ATOM(0, w, 8); MT_ATOM(0, w, 9); MT_ATOM(0, w, 10); MT_ATOM(0, w, 11); MT_ATOM(0, w, 12); MT_ATOM(0, w, 13); MT_ATOM(0, w, 14);
                if (nfeat == 16) MT_ATOM(0, w, 15);
            } else {
#pragma unroll
                for (int j = 0; j < 14; ++j) if (j < nfeat) MT_ATOM(0, w, j);
            }
            if (!ROT && ACC2) {
                if (nfeat1 >= 15) {
                    MT_ATOM(NACC - 1, w1, 0); MT_ATOM(NACC - 1, w1, 1); MT_ATOM(NACC - 1, w1, 2); MT_ATOM(NACC - 1, w1, 3); MT_ATOM(NACC - 1, w1, 4); MT_ATOM(NACC - 1, w1, 5);
                    MT_ATOM(NACC - 1, w1, 6); MT_ATOM(NACC - 1, w1, 7); MT_ATOM(NACC - 1, w1, 8); MT_ATOM(NACC - 1, w1, 9); MT_ATOM(NACC - 1, w1, 10); MT_ATOM(NACC - 1, w1, 11);
                    MT_ATOM(NACC - 1, w1, 12); MT_ATOM(NACC - 1, w1, 13); MT_ATOM(NACC - 1, w1, 14);
                    if (nfeat1 == 16) MT_ATOM(NACC - 1, w1, 15);
                } else {
#pragma unroll
                    for (int j = 0; j < 14; ++j) if (j < nfeat1) MT_ATOM(NACC - 1, w1, j);
                }
            }
#undef MT_ATOM
#undef MT_ATOMIC_ADD
        }
    };
    auto run_batch = [&](int nb) __attribute__((always_inline)) {
        run_batch_of(wave, r_head, nb);
        r_head = (r_head + nb) & (MT_RING - 1); r_cnt -= nb;
    };

    if (SPEC && wave >= NPROD) {
        // ---- consumer wave: the batches of the rings cw, cw + NCONS, ...  A producer publishes how many entries it has appended
        // (rsync[ring]) after writing them, this wave how many it has consumed (rsync[16 + ring]); LDS operations of one wave are
        // carried out in order, so an entry is visible before the count that covers it.
        const int cw = wave - NPROD;
        constexpr int NR = SPEC ? (NPROD + MT_CONSUMERS - 1) / MT_CONSUMERS : 1;
        uint32_t head[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) head[i] = 0u;
        for (;;) {
            bool any = false, all_done = true;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int rid = cw + i * MT_CONSUMERS;
                if (rid >= NPROD) continue;
                const uint32_t done = (uint32_t)__builtin_amdgcn_readfirstlane((int)RS_LOAD(32 + rid));     // read BEFORE the count: done => the count is final
                asm volatile("" ::: "memory");
                const uint32_t tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)RS_LOAD(rid));
                asm volatile("" ::: "memory");
                uint32_t avail = tail - head[i];
                while (avail >= 64u) { run_batch_of(rid, (int)head[i], 64); head[i] += 64u; avail -= 64u; any = true; }
                if (done && avail > 0u) { run_batch_of(rid, (int)head[i], (int)avail); head[i] += avail; avail = 0u; any = true; }
                if (any) { asm volatile("" ::: "memory"); if (lane == 0) RS_STORE(16 + rid, head[i]); }
                if (!done || avail > 0u) all_done = false;
            }
            if (all_done) break;
            if (!any) __builtin_amdgcn_s_sleep(4);
        }
    }

    uint32_t p_tail = 0u, p_head_seen = 0u;                       // producer of a wave-specialised pass: entries appended / known to be consumed
    const long long N = c.N, NS = c.NS, NG = c.NG;
    const long long nwt_all = (N + MT_WT_ROWS - 1) / MT_WT_ROWS;
    const long long wt_lo = nwt_all * rb / c.gx, wt_hi = nwt_all * (rb + 1) / c.gx;
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(rec);
    const uint4* rec_acc = rec + (long long)ch * N;                                  // the chunk whose features are accumulated
    typedef float v4f __attribute__((ext_vector_type(4)));

    // ---- the three stages of a step = (wave tile of 256 rows: 4 consecutive rows per lane, class tree kk of the workgroup)
    // records of a wave tile (+ the bag bits of its rows); rows past the end of the table read the last row and are masked out later
    auto load_rec = [&](long long wt, uint4 (&ra)[4], uint4 (&r1)[4], uint32_t& bagmask) __attribute__((always_inline)) {
        const long long tile0 = wt * MT_WT_ROWS;                 // scalar
        bagmask = 0xFu;
        if (BAG) bagmask = 0u;
        const long long row0 = tile0 + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long long rr = row0 + j; if (rr >= N) rr = N - 1;
            ra[j] = rec_acc[rr];
            if (ACC2 || (route && NCHR == 2)) r1[j] = rec[N + rr]; else r1[j] = make_uint4(0, 0, 0, 0);
            if (BAG) bagmask |= (inbag[rr] ? 1u : 0u) << j;
        }
    };
    // node ids and (g, h) of the tile's rows in class tree kk: read once per level, so non-temporal (they must not push the records --
    // which the other class tree groups of the row block re-read -- out of the XCD's L2).  The rows of both arrays are padded to whole
    // wave tiles (NS, NG), so the loads need neither a bounds check nor an alignment case: straight-line code, exact vmcnt bookkeeping.
    auto load_tree = [&](long long wt, int kk, uint32_t& n4, float4& g0, float4& g1) __attribute__((always_inline)) {
        // (the workgroup's class trees are not in class order when some of them are sparse: the class comes from the tree's packed entry)
        const long long tile0 = wt * MT_WT_ROWS, kq = SPARSE ? (long long)(tree_entry(kk).y >> 16) : (long long)(k0 + kk);      // scalar
        const unsigned lo = (unsigned)lane * 4u;
        n4 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(node + kq * NS + tile0 + lo));
        const float2* gp = gh + kq * NG + tile0 + lo;
        const v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(gp)), b2 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(gp + 2));
        g0 = make_float4(a.x, a.y, a.z, a.w); g1 = make_float4(b2.x, b2.y, b2.z, b2.w);
    };
    // table indices of the four rows -> LDS reads of their entries.  A routing launch indexes by the node id, a later launch by id - child_first; whatever lies
    // past the end of the class tree's table -- LV_INACTIVE, which also pads the node-id arrays behind the last row and fills the idle lanes of a sparse
    // batch -- is clamped to the dummy entry: no "is the row in a node of this level" test, no row mask, no select.
    auto lookup = [&](uint32_t n4, const uint2 tq, uint2 (&e)[4]) __attribute__((always_inline)) {
        const uint32_t tq0 = tq.x, tq1 = tq.y;       // scalar (tree_entry)
        const uint32_t base = tq0 & 0xFFu, tabn = (tq0 >> 17) & 0xFFu, rt_off = tq1 & 0xFFFFu;
        const uint2* rtk = rt + rt_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t id = (n4 >> (8 * j)) & 0xFFu;
            if (!ROUTE) id -= base;                  // (unsigned: an id below the children wraps around and is clamped as well)
            e[j] = rtk[id < tabn ? id : tabn];
        }
    };
    // ---- stage C of a step, in two halves: route4 (pure VALU: where do the four rows go, which of them fall into a built child) and append4 (the built rows
    // go to the wave's ring; in the plain pass full batches leave it as histogram updates).  Between the two the flat pipeline issues the route
    // lookups of the NEXT step into the same registers the entries of this step just left.
    // route4 (routing launch): liv[j] = workgroup-local built slot of row j's child, 0xFF = not built; the new node ids are stored
    auto route4 = [&](const long long row0 /* first of the lane's four rows */, const uint4 (&ra)[4], const uint4 (&r1)[4], uint32_t n4, const uint2 (&e)[4], const uint2 tq,
                      uint32_t (&liv)[4]) __attribute__((always_inline)) {
        uint32_t selv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // every row takes the entry it looked up: a row outside the level's nodes got an entry that routes to itself (or the dummy) and has no built slot
            const uint32_t ex = e[j].x, ey = e[j].y;
            unsigned bin;
            if (NCHR == 0) {
                bin = 0u;
                if (ex & (1u << 24)) { const unsigned f = ex & 0xFFu; bin = rec8[((long long)(f >> 4) * N + row0 + j) * 16 + (f & 15u)]; }
            } else {
                uint32_t rx = ra[j].x, ry = ra[j].y, rz = ra[j].z, rw = ra[j].w;
                if (NCHR == 2) { const bool second = (ex & (1u << 30)) != 0u; rx = second ? r1[j].x : rx; ry = second ? r1[j].y : ry; rz = second ? r1[j].z : rz; rw = second ? r1[j].w : rw; }
                const bool hi = (int32_t)ex < 0;
                const uint32_t lo32 = hi ? rz : rx, hi32 = hi ? rw : ry;
                bin = __builtin_amdgcn_perm(hi32, lo32, ex) & 0xFFu;      // selector byte 0 = f & 7: byte (f & 15) of the record (the other result bytes are not used)
            }
            // left = (bin == nan bin) ? default-left : (bin <= theta), as one byte compare (the route entry holds off and thr: see the table fill)
            const bool left = ((bin - ((ex >> 16) & 0xFFu)) & 0xFFu) <= ((ex >> 8) & 0xFFu);
            selv[j] = left ? (ey & 0xFFFFu) : (ey >> 16);          // child in bits 0..7, workgroup-local built slot in bits 8..15
            liv[j] = selv[j] >> 8;
        }
        // the four child bytes side by side
        const uint32_t out4 = __builtin_amdgcn_perm(selv[1], selv[0], 0x0C0C0400u) | __builtin_amdgcn_perm(selv[3], selv[2], 0x04000C0Cu);
        if (out4 != n4) __builtin_nontemporal_store(out4, reinterpret_cast<uint32_t*>(node + (long long)(tq.y >> 16) * NS + row0));
    };
    auto append4 = [&](const uint4 (&ra)[4], const uint4 (&r1)[4], uint32_t bagmask, const float4 g0, const float4 g1, const uint32_t (&liv)[4]) __attribute__((always_inline)) {
        const float gg[4] = {g0.x, g0.z, g1.x, g1.z}, hh[4] = {g0.y, g0.w, g1.y, g1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned li = liv[j];
            bool built = li != 0xFFu;
            if (BAG) built = built && ((bagmask >> j) & 1u);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(built);
            if (m != 0ull) {                                      // uniform
                if (SPEC) {   // room for this row step's entries?  (the consumer is at most one batch behind a full ring)
                    const uint32_t n_new = (uint32_t)__popcll(m);
                    while (p_tail + n_new - p_head_seen > (uint32_t)MT_RING) {
                        asm volatile("" ::: "memory");
                        if (lane == 0) RS_STORE(wave, p_tail);        // the consumer must see the entries of this step's earlier rows to make room
                        p_head_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)RS_LOAD(16 + wave));
                        if (p_tail + n_new - p_head_seen > (uint32_t)MT_RING) __builtin_amdgcn_s_sleep(2);
                    }
                }
                // a ring smaller than 128 entries (MT_RING_N=64: 26 KB more LDS for histograms) may not hold this row step's entries next to the
                // waiting ones: those leave first, as a partial batch (never taken with 128 entries: <= 63 wait, <= 64 arrive)
                if (!SPEC && MT_RING < 128 && r_cnt + (int)__popcll(m) > MT_RING) run_batch(r_cnt);
                if (built) {
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));     // built lanes below this one
                    const int pos = SPEC ? (int)((p_tail + below) & (uint32_t)(MT_RING - 1)) : (r_head + r_cnt + (int)below) & (MT_RING - 1);
                    uint4 q0 = ra[j], q1 = r1[j];
                    // (one v_perm: bytes 0..2 of the word, the slot as byte 3)
                    if (li_in_rec) { if (ACC2) q1.w = __builtin_amdgcn_perm(li, q1.w, 0x04020100u); else q0.w = __builtin_amdgcn_perm(li, q0.w, 0x04020100u); }
                    else ring_li[pos] = (uint16_t)li;
                    ring_rec[pos] = q0;
                    if (ACC2) ring_rec1[pos] = q1;
                    ring_gh[pos] = make_uint2(__float_as_uint(gg[j]), __float_as_uint(hh[j]));
                }
                if (SPEC) p_tail += (uint32_t)__popcll(m);
                else { r_cnt += (int)__popcll(m); if (r_cnt >= 64) run_batch(64); }
            }
        }
        if (SPEC) { asm volatile("" ::: "memory"); if (lane == 0) RS_STORE(wave, p_tail); }     // publish (after the entries: in-order LDS)
    };
    // both halves back to back (the batches of the sparse sweep)
    auto stage_c = [&](const long long row0, const uint4 (&ra)[4], const uint4 (&r1)[4], uint32_t bagmask, uint32_t n4, const float4 g0, const float4 g1,
                       const uint2 (&e)[4], const uint2 tq) __attribute__((always_inline)) {
        if ((tq.x >> 31) == 0u) return;           // (scalar) the class tree is finished or has nothing to split at this level
        uint32_t liv[4];
        if (ROUTE) route4(row0, ra, r1, n4, e, tq, liv);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) liv[j] = e[j].y & 0xFFu;
        }
        append4(ra, r1, bagmask, g0, g1, liv);
    };

    // ---- lock-step of the class-tree groups of a row block (round 5; wave-specialised pass; off by default).  Every group's workgroup reads the block's bin
    // records; they run on one XCD at the same time (launch order), but nothing keeps them at the same place: on the 100M x 32 shape the groups
    // drifted further apart than the XCD's 4 MB L2 holds and a level pass moved 25.5 GB for 8.3 GB of (node id, g, h) stream and
    // 3.2 GB of records (profiles/traffic.json, round 4).  Wave 0 of a workgroup publishes the tile round it is in (a word per
    // (row block, group), tagged with the launch's epoch), and every producer wave looks at the block's words every MT_LOCK_EVERY
    // rounds: it sleeps while the slowest group that HAS STARTED in this launch and has not finished is more than mt_window rounds
    // behind.  The slowest started group never waits, so the wait ends; a group that is not resident yet is not waited for.
    // Measured: -63 % fetch, +23 % time (profiles/r5e_*): the pass is not bound by that traffic.
    const bool lock = SPEC && c.mt_window > 0 && prog != nullptr && c.mt_G > 1;
    uint32_t* prog_rb = prog + (size_t)rb * (size_t)c.mt_G;
    const uint32_t ep_tag = c.mt_epoch << 20;
    auto lock_step = [&](const uint32_t round) __attribute__((always_inline)) {
        if ((round & (MT_LOCK_EVERY - 1)) != 0u) return;
        if (wave == 0 && lane == 0) __hip_atomic_store(prog_rb + grp, ep_tag | (round < 0xFFFFEu ? round + 1u : 0xFFFFEu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; spin < 4096; ++spin) {        // (bounded: a lost update can cost time, never the pass)
            uint32_t slow = 0xFFFFFFFFu;
            for (int g0 = 0; g0 < c.mt_G; g0 += 64) {
                uint32_t v = 0xFFFFFFFFu;
                if (g0 + lane < c.mt_G) {
                    const uint32_t w = __hip_atomic_load(prog_rb + g0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((w >> 20) == c.mt_epoch && (w & 0xFFFFFu) != 0xFFFFFu) v = w & 0xFFFFFu;     // started in this launch, not finished
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { const uint32_t v2 = (uint32_t)__shfl_xor((int)v, o); v = v2 < v ? v2 : v; }
                slow = v < slow ? v : slow;
            }
            if (slow == 0xFFFFFFFFu || round + 1u <= slow + (uint32_t)c.mt_window) break;
            __builtin_amdgcn_s_sleep(32);
        }
    };

        // ---- routing launch: ONE software pipeline over all steps of the wave (tile-major, class trees inside).  What bounds the pass is the number of
        // bytes a CU keeps in flight (round 6: a wave's loads were waited for at the end of the step that issued them -- the register copies of a
        // rotating pipeline need the loaded values -- so a CU held ~1 step x 16 waves = 37 KB in flight for half of the time: 2.2 TB/s).  Here the
        // three register sets of (node ids, g, h) rotate by NAME (the loop body is written three times), so a set loaded in step q is first touched
        // in step q + 1 (node ids: the route lookup) / q + 2 (g, h): two whole steps in flight, also across wave tiles; the records of the next tile
        // are requested when the last class tree of a tile has been routed.
    if (wave < NPROD) {
        const int nkw = nkd;          // class trees walked tile by tile (the sparse ones are swept below, the finished ones not at all)
        {
            // (the wave index through readfirstlane: tile numbers, step counters and everything derived from them are then SCALAR -- as a VGPR value
            // the compiler keeps the whole step bookkeeping in 64-bit VALU operations)
            const long long my_first = wt_lo + __builtin_amdgcn_readfirstlane(wave);
            const long long ntile_w = (nkw > 0 && my_first < wt_hi) ? (wt_hi - my_first + NPROD - 1) / NPROD : 0;
            const long long Q = ntile_w * nkw;
            if (Q > 0) {
                struct TS { uint32_t n4; float4 g0, g1; };
                TS S0, S1, S2;
                uint2 e_cur[4];
                bool cross = false;                                                               // (uniform) the step about to start is the first of a wave tile: its records wait in rn
                long long wt_c = my_first, wt_n = my_first, wt_l = my_first; int kk_c = 0, kk_n = 0, kk_l = 0;     // steps q, min(q + 1, Q - 1), min(q + 2, Q - 1)
                auto fwd = [&](long long& wt, int& kk) __attribute__((always_inline)) { if (++kk == nkw) { kk = 0; wt += NPROD; } };
                if (Q > 1) fwd(wt_n, kk_n);
                wt_l = wt_n; kk_l = kk_n;
                if (Q > 2) fwd(wt_l, kk_l);
                uint4 ra[4], r1[4], rn[4], r1n[4]; uint32_t bagmask, bagmask_n = 0u;
                load_rec(wt_c, ra, r1, bagmask);
#pragma unroll
                for (int j = 0; j < 4; ++j) { rn[j] = ra[j]; r1n[j] = r1[j]; }
                load_tree(wt_c, kk_c, S0.n4, S0.g0, S0.g1);
                load_tree(wt_n, kk_n, S1.n4, S1.g0, S1.g1);
                lookup(S0.n4, tree_entry(kk_c), e_cur);
                long long q = 0;
                auto step = [&](TS& cur, TS& nxt, TS& in) __attribute__((always_inline)) {
                    if (SPEC && lock && kk_c == 0) lock_step((uint32_t)((wt_c - my_first) / NPROD));
                    load_tree(wt_l, kk_l, in.n4, in.g0, in.g1);                              // step q + 2 (past the end: the last step once more, never used)
                    if (cross) {      // (AFTER this step's loads have been requested: the wait in front of these copies then lets those three stay in flight -- in front
                                      // of them it was a vmcnt(0) at every tile boundary, i.e. at every step or second step of the deep levels, where T is 1 or 2)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { ra[j] = rn[j]; r1[j] = r1n[j]; }
                        bagmask = bagmask_n;
                    }
                    if (kk_c == 0 && wt_c + NPROD < wt_hi) load_rec(wt_c + NPROD, rn, r1n, bagmask_n);     // (uniform) first step of a wave tile: the records of the wave's next tile
                    const uint2 tq_c = tree_entry(kk_c);
                    const bool live = (tq_c.x >> 31) != 0u;                                  // (scalar) else: the class tree is finished or has nothing to split at this level
                    uint32_t liv[4] = {0xFFu, 0xFFu, 0xFFu, 0xFFu};
                    if (ROUTE) { if (live) route4(wt_c * MT_WT_ROWS + lane * 4, ra, r1, cur.n4, e_cur, tq_c, liv); }
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) liv[j] = e_cur[j].y & 0xFFu;
                    }
                    lookup(nxt.n4, tree_entry(kk_n), e_cur);                                  // step q + 1, into the registers this step's entries just left
                    if (live) append4(ra, r1, bagmask, cur.g0, cur.g1, liv);
                    cross = wt_n != wt_c;
                    wt_c = wt_n; kk_c = kk_n; wt_n = wt_l; kk_n = kk_l;
                    if (q + 3 < Q) fwd(wt_l, kk_l);
                    ++q;
                };
                while (q < Q) {
                    step(S0, S1, S2);
                    if (q < Q) step(S1, S2, S0);
                    if (q < Q) step(S2, S0, S1);
                }
            }
        }
    }
    if (!SPEC) {
        // ---- sparse sweep: the class trees [nkd, nke) of the workgroup, whose expanded parents hold < 1/MT_SPARSE_DIV of the rows.  Per class tree the wave
        // streams ONLY the node ids of its rows (1 B per row: sixteen rows per lane and step), tests them against the tree's 64-bit mask of
        // live nodes and collects the 4-row groups that hold a live row in a REGISTER of the wave (lanes [0, sp_cnt) hold pending groups; new
        // ones are pushed to the next free lanes with one ds_permute: no LDS memory -- 8 KB of rings cost the K = 64 passes 5 % through the
        // histograms' replication, profiles/r05d_*); 64 such groups are one dense step: every lane
        // fetches its group's records, node ids and (g, h) and goes through the same lookup + route + append code as the row loop.  A pass
        // over a class tree with 1 % live rows costs its node-id stream instead of records + (g, h) + ~330 instructions per 256 rows.
        if (SPARSE && nkd < nke) {
            const long long row_lo = wt_lo * MT_WT_ROWS, row_hi = (wt_hi * MT_WT_ROWS < N) ? wt_hi * MT_WT_ROWS : N;
            const long long nst = (wt_hi * MT_WT_ROWS - row_lo + 1023) / 1024;        // super tiles of 1024 rows (the block's rows are whole wave tiles)
            for (int kk = nkd; kk < nke; ++kk) {
                const uint32_t tq0 = (uint32_t)__builtin_amdgcn_readlane((int)tpk_v.x, kk < 64 ? kk : 63), tq1 = (uint32_t)__builtin_amdgcn_readlane((int)tpk_v.y, kk < 64 ? kk : 63);
                const uint2 tq = make_uint2(tq0, tq1);
                const uint32_t base = tq0 & 0xFFu, nlev = (tq0 >> 8) & 0x1FFu;
                const unsigned long long xm_v = xmask[kk];
                const unsigned long long xm = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xm_v) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(xm_v >> 32)) << 32;
                const uint8_t* nd_k = node + (long long)(tq1 >> 16) * NS;
                const float2* gh_k = gh + (long long)(tq1 >> 16) * NG;
                int sp_cnt = 0;                                       // wave-uniform: lanes [0, sp_cnt) hold a pending group in `pend`
                uint32_t pend = 0u;
                auto sparse_batch = [&](int nb) __attribute__((always_inline)) {      // all pending groups: nb = sp_cnt (<= 64)
                    const bool on = lane < nb;
                    const uint32_t g = pend;
                    const long long row0 = on ? (long long)g * 4 : row_lo;
                    uint4 ra[4], r1[4]; uint32_t bagmask = BAG ? 0u : 0xFu;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        long long rr = row0 + j; if (rr >= N) rr = N - 1;
                        ra[j] = rec_acc[rr]; r1[j] = make_uint4(0, 0, 0, 0);
                        if (BAG) bagmask |= (inbag[rr] ? 1u : 0u) << j;
                    }
                    const uint32_t n4 = on ? *reinterpret_cast<const uint32_t*>(nd_k + row0) : 0xFFFFFFFFu;     // (idle lanes: LV_INACTIVE, the dummy entry)
                    const float4 g0 = *reinterpret_cast<const float4*>(gh_k + row0), g1 = *reinterpret_cast<const float4*>(gh_k + row0 + 2);
                    uint2 e[4];
                    lookup(n4, tq, e);
                    stage_c(row0, ra, r1, bagmask, n4, g0, g1, e, tq);
                    sp_cnt = 0;
                };
                auto load16 = [&](long long st) __attribute__((always_inline)) -> uint4 {
                    const long long r16 = row_lo + st * 1024 + lane * 16;
                    uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                    if (st < nst && r16 < row_hi) v = *reinterpret_cast<const uint4*>(nd_k + r16);      // (rows are padded to whole wave tiles: NS)
                    return v;
                };
                uint4 n16 = load16(wave);
                for (long long st = wave; st < nst; st += WAVES) {
                    const long long r16 = row_lo + st * 1024 + lane * 16;          // this lane's sixteen rows
                    const uint4 nxt = load16(st + WAVES);                          // (requested one step ahead)
                    const long long left = row_hi - r16;
                    const int nvalid = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
                    const uint32_t w4[4] = {n16.x, n16.y, n16.z, n16.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        bool hit = false;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t idx = ((w4[d] >> (8 * j)) & 0xFFu) - base;
                            hit = hit || (idx < nlev && ((xm >> (idx & 63u)) & 1ull) != 0ull && d * 4 + j < nvalid);
                        }
                        const unsigned long long m = __ballot(hit);
                        if (m != 0ull) {                                  // uniform: every lane takes part in the permute
                            // one permutation of the 64 lanes: the np hit lanes push their group to the lanes sp_cnt .. sp_cnt + np - 1 (mod 64), the
                            // others fill the rest (their values are never taken)
                            const int np = (int)__popcll(m), r_hit = (int)__popcll(m & lane_lt);
                            const int dest = hit ? sp_cnt + r_hit : sp_cnt + np + (lane - r_hit);
                            const uint32_t rx = (uint32_t)__builtin_amdgcn_ds_permute((dest & 63) << 2, (int)(uint32_t)((r16 + d * 4) >> 2));
                            const int rel = (lane - sp_cnt) & 63;         // lane receives the rel-th new group if rel < np
                            if (sp_cnt + np < 64) { if (rel < np) pend = rx; sp_cnt += np; }
                            else {
                                const int rem = sp_cnt + np - 64;          // groups that wrapped around to the lanes [0, rem)
                                if (rel < 64 - sp_cnt) pend = rx;
                                sparse_batch(64);
                                if (lane < rem) pend = rx;
                                sp_cnt = rem;
                            }
                        }
                    }
                    n16 = nxt;
                }
                if (sp_cnt > 0) sparse_batch(sp_cnt);
            }
        }
    }
    if (SPEC) { if (wave < NPROD) { asm volatile("" ::: "memory"); if (lane == 0) { RS_STORE(wave, p_tail); asm volatile("" ::: "memory"); RS_STORE(32 + wave, 1u); } }
                if (c.mt_window > 0 && prog != nullptr && wave == 0 && lane == 0)      // lock-step: this group is done with the block (nobody waits for it any more)
                    __hip_atomic_store(prog + (size_t)rb * (size_t)c.mt_G + grp, (c.mt_epoch << 20) | 0xFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#undef RS_LOAD
#undef RS_STORE
    else while (r_cnt > 0) run_batch(r_cnt < 64 ? r_cnt : 64);
    __syncthreads();
    // ---- flush this workgroup's partial histograms (plain stores: no global atomics, no zeroing) and the built-row counts
    constexpr int NQ = NACC * 16;
    for (int i = tid; i < total * wb; i += THREADS) {
        const int ln = i / wb, b = i - ln * wb;
        const MtTree t = ti[nd_tree[ln]];
        int j = 0;                                                  // accumulated feature of wide bin b: the last one whose first wide bin is <= b
#pragma unroll
        for (int q = 1; q < NQ; ++q) j += (b >= ftab[q]) ? 1 : 0;
        const int shb = ftab[64 + j], s0 = ftab[32 + j] + ((b - ftab[j]) << shb);
        long long tg = 0, th = 0;
        const unsigned long long* sg_ = hist_g + (size_t)ln * spn + s0; const unsigned long long* sh_ = hist_h + (size_t)ln * spn + s0;
        for (int r2 = 0; r2 < (1 << shb); ++r2) { tg += (long long)sg_[r2]; th += (long long)sh_[r2]; }
        HistBin o; o.g = tg; o.h = th;
        part[(((long long)t.k * c.gx + rb) * c.max_built + (c.mt_slot0 + ln - t.slot0)) * c.totbins + ftab[96 + j] + b] = o;
    }
    if (ch == 0) {   // exact row counts of the built children (k_level_plan numbers the children of parent ei as child_first + 2 ei, + 1)
        for (int ln = tid; ln < total; ln += THREADS) {
            int tot = 0;
            for (int r2 = 0; r2 < MT_CNT_REP; ++r2) tot += cnt[ln * MT_CNT_REP + r2];
            const MtTree t = ti[nd_tree[ln]];
            const int ei = c.mt_slot0 + ln - t.slot0;
            if (tot) atomicAdd(&count[(long long)t.k * 256 + t.child_first + 2 * ei + (plan[t.k].built_is_left[ei] ? 0 : 1)], tot);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_level_reduce: sum the gx workgroup partials of every built child into one compact [K][nb][totbins] buffer.
// Used (a) when a class tree has many workgroups (binary / few-class targets: gx up to 256 -- summing them inside
// the one-wave-per-feature split kernel took longer than the pass itself) and (b) for row-sharded multi-GPU
// training, where the compact buffer is what gets all-reduced (exact integer sums) across the ranks.
// grid (ceil(totbins/64), nb, K), block 256 = 64 bins x 4 partial-lanes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_level_reduce(const HistBin* __restrict__ part, HistBin* __restrict__ red, const LvPlan* __restrict__ plan,
                                                      const int32_t* __restrict__ count, int is_root, int nb, LevelConst c) {
    __shared__ long long sg[4][64], sh[4][64];
    const int k = blockIdx.z, bslot = blockIdx.y, bl = threadIdx.x & 63, xl = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + bl;
    // the local child counts ride in the same buffer (row-sharded: same all-reduce), as int64 words behind the histograms
    if (blockIdx.x == 0 && blockIdx.y == 0)
        reinterpret_cast<long long*>(red + (long long)c.K * nb * c.totbins)[(long long)k * 256 + threadIdx.x] = (long long)count[(long long)k * 256 + threadIdx.x];
    const LvPlan* pp = &plan[k];
    const int n_built = pp->done ? 0 : (is_root ? 1 : pp->n_built);
    long long ag = 0, ah = 0;
    if (b < c.totbins && bslot < n_built) {
        const HistBin* src = part + (((long long)k * c.gx) * c.max_built + bslot) * c.totbins + b;
        const long long xs = (long long)c.max_built * c.totbins;
#pragma unroll 8
        for (int x = xl; x < c.gx; x += 4) { const HistBin v = src[x * xs]; ag += v.g; ah += v.h; }
    }
    sg[xl][bl] = ag; sh[xl][bl] = ah;
    __syncthreads();
    if (xl == 0 && b < c.totbins) {
        HistBin acc; acc.g = sg[0][bl] + sg[1][bl] + sg[2][bl] + sg[3][bl]; acc.h = sh[0][bl] + sh[1][bl] + sh[2][bl] + sh[3][bl];
        red[((long long)k * nb + bslot) * c.totbins + b] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Joint bins for the root pass.  The root pass is bound by the LDS-atomic rate: one packed atomic per feature and row
// (15 for the synthetic table).  Features with few bins are therefore COMBINED: a group of features whose bin counts
// multiply to <= 256 shares one byte of a second, "joint" record (code = sum of bin_f * stride_f) and one histogram of
// prod(nbins) joint bins, so a row costs one atomic per GROUP (7 instead of 15).  The sums are exact integers, so the
// histogram of every real feature is recovered exactly as a marginal of its group's joint histogram; that happens here,
// in the reduction over the workgroup partials that the root pass needs anyway.  Only the root pass uses the joint record
// (one built node: the joint histograms fit the LDS with room for replication); routing, the level passes and the
// predictor keep the plain record.
// ------------------------------------------------------------------------------------------------
struct JointFeat { int32_t voff /* offset of the group's joint histogram */, stride, nbins /* of this feature */, nbv /* joint bins of the group */, hoff, vbyte /* byte of the joint record */, pad0, pad1; };

// joint record of every row from its plain bin record(s); thread per row
constexpr int JOINT_WIDE_CAP = 1024;     // joint bins of a group with 16-bit codes (k_level_root<true>)
__global__ __launch_bounds__(256) void k_pack_joint(const uint4* __restrict__ rec, long long N, int F, const JointFeat* __restrict__ jf, uint4* __restrict__ rec_joint, int wide,
                                                    int mult_chunk = -1 /* >= 0: byte 15 of that chunk's record holds the row's multiplicity and moves to byte 15 of the joint record */) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(rec);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int f = 0; f < F; ++f) {
        const uint32_t bin = rec8[((long long)(f >> 4) * N + i) * 16 + (f & 15)];
        const int vb = jf[f].vbyte;
        if (wide) w[vb >> 1] += (bin * (uint32_t)jf[f].stride) << (16 * (vb & 1));   // (a group's code stays below 65536)
        else w[vb >> 2] += (bin * (uint32_t)jf[f].stride) << (8 * (vb & 3));    // a group's code stays below 256: no carry into the next byte
    }
    if (mult_chunk >= 0) w[3] = (w[3] & 0x00FFFFFFu) | ((uint32_t)rec8[((long long)mult_chunk * N + i) * 16 + 15] << 24);
    rec_joint[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

// After k_level_reduce has summed the workgroup partials of a root pass over the joint record (same kernel, joint bin space):
// the histogram of every real feature is the marginal of its group's joint histogram.  grid (ceil(totbins / 256), K), block 256:
// one thread per real bin, <= 256 / nbins terms each.
__global__ __launch_bounds__(256) void k_level_marginal(const HistBin* __restrict__ red_j /* [K][vtotbins] */, HistBin* __restrict__ red, const LvPlan* __restrict__ plan,
                                                        const int32_t* __restrict__ count, const JointFeat* __restrict__ jf, const int16_t* __restrict__ bin_feat /* [totbins] */,
                                                        int vtotbins, LevelConst c) {
    const int k = blockIdx.y;
    if (blockIdx.x == 0)   // the local child counts ride behind the histograms (k_level_reduce does the same)
        reinterpret_cast<long long*>(red + (long long)c.K * c.totbins)[(long long)k * 256 + threadIdx.x] = (long long)count[(long long)k * 256 + threadIdx.x];
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= c.totbins) return;
    HistBin acc; acc.g = 0; acc.h = 0;
    if (!plan[k].done) {
        const JointFeat f = jf[bin_feat[b]];
        const int digit = b - f.hoff;
        const int period = f.stride * f.nbins;               // joint codes with this digit: hi * period + digit * stride + lo, lo < stride
        const HistBin* src = red_j + (long long)k * vtotbins + f.voff;
        for (int base = digit * f.stride; base < f.nbv; base += period)
            for (int lo = 0; lo < f.stride; ++lo) { const HistBin v = src[base + lo]; acc.g += v.g; acc.h += v.h; }
    }
    red[(long long)k * c.totbins + b] = acc;
}

__global__ __launch_bounds__(256) void k_counts_unpack(const long long* __restrict__ cnt64, int32_t* __restrict__ count_g) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    count_g[i] = (int32_t)cnt64[i];
}

// ------------------------------------------------------------------------------------------------
// k_level_split: sum the workgroup partials of the built child, derive the sibling by subtraction,
// scan both (FindBestThreshold).  grid (ROOT ? ceil(F/4) : ceil(F/2), ROOT ? 1 : max parents, K), block 256.
// ------------------------------------------------------------------------------------------------
template <bool ROOT>
__global__ __launch_bounds__(256) void k_level_split(const HistBin* __restrict__ part, HistBin* __restrict__ pool, LvPlan* __restrict__ plan,
                                                     SNode* __restrict__ nodes, int32_t* __restrict__ count, int32_t* __restrict__ count_local,
                                                     const FeatMeta* __restrict__ fmeta, const uint8_t* __restrict__ used_all /* [NE][K][F] */,
                                                     Cand* __restrict__ cand /* [K][256][F] */, unsigned long long* __restrict__ stat_rows,
                                                     const int32_t* __restrict__ itp /* device-side iteration counter */,
                                                     int n_hnodes, const FxScale* __restrict__ fxs, TrainConst c_model, LevelConst lc) {
    const int k = blockIdx.z, pi = blockIdx.y;
    const TrainConst c = tree_const(c_model, fxs, k);          // sums -> doubles on this class tree's grid
    const uint8_t* used = used_all + (long long)(*itp) * c.K * c.F;
    // ROOT: one wave per feature (4 per block).  Otherwise one wave per (feature, child side): 2 features per block,
    // so the two FindBestThreshold scans of a parent run side by side instead of back to back.
    const int wv = threadIdx.x >> 6;
    const int f = ROOT ? blockIdx.x * 4 + wv : blockIdx.x * 2 + (wv >> 1);
    const int side = wv & 1;   // 0 = left child, 1 = right child
    const LvPlan* pp = &plan[k];
    if (pp->done) return;
    if (!ROOT && pi >= pp->n_exp) return;
    if (f >= c.F) return;
    const int lane = lane_id();
    const FeatMeta fm = fmeta[f];
    SNode* nk = nodes + (long long)k * 256;
    Cand* ck = cand + (long long)k * 256 * c.F;
    HistBin* pk = pool + (long long)k * n_hnodes * c.totbins;
    const bool is_used = used[(long long)k * c.F + f] != 0;
    long long ag[4], ah[4];
    const int bslot = ROOT ? 0 : pi;
    // built child = sum of the gx workgroup partials
#pragma unroll
    for (int j = 0; j < 4; ++j) { ag[j] = 0; ah[j] = 0; }
    for (int x = 0; x < lc.gx; ++x) {
        const HistBin* src = part + (((long long)k * lc.gx + x) * lc.max_built + bslot) * c.totbins + fm.hoff;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int b = lane * 4 + j; if (b < fm.nbins) { const HistBin v = src[b]; ag[j] += v.g; ah[j] += v.h; } }
    }
    if (ROOT) {
        HistBin* h0 = pk + fm.hoff;
        long long sg_ = 0, sh_ = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int b = lane * 4 + j; if (b < fm.nbins) { HistBin v; v.g = ag[j]; v.h = ah[j]; h0[b] = v; } sg_ += ag[j]; sh_ += ah[j]; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { sg_ += __shfl_xor(sg_, o); sh_ += __shfl_xor(sh_, o); }
        if (f == 0 && lane == 0) { nk[0].Gq = sg_; nk[0].Hq = sh_; nk[0].searched = 1; atomicAdd(stat_rows, (unsigned long long)pp->n_in); }
        if (!is_used) { if (lane == 0) ck[f].gain = -INFINITY; return; }
        scan_child(ag, ah, fm, sg_, sh_, pp->n_in, c, &ck[f]);
        return;
    }
    const int p = pp->exp[pi];
    const SNode P = nk[p];
    const int l = P.left, r = P.right;
    const bool bl = pp->built_is_left[pi] != 0;
    const int me = side == 0 ? l : r;
    const bool i_am_built = (side == 0) == bl;
    const HistBin* hp = pk + (long long)P.hslot * c.totbins + fm.hoff;
    HistBin* hm = pk + (long long)nk[me].hslot * c.totbins + fm.hoff;
    if (!i_am_built) {   // the sibling: parent - built (exact integers)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = lane * 4 + j;
            if (b < fm.nbins) { const HistBin par = hp[b]; ag[j] = par.g - ag[j]; ah[j] = par.h - ah[j]; } else { ag[j] = 0; ah[j] = 0; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int b = lane * 4 + j; if (b < fm.nbins) { HistBin v; v.g = ag[j]; v.h = ah[j]; hm[b] = v; } }
    int nl = count[(long long)k * 256 + l], nr = count[(long long)k * 256 + r];
    {
        // only the built child was counted (k_level_mt); its sibling holds the rest of the parent's rows.  The derived
        // count is published for the leaf counts; row-sharded training sums the LOCAL arrays, so exactly one rank also
        // stores it there (lc.sib_local).
        const int nb = bl ? nl : nr, ns = P.count - nb;
        if (bl) nr = ns; else nl = ns;
        if (f == 0 && lane == 0 && side == 0) {
            const int sib = bl ? r : l;
            count[(long long)k * 256 + sib] = ns;
            if (lc.sib_local && count_local != count) count_local[(long long)k * 256 + sib] = ns;
        }
    }
    // SerialTreeLearner::BeforeFindBestSplit: both children too small -> neither is searched
    const bool go = !(nr < c.min_data_in_leaf * 2 && nl < c.min_data_in_leaf * 2);
    if (f == 0 && lane == 0) {
        nk[me].count = side == 0 ? nl : nr; nk[me].searched = go ? 1 : 0;
        if (side == 0) atomicAdd(stat_rows, (unsigned long long)(bl ? nl : nr));
    }
    if (!go) return;
    if (!is_used) { if (lane == 0) ck[(long long)me * c.F + f].gain = -INFINITY; return; }
    const long long lGq = P.best.left_gq, lHq = P.best.left_hq;
    if (side == 0) scan_child(ag, ah, fm, lGq, lHq, (long long)nl, c, &ck[(long long)l * c.F + f]);
    else scan_child(ag, ah, fm, P.Gq - lGq, P.Hq - lHq, (long long)nr, c, &ck[(long long)r * c.F + f]);
}

// ------------------------------------------------------------------------------------------------
// k_level_plan: one wave per class tree, before pass `level` (which routes depth level-1 -> level).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_level_plan(LvPlan* __restrict__ plan, SNode* __restrict__ nodes,
                                                    const Cand* __restrict__ cand, const FeatMeta* __restrict__ fmeta,
                                                    int level, TrainConst c, LevelConst lc) {
    __shared__ double pm[256];
    const int k = blockIdx.x, lane = lane_id(), wave = threadIdx.x >> 6;
    LvPlan* pp = &plan[k];
    if (pp->done) return;
    SNode* nk = nodes + (long long)k * 256;
    const Cand* ck = cand + (long long)k * 256 * c.F;
    const int first = pp->lvl_first, end = pp->lvl_end, nlev = end - first;   // nodes of depth level-1 (<= 64)
    // 1. best split of every node of the level (SplitInfo::operator>: gain, then smaller feature); one wave per node, 4 at a time
    for (int n = first + wave; n < end; n += 4) {
        if (!nk[n].searched) { if (lane == 0) { nk[n].best.gain = -INFINITY; nk[n].best_feature = -1; } continue; }
        const Cand* cf = ck + (long long)n * c.F;
        double bg = -INFINITY; int bf = -1;
        for (int f = lane; f < c.F; f += 64) { const double g = cf[f].gain; if (g > -INFINITY && leaf_better(g, f, 0, bg, bf, 0)) { bg = g; bf = f; } }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const double g2 = __shfl_xor(bg, o); const int f2 = __shfl_xor(bf, o); if (leaf_better(g2, f2, 0, bg, bf, 0)) { bg = g2; bf = f2; } }
        if (lane == 0) {
            if (bf >= 0) { nk[n].best = cf[bf]; nk[n].best_feature = bf; } else { nk[n].best.gain = -INFINITY; nk[n].best_feature = -1; }
        }
    }
    __syncthreads();
    if (wave != 0) return;   // the rest is one wave's work (the barriers below only count live waves)
    // 2. path-min gains
    for (int n = lane; n < first; n += 64) pm[n] = nk[n].pmin;
    __syncthreads();
    if (lane < nlev) {
        const int n = first + lane;
        const double g = nk[n].best.gain;
        const int par = nk[n].parent;
        double v = g;
        if (par >= 0) { const double pv = pm[par]; v = (pv < g) ? pv : g; }
        if (!(g > -INFINITY)) v = -INFINITY;
        pm[n] = v; nk[n].pmin = v;
    }
    __syncthreads();
    // 3. expansion test: X is dead once num_leaves-1 known nodes are split before it
    bool expand = false;
    if (lane < nlev) {
        const double v = pm[first + lane];
        if (v > 0.0) {
            int rank = 0;
            for (int y = 0; y < end; ++y) rank += (pm[y] > v) ? 1 : 0;
            expand = rank < c.num_leaves - 1;
        }
    }
    const unsigned long long em = __ballot(expand);
    const int n_exp = __popcll(em);
    for (int i = lane; i < 256; i += 64) { pp->route0[i] = 0; pp->route1[i] = 0xFFFFFFFFu; }
    __syncthreads();
    if (n_exp == 0) { if (lane == 0) { pp->done = 1; pp->n_exp = 0; pp->n_built = 0; } return; }
    const int child_first = pp->n_nodes;
    const bool with_hist = level < c.max_depth;   // children of depth max_depth are never searched
    const int hs0 = pp->n_hslots;
    if (expand) {
        const int ei = __popcll(em & ((1ull << lane) - 1ull));
        const int n = first + lane;
        const SNode P = nk[n];
        const int l = child_first + 2 * ei, r = l + 1;
        const int f = P.best_feature;
        // hessian-estimated smaller child is the one whose histogram is built
        const bool built_left = P.best.left_hq * 2 <= P.Hq;
        nk[n].left = l; nk[n].right = r;
        SNode L; memset(&L, 0, sizeof(L));
        L.depth = P.depth + 1; L.parent = n; L.left = -1; L.right = -1; L.best_feature = -1; L.searched = 0; L.best.gain = -INFINITY; L.pmin = -INFINITY;
        SNode R = L;
        L.is_left = 1; L.Gq = P.best.left_gq; L.Hq = P.best.left_hq; L.hslot = with_hist ? hs0 + 2 * ei : -1;
        R.is_left = 0; R.Gq = P.Gq - P.best.left_gq; R.Hq = P.Hq - P.best.left_hq; R.hslot = with_hist ? hs0 + 2 * ei + 1 : -1;
        nk[l] = L; nk[r] = R;
        pp->exp[ei] = (uint8_t)n; pp->built_is_left[ei] = built_left ? 1 : 0;
        const int nanbin = fmeta[f].has_nan ? fmeta[f].V : 255;
        pp->route0[n] = (uint32_t)f | ((uint32_t)(P.best.theta + 1) & 0xFFu) << 8 | (uint32_t)nanbin << 16 | 1u << 24 | (uint32_t)(P.best.dleft ? 1 : 0) << 25;
        const uint32_t ls = with_hist && built_left ? (uint32_t)ei : 0xFFu, rs = with_hist && !built_left ? (uint32_t)ei : 0xFFu;
        pp->route1[n] = (uint32_t)l | (uint32_t)r << 8 | ls << 16 | rs << 24;
    }
    const int n_built = with_hist ? n_exp : 0;
    // rows the pass of this level has to touch: those of the expanded parents (k_level_mt sweeps a class tree whose share is small
    // through its node ids only)
    long long live = expand ? (long long)nk[first + lane].count : 0ll;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) live += __shfl_xor(live, o);
    if (lane == 0) {
        pp->live_rows = (int32_t)(live > 0x7FFFFFFFll ? 0x7FFFFFFFll : live);
        pp->n_exp = n_exp; pp->n_built = n_built;
        pp->child_first = child_first; pp->n_nodes = child_first + 2 * n_exp;
        pp->lvl_first = child_first; pp->lvl_end = child_first + 2 * n_exp;
        if (with_hist) pp->n_hslots = hs0 + 2 * n_exp;
    }
}

// ------------------------------------------------------------------------------------------------
// k_level_replay: LightGBM's best-first growth replayed over the speculative nodes; emits the
// tree (Tree::Split numbering), applies Shrinkage / AddBias, and the node -> score-delta table.
// One wave per class tree.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_level_replay(LvPlan* __restrict__ plan, const SNode* __restrict__ nodes, const int32_t* __restrict__ count,
                                                     TreeOut out, const double* __restrict__ init, double* __restrict__ node_delta /* [K][256] */,
                                                     int32_t* __restrict__ leaf_node_out /* [K][LV_MAX_LEAVES] */,
                                                     int32_t* __restrict__ any_split, int32_t* __restrict__ err_flag, const int32_t* __restrict__ itp, TrainConst c) {
    const int it = *itp;
    __shared__ int leaf_node[LV_MAX_LEAVES], leaf_parent[LV_MAX_LEAVES], leaf_isleft[LV_MAX_LEAVES];
    __shared__ int node_leaf[256];
    __shared__ double upd[LV_MAX_LEAVES];
    const int k = blockIdx.x, lane = lane_id();
    LvPlan* pp = &plan[k];
    const SNode* nk = nodes + (long long)k * 256;
    const long long tbase = (long long)it * c.K + k;
    const long long nb = tbase * (c.num_leaves - 1);
    double* lv = out.leaf_value + tbase * c.num_leaves;
    const int n_nodes = pp->n_nodes;
    // the whole selection loop runs out of LDS: one coalesced sweep over the speculative nodes first
    __shared__ double s_gain[256], s_lout[256], s_rout[256], s_lv[LV_MAX_LEAVES];
    __shared__ int s_feat[256], s_theta[256];
    __shared__ short s_left[256], s_right[256], s_parent[256];
    __shared__ unsigned char s_dleft[256];
    for (int n = lane; n < 256; n += 64) {
        node_leaf[n] = -1;
        if (n < n_nodes) {
            const SNode& sn = nk[n];
            const bool se = sn.searched != 0;
            s_gain[n] = se ? sn.best.gain : -INFINITY; s_feat[n] = se ? sn.best_feature : -1;
            s_theta[n] = sn.best.theta; s_dleft[n] = (unsigned char)(sn.best.dleft ? 1 : 0);
            s_lout[n] = sn.best.left_out; s_rout[n] = sn.best.right_out;
            s_left[n] = (short)sn.left; s_right[n] = (short)sn.right; s_parent[n] = (short)sn.parent;
        }
    }
    if (lane == 0) { leaf_node[0] = 0; leaf_parent[0] = -1; leaf_isleft[0] = 0; }
    __syncthreads();
    if (lane == 0) node_leaf[0] = 0;
    __syncthreads();
    int L = 1;
    const int max_leaves = c.num_leaves < LV_MAX_LEAVES ? c.num_leaves : LV_MAX_LEAVES;
    while (L < max_leaves) {
        double bg = -INFINITY; int bf = -1, bl = 0x7FFFFFFF;
        for (int l = lane; l < L; l += 64) {
            const int sn = leaf_node[l];
            const double g = s_gain[sn]; const int f = s_feat[sn];
            if (bl == 0x7FFFFFFF || leaf_better(g, f, l, bg, bf, bl)) { bg = g; bf = f; bl = l; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double g2 = __shfl_xor(bg, o); const int f2 = __shfl_xor(bf, o), l2 = __shfl_xor(bl, o);
            if (l2 != 0x7FFFFFFF && (bl == 0x7FFFFFFF || leaf_better(g2, f2, l2, bg, bf, bl))) { bg = g2; bf = f2; bl = l2; }
        }
        if (!(bg > 0.0)) break;
        const int sn = leaf_node[bl];
        const int sl = s_left[sn], sr = s_right[sn];
        if (sl < 0) { if (lane == 0) { pp->error = 1; atomicOr(err_flag, 1); } break; }   // the expansion bound was violated (must never happen)
        const int node = L - 1, right_leaf = L;
        __syncthreads();   // every lane has read leaf_node[] before lane 0 rewrites it
        if (lane == 0) {
            out.feat[nb + node] = bf; out.theta[nb + node] = s_theta[sn]; out.dleft[nb + node] = (int)s_dleft[sn]; out.gain[nb + node] = s_gain[sn];
            out.left[nb + node] = ~bl; out.right[nb + node] = ~right_leaf;
            const int pn = leaf_parent[bl];
            if (pn >= 0) { if (leaf_isleft[bl]) out.left[nb + pn] = node; else out.right[nb + pn] = node; }
            s_lv[bl] = s_lout[sn]; s_lv[right_leaf] = s_rout[sn];
            leaf_node[bl] = sl; leaf_node[right_leaf] = sr;
            leaf_parent[bl] = node; leaf_isleft[bl] = 1; leaf_parent[right_leaf] = node; leaf_isleft[right_leaf] = 0;
            node_leaf[sl] = bl; node_leaf[sr] = right_leaf;
        }
        ++L;
        __syncthreads();
    }
    __syncthreads();
    if (lane == 0) out.L[tbase] = L;
    for (int l = lane; l < L; l += 64) leaf_node_out[(long long)k * LV_MAX_LEAVES + l] = leaf_node[l];   // leaf counts follow after the final pass
    double* nd = node_delta + (long long)k * 256;
    if (L <= 1) {
        if (lane == 0) lv[0] = (it == 0) ? init[k] : 0.0;
        for (int i = lane; i < 256; i += 64) nd[i] = 0.0;
        return;
    }
    if (lane == 0) atomicOr(any_split + it, 1);
    for (int l = lane; l < L; l += 64) {
        double v = s_lv[l] * c.learning_rate;    // Tree::Shrinkage
        upd[l] = v;
        if (it == 0 && fabs(init[k]) > k_eps()) v += init[k];   // Tree::AddBias (model only; scores already hold init)
        lv[l] = v;
    }
    __syncthreads();
    // rows that sit below a final leaf (speculative descendants) inherit that leaf
    for (int n = lane; n < 256; n += 64) {
        double d = 0.0;
        if (n < n_nodes) {
            int a = n;
            while (a >= 0 && node_leaf[a] < 0) a = s_parent[a];
            // a split node's own entry is overwritten by its children only when it was split in the final tree;
            // node_leaf of a split node still names the leaf index its LEFT child inherited, so walk DOWN is never needed:
            // rows only ever sit in the deepest expanded node, whose nearest assigned ancestor-or-self is a final leaf.
            if (a >= 0) d = upd[node_leaf[a]];
        }
        nd[n] = d;
    }
}

// ------------------------------------------------------------------------------------------------
// k_level_final: the last routing step (depth max_depth-1 -> max_depth, no histogram), the exact
// row counts of the deepest children and ScoreUpdater::AddScore in ONE streaming pass: every
// training row ends in its final speculative node, whose score delta the replay has tabulated.
// grid (gx, K), block 256, 4 rows per thread.
// ------------------------------------------------------------------------------------------------
// (A variant that also wrote the NEXT iteration's gradients -- one read of the K x N scores instead of two -- was built twice in round 5 and
// measured slower both times (profiles/r5b_*, profiles/EXPERIMENTS.md); it left the tree with numerics v2.2, whose grid is measured by
// the gradient kernels.)
__global__ __launch_bounds__(256) void k_level_final(const uint4* __restrict__ rec, const uint8_t* __restrict__ node_all,
                                                     const uint8_t* __restrict__ inbag, const LvPlan* __restrict__ plan, const TreeOut out,
                                                     const double* __restrict__ node_delta, double* __restrict__ score, int32_t* __restrict__ count,
                                                     const int32_t* __restrict__ itp, LevelConst c) {
    const int it = *itp;
    __shared__ double nd[256];
    __shared__ uint32_t route0[256], route1[256];
    __shared__ int32_t cnt[2 * LV_MAX_EXP * LV_CNT_REP];
    const int k = blockIdx.y;
    if (out.L[(long long)it * c.K + k] <= 1) return;   // no split: no score change, nothing to count
    const LvPlan* pp = &plan[k];
    const bool route = !pp->done;                      // plan(max_depth) expanded at least one node
    const int n_exp = route ? pp->n_exp : 0, child_first = pp->child_first;
    const int tid = threadIdx.x, lane = tid & 63;
    nd[tid] = node_delta[(long long)k * 256 + tid];
    route0[tid] = route ? pp->route0[tid] : 0u; route1[tid] = pp->route1[tid];
    for (int i = tid; i < 2 * n_exp * LV_CNT_REP; i += 256) cnt[i] = 0;
    __syncthreads();
    const long long N = c.N;
    const uint8_t* node = node_all + (long long)k * c.NS;
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(rec);
    double* sk = score + (long long)k * N;
    // 4 rows per thread; node ids and scores are loaded together (independent loads), then routed and written back
    const bool aligned16 = (((unsigned long long)sk) & 15ull) == 0ull;   // uniform
    for (long long i = ((long long)blockIdx.x * 256 + tid) * 4; i < N; i += (long long)gridDim.x * 1024) {
        if (i + 3 < N && aligned16) {
            // (non-temporal: every score is read and written once per iteration -- 19 GB per step of the 10M x 16 job that would otherwise push the bin records the
            // level passes of the other targets in flight re-read out of the L2)
            typedef double v2d __attribute__((ext_vector_type(2)));
            const uint32_t n4 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(node + i));
            v2d s01 = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(sk + i)), s23 = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(sk + i + 2));
            if (n4 == 0xFFFFFFFFu) continue;
            double sv[4] = {s01.x, s01.y, s23.x, s23.y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int n = (int)((n4 >> (8 * j)) & 0xFFu);
                if (n == LV_INACTIVE) continue;
                const long long row = i + j;
                const uint32_t w0 = route0[n];
                if (w0 & (1u << 24)) {
                    const int f = (int)(w0 & 0xFFu), theta1 = (int)((w0 >> 8) & 0xFFu), nanbin = (int)((w0 >> 16) & 0xFFu);
                    const int bin = (int)rec8[((long long)(f >> 4) * N + row) * 16 + (f & 15)];
                    const bool left = (bin == nanbin) ? ((w0 >> 25) & 1u) != 0u : (bin < theta1);
                    const uint32_t w1 = route1[n];
                    n = left ? (int)(w1 & 0xFFu) : (int)((w1 >> 8) & 0xFFu);
                    if (!inbag || inbag[row]) atomicAdd(&cnt[(n - child_first) * LV_CNT_REP + (lane & (LV_CNT_REP - 1))], c.has_mult ? (int)rec8[((long long)(c.nchunk - 1) * N + row) * 16 + 15] : 1);
                }
                sv[j] += nd[n];
            }
            s01.x = sv[0]; s01.y = sv[1]; s23.x = sv[2]; s23.y = sv[3];
            __builtin_nontemporal_store(s01, reinterpret_cast<v2d*>(sk + i)); __builtin_nontemporal_store(s23, reinterpret_cast<v2d*>(sk + i + 2));
            continue;
        }
        for (long long row = i; row < N && row < i + 4; ++row) {
            int n = node[row];
            if (n == LV_INACTIVE) continue;
            const uint32_t w0 = route0[n];
            if (w0 & (1u << 24)) {
                const int f = (int)(w0 & 0xFFu), theta1 = (int)((w0 >> 8) & 0xFFu), nanbin = (int)((w0 >> 16) & 0xFFu);
                const int bin = (int)rec8[((long long)(f >> 4) * N + row) * 16 + (f & 15)];
                const bool left = (bin == nanbin) ? ((w0 >> 25) & 1u) != 0u : (bin < theta1);
                const uint32_t w1 = route1[n];
                n = left ? (int)(w1 & 0xFFu) : (int)((w1 >> 8) & 0xFFu);
                if (!inbag || inbag[row]) atomicAdd(&cnt[(n - child_first) * LV_CNT_REP + (lane & (LV_CNT_REP - 1))], c.has_mult ? (int)rec8[((long long)(c.nchunk - 1) * N + row) * 16 + 15] : 1);
            }
            sk[row] += nd[n];
        }
    }
    __syncthreads();
    for (int ci = tid; ci < 2 * n_exp; ci += 256) {
        int tot = 0;
        for (int r2 = 0; r2 < LV_CNT_REP; ++r2) tot += cnt[ci * LV_CNT_REP + r2];
        if (tot) atomicAdd(&count[(long long)k * 256 + child_first + ci], tot);
    }
}

// ------------------------------------------------------------------------------------------------
// k_level_last: what is left of k_level_final when the score update rides in the next iteration's gradient kernel (PendingScore, rgbm_kernels.h): the
// last DataPartition::Split -- the rows of the nodes expanded at depth max_depth - 1 move to their children, in place -- and the exact counts of those
// children.  Reads the node ids (1 B per (row, class tree)) of the class trees that expanded anything at that depth, and the split byte of the rows concerned.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_level_last(const uint4* __restrict__ rec, uint8_t* __restrict__ node_all, const uint8_t* __restrict__ inbag,
                                                    const LvPlan* __restrict__ plan, const TreeOut out, int32_t* __restrict__ count, const int32_t* __restrict__ itp, LevelConst c) {
    const int it = *itp;
    __shared__ uint32_t route0[256], route1[256];
    __shared__ int32_t cnt[2 * LV_MAX_EXP * LV_CNT_REP];
    const int k = blockIdx.y;
    if (out.L[(long long)it * c.K + k] <= 1) return;   // no split
    const LvPlan* pp = &plan[k];
    if (pp->done) return;                              // plan(max_depth) expanded nothing: every row already sits in its leaf
    const int n_exp = pp->n_exp, child_first = pp->child_first;
    const int tid = threadIdx.x, lane = tid & 63;
    route0[tid] = pp->route0[tid]; route1[tid] = pp->route1[tid];
    for (int i = tid; i < 2 * n_exp * LV_CNT_REP; i += 256) cnt[i] = 0;
    __syncthreads();
    const long long N = c.N;
    uint8_t* node = node_all + (long long)k * c.NS;
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(rec);
    // 16 rows per thread and step (the node-id arrays are padded with LV_INACTIVE to whole wave tiles)
    for (long long i = ((long long)blockIdx.x * 256 + tid) * 16; i < N; i += (long long)gridDim.x * 4096) {
        uint4 v = *reinterpret_cast<const uint4*>(node + i);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        bool changed = false;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = (int)((w[d] >> (8 * j)) & 0xFFu);
                const uint32_t w0 = route0[n];             // (LV_INACTIVE: never expanded)
                if (w0 & (1u << 24)) {
                    const long long row = i + d * 4 + j;
                    const int f = (int)(w0 & 0xFFu), theta1 = (int)((w0 >> 8) & 0xFFu), nanbin = (int)((w0 >> 16) & 0xFFu);
                    const int bin = (int)rec8[((long long)(f >> 4) * N + row) * 16 + (f & 15)];
                    const bool left = (bin == nanbin) ? ((w0 >> 25) & 1u) != 0u : (bin < theta1);
                    const uint32_t w1 = route1[n];
                    const int nn = left ? (int)(w1 & 0xFFu) : (int)((w1 >> 8) & 0xFFu);
                    if (!inbag || inbag[row]) atomicAdd(&cnt[(nn - child_first) * LV_CNT_REP + (lane & (LV_CNT_REP - 1))], c.has_mult ? (int)rec8[((long long)(c.nchunk - 1) * N + row) * 16 + 15] : 1);
                    w[d] = (w[d] & ~(0xFFu << (8 * j))) | ((uint32_t)nn << (8 * j));
                    changed = true;
                }
            }
        }
        if (changed) *reinterpret_cast<uint4*>(node + i) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    for (int ci = tid; ci < 2 * n_exp; ci += 256) {
        int tot = 0;
        for (int r2 = 0; r2 < LV_CNT_REP; ++r2) tot += cnt[ci * LV_CNT_REP + r2];
        if (tot) atomicAdd(&count[(long long)k * 256 + child_first + ci], tot);
    }
}

// leaf counts of the finished tree (Tree::leaf_count_), once every child count is final
__global__ __launch_bounds__(LV_MAX_LEAVES) void k_level_leafcount(const LvPlan* __restrict__ plan, const int32_t* __restrict__ count,
                                                                   const int32_t* __restrict__ leaf_node, TreeOut out, const int32_t* __restrict__ itp, TrainConst c) {
    const int it = *itp;
    const int k = blockIdx.x, l = threadIdx.x;
    const long long tbase = (long long)it * c.K + k;
    const int L = out.L[tbase];
    if (l >= L) return;
    const int n = leaf_node[(long long)k * LV_MAX_LEAVES + l];
    out.leaf_count[tbase * c.num_leaves + l] = (n == 0) ? (int)plan[k].n_in : count[(long long)k * 256 + n];
}

// the iteration counter lives on the device so that one boosting iteration is the same launch sequence every time (hipGraph)
__global__ void k_next_iteration(int32_t* it) { if (threadIdx.x == 0 && blockIdx.x == 0) it[0] += 1; }

}  // namespace rg
