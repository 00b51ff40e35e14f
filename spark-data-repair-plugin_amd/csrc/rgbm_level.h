// rgbm_level.h -- the level-synchronous ("streaming") tree grower for gfx950.
//
// LightGBM grows a tree leaf-wise (best-first).  Re-creating that literally on a GPU means one
// partition + one small gathered histogram pass per split: 150 dependent launches per boosting
// iteration, random 16-byte gathers and launch-latency-bound small leaves (measured round 1:
// k_partition 50 %, k_hist 24 % of the time, profiles/r01a_*).  This grower produces the SAME tree,
// bit for bit, from at most max_depth+1 fully coalesced streaming passes over the row block:
//
//   * every row carries the id of the speculative node it sits in (u8 [K][N], ping-pong);
//   * pass L routes every row from its depth-(L-1) node to the depth-L child (one byte compare on
//     the record that is loaded anyway), counts rows per child exactly, and accumulates the
//     histogram of ONE child per expanded parent (the other is parent - child: sums are exact
//     integers, so which child is built never changes a bit of the result);
//   * k_level_split scans both children of every expanded parent (same FindBestThreshold code as
//     the leaf-wise path);
//   * k_level_plan decides which nodes of the new level to expand.  A node is expanded unless it
//     PROVABLY cannot be split by best-first growth: with pm(X) = min gain on the path root..X,
//     every known node Y with pm(Y) > pm(X) is split before X (induction on the best-first
//     queue), so X is dead once num_leaves-1 such nodes exist.  Expansion is a superset of the
//     final tree, never a subset;
//   * k_level_replay runs LightGBM's best-first selection (ArrayArgs::ArgMax + Tree::Split
//     numbering) over the speculative nodes and emits the tree in exactly the leaf-wise order.
//
// Used when 1 <= max_depth <= 7 and F <= 255 (the reference fixes max_depth = 7, train.py:109);
// every other configuration takes the leaf-wise path of rgbm_kernels.h.
//
// LDS per workgroup (1024 threads, one workgroup per CU): route table | child counters |
// packed 32+32-bit histogram slots [built node][feature][bin][replica] | 32-bit carry words per
// bin.  Packed slots are drained lazily: every lane budgets the |g| and h it has added since the
// last drain so that no field can exceed 2047 + sum of the 1024 lane budgets < 2^31 (2^32 for h);
// a drain moves the bits >= 2^11 to the carry words (see k_level_pass).
#pragma once
#include "rgbm_kernels.h"

namespace rg {

constexpr int LV_INACTIVE = 255;     // node id of rows that do not take part (target cell NULL)
constexpr int LV_MAX_EXP = 64;       // expanded parents per level (depth <= 6)
constexpr int LV_MAX_BUILT = 32;     // built children per level (parents of depth <= 5)
constexpr int LV_CNT_REP = 16;
#ifndef LV_THREADS_N
#define LV_THREADS_N 1024    // threads of a level-pass workgroup
#endif
#ifndef LV_BLOCKS_PER_CU
#define LV_BLOCKS_PER_CU 1   // resident level-pass workgroups per CU; they share the 160 KB of LDS
#endif
constexpr int LV_THREADS = LV_THREADS_N;
constexpr int LV_TILE = 2 * LV_THREADS;      // two rows per lane and tile
constexpr int LV_LDS_TOTAL = 160 * 1024;
constexpr int LV_LDS_BYTES = (LV_LDS_TOTAL / LV_BLOCKS_PER_CU) & ~1023;
constexpr int LV_CARRY_SHIFT = 11;
constexpr int LV_MAX_DEPTH = 7;
constexpr int LV_MAX_LEAVES = 128;
// split mode (k_level_route + k_level_pass<STREAM>)
constexpr int LV_SRING = 192;                           // entries of a wave's built-row ring (k_level_pass<STREAM>): <= 64 pending + 128 appended per tile
constexpr int LV_SRING_BYTES = (LV_THREADS / 64) * LV_SRING * 12;   // entry = g, h, row (4 B each)
constexpr int RT_KS = 32;                               // class trees per route workgroup (LDS route tables: 512 B each)
constexpr int RT_THREADS = 256;
constexpr int RT_WT_ROWS = 256;                         // rows of one wave tile: 4 consecutive rows per lane
#ifndef LV_RING
#define LV_RING 0       // 1: level passes compact the rows that feed a histogram into full waves (per-wave LDS ring).
                        // Measured on MI355X (K=64, 10M rows): halves the LDS atomic instructions but the pass time is unchanged
                        // (2.9 ms either way: VALU issue + s_waitcnt bound, not LDS bound), so the simpler path is the default.
#endif
constexpr int LV_LIST = 256;                                   // ring entries per wave
constexpr int LV_LIST_BYTES = LV_RING ? (LV_THREADS / 64) * LV_LIST * 4 : 0;
static_assert(!LV_RING || LV_TILE <= 2048, "ring entries hold an 11-bit row offset");

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
    int32_t n_nodes, lvl_first, lvl_end, n_exp, n_built, n_groups, npg, buf, buf_in, done, error, child_first, n_hslots, pad;
    long long n_in;
    uint8_t exp[LV_MAX_EXP];           // expanded parents (ascending node id)
    uint8_t built_is_left[LV_MAX_EXP];
    uint32_t route0[256];              // feature | (theta+1)<<8 | nanbin<<16 | expanded<<24 | dleft<<25
    uint32_t route1[256];              // left | right<<8 | left built slot<<16 | right built slot<<24  (0xFF = none)
};

struct LvLayout {   // per (class tree, chunk): packed-slot layout of the coming pass
    int32_t spn;                       // packed slots per built node
    int32_t sh[16], fbase[16];
};

struct LevelConst {
    int32_t gx, max_built, nchunk, K, F, totbins, num_leaves, max_depth, min_data_in_leaf, lds_bytes;
    int32_t drain_shift, pad0;   // testing: the per-lane drain budgets are shifted right by this much (0 in production), which forces drains on small inputs
    int32_t split_mode;          // 1: route + stream-accumulate kernels (k_level_route / k_level_pass<STREAM>); sibling counts are parent - built
    int32_t sib_local;           // split mode: k_level_split also writes the derived sibling count into the LOCAL count array (row-sharded: rank 0 only)
    long long N, NS;   // rows; row stride of the node-id arrays (multiple of 16)
    // GONLY passes (split mode): the gradient buffer holds int32 g only; h = h_from_g(g, label, weight) for the rows that are accumulated
    double inv_sg, sh, factor;
    int32_t objective, n_labels;
};

__device__ __forceinline__ uint32_t rec_byte(const uint4& r, int j) {
    uint32_t w = (j < 8) ? ((j < 4) ? r.x : r.y) : ((j < 12) ? r.z : r.w);
    return (w >> (8 * (j & 3))) & 0xFFu;
}

// ------------------------------------------------------------------------------------------------
// packed-slot layout for `ng` built nodes of one chunk inside `avail` bytes of LDS
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline int lv_cap_shift(int nbins) { int s = 0; while (s < 5 && (nbins << (s + 1)) <= 2048) ++s; return s; }

__host__ __device__ inline int lv_slots(const FeatMeta* fm, int nfeat, int s) {
    int t = 0;
    for (int j = 0; j < nfeat; ++j) { int cs = lv_cap_shift(fm[j].nbins); t += fm[j].nbins << (s < cs ? s : cs); }
    return t;
}

// bytes one built node needs in chunk `cm` at replication cap s (packed slots + carry words + slot->bin map share)
__host__ __device__ inline long long lv_node_bytes(const FeatMeta* fm, const ChunkMeta& cm, int s) {
    return (long long)lv_slots(fm, cm.nfeat, s) * 8 + (long long)cm.wide_bins * 8;
}

// everything that depends on the replication cap s for `nodes` built nodes: packed slots + carry words per node, plus the slot->wide map
__host__ __device__ inline long long lv_layout_bytes(const FeatMeta* fm, const ChunkMeta& cm, int s, long long nodes) {
    return nodes * lv_node_bytes(fm, cm, s) + (long long)lv_slots(fm, cm.nfeat, s) * 2;
}

__host__ __device__ inline long long lv_fixed_bytes(const ChunkMeta& cm, int n_exp, const FeatMeta* fm) {
    // route tables + child counters + (wide->slot, wide->hoff) tables + alignment slack (the slot->wide map is charged to the layout: lv_layout_bytes)
    (void)fm;
    return 2048 + 16 + LV_LIST_BYTES + LV_SRING_BYTES + 256 + (long long)2 * n_exp * LV_CNT_REP * 4 + (long long)cm.wide_bins * 8 + 64;
}

// Packed-slot layout of one chunk for `nodes` built nodes inside `avail` bytes: the largest uniform replication 2^s (s <= 5,
// at most 2048 slots per feature) that fits.  (A greedy variant that replicates the feature with the fewest slots first was
// measured 10 % slower: the synthetic columns are correlated, so high-cardinality features collide as well, and the drain
// budget of k_level_pass scales with the SMALLEST replication factor.)  Computed on the host, once per group size.
__host__ __device__ inline void lv_choose_layout(const FeatMeta* fm, const ChunkMeta& cm, long long nodes, long long avail, LvLayout& L) {
    int s = 5;
    while (s > 0 && lv_layout_bytes(fm, cm, s, nodes) > avail) --s;
    int off = 0;
    for (int j = 0; j < 16; ++j) {
        if (j < cm.nfeat) { const int cs = lv_cap_shift(fm[j].nbins); L.sh[j] = s < cs ? s : cs; L.fbase[j] = off; off += fm[j].nbins << L.sh[j]; }
        else { L.sh[j] = 0; L.fbase[j] = 0; }
    }
    L.spn = off;
}

// ------------------------------------------------------------------------------------------------
// k_level_init: per class tree, start of a boosting iteration
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_level_init(LvPlan* __restrict__ plan, LvLayout* __restrict__ layout, const LvLayout* __restrict__ lay_table /* [nchunk][LV_MAX_BUILT+1] */,
                                                   SNode* __restrict__ nodes, const FeatMeta* __restrict__ fmeta, const ChunkMeta* __restrict__ cmeta,
                                                   const unsigned int* __restrict__ n_in_ptr, long long n_train, LevelConst c) {
    const int k = blockIdx.x, lane = lane_id();
    LvPlan* pp = &plan[k];
    const long long n_in = n_in_ptr ? (long long)n_in_ptr[0] : n_train;
    for (int i = lane; i < 256; i += 64) { pp->route0[i] = 0; pp->route1[i] = 0xFFFFFFFFu; }
    if (lane < c.nchunk) {
        layout[(long long)k * c.nchunk + lane] = lay_table[lane * (LV_MAX_BUILT + 1) + 1];   // one built node (the root)
    }
    if (lane == 0) {
        pp->n_nodes = 1; pp->lvl_first = 0; pp->lvl_end = 1; pp->n_exp = 0; pp->n_built = 1; pp->n_groups = 1; pp->npg = 1;
        pp->buf = 0; pp->buf_in = 0; pp->error = 0; pp->child_first = 1; pp->n_hslots = 1; pp->n_in = n_in;
        pp->done = (n_in < (long long)c.min_data_in_leaf * 2) ? 1 : 0;   // BeforeFindBestSplit on the root
        SNode r; memset(&r, 0, sizeof(r));
        r.count = (int)n_in; r.depth = 0; r.parent = -1; r.left = -1; r.right = -1; r.best_feature = -1; r.searched = 0; r.hslot = 0;
        r.best.gain = -INFINITY; r.pmin = -INFINITY;
        nodes[(long long)k * 256] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// k_level_pass: THE roofline kernel of this grower.  grid (gx, K, nchunk * groups), block 1024.
//   ROOT: every active row sits in node 0, whose histogram is built (no routing, no counting).
//   else: route + count (chunk 0 / group 0 blocks write the new node ids) + histogram of the
//         built children that belong to this block's group.
// Algorithmic bytes per accumulated row: F bin bytes + 8 B (g,h); the pass also streams the node
// ids (1 B in, 1 B out) and the records of rows it only routes.
// ------------------------------------------------------------------------------------------------
//   STREAM (split mode): no routing -- k_level_route has already moved every row to its child.  The pass streams the node
//         ids (1 B) and (g,h) (8 B) of every row, fully coalesced, and looks the node id up in an LDS table: rows that sit
//         in a BUILT child are appended to the wave's LDS ring (g, h, row); whenever 64 of them are waiting, their records
//         are re-read (L2: the 64 class trees walk the rows in lock step) and the packed atomics run on a FULL wave.  A wave
//         instruction of LDS atomics costs the same for 6 active lanes as for 64 (profiles/r01_lds_atomic_active_lanes.txt),
//         and a built child holds ~12 % of the rows: the fused pass pays that instruction for every 64-row step, this one
//         for every 64 built rows.  Rows per built child are counted here; the sibling is parent - built (k_level_split).
//   GONLY: the (g,h) buffer holds the quantised gradient alone (int32 [K][N], TrainConst::g_only): the pass streams 4 B instead of 8 per
//         (row, class tree) and derives h from g, the row's label (u8 `ylab`) and the label's weight (`cw32`, float32-rounded; LDS copy)
//         for the rows it accumulates -- numerics v1.02 defines h that way for every path, so the histograms are the same integers.
template <bool ROOT, bool BAG, int MULTI /* 0: one 16-feature chunk; 2: exactly two (the other chunk's record is prefetched too); 3: more */, bool STREAM = false, bool GONLY = false>
__global__ __launch_bounds__(LV_THREADS, LV_BLOCKS_PER_CU) void k_level_pass(const uint4* __restrict__ rec, const int2* __restrict__ gh,
                                                           uint8_t* __restrict__ node_a, uint8_t* __restrict__ node_b,
                                                           const uint8_t* __restrict__ inbag, const LvPlan* __restrict__ plan,
                                                           const LvLayout* __restrict__ layout, HistBin* __restrict__ part,
                                                           int32_t* __restrict__ count, const FeatMeta* __restrict__ fmeta,
                                                           const ChunkMeta* __restrict__ cmeta, int with_hist, LevelConst c,
                                                           const uint8_t* __restrict__ ylab = nullptr, const double* __restrict__ cw32 = nullptr) {
    static_assert(!GONLY || ROOT || STREAM, "g-only buffers exist in split mode only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Block -> (class tree, row block).  Fused mode: grid (gx, K): block x walks the tiles x, x + gx, ... of class tree blockIdx.y.
    // Split mode: 1-D grid of K * gx blocks (gx a multiple of 8), every block owns a CONTIGUOUS run of tiles, and the decode makes
    // the launch order walk the table front to back with ALL class trees of a row block on ONE XCD (blocks land on XCD id % 8):
    //     id -> xl = id % 8,  r = id / 8,  class tree = r % K,  row block = (r / K) * 8 + xl
    // The class trees re-read the bin records of the rows they accumulate; spread over the chip and drifting apart over a whole
    // pass they missed L2 half of the time (2-4 GB of a level pass's 8-10 GB).  Now the ~32 workgroups that are resident on an XCD
    // share one 4 MB slice of records, start together and finish within ~0.2 ms.
    int k, bx, nbx;
    if (c.split_mode) {
        const unsigned id = blockIdx.x, r = id >> 3;
        k = (int)(r % (unsigned)c.K); bx = (int)(r / (unsigned)c.K) * 8 + (int)(id & 7u); nbx = c.gx;
    } else { k = blockIdx.y; bx = blockIdx.x; nbx = gridDim.x; }
    const int ch = blockIdx.z % c.nchunk, grp = blockIdx.z / c.nchunk;
    const LvPlan* pp = &plan[k];
    if (pp->done) return;
    const int n_exp = (ROOT || STREAM) ? 0 : pp->n_exp;
    const int n_built = with_hist ? pp->n_built : 0;
    const int npg = pp->npg;
    const bool writer = !ROOT && !STREAM && ch == 0 && grp == 0;
    const int g0 = grp * npg;
    int ng = n_built - g0; if (ng > npg) ng = npg; if (ng < 0) ng = 0;
    if (grp >= pp->n_groups && !writer) return;
    if (ng == 0 && !writer) return;

    const ChunkMeta cm = cmeta[ch];
    const FeatMeta* fm = fmeta + cm.first_feat;
    const LvLayout lay = layout[(long long)k * c.nchunk + ch];
    const int spn = lay.spn, wb = cm.wide_bins;
    const int tid = threadIdx.x, lane = tid & 63;

    // ---- LDS carve-up
    uint2* route = reinterpret_cast<uint2*>(smem);                                      // [256] (w0, w1) of LvPlan::route0/1
    int32_t* drain_flag = reinterpret_cast<int32_t*>(route + 256);                      // [4] (16 B), relaxed atomic accesses
    uint32_t* lst = reinterpret_cast<uint32_t*>(route + 256) + 4 + (tid >> 6) * LV_LIST;   // this wave's ring (LV_RING)
    uint32_t* sring = reinterpret_cast<uint32_t*>(route + 256) + 4 + LV_LIST_BYTES / 4 + (tid >> 6) * (LV_SRING * 3);   // STREAM: [3][LV_SRING] g | h | row
    uint8_t* bslot = reinterpret_cast<uint8_t*>(reinterpret_cast<uint32_t*>(route + 256) + 4 + LV_LIST_BYTES / 4 + LV_SRING_BYTES / 4);   // STREAM: node id -> group-local built slot
    int32_t* cnt = reinterpret_cast<int32_t*>(route + 256) + 4 + LV_LIST_BYTES / 4 + LV_SRING_BYTES / 4 + 64;
    const int ncnt = STREAM ? ng * LV_CNT_REP : 2 * n_exp * LV_CNT_REP;   // STREAM: rows per built child of this group
    int32_t* wide_g = cnt + ncnt;
    uint32_t* wide_h = reinterpret_cast<uint32_t*>(wide_g + (size_t)ng * wb);
    uint32_t* w_slot = wide_h + (size_t)ng * wb;          // [wb] first packed slot of the bin | sh << 24
    uint32_t* w_hoff = w_slot + wb;                       // [wb] offset of the bin inside a node histogram
    uint16_t* s2w = reinterpret_cast<uint16_t*>(w_hoff + wb);   // [spn] packed slot -> carry-word index
    size_t off = reinterpret_cast<unsigned char*>(s2w + spn) - smem;
    off = (off + 15) & ~(size_t)15;
    unsigned long long* fast = reinterpret_cast<unsigned long long*>(smem + off);

    if (!ROOT && !STREAM) for (int i = tid; i < 256; i += LV_THREADS) {
        // LDS copy of the route table, specialised for this block: built slots become group-local (0xFF = the child's
        // histogram is not this block's business) and an unexpanded node routes to itself, so the row loop needs no selects
        const uint32_t w0 = pp->route0[i]; uint32_t w1 = pp->route1[i];
        if (w0 & (1u << 24)) {
            const int ls = (int)((w1 >> 16) & 0xFFu), rs = (int)(w1 >> 24);
            const uint32_t l2 = (ls != 0xFF && ls >= g0 && ls - g0 < ng) ? (uint32_t)(ls - g0) : 0xFFu;
            const uint32_t r2 = (rs != 0xFF && rs >= g0 && rs - g0 < ng) ? (uint32_t)(rs - g0) : 0xFFu;
            w1 = (w1 & 0xFFFFu) | l2 << 16 | r2 << 24;
        } else w1 = (uint32_t)i | (uint32_t)i << 8 | 0xFFFF0000u;
        route[i] = make_uint2(w0, w1);
    }
    if (tid < 4) drain_flag[tid] = 0;
    double2* wtab = reinterpret_cast<double2*>(route);     // GONLY: (weight, 1 / weight) of every label (<= 128; the route table's 2 KB are free in ROOT / STREAM passes)
    if (GONLY) for (int i = tid; i < 128; i += LV_THREADS) { const double w = (cw32 && i < c.n_labels) ? cw32[i] : 1.0; wtab[i] = make_double2(w, rg_inv_weight(w)); }
    if (STREAM) {   // k_level_plan numbers the children of expanded parent ei as child_first + 2 ei (left), + 1 (right); one of them is built
        for (int i = tid; i < 256; i += LV_THREADS) {
            const int d = i - pp->child_first;
            uint8_t v = 0xFF;
            if (d >= 0 && d < 2 * n_built) {
                const int ei = d >> 1;
                if ((pp->built_is_left[ei] ? 0 : 1) == (d & 1) && ei >= g0 && ei - g0 < ng) v = (uint8_t)(ei - g0);
            }
            bslot[i] = v;
        }
    }
#define LV_FLAG_LOAD() __hip_atomic_load(drain_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LV_FLAG_STORE(v) __hip_atomic_store(drain_flag, (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
    for (int i = tid; i < ncnt; i += LV_THREADS) cnt[i] = 0;
    for (int i = tid; i < ng * wb; i += LV_THREADS) { wide_g[i] = 0; wide_h[i] = 0u; }
    for (int i = tid; i < ng * spn; i += LV_THREADS) fast[i] = 0ull;
    for (int j = 0; j < cm.nfeat; ++j) {
        const int nb = fm[j].nbins, sh = lay.sh[j], fb = lay.fbase[j], wo = fm[j].wide_off;
        for (int b = tid; b < nb; b += LV_THREADS) { w_slot[wo + b] = (uint32_t)(fb + (b << sh)) | ((uint32_t)sh << 24); w_hoff[wo + b] = (uint32_t)(fm[j].hoff + b); }
        for (int s = tid; s < (nb << sh); s += LV_THREADS) s2w[fb + s] = (uint16_t)((wo + (s >> sh)) | (sh > 0 ? 0x8000 : 0));
    }
    // per-feature lane constants: byte offset of this lane's replica of bin 0 (relative to the node's slots), and
    // the shift that turns a bin into a byte offset
    int cj[16], fsh3[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { fsh3[j] = lay.sh[j] + 3; cj[j] = (lay.fbase[j] + (lane & ((1 << lay.sh[j]) - 1))) * 8; }
    const int nfeat = cm.nfeat;
    __syncthreads();

    const long long N = c.N;
    const uint8_t* node_in = (pp->buf_in ? node_b : node_a) + (long long)k * c.NS;
    uint8_t* node_out = (pp->buf ? node_b : node_a) + (long long)k * c.NS;
    if (ROOT) node_in = (pp->buf ? node_b : node_a) + (long long)k * c.NS;
    const uint4* recc = rec + (long long)ch * N;
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(rec);
    const int2* ghk = gh + (long long)k * N;
    const int child_first = pp->child_first;
    const long long ntiles_all = (N + LV_TILE - 1) / LV_TILE;
    // this block's tiles: t = tbeg, tbeg + tstep, ... < ntiles
    const long long tbeg = c.split_mode ? ntiles_all * bx / nbx : bx;
    const long long ntiles = c.split_mode ? ntiles_all * (bx + 1) / nbx : ntiles_all;
    const long long tstep = c.split_mode ? 1 : nbx;

    // Packed-slot overflow control without per-tile barriers.  Every lane keeps the sums of |g| and h it has
    // added since the last drain; a field of any slot is at most 2047 (drain remainder) + the sum over all 1024
    // lanes, so it cannot overflow while every lane stays within LB_G / LB_H.  A lane that would exceed its
    // budget raises the drain flag; waves poll the flag once per row step and rendezvous at a barrier, drain
    // cooperatively and continue.  Drains are rare for small gradients (multiclass), at worst every 2 rows/lane.
    // A slot of feature f only receives lanes with the same (lane mod rep_f): 1024 / rep_f lanes, so the budget grows with the
    // smallest replication factor of this block's layout (x16..32 at the root and the shallow levels).
    int rep_min = 32;
    for (int j = 0; j < cm.nfeat; ++j) { const int r = 1 << lay.sh[j]; if (r < rep_min) rep_min = r; }
    const unsigned LB_G = ((((1u << 31) - 2048u) / LV_THREADS) * (unsigned)rep_min) >> c.drain_shift;
    const unsigned LB_H = ((unsigned)((((1ull << 32) - 2048ull)) / LV_THREADS) * (unsigned)rep_min) >> c.drain_shift;
    unsigned acc_g = 0, acc_h = 0;
    auto drain = [&]() {
        // move the bits above 2^11 of both fields into the per-bin carry words
        for (int li = 0; li < ng; ++li) {
            for (int s2 = tid; s2 < spn; s2 += LV_THREADS) {
                const size_t i = (size_t)li * spn + s2;
                const unsigned long long v = fast[i];
                const int g32 = (int)(v >> 32);
                const unsigned int h32 = (unsigned int)(v & 0xFFFFFFFFull);
                const int cg = g32 >> LV_CARRY_SHIFT;
                const unsigned int chh = h32 >> LV_CARRY_SHIFT;
                if (cg != 0 || chh != 0u) {
                    fast[i] = ((unsigned long long)(unsigned int)(g32 & ((1 << LV_CARRY_SHIFT) - 1)) << 32) | (unsigned long long)(h32 & ((1u << LV_CARRY_SHIFT) - 1u));
                    const int sw = (int)s2w[s2];
                    const int wi = li * wb + (sw & 0x7FFF);
                    if (sw & 0x8000) { if (cg != 0) atomicAdd(&wide_g[wi], cg); if (chh != 0u) atomicAdd(&wide_h[wi], chh); }
                    else { wide_g[wi] += cg; wide_h[wi] += chh; }   // un-replicated bin: this thread is its only writer
                }
            }
        }
    };
    // every wave that reaches a rendezvous executes exactly: barrier, [flag set: drain, barrier, clear, barrier]
    auto rendezvous = [&]() -> bool {
        __syncthreads();
        if (!LV_FLAG_LOAD()) return false;
        drain();
        __syncthreads();
        if (tid == 0) LV_FLAG_STORE(0);
        __syncthreads();
        acc_g = 0; acc_h = 0;
        return true;
    };

    // one histogram update of a (row, built node) pair held by this lane: 15-16 packed LDS atomics, 3 instructions each
    auto accumulate = [&](bool on, int li, const uint4& r, const int2& g_in) __attribute__((always_inline)) {
        int2 g = g_in;
        if (GONLY) {   // g_in = (quantised gradient, label): h is a function of both and of the label's weight (numerics v1.02)
            const int yl = on ? (g_in.y & 0x7F) : 0;
            const double2 ww = wtab[yl];
            g.y = on ? h_from_g(g_in.x, yl == k, ww.x, ww.y, c.objective, c.inv_sg, c.sh, c.factor) : 0;
        }
        const unsigned long long packed = ((unsigned long long)(long long)g.x << 32) + (unsigned long long)(unsigned int)g.y;
        const bool need = on && packed != 0ull;
        const unsigned ag = (unsigned)(g.x < 0 ? -g.x : g.x), ah = (unsigned)g.y;
        const bool over = need && (acc_g + ag > LB_G || acc_h + ah > LB_H);
        if (__any(over)) { if (lane == 0) LV_FLAG_STORE(1); rendezvous(); }   // (the flag itself is polled once per tile, see tile_step)
        if (need) {
            acc_g += ag; acc_h += ah;
            unsigned char* fb = reinterpret_cast<unsigned char*>(fast) + (unsigned)li * (unsigned)(spn * 8);
            const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#ifdef LV_DBG_NO_ATOM   // timing experiment: everything but the LDS atomics (results are wrong)
#define LV_ATOM(j) asm volatile("" :: "v"(fb + cj[j] + (int)(((w[(j) >> 2] >> (8 * ((j) & 3))) & 0xFFu) << fsh3[j])), "v"(packed))
#else
#define LV_ATOM(j) atomicAdd(reinterpret_cast<unsigned long long*>(fb + cj[j] + (int)(((w[(j) >> 2] >> (8 * ((j) & 3))) & 0xFFu) << fsh3[j])), packed)
#endif
            if (nfeat >= 15) {   // the common shapes (full chunk, or 15 features): no per-feature branches
                LV_ATOM(0); LV_ATOM(1); LV_ATOM(2); LV_ATOM(3); LV_ATOM(4); LV_ATOM(5); LV_ATOM(6); LV_ATOM(7);
                LV_ATOM(8); LV_ATOM(9); LV_ATOM(10); LV_ATOM(11); LV_ATOM(12); LV_ATOM(13); LV_ATOM(14);
                if (nfeat == 16) LV_ATOM(15);
            } else {
#pragma unroll
                for (int j = 0; j < 14; ++j) if (j < nfeat) LV_ATOM(j);
            }
#undef LV_ATOM
        }
    };

    // Software pipeline: the loads of the next tile are in flight while this tile's LDS atomics run (one workgroup
    // per CU, so nothing else would hide the HBM latency).  Addresses are a uniform tile base + a 32-bit lane offset.
    constexpr int RPT = LV_TILE / LV_THREADS;
    int cur_n[RPT], nxt_n[RPT], cur_ib[RPT], nxt_ib[RPT]; uint4 cur_r[RPT], nxt_r[RPT], cur_r2[RPT], nxt_r2[RPT]; int2 cur_g[RPT], nxt_g[RPT];
    const uint4* rec_other = rec + (long long)(MULTI == 2 ? 1 - ch : ch) * N;   // MULTI == 2: the record that holds the other 16 features
    // straight-line loads (clamped offsets, no branches) so that the in-order vmcnt bookkeeping stays exact
    auto fetch = [&](long long t, int (&fn)[RPT], uint4 (&fr)[RPT], uint4 (&fr2)[RPT], int2 (&fg)[RPT], int (&fib)[RPT]) __attribute__((always_inline)) {
        const bool tv = t < ntiles;                                   // uniform
        const long long pb = tv ? t * LV_TILE : 0;
        const long long left_rows = N - pb;
        const unsigned lim = (unsigned)(left_rows < LV_TILE ? left_rows : LV_TILE) - 1u;   // last valid offset in the tile
        const uint8_t* nb_ = node_in + pb; const uint4* rb_ = recc + pb; const int2* gb_ = ghk + pb;
        const uint8_t* ib_ = BAG ? inbag + pb : nullptr;
#pragma unroll
        for (int s = 0; s < RPT; ++s) {
            const unsigned o = (unsigned)(s * LV_THREADS + tid);
            const unsigned oc = o < lim ? o : lim;
            // STREAM: node ids and (g,h) are use-once data; non-temporal loads leave more of L2 / Infinity Cache to the bin records the
            // batches re-read (measured: -7 % HBM fetch, -2 % time; profiles/r02_stream_pass_experiments.txt)
            const int nv = STREAM ? (int)__builtin_nontemporal_load(nb_ + oc) : (int)nb_[oc];
            if (!STREAM) fr[s] = rb_[oc]; else fr[s] = make_uint4(0, 0, 0, 0);
            if (!ROOT && !STREAM && MULTI == 2) fr2[s] = (rec_other + pb)[oc]; else fr2[s] = make_uint4(0, 0, 0, 0);
            if (GONLY) {
                const int32_t* g32 = reinterpret_cast<const int32_t*>(gh) + (long long)k * N + pb;
                const int gv = STREAM ? __builtin_nontemporal_load(g32 + oc) : g32[oc];
                fg[s] = make_int2(gv, ROOT ? (int)ylab[pb + oc] : 0);                 // ROOT: the label rides in the h slot (accumulate)
            }
            else if (STREAM) { const long long v = __builtin_nontemporal_load(reinterpret_cast<const long long*>(gb_ + oc)); fg[s] = make_int2((int)(v & 0xFFFFFFFFll), (int)(v >> 32)); }
            else if (ROOT || STREAM || !LV_RING) fg[s] = gb_[oc]; else fg[s] = make_int2(0, 0);
            fib[s] = BAG ? (int)ib_[oc] : 1;
            fn[s] = (tv && o <= lim) ? nv : LV_INACTIVE;
        }
    };
    if (STREAM) {
        // ---- split mode: coalesced stream over (node id, g, h) of every row; built rows go through the wave's LDS ring.
        // Ring invariant: at most 64 entries wait when a tile starts (<= 128 are appended per tile, LV_SRING = 192).  Whenever
        // >= 64 wait after a tile, one batch is taken out (its record gather is issued at once and consumed at the START of
        // the next tile, so an L2 round trip hides behind that tile's loads); more than 128 waiting (a tile of built rows
        // only: clustered data) are worked off on the spot.
        uint32_t* ring_g = sring; uint32_t* ring_h = sring + LV_SRING; uint32_t* ring_r = sring + 2 * LV_SRING;
        int r_head = 0, r_cnt = 0;                       // wave-uniform
        bool pend = false; uint4 p_rec = make_uint4(0, 0, 0, 0); int2 p_gh = make_int2(0, 0); int p_li = 0;
        const unsigned long long lane_lt = (1ull << lane) - 1ull;
        auto take_batch = [&](int nb) __attribute__((always_inline)) {     // nb = min(r_cnt, 64) entries -> p_*, gather issued
            const bool on = lane < nb;
            int pos = r_head + lane; if (pos >= LV_SRING) pos -= LV_SRING;
            const uint32_t row = ring_r[pos], hw = ring_h[pos];             // h < 2^21 (HQ_MAX): the built slot rides in bits 24..31
            p_gh = make_int2((int)ring_g[pos], GONLY ? (int)ylab[on ? row : 0u] : (int)(hw & 0xFFFFFFu));   // GONLY: (g, label)
            p_rec = recc[on ? row : 0u];
            p_li = on ? (int)(hw >> 24) : -1;
            r_head += nb; if (r_head >= LV_SRING) r_head -= LV_SRING;
            r_cnt -= nb; pend = true;
        };
        auto run_batch = [&]() __attribute__((always_inline)) {
            if (ch == 0 && p_li >= 0) atomicAdd(&cnt[p_li * LV_CNT_REP + (lane & (LV_CNT_REP - 1))], 1);
            accumulate(p_li >= 0, p_li < 0 ? 0 : p_li, p_rec, p_gh);
            pend = false;
        };
        // the loads of TWO tiles are in flight while a tile is processed (three register sets take turns): with one workgroup
        // per CU, 9 B per row and ~1.5 us of HBM latency, one tile ahead kept only ~18 KB per CU in flight = 3.7 TB/s
        auto stream_step = [&](long long t, int (&Cn)[RPT], int2 (&Cg)[RPT], int (&Cib)[RPT], int (&Xn)[RPT], uint4 (&Xr)[RPT], uint4 (&Xr2)[RPT], int2 (&Xg)[RPT], int (&Xib)[RPT]) __attribute__((always_inline)) {
            const long long p0 = t * LV_TILE;
            if (LV_FLAG_LOAD()) rendezvous();
            fetch(t + 2 * tstep, Xn, Xr, Xr2, Xg, Xib);
            if (pend) run_batch();
#pragma unroll
            for (int s = 0; s < RPT; ++s) {
                const unsigned o = (unsigned)(s * LV_THREADS + tid);
                const int n = Cn[s];
                const uint32_t bs = bslot[n];                                  // inactive rows carry id 255, which never names a child
                const bool built = bs != 0xFFu && (!BAG || Cib[s] != 0) && c.pad0 != 2;   // (pad0 == 2: timing experiment, nothing is appended)
                const unsigned long long m = __ballot(built);
                if (built) {
                    int pos = r_head + r_cnt + (int)__popcll(m & lane_lt); if (pos >= LV_SRING) pos -= LV_SRING; if (pos >= LV_SRING) pos -= LV_SRING;
                    ring_g[pos] = (uint32_t)Cg[s].x; ring_h[pos] = (GONLY ? 0u : (uint32_t)Cg[s].y) | (bs << 24); ring_r[pos] = (uint32_t)(p0 + o);
                }
                r_cnt += (int)__popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // ring entries are read by other lanes of this wave
            if (c.pad0 != 0) { r_cnt = 0; r_head = 0; }                  // timing experiment (RGBM_DBG_STREAM): the stream + ring without the batches
            while (r_cnt > 128) { take_batch(64); run_batch(); }
            if (r_cnt >= 64) take_batch(64);
        };
        int thd_n[RPT], thd_ib[RPT]; uint4 thd_r[RPT], thd_r2[RPT]; int2 thd_g[RPT];
        long long t = tbeg;
        fetch(t, cur_n, cur_r, cur_r2, cur_g, cur_ib);
        fetch(t + tstep, nxt_n, nxt_r, nxt_r2, nxt_g, nxt_ib);
        while (t < ntiles) {
            stream_step(t, cur_n, cur_g, cur_ib, thd_n, thd_r, thd_r2, thd_g, thd_ib); t += tstep; if (t >= ntiles) break;
            stream_step(t, nxt_n, nxt_g, nxt_ib, cur_n, cur_r, cur_r2, cur_g, cur_ib); t += tstep; if (t >= ntiles) break;
            stream_step(t, thd_n, thd_g, thd_ib, nxt_n, nxt_r, nxt_r2, nxt_g, nxt_ib); t += tstep;
        }
        if (pend) run_batch();
        while (r_cnt > 0) { take_batch(r_cnt < 64 ? r_cnt : 64); run_batch(); }
    } else if (ROOT || !LV_RING) {
        // one tile: prefetch the tile after it into the other register set, then process this one (the two sets swap
        // roles from call to call, so nothing is copied)
        auto tile_step = [&](long long t, int (&Cn)[RPT], uint4 (&Cr)[RPT], uint4 (&Cr2)[RPT], int2 (&Cg)[RPT], int (&Cib)[RPT],
                             int (&Xn)[RPT], uint4 (&Xr)[RPT], uint4 (&Xr2)[RPT], int2 (&Xg)[RPT], int (&Xib)[RPT]) __attribute__((always_inline)) {
            const long long p0 = t * LV_TILE;
            // one poll of the drain flag per tile, before this tile's atomics are queued: an LDS read returns behind every
            // LDS atomic issued before it, so polling inside the row steps would serialise the atomics of consecutive steps
            if (ng > 0 && LV_FLAG_LOAD()) rendezvous();
            fetch(t + tstep, Xn, Xr, Xr2, Xg, Xib);
            uint8_t* ob_ = node_out + p0;
#pragma unroll
            for (int s = 0; s < RPT; ++s) {
                const unsigned o = (unsigned)(s * LV_THREADS + tid);
                const int n = Cn[s];
                if (ROOT) { accumulate(n != LV_INACTIVE, 0, Cr[s], Cg[s]); continue; }
                const bool inrange = p0 + o < N;
                // branch-free routing (for !MULTI): unexpanded nodes (and the inactive id 255) route to themselves
                const uint2 e = route[n];
                const bool expd = (e.x & (1u << 24)) != 0u;
                const unsigned f = e.x & 0xFFu;
                unsigned bin;
                if (MULTI == 0 || MULTI == 2) {
                    const bool hi = (f & 8u) != 0u;
                    uint32_t rx = Cr[s].x, ry = Cr[s].y, rz = Cr[s].z, rw = Cr[s].w;
                    if (MULTI == 2) {   // the split feature may live in the other chunk's record (prefetched alongside)
                        const bool mine = (f >> 4) == (unsigned)ch;
                        rx = mine ? rx : Cr2[s].x; ry = mine ? ry : Cr2[s].y; rz = mine ? rz : Cr2[s].z; rw = mine ? rw : Cr2[s].w;
                    }
                    const uint32_t lo32 = hi ? rz : rx, hi32 = hi ? rw : ry;
                    bin = __builtin_amdgcn_perm(hi32, lo32, (f & 7u) | 0x0C0C0C00u);   // byte (f & 15) of the record
                } else {
                    bin = 0;
                    if (expd) {
                        if ((f >> 4) == (unsigned)ch) {
                            const bool hi = (f & 8u) != 0u;
                            const uint32_t rx = Cr[s].x, ry = Cr[s].y, rz = Cr[s].z, rw = Cr[s].w;
                            const uint32_t lo32 = hi ? rz : rx, hi32 = hi ? rw : ry;
                            bin = __builtin_amdgcn_perm(hi32, lo32, (f & 7u) | 0x0C0C0C00u);
                        } else bin = rec8[((long long)(f >> 4) * N + p0 + o) * 16 + (f & 15u)];
                    }
                }
                const bool left = (bin == ((e.x >> 16) & 0xFFu)) ? ((e.x >> 25) & 1u) != 0u : (bin < ((e.x >> 8) & 0xFFu));
                const unsigned sel = left ? e.y : (e.y >> 8);          // child in bits 0..7, group-local built slot in bits 16..23
                const int child = (int)(sel & 0xFFu);
                const int li = (int)((sel >> 16) & 0xFFu);             // 0xFF: nothing to accumulate here
                if (writer && expd && Cib[s]) atomicAdd(&cnt[(child - child_first) * LV_CNT_REP + (lane & (LV_CNT_REP - 1))], 1);
                if (writer && inrange) ob_[o] = (uint8_t)child;
                if (ng > 0) accumulate(li != 0xFF, li, Cr[s], Cg[s]);
            }
        };
        long long t = tbeg;
        fetch(t, cur_n, cur_r, cur_r2, cur_g, cur_ib);
        while (t < ntiles) {
            tile_step(t, cur_n, cur_r, cur_r2, cur_g, cur_ib, nxt_n, nxt_r, nxt_r2, nxt_g, nxt_ib); t += tstep; if (t >= ntiles) break;
            tile_step(t, nxt_n, nxt_r, nxt_r2, nxt_g, nxt_ib, cur_n, cur_r, cur_r2, cur_g, cur_ib); t += tstep;
        }
    } else {
        // ---- level pass with compaction.  Phase 1 (every row): route, count, store the new node id; rows that feed a
        // histogram of this block's group are appended to this wave's LDS ring (4 B: tile index | row offset | slot).
        // Phase 2: full waves of 64 listed rows reload (rec, gh) -- L2 hits, the records were streamed a moment ago --
        // and run the packed atomics.  Order inside one tile step: [batch loads] [far prefetch] [routing] [atomics]:
        // vmcnt retires in order, so waiting for the batch never waits for the prefetch issued after it.
        const long long gstep = gridDim.x;
        int lcount = 0, lhead = 0;            // wave-uniform ring state
        uint32_t ti = 0;                      // index of the current tile among this block's tiles
        fetch(blockIdx.x, cur_n, cur_r, cur_r2, cur_g, cur_ib);
        for (long long t = blockIdx.x; t < ntiles; t += gstep, ++ti) {
            const long long p0 = t * LV_TILE;
            bool b_on[2]; uint4 b_r[2]; int2 b_g[2]; int b_li[2]; int b_n[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int nb = ng > 0 ? (lcount < 64 ? lcount : 64) : 0;
                b_n[b] = nb; b_on[b] = lane < nb;
                const uint32_t e = lst[(lhead + (b_on[b] ? lane : 0)) & (LV_LIST - 1)];
                const long long row = b_on[b] ? ((long long)blockIdx.x + (long long)(e >> 16) * gstep) * LV_TILE + ((e >> 5) & 0x7FFu) : p0;
                b_li[b] = (int)(e & 31u);
                b_r[b] = recc[row]; b_g[b] = ghk[row];
                lhead += nb; lcount -= nb;
            }
            fetch(t + gstep, nxt_n, nxt_r, nxt_r2, nxt_g, nxt_ib);
            uint8_t* ob_ = node_out + p0;
#pragma unroll
            for (int s = 0; s < RPT; ++s) {
                const unsigned o = (unsigned)(s * LV_THREADS + tid);
                const int n = cur_n[s];
                int li = -1;
                const bool inrange = p0 + o < N;
                int child = n;
                if (n != LV_INACTIVE) {
                    const uint2 e = route[n];
                    if (e.x & (1u << 24)) {
                        const unsigned f = e.x & 0xFFu;
                        unsigned bin;
                        if (!MULTI || (f >> 4) == (unsigned)ch) {
                            const bool hi = (f & 8u) != 0u;
                            const uint32_t rx = cur_r[s].x, ry = cur_r[s].y, rz = cur_r[s].z, rw = cur_r[s].w;
                            const uint32_t lo32 = hi ? rz : rx, hi32 = hi ? rw : ry;
                            bin = __builtin_amdgcn_perm(hi32, lo32, (f & 7u) | 0x0C0C0C00u);
                        } else bin = rec8[((long long)(f >> 4) * N + p0 + o) * 16 + (f & 15u)];
                        const bool left = (bin == ((e.x >> 16) & 0xFFu)) ? ((e.x >> 25) & 1u) != 0u : (bin < ((e.x >> 8) & 0xFFu));
                        const unsigned sel = left ? e.y : (e.y >> 8);
                        child = (int)(sel & 0xFFu);
                        const int bs = (int)((sel >> 16) & 0xFFu);
                        if (writer && cur_ib[s]) atomicAdd(&cnt[(child - child_first) * LV_CNT_REP + (lane & (LV_CNT_REP - 1))], 1);
                        if (bs != 0xFF) li = bs;   // group-local already (LDS copy of the route table)
                    }
                }
                if (writer && inrange) ob_[o] = (uint8_t)child;
                if (ng > 0) {
                    const unsigned long long m = __ballot(li >= 0);
                    if (li >= 0) lst[(lhead + lcount + __popcll(m & ((1ull << lane) - 1ull))) & (LV_LIST - 1)] = (ti << 16) | (o << 5) | (uint32_t)li;
                    lcount += __popcll(m);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // ring entries are read by other lanes of this wave
            if (ng > 0) {
                if (b_n[0] == 0) { if (LV_FLAG_LOAD()) rendezvous(); }
                else {
                    accumulate(b_on[0], b_li[0], b_r[0], b_g[0]);
                    if (b_n[1] > 0) accumulate(b_on[1], b_li[1], b_r[1], b_g[1]);
                }
            }
#pragma unroll
            for (int s = 0; s < RPT; ++s) { cur_n[s] = nxt_n[s]; cur_r[s] = nxt_r[s]; cur_ib[s] = nxt_ib[s]; }
        }
        // leftovers in the ring
        while (ng > 0 && lcount > 0) {
            const int nb = lcount < 64 ? lcount : 64;
            const bool on = lane < nb;
            const uint32_t e = lst[(lhead + (on ? lane : 0)) & (LV_LIST - 1)];
            const long long row = on ? ((long long)blockIdx.x + (long long)(e >> 16) * gstep) * LV_TILE + ((e >> 5) & 0x7FFu) : 0;
            const uint4 br = recc[row]; const int2 bg = ghk[row];
            lhead += nb; lcount -= nb;
            accumulate(on, (int)(e & 31u), br, bg);
        }
    }
    // epilogue rendezvous: leave only when every wave has finished its rows and no drain is pending
    if (ng > 0) { while (rendezvous()) {} } else __syncthreads();
    if (STREAM && c.pad0 != 0) return;   // timing experiment: results are discarded
    // ---- flush this workgroup's partial histograms (plain stores: no global atomics, no zeroing)
    for (int li = 0; li < ng; ++li) {
        HistBin* dst = part + (((long long)k * c.gx + bx) * c.max_built + (g0 + li)) * c.totbins;
        for (int b = tid; b < wb; b += LV_THREADS) {
            const uint32_t ws = w_slot[b];
            const int sh = (int)(ws >> 24), s0 = (int)(ws & 0xFFFFFFu);
            long long tg = (long long)wide_g[li * wb + b] << LV_CARRY_SHIFT;
            long long th = (long long)(unsigned long long)wide_h[li * wb + b] << LV_CARRY_SHIFT;
            const unsigned long long* fb = fast + (size_t)li * spn + s0;
            for (int r2 = 0; r2 < (1 << sh); ++r2) { const unsigned long long v = fb[r2]; tg += (long long)(int)(v >> 32); th += (long long)(unsigned int)(v & 0xFFFFFFFFull); }
            HistBin o; o.g = tg; o.h = th;
            dst[w_hoff[b]] = o;
        }
    }
    if (writer) {
        for (int ci = tid; ci < 2 * n_exp; ci += LV_THREADS) {
            int tot = 0;
            for (int r2 = 0; r2 < LV_CNT_REP; ++r2) tot += cnt[ci * LV_CNT_REP + r2];
            if (tot) atomicAdd(&count[(long long)k * 256 + child_first + ci], tot);
        }
    }
    if (STREAM && ch == 0) {   // exact row counts of the built children of this group (k_level_plan numbers the children of parent ei as child_first + 2 ei, + 1)
        for (int ci = tid; ci < ng; ci += LV_THREADS) {
            int tot = 0;
            for (int r2 = 0; r2 < LV_CNT_REP; ++r2) tot += cnt[ci * LV_CNT_REP + r2];
            const int ei = g0 + ci;
            if (tot) atomicAdd(&count[(long long)k * 256 + child_first + 2 * ei + (pp->built_is_left[ei] ? 0 : 1)], tot);
        }
    }
}

#undef LV_FLAG_LOAD
#undef LV_FLAG_STORE


// ------------------------------------------------------------------------------------------------
// k_level_route (split mode): DataPartition::Split of a whole level for ALL class trees of a row tile.
// A lane owns 4 consecutive rows: their bin records are loaded ONCE and stay in registers while the
// lane walks the class trees of its slice -- per class tree it reads one dword of node ids, looks the
// (at most 64) nodes of the level up in an LDS route table and moves the rows to their children in place
// (only changed dwords are stored).  No (g,h), no histogram, no counting: 1-2 B per (row, class tree) of
// HBM traffic, ~40 VALU instructions per 64 (row, class tree) pairs (the fused pass spends ~100 on the
// same routing).  k_level_pass<STREAM> then builds the histograms of the built children.
// (A first cut also appended the built rows to per-class-tree lists here, for a gather-based accumulate:
// the returning global atomic per (wave tile, class tree) cost 1.9 of its 2.4 ms and the gathers moved as
// many HBM bytes as a full stream -- profiles/r02b_split_first_cut_profile.txt.)
// grid (persistent, ceil(K / RT_KS)), block RT_THREADS.
// ------------------------------------------------------------------------------------------------
template <int NCH /* 1, 2: the row's one / two 16-byte records live in registers; 0: any number of chunks, the split byte is gathered */>
__global__ __launch_bounds__(RT_THREADS) void k_level_route(const uint4* __restrict__ rec, uint8_t* __restrict__ node /* [K][NS], updated in place */,
                                                            const LvPlan* __restrict__ plan, LevelConst c) {
    __shared__ uint2 rt[RT_KS][64];
    __shared__ int s_base[RT_KS], s_live[RT_KS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.y * RT_KS;
    const int nk = (c.K - k0) < RT_KS ? (c.K - k0) : RT_KS;
    for (int i = tid; i < nk * 64; i += RT_THREADS) {
        const int kk = i >> 6, j = i & 63;
        const LvPlan* pp = &plan[k0 + kk];
        const bool live = !pp->done && pp->n_exp > 0;
        const int base = live ? (int)pp->exp[0] : 0;          // expanded parents are listed in ascending node id; a level has <= 64 nodes
        const int n = base + j;
        uint2 e = make_uint2(0u, 0u);
        if (live && n < 256) e = make_uint2(pp->route0[n], pp->route1[n]);
        rt[kk][j] = e;
        if (j == 0) { s_base[kk] = base; s_live[kk] = live ? 1 : 0; }
    }
    __syncthreads();
    const long long N = c.N, NS = c.NS;
    const long long nwt = (N + RT_WT_ROWS - 1) / RT_WT_ROWS;
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(rec);
    for (long long wt = (long long)blockIdx.x * (RT_THREADS / 64) + wave; wt < nwt; wt += (long long)gridDim.x * (RT_THREADS / 64)) {
        const long long row0 = wt * RT_WT_ROWS + lane * 4;
        const bool lane_on = row0 < N;
        uint4 r[4], r2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long long rr = row0 + j; if (rr >= N) rr = N - 1;
            if (NCH >= 1) r[j] = rec[rr]; else r[j] = make_uint4(0, 0, 0, 0);
            if (NCH == 2) r2[j] = rec[N + rr]; else r2[j] = make_uint4(0, 0, 0, 0);
        }
        // rows past the end of the table never take part (their node bytes are uninitialised)
        uint32_t rowmask = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (row0 + j < N) rowmask |= 0xFFu << (8 * j);
        // node ids of the NEXT class tree are requested before this class tree's store is issued
        uint32_t n4_next = 0xFFFFFFFFu;
        if (lane_on) n4_next = *reinterpret_cast<const uint32_t*>(node + (long long)k0 * NS + row0);
        for (int kk = 0; kk < nk; ++kk) {
            uint32_t* np = reinterpret_cast<uint32_t*>(node + (long long)(k0 + kk) * NS + row0);
            const uint32_t n4 = n4_next;
            if (kk + 1 < nk && lane_on) n4_next = *reinterpret_cast<const uint32_t*>(node + (long long)(k0 + kk + 1) * NS + row0);
            if (!s_live[kk]) continue;                                   // uniform
            const uint32_t base = (uint32_t)s_base[kk];
            uint32_t idx[4]; bool in[4]; bool any_in = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                idx[j] = ((n4 >> (8 * j)) & 0xFFu) - base;
                in[j] = idx[j] < 64u && ((rowmask >> (8 * j)) & 1u);
                any_in |= in[j];
            }
            if (__ballot(any_in) == 0ull) continue;                      // no row of this wave tile sits in a node of the level
            uint32_t out4 = n4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint2 e = rt[kk][idx[j] & 63u];
                const bool expd = in[j] && (e.x & (1u << 24)) != 0u;
                const unsigned f = e.x & 0xFFu;
                unsigned bin;
                if (NCH == 0) {
                    bin = 0u;
                    if (expd) bin = rec8[((long long)(f >> 4) * N + row0 + j) * 16 + (f & 15u)];
                } else {
                    uint32_t rx = r[j].x, ry = r[j].y, rz = r[j].z, rw = r[j].w;
                    if (NCH == 2) { const bool second = (f >> 4) != 0u; rx = second ? r2[j].x : rx; ry = second ? r2[j].y : ry; rz = second ? r2[j].z : rz; rw = second ? r2[j].w : rw; }
                    const bool hi = (f & 8u) != 0u;
                    const uint32_t lo32 = hi ? rz : rx, hi32 = hi ? rw : ry;
                    bin = __builtin_amdgcn_perm(hi32, lo32, (f & 7u) | 0x0C0C0C00u);   // byte (f & 15) of the record
                }
                const bool left = (bin == ((e.x >> 16) & 0xFFu)) ? ((e.x >> 25) & 1u) != 0u : (bin < ((e.x >> 8) & 0xFFu));
                const unsigned sel = left ? e.y : (e.y >> 8);          // child in bits 0..7
                if (expd) out4 = (out4 & ~(0xFFu << (8 * j))) | ((sel & 0xFFu) << (8 * j));
            }
            if (out4 != n4) *np = out4;
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
__global__ __launch_bounds__(256) void k_pack_joint(const uint4* __restrict__ rec, long long N, int F, const JointFeat* __restrict__ jf, uint4* __restrict__ rec_joint) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(rec);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int f = 0; f < F; ++f) {
        const uint32_t bin = rec8[((long long)(f >> 4) * N + i) * 16 + (f & 15)];
        const int vb = jf[f].vbyte;
        w[vb >> 2] += (bin * (uint32_t)jf[f].stride) << (8 * (vb & 3));    // a group's code stays below 256: no carry into the next byte
    }
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
                                                     int n_hnodes, TrainConst c, LevelConst lc) {
    const uint8_t* used = used_all + (long long)(*itp) * c.K * c.F;
    const int k = blockIdx.z, pi = blockIdx.y;
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
    if (lc.split_mode) {
        // only the built child was counted (k_level_pass<STREAM>); its sibling holds the rest of the parent's rows.  The derived
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
__global__ __launch_bounds__(256) void k_level_plan(LvPlan* __restrict__ plan, LvLayout* __restrict__ layout, const LvLayout* __restrict__ lay_table, SNode* __restrict__ nodes,
                                                    const Cand* __restrict__ cand, const FeatMeta* __restrict__ fmeta,
                                                    const ChunkMeta* __restrict__ cmeta, int level, TrainConst c, LevelConst lc) {
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
    // 4. LDS layout of the pass
    const int n_built = with_hist ? n_exp : 0;
    int npg = 1, n_groups = 1;
    if (n_built > 0) {
        npg = n_built;
        for (int chn = 0; chn < c.nchunk; ++chn) {
            const ChunkMeta cm = cmeta[chn]; const FeatMeta* fm = fmeta + cm.first_feat;
            const long long avail = lc.lds_bytes - lv_fixed_bytes(cm, LV_MAX_EXP, fm);
            long long fit = (avail - (long long)lv_slots(fm, cm.nfeat, 0) * 2) / lv_node_bytes(fm, cm, 0);
            if (fit < 1) fit = 1;
            if (fit < npg) npg = (int)fit;
        }
        n_groups = (n_built + npg - 1) / npg;
    }
    if (lane < c.nchunk) {
        layout[(long long)k * c.nchunk + lane] = lay_table[lane * (LV_MAX_BUILT + 1) + (npg < 1 ? 1 : npg)];   // precomputed on the host (lv_choose_layout)
    }
    if (lane == 0) {
        pp->n_exp = n_exp; pp->n_built = n_built; pp->npg = npg; pp->n_groups = n_groups;
        pp->child_first = child_first; pp->n_nodes = child_first + 2 * n_exp;
        pp->lvl_first = child_first; pp->lvl_end = child_first + 2 * n_exp;
        if (with_hist) pp->n_hslots = hs0 + 2 * n_exp;
        pp->buf_in = pp->buf; pp->buf = 1 - pp->buf;
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
template <bool SCORE /* false: store the final node ids instead; the next k_grad_mc applies the deltas (lazy AddScore) */>
__global__ __launch_bounds__(256) void k_level_final(const uint4* __restrict__ rec, uint8_t* __restrict__ node_a, uint8_t* __restrict__ node_b,
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
    if (!SCORE && !route) return;                      // the ids in pp->buf are final already
    const int n_exp = route ? pp->n_exp : 0, child_first = pp->child_first;
    const int tid = threadIdx.x, lane = tid & 63;
    nd[tid] = node_delta[(long long)k * 256 + tid];
    route0[tid] = route ? pp->route0[tid] : 0u; route1[tid] = pp->route1[tid];
    for (int i = tid; i < 2 * n_exp * LV_CNT_REP; i += 256) cnt[i] = 0;
    __syncthreads();
    const long long N = c.N;
    const uint8_t* node = ((route ? pp->buf_in : pp->buf) ? node_b : node_a) + (long long)k * c.NS;
    uint8_t* node_out = (pp->buf ? node_b : node_a) + (long long)k * c.NS;   // !SCORE: receives the routed ids
    const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(rec);
    double* sk = score + (long long)k * N;
    // 4 rows per thread; node ids and scores are loaded together (independent loads), then routed and written back
    const bool aligned16 = (((unsigned long long)sk) & 15ull) == 0ull;   // uniform
    for (long long i = ((long long)blockIdx.x * 256 + tid) * 4; i < N; i += (long long)gridDim.x * 1024) {
        if (i + 3 < N && aligned16) {
            const uint32_t n4 = *reinterpret_cast<const uint32_t*>(node + i);
            double2 s01 = make_double2(0, 0), s23 = make_double2(0, 0);
            if (SCORE) { s01 = *reinterpret_cast<const double2*>(sk + i); s23 = *reinterpret_cast<const double2*>(sk + i + 2); }
            if (n4 == 0xFFFFFFFFu) { if (!SCORE) *reinterpret_cast<uint32_t*>(node_out + i) = n4; continue; }
            double sv[4] = {s01.x, s01.y, s23.x, s23.y};
            uint32_t o4 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int n = (int)((n4 >> (8 * j)) & 0xFFu);
                if (n == LV_INACTIVE) { o4 |= 0xFFu << (8 * j); continue; }
                const long long row = i + j;
                const uint32_t w0 = route0[n];
                if (w0 & (1u << 24)) {
                    const int f = (int)(w0 & 0xFFu), theta1 = (int)((w0 >> 8) & 0xFFu), nanbin = (int)((w0 >> 16) & 0xFFu);
                    const int bin = (int)rec8[((long long)(f >> 4) * N + row) * 16 + (f & 15)];
                    const bool left = (bin == nanbin) ? ((w0 >> 25) & 1u) != 0u : (bin < theta1);
                    const uint32_t w1 = route1[n];
                    n = left ? (int)(w1 & 0xFFu) : (int)((w1 >> 8) & 0xFFu);
                    if (!inbag || inbag[row]) atomicAdd(&cnt[(n - child_first) * LV_CNT_REP + (lane & (LV_CNT_REP - 1))], 1);
                }
                if (SCORE) sv[j] += nd[n]; else o4 |= (uint32_t)n << (8 * j);
            }
            if (SCORE) {
                s01.x = sv[0]; s01.y = sv[1]; s23.x = sv[2]; s23.y = sv[3];
                *reinterpret_cast<double2*>(sk + i) = s01; *reinterpret_cast<double2*>(sk + i + 2) = s23;
            } else *reinterpret_cast<uint32_t*>(node_out + i) = o4;
            continue;
        }
        for (long long row = i; row < N && row < i + 4; ++row) {
            int n = node[row];
            if (n == LV_INACTIVE) { if (!SCORE) node_out[row] = (uint8_t)LV_INACTIVE; continue; }
            const uint32_t w0 = route0[n];
            if (w0 & (1u << 24)) {
                const int f = (int)(w0 & 0xFFu), theta1 = (int)((w0 >> 8) & 0xFFu), nanbin = (int)((w0 >> 16) & 0xFFu);
                const int bin = (int)rec8[((long long)(f >> 4) * N + row) * 16 + (f & 15)];
                const bool left = (bin == nanbin) ? ((w0 >> 25) & 1u) != 0u : (bin < theta1);
                const uint32_t w1 = route1[n];
                n = left ? (int)(w1 & 0xFFu) : (int)((w1 >> 8) & 0xFFu);
                if (!inbag || inbag[row]) atomicAdd(&cnt[(n - child_first) * LV_CNT_REP + (lane & (LV_CNT_REP - 1))], 1);
            }
            if (SCORE) sk[row] += nd[n]; else node_out[row] = (uint8_t)n;
        }
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
