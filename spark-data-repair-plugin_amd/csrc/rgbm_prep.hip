// rgbm_prep.hip -- the relational steps either side of the repair models, on the HBM-resident code table
// (SURVEY.md 8(f) rows 2-4).  Everything here is HBM-bound integer work over int32 codes [c][n]:
//
//   detect_nulls          NullErrorDetector                      (reference: ErrorDetectorApi.scala:128-157)
//   detect_constraint     ConstraintErrorDetector for  X1..Xm -> Y  style denial constraints
//                         t1&t2&EQ(t1.X,t2.X)&..&IQ(t1.Y,t2.Y)     (ErrorDetectorApi.scala:189-244)
//   null_cells            convertErrorCellsToNull                 (RepairApi.scala:171-211)
//   rows_of_cells         clean / dirty row split                 (python/repair/model.py:549-553)
//   gather_rows           the dirty-row table
//   count_codes           per-code row counts of a column (class weights, domain statistics)
//   encode_dict           dictionary indices -> sorted-rank codes (replaces the pandas encoders, model.py:701-729)
//
// Result lists are ORDERED (by position in the column list, then ascending row), so the output is a deterministic
// function of the input: the device stream compaction is two passes (coalesced flag pass that leaves 64-row ballots
// behind, exclusive scan of the per-block counts, emit pass over the ballots) and never uses arrival order.
#include "rgbm_host.h"

#include <algorithm>
#include <cstring>

namespace {

using namespace rgh;

constexpr int PB = 256;                 // threads per block
constexpr int PSUB = 16;                // 256-row sub-tiles per block
constexpr int PROWS = PB * PSUB;        // 4096 rows per block
constexpr int PBAL = PROWS / 64;        // 64 ballots per block

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// ---------------------------------------------------------------------------------------------
// stream compaction, pass 1: flags -> 64-row ballots + per-block count.   grid (nblk, ncols)
//   MODE 0: flag = cell of column cols[blockIdx.y] is NULL;  MODE 1: flag = mask[row] != 0 (blockIdx.y == 0)
// Algorithmic bytes: 4 B per (row, column) for MODE 0, 1 B per row for MODE 1.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(PB) void k_flag(const int32_t* __restrict__ codes, const uint8_t* __restrict__ mask,
                                             const int32_t* __restrict__ cols, long long n, long long nblk,
                                             unsigned long long* __restrict__ ballots, unsigned* __restrict__ bcount) {
    const long long b = blockIdx.x;
    const int j = blockIdx.y;
    const long long base = b * PROWS;
    const int32_t* col = MODE == 0 ? codes + (long long)cols[j] * n : nullptr;
    bool f[PSUB];
#pragma unroll
    for (int s = 0; s < PSUB; ++s) {             // 16 independent coalesced loads in flight per lane
        const long long r = base + (long long)s * PB + threadIdx.x;
        if (MODE == 0) f[s] = r < n ? (col[r] < 0) : false;
        else f[s] = r < n ? (mask[r] != 0) : false;
    }
    __shared__ unsigned wsum[PB / 64];
    unsigned cnt = 0;
    unsigned long long* bo = ballots + ((long long)j * nblk + b) * PBAL;
#pragma unroll
    for (int s = 0; s < PSUB; ++s) {
        const unsigned long long m = __ballot(f[s]);
        if (lane_id() == 0) { bo[s * (PB / 64) + (threadIdx.x >> 6)] = m; cnt += (unsigned)__popcll(m); }
    }
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned t = 0; for (int w = 0; w < PB / 64; ++w) t += wsum[w]; bcount[(long long)j * nblk + b] = t; }
}

// exclusive scan of m block counts (one workgroup; 8 entries per thread and step)
__global__ __launch_bounds__(1024) void k_scan_counts(const unsigned* __restrict__ cnt, long long m, long long* __restrict__ off,
                                                      long long* __restrict__ total) {
    __shared__ long long wtot[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (long long base = 0; base < m; base += 8192) {
        long long v[8]; long long s = 0;
        const long long i0 = base + (long long)tid * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = (i0 + k < m) ? (long long)cnt[i0 + k] : 0; s += v[k]; }
        long long incl = s;                               // inclusive wave scan of the per-thread sums
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63) wtot[w] = incl;
        __syncthreads();
        long long wbase = 0;
        for (int q = 0; q < w; ++q) wbase += wtot[q];
        long long run = carry_s + wbase + incl - s;
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (i0 + k < m) off[i0 + k] = run; run += v[k]; }
        __syncthreads();
        if (tid == 1023) carry_s = run;
        __syncthreads();
    }
    if (tid == 0) *total = carry_s;
}

// pass 2: ballots -> ordered (row, column) cells.   grid (nblk, ncols), one workgroup of 64 lanes x 4 waves
__global__ __launch_bounds__(PB) void k_emit(const unsigned long long* __restrict__ ballots, const long long* __restrict__ off,
                                             const int32_t* __restrict__ cols, long long nblk,
                                             long long* __restrict__ out_rows, int32_t* __restrict__ out_cols) {
    const long long b = blockIdx.x;
    const int j = blockIdx.y;
    const long long e = (long long)j * nblk + b;
    __shared__ unsigned pre[PBAL];
    __shared__ unsigned long long bal[PBAL];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < PBAL) {                        // wave 0: exclusive scan of the 64 ballot popcounts
        const unsigned long long m = ballots[e * PBAL + tid];
        bal[tid] = m;
        const unsigned c = (unsigned)__popcll(m);
        unsigned incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        pre[tid] = incl - c;
    }
    __syncthreads();
    const long long o0 = off[e];
    const int32_t cj = cols ? cols[j] : -1;
#pragma unroll
    for (int s = 0; s < PSUB; ++s) {
        const int q = s * (PB / 64) + w;     // ballot q covers rows [b*4096 + q*64, +64): ascending in q
        const unsigned long long m = bal[q];
        if ((m >> lane) & 1ull) {
            const long long p = o0 + pre[q] + __popcll(m & ((1ull << lane) - 1ull));
            out_rows[p] = b * PROWS + (long long)q * 64 + lane;
            if (out_cols) out_cols[p] = cj;
        }
    }
}

// cells = rows x columns (column-major): the constraint detector reports every given attribute of a violating row
__global__ void k_replicate(const long long* __restrict__ rows, long long m, const int32_t* __restrict__ cols, int ncols,
                            long long* __restrict__ out_rows, int32_t* __restrict__ out_cols) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const long long r = rows[i];
    for (int j = 0; j < ncols; ++j) { out_rows[(long long)j * m + i] = r; out_cols[(long long)j * m + i] = cols[j]; }
}

// ---------------------------------------------------------------------------------------------
// denial constraints EQ(X1)..EQ(Xm) & IQ(Y) on two tuples: open-addressing hash table keyed by the NULL-safe
// mixed-radix code of (X1..Xm); a group remembers the first Y it has seen and a flag "another Y exists"
// .  The violation mask is a pure function of the table -- insertion order is irrelevant.
// Algorithmic bytes per row: 4 B per EQ/IQ column + ~12 B of table traffic per probe, twice (insert, lookup).
// ---------------------------------------------------------------------------------------------
struct KeySpec { int32_t ncols; int32_t col[12]; unsigned long long radix[12]; };
constexpr unsigned long long HT_EMPTY = ~0ull;

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x;
}
__device__ __forceinline__ unsigned long long row_key(const int32_t* __restrict__ codes, long long n, const KeySpec& ks, long long i) {
    unsigned long long k = 0;
    for (int c = 0; c < ks.ncols; ++c) k = k * ks.radix[c] + (unsigned long long)(codes[(long long)ks.col[c] * n + i] + 1);   // NULL (-1) -> digit 0
    return k;
}

__global__ __launch_bounds__(256) void k_ht_insert(const int32_t* __restrict__ codes, long long n, KeySpec ks, int iq_col,
                                                   unsigned long long* __restrict__ keys, unsigned* __restrict__ state, unsigned long long cap_mask) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = row_key(codes, n, ks, i);
    unsigned long long slot = mix64(key) & cap_mask;
    // A slot only ever goes EMPTY -> key, and its state only grows (0 -> first value -> | "another value"),
    // so a plain read that already shows the final answer makes the atomic unnecessary: low-cardinality keys (few
    // groups, millions of rows each) would otherwise serialise on a handful of addresses.
    for (;;) {
        unsigned long long prev = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read at L2, never a stale L1 line
        if (prev == key) break;
        if (prev == HT_EMPTY) { prev = atomicCAS(&keys[slot], HT_EMPTY, key); if (prev == HT_EMPTY || prev == key) break; }
        slot = (slot + 1) & cap_mask;
    }
    const unsigned v = (unsigned)(codes[(long long)iq_col * n + i] + 1) + 1u;            // >= 1; NULL is a value of its own (<=>)
    unsigned old = __hip_atomic_load(&state[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old >> 31) || old == v) return;
    if (old == 0u) old = atomicCAS(&state[slot], 0u, v);
    if (old != 0u && (old & 0x7FFFFFFFu) != v) atomicOr(&state[slot], 0x80000000u);
}

__global__ __launch_bounds__(256) void k_ht_lookup(const int32_t* __restrict__ codes, long long n, KeySpec ks,
                                                   const unsigned long long* __restrict__ keys, const unsigned* __restrict__ state,
                                                   unsigned long long cap_mask, uint8_t* __restrict__ mask) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = row_key(codes, n, ks, i);
    unsigned long long slot = mix64(key) & cap_mask;
    while (keys[slot] != key) slot = (slot + 1) & cap_mask;
    const unsigned st = state[slot];
    mask[i] = (uint8_t)(st >> 31);
}

// ---------------------------------------------------------------------------------------------
// small scatter / gather kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_null_cells(int32_t* __restrict__ codes, long long n, int c, const long long* __restrict__ rows,
                             const int32_t* __restrict__ cols, long long m, const uint8_t* __restrict__ is_target) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const long long r = rows[i]; const int cc = cols[i];
    if (r < 0 || r >= n || cc < 0 || cc >= c || !is_target[cc]) return;     // cells outside the table / the targets are ignored (join semantics)
    codes[(long long)cc * n + r] = -1;
}

__global__ void k_write_cells(int32_t* __restrict__ codes, long long n, int c, const long long* __restrict__ rows,
                              const int32_t* __restrict__ cols, const int32_t* __restrict__ vals, long long m) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const long long r = rows[i]; const int cc = cols[i];
    if (r < 0 || r >= n || cc < 0 || cc >= c) return;
    codes[(long long)cc * n + r] = vals[i];
}

__global__ void k_read_cells(const int32_t* __restrict__ codes, long long n, int c, const long long* __restrict__ rows,
                             const int32_t* __restrict__ cols, long long m, int32_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const long long r = rows[i]; const int cc = cols[i];
    out[i] = (r < 0 || r >= n || cc < 0 || cc >= c) ? -1 : codes[(long long)cc * n + r];
}

__global__ void k_mark_rows(uint8_t* __restrict__ mask, long long n, const long long* __restrict__ rows, long long m) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const long long r = rows[i];
    if (r >= 0 && r < n) mask[r] = 1;
}

__global__ void k_gather_rows(const int32_t* __restrict__ in, long long n_in, int32_t* __restrict__ out, long long n_out,
                              const long long* __restrict__ rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const int cc = blockIdx.y;
    out[(long long)cc * n_out + i] = in[(long long)cc * n_in + rows[i]];
}

// per-code counts of one column: LDS histogram per workgroup when the domain fits, flushed with global atomics
constexpr int CC_LDS = 8192;
__global__ __launch_bounds__(256) void k_count_codes(const int32_t* __restrict__ col, long long n, int n_codes,
                                                     unsigned long long* __restrict__ counts /* [n_codes + 1], last = NULL */) {
    __shared__ unsigned h[CC_LDS + 1];
    const bool lds = n_codes <= CC_LDS;
    if (lds) { for (int i = threadIdx.x; i <= n_codes; i += blockDim.x) h[i] = 0; __syncthreads(); }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int v = col[i];
        const int b = (v < 0 || v >= n_codes) ? n_codes : v;
        if (lds) atomicAdd(&h[b], 1u); else atomicAdd(&counts[b], 1ull);
    }
    if (lds) {
        __syncthreads();
        for (int i = threadIdx.x; i <= n_codes; i += blockDim.x) if (h[i]) atomicAdd(&counts[i], (unsigned long long)h[i]);
    }
}

// dictionary indices -> codes through a per-column remap table (index < 0 or >= dict size -> NULL)
__global__ void k_encode_dict(const int32_t* __restrict__ idx, long long n, const int32_t* __restrict__ remap,
                              const long long* __restrict__ remap_off, const int32_t* __restrict__ dict_size, int32_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cc = blockIdx.y;
    const int v = idx[(long long)cc * n + i];
    out[(long long)cc * n + i] = (v < 0 || v >= dict_size[cc]) ? -1 : remap[remap_off[cc] + v];
}

// ---------------------------------------------------------------------------------------------
// candidate distributions (python/repair/model.py:1196-1212): per cell, classes by descending probability (ties keep
// class order), those > threshold, at most top_k.  One wave per cell: every step each lane proposes its best class
// that comes AFTER the previous pick in (prob desc, class asc) order, a wave arg-max picks the next one.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_top_k_pmf(const double* __restrict__ proba, long long m, int K, int top_k, double threshold,
                                                   const int32_t* __restrict__ cur_code, int32_t* __restrict__ cls_out,
                                                   double* __restrict__ prob_out, double* __restrict__ cur_prob_out) {
    const long long cell = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (cell >= m) return;
    const int lane = lane_id();
    const double* p = proba + cell * K;
    double last_p = INFINITY; int last_c = -1;
    for (int j = 0; j < top_k; ++j) {
        double bp = -1.0; int bc = 0x7FFFFFFF;
        for (int c = lane; c < K; c += 64) {
            const double v = p[c];
            const bool after = v < last_p || (v == last_p && c > last_c);
            if (after && v > threshold && (v > bp || (v == bp && c < bc))) { bp = v; bc = c; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double op = __shfl_xor(bp, d); const int oc = __shfl_xor(bc, d);
            if (op > bp || (op == bp && oc < bc)) { bp = op; bc = oc; }
        }
        const bool found = bc != 0x7FFFFFFF;
        if (lane == 0) { cls_out[cell * top_k + j] = found ? bc : -1; prob_out[cell * top_k + j] = found ? bp : 0.0; }
        if (!found) {                                   // nothing left above the threshold: pad the rest
            for (int r = j + 1 + lane; r < top_k; r += 64) { cls_out[cell * top_k + r] = -1; prob_out[cell * top_k + r] = 0.0; }
            break;
        }
        last_p = bp; last_c = bc;
    }
    if (cur_prob_out && lane == 0) {
        const int cc = cur_code ? cur_code[cell] : -1;
        cur_prob_out[cell] = (cc >= 0 && cc < K) ? p[cc] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
inline unsigned nblocks(long long n, int per) { return (unsigned)((n + per - 1) / per); }

hipStream_t table_stream(const rgbm_table& t) {
    if (!t.stream) HIPCHK(hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking));
    return t.stream;
}
// scratch slot `slot` of the table, at least `count` elements of T (grown, never shrunk)
template <typename T>
T* scr(const rgbm_table& t, int slot, size_t count) {
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (t.scratch[slot].n < bytes) t.scratch[slot].alloc(bytes + bytes / 4);
    return reinterpret_cast<T*>(t.scratch[slot].p);
}
template <typename T>
T* scr_upload(const rgbm_table& t, int slot, const T* host, size_t count, hipStream_t s) {
    T* d = scr<T>(t, slot, count);
    if (count) HIPCHK(hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, s));
    return d;
}

// ordered compaction of the flagged (row, column) cells into t.cell_rows / t.cell_cols; returns the cell count
template <int MODE>
long long compact(rgbm_table& t, const uint8_t* d_mask, const int32_t* d_cols, int ncols, bool want_cols, hipStream_t s) {
    const long long n = t.n, nblk = (n + PROWS - 1) / PROWS, m = nblk * ncols;
    unsigned long long* ballots = scr<unsigned long long>(t, 0, (size_t)m * PBAL);
    unsigned* bcount = scr<unsigned>(t, 1, (size_t)m);
    long long* off = scr<long long>(t, 2, (size_t)m + 1);
    hipLaunchKernelGGL(k_flag<MODE>, dim3((unsigned)nblk, (unsigned)ncols), dim3(PB), 0, s, t.codes.p, d_mask, d_cols, n, nblk, ballots, bcount);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, bcount, m, off, off + m);
    long long total = 0;
    HIPCHK(hipMemcpyAsync(&total, off + m, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (t.cell_rows.n < (size_t)std::max<long long>(total, 1)) t.cell_rows.alloc((size_t)std::max<long long>(total, 1) * 5 / 4);
    if (want_cols) { if (t.cell_cols.n < (size_t)std::max<long long>(total, 1)) t.cell_cols.alloc((size_t)std::max<long long>(total, 1) * 5 / 4); }
    else t.cell_cols.release();
    if (total > 0)
        hipLaunchKernelGGL(k_emit, dim3((unsigned)nblk, (unsigned)ncols), dim3(PB), 0, s, ballots, off, want_cols ? d_cols : nullptr, nblk,
                           t.cell_rows.p, want_cols ? t.cell_cols.p : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    t.n_cells = total;
    return total;
}

void check_cols(const rgbm_table& t, const int32_t* cols, int n, const char* what) {
    for (int i = 0; i < n; ++i) if (cols[i] < 0 || cols[i] >= t.c) throw std::invalid_argument(std::string(what) + ": column index out of range");
}

}  // namespace

extern "C" {

RGBM_EXPORT int rgbm_table_detect_nulls(rgbm_table* t, const int32_t* cols, int32_t n_cols, int64_t* n_cells_out) {
    if (!t || !n_cells_out || n_cols < 0 || (n_cols > 0 && !cols)) return fail(RGBM_ERR_ARG, "rgbm_table_detect_nulls: bad argument");
    return guarded([&]() {
        use_device(t->device);
        check_cols(*t, cols, n_cols, "rgbm_table_detect_nulls");
        if (n_cols == 0) { t->n_cells = 0; *n_cells_out = 0; return RGBM_OK; }
        std::lock_guard<std::mutex> prep_lk(t->prep_mu); hipStream_t s = table_stream(*t);
        const int32_t* d_cols = scr_upload<int32_t>(*t, 6, cols, (size_t)n_cols, s);
        *n_cells_out = compact<0>(*t, nullptr, d_cols, n_cols, true, s);
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_detect_constraint(rgbm_table* t, const int32_t* eq_cols, int32_t n_eq, int32_t iq_col,
                                             const int32_t* cell_cols, int32_t n_cell_cols, int64_t* n_rows_out, int64_t* n_cells_out) {
    if (!t || n_eq < 0 || n_eq > 12 || (n_eq > 0 && !eq_cols) || n_cell_cols < 0 || (n_cell_cols > 0 && !cell_cols) || !n_cells_out)
        return fail(RGBM_ERR_ARG, "rgbm_table_detect_constraint: bad argument (at most 12 EQ attributes)");
    return guarded([&]() {
        use_device(t->device);
        check_cols(*t, eq_cols, n_eq, "rgbm_table_detect_constraint");
        check_cols(*t, cell_cols, n_cell_cols, "rgbm_table_detect_constraint");
        if (iq_col < 0 || iq_col >= t->c) throw std::invalid_argument("rgbm_table_detect_constraint: IQ column out of range");
        KeySpec ks; memset(&ks, 0, sizeof(ks)); ks.ncols = n_eq;
        unsigned __int128 span = 1;
        for (int i = 0; i < n_eq; ++i) {
            ks.col[i] = eq_cols[i]; ks.radix[i] = (unsigned long long)t->n_codes[eq_cols[i]] + 1ull;
            span *= ks.radix[i];
            if (span >= ((unsigned __int128)1 << 63)) throw std::invalid_argument("rgbm_table_detect_constraint: the EQ attributes span more than 2^63 value combinations");
        }
        std::lock_guard<std::mutex> prep_lk(t->prep_mu); hipStream_t s = table_stream(*t);
        const long long n = t->n;
        // the table never needs more slots than twice the number of possible keys
        unsigned long long want = (unsigned long long)n * 2ull;
        if (span < (unsigned __int128)want) want = (unsigned long long)span * 2ull;
        unsigned long long cap = 1024; while (cap < want) cap <<= 1;
        unsigned long long* keys = scr<unsigned long long>(*t, 3, (size_t)cap);
        unsigned* state = scr<unsigned>(*t, 4, (size_t)cap);
        uint8_t* mask = scr<uint8_t>(*t, 5, (size_t)n);
        HIPCHK(hipMemsetAsync(keys, 0xFF, (size_t)cap * 8, s));
        HIPCHK(hipMemsetAsync(state, 0, (size_t)cap * 4, s));
        const unsigned nb = nblocks(n, 256);
        hipLaunchKernelGGL(k_ht_insert, dim3(nb), dim3(256), 0, s, t->codes.p, n, ks, iq_col, keys, state, cap - 1);
        hipLaunchKernelGGL(k_ht_lookup, dim3(nb), dim3(256), 0, s, t->codes.p, n, ks, keys, state, cap - 1, mask);
        const long long m = compact<1>(*t, mask, nullptr, 1, false, s);     // ascending violating rows
        if (n_rows_out) *n_rows_out = m;
        if (n_cell_cols > 0) {
            const size_t tot = (size_t)std::max<long long>(m * n_cell_cols, 1);
            DevBuf<long long> rows_r(tot); DevBuf<int32_t> cols_r(tot);
            const int32_t* d_cc = scr_upload<int32_t>(*t, 6, cell_cols, (size_t)n_cell_cols, s);
            if (m > 0) hipLaunchKernelGGL(k_replicate, dim3(nblocks(m, 256)), dim3(256), 0, s, t->cell_rows.p, m, d_cc, n_cell_cols, rows_r.p, cols_r.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(s));
            t->cell_rows.swap(rows_r);
            t->cell_cols.swap(cols_r);
            t->n_cells = m * n_cell_cols;
        }
        *n_cells_out = t->n_cells;
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_rows_of_cells(rgbm_table* t, const int64_t* rows, int64_t n_cells, int64_t* n_rows_out) {
    if (!t || !n_rows_out || n_cells < 0 || (n_cells > 0 && !rows)) return fail(RGBM_ERR_ARG, "rgbm_table_rows_of_cells: bad argument");
    return guarded([&]() {
        use_device(t->device);
        std::lock_guard<std::mutex> prep_lk(t->prep_mu); hipStream_t s = table_stream(*t);
        uint8_t* mask = scr<uint8_t>(*t, 5, (size_t)t->n);
        HIPCHK(hipMemsetAsync(mask, 0, (size_t)t->n, s));
        static_assert(sizeof(long long) == sizeof(int64_t), "row positions are 64-bit");
        const long long* d_rows = scr_upload<long long>(*t, 7, reinterpret_cast<const long long*>(rows), (size_t)n_cells, s);
        if (n_cells > 0) hipLaunchKernelGGL(k_mark_rows, dim3(nblocks(n_cells, 256)), dim3(256), 0, s, mask, (long long)t->n, d_rows, (long long)n_cells);
        *n_rows_out = compact<1>(*t, mask, nullptr, 1, false, s);
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_cells_fetch(const rgbm_table* t, int64_t* rows_out, int32_t* cols_out) {
    if (!t) return fail(RGBM_ERR_ARG, "rgbm_table_cells_fetch: bad argument");
    return guarded([&]() {
        use_device(t->device);
        std::lock_guard<std::mutex> prep_lk(t->prep_mu);
        if (t->n_cells > 0) {
            if (rows_out) HIPCHK(hipMemcpy(rows_out, t->cell_rows.p, (size_t)t->n_cells * 8, hipMemcpyDeviceToHost));
            if (cols_out) {
                if (!t->cell_cols.p) throw std::invalid_argument("rgbm_table_cells_fetch: the last result is a row list (no columns)");
                HIPCHK(hipMemcpy(cols_out, t->cell_cols.p, (size_t)t->n_cells * 4, hipMemcpyDeviceToHost));
            }
        }
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_null_cells(rgbm_table* t, const int64_t* rows, const int32_t* cols, int64_t n_cells,
                                      const int32_t* target_cols, int32_t n_targets) {
    if (!t || n_cells < 0 || (n_cells > 0 && (!rows || !cols)) || n_targets < 0 || (n_targets > 0 && !target_cols))
        return fail(RGBM_ERR_ARG, "rgbm_table_null_cells: bad argument");
    return guarded([&]() {
        use_device(t->device);
        check_cols(*t, target_cols, n_targets, "rgbm_table_null_cells");
        if (n_cells == 0 || n_targets == 0) return RGBM_OK;
        std::lock_guard<std::mutex> prep_lk(t->prep_mu); hipStream_t s = table_stream(*t);
        std::vector<uint8_t> is_t((size_t)t->c, 0);
        for (int i = 0; i < n_targets; ++i) is_t[target_cols[i]] = 1;
        const uint8_t* d_t = scr_upload<uint8_t>(*t, 6, is_t.data(), is_t.size(), s);
        const long long* d_rows = scr_upload<long long>(*t, 7, reinterpret_cast<const long long*>(rows), (size_t)n_cells, s);
        const int32_t* d_cols = scr_upload<int32_t>(*t, 8, cols, (size_t)n_cells, s);
        hipLaunchKernelGGL(k_null_cells, dim3(nblocks(n_cells, 256)), dim3(256), 0, s, t->codes.p, (long long)t->n, (int)t->c, d_rows, d_cols,
                           (long long)n_cells, d_t);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));      // is_t and the caller's arrays are read by the async copies
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_write_cells(rgbm_table* t, const int64_t* rows, const int32_t* cols, const int32_t* codes, int64_t n_cells) {
    if (!t || n_cells < 0 || (n_cells > 0 && (!rows || !cols || !codes))) return fail(RGBM_ERR_ARG, "rgbm_table_write_cells: bad argument");
    return guarded([&]() {
        use_device(t->device);
        if (n_cells == 0) return RGBM_OK;
        for (int64_t i = 0; i < n_cells; ++i)
            if (cols[i] >= 0 && cols[i] < t->c && codes[i] >= t->n_codes[cols[i]]) throw std::invalid_argument("rgbm_table_write_cells: code outside the column's dictionary");
        std::lock_guard<std::mutex> prep_lk(t->prep_mu); hipStream_t s = table_stream(*t);
        const long long* d_rows = scr_upload<long long>(*t, 7, reinterpret_cast<const long long*>(rows), (size_t)n_cells, s);
        const int32_t* d_cols = scr_upload<int32_t>(*t, 8, cols, (size_t)n_cells, s);
        const int32_t* d_vals = scr_upload<int32_t>(*t, 9, codes, (size_t)n_cells, s);
        hipLaunchKernelGGL(k_write_cells, dim3(nblocks(n_cells, 256)), dim3(256), 0, s, t->codes.p, (long long)t->n, (int)t->c, d_rows, d_cols, d_vals, (long long)n_cells);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_read_cells(const rgbm_table* t, const int64_t* rows, const int32_t* cols, int64_t n_cells, int32_t* codes_out) {
    if (!t || n_cells < 0 || (n_cells > 0 && (!rows || !cols || !codes_out))) return fail(RGBM_ERR_ARG, "rgbm_table_read_cells: bad argument");
    return guarded([&]() {
        use_device(t->device);
        if (n_cells == 0) return RGBM_OK;
        std::lock_guard<std::mutex> prep_lk(t->prep_mu); hipStream_t s = table_stream(*t);
        const long long* d_rows = scr_upload<long long>(*t, 7, reinterpret_cast<const long long*>(rows), (size_t)n_cells, s);
        const int32_t* d_cols = scr_upload<int32_t>(*t, 8, cols, (size_t)n_cells, s);
        int32_t* d_out = scr<int32_t>(*t, 9, (size_t)n_cells);
        hipLaunchKernelGGL(k_read_cells, dim3(nblocks(n_cells, 256)), dim3(256), 0, s, t->codes.p, (long long)t->n, (int)t->c, d_rows, d_cols,
                           (long long)n_cells, d_out);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(codes_out, d_out, (size_t)n_cells * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_gather_rows(const rgbm_table* t, const int64_t* rows, int64_t n_rows, rgbm_table** out) {
    if (!t || !out || n_rows <= 0 || !rows) return fail(RGBM_ERR_ARG, "rgbm_table_gather_rows: bad argument (at least one row)");
    return guarded([&]() {
        use_device(t->device);
        for (int64_t i = 0; i < n_rows; ++i) if (rows[i] < 0 || rows[i] >= t->n) throw std::invalid_argument("rgbm_table_gather_rows: row position out of range");
        std::lock_guard<std::mutex> prep_lk(t->prep_mu); hipStream_t s = table_stream(*t);
        std::unique_ptr<rgbm_table> o(new rgbm_table());
        o->device = t->device; o->n = n_rows; o->c = t->c; o->n_codes = t->n_codes; o->col_values = t->col_values; o->col_kind = t->col_kind;
        o->codes.alloc((size_t)n_rows * t->c);
        const long long* d_rows = scr_upload<long long>(*t, 7, reinterpret_cast<const long long*>(rows), (size_t)n_rows, s);
        hipLaunchKernelGGL(k_gather_rows, dim3(nblocks(n_rows, 256), (unsigned)t->c), dim3(256), 0, s, t->codes.p, (long long)t->n, o->codes.p,
                           (long long)n_rows, d_rows);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        *out = o.release();
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_count_codes(const rgbm_table* t, int32_t col, int64_t* counts_out, int64_t* n_null_out) {
    if (!t || !counts_out || col < 0 || col >= t->c) return fail(RGBM_ERR_ARG, "rgbm_table_count_codes: bad argument");
    return guarded([&]() {
        use_device(t->device);
        std::lock_guard<std::mutex> prep_lk(t->prep_mu); hipStream_t s = table_stream(*t);
        const int nc = t->n_codes[col];
        unsigned long long* d_cnt = scr<unsigned long long>(*t, 9, (size_t)nc + 1);
        HIPCHK(hipMemsetAsync(d_cnt, 0, ((size_t)nc + 1) * 8, s));
        const unsigned nb = std::min<unsigned>(nblocks(t->n, 256 * 16), 256u * 8u);
        hipLaunchKernelGGL(k_count_codes, dim3(std::max(nb, 1u)), dim3(256), 0, s, t->codes.p + (size_t)col * t->n, (long long)t->n, nc, d_cnt);
        HIPCHK(hipGetLastError());
        std::vector<unsigned long long> h((size_t)nc + 1);
        HIPCHK(hipMemcpyAsync(h.data(), d_cnt, h.size() * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (int i = 0; i < nc; ++i) counts_out[i] = (int64_t)h[i];
        if (n_null_out) *n_null_out = (int64_t)h[nc];
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_create_dict(const int32_t* idx_colmajor, int64_t n, int32_t c, const int32_t* const* remap,
                                       const int32_t* dict_size, int32_t device_id, rgbm_table** out) {
    if (!idx_colmajor || !remap || !dict_size || !out || n <= 0 || c <= 0) return fail(RGBM_ERR_ARG, "rgbm_table_create_dict: bad argument");
    return guarded([&]() {
        use_device(device_id);
        StreamGuard sg;
        std::vector<long long> roff((size_t)c); std::vector<int32_t> flat; std::vector<int32_t> ncodes((size_t)c);
        for (int j = 0; j < c; ++j) {
            if (dict_size[j] < 0 || (dict_size[j] > 0 && !remap[j])) throw std::invalid_argument("rgbm_table_create_dict: bad dictionary");
            roff[j] = (long long)flat.size();
            int mx = -1;
            for (int v = 0; v < dict_size[j]; ++v) {
                const int r = remap[j][v];
                if (r < -1) throw std::invalid_argument("rgbm_table_create_dict: remap entries must be codes >= 0 or -1 (NULL)");
                flat.push_back(r); mx = std::max(mx, r);
            }
            ncodes[j] = std::max(mx + 1, 1);      // an all-NULL column still counts as a one-code domain (repair.encode does the same)
        }
        std::unique_ptr<rgbm_table> t(new rgbm_table());
        t->device = device_id; t->n = n; t->c = c; t->n_codes = ncodes;
        t->codes.alloc((size_t)n * c);
        DevBuf<int32_t> d_idx((size_t)n * c); HIPCHK(hipMemcpyAsync(d_idx.p, idx_colmajor, (size_t)n * c * 4, hipMemcpyHostToDevice, sg.s));
        DevBuf<int32_t> d_flat(std::max<size_t>(flat.size(), 1)); d_flat.upload(flat.data(), flat.size(), sg.s);
        DevBuf<long long> d_off((size_t)c); d_off.upload(roff.data(), roff.size(), sg.s);
        DevBuf<int32_t> d_ds((size_t)c); d_ds.upload(dict_size, (size_t)c, sg.s);
        hipLaunchKernelGGL(k_encode_dict, dim3(nblocks(n, 256), (unsigned)c), dim3(256), 0, sg.s, d_idx.p, (long long)n, d_flat.p, d_off.p, d_ds.p, t->codes.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(sg.s));
        *out = t.release();
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_repair_pmf(rgbm_table* t, const rgbm_model* m, int32_t target_col, const int32_t* feat_cols, int32_t f, int32_t top_k,
                                      double threshold, const int32_t* cur_code, int64_t cap, int64_t* n_cells_out, int64_t* rows_out,
                                      int32_t* class_out, double* prob_out, double* cur_prob_out) {
    if (!t || !m || !feat_cols || f <= 0 || target_col < 0 || target_col >= t->c || top_k <= 0 || !n_cells_out || cap < 0)
        return fail(RGBM_ERR_ARG, "rgbm_table_repair_pmf: bad argument");
    return guarded([&]() {
        use_device(t->device);
        check_cols(*t, feat_cols, f, "rgbm_table_repair_pmf");
        int32_t obj = 0, K = 0, F = 0;
        model_shape(m, &obj, &K, &F);
        if (obj == 2) throw std::invalid_argument("rgbm_table_repair_pmf: a regressor has no class distribution (model.py:1214-1221 handles continuous attributes)");
        if (F != f) throw std::invalid_argument("rgbm_table_repair_pmf: the model was trained on a different number of features");
        std::lock_guard<std::mutex> prep_lk(t->prep_mu); struct { hipStream_t s; } sg{table_stream(*t)};
        const int32_t* d_tc = scr_upload<int32_t>(*t, 6, &target_col, 1, sg.s);
        const long long cells = compact<0>(*t, nullptr, d_tc, 1, false, sg.s);      // ascending rows whose target cell is NULL
        *n_cells_out = cells;
        if (cells == 0) return RGBM_OK;
        if (cells > cap) throw std::invalid_argument("rgbm_table_repair_pmf: output capacity too small (n_cells_out holds the needed size)");
        if (!rows_out || !class_out || !prob_out) throw std::invalid_argument("rgbm_table_repair_pmf: output arrays missing");
        // the cells' rows as a small code block, scored in one go
        DevBuf<int32_t> sub((size_t)cells * t->c);
        hipLaunchKernelGGL(k_gather_rows, dim3(nblocks(cells, 256), (unsigned)t->c), dim3(256), 0, sg.s, t->codes.p, (long long)t->n, sub.p, cells, t->cell_rows.p);
        DevBuf<int32_t> d_fc((size_t)f); d_fc.upload(feat_cols, (size_t)f, sg.s);
        DevBuf<double> proba((size_t)cells * K);
        predict_proba_device(m, t->device, sg.s, sub.p, cells, d_fc.p, proba.p);
        DevBuf<int32_t> d_cls((size_t)cells * top_k); DevBuf<double> d_pr((size_t)cells * top_k);
        DevBuf<int32_t> d_cur; DevBuf<double> d_cp;
        if (cur_prob_out) { d_cp.alloc((size_t)cells); if (cur_code) { d_cur.alloc((size_t)cells); d_cur.upload(cur_code, (size_t)cells, sg.s); } }
        hipLaunchKernelGGL(k_top_k_pmf, dim3(nblocks(cells, 4)), dim3(256), 0, sg.s, proba.p, cells, (int)K, (int)top_k, threshold,
                           d_cur.p, d_cls.p, d_pr.p, d_cp.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(rows_out, t->cell_rows.p, (size_t)cells * 8, hipMemcpyDeviceToHost, sg.s));
        d_cls.download(class_out, (size_t)cells * top_k, sg.s);
        d_pr.download(prob_out, (size_t)cells * top_k, sg.s);
        if (cur_prob_out) d_cp.download(cur_prob_out, (size_t)cells, sg.s);
        HIPCHK(hipStreamSynchronize(sg.s));
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_shape(const rgbm_table* t, int64_t* n_out, int32_t* c_out, int32_t* n_codes_out) {
    if (!t) return fail(RGBM_ERR_ARG, "rgbm_table_shape: bad argument");
    if (n_out) *n_out = t->n;
    if (c_out) *c_out = t->c;
    if (n_codes_out) for (int j = 0; j < t->c; ++j) n_codes_out[j] = t->n_codes[j];
    return RGBM_OK;
}

}  // extern "C"
