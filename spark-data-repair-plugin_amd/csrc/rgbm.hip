// rgbm.hip -- librepairgbm.so: host orchestration + C-ABI (include/rgbm.h) over the gfx950 kernels.
//
// One training call = one HIP stream; the whole leaf-wise growth of the K class trees of a
// boosting iteration runs on the device in lock step (grid.y = class tree) without any host
// round trip: the per-tree control blocks (rg::TreeState) live in HBM and every kernel of the
// fixed launch sequence
//     hist -> split_find -> tree_step -> partition -> finish_split     (num_leaves-1 times)
// reads its work description from there.  The host only enqueues.
//
// There is NO CPU fallback in this library: without a HIP device every compute entry point
// returns RGBM_ERR_NO_DEVICE (the CPU oracle under oracle/ is test infrastructure and is never
// linked or loaded from here).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rgbm_host.h"
#include "rgbm_kernels.h"
#include "rgbm_level.h"
#include "rgbm_small.h"

#define RGBM_VERSION 220   // numerics spec v2.2: float32 (g, h) as LightGBM computes them, exact integer histogram sums on a fixed-point grid chosen per class tree and boosting iteration (rgbm_numerics.h)

namespace {

thread_local std::string g_err;
}  // namespace
std::string& rgh::last_error() { return g_err; }

// ---- caching device-memory pool (rgbm_host.h)
namespace {
struct PoolBlock { void* p; size_t bytes; unsigned long long stamp; };
struct DevPool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, PoolBlock> free_blocks;     // (device, bytes) -> block
    size_t cached = 0; unsigned long long clock = 0;
    size_t cap() {
        static const size_t c = [] { const char* e = getenv("RGBM_POOL_MB"); long long mb = e ? atoll(e) : 65536; return (size_t)std::max<long long>(mb, 0) << 20; }();
        return c;
    }
    // drop the least recently freed blocks until at most `keep` bytes stay cached (mu held)
    void trim_locked(size_t keep) {
        while (cached > keep && !free_blocks.empty()) {
            auto victim = free_blocks.begin();
            for (auto it = free_blocks.begin(); it != free_blocks.end(); ++it) if (it->second.stamp < victim->second.stamp) victim = it;
            int cur = 0; (void)hipGetDevice(&cur);
            if (cur != victim->first.first) (void)hipSetDevice(victim->first.first);
            (void)hipFree(victim->second.p);
            if (cur != victim->first.first) (void)hipSetDevice(cur);
            cached -= victim->second.bytes;
            free_blocks.erase(victim);
        }
    }
    ~DevPool() { /* process exit: the driver reclaims device memory; HIP may already be shut down here */ }
};
DevPool& pool() { static DevPool* p = new DevPool(); return *p; }
}  // namespace

// RGBM_GUARD=1 (debugging): every buffer sits between two 2 MB guard zones filled with 0xC3, and so is the slack between the bytes asked
// for and the end of the pool block; pool_free checks that nobody wrote there and says so on stderr.  A kernel that READS out of
// bounds sees the same 0xC3 bytes in every process instead of a neighbour's live data.
namespace {
constexpr size_t GUARD = 2u << 20;
bool guard_on() { static const bool g = getenv("RGBM_GUARD") != nullptr; return g; }
std::mutex g_guard_mu; std::map<void*, size_t> g_guard_bytes;
__global__ void k_guard_check(const unsigned char* p, size_t n, unsigned long long* bad /* [0] count, [1] first offset */) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != 0xC3) { atomicAdd(&bad[0], 1ull); atomicMin(&bad[1], (unsigned long long)i); }
}
// RGBM_TRACE (debugging): wrapping 64-bit sum of a buffer's words
__global__ void k_trace_sum(const unsigned long long* p, size_t nwords, unsigned long long* out) {
    unsigned long long a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) a += p[i] * (2 * (unsigned long long)i + 1);
    atomicAdd(out, a);
}
void* pool_alloc_raw(size_t bytes, size_t* block_bytes, int* device);
void pool_free_raw(void* p, size_t block_bytes, int device);
}  // namespace

void* rgh::pool_alloc(size_t bytes, size_t* block_bytes, int* device) {
    if (!guard_on()) return pool_alloc_raw(bytes, block_bytes, device);
    unsigned char* raw = static_cast<unsigned char*>(pool_alloc_raw(bytes + 2 * GUARD, block_bytes, device));
    if (hipMemset(raw, 0xC3, GUARD) != hipSuccess || hipMemset(raw + GUARD + bytes, 0xC3, *block_bytes - GUARD - bytes) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)
        throw std::runtime_error("RGBM_GUARD: hipMemset failed");
    { std::lock_guard<std::mutex> lk(g_guard_mu); g_guard_bytes[raw + GUARD] = bytes; }
    return raw + GUARD;
}

void rgh::pool_free(void* p, size_t block_bytes, int device) {
    if (!p) return;
    if (!guard_on()) { pool_free_raw(p, block_bytes, device); return; }
    size_t bytes = 0;
    { std::lock_guard<std::mutex> lk(g_guard_mu); auto it = g_guard_bytes.find(p); if (it != g_guard_bytes.end()) { bytes = it->second; g_guard_bytes.erase(it); } }
    unsigned char* raw = static_cast<unsigned char*>(p) - GUARD;
    static thread_local unsigned long long* d_bad = nullptr;
    if (!d_bad) (void)hipMalloc(&d_bad, 16);
    const size_t tail = block_bytes - GUARD - bytes;
    const struct { const unsigned char* q; size_t n; const char* what; } zones[2] = {{raw, GUARD, "front guard"}, {raw + GUARD + bytes, tail, "slack + back guard"}};
    for (const auto& z : zones) {
        unsigned long long h[2] = {0ull, ~0ull};
        (void)hipMemcpy(d_bad, h, 16, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_guard_check, dim3(1024), dim3(256), 0, nullptr, z.q, z.n, d_bad);
        (void)hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost);
        if (h[0]) fprintf(stderr, "[rgbm guard] buffer of %zu bytes (block %zu): %llu byte(s) of the %s were overwritten, first at offset %lld from the buffer's %s\n",
                          bytes, block_bytes, h[0], z.what, z.q == raw ? (long long)h[1] - (long long)GUARD : (long long)h[1], z.q == raw ? "start" : "end");
    }
    pool_free_raw(raw, block_bytes, device);
}

namespace {
void* pool_alloc_raw(size_t bytes, size_t* block_bytes, int* device) {
    int dev = 0; (void)hipGetDevice(&dev);
    *device = dev;
    DevPool& P = pool();
    // round up so that nearly equal requests can share blocks: 256 B below 1 MB, 2 MB above
    const size_t want = bytes < (1u << 20) ? ((bytes + 255) & ~(size_t)255) : ((bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1));
    if (P.cap() > 0) {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.free_blocks.lower_bound(std::make_pair(dev, want));
        // a request never swallows a much larger block (up to 8x its size, 4x below 1 MB)
        if (it != P.free_blocks.end() && it->first.first == dev && it->first.second <= want * (want >= (1u << 20) ? 8 : 4)) {
            void* p = it->second.p; *block_bytes = it->second.bytes;
            P.cached -= it->second.bytes;
            P.free_blocks.erase(it);
            return p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess && P.cap() > 0) {       // out of memory with blocks parked in the pool: give them back and retry once
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> lk(P.mu); P.trim_locked(0); }
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); char b[256]; snprintf(b, sizeof(b), "hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e)); throw std::runtime_error(b); }
    *block_bytes = want;
    return p;
}

void pool_free_raw(void* p, size_t block_bytes, int device) {
    if (!p) return;
    DevPool& P = pool();
    if (P.cap() == 0 || block_bytes > P.cap()) {
        int cur = 0; (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        (void)hipFree(p);
        if (cur != device) (void)hipSetDevice(cur);
        return;
    }
    std::lock_guard<std::mutex> lk(P.mu);
    P.free_blocks.emplace(std::make_pair(device, block_bytes), PoolBlock{p, block_bytes, ++P.clock});
    P.cached += block_bytes;
    if (P.cached > P.cap()) P.trim_locked(P.cap());
}

}  // namespace

void rgh::pool_trim(size_t keep_bytes) { DevPool& P = pool(); std::lock_guard<std::mutex> lk(P.mu); P.trim_locked(keep_bytes); }
namespace {
using rgh::fail; using rgh::DevBuf; using rgh::use_device; using rgh::StreamGuard; using rgh::guarded;

// ---------------------------------------------------------------------------------------------
// Row-sharded multi-GPU training (DESIGN.md "Multi-GPU"): every rank holds a row shard of the table and grows the
// SAME trees; the only exchanges are integer all-reduces of (a) the code counts used for binning, (b) the built
// children's histograms after each level pass and (c) the exact child row counts.  Sums are exact integers, so the
// model is bit-identical to single-GPU training for any number of ranks and any row split.
// Two transports: RCCL (one process per GPU, all-reduce enqueued on the training stream: no host round trip) and an
// in-process thread group (one rank per host thread on one device) that exists so the sharding logic can be
// verified on a single GPU.
// ---------------------------------------------------------------------------------------------
struct LocalGroup {
    int nranks = 0, device = 0;
    std::mutex mu; std::condition_variable cv;
    int arrived = 0; long long generation = 0;
    std::vector<void*> ptr;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const long long gen = generation;
        if (++arrived == nranks) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};

struct Fusion;
struct Comm {
    int kind = 0;   // 0 none, 1 RCCL, 2 local thread group, 3 member of a fusion group (the rank's communicator lives in the group)
    int rank = 0, nranks = 1;
    ncclComm_t nccl = nullptr;
    LocalGroup* lg = nullptr;
    void* tmp = nullptr; size_t tmp_bytes = 0;
    Fusion* fusion = nullptr; int member = 0;      // kind 3
};
thread_local Comm g_comm;

enum { AR_I64 = 0, AR_U32 = 1, AR_I32 = 2 };

// ---------------------------------------------------------------------------------------------
// Fusion group (round 5): the row-sharded training calls of ONE rank that run at the same time -- one host thread and one HIP stream
// each -- and exchange in LOCK STEP.  Every row-sharded training call makes the same sequence of collectives (code counts once, then per
// boosting iteration one all-reduce per level and one of the deepest counts: the sequence is fixed by the parameters, not by the
// data), so the i-th collective of every member is carried by ONE all-reduce of the rank's communicator over the members' buffers laid
// side by side: a member records an event behind its producers and waits at a host barrier; the last one to arrive copies all parts
// into a staging buffer on the group's stream, enqueues the collective and the copies back and records a `done` event; every member's
// stream then waits for that event.  Nothing synchronises with the device: the host threads only meet at enqueue time.
// What it buys: the row-sharded targets of a rank trained one after another on the thread that owns the
// communicator -- the in-rank concurrency (22 % at N = 1) was gone exactly where every target is row-sharded, and a job paid 8
// collectives per target-iteration.  Now they overlap like the target-sharded ones and a rank issues 8 collectives per iteration of
// ALL its row-sharded targets.  The members' order inside the fused buffer is their index, so it is the same on every rank.
// A member that fails breaks the group: the others raise at their next collective (and the RCCL communicator is aborted, as before).
// ---------------------------------------------------------------------------------------------
struct Fusion {
    static constexpr int MAX_MEMBERS = 64;
    Comm comm;                  // the rank's communicator (moved in by rgbm_fusion_create, moved back by rgbm_fusion_free)
    int device = 0;
    int n = 0;                  // members still taking part
    int n_members = 0;          // members the group was created for (rgbm_fusion_join checks the index against it)
    bool joined[MAX_MEMBERS] = {};
    std::mutex mu; std::condition_variable cv;
    int arrived = 0; long long gen = 0; bool broken = false; std::string why;
    struct Part { void* buf = nullptr; size_t count = 0; int type = 0; hipStream_t s = nullptr; hipEvent_t ready = nullptr; bool in = false; } part[MAX_MEMBERS];
    void* stage = nullptr; size_t stage_bytes = 0;
    hipStream_t cs = nullptr; hipEvent_t done = nullptr;
    long long collectives = 0, fused_parts = 0;    // statistics
};

void all_reduce_on(Comm& c, void* buf, size_t count, int type, hipStream_t s);

// the collective of one step for every member that is parked in it (called with f->mu held, by the last member to arrive -- or by a member
// that leaves while all the remaining ones are parked).  Members may be at different points of their sequences (one is counting codes for
// its next target while another is in the middle of an iteration): the parts are grouped by element type, one all-reduce per type present,
// in a fixed type order -- the same on every rank, because which member takes part in the k-th step only depends on the sequences.
void fusion_run_step_locked(Fusion* f) {
    try {
        for (int type = 0; type < 3; ++type) {
            const size_t esz = type == AR_I64 ? 8 : 4;
            size_t total = 0; int nparts = 0;
            for (int j = 0; j < Fusion::MAX_MEMBERS; ++j) if (f->part[j].in && f->part[j].type == type) { total += f->part[j].count; ++nparts; }
            if (nparts == 0) continue;
            if (total * esz > f->stage_bytes) {
                HIPCHK(hipStreamSynchronize(f->cs));
                if (f->stage) (void)hipFree(f->stage);
                f->stage_bytes = std::max<size_t>(total * esz * 2, 1 << 20);
                HIPCHK(hipMalloc(&f->stage, f->stage_bytes));
            }
            size_t off = 0;
            for (int j = 0; j < Fusion::MAX_MEMBERS; ++j) {
                Fusion::Part& q = f->part[j];
                if (!q.in || q.type != type) continue;
                HIPCHK(hipStreamWaitEvent(f->cs, q.ready, 0));
                if (q.count) HIPCHK(hipMemcpyAsync(static_cast<char*>(f->stage) + off * esz, q.buf, q.count * esz, hipMemcpyDeviceToDevice, f->cs));
                off += q.count;
            }
            all_reduce_on(f->comm, f->stage, total, type, f->cs);
            off = 0;
            for (int j = 0; j < Fusion::MAX_MEMBERS; ++j) {
                Fusion::Part& q = f->part[j];
                if (!q.in || q.type != type) continue;
                if (q.count) HIPCHK(hipMemcpyAsync(q.buf, static_cast<char*>(f->stage) + off * esz, q.count * esz, hipMemcpyDeviceToDevice, f->cs));
                off += q.count; f->fused_parts += 1;
            }
            f->collectives += 1;
        }
        for (int j = 0; j < Fusion::MAX_MEMBERS; ++j) f->part[j].in = false;
        HIPCHK(hipEventRecord(f->done, f->cs));
    } catch (const std::exception& e) { f->broken = true; f->why = e.what(); }
    f->arrived = 0; ++f->gen; f->cv.notify_all();
}

// one fused collective; called by every member with its own buffer
void fused_all_reduce(Fusion* f, int me, void* buf, size_t count, int type, hipStream_t s) {
    Fusion::Part& mine = f->part[me];
    if (!mine.ready) HIPCHK(hipEventCreateWithFlags(&mine.ready, hipEventDisableTiming));
    HIPCHK(hipEventRecord(mine.ready, s));
    std::unique_lock<std::mutex> lk(f->mu);
    if (f->broken) throw std::runtime_error("fusion group: a concurrent row-sharded training call of this rank failed (" + f->why + ")");
    mine.buf = buf; mine.count = count; mine.type = type; mine.s = s; mine.in = true;
    const long long gen = f->gen;
    if (++f->arrived >= f->n) fusion_run_step_locked(f);
    else {
        // bounded: a member of this rank that never arrives and never leaves -- its thread died before its next collective -- must not
        // park the others (and, through them, the peer ranks) for ever: after RGBM_COMM_TIMEOUT_S the group is broken and everybody raises
        static const double limit = [] { const char* e = getenv("RGBM_COMM_TIMEOUT_S"); double v = e ? atof(e) : 600.0; return v > 0.0 ? v : 600.0; }();
        if (!f->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return f->gen != gen || f->broken; })) {
            f->broken = true; f->why = "a member of this rank did not reach its next collective within " + std::to_string((int)limit) + " s";
            if (f->comm.kind == 1 && f->comm.nccl) { (void)ncclCommAbort(f->comm.nccl); f->comm.nccl = nullptr; f->comm.kind = 0; }
            f->cv.notify_all();
        }
    }
    if (f->broken) throw std::runtime_error("fusion group: a concurrent row-sharded training call of this rank failed (" + f->why + ")");
    HIPCHK(hipStreamWaitEvent(s, f->done, 0));
}

// a member is done (its training call returned or failed): the others go on among themselves
void fusion_leave(bool failed, const char* why) {
    Comm& c = g_comm;
    if (c.kind != 3 || !c.fusion) return;
    Fusion* f = c.fusion;
    {
        std::lock_guard<std::mutex> lk(f->mu);
        f->part[c.member].in = false;
        if (failed && !f->broken) { f->broken = true; f->why = why ? why : "unknown error"; if (f->comm.kind == 1 && f->comm.nccl) { (void)ncclCommAbort(f->comm.nccl); f->comm.nccl = nullptr; f->comm.kind = 0; } }
        f->n -= 1;
        if (f->broken) f->cv.notify_all();
        else if (f->n > 0 && f->arrived >= f->n) fusion_run_step_locked(f);     // the remaining members are all parked in their next step: it is theirs alone
    }
    g_comm = Comm();
}

// Fail-safe for the RCCL transport.  A rank that dies (or throws) in the middle of a training call never enqueues its next
// all-reduce, and its peers would sit in theirs for ever.  So (a) a rank that fails ABORTS its communicator before the error
// leaves the library (comm_abort_on_failure), and (b) every rank waits for its training stream through a watchdog instead of
// a blocking synchronise: it polls the stream, asks RCCL for asynchronous errors and gives up after RGBM_COMM_TIMEOUT_S seconds
// (default 600) without progress -- then it aborts its own communicator (which takes the stuck collective kernel off the stream)
// and raises.  After an abort the thread has no communicator: the caller falls back to target sharding (repair/dist.py).
void comm_abort() {
    Comm& c = g_comm;
    if (c.kind == 1 && c.nccl) { (void)ncclCommAbort(c.nccl); c.nccl = nullptr; c.kind = 0; c.nranks = 1; c.rank = 0; }
    if (c.kind == 3 && c.fusion) fusion_leave(true, "a member's stream failed or timed out inside a collective");
}

void stream_sync_watchdog(hipStream_t s) {
    Comm& c = g_comm;
    const bool fused_rccl = c.kind == 3 && c.fusion && c.fusion->comm.kind == 1;
    if (c.kind != 1 && !fused_rccl) { HIPCHK(hipStreamSynchronize(s)); return; }
    static const double limit = [] { const char* e = getenv("RGBM_COMM_TIMEOUT_S"); double v = e ? atof(e) : 600.0; return v > 0.0 ? v : 600.0; }();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); comm_abort(); throw std::runtime_error(std::string("training stream failed during a collective: ") + hipGetErrorString(q)); }
        ncclResult_t ae = ncclSuccess; bool have_ae = false;
        if (fused_rccl) {
            // the group's state belongs to its mutex: another member may be leaving right now -- writing `why`, aborting the communicator
            // and clearing the handle -- so the flag, the text and the asynchronous-error query are taken under the lock
            bool broken; std::string w;
            {
                std::lock_guard<std::mutex> lk(c.fusion->mu);
                broken = c.fusion->broken; if (broken) w = c.fusion->why;
                if (!broken && c.fusion->comm.nccl) have_ae = ncclCommGetAsyncError(c.fusion->comm.nccl, &ae) == ncclSuccess;
            }
            if (broken) { comm_abort(); throw std::runtime_error("fusion group broken: " + w); }
        } else if (c.nccl) have_ae = ncclCommGetAsyncError(c.nccl, &ae) == ncclSuccess;
        if (have_ae && ae != ncclSuccess && ae != ncclInProgress) {
            const std::string what = ncclGetErrorString(ae);
            comm_abort();
            throw std::runtime_error("RCCL reported an asynchronous error (a peer rank failed?): " + what);
        }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > limit) {
            comm_abort();
            throw std::runtime_error("row-sharded training: no progress for " + std::to_string((int)limit) + " s inside a collective (a peer rank is gone); communicator aborted");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(waited < 0.05 ? 50 : 1000));
    }
}

struct PtrList { const void* p[16]; };
template <typename T>
__global__ void k_sum_ranks(PtrList src, int n, T* __restrict__ out, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T acc = 0;
    for (int r = 0; r < n; ++r) acc += reinterpret_cast<const T*>(src.p[r])[i];
    out[i] = acc;
}

// in-place sum all-reduce of `count` elements on stream s
void all_reduce(void* buf, size_t count, int type, hipStream_t s) {
    Comm& c = g_comm;
    if (c.kind == 3) { fused_all_reduce(c.fusion, c.member, buf, count, type, s); return; }     // (also with count == 0: every member takes part in every step)
    all_reduce_on(c, buf, count, type, s);
}
void all_reduce_on(Comm& c, void* buf, size_t count, int type, hipStream_t s) {
    if (c.kind == 0 || count == 0) return;
    const size_t esz = type == AR_I64 ? 8 : 4;
    if (c.kind == 1) {
        const ncclDataType_t dt = type == AR_I64 ? ncclInt64 : (type == AR_U32 ? ncclUint32 : ncclInt32);
        ncclResult_t r = ncclAllReduce(buf, buf, count, dt, ncclSum, c.nccl, s);
        if (r != ncclSuccess) throw std::runtime_error(std::string("ncclAllReduce failed: ") + ncclGetErrorString(r));
        return;
    }
    LocalGroup* g = c.lg;
    if (c.tmp_bytes < count * esz) { if (c.tmp) (void)hipFree(c.tmp); HIPCHK(hipMalloc(&c.tmp, count * esz)); c.tmp_bytes = count * esz; }
    HIPCHK(hipStreamSynchronize(s));
    { std::lock_guard<std::mutex> lk(g->mu); g->ptr[c.rank] = buf; }
    g->barrier();
    PtrList pl; for (int r = 0; r < g->nranks; ++r) pl.p[r] = g->ptr[r];
    const unsigned blocks = (unsigned)((count + 255) / 256);
    if (type == AR_I64) hipLaunchKernelGGL(k_sum_ranks<long long>, dim3(blocks), dim3(256), 0, s, pl, g->nranks, (long long*)c.tmp, count);
    else if (type == AR_U32) hipLaunchKernelGGL(k_sum_ranks<unsigned int>, dim3(blocks), dim3(256), 0, s, pl, g->nranks, (unsigned int*)c.tmp, count);
    else hipLaunchKernelGGL(k_sum_ranks<int>, dim3(blocks), dim3(256), 0, s, pl, g->nranks, (int*)c.tmp, count);
    HIPCHK(hipStreamSynchronize(s));
    g->barrier();                                   // everyone has read every buffer
    HIPCHK(hipMemcpyAsync(buf, c.tmp, count * esz, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    g->barrier();
}

// all-gather of equal-sized device blocks over the calling thread's own communicator (not from inside a fusion group): recv = [nranks][bytes].
// RCCL: ncclAllGather on the library's communicator, enqueued on s; thread group: every rank copies every rank's block (device copies
// between two barriers); no communicator: a copy.  What the C1 / C2 exchanges of a multi-GPU job ride on (model blobs, repaired cells).
struct GatherStats { std::atomic<long long> bytes{0}, ns{0}, calls{0}; };
GatherStats& gather_stats() { static GatherStats g; return g; }
void all_gather_on(Comm& c, const void* send, void* recv, size_t bytes, hipStream_t s) {
    if (c.kind == 3) throw std::invalid_argument("all-gather from inside a fusion group");
    if (bytes == 0) return;
    if (c.kind == 0 || c.nranks <= 1) { if (c.kind != 1) { HIPCHK(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s)); return; } }
    if (c.kind == 1) {
        ncclResult_t r = ncclAllGather(send, recv, bytes, ncclUint8, c.nccl, s);
        if (r != ncclSuccess) throw std::runtime_error(std::string("ncclAllGather failed: ") + ncclGetErrorString(r));
        return;
    }
    LocalGroup* g = c.lg;
    HIPCHK(hipStreamSynchronize(s));
    { std::lock_guard<std::mutex> lk(g->mu); g->ptr[c.rank] = const_cast<void*>(send); }
    g->barrier();
    for (int r = 0; r < g->nranks; ++r) HIPCHK(hipMemcpyAsync(static_cast<char*>(recv) + (size_t)r * bytes, g->ptr[r], bytes, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    g->barrier();                                   // everyone has read every block
}

struct Feat {
    int32_t n_codes = 0, V = 0, has_nan = 0; std::vector<int32_t> ub;
    // CATEGORICAL features only (else empty): bit c set => no training row held code c.  Such a category is MISSING for this model at
    // prediction time (it has no place in the order of the codes) -- what a per-model dictionary does with a value it does not hold.
    std::vector<uint32_t> unseen;
    bool is_unseen(int32_t c) const { return !unseen.empty() && ((unseen[(size_t)c >> 5] >> (c & 31)) & 1u); }
};
// One tree: L leaves, L - 1 internal nodes.  Its eight arrays sit in ONE block -- the tree's own, or a slice of the model's arena
// (a model is 300 x K trees: eight std::vectors a tree were 576 000 allocations for the 16 models of a default job).
struct Tree {
    int32_t L = 1;
    double *gain = nullptr, *leaf_value = nullptr;
    int32_t *feat = nullptr, *theta = nullptr, *dleft = nullptr, *left = nullptr, *right = nullptr, *leaf_count = nullptr;
    std::unique_ptr<double[]> own;
    static size_t doubles(int L) { const size_t n = (size_t)L - 1; return n + (size_t)L + (5 * n + (size_t)L + 1) / 2; }
    void place(double* p, int L_) {
        L = L_; const size_t n = (size_t)L - 1;
        gain = p; leaf_value = p + n;
        int32_t* q = reinterpret_cast<int32_t*>(p + n + L);
        feat = q; theta = q + n; dleft = q + 2 * n; left = q + 3 * n; right = q + 4 * n; leaf_count = q + 5 * n;
    }
    void alloc(int L_) { own.reset(new double[doubles(L_)]); place(own.get(), L_); }
};

struct DeviceModel {   // predictor mirror of a model on one device
    DevBuf<rg::PNode> nodes; DevBuf<double> leaf_value;
    // bit-vector scoring tables (k_predict_qs): per tree the AND-masks of every (feature, bin) and the leaf values in left-to-right order
    DevBuf<uint32_t> qs_masks, qs_used; DevBuf<double> qs_leaves; DevBuf<int32_t> qs_foff; int qs_S = 0, qs_MW = 0;   // qs_MW = 0: not built
    DevBuf<uint8_t> lut; DevBuf<long long> lut_off; DevBuf<int32_t> n_codes; DevBuf<uint8_t> miss; DevBuf<int32_t> ident;
    int node_stride = 1, leaf_stride = 1;
};

}  // namespace

struct rgbm_model {
    int32_t objective = 0, num_class = 0, K = 1, n_iter = 0, F = 0;
    std::vector<Feat> feats;
    std::vector<Tree> trees;
    std::unique_ptr<double[]> tree_arena;      // storage of the trees of a trained model (a loaded model's trees own theirs)
    std::mutex mu;
    std::map<int, DeviceModel*> dev;
    ~rgbm_model() { for (auto& kv : dev) { (void)hipSetDevice(kv.first); delete kv.second; } }
};


namespace {

// ---------------------------------------------------------------------------------------------
// Host logic: bin finding (LightGBM bin.cpp GreedyFindBin / FindBinWithZeroAsOneBin restated in
// code space; see DESIGN.md "Binning").  Input: counts per code over the training rows.
// ---------------------------------------------------------------------------------------------
// Bin upper bound between two neighbouring training codes a < b.  Codes of a NUMERIC column are ranks of its distinct values and
// LightGBM puts the bound at the midpoint of the VALUES: with the column's value dictionary (rgbm_table_set_column_values) the bound
// in code space is the largest code whose value is <= (val[a] + val[b]) / 2, so a value that only rows outside the training set
// hold falls on the side LightGBM would send it to.  Without one (categorical columns, host-array calls) the codes are the scale.
int32_t mid_code(int32_t a, int32_t b, const double* vals) {
    if (!vals) return (int32_t)(((int64_t)a + (int64_t)b) / 2);
    const double m = (vals[a] + vals[b]) / 2.0;
    int32_t lo = a, hi = b - 1;
    while (lo < hi) { const int32_t c = lo + (hi - lo + 1) / 2; if (vals[c] <= m) lo = c; else hi = c - 1; }
    return lo;
}

int greedy_find_bin(const std::vector<int32_t>& dv, const std::vector<int64_t>& cnt, int max_bin, int64_t total_cnt,
                    int min_data_in_bin, std::vector<int32_t>& ub, const double* vals) {
    const int nd = (int)dv.size();
    ub.clear();
    if (nd <= 0) return 0;
    if (nd <= max_bin) {
        int64_t cur = 0;
        for (int i = 0; i + 1 < nd; ++i) {
            cur += cnt[i];
            if (cur >= min_data_in_bin) { ub.push_back(mid_code(dv[i], dv[i + 1], vals)); cur = 0; }
        }
        ub.push_back(INT32_MAX);
        return (int)ub.size();
    }
    if (min_data_in_bin > 0) {
        max_bin = (int)std::min<int64_t>(max_bin, total_cnt / min_data_in_bin);
        max_bin = std::max(max_bin, 1);
    }
    double mean_bin_size = (double)total_cnt / max_bin;
    int rest_bin_cnt = max_bin;
    int64_t rest_sample_cnt = total_cnt;
    std::vector<char> big(nd, 0);
    for (int i = 0; i < nd; ++i)
        if ((double)cnt[i] >= mean_bin_size) { big[i] = 1; --rest_bin_cnt; rest_sample_cnt -= cnt[i]; }
    mean_bin_size = (double)rest_sample_cnt / rest_bin_cnt;
    std::vector<int32_t> upper(max_bin, INT32_MAX), lower(max_bin, INT32_MAX);
    int bin_cnt = 0;
    lower[0] = dv[0];
    int64_t cur = 0;
    for (int i = 0; i + 1 < nd; ++i) {
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
    for (int i = 0; i + 1 < bin_cnt; ++i) {
        int32_t v = mid_code(upper[i], lower[i + 1], vals);
        if (ub.empty() || ub.back() != v) ub.push_back(v);
    }
    ub.push_back(INT32_MAX);
    return (int)ub.size();
}

void find_bin(const unsigned int* cnt, int32_t n_codes, int64_t n_train, const rgbm_params& p, Feat& f, const double* vals) {
    std::vector<int32_t> dv; std::vector<int64_t> dc;
    int64_t seen = 0;
    for (int32_t c = 0; c < n_codes; ++c) if (cnt[c]) { dv.push_back(c); dc.push_back(cnt[c]); seen += cnt[c]; }
    const int64_t na = n_train - seen;
    int mb = p.max_bin - (na > 0 ? 1 : 0) - 1;   // NaN bin, zero bin (FindBinWithZeroAsOneBin)
    if (mb < 1) mb = 1;
    f.n_codes = n_codes; f.has_nan = na > 0 ? 1 : 0;
    f.V = greedy_find_bin(dv, dc, mb, seen, p.min_data_in_bin, f.ub, vals);
}


// LightGBM utils/random.h (as remembered)
// Test / experiment switches, read from the environment at the start of every training call (tests flip them between calls); none
// is needed in normal use.  RGBM_GROWER=leafwise|level, RGBM_TIMING=1 (host wall-clock of the phases to stderr), RGBM_LV_LDS=bytes
// (shrinks the LDS pool of the level passes: more built-slot windows per level), RGBM_LV_BLOCKS / RGBM_MT_BLOCKS (row blocks per class
// tree of the root / level passes), RGBM_MT_TREES (cap on the class trees per level-pass workgroup), RGBM_MT_REP (LDS replication the
// level passes are sized for, default 8), RGBM_JOINT_ROOT=0, RGBM_MT_ACC2=0 (tables of 17..32 features: one level pass per 16-feature chunk
// instead of one pass that accumulates both), RGBM_MT_SPEC=0|1 (wave-specialised level pass), RGBM_MT_SPARSE=0 (no sparse sweep: class trees
// with few live rows are walked tile by tile like the others).  Round 5: RGBM_MT_ROT=-1|0|1 (feature rotation of a level pass's histogram updates:
// per launch where fewer than RGBM_MT_ROT_COPIES2 / 2 copies of its histograms fit the LDS -- the default --, never, every launch that has the
// instantiation; RGBM_MT_ROT_T=0 keeps the replicated layout's class trees per workgroup), RGBM_MT_LOCK=<rounds> (lock-step of the class-tree groups
// of a row block, wave-specialised pass), RGBM_JOINT_WIDE=1 (16-bit joint codes in the root pass), RGBM_FX_MEASURE=separate (the coarse gradient sums of
// numerics v2.2 from a pass of their own instead of out of the gradient kernels), round 6: RGBM_DEFER_SCORE=1 (AddScore inside the next iteration's gradient kernel; slower, off), RGBM_TEST_HOOKS=1 + RGBM_FX_ROWS=R (TEST hook: the fixed-point grid of a
// table of R rows; the oracle reads the same pair; a warning goes to stderr when it is active).
constexpr int LV_THREADS_DEFAULT = 1024;
struct RunSwitches { int grower = 0; int lv_lds = 0; long long lv_blocks = 0; long long mt_blocks = 0; int mt_T = 0; int mt_rep = MT_ROT ? 4 : 8 /* replicas the level passes are sized for: under feature rotation four resolve every conflict */; bool joint_root = true; bool timing = false; bool mt_acc2 = true; int mt_threads = LV_THREADS_DEFAULT; int mt_spec = -1 /* -1: wave-specialised pass for two-chunk tables only (measured) */; bool mt_sparse = true /* class trees with < 1/MT_SPARSE_DIV (= 1/16) live rows are swept through their node ids */; int mt_rot = -1 /* RGBM_MT_ROT: feature rotation of the level pass atomics: -1 = where the LDS holds fewer than mt_rot_copies2 / 2 = EIGHT copies of the launch's worst-case histograms (default), 0 = never, 1 = every pass that has the instantiation */; bool joint_wide = false /* RGBM_JOINT_WIDE=1: 16-bit joint codes in the root pass (groups of <= 1024 joint bins, 10 atomics per row instead of 14 on the synthetic table): measured neutral (the two copies that fit conflict more), off */; int mt_rot_tmin = 2 /* RGBM_MT_ROT_TMIN: a launch whose replicated layout holds fewer class trees per workgroup than this (2: a single one) rotates when one copy holds at least twice as many (0 = off).  Measured (profiles/EXPERIMENTS.md, r7m / r7n): 0 / 2 / 4: bench step 80.4-80.8 / 80.0-80.1 / 80.8-81.2 ms, roofline frac 0.196 / 0.200 / 0.200 */; int mt_rot_tdiv = 1 /* RGBM_MT_ROT_TDIV=d: a rotated launch takes 1/d of the class trees one copy would allow and replicates with the rest (measured: see profiles/EXPERIMENTS.md) */; bool mt_rot_T = true /* RGBM_MT_ROT_T=0: rotated launches keep the class trees per workgroup of the replicated layout */; int mt_rot_copies2 = 16 /* RGBM_MT_ROT_COPIES2: twice the number of plain copies of a launch's histograms below which it rotates (16 = eight copies: every launch that cannot have the eight copies the replicated layout is sized for; round 5 had 6 -- the flat pipeline of round 6 moved the balance, profiles/r6r_*) */; int mt_lock = -1 /* RGBM_MT_LOCK=<tile rounds>: lock-step window of the class-tree groups of a row block in the wave-specialised pass; -1 / 0 = off (default) */; bool defer_score = false /* RGBM_DEFER_SCORE=1: the level grower adds a tree's output to the scores inside the NEXT iteration's gradient kernel (PendingScore; k_level_last finishes routing + counts on the node ids alone) instead of in a pass of its own (k_level_final) -- same scores, same models; the tests run both.  Measured SLOWER (round 6, profiles/r7b_*: k_grad_mc 2.08 -> 3.91 ms per launch for 0.81 ms saved in the last pass; bench step 80.8 / 81.1 -> 83.1 / 83.3 ms): the gradient kernel is FP64-bound at 3.6 TB/s, the pass of its own streams at 4.7 TB/s -- the third fusion of these two that lost; off */; bool fx_separate = false /* RGBM_FX_MEASURE=separate: the coarse sums behind every class tree's fixed-point grid come from a pass of their own over the (g, h) array (k_fx_measure) instead of out of the gradient kernels -- same sums, same models; the tests run both */; };
RunSwitches read_switches() {
    RunSwitches w;
    if (const char* e = getenv("RGBM_GROWER")) w.grower = strcmp(e, "leafwise") == 0 ? 2 : (strcmp(e, "level") == 0 ? 1 : 0);
    if (const char* e = getenv("RGBM_LV_LDS")) w.lv_lds = atoi(e);
    if (const char* e = getenv("RGBM_LV_BLOCKS")) w.lv_blocks = atoll(e);
    if (const char* e = getenv("RGBM_MT_BLOCKS")) w.mt_blocks = atoll(e);
    if (const char* e = getenv("RGBM_MT_TREES")) w.mt_T = atoi(e);
    if (const char* e = getenv("RGBM_MT_REP")) w.mt_rep = atoi(e);
    if (const char* e = getenv("RGBM_JOINT_ROOT")) w.joint_root = atoi(e) != 0;
    if (const char* e = getenv("RGBM_MT_ACC2")) w.mt_acc2 = atoi(e) != 0;
    if (const char* e = getenv("RGBM_MT_THREADS")) w.mt_threads = atoi(e) == 768 ? 768 : 1024;
    if (const char* e = getenv("RGBM_MT_SPEC")) w.mt_spec = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("RGBM_MT_SPARSE")) w.mt_sparse = atoi(e) != 0;
    if (const char* e = getenv("RGBM_FX_MEASURE")) w.fx_separate = strcmp(e, "separate") == 0;
    if (const char* e = getenv("RGBM_DEFER_SCORE")) w.defer_score = atoi(e) != 0;
    if (const char* e = getenv("RGBM_MT_LOCK")) w.mt_lock = atoi(e);
    if (const char* e = getenv("RGBM_MT_ROT")) w.mt_rot = atoi(e);
    if (const char* e = getenv("RGBM_MT_ROT_COPIES2")) w.mt_rot_copies2 = atoi(e);
    if (const char* e = getenv("RGBM_MT_ROT_TDIV")) w.mt_rot_tdiv = std::max(1, atoi(e));
    if (const char* e = getenv("RGBM_MT_ROT_TMIN")) w.mt_rot_tmin = atoi(e);
    if (const char* e = getenv("RGBM_MT_ROT_T")) w.mt_rot_T = atoi(e) != 0;
    if (const char* e = getenv("RGBM_JOINT_WIDE")) w.joint_wide = atoi(e) != 0;
    w.timing = getenv("RGBM_TIMING") != nullptr;
    return w;
}

struct LgbRand {
    uint32_t x;
    explicit LgbRand(uint32_t seed) : x(seed) {}
    int rnd16() { x = 214013u * x + 2531011u; return (int)((x >> 16) & 0x7FFF); }
    int rnd32() { x = 214013u * x + 2531011u; return (int)(x & 0x7FFFFFFF); }
    float next_float() { return (float)rnd16() / 32768.0f; }
    int next_int(int lo, int hi) { return rnd32() % (hi - lo) + lo; }
    std::vector<int> sample(int N, int K) {
        std::vector<int> ret;
        if (K > N || K <= 0) return ret;
        if (K == N) { for (int i = 0; i < N; ++i) ret.push_back(i); return ret; }
        if (K > 1 && (double)K > ((double)N / std::log2((double)K))) {
            for (int i = 0; i < N; ++i) {
                double prob = (double)(K - (int)ret.size()) / (double)(N - i);
                if (next_float() < prob) ret.push_back(i);
            }
            return ret;
        }
        std::vector<char> in(N, 0);
        for (int r = N - K; r < N; ++r) { int v = next_int(0, r); if (in[v]) in[r] = 1; else in[v] = 1; }
        for (int i = 0; i < N; ++i) if (in[i]) ret.push_back(i);
        return ret;
    }
};


void check_params(const rgbm_params& p) {
    if (p.objective < 0 || p.objective > 2) throw std::invalid_argument("objective must be 0 (binary), 1 (multiclass) or 2 (regression)");
    if (p.max_bin < 2 || p.max_bin > 255) throw std::invalid_argument("max_bin must be in [2, 255]");
    if (p.num_leaves < 2 || p.num_leaves > 32767) throw std::invalid_argument("num_leaves must be in [2, 32767]");
    if (p.n_estimators < 1) throw std::invalid_argument("n_estimators must be positive");
    if (!(p.learning_rate > 0.0)) throw std::invalid_argument("learning_rate must be positive");
    // regression: the fixed-point grid of the histogram sums is sized for |score - y| <= (max y - min y), which a shrinkage above 1 can break
    if (p.objective == 2 && p.learning_rate > 1.0) throw std::invalid_argument("regression with learning_rate > 1 is not supported");
    if (p.objective == 1 && p.num_class < 2) throw std::invalid_argument("multiclass needs num_class >= 2");
    if (p.min_data_in_leaf < 0 || p.lambda_l1 < 0 || p.lambda_l2 < 0) throw std::invalid_argument("negative regularisation / min_data_in_leaf");
    if (p.bagging_fraction <= 0.0 || p.bagging_fraction > 1.0) throw std::invalid_argument("bagging_fraction must be in (0, 1]");
    if (p.feature_fraction <= 0.0 || p.feature_fraction > 1.0) throw std::invalid_argument("feature_fraction must be in (0, 1]");
}

struct HostLabelStats {   // row-order weight sums, only used with per-row sample weights
    bool valid = false;
    std::vector<double> tot; double suml = 0.0, w_max = 0.0;
};

// ---------------------------------------------------------------------------------------------
// What a fit derives from the code counts of its training rows before it touches row data (GBDT::Init / Dataset construction /
// BoostFromScore): bins, lookup tables, chunk layout, label statistics, initial scores, the fixed-point grid.  Shared by the
// single-fit trainer (train_core) and the batched small-table trainer (train_batch_small).
// ---------------------------------------------------------------------------------------------
struct FitHost {
    int64_t n_train = 0; int nchunk = 1; size_t lds_hist = 0;
    std::vector<int32_t> cols, ncod; std::vector<long long> cnt_off, lut_off; std::vector<unsigned int> cnt;
    std::vector<rg::FeatMeta> fmeta; std::vector<rg::ChunkMeta> cmeta; std::vector<uint8_t> lut, miss, trivial;
    std::vector<double> init, yv32; rg::TrainConst tc;
    std::unique_ptr<rgbm_model> model;
};

void fit_setup(const rgbm_table& tab, const double* y_value_in, const double* class_weight, const HostLabelStats* hls, const rgbm_params& p, int32_t F, FitHost& h) {
    using namespace rg;
    const int64_t N = tab.n;
    const int obj = p.objective;
    const std::vector<int32_t>& cols = h.cols; const std::vector<int32_t>& ncod = h.ncod; const std::vector<long long>& cnt_off = h.cnt_off;
    const std::vector<unsigned int>& cnt = h.cnt;
    const int n_y = ncod[F];
    const int K = obj == 1 ? p.num_class : 1;
    const double* y_value = nullptr;
    if (obj == 2) { h.yv32.resize(std::max(n_y, 1)); for (int c = 0; c < n_y; ++c) h.yv32[c] = (double)(float)y_value_in[c]; y_value = h.yv32.data(); }
    const unsigned int* ycnt = cnt.data() + cnt_off[F];
    int64_t& n_train = h.n_train; n_train = 0;
    for (int c = 0; c < n_y; ++c) n_train += ycnt[c];
    if (n_train <= 0) throw std::out_of_range("no training rows (every target cell is NULL)");
    if (N >= (1ll << 31) - 4096) throw std::invalid_argument("more than 2^31 rows per table are not supported");

    // ---- 2. bins
    h.model.reset(new rgbm_model());
    rgbm_model* model = h.model.get();
    model->objective = obj; model->num_class = obj == 1 ? K : (obj == 0 ? 2 : 1); model->K = K; model->F = F;
    model->feats.resize(F);
    std::vector<FeatMeta>& fmeta = h.fmeta; fmeta.assign(F, FeatMeta());
    std::vector<long long>& lut_off = h.lut_off; lut_off.assign(F + 1, 0);
    std::vector<uint8_t>& miss = h.miss; std::vector<uint8_t>& trivial = h.trivial; miss.assign(F, 0); trivial.assign(F, 0);
    int totbins = 0;
    for (int f = 0; f < F; ++f) {
        Feat& ft = model->feats[f];
        const std::vector<double>* cv = (size_t)cols[f] < tab.col_values.size() ? &tab.col_values[cols[f]] : nullptr;
        find_bin(cnt.data() + cnt_off[f], ncod[f], n_train, p, ft, (cv && (int32_t)cv->size() == ncod[f] && ncod[f] > 0) ? cv->data() : nullptr);
        if ((size_t)cols[f] < tab.col_kind.size() && tab.col_kind[cols[f]] == 1) {
            const unsigned int* cf = cnt.data() + cnt_off[f];
            bool any = false;
            for (int32_t c = 0; c < ncod[f]; ++c) any |= cf[c] == 0;
            if (any) {
                ft.unseen.assign((size_t)(ncod[f] + 31) / 32, 0u);
                for (int32_t c = 0; c < ncod[f]; ++c) if (cf[c] == 0) ft.unseen[(size_t)c >> 5] |= 1u << (c & 31);
            }
        }
        fmeta[f].V = ft.V; fmeta[f].has_nan = ft.has_nan; fmeta[f].nbins = std::max(ft.V + ft.has_nan, 1); fmeta[f].hoff = totbins;
        totbins += fmeta[f].nbins;
        trivial[f] = (ft.V + ft.has_nan <= 1) || ft.V == 0;
        miss[f] = (uint8_t)(ft.has_nan ? ft.V : 0);
        lut_off[f + 1] = lut_off[f] + std::max(ncod[f], 1);
    }
    std::vector<uint8_t>& lut = h.lut; lut.assign(lut_off[F], 0);
    for (int f = 0; f < F; ++f) {
        const Feat& ft = model->feats[f];
        int b = 0;
        for (int c = 0; c < ncod[f]; ++c) { while (b < ft.V - 1 && c > ft.ub[b]) ++b; lut[lut_off[f] + c] = (uint8_t)(ft.V > 0 ? b : 0); }
    }
    const int nchunk = (F + 15) / 16; h.nchunk = nchunk;
    std::vector<ChunkMeta>& cmeta = h.cmeta; cmeta.assign(nchunk, ChunkMeta());
    size_t& lds_hist = h.lds_hist; lds_hist = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
        ChunkMeta& cm = cmeta[ch]; cm.first_feat = ch * 16; cm.nfeat = std::min(16, F - ch * 16); cm.fast_slots = 0; cm.wide_bins = 0;
        for (int j = 0; j < cm.nfeat; ++j) {
            FeatMeta& m = fmeta[ch * 16 + j];
            int sh = 0; while (sh < 5 && (m.nbins << (sh + 1)) <= 256) ++sh;
            m.rep_shift = sh; m.fast_base = cm.fast_slots; m.wide_off = cm.wide_bins;
            cm.fast_slots += m.nbins << sh; cm.wide_bins += m.nbins;
        }
        lds_hist = std::max(lds_hist, (size_t)cm.fast_slots * 16);      // leaf-wise k_hist: replicated (g, h) slots, 16 B each
    }

    // ---- 3. label statistics, init scores (BoostFromScore), quantisation scales
    const int nl = std::max({n_y, obj == 0 ? 2 : 1, obj == 1 ? K : 1});
    std::vector<double> tot(nl, 0.0);
    double w_max = 0.0;
    if (hls && hls->valid) { tot = hls->tot; tot.resize(nl, 0.0); w_max = hls->w_max; }
    else {
        for (int c = 0; c < n_y; ++c) {
            const double cw32 = class_weight ? (double)(float)class_weight[c] : 1.0;   // float32 weights (LightGBM label_t)
            tot[c] = (double)ycnt[c] * cw32;
            if (ycnt[c]) w_max = std::max(w_max, cw32);
        }
    }
    if (!(w_max > 0.0)) w_max = 1.0;
    double sumw = 0.0;
    for (int c = 0; c < nl; ++c) sumw += tot[c];
    std::vector<double>& init = h.init; init.assign(K, 0.0);
    double ymin = 0.0, ymax = 0.0;
    const double keps = k_eps();
    if (obj == 2) {
        bool first = true; double suml = 0.0;
        for (int c = 0; c < n_y; ++c) {
            if (!ycnt[c]) continue;
            if (first) { ymin = ymax = y_value[c]; first = false; }
            ymin = std::min(ymin, y_value[c]); ymax = std::max(ymax, y_value[c]);
        }
        if (hls && hls->valid) suml = hls->suml; else for (int c = 0; c < nl; ++c) suml += tot[c] * (c < n_y ? y_value[c] : 0.0);
        init[0] = suml / sumw;
    } else if (obj == 0) {
        double pavg = tot[1] / sumw;
        if (pavg > 1.0 - keps) pavg = 1.0 - keps;
        if (pavg < keps) pavg = keps;
        init[0] = std::log(pavg / (1.0 - pavg));
    } else {
        for (int k = 0; k < K; ++k) { double pr = tot[k] / sumw; init[k] = std::log(pr > keps ? pr : keps); }
    }
    const double factor = obj == 1 ? (double)K / (double)(K - 1) : 1.0;
    double bound_g, bound_h;
    if (obj == 2) { bound_g = (ymax - ymin) * w_max; if (!(bound_g > 0.0)) bound_g = 1.0; bound_h = w_max; }
    else if (obj == 0) { bound_g = w_max; bound_h = 0.25 * w_max; }
    else { bound_g = w_max; bound_h = factor * 0.25 * w_max; }
    // fixed-point grid of the histogram sums (numerics v2.2, rgbm_numerics.h).  The FLOOR of every class tree's exponent is v2.1's: |g_i| <= (bound_g / w_max) * w_i
    // and h_i <= (bound_h / w_max) * w_i hold for every row i, so with e = min(50 - ceil_log2(bound), 62 - ceil_log2(bound * sum_w / w_max)) every converted value is
    // at most 2^50 in magnitude (the range of the rint trick) and an int64 sum over all training rows (of all ranks) stays below 2^62.  The exponent a class tree
    // actually gets in an iteration comes from the measured coarse sums of its gradients (FxGrid -> k_fx_scale / k_small_tree).
    double wr = sumw / w_max;
    long long q_mult = 1;
    // TEST hook (tests/test_numerics_bound.py, tools/numerics_scale.py; the oracle reads the same pair): RGBM_TEST_HOOKS=1 + RGBM_FX_ROWS = R sizes the grid as if the
    // table held R training rows with this table's gradient distribution -- the grid a 10M / 100M-row table of this kind gets, on a table that trains in seconds.
    // It changes every model: only honoured with the explicit switch, and said on stderr.
    if (const char* th = std::getenv("RGBM_TEST_HOOKS")) if (std::atoi(th) != 0) if (const char* ev = std::getenv("RGBM_FX_ROWS")) {
        const double R = std::atof(ev);
        if (R > (double)n_train && R <= 4294967296.0 && n_train > 0) {
            wr *= R / (double)n_train; q_mult = ((long long)R + n_train - 1) / n_train;
            static std::atomic<bool> said{false};
            if (!said.exchange(true)) fprintf(stderr, "[rgbm] TEST HOOK active: RGBM_FX_ROWS=%s coarsens the fixed-point grid of every model trained by this process\n", ev);
        }
    }
    const int e_g = rg::fx_exponent(bound_g, wr), e_h = rg::fx_exponent(bound_h, wr);

    TrainConst& tc = h.tc; memset(&tc, 0, sizeof(tc));
    tc.sg = std::ldexp(1.0, e_g); tc.sh = std::ldexp(1.0, e_h); tc.inv_sg = std::ldexp(1.0, -e_g); tc.inv_sh = std::ldexp(1.0, -e_h);
    tc.l1 = p.lambda_l1; tc.l2 = p.lambda_l2; tc.min_gain_to_split = p.min_gain_to_split; tc.min_sum_hessian = p.min_sum_hessian_in_leaf;
    tc.learning_rate = p.learning_rate; tc.factor = factor; tc.min_data_in_leaf = p.min_data_in_leaf; tc.max_depth = p.max_depth;
    tc.num_leaves = p.num_leaves; tc.F = F; tc.K = K; tc.totbins = std::max(totbins, 1); tc.nchunk = nchunk; tc.objective = obj;
    tc.N = N; tc.n_train = n_train; tc.NG = N;
    tc.fx.c_g = 24 - rg::fx_ceil_log2(bound_g); tc.fx.c_h = 24 - rg::fx_ceil_log2(bound_h);
    tc.fx.e_g_min = e_g; tc.fx.e_h_min = e_h;
    tc.fx.e_g_max = 50 - rg::fx_ceil_log2(bound_g); tc.fx.e_h_max = 50 - rg::fx_ceil_log2(bound_h);
    tc.fx.q_mult = q_mult;
}

// the trees as the trainer leaves them in its flat device arrays (TreeOut), copied to the host: -> the model's tree list
struct HostTrees { const int32_t *L, *feat, *theta, *dleft, *left, *right, *cnt; const double *gain, *val; const int32_t* any; };
// two steps, so that the (many) trees of a batch of fits can be filled by several host threads: size the list, then fill ranges of it
size_t model_trees_begin(rgbm_model* model, const HostTrees& h, int NE, int K) {
    int n_iter = NE;
    for (int it = 0; it < NE; ++it) if (!h.any[it]) { n_iter = it > 0 ? it : 1; break; }   // "no more leaves that meet the split requirements"
    model->n_iter = n_iter;
    const size_t nt = (size_t)n_iter * K;
    model->trees.clear(); model->trees.resize(nt);
    size_t total = 0;
    for (size_t t = 0; t < nt; ++t) total += Tree::doubles(std::max(h.L[t], 1));
    model->tree_arena.reset(new double[std::max<size_t>(total, 1)]);
    size_t off = 0;
    for (size_t t = 0; t < nt; ++t) { const int L = std::max(h.L[t], 1); model->trees[t].place(model->tree_arena.get() + off, L); off += Tree::doubles(L); }
    return nt;
}
void model_trees_fill(rgbm_model* model, const HostTrees& h, int NL, size_t t0, size_t t1) {
    for (size_t t = t0; t < t1; ++t) {
        Tree& tr = model->trees[t];               // (placed by model_trees_begin)
        const size_t n = (size_t)tr.L - 1, nb = t * (NL - 1), lb = t * NL;
        memcpy(tr.feat, h.feat + nb, 4 * n); memcpy(tr.theta, h.theta + nb, 4 * n); memcpy(tr.dleft, h.dleft + nb, 4 * n);
        memcpy(tr.left, h.left + nb, 4 * n); memcpy(tr.right, h.right + nb, 4 * n); memcpy(tr.gain, h.gain + nb, 8 * n);
        memcpy(tr.leaf_value, h.val + lb, 8 * (size_t)tr.L); memcpy(tr.leaf_count, h.cnt + lb, 4 * (size_t)tr.L);
    }
}
void model_from_trees(rgbm_model* model, const HostTrees& h, int NE, int K, int NL) {
    const size_t n = model_trees_begin(model, h, NE, K);
    model_trees_fill(model, h, NL, 0, n);
}

// per-tree feature masks (ColSampler::ResetByTree), generated in LightGBM's draw order: [n_estimators * K][F]
std::vector<uint8_t> make_used_masks(const rgbm_params& p, size_t NT, int F, const std::vector<uint8_t>& trivial) {
    std::vector<uint8_t> used(NT * F, 0);
    LgbRand sr((uint32_t)p.seed);
    sr.rnd16(); sr.rnd16(); sr.rnd16();
    LgbRand ff((uint32_t)sr.rnd16());
    std::vector<int> valid; for (int f = 0; f < F; ++f) if (!trivial[f]) valid.push_back(f);
    for (size_t t = 0; t < NT; ++t) {
        uint8_t* u = used.data() + t * F;
        if (p.feature_fraction < 1.0) {
            int cntf = (int)std::floor((double)valid.size() * p.feature_fraction + 0.5);
            if (cntf < 1) cntf = 1;
            for (int i : ff.sample((int)valid.size(), cntf)) u[valid[i]] = 1;
        } else for (int f : valid) u[f] = 1;
    }
    return used;
}

// ---------------------------------------------------------------------------------------------
// The trainer (GBDT::Train) on a resident table.
// ---------------------------------------------------------------------------------------------
rgbm_model* train_core(const rgbm_table& tab, int32_t target_col, const int32_t* feat_cols, int32_t F,
                       const double* y_value, const double* class_weight, const double* sample_weight_host,
                       const HostLabelStats* hls, const rgbm_params& p, rgbm_train_stats* stats) {
    using namespace rg;
    check_params(p);
    if (F <= 0) throw std::invalid_argument("no feature columns");
    if (target_col < 0 || target_col >= tab.c) throw std::invalid_argument("target column out of range");
    for (int f = 0; f < F; ++f) if (feat_cols[f] < 0 || feat_cols[f] >= tab.c) throw std::invalid_argument("feature column out of range");
    if (F > 65535) throw std::invalid_argument("more than 65535 feature columns");
    const int64_t N = tab.n;
    const int obj = p.objective;
    const int n_y = tab.n_codes[target_col];
    const int K = obj == 1 ? p.num_class : 1;
    if (obj == 1 && n_y > p.num_class) throw std::out_of_range("target has more label codes than num_class");
    if (obj == 0 && n_y > 2) throw std::out_of_range("binary objective with more than 2 label codes");
    if (obj == 2 && !y_value) throw std::out_of_range("regression needs the y_value dictionary");
    StreamGuard sg_; hipStream_t s = sg_.s;
    hipEvent_t ev_begin, ev_end; HIPCHK(hipEventCreate(&ev_begin)); HIPCHK(hipEventCreate(&ev_end));
    HIPCHK(hipEventRecord(ev_begin, s));

    const RunSwitches sw = read_switches();
    const bool timing = sw.timing;   // host wall-clock of the phases, to stderr
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    // ---- 1. code frequencies of the training rows (features + the target itself)
    std::vector<int32_t> cols(feat_cols, feat_cols + F); cols.push_back(target_col);
    std::vector<int32_t> ncod(F + 1); std::vector<long long> cnt_off(F + 2, 0);
    for (int f = 0; f <= F; ++f) { ncod[f] = tab.n_codes[cols[f]]; cnt_off[f + 1] = cnt_off[f] + std::max(ncod[f], 1); }
    DevBuf<int32_t> d_cols(F + 1), d_ncod(F + 1); DevBuf<long long> d_cnt_off(F + 2); DevBuf<unsigned int> d_cnt(cnt_off[F + 1]);
    d_cols.upload(cols.data(), F + 1, s); d_ncod.upload(ncod.data(), F + 1, s); d_cnt_off.upload(cnt_off.data(), F + 2, s); d_cnt.zero(s);
    const int32_t* d_ycol = tab.codes.p + (long long)target_col * N;
    {
        int gx = (int)std::min<int64_t>((N + 255) / 256, 1024);
        hipLaunchKernelGGL(k_count_codes, dim3(gx, F + 1), dim3(256), 0, s, tab.codes.p, (long long)N, d_ycol, d_cols.p, d_ncod.p, d_cnt_off.p, d_cnt.p,
                           tab.has_mult ? tab.mult.p : (const uint8_t*)nullptr);
    }
    // row-sharded: this table is one rank's row shard (rgbm_params.reserved bit 0 = RGBM_FLAG_ROW_SHARDED)
    if ((p.reserved & RGBM_FLAG_ROW_SHARDED) && g_comm.kind == 0) throw std::invalid_argument("row-sharded training requested but this thread has no communicator (rgbm_comm_init)");
    const bool dp = (p.reserved & RGBM_FLAG_ROW_SHARDED) != 0;
    if (dp) all_reduce(d_cnt.p, (size_t)cnt_off[F + 1], AR_U32, s);
    std::vector<unsigned int> cnt(cnt_off[F + 1]);
    d_cnt.download(cnt.data(), cnt.size(), s);
    if (dp) stream_sync_watchdog(s); else HIPCHK(hipStreamSynchronize(s));
    FitHost fh; fh.cols = cols; fh.ncod = ncod; fh.cnt_off = cnt_off; fh.cnt = cnt;
    fit_setup(tab, y_value, class_weight, hls, p, F, fh);
    if (obj == 2) y_value = fh.yv32.data();          // LightGBM keeps labels as float32: the dictionary rounded once
    const int64_t n_train = fh.n_train;
    rgbm_model* model = fh.model.get();
    std::unique_ptr<rgbm_model>& guard = fh.model;
    std::vector<FeatMeta>& fmeta = fh.fmeta; std::vector<ChunkMeta>& cmeta = fh.cmeta;
    std::vector<long long>& lut_off = fh.lut_off; std::vector<uint8_t>& lut = fh.lut; std::vector<uint8_t>& miss = fh.miss; std::vector<uint8_t>& trivial = fh.trivial;
    const int nchunk = fh.nchunk; const size_t lds_hist = fh.lds_hist;
    std::vector<double>& init = fh.init; TrainConst tc = fh.tc;
    const int NL = p.num_leaves, NE = p.n_estimators;

    const double t_bins = now();
    // ---- 4. device state
    DevBuf<FeatMeta> d_fmeta(F); DevBuf<ChunkMeta> d_cmeta(nchunk); DevBuf<long long> d_lut_off(F + 1); DevBuf<uint8_t> d_lut(lut.size()), d_miss(F);
    d_fmeta.upload(fmeta.data(), F, s); d_cmeta.upload(cmeta.data(), nchunk, s); d_lut_off.upload(lut_off.data(), F + 1, s);
    d_lut.upload(lut.data(), lut.size(), s); d_miss.upload(miss.data(), F, s);
    DevBuf<uint4> d_rec((size_t)nchunk * N);
    // rows with multiplicities (rgbm_table_set_row_multiplicity): the multiplicity rides in byte 15 of the row's LAST bin record, so that chunk
    // must leave the byte free; the level grower only (one chunk, or two chunks in one pass), no bagging (LightGBM draws per ORIGINAL row), no per-row weights
    const bool wm = tab.has_mult;
    const uint8_t* d_mult = wm ? tab.mult.p : (const uint8_t*)nullptr;
    if (wm) {
        const bool lvl = p.max_depth >= 1 && p.max_depth <= LV_MAX_DEPTH && F <= 255 && read_switches().grower != 2;
        if (!lvl || nchunk > 2 || (F % 16) == 0 || sample_weight_host || (p.bagging_freq > 0 && p.bagging_fraction < 1.0))
            throw std::invalid_argument("a table with row multiplicities trains with the level grower (1 <= max_depth <= 7), at most 32 features of which the last 16-feature "
                                        "chunk holds at most 15, without bagging and without per-row weights");
    }
    hipLaunchKernelGGL(k_pack_bins, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, tab.codes.p, (long long)N, 0ll, (long long)N,
                       d_cols.p, d_ncod.p, d_lut_off.p, d_lut.p, d_miss.p, F, nchunk, d_rec.p, d_mult);
    // grower choice: the level-synchronous streaming grower (rgbm_level.h) whenever it applies; RGBM_GROWER=leafwise forces the
    // index-list grower (both are HIP; they produce identical models)
    const bool use_bagging = p.bagging_freq > 0 && p.bagging_fraction < 1.0;
    const bool level_mode = p.max_depth >= 1 && p.max_depth <= LV_MAX_DEPTH && F <= 255 && sw.grower != 2;
    if (dp && (!level_mode || sample_weight_host))
        throw std::invalid_argument("row-sharded training supports the level grower (1 <= max_depth <= 7) without per-row weights");
    DevBuf<int32_t> d_base; DevBuf<unsigned int> d_counter(1); d_counter.zero(s);
    if (!level_mode || use_bagging) {
        d_base.alloc(n_train);
        hipLaunchKernelGGL(k_iota_train, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d_ycol, (long long)N, d_base.p, d_counter.p);
    }
    if (level_mode) tc.NG = (N + 255) & ~255ll;            // level grower: every wave tile of a class tree's (g, h) row is in bounds and 32-byte aligned
    DevBuf<float2> d_gh((size_t)K * tc.NG); d_gh.zero(s);      // float32 (g, h) of every (row, class tree); non-training rows stay (0, 0)
    DevBuf<double> d_score((size_t)K * N), d_init(K);
    d_init.upload(init.data(), K, s);
    DevBuf<unsigned long long> d_fxq((size_t)K * 2); d_fxq.zero(s);      // numerics v2.2: coarse gradient sums of the iteration's class trees (k_fx_scale zeroes them again)
    DevBuf<FxScale> d_fxs(K);                                            // ... and the grid they give every class tree
    hipLaunchKernelGGL(k_init_score, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d_score.p, (long long)N, K, d_init.p);
    DevBuf<int32_t> d_idx0, d_idx1, d_sorted, d_any(NE); d_any.zero(s);
    DevBuf<HistBin> d_pool; DevBuf<TreeState> d_state; DevBuf<Leaf> d_leaves; DevBuf<Cand> d_cand; DevBuf<double> d_upd;
    if (!level_mode) {
        d_idx0.alloc((size_t)K * n_train); d_idx1.alloc((size_t)K * n_train);
        d_pool.alloc((size_t)K * NL * tc.totbins);
        d_state.alloc(K); d_leaves.alloc((size_t)K * NL); d_cand.alloc((size_t)K * 2 * F);
        d_upd.alloc((size_t)K * NL); d_sorted.alloc((size_t)K * NL * 3);
    }
    // ---- level grower state
    LevelConst lc; memset(&lc, 0, sizeof(lc));
    DevBuf<uint8_t> d_node; DevBuf<LvPlan> d_plan; DevBuf<SNode> d_snodes; DevBuf<Cand> d_lcand;
    DevBuf<HistBin> d_part, d_lpool; DevBuf<int32_t> d_count, d_count_g, d_err, d_leafnode; DevBuf<HistBin> d_part_red; DevBuf<double> d_ndelta; DevBuf<unsigned long long> d_statrows;
    DevBuf<uint32_t> d_prog; uint32_t mt_epoch = 0;   // lock-step progress words of the wave-specialised level pass: [row blocks][tree groups]
    int n_hnodes = 1;
    bool use_reduce = false;   // root pass: sum the per-workgroup partials in a separate kernel (many workgroups per class tree, joint bins, or row-sharded)
    // joint bins for the root pass (rgbm_level.h, k_pack_joint): a second record whose bytes hold GROUPS of low-cardinality features
    bool joint_root = false, joint_wide = false; int vtotbins = 0;
    LevelConst lcj; memset(&lcj, 0, sizeof(lcj));
    DevBuf<FeatMeta> d_vfmeta; DevBuf<ChunkMeta> d_vcmeta; DevBuf<uint4> d_rec_j; DevBuf<HistBin> d_part_j, d_red_j;
    DevBuf<JointFeat> d_jf; DevBuf<int16_t> d_binfeat;
    // one k_level_mt launch of a level: (chunk, window of built slots); the first one of a level routes
    struct MtLaunch { int ch, slot0, nslots, route, T, G, gx, acc2, rot; };
    std::vector<std::vector<MtLaunch>> mt_plan(LV_MAX_DEPTH + 1);
    std::vector<int> mt_gx(LV_MAX_DEPTH + 1, 8);           // row blocks per class tree of a level's launches (one value per level: the partials share it)
    if (level_mode) {
        const long long ntiles = (N + LV_TILE - 1) / LV_TILE;
        lc.lds_bytes = LV_LDS_BYTES;
        if (sw.lv_lds >= 65536 && sw.lv_lds <= LV_LDS_BYTES) lc.lds_bytes = sw.lv_lds;      // testing: a smaller LDS pool forces several built-slot windows per level
        lc.nchunk = nchunk; lc.K = K; lc.F = F; lc.totbins = tc.totbins;
        lc.num_leaves = NL; lc.max_depth = p.max_depth; lc.min_data_in_leaf = p.min_data_in_leaf; lc.N = N; lc.NS = (N + 255) & ~255ll; lc.NG = lc.NS;
        lc.sib_local = (dp && g_comm.rank == 0) ? 1 : 0;
        lc.has_mult = wm ? 1u : 0u;
        n_hnodes = (1 << p.max_depth) - 1;
        // ---- root pass: one 1024-thread workgroup per CU and (class tree, row block, chunk).  Contiguous row blocks, a multiple of 8 per
        // class tree with all class trees of a row block on one XCD (k_level_root), about four rounds of 256 workgroups when the class
        // trees alone nearly fill the chip (measured at K = 64, 10M rows: 8 / 16 / 40 blocks per class tree = 23.0 / 21.0 / 23.2 ms per
        // iteration), else the fewest blocks that fill one round; <= 2^22 rows per workgroup.  RGBM_LV_BLOCKS overrides.
        {
            const long long cu_slots = 256, per = (long long)K * nchunk;
            const long long gmin = std::max<long long>(1, (N + (1ll << 22) - 1) >> 22);
            auto eff_of = [&](long long g) { const long long tot = g * per, rounds = (tot + cu_slots - 1) / cu_slots; return (double)tot / (double)(rounds * cu_slots); };
            const long long gfloor = std::max<long long>(8, (gmin + 7) / 8 * 8), gcap = std::max<long long>(gfloor, std::min<long long>(512, (ntiles + 7) / 8 * 8));
            long long g2 = gfloor; double best2 = -1.0;
            for (long long g = gfloor; g <= gcap; g += 8) {
                if (per >= 16 && (g * per < 3 * cu_slots || g * per > 5 * cu_slots)) continue;
                const double e = eff_of(g);
                if (e > best2 + 1e-9) { best2 = e; g2 = g; }
                if (e >= 0.99) break;
            }
            if (best2 < 0.0) for (long long g = gfloor; g <= gcap; g += 8) { const double e = eff_of(g); if (e > best2 + 1e-9) { best2 = e; g2 = g; } if (e >= 0.99) break; }
            if (sw.lv_blocks >= 1) g2 = (sw.lv_blocks + 7) / 8 * 8;
            lc.gx = (int)g2; lc.xcd_blocks = 1; lc.max_built = 1;
        }
        for (int ch = 0; ch < nchunk; ++ch)
            if ((long long)lv_slots(fmeta.data() + cmeta[ch].first_feat, cmeta[ch].nfeat, 0) * 16 + LV_ROOT_FIXED > lc.lds_bytes)
                throw std::invalid_argument("histogram of one node exceeds LDS");
        // ---- level passes (k_level_mt).  Per chunk: how many built nodes the LDS holds next to the rings (cap), hence how many class
        // trees share a workgroup (T = cap / worst-case built nodes of the level, 2^(L-1)) or, when one class tree's nodes do not fit,
        // how many launches ("windows" of built slots) the level takes.  Every class tree group gets the same row blocks, a multiple of 8
        // (one XCD each), enough to fill the chip once (<= 2^22 rows per workgroup).
        size_t part_items = (size_t)K * lc.gx;                   // root partials: one node per (class tree, row block)
        for (int level = 1; level < p.max_depth; ++level) {
            const int worst = 1 << (level - 1);
            int G_first = 1;
            // tables of 17..32 features: ONE pass per level accumulates both 16-feature chunks (k_level_mt<2, ., ., MT_THREADS_ACC2, true>: both
            // records of a row in registers and in the ring) instead of one pass per chunk that streams every (node id, g, h) again
            bool acc2 = nchunk == 2 && sw.mt_acc2;
            if (acc2) {   // ... when one node's histograms of both chunks fit the LDS of the 768-thread workgroup (else: one pass per chunk, as for > 32 features)
                const long long nb2 = ((long long)lv_slots(fmeta.data() + cmeta[0].first_feat, cmeta[0].nfeat, 0) + lv_slots(fmeta.data() + cmeta[1].first_feat, cmeta[1].nfeat, 0) + MT_ROT_DUMMY) * 16;
                if (nb2 > mt_hist_bytes(lc.lds_bytes, MT_THREADS_ACC2, true, sw.mt_spec != 0, cmeta[1].nfeat > 15 || wm)) acc2 = false;
            }
            if (wm && nchunk == 2 && !acc2) throw std::invalid_argument("a two-chunk table with row multiplicities needs the one-pass level form (the multiplicity rides in the second record)");
            for (int ch = 0; ch < (acc2 ? 1 : nchunk); ++ch) {
                const FeatMeta* fm = fmeta.data() + cmeta[ch].first_feat;
                long long node_bytes = (long long)lv_slots(fm, cmeta[ch].nfeat, 0) * 16 + MT_ROT_DUMMY * 16;
                if (acc2) node_bytes += (long long)lv_slots(fmeta.data() + cmeta[1].first_feat, cmeta[1].nfeat, 0) * 16;
                const int mt_thr = acc2 ? MT_THREADS_ACC2 : ((nchunk == 1 && sw.mt_threads == 768 && sw.mt_spec != 1) ? 768 : LV_THREADS);
                const bool spec = acc2 ? sw.mt_spec != 0 : (nchunk == 1 && sw.mt_spec == 1);
                const bool li_arr = cmeta[acc2 ? 1 : ch].nfeat > 15 || wm;          // the ring needs a slot array (k_level_mt: !li_in_rec)
                const long long hist_room = mt_hist_bytes(lc.lds_bytes, mt_thr, acc2, spec, li_arr);
                long long cap = hist_room / std::max<long long>(node_bytes, 1);
                cap = std::min<long long>(cap, MT_MAX_NODES);
                if (cap < 1) throw std::invalid_argument("histogram of one node exceeds LDS");
                const int win = (int)std::min<long long>(worst, cap);                    // built slots per launch
                // LDS left over after the T class trees' nodes becomes replication (2^s copies of every bin).  A batch of 64 built rows comes from
                // one or two class trees (a wave walks them one after another) and, at the shallow levels, from one or two nodes: without
                // replication its lanes pile up on a handful of addresses (measured at K = 64: 23-31 cycles per atomic instruction at
                // replication 1 against 12.8 in the root pass).  So T is sized for `mt_rep` copies first; the records the extra workgroups
                // re-read come out of the XCD's L2 (all class tree groups of a row block run on one XCD at the same time).
                const long long rep_target = std::max(1, sw.mt_rep);
                const long long t_nodes = std::max<long long>(win, cap / rep_target);
                int T = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(t_nodes / win, MT_MAX_T), std::min<long long>(K, MT_RT_BUDGET / (2 * worst))));
                if (sw.mt_T >= 1) T = std::max(1, std::min(T, sw.mt_T));
                // Feature rotation of the histogram updates (rgbm_level.h, MT_ROT) for the launches whose LDS holds fewer than eight copies (RGBM_MT_ROT_COPIES2 / 2) of their
                // worst-case histograms -- the deepest dense levels, where rows of one cluster pile up on one address per feature: measured -11 % at
                // level 5 of the K = 64 target and +35-60 % where there IS room for replicas (profiles/r5c_*), hence per launch.  Plain one-chunk pass only.
                const bool plain1 = (!acc2 && nchunk == 1 && !spec && mt_thr == LV_THREADS) || (acc2 && spec);      // the two instantiations that exist with rotation
                // (a rotated step works on sixteen feature slots per chunk whatever the chunk holds: with few features most lanes of a step idle -- rotate only chunks
                // that fill at least three quarters of them)
                const bool rot_full = cmeta[ch].nfeat >= 12 && (!acc2 || cmeta[1].nfeat >= 12);
                // ... and (round 6) where the replicated layout would leave the workgroup fewer than RGBM_MT_ROT_TMIN (2: a single one) class trees although one copy holds at least twice
                // as many: every doubling of the class trees per workgroup is ~5 % of a deep level (the records are read once per T steps), which eight copies do not buy back
                const long long cap_rot1 = std::min<long long>(hist_room / (node_bytes + mt_rot_dummy(true) * 16), MT_MAX_NODES);
                const long long T_rot1 = std::min<long long>(std::min<long long>(cap_rot1 / win, MT_MAX_T), std::min<long long>(K, MT_RT_BUDGET / (2 * worst)));
                const bool few_trees = T < sw.mt_rot_tmin && T_rot1 >= 2 * (long long)T;
                const bool rot = plain1 && (sw.mt_rot == 1 || (sw.mt_rot < 0 && rot_full && (cap * 2 < (long long)sw.mt_rot_copies2 * T * win || few_trees))) &&
                                 (long long)T * win * (node_bytes + mt_rot_dummy(true) * 16) <= hist_room;
                // a rotated launch needs ONE copy: as many class trees per workgroup as the LDS holds (RGBM_MT_ROT_T=0: keep the T sized for replication)
                if (rot && sw.mt_rot_T && sw.mt_T < 1) {
                    const long long cap_rot = std::min<long long>(hist_room / (node_bytes + mt_rot_dummy(true) * 16), MT_MAX_NODES);
                    T = (int)std::max<long long>(T, std::min<long long>(std::min<long long>(cap_rot / win / std::max(1, sw.mt_rot_tdiv), MT_MAX_T), std::min<long long>(K, MT_RT_BUDGET / (2 * worst))));
                }
                const int G = (K + T - 1) / T;
                if (ch == 0) G_first = G;
                for (int s0 = 0; s0 < worst; s0 += win)
                    mt_plan[level].push_back(MtLaunch{ch, s0, win, (ch == 0 && s0 == 0) ? 1 : 0, T, G, 0, acc2 ? 1 : 0, rot ? 1 : 0});
            }
            const long long gmin = std::max<long long>(1, (N + (1ll << 22) - 1) >> 22);
            // row blocks per class tree: a multiple of 8 (one XCD each); G * gx workgroups should fill whole rounds of 256 CUs (G = 24: 8 row
            // blocks leave a quarter of the chip idle, 32 make three full rounds) without cutting the table into slivers
            const long long nwt = (N + MT_WT_ROWS - 1) / MT_WT_ROWS;
            const long long gx_lo = std::max<long long>(8, (gmin + 7) / 8 * 8), gx_hi = std::max<long long>(gx_lo, std::min<long long>(512, (nwt / (4 * (LV_THREADS / 64)) + 7) / 8 * 8));
            long long gx = gx_lo; double best = -1.0;
            for (long long g = gx_lo; g <= gx_hi; g += 8) {
                const long long tot = g * G_first, rounds = (tot + 255) / 256;
                const double eff = (double)tot / (double)(rounds * 256) - 0.01 * (double)rounds;   // fuller rounds first, then fewer of them
                if (eff > best + 1e-9) { best = eff; gx = g; }
                if (tot >= 256 && (double)tot / (double)(rounds * 256) >= 0.97) break;
            }
            if (sw.mt_blocks >= 1) gx = (sw.mt_blocks + 7) / 8 * 8;
            mt_gx[level] = (int)gx;
            for (auto& L : mt_plan[level]) L.gx = (int)gx;
            part_items = std::max(part_items, (size_t)K * (size_t)gx * (size_t)worst);
        }
        if (sw.timing) for (int level = 1; level < p.max_depth; ++level) for (const auto& L : mt_plan[level])
            fprintf(stderr, "[rgbm] level %d launch: chunk %d slots %d..%d route %d T %d G %d gx %d acc2 %d rot %d\n", level, L.ch, L.slot0, L.slot0 + L.nslots - 1, L.route, L.T, L.G, L.gx, L.acc2, L.rot);
        {
            size_t prog_words = 0;
            for (int level = 1; level < p.max_depth; ++level) for (const auto& L : mt_plan[level]) prog_words = std::max(prog_words, (size_t)L.G * (size_t)L.gx);
            d_prog.alloc(std::max<size_t>(prog_words, 1)); d_prog.zero(s);
        }
        d_node.alloc((size_t)K * lc.NS);
        // the padding rows [N, NS) of every class tree stay LV_INACTIVE for the whole fit (the gradient kernels reset rows < N only): the level pass
        // routes rows by their id alone, and an id past the end of a class tree's table takes the dummy entry
        HIPCHK(hipMemsetAsync(d_node.p, 0xFF, (size_t)K * lc.NS, s));
        d_plan.alloc(K); d_snodes.alloc((size_t)K * 256); d_lcand.alloc((size_t)K * 256 * F);
        d_part.alloc(part_items * tc.totbins); d_lpool.alloc((size_t)K * n_hnodes * tc.totbins);
        d_count.alloc((size_t)K * 256); use_reduce = dp || lc.gx > 4;
        if (dp) d_count_g.alloc((size_t)K * 256);
        const int max_built_all = 1 << std::max(0, p.max_depth - 2);
        d_part_red.alloc((size_t)K * max_built_all * tc.totbins + (size_t)K * 128 /* K*256 int64 counts */);
        d_leafnode.alloc((size_t)K * LV_MAX_LEAVES); d_err.alloc(1); d_err.zero(s); d_ndelta.alloc((size_t)K * 256); d_statrows.alloc(1); d_statrows.zero(s);
        // Joint bins for the root pass (the tables where the root pass runs at the LDS-atomic rate).  Best-fit-decreasing packing of the
        // features into groups whose bin counts multiply to <= 256; worth it when it saves at least two atomics per row and the groups
        // fit one 16-byte record.  RGBM_JOINT_ROOT=0 disables it (same models either way: the sums are exact integers).
        joint_root = sw.joint_root && ((long long)K * N >= (1ll << 21) || (wm && nchunk == 2));
        if (joint_root) {
            std::vector<int> order(F); for (int f = 0; f < F; ++f) order[f] = f;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return fmeta[a].nbins > fmeta[b].nbins; });
            std::vector<std::vector<int>> groups; std::vector<int> prod;
            auto pack = [&](int cap, std::vector<std::vector<int>>& gs, std::vector<int>& pr) {      // best-fit-decreasing into groups whose bin counts multiply to <= cap
                gs.clear(); pr.clear();
                for (int f : order) {
                    int best = -1;
                    for (size_t g = 0; g < gs.size(); ++g)
                        if ((long long)pr[g] * fmeta[f].nbins <= cap && (best < 0 || pr[g] > pr[best])) best = (int)g;
                    if (best < 0) { gs.emplace_back(); pr.push_back(1); best = (int)gs.size() - 1; }
                    gs[best].push_back(f); pr[best] *= fmeta[f].nbins;
                }
            };
            pack(256, groups, prod);
            // WIDE joint codes (round 5): 16-bit fields, groups of up to 1024 joint bins, at most eight per 16-byte record -- fewer groups = fewer atomics per row
            // (the synthetic table: 5 groups instead of 7, 10 atomics instead of 14) as long as two copies of the joint histograms fit the LDS
            {
                std::vector<std::vector<int>> gw; std::vector<int> pw;
                pack(JOINT_WIDE_CAP, gw, pw);
                long long slots2 = 0; for (int v : pw) slots2 += (long long)v * 2;
                if (sw.joint_wide && !wm && gw.size() <= 8 && gw.size() + 2 <= groups.size() && slots2 * 16 + LV_ROOT_FIXED <= lc.lds_bytes) { groups.swap(gw); prod.swap(pw); joint_wide = true; }
            }
            const int VF = (int)groups.size();
            if (VF > (joint_wide ? 8 : (wm ? 15 : 16)) || VF > F - 2) joint_root = false;
            else {
                std::vector<FeatMeta> vfm(VF); std::vector<JointFeat> jf(F); std::vector<int16_t> binfeat(tc.totbins, 0);
                ChunkMeta vcm; vcm.first_feat = 0; vcm.nfeat = VF; vcm.fast_slots = 0; vcm.wide_bins = 0;
                for (int v = 0; v < VF; ++v) {
                    FeatMeta& m = vfm[v]; memset(&m, 0, sizeof(m));
                    m.V = prod[v]; m.has_nan = 0; m.nbins = prod[v]; m.hoff = vcm.wide_bins; m.wide_off = vcm.wide_bins;
                    int sh = 0; while (sh < 5 && (m.nbins << (sh + 1)) <= 256) ++sh;
                    m.rep_shift = sh; m.fast_base = vcm.fast_slots; vcm.fast_slots += m.nbins << sh; vcm.wide_bins += m.nbins;
                    int stride = 1;
                    for (int f : groups[v]) {
                        JointFeat& j = jf[f]; memset(&j, 0, sizeof(j));
                        j.voff = m.hoff; j.stride = stride; j.nbins = fmeta[f].nbins; j.nbv = prod[v]; j.hoff = fmeta[f].hoff; j.vbyte = v;
                        stride *= fmeta[f].nbins;
                        for (int b = 0; b < fmeta[f].nbins; ++b) binfeat[fmeta[f].hoff + b] = (int16_t)f;
                    }
                }
                vtotbins = vcm.wide_bins;
                if ((long long)lv_slots(vfm.data(), VF, 0) * 16 + LV_ROOT_FIXED > lc.lds_bytes) joint_root = false;
                else {
                    d_vfmeta.alloc(VF); d_vfmeta.upload(vfm.data(), VF, s); d_vcmeta.alloc(1); d_vcmeta.upload(&vcm, 1, s);
                    d_jf.alloc(F); d_jf.upload(jf.data(), F, s); d_binfeat.alloc(binfeat.size()); d_binfeat.upload(binfeat.data(), binfeat.size(), s);
                    d_rec_j.alloc((size_t)N);
                    hipLaunchKernelGGL(k_pack_joint, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d_rec.p, (long long)N, F, d_jf.p, d_rec_j.p, joint_wide ? 1 : 0, wm ? nchunk - 1 : -1);
                    d_part_j.alloc((size_t)K * lc.gx * vtotbins); d_red_j.alloc((size_t)K * vtotbins + (size_t)K * 128 /* k_level_reduce parks the counts behind the bins */);
                    lcj = lc; lcj.nchunk = 1; lcj.F = VF; lcj.totbins = vtotbins; lcj.max_built = 1;
                    use_reduce = true;
                    HIPCHK(hipStreamSynchronize(s));   // the vectors above are locals
                }
            }
        }
        if (wm && nchunk == 2 && !joint_root)
            throw std::invalid_argument("a two-chunk table with row multiplicities needs the joint-bin root pass (its record carries the multiplicity); this feature set does not pack into 15 groups");
        // once per process and device: the attribute belongs to the function, not to the call, and other threads are launching these
        // kernels while a new training call sets up
        {
            static std::mutex attr_mu; static std::vector<char> attr_done(64, 0);
            std::lock_guard<std::mutex> lk(attr_mu);
            if (!attr_done[tab.device & 63]) {
                HIPCHK(hipFuncSetAttribute((const void*)k_level_root<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LV_LDS_BYTES));
                HIPCHK(hipFuncSetAttribute((const void*)k_level_root<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LV_LDS_BYTES));
#define RGBM_MT_ATTR1(NCHR, BAG, ROUTE, THR, ACC, ...) HIPCHK(hipFuncSetAttribute((const void*)k_level_mt<NCHR, BAG, ROUTE, THR, ACC, __VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, LV_LDS_BYTES))
#define RGBM_MT_ATTR(NCHR, THR, ACC, SPEC) RGBM_MT_ATTR1(NCHR, false, true, THR, ACC, SPEC); RGBM_MT_ATTR1(NCHR, false, false, THR, ACC, SPEC); \
                                           RGBM_MT_ATTR1(NCHR, true, true, THR, ACC, SPEC); RGBM_MT_ATTR1(NCHR, true, false, THR, ACC, SPEC)
                RGBM_MT_ATTR(0, LV_THREADS, false, false); RGBM_MT_ATTR(1, LV_THREADS, false, false); RGBM_MT_ATTR(2, LV_THREADS, false, false);
                RGBM_MT_ATTR(2, MT_THREADS_ACC2, true, false); RGBM_MT_ATTR(1, 768, false, false);
                RGBM_MT_ATTR(1, LV_THREADS, false, true); RGBM_MT_ATTR(2, MT_THREADS_ACC2, true, true);
                RGBM_MT_ATTR1(1, false, true, LV_THREADS, false, false, true); RGBM_MT_ATTR1(1, false, false, LV_THREADS, false, false, true);
                RGBM_MT_ATTR1(1, true, true, LV_THREADS, false, false, true); RGBM_MT_ATTR1(1, true, false, LV_THREADS, false, false, true);
                RGBM_MT_ATTR1(2, false, true, MT_THREADS_ACC2, true, true, true); RGBM_MT_ATTR1(2, false, false, MT_THREADS_ACC2, true, true, true);
                RGBM_MT_ATTR1(2, true, true, MT_THREADS_ACC2, true, true, true); RGBM_MT_ATTR1(2, true, false, MT_THREADS_ACC2, true, true, true);
#undef RGBM_MT_ATTR
#undef RGBM_MT_ATTR1
                attr_done[tab.device & 63] = 1;
            }
        }
    }
    const size_t NT = (size_t)NE * K;
    DevBuf<int32_t> t_L(NT), t_feat(NT * (NL - 1)), t_theta(NT * (NL - 1)), t_dleft(NT * (NL - 1)), t_left(NT * (NL - 1)), t_right(NT * (NL - 1)), t_cnt(NT * NL);
    DevBuf<double> t_gain(NT * (NL - 1)), t_val(NT * NL);
    t_cnt.zero(s); t_val.zero(s);
    TreeOut to{t_L.p, t_feat.p, t_theta.p, t_dleft.p, t_left.p, t_right.p, t_gain.p, t_val.p, t_cnt.p};
    DevBuf<double> d_cw, d_yv, d_sw;
    if (class_weight) { d_cw.alloc(n_y); d_cw.upload(class_weight, n_y, s); }
    if (y_value) { d_yv.alloc(n_y); d_yv.upload(y_value, n_y, s); }
    if (sample_weight_host) { d_sw.alloc(N); d_sw.upload(sample_weight_host, N, s); }

    const std::vector<uint8_t> used = make_used_masks(p, NT, F, trivial);
    DevBuf<uint8_t> d_used(used.size()); d_used.upload(used.data(), used.size(), s);

    // bagging state (GBDT::Bagging): stable training-row order, one LCG per 1024 positions
    DevBuf<int32_t> d_sorted_rows, d_oob; DevBuf<unsigned int> d_blk, d_rand, d_bagcnt, d_bagcnt_g; DevBuf<uint8_t> d_inbag;
    long long bag_off = 0, bag_local = 0, bag_nrb = 1;
    if (use_bagging) {
        const long long nblk = (N + 1023) / 1024;
        d_blk.alloc(nblk); d_sorted_rows.alloc(n_train); d_oob.alloc(n_train); d_inbag.alloc(N); d_bagcnt.alloc(2);
        hipLaunchKernelGGL(k_block_count, dim3((unsigned)nblk), dim3(256), 0, s, d_ycol, (long long)N, d_blk.p);
        hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, s, d_blk.p, nblk);
        hipLaunchKernelGGL(k_stable_compact, dim3((unsigned)nblk), dim3(256), 0, s, d_ycol, (long long)N, d_blk.p, d_sorted_rows.p);
        if (dp) {
            // GBDT::Bagging draws per training-row POSITION (ascending row order over the whole table): this rank's rows hold the positions
            // [bag_off, bag_off + bag_local), bag_off = training rows of the ranks before it (one all-reduce of a per-rank count vector)
            unsigned int h_local = 0; d_counter.download(&h_local, 1, s); stream_sync_watchdog(s);
            bag_local = (long long)h_local;
            std::vector<long long> rc((size_t)g_comm.nranks, 0); rc[(size_t)g_comm.rank] = bag_local;
            DevBuf<long long> d_rc(rc.size()); d_rc.upload(rc.data(), rc.size(), s);
            all_reduce(d_rc.p, rc.size(), AR_I64, s);
            d_rc.download(rc.data(), rc.size(), s); stream_sync_watchdog(s);
            long long tot = 0;
            for (int r = 0; r < g_comm.nranks; ++r) { if (r < g_comm.rank) bag_off += rc[(size_t)r]; tot += rc[(size_t)r]; }
            if (tot != n_train) throw std::runtime_error("row-sharded bagging: the ranks' training-row counts do not add up");
            d_bagcnt_g.alloc(2);
        } else bag_local = n_train;
        bag_nrb = std::max<long long>(1, (bag_off + bag_local + 1023) / 1024 - bag_off / 1024);
        LgbRand sr2((uint32_t)p.seed); sr2.rnd16();
        const int bagging_seed = sr2.rnd16();
        std::vector<unsigned int> st(bag_nrb);
        for (long long b = 0; b < bag_nrb; ++b) st[b] = (unsigned int)(bagging_seed + bag_off / 1024 + b);
        d_rand.alloc(bag_nrb); d_rand.upload(st.data(), bag_nrb, s);
        HIPCHK(hipStreamSynchronize(s));
    }
    // rows in the bag of the current iteration, over ALL ranks when row-sharded (the root count of every tree)
    const unsigned int* n_in_ptr = use_bagging ? (dp ? d_bagcnt_g.p : d_bagcnt.p) : nullptr;

    if (!level_mode) {
        if (lds_hist > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hist));
        if (lds_hist > 160 * 1024) throw std::invalid_argument("histogram working set exceeds LDS");
    }

    const long long root_tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
    const int hist_gx = (int)std::max<long long>(1, std::min<long long>(root_tiles, (1536 + (long long)K * nchunk - 1) / ((long long)K * nchunk)));
    const int part_gx = (int)std::max<long long>(1, std::min<long long>((n_train + 1023) / 1024, (1024 + K - 1) / K));
    const int upd_gx = (int)std::max<long long>(1, std::min<long long>((n_train + 255) / 256, (2048 + K - 1) / K));
    const int grad_gx = (int)std::min<long long>((N + 255) / 256, 4096);
    const size_t upd_lds = (size_t)(3 * NL + (NL & 1)) * 4 + (size_t)NL * 8;

    std::vector<std::pair<hipEvent_t, hipEvent_t>> hist_ev;   // only when stats are requested
    std::vector<char> hist_ev_root;
    auto launch_hist = [&](bool root) {
        hipEvent_t a = nullptr, b = nullptr;
        if (stats) { HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); HIPCHK(hipEventRecord(a, s)); }
        hipLaunchKernelGGL(k_hist, dim3(hist_gx, K, nchunk), dim3(256), lds_hist, s, d_rec.p, d_gh.p, d_idx0.p, d_idx1.p, d_base.p,
                           d_state.p, d_pool.p, d_fmeta.p, d_cmeta.p, d_fxs.p, tc);
        if (stats) { HIPCHK(hipEventRecord(b, s)); hist_ev.emplace_back(a, b); hist_ev_root.push_back(root ? 1 : 0); }
    };

    const int score_gx = (int)std::max<long long>(1, std::min<long long>((N + 1023) / 1024, (2048 + K - 1) / K));
    const int last_gx = (int)std::max<long long>(1, std::min<long long>((N + 4095) / 4096, (2048 + K - 1) / K));      // k_level_last: 16 rows per thread and step
    const bool defer_score = sw.defer_score;
    auto timed = [&](bool root, auto&& fn) {   // HIP events around one histogram launch, on the stream it is launched on
        hipEvent_t a = nullptr, b = nullptr;
        if (stats) { HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); HIPCHK(hipEventRecord(a, s)); }
        fn();
        if (stats) { HIPCHK(hipEventRecord(b, s)); hist_ev.emplace_back(a, b); hist_ev_root.push_back(root ? 1 : 0); }
    };
    auto launch_root = [&]() {
        timed(true, [&]() {
            if (joint_root && joint_wide)
                hipLaunchKernelGGL(k_level_root<true>, dim3((unsigned)lc.gx * (unsigned)K, 1, 1), dim3(LV_THREADS), lc.lds_bytes, s, d_rec_j.p, d_gh.p, d_node.p, d_plan.p, d_part_j.p,
                                   d_vfmeta.p, d_vcmeta.p, d_fxs.p, lcj);
            else if (joint_root)   // the root pass over the joint record: one pair of atomics per feature GROUP and row
                hipLaunchKernelGGL(k_level_root<false>, dim3((unsigned)lc.gx * (unsigned)K, 1, 1), dim3(LV_THREADS), lc.lds_bytes, s, d_rec_j.p, d_gh.p, d_node.p, d_plan.p, d_part_j.p,
                                   d_vfmeta.p, d_vcmeta.p, d_fxs.p, lcj);
            else
                hipLaunchKernelGGL(k_level_root<false>, dim3((unsigned)lc.gx * (unsigned)K, 1, nchunk), dim3(LV_THREADS), lc.lds_bytes, s, d_rec.p, d_gh.p, d_node.p, d_plan.p, d_part.p,
                                   d_fmeta.p, d_cmeta.p, d_fxs.p, lc);
        });
    };
    // the k_level_mt launches of one level; returns the LevelConst that describes the partials they wrote
    auto launch_level = [&](int level) -> LevelConst {
        LevelConst ll = lc;
        ll.xcd_blocks = 0; ll.gx = mt_gx[level]; ll.max_built = 1 << (level - 1);
        for (const MtLaunch& L : mt_plan[level]) {
            LevelConst l1 = ll;
            l1.mt_T = L.T; l1.mt_G = L.G; l1.mt_ch = L.ch; l1.mt_slot0 = L.slot0; l1.mt_nslots = L.nslots; l1.mt_route = L.route; l1.mt_sparse = sw.mt_sparse ? 1 : 0;
            // lock-step of the class-tree groups of a row block (wave-specialised pass): window in tile rounds, one round = 8 wave tiles of records
            // off by default: measured -63 % HBM fetch and +23 % time on 100M x 32 (profiles/r5e_*): the pass is not bound by that traffic
            l1.mt_window = sw.mt_lock < 0 ? 0 : sw.mt_lock;
            l1.mt_epoch = (mt_epoch++ % 4095u) + 1u;      // (never 0: the words start zeroed)
            const dim3 grid((unsigned)L.G * (unsigned)L.gx);
            const int nchr = nchunk == 1 ? 1 : (nchunk == 2 ? 2 : 0);
            timed(false, [&]() {
#define RGBM_LAUNCH_MT2(NCHR, BAG, INBAG, THR, ACC, ...) do { if (L.route) hipLaunchKernelGGL((k_level_mt<NCHR, BAG, true, THR, ACC, __VA_ARGS__>), grid, dim3(THR), LV_LDS_BYTES /* (always the CU's whole LDS: the hessian sums sit a compile-time distance behind the gradient sums; lc.lds_bytes, the test hook, only sizes the histograms) */, s, d_rec.p, d_gh.p, d_node.p, (const uint8_t*)(INBAG), d_plan.p, \
                                                                          d_part.p, d_count.p, d_fmeta.p, d_cmeta.p, d_err.p, d_prog.p, d_fxs.p, l1); \
                                           else hipLaunchKernelGGL((k_level_mt<NCHR, BAG, false, THR, ACC, __VA_ARGS__>), grid, dim3(THR), LV_LDS_BYTES, s, d_rec.p, d_gh.p, d_node.p, (const uint8_t*)(INBAG), d_plan.p, \
                                                                   d_part.p, d_count.p, d_fmeta.p, d_cmeta.p, d_err.p, d_prog.p, d_fxs.p, l1); } while (0)
#define RGBM_LAUNCH_MT(NCHR, THR, ACC, ...) do { if (use_bagging) RGBM_LAUNCH_MT2(NCHR, true, d_inbag.p, THR, ACC, __VA_ARGS__); else RGBM_LAUNCH_MT2(NCHR, false, nullptr, THR, ACC, __VA_ARGS__); } while (0)
                if (L.acc2) { if (sw.mt_spec != 0 && L.rot) RGBM_LAUNCH_MT(2, MT_THREADS_ACC2, true, true, true); else if (sw.mt_spec != 0) RGBM_LAUNCH_MT(2, MT_THREADS_ACC2, true, true); else RGBM_LAUNCH_MT(2, MT_THREADS_ACC2, true, false); }
                else if (nchr == 1 && sw.mt_spec == 1) RGBM_LAUNCH_MT(1, LV_THREADS, false, true);
                else if (nchr == 1 && sw.mt_threads == 768) RGBM_LAUNCH_MT(1, 768, false, false);
                else if (nchr == 1 && L.rot) RGBM_LAUNCH_MT(1, LV_THREADS, false, false, true);
                else if (nchr == 1) RGBM_LAUNCH_MT(1, LV_THREADS, false, false);
                else if (nchr == 2) RGBM_LAUNCH_MT(2, LV_THREADS, false, false);
                else RGBM_LAUNCH_MT(0, LV_THREADS, false, false);
#undef RGBM_LAUNCH_MT2
#undef RGBM_LAUNCH_MT
            });
        }
        return ll;
    };

    DevBuf<int32_t> d_it(1); d_it.zero(s);   // device-side iteration counter (k_next_iteration)
    // RGBM_TRACE=<dir> (debugging; with RGBM_TRACE_K=<class trees> and RGBM_TRACE_ITERS=lo:hi): snapshots of the grower's state after every
    // level of the iterations lo..hi and a checksum of (score, g/h, node ids) after every iteration, written to <dir>/target<col>.{bin,idx}
    const char* trace_dir = getenv("RGBM_TRACE");
    const bool trace_on = trace_dir && level_mode && (!getenv("RGBM_TRACE_K") || atoi(getenv("RGBM_TRACE_K")) == K);
    int trace_lo = 44, trace_hi = 44, cur_it = 0;
    if (trace_on && getenv("RGBM_TRACE_ITERS")) sscanf(getenv("RGBM_TRACE_ITERS"), "%d:%d", &trace_lo, &trace_hi);
    struct TraceRec { std::string name; int it, level; size_t off, bytes; };
    std::vector<TraceRec> trace_idx; size_t trace_used = 0;
    DevBuf<unsigned char> d_trace;
    if (trace_on) d_trace.alloc(((size_t)K * n_hnodes * tc.totbins * 16 + (size_t)K * 256 * (sizeof(SNode) + 4) + (size_t)K * sizeof(LvPlan) + 4096) * (size_t)(p.max_depth + 1) * (size_t)(trace_hi - trace_lo + 1) + (size_t)NE * 64 + 4096);
    auto trace_copy = [&](const char* name, int level, const void* src, size_t bytes) {
        if (!trace_on || cur_it < trace_lo || cur_it > trace_hi) return;
        if (trace_used + bytes > d_trace.n) return;
        HIPCHK(hipMemcpyAsync(d_trace.p + trace_used, src, bytes, hipMemcpyDeviceToDevice, s));
        trace_idx.push_back({name, cur_it, level, trace_used, bytes}); trace_used += (bytes + 15) & ~(size_t)15;
    };
    auto trace_sum = [&](const char* name, const void* src, size_t bytes) {
        if (!trace_on || trace_used + 16 > d_trace.n) return;
        HIPCHK(hipMemsetAsync(d_trace.p + trace_used, 0, 8, s));
        hipLaunchKernelGGL(k_trace_sum, dim3(2048), dim3(256), 0, s, (const unsigned long long*)src, bytes / 8, (unsigned long long*)(d_trace.p + trace_used));
        trace_idx.push_back({name, cur_it, -1, trace_used, 8}); trace_used += 16;
    };
    auto trace_level = [&](int level) {
        trace_copy("count", level, d_count.p, (size_t)K * 256 * 4);
        trace_copy("plan", level, d_plan.p, (size_t)K * sizeof(LvPlan));
        trace_copy("snodes", level, d_snodes.p, (size_t)K * 256 * sizeof(SNode));
        trace_copy("lpool", level, d_lpool.p, (size_t)K * n_hnodes * tc.totbins * 16);
    };
    // one boosting iteration of the level grower after the gradients: an iteration-invariant launch sequence
    auto enqueue_level_growth = [&]() {
            hipLaunchKernelGGL(k_level_init, dim3(K), dim3(64), 0, s, d_plan.p, d_snodes.p, d_count.p, n_in_ptr, (long long)n_train, d_fxq.p, tc.fx, d_fxs.p, lc);
            int32_t* cntg = dp ? d_count_g.p : d_count.p;      // child row counts seen by split / leaf-count (global when row-sharded)
            // partials of this rank -> compact buffer (-> integer all-reduce when row-sharded); the split kernel then sees ONE partial
            auto exchange = [&](bool root, int nb, const LevelConst& lp) -> std::pair<const HistBin*, LevelConst> {
                if (root && !use_reduce) return {d_part.p, lp};
                if (root && joint_root) {   // partials -> one joint histogram per class tree (k_level_reduce in the joint bin space) -> marginals of the real features
                    hipLaunchKernelGGL(k_level_reduce, dim3((vtotbins + 63) / 64, 1, K), dim3(256), 0, s, d_part_j.p, d_red_j.p, d_plan.p, d_count.p, 1, 1, lcj);
                    hipLaunchKernelGGL(k_level_marginal, dim3((tc.totbins + 255) / 256, K), dim3(256), 0, s, d_red_j.p, d_part_red.p, d_plan.p, d_count.p, d_jf.p, d_binfeat.p, vtotbins, lc);
                } else
                hipLaunchKernelGGL(k_level_reduce, dim3((tc.totbins + 63) / 64, nb, K), dim3(256), 0, s, d_part.p, d_part_red.p, d_plan.p, d_count.p, root ? 1 : 0, nb, lp);
                if (dp) {
                    const size_t nh = (size_t)K * nb * tc.totbins * 2;          // int64 words of histograms, then K*256 child counts
                    all_reduce(d_part_red.p, nh + (size_t)K * 256, AR_I64, s);
                    hipLaunchKernelGGL(k_counts_unpack, dim3(K), dim3(256), 0, s, reinterpret_cast<const long long*>(d_part_red.p) + nh, d_count_g.p);
                }
                LevelConst r = lp; r.gx = 1; r.max_built = nb;
                return {d_part_red.p, r};
            };
            launch_root();
            {
                auto ex = exchange(true, 1, lc);
                hipLaunchKernelGGL(k_level_split<true>, dim3((F + 3) / 4, 1, K), dim3(256), 0, s, ex.first, d_lpool.p, d_plan.p, d_snodes.p, cntg, d_count.p, d_fmeta.p,
                                   d_used.p, d_lcand.p, d_statrows.p, d_it.p, n_hnodes, d_fxs.p, tc, ex.second);
                trace_level(0);
            }
            for (int level = 1; level < p.max_depth; ++level) {
                hipLaunchKernelGGL(k_level_plan, dim3(K), dim3(256), 0, s, d_plan.p, d_snodes.p, d_lcand.p, d_fmeta.p, level, tc, lc);
                const LevelConst lp = launch_level(level);
                auto ex = exchange(false, 1 << (level - 1), lp);
                hipLaunchKernelGGL(k_level_split<false>, dim3((F + 1) / 2, 1 << (level - 1), K), dim3(256), 0, s, ex.first, d_lpool.p, d_plan.p, d_snodes.p,
                                   cntg, d_count.p, d_fmeta.p, d_used.p, d_lcand.p, d_statrows.p, d_it.p, n_hnodes, d_fxs.p, tc, ex.second);
                trace_level(level);
            }
            // last level: plan -> replay (leaf values never depend on the deepest counts) -> route + count + score in one pass
            hipLaunchKernelGGL(k_level_plan, dim3(K), dim3(256), 0, s, d_plan.p, d_snodes.p, d_lcand.p, d_fmeta.p, p.max_depth, tc, lc);
            hipLaunchKernelGGL(k_level_replay, dim3(K), dim3(64), 0, s, d_plan.p, d_snodes.p, cntg, to, d_init.p, d_ndelta.p, d_leafnode.p, d_any.p, d_err.p, d_it.p, tc);
            // AddScore: the pass that routes, counts and adds in one (k_level_final).  RGBM_DEFER_SCORE=1 (measured slower, off): folded into the NEXT iteration's
            // gradient kernel, with k_level_last for the last routing step + the deepest counts; the last iteration has no next one and keeps k_level_final
            if (defer_score && cur_it + 1 < NE)
                hipLaunchKernelGGL(k_level_last, dim3(last_gx, K), dim3(256), 0, s, d_rec.p, d_node.p, use_bagging ? d_inbag.p : (const uint8_t*)nullptr, d_plan.p, to, d_count.p, d_it.p, lc);
            else
            hipLaunchKernelGGL(k_level_final, dim3(score_gx, K), dim3(256), 0, s, d_rec.p, d_node.p, use_bagging ? d_inbag.p : (const uint8_t*)nullptr,
                               d_plan.p, to, d_ndelta.p, d_score.p, d_count.p, d_it.p, lc);
            if (dp) { hipLaunchKernelGGL(k_copy_i32, dim3(K), dim3(256), 0, s, d_count.p, d_count_g.p, (long long)K * 256); all_reduce(d_count_g.p, (size_t)K * 256, AR_I32, s); }
            hipLaunchKernelGGL(k_level_leafcount, dim3(K), dim3(LV_MAX_LEAVES), 0, s, d_plan.p, cntg, d_leafnode.p, to, d_it.p, tc);
            trace_level(p.max_depth);
            trace_sum("score", d_score.p, (size_t)K * N * 8); trace_sum("node", d_node.p, (size_t)K * lc.NS);
    };

    // numerics v2.2: every class tree of an iteration gets its own fixed-point grid from the coarse sums Q_g, Q_h of its (g, h) (rgbm_numerics.h).  The
    // gradient kernels leave the sums per workgroup / wave (d_qpart [parts][K][2]: plain stores, no atomics), k_fx_reduce adds them up into d_fxq [K][2];
    // k_grad<1> (softmax with K > 112) and RGBM_FX_MEASURE=separate take them from a pass of their own over the (g, h) array (k_fx_measure).  Row-sharded:
    // ONE more integer all-reduce per iteration (2 K words).  k_fx_scale turns the sums into the FxScale table every accumulating / searching kernel reads.
    const bool mc_tile = obj == 1 && K >= 16 && K <= 112, mc_rows = obj == 1 && K < 16;
    const bool fx_fused = !sw.fx_separate && (obj != 1 || mc_tile || mc_rows);
    const long long fx_parts = !fx_fused ? 0 : (mc_tile ? (N + 63) / 64 : (mc_rows ? ((N + 255) / 256) * 4 : (long long)grad_gx));
    DevBuf<unsigned long long> d_qpart; if (fx_fused) d_qpart.alloc((size_t)fx_parts * K * 2);
    auto enqueue_grad = [&]() {
        const double* cw = class_weight ? d_cw.p : nullptr; const double* yv = y_value ? d_yv.p : nullptr; const double* sw_ = sample_weight_host ? d_sw.p : nullptr;
        const uint8_t* inbag = use_bagging ? d_inbag.p : nullptr;
        uint8_t* node0 = level_mode ? d_node.p : nullptr;
        unsigned long long* qp = fx_fused ? d_qpart.p : nullptr;
        // level grower: the AddScore of the previous iteration's trees rides in this kernel (PendingScore, rgbm_kernels.h)
        PendingScore pd{nullptr, nullptr, nullptr, 0};
        if (level_mode && defer_score && cur_it > 0) pd = PendingScore{d_node.p, d_ndelta.p, to.L + (size_t)(cur_it - 1) * K, lc.NS};
        if (obj == 0) hipLaunchKernelGGL(k_grad<0>, dim3(grad_gx), dim3(256), 0, s, d_score.p, d_ycol, yv, cw, sw_, inbag, d_gh.p, node0, lc.NS, qp, d_mult, pd, tc);
        else if (mc_rows)
            hipLaunchKernelGGL(k_grad_mc_rows<256>, dim3((unsigned)((N + 255) / 256)), dim3(256), (size_t)K * 256 * 8, s, d_score.p, d_ycol, cw, sw_, inbag, d_gh.p, node0, lc.NS, qp, d_mult, pd, tc);
        else if (mc_tile)
            hipLaunchKernelGGL(k_grad_mc, dim3((unsigned)((N + 63) / 64)), dim3(256), (size_t)(K * 64 + 320) * 8, s, d_score.p, d_ycol, cw, sw_, inbag, d_gh.p, node0, lc.NS, qp, d_mult, pd, tc);
        else if (obj == 1) hipLaunchKernelGGL(k_grad<1>, dim3(grad_gx), dim3(256), 0, s, d_score.p, d_ycol, yv, cw, sw_, inbag, d_gh.p, node0, lc.NS, (unsigned long long*)nullptr, d_mult, pd, tc);
        else hipLaunchKernelGGL(k_grad<2>, dim3(grad_gx), dim3(256), 0, s, d_score.p, d_ycol, yv, cw, sw_, inbag, d_gh.p, node0, lc.NS, qp, d_mult, pd, tc);
        if (fx_fused) {
            const long long total = fx_parts * K * 2;
            const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>(256, (total + 8191) / 8192));
            hipLaunchKernelGGL(k_fx_reduce, dim3(gx), dim3(256), (size_t)K * 2 * 8, s, d_qpart.p, fx_parts, 2 * K, d_fxq.p);
        } else {
            const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>(128, (N + 4095) / 4096));
            hipLaunchKernelGGL(k_fx_measure, dim3(gx, K), dim3(256), 0, s, d_gh.p, (long long)N, (long long)tc.NG, tc.fx, d_fxq.p, d_mult);
        }
        if (dp) all_reduce(d_fxq.p, (size_t)K * 2, AR_I64, s);
        if (!level_mode) hipLaunchKernelGGL(k_fx_scale, dim3(1), dim3(256), 0, s, d_fxq.p, K, tc.fx, d_fxs.p);      // (level grower: k_level_init does it)
    };

    if (timing) HIPCHK(hipStreamSynchronize(s));
    const double t_setup = now();
    // ---- 5. boosting iterations: everything below is enqueue-only.
    // hipGraph replay of an iteration was built and measured twice (rounds 1 and 2, MI355X / ROCm 7.2) and removed: a 10 000-row fit
    // takes the same time either way (the chain of ~30 small dependent kernels per iteration is bound by GPU-side latency, not by
    // the host launches), 24 concurrent host threads gain nothing (8.8 vs 9.2 ms per 60-iteration fit), and captures made while
    // other threads train fail or replay wrongly (20 bad models in 150) even with every graph call behind one mutex.
    auto enqueue_bagging = [&]() {
        hipLaunchKernelGGL(k_bagging, dim3((unsigned)((bag_nrb + 63) / 64)), dim3(64), 0, s, d_rand.p, bag_local, p.bagging_fraction, d_sorted_rows.p, d_inbag.p, d_bagcnt.p,
                           bag_off, (long long)n_train);
        hipLaunchKernelGGL(k_bag_lists, dim3((unsigned)((std::max<long long>(bag_local, 1) + 255) / 256)), dim3(256), 0, s, d_sorted_rows.p, bag_local, d_inbag.p, d_base.p, d_oob.p, d_bagcnt.p);
        if (dp) {
            hipLaunchKernelGGL(k_copy_i32, dim3(1), dim3(256), 0, s, reinterpret_cast<const int32_t*>(d_bagcnt.p), reinterpret_cast<int32_t*>(d_bagcnt_g.p), 2ll);
            all_reduce(d_bagcnt_g.p, 2, AR_U32, s);
        }
    };
    for (int it = 0; it < NE; ++it) {
        if (use_bagging && it % p.bagging_freq == 0) enqueue_bagging();
        cur_it = it;
        enqueue_grad();
        trace_sum("gh", d_gh.p, (size_t)K * tc.NG * 8);
        const uint8_t* usedp = d_used.p + (size_t)it * K * F;
        if (level_mode) {
            enqueue_level_growth();
            hipLaunchKernelGGL(k_next_iteration, dim3(1), dim3(1), 0, s, d_it.p);
            continue;
        }
        hipLaunchKernelGGL(k_init_iter, dim3(K), dim3(64), 0, s, d_state.p, d_leaves.p, d_pool.p, to, n_in_ptr, it, tc);
        for (int step = 0; step < NL - 1; ++step) {
            launch_hist(step == 0);
            hipLaunchKernelGGL(k_split_find, dim3((F + 3) / 4, K), dim3(256), 0, s, d_pool.p, d_state.p, d_leaves.p, d_fmeta.p, usedp, d_cand.p, d_fxs.p, tc);
            hipLaunchKernelGGL(k_tree_step, dim3(K), dim3(64), 0, s, d_state.p, d_leaves.p, d_cand.p, d_pool.p, d_fmeta.p, to, it, tc);
            hipLaunchKernelGGL(k_partition, dim3(part_gx, K), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(d_rec.p), d_idx0.p, d_idx1.p, d_base.p, d_state.p, tc);
            hipLaunchKernelGGL(k_finish_split, dim3(K), dim3(64), 0, s, d_state.p, d_leaves.p, to, it, tc);
        }
        hipLaunchKernelGGL(k_finalize_tree, dim3(K), dim3(64), 0, s, d_state.p, d_leaves.p, to, d_init.p, d_upd.p, d_sorted.p, d_any.p, it, tc);
        hipLaunchKernelGGL(k_score_update, dim3(upd_gx, K), dim3(256), upd_lds, s, d_state.p, d_leaves.p, d_sorted.p, d_upd.p, d_idx0.p, d_idx1.p, d_base.p, d_score.p, n_in_ptr, tc);
        if (use_bagging)
            hipLaunchKernelGGL(k_score_update_oob, dim3(upd_gx, K), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(d_rec.p), d_oob.p, d_bagcnt.p,
                               d_state.p, to, d_fmeta.p, d_upd.p, d_score.p, it, tc);
    }
    HIPCHK(hipGetLastError());

    const double t_enq = now();
    // ---- 6. trees back to the host
    std::vector<int32_t> hL(NT), hfeat(NT * (NL - 1)), htheta(NT * (NL - 1)), hdleft(NT * (NL - 1)), hleft(NT * (NL - 1)), hright(NT * (NL - 1)), hcnt(NT * NL), hany(NE);
    std::vector<double> hgain(NT * (NL - 1)), hval(NT * NL);
    t_L.download(hL.data(), hL.size(), s); t_feat.download(hfeat.data(), hfeat.size(), s); t_theta.download(htheta.data(), htheta.size(), s);
    t_dleft.download(hdleft.data(), hdleft.size(), s); t_left.download(hleft.data(), hleft.size(), s); t_right.download(hright.data(), hright.size(), s);
    t_cnt.download(hcnt.data(), hcnt.size(), s); t_gain.download(hgain.data(), hgain.size(), s); t_val.download(hval.data(), hval.size(), s);
    d_any.download(hany.data(), NE, s);
    int32_t h_err = 0; unsigned long long h_statrows = 0;
    if (level_mode) { d_err.download(&h_err, 1, s); d_statrows.download(&h_statrows, 1, s); }
    HIPCHK(hipEventRecord(ev_end, s));
    if (dp) stream_sync_watchdog(s); else HIPCHK(hipStreamSynchronize(s));
    if (timing) fprintf(stderr, "[rgbm] target %d K=%d: count+bins %.1f ms, alloc+pack %.1f ms, enqueue %.1f ms, drain+download %.1f ms\n", target_col, K, t_bins - t_start, t_setup - t_bins, t_enq - t_setup, now() - t_enq);
    if (trace_on && trace_used) {
        std::vector<unsigned char> h(trace_used);
        HIPCHK(hipMemcpy(h.data(), d_trace.p, trace_used, hipMemcpyDeviceToHost));
        char path[512];
        snprintf(path, sizeof(path), "%s/target%d.bin", trace_dir, target_col);
        if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 1, h.size(), f); fclose(f); }
        snprintf(path, sizeof(path), "%s/target%d.idx", trace_dir, target_col);
        if (FILE* f = fopen(path, "w")) { for (const auto& r : trace_idx) fprintf(f, "%s %d %d %zu %zu\n", r.name.c_str(), r.it, r.level, r.off, r.bytes); fclose(f); }
    }
    if (h_err & 2) throw std::runtime_error("level grower: a level pass was launched with more built nodes / route entries than its workgroup tables hold (sizing violated)");
    if (h_err) throw std::runtime_error("level grower: a node outside the speculative expansion was selected (expansion bound violated)");

    HostTrees ht{hL.data(), hfeat.data(), htheta.data(), hdleft.data(), hleft.data(), hright.data(), hcnt.data(), hgain.data(), hval.data(), hany.data()};
    model_from_trees(model, ht, NE, K, NL);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, ev_begin, ev_end)); stats->total_ms = ms;
        stats->hist_launches = (int64_t)hist_ev.size();
        for (size_t i = 0; i < hist_ev.size(); ++i) {
            float m2 = 0.f; HIPCHK(hipEventElapsedTime(&m2, hist_ev[i].first, hist_ev[i].second));
            stats->hist_ms += m2; if (hist_ev_root[i]) stats->root_ms += m2;
            (void)hipEventDestroy(hist_ev[i].first); (void)hipEventDestroy(hist_ev[i].second);
        }
        if (level_mode) {
            // rows whose (g,h) were accumulated: counted on the device (k_level_split); root passes = trees with a searched root
            int64_t root_rows = 0;
            const bool root_done = n_train >= (int64_t)p.min_data_in_leaf * 2;
            for (int it = 0; it < NE; ++it) for (int k = 0; k < K; ++k) { if (root_done && !use_bagging) root_rows += n_train; stats->trees += 1; }
            if (use_bagging) root_rows = 0;   // bag sizes vary; not tracked separately
            // row-sharded: the device counter sums GLOBAL child counts on every rank; report this rank's share
            const int64_t rows_acc = dp ? (int64_t)(h_statrows / (unsigned long long)g_comm.nranks) : (int64_t)h_statrows;
            if (dp) root_rows /= g_comm.nranks;
            stats->hist_rows = rows_acc; stats->root_rows = root_rows;
            stats->root_atomics_per_row = 2 * (int64_t)(joint_root ? lcj.F : F); stats->level_atomics_per_row = 2 * (int64_t)F;
            // algorithmic bytes: F bin bytes + 8 B (g,h) per accumulated row
            stats->hist_bytes = rows_acc * ((int64_t)F + 8);
            (void)hipEventDestroy(ev_begin); (void)hipEventDestroy(ev_end);
            return guard.release();
        }
        // rows scanned: root = N per class tree; otherwise the smaller child of every split, which the
        // trees record (leaf_count of both children is known at the end only for final leaves), so
        // replay the growth order on the host from the stored counts.
        int64_t rows = 0, root_rows = 0, bytes = 0;
        for (int it = 0; it < NE; ++it) for (int k = 0; k < K; ++k) {
            const size_t t = (size_t)it * K + k;
            const int L = hL[t];
            const bool root_done = n_train >= (int64_t)p.min_data_in_leaf * 2;
            if (root_done) { rows += N; root_rows += N; bytes += N * ((int64_t)nchunk * 16 + 8); }
            // subtree sizes by unfolding the splits in creation order
            std::vector<int64_t> cntl(NL, 0); std::vector<int> depth(NL, 0);
            cntl[0] = n_train;
            std::vector<int64_t> fin(hcnt.begin() + t * NL, hcnt.begin() + t * NL + NL);
            // final leaf counts -> internal counts: node j created leaf j+1 from some leaf; recover via children links
            std::vector<int64_t> nodecnt(std::max(L - 1, 1), 0);
            for (int j = L - 2; j >= 0; --j) {
                auto sub = [&](int ch) -> int64_t { return ch < 0 ? fin[~ch] : nodecnt[ch]; };
                nodecnt[j] = sub(hleft[t * (NL - 1) + j]) + sub(hright[t * (NL - 1) + j]);
            }
            std::vector<int> node_depth(std::max(L - 1, 1), 0);
            for (int j = 0; j < L - 1; ++j) {
                int lc = hleft[t * (NL - 1) + j], rc = hright[t * (NL - 1) + j];
                if (lc >= 0) node_depth[lc] = node_depth[j] + 1;
                if (rc >= 0) node_depth[rc] = node_depth[j] + 1;
                auto sub = [&](int ch) -> int64_t { return ch < 0 ? fin[~ch] : nodecnt[ch]; };
                const int64_t a = sub(lc), b = sub(rc);
                const bool last = (j + 2 >= NL);
                bool go = !last;
                if (p.max_depth > 0 && node_depth[j] + 1 >= p.max_depth) go = false;
                if (a < (int64_t)p.min_data_in_leaf * 2 && b < (int64_t)p.min_data_in_leaf * 2) go = false;
                if (go) { int64_t sm = std::min(a, b); rows += sm; bytes += sm * ((int64_t)nchunk * 16 + 8 + 4); }
            }
            stats->trees += 1;
        }
        stats->hist_rows = rows; stats->root_rows = root_rows;
        // algorithmic bytes use the USEFUL feature bytes (F), not the padded record
        stats->hist_bytes = rows * ((int64_t)F + 8) + (rows - root_rows) * 4;
        (void)bytes;
    }
    (void)hipEventDestroy(ev_begin); (void)hipEventDestroy(ev_end);
    return guard.release();
}


struct PredictScratch;
void predict_device(rgbm_model* m, int device, hipStream_t s, const int32_t* d_codes, long long Ntab, long long row0, long long n,
                    const int32_t* d_feat_cols, double* d_proba, int32_t* d_label, double* d_top, PredictScratch* scratch);

// ---------------------------------------------------------------------------------------------
// The batched small-table trainer (rgbm_small.h): many fits, three launches per boosting iteration for all of them.
// Replaces the loop over fits of python/repair/train.py:158-209 (cross_val_score inside the hyper-parameter search: folds x
// trials) and python/repair/model.py:768-815 on <= 10 000-row samples.  Every fit is an rgbm_table_train call in all but time:
// the models are the same bytes.
// ---------------------------------------------------------------------------------------------
struct SmallFitDev {   // device state of one fit of a batch
    FitHost h; int F = 0, K = 1, NL = 0, NE = 0; size_t NT = 0; bool bag = false; long long bag_nrb = 1;
    DevBuf<int32_t> cols, ncod; DevBuf<long long> cnt_off, lut_off; DevBuf<unsigned int> cnt, counter;
    DevBuf<rg::FeatMeta> fmeta; DevBuf<rg::ChunkMeta> cmeta; DevBuf<uint8_t> lut, miss, used, inbag;
    DevBuf<uint4> rec; DevBuf<float2> gh; DevBuf<double> score, init, upd, cw, yv;
    DevBuf<int32_t> idx0, idx1, base, tree_L, any, sorted_rows, oob; DevBuf<rg::HistBin> pool;
    DevBuf<unsigned int> blk, rand, bagcnt;
    DevBuf<rg::ScanLane> scan_map; std::vector<rg::ScanLane> h_scan; int scan_waves = 0;
    DevBuf<uint4> vrec; DevBuf<double> vscore, vtop; DevBuf<int32_t> vlabel; DevBuf<uint8_t> vlut, vmiss; std::vector<uint8_t> h_vlut, h_vmiss; long long n_valid = 0;
    DevBuf<int32_t> t_L, t_feat, t_theta, t_dleft, t_left, t_right, t_cnt; DevBuf<double> t_gain, t_val;
    std::vector<int32_t> hL, hfeat, htheta, hdleft, hleft, hright, hcnt, hany; std::vector<double> hgain, hval;
    std::vector<uint8_t> h_used; std::vector<unsigned int> h_rand;      // host sources of asynchronous uploads: alive until the batch has drained
};

// validation rows of a fit that trained outside the fused kernels: the predictor on its model (same labels / values)
void score_valid_with_model(const rgbm_fit_spec& sp, rgbm_model* m) {
    if (!sp.valid_table || !m || sp.valid_table->n <= 0 || (!sp.valid_label_out && !sp.valid_value_out)) return;
    const rgbm_table& vt = *sp.valid_table;
    StreamGuard sg_; hipStream_t s = sg_.s;
    DevBuf<int32_t> d_fc(sp.n_features), d_lab(vt.n); DevBuf<double> d_top(vt.n);
    d_fc.upload(sp.feat_cols, sp.n_features, s);
    predict_device(m, vt.device, s, vt.codes.p, vt.n, 0, vt.n, d_fc.p, nullptr, d_lab.p, d_top.p, nullptr);
    if (sp.valid_label_out) d_lab.download(sp.valid_label_out, vt.n, s);
    if (sp.valid_value_out) d_top.download(sp.valid_value_out, vt.n, s);
    HIPCHK(hipStreamSynchronize(s));
}

bool small_fit_eligible(const rgbm_table& tab, int32_t F, const rgbm_params& p, long long small_rows) {
    if (tab.n > small_rows || tab.n >= (1ll << 31) - 4096) return false;
    if (tab.has_mult) return false;      // (rows with multiplicities: the level grower of the single-fit trainer)
    if (p.num_leaves > rg::SM_MAX_LEAVES || F > rg::SM_MAX_FEATS) return false;
    if (p.reserved & RGBM_FLAG_ROW_SHARDED) return false;
    return true;
}

// status[i] = RGBM_OK or the error code of fit i (its message is the thread's last error when exactly one fit fails; the Python
// binding raises per fit).  Fits the fused kernel does not cover (large tables, > 256 leaves) run through train_core one by one.
// page-locked host block for the tree harvest of a batch of fits (see phase E of train_batch_small)
struct HarvestArena {
    std::mutex mu; void* p = nullptr; size_t cap = 0; int device = -1;
    static constexpr size_t KEEP_MAX = (size_t)1 << 30;
    // a block of at least `bytes` (null: not available -- the caller falls back to pageable memory)
    void* get(size_t bytes, int dev) {
        if (bytes == 0 || bytes > KEEP_MAX || getenv("RGBM_NO_PIN") != nullptr) return nullptr;
        if (p && cap >= bytes && device == dev) return p;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = std::min(KEEP_MAX, bytes + bytes / 4);
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return nullptr; }
        cap = want; device = dev;
        return p;
    }
};
HarvestArena& harvest_arena() { static HarvestArena* a = new HarvestArena(); return *a; }

void train_batch_small(const rgbm_fit_spec* specs, int32_t n_fits, rgbm_model** out, int32_t* status) {
    using namespace rg;
    long long small_rows = 65536;
    if (const char* e = getenv("RGBM_SMALL_ROWS")) small_rows = atoll(e);
    const int device = specs[0].table->device;
    std::vector<int> batch;                                           // fits that go through the fused kernels
    auto record_error = [&](int i, const std::exception& e, int code) { status[i] = code; rgh::last_error() = e.what(); };
    auto run_single = [&](int i) {
        const rgbm_fit_spec& sp = specs[i];
        try {
            out[i] = train_core(*sp.table, sp.target_col, sp.feat_cols, sp.n_features, sp.y_value, sp.class_weight, nullptr, nullptr, *sp.params, nullptr); status[i] = RGBM_OK;
            score_valid_with_model(sp, out[i]);
        }
        catch (const std::invalid_argument& e) { record_error(i, e, RGBM_ERR_PARAM); }
        catch (const std::out_of_range& e) { record_error(i, e, RGBM_ERR_LABEL); }
        catch (const std::domain_error& e) { record_error(i, e, RGBM_ERR_NO_DEVICE); }
        catch (const std::exception& e) { record_error(i, e, RGBM_ERR_HIP); }
    };
    for (int i = 0; i < n_fits; ++i) {
        out[i] = nullptr; status[i] = RGBM_OK;
        const rgbm_fit_spec& sp = specs[i];
        if (!sp.table || !sp.feat_cols || !sp.params) { status[i] = RGBM_ERR_ARG; rgh::last_error() = "rgbm_table_train_batch: bad fit spec"; continue; }
        if (sp.table->device != device) throw std::invalid_argument("rgbm_table_train_batch: all tables of a batch must live on one device");
        if (small_fit_eligible(*sp.table, sp.n_features, *sp.params, small_rows)) batch.push_back(i); else run_single(i);
    }
    if (batch.empty()) return;
    if (batch.size() == 1 && !getenv("RGBM_SMALL_ALWAYS")) { run_single(batch[0]); return; }    // one fit: the level grower's kernel chain is faster than one workgroup per class tree

    StreamGuard sg_; hipStream_t s = sg_.s;
    const bool timing = getenv("RGBM_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    std::vector<std::unique_ptr<SmallFitDev>> dev(n_fits);
    // ---- A. argument checks + code frequencies of every fit's training rows (one launch per fit, ONE synchronisation for all)
    std::vector<int> live;
    for (int i : batch) {
        const rgbm_fit_spec& sp = specs[i]; const rgbm_table& tab = *sp.table; const rgbm_params& p = *sp.params; const int F = sp.n_features;
        try {
            check_params(p);
            if (F <= 0) throw std::invalid_argument("no feature columns");
            if (sp.target_col < 0 || sp.target_col >= tab.c) throw std::invalid_argument("target column out of range");
            for (int f = 0; f < F; ++f) if (sp.feat_cols[f] < 0 || sp.feat_cols[f] >= tab.c) throw std::invalid_argument("feature column out of range");
            const int obj = p.objective, n_y = tab.n_codes[sp.target_col];
            if (obj == 1 && n_y > p.num_class) throw std::out_of_range("target has more label codes than num_class");
            if (obj == 0 && n_y > 2) throw std::out_of_range("binary objective with more than 2 label codes");
            if (obj == 2 && !sp.y_value) throw std::out_of_range("regression needs the y_value dictionary");
        }
        catch (const std::invalid_argument& e) { record_error(i, e, RGBM_ERR_PARAM); continue; }
        catch (const std::out_of_range& e) { record_error(i, e, RGBM_ERR_LABEL); continue; }
        catch (const std::exception& e) { record_error(i, e, RGBM_ERR_HIP); continue; }
        dev[i].reset(new SmallFitDev());
        SmallFitDev& d = *dev[i]; FitHost& h = d.h;
        d.F = F; d.K = p.objective == 1 ? p.num_class : 1; d.NL = p.num_leaves; d.NE = p.n_estimators; d.NT = (size_t)d.NE * d.K;
        h.cols.assign(sp.feat_cols, sp.feat_cols + F); h.cols.push_back(sp.target_col);
        h.ncod.resize(F + 1); h.cnt_off.assign(F + 2, 0);
        for (int f = 0; f <= F; ++f) { h.ncod[f] = tab.n_codes[h.cols[f]]; h.cnt_off[f + 1] = h.cnt_off[f] + std::max(h.ncod[f], 1); }
        d.cols.alloc(F + 1); d.ncod.alloc(F + 1); d.cnt_off.alloc(F + 2); d.cnt.alloc(h.cnt_off[F + 1]);
        d.cols.upload(h.cols.data(), F + 1, s); d.ncod.upload(h.ncod.data(), F + 1, s); d.cnt_off.upload(h.cnt_off.data(), F + 2, s); d.cnt.zero(s);
        const int32_t* d_ycol = tab.codes.p + (long long)sp.target_col * tab.n;
        const int gx = (int)std::min<int64_t>((tab.n + 255) / 256, 1024);
        hipLaunchKernelGGL(k_count_codes, dim3(gx, F + 1), dim3(256), 0, s, tab.codes.p, (long long)tab.n, d_ycol, d.cols.p, d.ncod.p, d.cnt_off.p, d.cnt.p);
        h.cnt.resize(h.cnt_off[F + 1]);
        d.cnt.download(h.cnt.data(), h.cnt.size(), s);
        live.push_back(i);
    }
    HIPCHK(hipStreamSynchronize(s));
    // ---- B. per fit: bins, tables, records, state
    std::vector<int> ok;
    int NE_max = 0; size_t lds_max = 0; long long N_max = 1, ntrain_max = 1, nrb_max = 1, nvalid_max = 0; bool any_bag = false;
    for (int i : live) {
        const rgbm_fit_spec& sp = specs[i]; const rgbm_table& tab = *sp.table; const rgbm_params& p = *sp.params;
        SmallFitDev& d = *dev[i]; FitHost& h = d.h; const int F = d.F, K = d.K, NL = d.NL, NE = d.NE; const long long N = tab.n;
        // one failing fit does not fail the batch: whatever this fit's set-up throws (arguments, labels, HIP) is ITS status; uploads queued from
        // its host vectors are drained before they are destroyed
        auto drop_fit = [&](const std::exception& e, int code) { (void)hipStreamSynchronize(s); record_error(i, e, code); dev[i].reset(); };
        try {
        fit_setup(tab, sp.y_value, sp.class_weight, nullptr, p, F, h);
        // lane map of the packed threshold scan (rgbm_small.h): a lane owns 4 bins of one feature, a feature does not straddle waves;
        // used when a child's features fit half of the workgroup's waves (both children are scanned at once) and the two compact
        // histograms fit next to the rest in LDS -- otherwise one wave per feature (split_find_body)
        {
            std::vector<ScanLane> sm; int lanes_used = 0;
            for (int f = 0; f < F; ++f) {
                const int nl_f = std::max(1, (h.fmeta[f].V + 3) / 4);
                if (lanes_used % 64 + nl_f > 64) while (lanes_used % 64) { sm.push_back(ScanLane{-1, 0, 1, 0}); ++lanes_used; }
                const int first = lanes_used % 64;
                for (int q = 0; q < nl_f; ++q) { sm.push_back(ScanLane{f, 4 * q, q == 0 ? 1 : 0, first + nl_f - 1}); ++lanes_used; }
            }
            while (lanes_used % 64) { sm.push_back(ScanLane{-1, 0, 1, 0}); ++lanes_used; }
            const int Wn = lanes_used / 64;
            const bool off_switch = getenv("RGBM_SMALL_PACKED") && atoi(getenv("RGBM_SMALL_PACKED")) == 0;
            if (!off_switch && 2 * Wn <= SM_WAVES && h.tc.totbins <= 2048 && sm_lds_bytes(h.lds_hist, NL, F, h.tc.totbins) <= 150 * 1024) { d.scan_waves = Wn; d.h_scan = std::move(sm); }
        }
        const size_t lds = sm_lds_bytes(h.lds_hist, NL, F, d.scan_waves > 0 ? h.tc.totbins : 0);
        if (lds > 160 * 1024 - 2048) { dev[i].reset(); run_single(i); continue; }          // (histogram working set of one chunk + leaves exceed the LDS)
        const int nchunk = h.nchunk; const long long n_train = h.n_train;
        d.fmeta.alloc(F); d.cmeta.alloc(nchunk); d.lut_off.alloc(F + 1); d.lut.alloc(std::max<size_t>(h.lut.size(), 1)); d.miss.alloc(F);
        d.fmeta.upload(h.fmeta.data(), F, s); d.cmeta.upload(h.cmeta.data(), nchunk, s); d.lut_off.upload(h.lut_off.data(), F + 1, s);
        d.lut.upload(h.lut.data(), h.lut.size(), s); d.miss.upload(h.miss.data(), F, s);
        d.rec.alloc((size_t)nchunk * N);
        hipLaunchKernelGGL(k_pack_bins, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, tab.codes.p, N, 0ll, N, d.cols.p, d.ncod.p, d.lut_off.p, d.lut.p, d.miss.p, F, nchunk, d.rec.p);
        const int32_t* d_ycol = tab.codes.p + (long long)sp.target_col * N;
        d.bag = p.bagging_freq > 0 && p.bagging_fraction < 1.0;
        d.counter.alloc(1); d.counter.zero(s);
        d.base.alloc(n_train);
        hipLaunchKernelGGL(k_iota_train, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d_ycol, N, d.base.p, d.counter.p);
        d.gh.alloc((size_t)K * N); d.gh.zero(s);
        d.score.alloc((size_t)K * N); d.init.alloc(K); d.init.upload(h.init.data(), K, s);
        hipLaunchKernelGGL(k_init_score, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d.score.p, N, K, d.init.p);
        d.idx0.alloc((size_t)K * n_train); d.idx1.alloc((size_t)K * n_train);
        d.pool.alloc((size_t)K * NL * h.tc.totbins); d.upd.alloc((size_t)K * NL); d.tree_L.alloc(K); d.any.alloc(NE); d.any.zero(s);
        const size_t NT = d.NT;
        d.t_L.alloc(NT); d.t_feat.alloc(NT * (NL - 1)); d.t_theta.alloc(NT * (NL - 1)); d.t_dleft.alloc(NT * (NL - 1)); d.t_left.alloc(NT * (NL - 1)); d.t_right.alloc(NT * (NL - 1));
        d.t_cnt.alloc(NT * NL); d.t_gain.alloc(NT * (NL - 1)); d.t_val.alloc(NT * NL); d.t_cnt.zero(s); d.t_val.zero(s);
        const int n_y = h.ncod[F];
        if (sp.class_weight) { d.cw.alloc(n_y); d.cw.upload(sp.class_weight, n_y, s); }
        if (p.objective == 2) { d.yv.alloc(n_y); d.yv.upload(h.yv32.data(), n_y, s); }
        if (d.scan_waves > 0) { d.scan_map.alloc(d.h_scan.size()); d.scan_map.upload(d.h_scan.data(), d.h_scan.size(), s); }
        if (sp.valid_table && sp.valid_table->n > 0 && (sp.valid_label_out || sp.valid_value_out)) {
            // the validation rows' bin records, in the predictor's convention: a NULL cell, a code outside the dictionary and a category
            // no training row held (Feat::unseen) are MISSING = bin 255 (device_model's lookup table)
            const rgbm_table& vt = *sp.valid_table;
            if (vt.c != tab.c) throw std::invalid_argument("rgbm_table_train_batch: valid_table must have the columns of the training table");
            d.n_valid = vt.n;
            d.h_vlut.assign(std::max<size_t>(h.lut.size(), 1), 0); d.h_vmiss.assign(F, 255);
            for (int f = 0; f < F; ++f) {
                const Feat& ft = h.model->feats[f]; int b = 0;
                for (int cde = 0; cde < h.ncod[f]; ++cde) { while (b < ft.V - 1 && cde > ft.ub[b]) ++b; d.h_vlut[h.lut_off[f] + cde] = ft.is_unseen(cde) ? (uint8_t)255 : (uint8_t)(ft.V > 0 ? b : 0); }
            }
            d.vlut.alloc(d.h_vlut.size()); d.vlut.upload(d.h_vlut.data(), d.h_vlut.size(), s); d.vmiss.alloc(F); d.vmiss.upload(d.h_vmiss.data(), F, s);
            d.vrec.alloc((size_t)nchunk * vt.n);
            hipLaunchKernelGGL(k_pack_bins, dim3((unsigned)((vt.n + 255) / 256)), dim3(256), 0, s, vt.codes.p, (long long)vt.n, 0ll, (long long)vt.n, d.cols.p, d.ncod.p, d.lut_off.p, d.vlut.p, d.vmiss.p, F, nchunk, d.vrec.p);
            d.vscore.alloc((size_t)K * vt.n); d.vlabel.alloc(vt.n); d.vtop.alloc(vt.n);
            hipLaunchKernelGGL(k_init_score, dim3((unsigned)((vt.n + 255) / 256)), dim3(256), 0, s, d.vscore.p, (long long)vt.n, K, d.init.p);
            nvalid_max = std::max<long long>(nvalid_max, vt.n);
        }
        d.h_used = make_used_masks(p, NT, F, h.trivial);
        d.used.alloc(d.h_used.size()); d.used.upload(d.h_used.data(), d.h_used.size(), s);
        if (d.bag) {   // GBDT::Bagging state: stable training-row order, one LCG per 1024 positions
            const long long nblk = (N + 1023) / 1024;
            d.blk.alloc(nblk); d.sorted_rows.alloc(n_train); d.oob.alloc(n_train); d.inbag.alloc(N); d.bagcnt.alloc(2);
            hipLaunchKernelGGL(k_block_count, dim3((unsigned)nblk), dim3(256), 0, s, d_ycol, N, d.blk.p);
            hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, s, d.blk.p, nblk);
            hipLaunchKernelGGL(k_stable_compact, dim3((unsigned)nblk), dim3(256), 0, s, d_ycol, N, d.blk.p, d.sorted_rows.p);
            d.bag_nrb = std::max<long long>(1, (n_train + 1023) / 1024);
            LgbRand sr2((uint32_t)p.seed); sr2.rnd16();
            const int bagging_seed = sr2.rnd16();
            d.h_rand.resize(d.bag_nrb);
            for (long long b = 0; b < d.bag_nrb; ++b) d.h_rand[b] = (unsigned int)(bagging_seed + b);
            d.rand.alloc(d.bag_nrb); d.rand.upload(d.h_rand.data(), d.bag_nrb, s);
            any_bag = true; nrb_max = std::max(nrb_max, d.bag_nrb);
        }
        NE_max = std::max(NE_max, NE); lds_max = std::max(lds_max, lds); N_max = std::max(N_max, N); ntrain_max = std::max(ntrain_max, n_train);
        ok.push_back(i);
        }
        catch (const std::invalid_argument& e) { drop_fit(e, RGBM_ERR_PARAM); continue; }
        catch (const std::out_of_range& e) { drop_fit(e, RGBM_ERR_LABEL); continue; }
        catch (const std::domain_error& e) { drop_fit(e, RGBM_ERR_NO_DEVICE); continue; }
        catch (const std::exception& e) { drop_fit(e, RGBM_ERR_HIP); continue; }
    }
    if (ok.empty()) return;
    if (timing) HIPCHK(hipStreamSynchronize(s));
    const double t_setup = now();
    // ---- C. descriptors
    std::vector<SmallFit> fits(ok.size()); std::vector<int32_t> tree2fit;
    for (size_t j = 0; j < ok.size(); ++j) {
        const int i = ok[j]; const rgbm_fit_spec& sp = specs[i]; const rgbm_params& p = *sp.params; SmallFitDev& d = *dev[i]; FitHost& h = d.h;
        SmallFit& f = fits[j]; memset(&f, 0, sizeof(f));
        f.c = h.tc; f.rec = d.rec.p; f.ycol = sp.table->codes.p + (long long)sp.target_col * sp.table->n; f.gh = d.gh.p; f.score = d.score.p;
        f.idx0 = d.idx0.p; f.idx1 = d.idx1.p; f.base_idx = d.base.p; f.pool = d.pool.p; f.fmeta = d.fmeta.p; f.cmeta = d.cmeta.p; f.used = d.used.p;
        f.out = TreeOut{d.t_L.p, d.t_feat.p, d.t_theta.p, d.t_dleft.p, d.t_left.p, d.t_right.p, d.t_gain.p, d.t_val.p, d.t_cnt.p};
        f.init = d.init.p; f.upd = d.upd.p; f.tree_L = d.tree_L.p; f.any_split = d.any.p;
        f.class_w = sp.class_weight ? d.cw.p : nullptr; f.y_value = p.objective == 2 ? d.yv.p : nullptr;
        f.rand_state = d.rand.p; f.sorted_rows = d.sorted_rows.p; f.inbag = d.inbag.p; f.bagcnt = d.bagcnt.p; f.oob = d.oob.p;
        f.bag_fraction = p.bagging_fraction; f.bag_nrb = d.bag_nrb; f.bag_freq = d.bag ? p.bagging_freq : 0;
        f.tree0 = (int32_t)tree2fit.size(); f.n_estimators = d.NE; f.lds_hist = (unsigned long long)h.lds_hist;
        f.scan_map = d.scan_map.p; f.scan_waves = d.scan_waves;
        f.vrec = d.vrec.p; f.vscore = d.vscore.p; f.n_valid = d.n_valid;
        for (int k = 0; k < d.K; ++k) tree2fit.push_back((int32_t)j);
    }
    DevBuf<unsigned long long> d_prof(8); d_prof.zero(s);
    for (auto& f : fits) f.prof = d_prof.p;
    DevBuf<SmallFit> d_fits(fits.size()); DevBuf<int32_t> d_t2f(tree2fit.size());
    d_fits.upload(fits.data(), fits.size(), s); d_t2f.upload(tree2fit.data(), tree2fit.size(), s);
    {
        static std::mutex attr_mu; static std::vector<char> attr_done(64, 0);
        std::lock_guard<std::mutex> lk(attr_mu);
        if (!attr_done[device & 63]) { HIPCHK(hipFuncSetAttribute((const void*)k_small_tree, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048)); attr_done[device & 63] = 1; }
    }
    if (lds_max > 160 * 1024 - 2048) throw std::invalid_argument("small-table batch: histogram working set exceeds LDS");
    // ---- D. the boosting iterations of the whole batch: enqueue only
    const unsigned nf = (unsigned)fits.size(), KT = (unsigned)tree2fit.size();
    const unsigned gx_rows = (unsigned)std::min<long long>((N_max + 255) / 256, 64), gx_train = (unsigned)std::min<long long>((ntrain_max + 255) / 256, 64);
    for (int it = 0; it < NE_max; ++it) {
        if (any_bag) {
            hipLaunchKernelGGL(k_small_bagging, dim3((unsigned)((nrb_max + 63) / 64), nf), dim3(64), 0, s, d_fits.p, it);
            hipLaunchKernelGGL(k_small_bag_lists, dim3(gx_train, nf), dim3(256), 0, s, d_fits.p, it);
        }
        hipLaunchKernelGGL(k_small_grad, dim3(gx_rows, nf), dim3(256), 0, s, d_fits.p, it);
        hipLaunchKernelGGL(k_small_tree, dim3(KT), dim3(SM_THREADS), lds_max, s, d_fits.p, d_t2f.p, it);
        if (any_bag) hipLaunchKernelGGL(k_small_oob, dim3(gx_train, KT), dim3(256), 0, s, d_fits.p, d_t2f.p, it);
        if (nvalid_max > 0) hipLaunchKernelGGL(k_small_valid, dim3((unsigned)std::min<long long>((nvalid_max + 255) / 256, 64), KT), dim3(256), 0, s, d_fits.p, d_t2f.p, it);
    }
    for (int i : ok) {   // ConvertOutput + arg-max of the validation rows (what model.predict returns for them)
        SmallFitDev& d = *dev[i]; const rgbm_fit_spec& sp = specs[i];
        if (d.n_valid <= 0) continue;
        hipLaunchKernelGGL(k_softmax_argmax, dim3((unsigned)((d.n_valid + 255) / 256)), dim3(256), 0, s, d.vscore.p, d.n_valid, sp.params->objective,
                           d.h.model->num_class, (double*)nullptr, d.vlabel.p, d.vtop.p);
        if (sp.valid_label_out) d.vlabel.download(sp.valid_label_out, d.n_valid, s);
        if (sp.valid_value_out) d.vtop.download(sp.valid_value_out, d.n_valid, s);
    }
    HIPCHK(hipGetLastError());
    const double t_enq = now();
    if (timing) HIPCHK(hipStreamSynchronize(s));
    const double t_iter = now();
    // ---- E. trees back to the host, one model per fit
    std::vector<int> want;                                            // fits whose model is wanted (a CV fold is wanted for its scores only)
    for (int i : ok) if (!((specs[i].params->reserved & RGBM_FLAG_NO_MODEL) && dev[i]->n_valid > 0)) want.push_back(i);
    // The trees of a batch are tens of MB (72 000 trees, 87 MB for the 16 fits of the 10M x 16 job): into fresh pageable vectors that
    // was zero-filling + page faults + 160 staged copies = most of 69 ms.  They land in ONE page-locked block kept for the life of the
    // process (grown on demand, at most 1 GiB kept; a second batch running at the same time uses pageable vectors).
    std::vector<HostTrees> hts(n_fits);
    HarvestArena& HA = harvest_arena();
    std::unique_lock<std::mutex> ha_lock(HA.mu, std::try_to_lock);
    {
        size_t need = 0;
        auto take = [&](size_t bytes) { const size_t o = need; need += (bytes + 63) & ~(size_t)63; return o; };
        struct Off { size_t L, feat, theta, dleft, left, right, cnt, gain, val, any; };
        std::vector<Off> offs(n_fits);
        for (int i : want) {
            SmallFitDev& d = *dev[i]; const size_t NT = d.NT, nn = NT * (size_t)(d.NL - 1), nl = NT * (size_t)d.NL;
            Off& o = offs[i];
            o.L = take(NT * 4); o.feat = take(nn * 4); o.theta = take(nn * 4); o.dleft = take(nn * 4); o.left = take(nn * 4); o.right = take(nn * 4);
            o.cnt = take(nl * 4); o.gain = take(nn * 8); o.val = take(nl * 8); o.any = take((size_t)d.NE * 4);
        }
        char* base = ha_lock.owns_lock() ? (char*)HA.get(need, device) : nullptr;
        for (int i : want) {
            SmallFitDev& d = *dev[i]; const size_t NT = d.NT, nn = NT * (size_t)(d.NL - 1), nl = NT * (size_t)d.NL;
            int32_t *hL, *hfeat, *htheta, *hdleft, *hleft, *hright, *hcnt, *hany; double *hgain, *hval;
            if (base) {
                const Off& o = offs[i];
                hL = (int32_t*)(base + o.L); hfeat = (int32_t*)(base + o.feat); htheta = (int32_t*)(base + o.theta); hdleft = (int32_t*)(base + o.dleft);
                hleft = (int32_t*)(base + o.left); hright = (int32_t*)(base + o.right); hcnt = (int32_t*)(base + o.cnt); hany = (int32_t*)(base + o.any);
                hgain = (double*)(base + o.gain); hval = (double*)(base + o.val);
            } else {
                d.hL.resize(NT); d.hfeat.resize(nn); d.htheta.resize(nn); d.hdleft.resize(nn); d.hleft.resize(nn); d.hright.resize(nn);
                d.hcnt.resize(nl); d.hany.resize(d.NE); d.hgain.resize(nn); d.hval.resize(nl);
                hL = d.hL.data(); hfeat = d.hfeat.data(); htheta = d.htheta.data(); hdleft = d.hdleft.data(); hleft = d.hleft.data(); hright = d.hright.data();
                hcnt = d.hcnt.data(); hany = d.hany.data(); hgain = d.hgain.data(); hval = d.hval.data();
            }
            d.t_L.download(hL, NT, s); d.t_feat.download(hfeat, nn, s); d.t_theta.download(htheta, nn, s);
            d.t_dleft.download(hdleft, nn, s); d.t_left.download(hleft, nn, s); d.t_right.download(hright, nn, s);
            d.t_cnt.download(hcnt, nl, s); d.t_gain.download(hgain, nn, s); d.t_val.download(hval, nl, s);
            d.any.download(hany, d.NE, s);
            hts[i] = HostTrees{hL, hfeat, htheta, hdleft, hleft, hright, hcnt, hgain, hval, hany};
        }
    }
    HIPCHK(hipStreamSynchronize(s));
    const double t_down = now();
    {   // the tree lists of the models: host work proportional to fits x iterations x class trees, spread over a few threads
        struct Item { int fit; size_t t0, t1; };
        std::vector<Item> items;
        for (int i : want) {
            SmallFitDev& d = *dev[i];
            const size_t n = model_trees_begin(d.h.model.get(), hts[i], d.NE, d.K);
            for (size_t t0 = 0; t0 < n; t0 += 512) items.push_back(Item{i, t0, std::min(n, t0 + 512)});
        }
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t j = next.fetch_add(1); j < items.size(); j = next.fetch_add(1))
                model_trees_fill(dev[items[j].fit]->h.model.get(), hts[items[j].fit], dev[items[j].fit]->NL, items[j].t0, items[j].t1);
        };
        const size_t nth = std::min<size_t>(std::min<size_t>(std::max<size_t>(items.size(), 1), 32), std::max(1u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (size_t q = 1; q < nth; ++q) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
    }
    for (int i : ok) status[i] = RGBM_OK;
    for (int i : want) out[i] = dev[i]->h.model.release();
#if defined(SM_PROF)
    { unsigned long long hp[8]; d_prof.download(hp, 8, s); HIPCHK(hipStreamSynchronize(s)); unsigned long long tot = 0; for (int q = 0; q < 8; ++q) tot += hp[q];
      fprintf(stderr, "[rgbm] k_small_tree phases (%% of thread-0 cycles): zero %.1f | accumulate %.1f | flush %.1f | split_find %.1f | reduce+pick %.1f | partition %.1f | finish %.1f | score %.1f\n",
              100.0 * hp[0] / tot, 100.0 * hp[1] / tot, 100.0 * hp[2] / tot, 100.0 * hp[3] / tot, 100.0 * hp[4] / tot, 100.0 * hp[5] / tot, 100.0 * hp[6] / tot, 100.0 * hp[7] / tot); }
#endif
    if (timing) fprintf(stderr, "[rgbm] batch of %zu fits, %u class trees, %d iterations: setup %.1f ms, enqueue %.1f ms, iterations drained after %.1f ms, download %.1f ms (%s), models %.1f ms\n",
                        ok.size(), KT, NE_max, t_setup - t_start, t_enq - t_setup, t_iter - t_setup, t_down - t_iter, ha_lock.owns_lock() ? "page-locked block" : "pageable", now() - t_down);
}

// ---------------------------------------------------------------------------------------------
// Predictor mirror
// ---------------------------------------------------------------------------------------------
// run `fn(t0, t1)` over [0, n) in ranges, on a few host threads when there is enough of it (the table builders below: a
// 19 200-tree model is ~20 ms of host work on one thread)
template <class Fn> void parallel_ranges(size_t n, size_t min_per_thread, Fn fn) {
    const size_t nth = std::min<size_t>(std::min<size_t>(8, std::max<size_t>(1, n / std::max<size_t>(min_per_thread, 1))), std::max(1u, std::thread::hardware_concurrency()));
    if (nth <= 1) { fn((size_t)0, n); return; }
    std::exception_ptr err; std::mutex emu; std::vector<std::thread> th;
    for (size_t q = 0; q < nth; ++q)
        th.emplace_back([&, q]() { try { fn(n * q / nth, n * (q + 1) / nth); } catch (...) { std::lock_guard<std::mutex> lk(emu); if (!err) err = std::current_exception(); } });
    for (auto& t : th) t.join();
    if (err) std::rethrow_exception(err);
}

// the index-linked node tables of the tree walk (k_predict_raw): built only for models the bit-vector scorer cannot take, or when
// a test asks for the walk (the caller holds m->mu)
void build_walk_tables(rgbm_model* m, DeviceModel* dm, hipStream_t s) {
    using namespace rg;
    if (dm->nodes.p) return;
    int maxL = 2;
    for (auto& t : m->trees) maxL = std::max(maxL, t.L);
    dm->node_stride = maxL - 1; dm->leaf_stride = maxL;
    const size_t NT = m->trees.size();
    std::vector<PNode> nodes(std::max<size_t>(NT, 1) * dm->node_stride); std::vector<double> lv(std::max<size_t>(NT, 1) * dm->leaf_stride, 0.0);
    parallel_ranges(NT, 2048, [&](size_t t0, size_t t1) {
        for (size_t t = t0; t < t1; ++t) {
            const Tree& tr = m->trees[t];
            PNode* nd = nodes.data() + t * dm->node_stride;
            if (tr.L <= 1) { nd[0].w0 = 0; nd[0].w1 = 0xFFFFFFFFu; /* both children = ~0 */ }
            for (int j = 0; j < tr.L - 1; ++j) {
                nd[j].w0 = (uint32_t)(tr.feat[j] & 0xFFFF) | ((uint32_t)(tr.theta[j] + 1) & 0x1FF) << 16 | (uint32_t)(tr.dleft[j] ? 1 : 0) << 25;
                nd[j].w1 = ((uint32_t)tr.left[j] & 0xFFFFu) | ((uint32_t)tr.right[j] & 0xFFFFu) << 16;
            }
            for (int l = 0; l < tr.L; ++l) lv[t * dm->leaf_stride + l] = tr.leaf_value[l];
        }
    });
    dm->nodes.alloc(nodes.size()); dm->nodes.upload(nodes.data(), nodes.size(), s);
    dm->leaf_value.alloc(lv.size()); dm->leaf_value.upload(lv.data(), lv.size(), s);
    HIPCHK(hipStreamSynchronize(s));     // the vectors are locals
}

DeviceModel* device_model(rgbm_model* m, int device, hipStream_t s) {
    std::lock_guard<std::mutex> lk(m->mu);
    auto it = m->dev.find(device);
    if (it != m->dev.end()) return it->second;
    using namespace rg;
    auto dm = new DeviceModel();
    std::unique_ptr<DeviceModel> guard(dm);
    const size_t NT = m->trees.size();
    const int F = m->F;
    std::vector<long long> lut_off(F + 1, 0); std::vector<int32_t> ncod(F), ident(F); std::vector<uint8_t> miss(F, 255);
    for (int f = 0; f < F; ++f) { ncod[f] = m->feats[f].n_codes; ident[f] = f; lut_off[f + 1] = lut_off[f] + std::max(ncod[f], 1); }
    std::vector<uint8_t> lut(lut_off[F]);
    for (int f = 0; f < F; ++f) {
        const Feat& ft = m->feats[f]; int b = 0;
        for (int c = 0; c < ncod[f]; ++c) { while (b < ft.V - 1 && c > ft.ub[b]) ++b; lut[lut_off[f] + c] = ft.is_unseen(c) ? (uint8_t)255 : (uint8_t)(ft.V > 0 ? b : 0); }
    }
    // the bit-vector scoring tables (k_predict_qs) when no tree has more than 64 leaves and the model has at most 32 features
    {
        int maxLeaves = 1;
        for (const Tree& tr : m->trees) maxLeaves = std::max(maxLeaves, tr.L);
        const int F_ = m->F;
        if (maxLeaves <= 64 && F_ <= 32 && NT > 0) {
            const int MW = maxLeaves <= 32 ? 1 : 2, LP = 32 * MW;
            std::vector<int32_t> foff(F_ + 1, 0);
            for (int f = 0; f < F_; ++f) foff[f + 1] = foff[f] + std::max(m->feats[f].V, 1) + 1;       // value bins + the missing entry
            const int S = foff[F_];
            // (uninitialised blocks, filled by the thread that owns the range: a 20 MB vector constructor is 5 ms of page faults on one thread)
            const size_t mk_n = (size_t)NT * S * MW, lv2_n = (size_t)NT * LP;
            std::unique_ptr<uint32_t[]> mk_(new uint32_t[mk_n]); std::unique_ptr<double[]> lv2_(new double[lv2_n]);
            uint32_t* const mk = mk_.get(); double* const lv2 = lv2_.get();
            std::vector<uint32_t> usedf(NT, 0u);
            parallel_ranges(NT, 2048, [&](size_t tr0, size_t tr1) {
            memset(mk + tr0 * S * MW, 0xFF, (tr1 - tr0) * S * MW * sizeof(uint32_t));
            std::fill(lv2 + tr0 * LP, lv2 + tr1 * LP, 0.0);
            std::vector<int> lo, mid;                                // per internal node: in-order ids [lo, mid) of the leaves of its left subtree
            std::vector<std::pair<int, int>> st;
            for (size_t t = tr0; t < tr1; ++t) {
                const Tree& tr = m->trees[t];
                uint32_t* tm = mk + t * S * MW; double* tl = lv2 + t * LP;
                if (tr.L <= 1) { tl[0] = tr.leaf_value ? tr.leaf_value[0] : 0.0; continue; }
                lo.assign(tr.L - 1, 0); mid.assign(tr.L - 1, 0);
                // iterative in-order traversal: (ref, state) with state 0 = enter, 1 = left subtree done
                int next_leaf = 0;
                st.clear(); st.emplace_back(0, 0);
                while (!st.empty()) {
                    auto& top = st.back();
                    const int ref = top.first;
                    if (ref < 0) { tl[next_leaf++] = tr.leaf_value[~ref]; st.pop_back(); continue; }
                    if (top.second == 0) { top.second = 1; lo[ref] = next_leaf; st.emplace_back(tr.left[ref], 0); }
                    else { mid[ref] = next_leaf; const int r = tr.right[ref]; st.pop_back(); st.emplace_back(r, 0); }
                }
                for (int j = 0; j < tr.L - 1; ++j) {
                    // mask of node j: zeros for the leaves of its left subtree
                    uint32_t w[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
                    for (int b = lo[j]; b < mid[j]; ++b) w[b >> 5] &= ~(1u << (b & 31));
                    const int f = tr.feat[j], nb = foff[f + 1] - foff[f] - 1;
                    usedf[t] |= 1u << f;
                    // the test of node j is FALSE (the row goes right) for bins above its threshold, and for a missing value unless it defaults left
                    for (int b = std::max(tr.theta[j] + 1, 0); b < nb; ++b) for (int q = 0; q < MW; ++q) tm[(size_t)(foff[f] + b) * MW + q] &= w[q];
                    if (!tr.dleft[j]) for (int q = 0; q < MW; ++q) tm[(size_t)(foff[f] + nb) * MW + q] &= w[q];
                }
            }
            });
            dm->qs_masks.alloc(mk_n); dm->qs_masks.upload(mk, mk_n, s);
            dm->qs_leaves.alloc(lv2_n); dm->qs_leaves.upload(lv2, lv2_n, s);
            dm->qs_used.alloc(usedf.size()); dm->qs_used.upload(usedf.data(), usedf.size(), s);
            dm->qs_foff.alloc(foff.size()); dm->qs_foff.upload(foff.data(), foff.size(), s);
            dm->qs_S = S; dm->qs_MW = MW;
            HIPCHK(hipStreamSynchronize(s));     // the vectors are locals
        }
    }
    if (dm->qs_MW == 0) build_walk_tables(m, dm, s);
    dm->lut.alloc(lut.size()); dm->lut.upload(lut.data(), lut.size(), s);
    dm->lut_off.alloc(F + 1); dm->lut_off.upload(lut_off.data(), F + 1, s);
    dm->n_codes.alloc(F); dm->n_codes.upload(ncod.data(), F, s);
    dm->miss.alloc(F); dm->miss.upload(miss.data(), F, s);
    dm->ident.alloc(F); dm->ident.upload(ident.data(), F, s);
    HIPCHK(hipStreamSynchronize(s));
    m->dev[device] = dm;
    return guard.release();
}

// score rows [row0, row0+n) of device codes (column-major, Ntab rows per column) with model m.
// feat_col_dev: device array of the F column indices.  Outputs are device pointers (any may be null).
// scratch of one scoring call; the chain allocates it once for all its models (hipMalloc of hundreds of MB per target was
// most of the repair time)
struct PredictScratch { DevBuf<uint4> rec; DevBuf<double> raw; };

void predict_device(rgbm_model* m, int device, hipStream_t s, const int32_t* d_codes, long long Ntab, long long row0, long long n,
                    const int32_t* d_feat_cols, double* d_proba, int32_t* d_label, double* d_top, PredictScratch* scratch = nullptr) {
    using namespace rg;
    if (n <= 0) return;
    DeviceModel* dm = device_model(m, device, s);
    const int F = m->F, nchunk = (F + 15) / 16, K = m->K;
    PredictScratch local;
    PredictScratch& sc = scratch ? *scratch : local;
    if (sc.rec.n < (size_t)nchunk * n) sc.rec.alloc((size_t)nchunk * n);
    if (sc.raw.n < (size_t)K * n) sc.raw.alloc((size_t)K * n);
    DevBuf<uint4>& rec = sc.rec; DevBuf<double>& raw = sc.raw;
    hipLaunchKernelGGL(k_pack_bins, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_codes, Ntab, row0, n,
                       d_feat_cols ? d_feat_cols : dm->ident.p, dm->n_codes.p, dm->lut_off.p, dm->lut.p, dm->miss.p, F, nchunk, rec.p);
    const char* pe = getenv("RGBM_PREDICTOR"); const bool force_walk = pe && strcmp(pe, "walk") == 0;   // tests: the index-linked walk
    bool qs = dm->qs_MW > 0 && !force_walk;
    int qs_tb = 8, qs_tw = 0;        // trees per LDS stage; padded mask words per tree (0: the dynamic-stride kernel)
    if (qs) {
        const int words = dm->qs_S * dm->qs_MW;
        // compile-time strides (k_predict_qs<., ., TW, TBN>) for the table sizes that occur: the tree's offset rides in the ds_read
        struct Fix { int mw, fmax, tw, tbn; };
        static const Fix fixes[] = {{1, 16, 256, 8}, {1, 16, 512, 8}, {1, 32, 512, 8}, {1, 32, 1024, 4}, {2, 16, 512, 8}, {2, 16, 1024, 4}, {2, 32, 1024, 4}, {2, 32, 2048, 2}};
        const bool no_fix = getenv("RGBM_QS_FIXED") && atoi(getenv("RGBM_QS_FIXED")) == 0;
        for (const Fix& fx : fixes)
            if (!no_fix && fx.mw == dm->qs_MW && fx.fmax == (F <= 16 ? 16 : 32) && words + dm->qs_MW <= fx.tw) { qs_tw = fx.tw; qs_tb = fx.tbn; break; }   // (+ one all-ones pad entry)
        const size_t per_tree = (size_t)(qs_tw ? qs_tw : words) * 4 + (size_t)32 * dm->qs_MW * 8 + 4;
        if (!qs_tw) while (qs_tb > 1 && 2 * qs_tb * per_tree + 16 > 48 * 1024) --qs_tb;
        if (2 * qs_tb * per_tree + 16 > 48 * 1024) qs = false;
    }
    if (!qs) { std::lock_guard<std::mutex> lk(m->mu); build_walk_tables(m, dm, s); }
    if (qs) {   // bit-vector scoring: no tree walk at all
        const size_t lds = (size_t)2 * qs_tb * ((size_t)(qs_tw ? qs_tw : dm->qs_S * dm->qs_MW) * 4 + (size_t)32 * dm->qs_MW * 8 + 4) + 16;
        const dim3 grid((unsigned)((n + 256 * QS_ROWS - 1) / (256 * QS_ROWS)), K);
        const uint8_t* r8 = reinterpret_cast<const uint8_t*>(rec.p);
#define RGBM_QS(MW, FM, TW, TBN) hipLaunchKernelGGL((k_predict_qs<MW, FM, TW, TBN>), grid, dim3(256), lds, s, r8, n, dm->qs_masks.p, dm->qs_leaves.p, dm->qs_used.p, dm->qs_foff.p, F, dm->qs_S, qs_tb, m->n_iter, K, raw.p)
        if (dm->qs_MW == 1 && F <= 16) { if (qs_tw == 256) RGBM_QS(1, 16, 256, 8); else if (qs_tw == 512) RGBM_QS(1, 16, 512, 8); else RGBM_QS(1, 16, 0, 0); }
        else if (dm->qs_MW == 1) { if (qs_tw == 512) RGBM_QS(1, 32, 512, 8); else if (qs_tw == 1024) RGBM_QS(1, 32, 1024, 4); else RGBM_QS(1, 32, 0, 0); }
        else if (F <= 16) { if (qs_tw == 512) RGBM_QS(2, 16, 512, 8); else if (qs_tw == 1024) RGBM_QS(2, 16, 1024, 4); else RGBM_QS(2, 16, 0, 0); }
        else { if (qs_tw == 1024) RGBM_QS(2, 32, 1024, 4); else if (qs_tw == 2048) RGBM_QS(2, 32, 2048, 2); else RGBM_QS(2, 32, 0, 0); }
#undef RGBM_QS
    }
    else if (nchunk == 1) hipLaunchKernelGGL(k_predict_raw<true>, dim3((unsigned)((n + 255) / 256), K), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(rec.p), n,
                                        dm->nodes.p, dm->leaf_value.p, m->n_iter, K, dm->node_stride, dm->leaf_stride, raw.p);
    else hipLaunchKernelGGL(k_predict_raw<false>, dim3((unsigned)((n + 255) / 256), K), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(rec.p), n,
                            dm->nodes.p, dm->leaf_value.p, m->n_iter, K, dm->node_stride, dm->leaf_stride, raw.p);
    hipLaunchKernelGGL(k_softmax_argmax, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, raw.p, n, m->objective, m->num_class, d_proba, d_label, d_top);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));   // rec/raw are freed on return
}


// version 2 = version 1 + the unseen-category bitmap of every feature; written only when a feature has one
int32_t blob_version(const rgbm_model& m) { for (const Feat& f : m.feats) if (!f.unseen.empty()) return 2; return 1; }

size_t serialised_size(const rgbm_model& m) {
    const int32_t ver = blob_version(m);
    size_t n = 28;
    for (const Feat& f : m.feats) n += 12 + 4 * (size_t)f.V + (ver == 2 ? 4 + 4 * f.unseen.size() : 0);
    for (const Tree& t : m.trees) n += 4 + ((size_t)t.L - 1) * 28 + (size_t)t.L * 12;
    return n;
}

// the model blob, written straight into the caller's buffer (a 19 200-tree model is 23 MB: no intermediate vector, one pass)
void serialise_into(const rgbm_model& m, uint8_t* dst) {
    uint8_t* p = dst;
    auto put = [&](const void* src, size_t n) { memcpy(p, src, n); p += n; };
    const int32_t ver = blob_version(m);
    int32_t hdr[7] = {0x4D424752, ver, m.objective, m.num_class, m.K, m.n_iter, m.F};
    put(hdr, sizeof(hdr));
    for (const Feat& f : m.feats) {
        int32_t h3[3] = {f.n_codes, f.V, f.has_nan}; put(h3, sizeof(h3)); put(f.ub.data(), 4 * (size_t)f.V);
        if (ver == 2) { int32_t nw = (int32_t)f.unseen.size(); put(&nw, 4); put(f.unseen.data(), 4 * (size_t)nw); }
    }
    for (const Tree& t : m.trees) {
        const size_t n = (size_t)t.L - 1;
        put(&t.L, 4);
        put(t.feat, 4 * n); put(t.theta, 4 * n); put(t.dleft, 4 * n);
        put(t.left, 4 * n); put(t.right, 4 * n); put(t.gain, 8 * n);
        put(t.leaf_value, 8 * (size_t)t.L); put(t.leaf_count, 4 * (size_t)t.L);
    }
}

}  // namespace

void rgh::predict_proba_device(const rgbm_model* m, int device, hipStream_t s, const int32_t* d_codes, long long n, const int32_t* d_feat_cols,
                               double* d_proba) {
    predict_device(const_cast<rgbm_model*>(m), device, s, d_codes, n, 0, n, d_feat_cols, d_proba, nullptr, nullptr);
}
void rgh::model_shape(const rgbm_model* m, int32_t* objective, int32_t* num_class, int32_t* n_features) {
    *objective = m->objective; *num_class = m->num_class; *n_features = m->F;
}

// Host -> HBM copy of a caller-owned block.  A pageable block handed to hipMemcpy is staged by the driver through one small
// bounce buffer (measured 4-7 GB/s for the 735 MB of the 10M x 16 tables); page-locking it in place (hipHostRegister) costs more
// than it saves for a one-off copy (4.4 GB/s).  So the library stages it itself: two page-locked 32 MB buffers kept for the
// life of the process, filled by a few host threads while the copy engine drains the other one.  Blocks that already are
// pinned (rgbm_host_alloc, or registered by the caller) go straight to the copy engine.  RGBM_NO_PIN=1: plain hipMemcpy.
namespace {
struct StageRing {
    std::mutex mu; void* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; hipStream_t s = nullptr; int device = -1; bool ok = false;
    static constexpr size_t CHUNK = 32u << 20;
    bool ready(int dev) {
        if (ok && device == dev) return true;
        if (ok) return false;                                        // ring belongs to another device: plain copy for this one
        if (hipHostMalloc(&buf[0], CHUNK, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&buf[1], CHUNK, hipHostMallocDefault) != hipSuccess ||
            hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        device = dev; ok = true;
        return true;
    }
};
StageRing& stage_ring() { static StageRing* r = new StageRing(); return *r; }
}  // namespace

static void upload_pinned(void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return;
    hipPointerAttribute_t attr; memset(&attr, 0, sizeof(attr));
    const bool pinned = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    int dev = 0; (void)hipGetDevice(&dev);
    StageRing& R = stage_ring();
    if (pinned || bytes < (8u << 20) || getenv("RGBM_NO_PIN") != nullptr) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return; }
    std::lock_guard<std::mutex> lk(R.mu);
    if (!R.ready(dev)) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return; }
    const int nthr = (int)std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency() / 2));
    size_t off = 0; int i = 0;
    while (off < bytes) {
        const size_t n = std::min(StageRing::CHUNK, bytes - off);
        const int b = i & 1;
        if (i >= 2) HIPCHK(hipEventSynchronize(R.ev[b]));            // the copy engine is done with this buffer
        std::vector<std::thread> th;
        const size_t per = (n + nthr - 1) / nthr;
        for (int t = 1; t < nthr; ++t) {
            const size_t o = per * t; if (o >= n) break;
            th.emplace_back([=, &R]() { memcpy((char*)R.buf[b] + o, (const char*)src + off + o, std::min(per, n - o)); });
        }
        memcpy(R.buf[b], (const char*)src + off, std::min(per, n));
        for (auto& t : th) t.join();
        HIPCHK(hipMemcpyAsync((char*)dst + off, R.buf[b], n, hipMemcpyHostToDevice, R.s));
        HIPCHK(hipEventRecord(R.ev[b], R.s));
        off += n; ++i;
    }
    HIPCHK(hipStreamSynchronize(R.s));
}

// page-locked staging of the chain's outputs (two blocks: model t's labels / probabilities leave the device and are copied into the
// caller's arrays while model t + 1 is scored); kept for the life of the process, at most 256 MiB a block
namespace {
struct ChainStage {
    std::mutex mu; void* p[2] = {nullptr, nullptr}; size_t cap = 0; int device = -1;
    static constexpr size_t KEEP_MAX = (size_t)256 << 20;
    bool get(size_t bytes, int dev) {
        if (bytes == 0 || bytes > KEEP_MAX || getenv("RGBM_NO_PIN") != nullptr) return false;
        if (p[0] && cap >= bytes && device == dev) return true;
        for (int i = 0; i < 2; ++i) if (p[i]) { (void)hipHostFree(p[i]); p[i] = nullptr; }
        cap = 0;
        const size_t want = std::min(KEEP_MAX, bytes + bytes / 8);
        if (hipHostMalloc(&p[0], want, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&p[1], want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            for (int i = 0; i < 2; ++i) if (p[i]) { (void)hipHostFree(p[i]); p[i] = nullptr; }
            return false;
        }
        cap = want; device = dev;
        return true;
    }
};
ChainStage& chain_stage() { static ChainStage* c = new ChainStage(); return *c; }
struct ThreadJoiner { std::vector<std::thread> th; ~ThreadJoiner() { for (auto& t : th) if (t.joinable()) t.join(); } };
}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {


RGBM_EXPORT int rgbm_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
RGBM_EXPORT const char* rgbm_last_error(void) { return g_err.c_str(); }
RGBM_EXPORT int rgbm_version(void) { return RGBM_VERSION; }
RGBM_EXPORT int rgbm_release_cache(void) {
    return guarded([&]() {
        rgh::pool_trim(0);
        // the page-locked blocks kept for the tree harvest of a batch and for the outputs of a repair chain (not while a call uses them)
        {
            HarvestArena& a = harvest_arena();
            std::unique_lock<std::mutex> lk(a.mu, std::try_to_lock);
            if (lk.owns_lock() && a.p) { (void)hipHostFree(a.p); a.p = nullptr; a.cap = 0; a.device = -1; }
        }
        {
            ChainStage& c = chain_stage();
            std::unique_lock<std::mutex> lk(c.mu, std::try_to_lock);
            if (lk.owns_lock()) { for (int i = 0; i < 2; ++i) if (c.p[i]) { (void)hipHostFree(c.p[i]); c.p[i] = nullptr; } c.cap = 0; c.device = -1; }
        }
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_create(const int32_t* codes, int64_t n, int32_t c, const int32_t* n_codes, int32_t device_id, rgbm_table** out) {
    if (!codes || !n_codes || !out || n <= 0 || c <= 0) return fail(RGBM_ERR_ARG, "rgbm_table_create: bad argument");
    return guarded([&]() {
        use_device(device_id);
        std::unique_ptr<rgbm_table> t(new rgbm_table());
        t->device = device_id; t->n = n; t->c = c; t->n_codes.assign(n_codes, n_codes + c);
        t->codes.alloc((size_t)n * c);
        upload_pinned(t->codes.p, codes, (size_t)n * c * sizeof(int32_t));
        *out = t.release();
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return fail(RGBM_ERR_ARG, "rgbm_host_alloc: bad argument");
    return guarded([&]() {
        void* p = nullptr;
        HIPCHK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        *out = p;
        return RGBM_OK;
    });
}

RGBM_EXPORT void rgbm_host_free(void* p) { if (p) (void)hipHostFree(p); }

RGBM_EXPORT int rgbm_table_set_column_kind(rgbm_table* t, int32_t col, int32_t kind) {
    if (!t || col < 0 || col >= t->c || kind < 0 || kind > 1) return fail(RGBM_ERR_ARG, "rgbm_table_set_column_kind: bad argument");
    if (t->col_kind.size() < (size_t)t->c) t->col_kind.resize(t->c, 0);
    t->col_kind[col] = (uint8_t)kind;
    return RGBM_OK;
}

RGBM_EXPORT int rgbm_table_set_column_values(rgbm_table* t, int32_t col, const double* values, int32_t n) {
    if (!t || col < 0 || col >= t->c || n < 0 || (n > 0 && !values)) return fail(RGBM_ERR_ARG, "rgbm_table_set_column_values: bad argument");
    return guarded([&]() {
        if (n != 0 && n != t->n_codes[col]) throw std::invalid_argument("rgbm_table_set_column_values: one value per code of the column is needed");
        for (int i = 1; i < n; ++i) if (!(values[i - 1] < values[i])) throw std::invalid_argument("rgbm_table_set_column_values: values must be strictly ascending");
        if (t->col_values.size() < (size_t)t->c) t->col_values.resize(t->c);
        t->col_values[col].assign(values, values + n);
        return RGBM_OK;
    });
}

RGBM_EXPORT void rgbm_table_free(rgbm_table* t) { if (t) { (void)hipSetDevice(t->device); delete t; } }

RGBM_EXPORT int rgbm_table_read_column(const rgbm_table* t, int32_t col, int32_t* out) {
    if (!t || !out || col < 0 || col >= t->c) return fail(RGBM_ERR_ARG, "rgbm_table_read_column: bad argument");
    return guarded([&]() {
        use_device(t->device);
        HIPCHK(hipMemcpy(out, t->codes.p + (size_t)col * t->n, (size_t)t->n * sizeof(int32_t), hipMemcpyDeviceToHost));
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_train(const rgbm_table* t, int32_t target_col, const int32_t* feat_cols, int32_t f, const double* y_value,
                                 const double* class_weight, const rgbm_params* p, rgbm_model** out, rgbm_train_stats* stats) {
    if (!t || !feat_cols || !p || !out) return fail(RGBM_ERR_ARG, "rgbm_table_train: bad argument");
    return guarded([&]() {
        use_device(t->device);
        try {
            *out = train_core(*t, target_col, feat_cols, f, y_value, class_weight, nullptr, nullptr, *p, stats);
        } catch (...) {
            // a rank that fails inside a row-sharded call must not leave its peers waiting in their next all-reduce
            if ((p->reserved & RGBM_FLAG_ROW_SHARDED) && (g_comm.kind == 1 || g_comm.kind == 3)) comm_abort();
            throw;
        }
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_train_batch(const rgbm_fit_spec* fits, int32_t n_fits, rgbm_model** out_models, int32_t* out_status) {
    if (!fits || n_fits <= 0 || !out_models || !out_status || !fits[0].table) return fail(RGBM_ERR_ARG, "rgbm_table_train_batch: bad argument");
    return guarded([&]() {
        use_device(fits[0].table->device);
        for (int32_t i = 0; i < n_fits; ++i) { out_models[i] = nullptr; out_status[i] = RGBM_OK; }
        // a failure of the batch as a whole (not of one fit) returns no models: the ones already trained are released here, the caller
        // only sees the error code
        try { train_batch_small(fits, n_fits, out_models, out_status); }
        catch (...) { for (int32_t i = 0; i < n_fits; ++i) { delete out_models[i]; out_models[i] = nullptr; } throw; }
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_train(const int32_t* X, int64_t n, int32_t f, const int32_t* n_codes, const int32_t* y_code, int32_t n_y_codes,
                           const double* y_value, const double* class_weight, const double* sample_weight, const rgbm_params* p,
                           rgbm_model** out, rgbm_train_stats* stats) {
    if (!X || !n_codes || !y_code || !p || !out || n <= 0 || f <= 0 || n_y_codes <= 0) return fail(RGBM_ERR_ARG, "rgbm_train: bad argument");
    return guarded([&]() {
        use_device(p->device_id);
        for (int64_t i = 0; i < n; ++i) if (y_code[i] < 0 || y_code[i] >= n_y_codes) throw std::out_of_range("y_code outside [0, n_y_codes)");
        rgbm_table tab; tab.device = p->device_id; tab.n = n; tab.c = f + 1;
        tab.n_codes.assign(n_codes, n_codes + f); tab.n_codes.push_back(n_y_codes);
        tab.codes.alloc((size_t)n * (f + 1));
        HIPCHK(hipMemcpy(tab.codes.p, X, (size_t)n * f * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(tab.codes.p + (size_t)n * f, y_code, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
        std::vector<int32_t> fc(f); for (int i = 0; i < f; ++i) fc[i] = i;
        HostLabelStats hls;
        if (sample_weight) {   // row-order sums (numerics spec: only defined on the host-array path)
            hls.valid = true; hls.tot.assign(n_y_codes, 0.0);
            for (int64_t i = 0; i < n; ++i) {
                double w = (class_weight ? class_weight[y_code[i]] : 1.0); w = w * sample_weight[i]; w = (double)(float)w;
                hls.tot[y_code[i]] += w; if (w > hls.w_max) hls.w_max = w;
                if (y_value) hls.suml += (double)(float)y_value[y_code[i]] * w;
            }
        }
        *out = train_core(tab, f, fc.data(), f, y_value, class_weight, sample_weight, &hls, *p, stats);
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_predict(const rgbm_model* m, const int32_t* X, int64_t n, int32_t f, int32_t device_id, double* out) {
    if (!m || !X || !out || n < 0) return fail(RGBM_ERR_ARG, "rgbm_predict: bad argument");
    if (f != m->F) return fail(RGBM_ERR_ARG, "rgbm_predict: feature count differs from the model's");
    if (n == 0) return RGBM_OK;
    return guarded([&]() {
        use_device(device_id);
        StreamGuard sg; hipStream_t s = sg.s;
        DevBuf<int32_t> codes((size_t)n * f); codes.upload(X, (size_t)n * f, s);
        const int ncol = m->objective == 2 ? 1 : m->num_class;
        DevBuf<double> proba((size_t)n * ncol);
        predict_device(const_cast<rgbm_model*>(m), device_id, s, codes.p, n, 0, n, nullptr, proba.p, nullptr, nullptr);
        HIPCHK(hipMemcpy(out, proba.p, (size_t)n * ncol * sizeof(double), hipMemcpyDeviceToHost));
        return RGBM_OK;
    });
}

static int chain_device(rgbm_model* const* models, int32_t T, const int32_t* target_col, const int32_t* feat_cols, const int32_t* feat_off,
                        const int32_t* class_code, const int32_t* class_off, int32_t* d_codes, long long Ntab, long long row0, long long n,
                        int device, hipStream_t s, int32_t* out_label, double* out_prob,
                        int32_t* d_label_all = nullptr, double* d_prob_all = nullptr, long long sink_stride = 0 /* device sinks [T][sink_stride]: the outputs STAY on the device (C2 on device buffers) */) {
    using namespace rg;
    for (int t = 0; t < T; ++t)
        if (feat_off[t + 1] - feat_off[t] != models[t]->F) throw std::invalid_argument("chain: feature list length differs from the model's feature count");
    PredictScratch scratch;
    {   // size the scratch once for the largest model of the chain
        size_t mr = 0, mk = 0;
        for (int t = 0; t < T; ++t) { mr = std::max(mr, (size_t)((models[t]->F + 15) / 16)); mk = std::max(mk, (size_t)models[t]->K); }
        scratch.rec.alloc(mr * (size_t)n); scratch.raw.alloc(mk * (size_t)n);
    }
    const bool timing = getenv("RGBM_TIMING") != nullptr;
    double t_score = 0, t_fill = 0, t_copy = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    // ---- outputs: double-buffered on the device and in page-locked host memory when the staging blocks are free
    ChainStage& CS = chain_stage();
    std::unique_lock<std::mutex> cs_lock(CS.mu, std::try_to_lock);
    const size_t lab_bytes = ((size_t)n * sizeof(int32_t) + 255) & ~(size_t)255, out_bytes = lab_bytes + (size_t)n * sizeof(double);
    const bool staged = (out_label || out_prob) && T > 1 && cs_lock.owns_lock() && CS.get(out_bytes, device);
    DevBuf<int32_t> d_label[2]; DevBuf<double> d_top[2];
    for (int b = 0; b < (staged ? 2 : 1); ++b) { d_label[b].alloc(n); d_top[b].alloc(n); }
    StreamGuard copy_stream;
    hipEvent_t e_copied[2] = {nullptr, nullptr};
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int i = 0; i < 2; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } evg{e_copied};
    if (staged) for (int b = 0; b < 2; ++b) HIPCHK(hipEventCreateWithFlags(&e_copied[b], hipEventDisableTiming));
    std::atomic<int> copy_err{0};
    std::thread helper[2];
    struct HelperJoin { std::thread* h; ~HelperJoin() { for (int i = 0; i < 2; ++i) if (h[i].joinable()) h[i].join(); } } hj{helper};
    // ---- predictor tables of the chain's models: built in the background, in chain order, on a few host threads (host work + small
    // uploads, 1-25 ms a model); predict_device -> device_model waits on the model's mutex for a build in flight, or builds it itself
    std::vector<rgbm_model*> todo;
    for (int t = 0; t < T; ++t) if (std::find(todo.begin(), todo.end(), models[t]) == todo.end()) todo.push_back(models[t]);
    std::atomic<size_t> next{0}; std::mutex emu; std::exception_ptr build_err;
    ThreadJoiner builders;
    {
        const size_t nth = std::min<size_t>(std::min<size_t>(todo.size(), 16), std::max(1u, std::thread::hardware_concurrency()));
        auto work = [&]() {
            try {
                use_device(device);
                StreamGuard own;
                for (size_t j = next.fetch_add(1); j < todo.size(); j = next.fetch_add(1)) device_model(todo[j], device, own.s);
            } catch (...) { std::lock_guard<std::mutex> lk(emu); if (!build_err) build_err = std::current_exception(); }
        };
        if (todo.size() > 1) for (size_t q = 0; q < nth; ++q) builders.th.emplace_back(work);
    }
    for (int t = 0; t < T; ++t) {
        rgbm_model* m = models[t];
        const int F = m->F, b = staged ? (t & 1) : 0;
        double t1 = now();
        if (helper[b].joinable()) helper[b].join();                  // model t - 2 has left d_label[b] / d_top[b] and the staging block b
        if (copy_err.load()) throw std::runtime_error("chain: copy of the repaired cells to the host failed");
        t_copy += now() - t1; t1 = now();
        DevBuf<int32_t> d_fc(F); d_fc.upload(feat_cols + feat_off[t], F, s);
        predict_device(m, device, s, d_codes, Ntab, row0, n, d_fc.p, nullptr, d_label[b].p, d_top[b].p, &scratch);
        double t2 = now(); t_score += t2 - t1;
        if (m->objective != 2) {
            int ncc = class_off ? class_off[t + 1] - class_off[t] : m->num_class;
            DevBuf<int32_t> d_cc(std::max(ncc, 1));
            std::vector<int32_t> cc(std::max(ncc, 1), 0);
            for (int i = 0; i < ncc; ++i) cc[i] = class_code ? class_code[class_off[t] + i] : i;
            d_cc.upload(cc.data(), cc.size(), s);
            hipLaunchKernelGGL(k_fill_cells, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_codes + (long long)target_col[t] * Ntab + row0, n, d_label[b].p, d_cc.p, ncc);
            HIPCHK(hipStreamSynchronize(s));
        }
        double t3 = now(); t_fill += t3 - t2;
        if (d_label_all) HIPCHK(hipMemcpyAsync(d_label_all + (size_t)t * sink_stride, d_label[b].p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        if (d_prob_all) HIPCHK(hipMemcpyAsync(d_prob_all + (size_t)t * sink_stride, d_top[b].p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
        if ((d_label_all || d_prob_all) && !staged) HIPCHK(hipStreamSynchronize(s));
        if (staged) {
            char* pin = (char*)CS.p[b];
            if (out_label) HIPCHK(hipMemcpyAsync(pin, d_label[b].p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, copy_stream.s));
            if (out_prob) HIPCHK(hipMemcpyAsync(pin + lab_bytes, d_top[b].p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, copy_stream.s));
            HIPCHK(hipEventRecord(e_copied[b], copy_stream.s));
            int32_t* ol = out_label ? out_label + (size_t)t * n : nullptr; double* op = out_prob ? out_prob + (size_t)t * n : nullptr;
            hipEvent_t ev = e_copied[b];
            helper[b] = std::thread([=, &copy_err]() {
                if (hipEventSynchronize(ev) != hipSuccess) { copy_err.store(1); return; }
                std::thread second;
                if (ol && op) second = std::thread([=]() { memcpy(op, pin + lab_bytes, (size_t)n * sizeof(double)); });
                else if (op) memcpy(op, pin + lab_bytes, (size_t)n * sizeof(double));
                if (ol) memcpy(ol, pin, (size_t)n * sizeof(int32_t));
                if (second.joinable()) second.join();
            });
        } else {
            if (out_label) HIPCHK(hipMemcpy(out_label + (size_t)t * n, d_label[b].p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
            if (out_prob) HIPCHK(hipMemcpy(out_prob + (size_t)t * n, d_top[b].p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        }
        t_copy += now() - t3;
    }
    {
        const double t4 = now();
        for (int b = 0; b < 2; ++b) if (helper[b].joinable()) helper[b].join();
        t_copy += now() - t4;
        if (copy_err.load()) throw std::runtime_error("chain: copy of the repaired cells to the host failed");
    }
    for (auto& x : builders.th) if (x.joinable()) x.join();
    if (build_err) std::rethrow_exception(build_err);
    if (timing) fprintf(stderr, "[rgbm] chain of %d models over %lld rows: %.1f ms (score incl. waiting for tables %.1f ms, fill %.1f ms, copy out not hidden %.1f ms, %s)\n",
                        T, n, (now() - t_begin) * 1e3, t_score * 1e3, t_fill * 1e3, t_copy * 1e3, staged ? "staged" : "direct");
    return RGBM_OK;
}

RGBM_EXPORT int rgbm_repair_chain(const rgbm_model* const* models, int32_t T, const int32_t* target_col, const int32_t* feat_cols,
                                  const int32_t* feat_off, const int32_t* class_code, const int32_t* class_off, int32_t* table, int64_t n,
                                  int32_t c, int32_t device_id, int32_t* out_label, double* out_prob) {
    if (!models || T < 0 || !target_col || !feat_cols || !feat_off || !class_off || !table || n < 0 || c <= 0) return fail(RGBM_ERR_ARG, "rgbm_repair_chain: bad argument");
    if (n == 0 || T == 0) return RGBM_OK;
    return guarded([&]() {
        use_device(device_id);
        StreamGuard sg; hipStream_t s = sg.s;
        DevBuf<int32_t> codes((size_t)n * c); codes.upload(table, (size_t)n * c, s);
        chain_device(const_cast<rgbm_model* const*>(models), T, target_col, feat_cols, feat_off, class_code, class_off, codes.p, n, 0, n, device_id, s, out_label, out_prob);
        HIPCHK(hipMemcpy(table, codes.p, (size_t)n * c * sizeof(int32_t), hipMemcpyDeviceToHost));
        return RGBM_OK;
    });
}

// Rows with multiplicities: row i of the table stands for mult[i] (1..255) identical rows of a larger table (repair.pipeline.distinct_rows makes
// such a table: identical (features, label) rows take identical paths and gradients in every tree, and every sum of the trainer is an exact
// integer, so m times a row's value IS the sum over its m copies).  Every later rgbm_table_train on this table weighs the row so -- code counts,
// child counts, histogram sums, the coarse sums behind the fixed-point grids -- and returns byte for byte the model the expanded table gives.
// NULL clears.  Level grower, <= 32 features with a free byte in the last record, no bagging (rgbm.hip train_core says so when it does not apply).
RGBM_EXPORT int rgbm_table_set_row_multiplicity(rgbm_table* t, const uint8_t* mult) {
    if (!t) return fail(RGBM_ERR_ARG, "rgbm_table_set_row_multiplicity: bad argument");
    return guarded([&]() {
        use_device(t->device);
        if (!mult) { t->has_mult = false; t->mult_total = 0; return RGBM_OK; }
        int64_t tot = 0;
        for (int64_t i = 0; i < t->n; ++i) { if (mult[i] == 0) throw std::invalid_argument("rgbm_table_set_row_multiplicity: a multiplicity of 0"); tot += mult[i]; }
        if (tot >= (1ll << 31) - 4096) throw std::invalid_argument("rgbm_table_set_row_multiplicity: the expanded table has more than 2^31 rows");
        StreamGuard sg;
        t->mult.alloc((size_t)t->n);
        t->mult.upload(mult, (size_t)t->n, sg.s);
        HIPCHK(hipStreamSynchronize(sg.s));
        t->has_mult = true; t->mult_total = tot;
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_table_repair_chain(rgbm_table* t, const rgbm_model* const* models, int32_t T, const int32_t* target_col,
                                        const int32_t* feat_cols, const int32_t* feat_off, int64_t row_begin, int64_t n_rows,
                                        int32_t* out_label, double* out_prob) {
    if (!t || !models || T < 0 || !target_col || !feat_cols || !feat_off || row_begin < 0 || n_rows < 0 || row_begin + n_rows > t->n)
        return fail(RGBM_ERR_ARG, "rgbm_table_repair_chain: bad argument");
    if (n_rows == 0 || T == 0) return RGBM_OK;
    return guarded([&]() {
        use_device(t->device);
        StreamGuard sg;
        return chain_device(const_cast<rgbm_model* const*>(models), T, target_col, feat_cols, feat_off, nullptr, nullptr, t->codes.p, t->n, row_begin, n_rows,
                            t->device, sg.s, out_label, out_prob);
    });
}

RGBM_EXPORT int rgbm_model_save(const rgbm_model* m, void* buf, size_t* len) {
    if (!m || !len) return fail(RGBM_ERR_ARG, "rgbm_model_save: bad argument");
    return guarded([&]() {
        const size_t need = serialised_size(*m);
        if (!buf) { *len = need; return RGBM_OK; }
        if (*len < need) { *len = need; return fail(RGBM_ERR_ARG, "rgbm_model_save: buffer too small"); }
        serialise_into(*m, (uint8_t*)buf); *len = need;
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_model_load(const void* buf, size_t len, rgbm_model** out) {
    if (!buf || !out) return fail(RGBM_ERR_ARG, "rgbm_model_load: bad argument");
    return guarded([&]() {
        const uint8_t* p = (const uint8_t*)buf; const uint8_t* end = p + len;
        auto need = [&](size_t n) { if ((size_t)(end - p) < n) throw std::length_error("truncated model"); };
        try {
            need(28); int32_t hdr[7]; memcpy(hdr, p, 28); p += 28;
            if (hdr[0] != 0x4D424752 || (hdr[1] != 1 && hdr[1] != 2)) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: bad magic/version");
            std::unique_ptr<rgbm_model> m(new rgbm_model());
            m->objective = hdr[2]; m->num_class = hdr[3]; m->K = hdr[4]; m->n_iter = hdr[5]; m->F = hdr[6];
            if (m->F < 0 || m->K < 1 || m->n_iter < 0 || m->F > 65535) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: bad header");
            // the predictor indexes its K x n score scratch by num_class: the header must be self-consistent
            if (m->objective < 0 || m->objective > 2 || m->num_class < 1) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: bad objective / num_class");
            if (m->objective == 1 ? m->K != m->num_class : m->K != 1) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: class-tree count does not match the objective");
            if (m->objective == 0 && m->num_class != 2) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: binary model with num_class != 2");
            // every tree needs at least 16 bytes (L + one leaf value + one leaf count): bound the allocation by the buffer
            if ((unsigned long long)m->n_iter * (unsigned long long)m->K > (unsigned long long)(end - p) / 16ull) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: tree count exceeds the buffer");
            m->feats.resize(m->F);
            for (Feat& f : m->feats) {
                need(12); int32_t h3[3]; memcpy(h3, p, 12); p += 12;
                f.n_codes = h3[0]; f.V = h3[1]; f.has_nan = h3[2];
                if (f.V < 0 || f.V > 255) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: bad bin count");
                need(4 * (size_t)f.V); f.ub.resize(f.V); memcpy(f.ub.data(), p, 4 * (size_t)f.V); p += 4 * (size_t)f.V;
                if (hdr[1] == 2) {
                    need(4); int32_t nw; memcpy(&nw, p, 4); p += 4;
                    if (nw < 0 || (nw != 0 && (f.n_codes < 0 || nw != (f.n_codes + 31) / 32))) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: bad unseen-category bitmap");
                    need(4 * (size_t)nw); f.unseen.resize(nw); memcpy(f.unseen.data(), p, 4 * (size_t)nw); p += 4 * (size_t)nw;
                }
            }
            m->trees.resize((size_t)m->n_iter * m->K);
            for (Tree& t : m->trees) {
                int32_t L_; need(4); memcpy(&L_, p, 4); p += 4;
                if (L_ < 1 || L_ > 32767) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: bad leaf count");
                const size_t n = (size_t)L_ - 1;
                need(n * 28 + (size_t)L_ * 12);
                t.alloc(L_);
                auto rd32 = [&](int32_t* v, size_t k) { memcpy(v, p, 4 * k); p += 4 * k; };
                auto rd64 = [&](double* v, size_t k) { memcpy(v, p, 8 * k); p += 8 * k; };
                rd32(t.feat, n); rd32(t.theta, n); rd32(t.dleft, n); rd32(t.left, n); rd32(t.right, n); rd64(t.gain, n);
                rd64(t.leaf_value, t.L); rd32(t.leaf_count, t.L);
                for (size_t j = 0; j < n; ++j) {
                    if (t.feat[j] < 0 || t.feat[j] >= m->F) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: node feature out of range");
                    if (t.theta[j] < -1 || t.theta[j] > 254) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: node threshold out of range");
                    auto okc = [&](int ch) { return ch < 0 ? (~ch) < t.L : ch < (int)n && ch > (int)j; };
                    if (!okc(t.left[j]) || !okc(t.right[j])) return fail(RGBM_ERR_FORMAT, "rgbm_model_load: bad child link");
                }
            }
            *out = m.release();
            return RGBM_OK;
        } catch (const std::length_error&) { return fail(RGBM_ERR_FORMAT, "rgbm_model_load: truncated buffer"); }
    });
}

// ---- row-sharded multi-GPU training: communicator of the CALLING THREAD --------------------------------------
RGBM_EXPORT int rgbm_comm_unique_id(void* id_out) {
    if (!id_out) return fail(RGBM_ERR_ARG, "rgbm_comm_unique_id: bad argument");
    return guarded([&]() {
        static_assert(sizeof(ncclUniqueId) <= RGBM_COMM_ID_BYTES, "ncclUniqueId larger than RGBM_COMM_ID_BYTES");
        ncclUniqueId id; memset(&id, 0, sizeof(id));
        ncclResult_t r = ncclGetUniqueId(&id);
        if (r != ncclSuccess) throw std::runtime_error(std::string("ncclGetUniqueId failed: ") + ncclGetErrorString(r));
        memset(id_out, 0, RGBM_COMM_ID_BYTES); memcpy(id_out, &id, sizeof(id));
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_comm_init(const void* id, int32_t rank, int32_t nranks, int32_t device_id) {
    if (!id || rank < 0 || nranks < 1 || rank >= nranks) return fail(RGBM_ERR_ARG, "rgbm_comm_init: bad argument");
    return guarded([&]() {
        if (g_comm.kind != 0) throw std::invalid_argument("rgbm_comm_init: this thread already has a communicator");
        use_device(device_id);
        ncclUniqueId uid; memcpy(&uid, id, sizeof(uid));
        ncclComm_t c = nullptr;
        ncclResult_t r = ncclCommInitRank(&c, nranks, uid, rank);
        if (r != ncclSuccess) throw std::runtime_error(std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r));
        g_comm.kind = 1; g_comm.rank = rank; g_comm.nranks = nranks; g_comm.nccl = c;
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_local_group_create(int32_t nranks, int32_t device_id, void** group_out) {
    if (!group_out || nranks < 1 || nranks > 16) return fail(RGBM_ERR_ARG, "rgbm_local_group_create: bad argument (1..16 ranks)");
    return guarded([&]() {
        use_device(device_id);
        LocalGroup* g = new LocalGroup(); g->nranks = nranks; g->device = device_id; g->ptr.assign(nranks, nullptr);
        *group_out = g;
        return RGBM_OK;
    });
}

RGBM_EXPORT void rgbm_local_group_free(void* group) { delete static_cast<LocalGroup*>(group); }

RGBM_EXPORT int rgbm_comm_init_local(void* group, int32_t rank) {
    LocalGroup* g = static_cast<LocalGroup*>(group);
    if (!g || rank < 0 || rank >= g->nranks) return fail(RGBM_ERR_ARG, "rgbm_comm_init_local: bad argument");
    return guarded([&]() {
        if (g_comm.kind != 0) throw std::invalid_argument("rgbm_comm_init_local: this thread already has a communicator");
        use_device(g->device);
        g_comm.kind = 2; g_comm.rank = rank; g_comm.nranks = g->nranks; g_comm.lg = g;
        return RGBM_OK;
    });
}

// ---- fusion group: concurrent row-sharded training calls of one rank, one collective per step for all of them (see struct Fusion)
RGBM_EXPORT int rgbm_fusion_create(int32_t n_members, void** out) {
    if (!out || n_members < 1 || n_members > Fusion::MAX_MEMBERS) return fail(RGBM_ERR_ARG, "rgbm_fusion_create: bad argument (1..64 members)");
    return guarded([&]() {
        if (g_comm.kind != 1 && g_comm.kind != 2) throw std::invalid_argument("rgbm_fusion_create: the calling thread has no communicator (rgbm_comm_init)");
        std::unique_ptr<Fusion> f(new Fusion());
        HIPCHK(hipGetDevice(&f->device));
        HIPCHK(hipStreamCreateWithFlags(&f->cs, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&f->done, hipEventDisableTiming));
        f->n = n_members; f->n_members = n_members;
        f->comm = g_comm; g_comm = Comm();         // the communicator lives in the group until rgbm_fusion_free hands it back
        *out = f.release();
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_fusion_join(void* fusion, int32_t member) {
    Fusion* f = static_cast<Fusion*>(fusion);
    if (!f || member < 0 || member >= Fusion::MAX_MEMBERS) return fail(RGBM_ERR_ARG, "rgbm_fusion_join: bad argument");
    return guarded([&]() {
        if (g_comm.kind != 0) throw std::invalid_argument("rgbm_fusion_join: this thread already has a communicator");
        {
            std::lock_guard<std::mutex> lk(f->mu);
            if (member >= f->n_members) throw std::invalid_argument("rgbm_fusion_join: member index beyond the members the group was created for");
            if (f->joined[member]) throw std::invalid_argument("rgbm_fusion_join: this member index is taken");
            f->joined[member] = true;
        }
        use_device(f->device);
        g_comm.kind = 3; g_comm.rank = f->comm.rank; g_comm.nranks = f->comm.nranks; g_comm.fusion = f; g_comm.member = member;
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_fusion_leave(int32_t failed) {
    fusion_leave(failed != 0, "a member reported a failure");
    return RGBM_OK;
}

RGBM_EXPORT int rgbm_fusion_info(void* fusion, int64_t* info /* [4] = {collectives issued, member parts carried, members still in, broken} */) {
    Fusion* f = static_cast<Fusion*>(fusion);
    if (!f || !info) return fail(RGBM_ERR_ARG, "rgbm_fusion_info: bad argument");
    std::lock_guard<std::mutex> lk(f->mu);
    info[0] = f->collectives; info[1] = f->fused_parts; info[2] = f->n; info[3] = f->broken ? 1 : 0;
    return RGBM_OK;
}

RGBM_EXPORT int rgbm_fusion_free(void* fusion) {
    Fusion* f = static_cast<Fusion*>(fusion);
    if (!f) return RGBM_OK;
    return guarded([&]() {
        (void)hipSetDevice(f->device);
        if (f->cs) { (void)hipStreamSynchronize(f->cs); (void)hipStreamDestroy(f->cs); }
        if (f->done) (void)hipEventDestroy(f->done);
        for (auto& q : f->part) if (q.ready) (void)hipEventDestroy(q.ready);
        if (f->stage) (void)hipFree(f->stage);
        // the communicator goes back to the calling thread (unless that thread has one already, or the group broke and aborted it)
        if (g_comm.kind == 0 && (f->comm.kind == 1 || f->comm.kind == 2)) g_comm = f->comm;
        else { if (f->comm.kind == 1 && f->comm.nccl) (void)ncclCommDestroy(f->comm.nccl); if (f->comm.tmp) (void)hipFree(f->comm.tmp); }
        delete f;
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_comm_finalize(void) {
    if (g_comm.kind == 3) { fusion_leave(false, nullptr); return RGBM_OK; }      // a member thread: leave the group (the communicator is the group's)
    if (g_comm.kind == 1 && g_comm.nccl) (void)ncclCommDestroy(g_comm.nccl);
    if (g_comm.tmp) (void)hipFree(g_comm.tmp);
    g_comm = Comm();
    return RGBM_OK;
}

// ---- C1 / C2 of a multi-GPU job on the library's OWN communicator ("C1 / C2 on device buffers and one communicator"): the
// serialised models (Spark broadcast, python/repair/model.py:1069) and the repaired cells (the union of the UDF outputs, model.py:1142) are
// all-gathered by ncclAllGather over xGMI on the communicator the row-sharded trainer uses -- a rank holds ONE RCCL communicator; torch's process
// group only carries control traffic (gloo).  Equal-sized blocks: the callers pad to the largest rank (sizes first, through the same path).
static void gather_host_blocks(const void* send, size_t my_bytes, size_t block_bytes, void* recv) {
    Comm& c = g_comm;
    const int nr = std::max(1, c.nranks);
    StreamGuard sg; hipStream_t s = sg.s;
    const auto t0 = std::chrono::steady_clock::now();
    DevBuf<unsigned char> d_send(std::max<size_t>(block_bytes, 1)), d_recv(std::max<size_t>(block_bytes, 1) * (size_t)nr);
    d_send.zero(s);
    if (my_bytes) HIPCHK(hipMemcpyAsync(d_send.p, send, my_bytes, hipMemcpyHostToDevice, s));
    all_gather_on(c, d_send.p, d_recv.p, block_bytes, s);
    if (block_bytes) HIPCHK(hipMemcpyAsync(recv, d_recv.p, block_bytes * (size_t)nr, hipMemcpyDeviceToHost, s));
    stream_sync_watchdog(s);
    GatherStats& G = gather_stats();
    G.bytes += (long long)(block_bytes * (size_t)nr); G.calls += 1;
    G.ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

RGBM_EXPORT int rgbm_comm_all_gather_sizes(const int64_t* mine, int32_t n, int64_t* all /* [nranks][n] */) {
    if (!mine || !all || n < 1) return fail(RGBM_ERR_ARG, "rgbm_comm_all_gather_sizes: bad argument");
    return guarded([&]() { gather_host_blocks(mine, (size_t)n * 8, (size_t)n * 8, all); return RGBM_OK; });
}

RGBM_EXPORT int rgbm_comm_all_gather_bytes(const void* send, int64_t my_bytes, int64_t block_bytes, void* recv /* [nranks][block_bytes] */) {
    if (my_bytes < 0 || block_bytes < my_bytes || !recv || (my_bytes > 0 && !send)) return fail(RGBM_ERR_ARG, "rgbm_comm_all_gather_bytes: bad argument");
    return guarded([&]() { gather_host_blocks(send, (size_t)my_bytes, (size_t)block_bytes, recv); return RGBM_OK; });
}

// The chained repair of this rank's rows with its outputs LEFT ON THE DEVICE, all-gathered over the communicator there, and only then copied to
// the host: out_label / out_prob = [nranks][T][max_rows] (a rank's block holds its n_rows rows of every model, padded to max_rows = the largest
// rank's row count -- the caller exchanged the counts with rgbm_comm_all_gather_sizes).  Same labels / probabilities as rgbm_table_repair_chain.
RGBM_EXPORT int rgbm_table_repair_chain_gather(rgbm_table* t, const rgbm_model* const* models, int32_t T, const int32_t* target_col,
                                               const int32_t* feat_cols, const int32_t* feat_off, int64_t row_begin, int64_t n_rows, int64_t max_rows,
                                               int32_t* out_label, double* out_prob) {
    if (!t || !models || T < 1 || !target_col || !feat_cols || !feat_off || row_begin < 0 || n_rows < 0 || row_begin + n_rows > t->n || max_rows < n_rows || max_rows < 1 || !out_label || !out_prob)
        return fail(RGBM_ERR_ARG, "rgbm_table_repair_chain_gather: bad argument");
    return guarded([&]() {
        use_device(t->device);
        Comm& c = g_comm;
        const int nr = std::max(1, c.nranks);
        StreamGuard sg; hipStream_t s = sg.s;
        const size_t blk = (size_t)T * (size_t)max_rows;
        DevBuf<int32_t> d_lab(blk), d_lab_all(blk * (size_t)nr); DevBuf<double> d_prob(blk), d_prob_all(blk * (size_t)nr);
        d_lab.zero(s); d_prob.zero(s);
        if (n_rows > 0)
            chain_device(const_cast<rgbm_model* const*>(models), T, target_col, feat_cols, feat_off, nullptr, nullptr, t->codes.p, t->n, row_begin, n_rows,
                         t->device, s, nullptr, nullptr, d_lab.p, d_prob.p, (long long)max_rows);
        const auto t0 = std::chrono::steady_clock::now();
        all_gather_on(c, d_lab.p, d_lab_all.p, blk * sizeof(int32_t), s);
        all_gather_on(c, d_prob.p, d_prob_all.p, blk * sizeof(double), s);
        HIPCHK(hipMemcpyAsync(out_label, d_lab_all.p, blk * (size_t)nr * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(out_prob, d_prob_all.p, blk * (size_t)nr * sizeof(double), hipMemcpyDeviceToHost, s));
        stream_sync_watchdog(s);
        GatherStats& G = gather_stats();
        G.bytes += (long long)(blk * (size_t)nr * 12); G.calls += 2;
        G.ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        return RGBM_OK;
    });
}

RGBM_EXPORT int rgbm_comm_gather_stats(int64_t* out /* [3] = {bytes received by all-gathers of this process, nanoseconds spent in them, collectives} */) {
    if (!out) return fail(RGBM_ERR_ARG, "rgbm_comm_gather_stats: bad argument");
    GatherStats& G = gather_stats();
    out[0] = G.bytes.load(); out[1] = G.ns.load(); out[2] = G.calls.load();
    return RGBM_OK;
}

RGBM_EXPORT int rgbm_comm_info(int32_t* info) {
    if (!info) return fail(RGBM_ERR_ARG, "rgbm_comm_info: bad argument");
    info[0] = g_comm.kind; info[1] = g_comm.rank; info[2] = g_comm.nranks;     // (kind 3: member of a fusion group)
    return RGBM_OK;
}

// ranks the calling thread's RCCL communicator really spans (ncclCommCount), 1 for no communicator: what a multi-GPU job prints so that a
// silent fall-back to fewer ranks is visible in its result line
RGBM_EXPORT int rgbm_comm_count(int32_t* n_out) {
    if (!n_out) return fail(RGBM_ERR_ARG, "rgbm_comm_count: bad argument");
    return guarded([&]() {
        *n_out = 1;
        const Comm& c = (g_comm.kind == 3 && g_comm.fusion) ? g_comm.fusion->comm : g_comm;
        if (c.kind == 2) *n_out = c.nranks;
        if (c.kind == 1 && c.nccl) {
            int n = 0;
            ncclResult_t r = ncclCommCount(c.nccl, &n);
            if (r != ncclSuccess) throw std::runtime_error(std::string("ncclCommCount failed: ") + ncclGetErrorString(r));
            *n_out = n;
        }
        return RGBM_OK;
    });
}

RGBM_EXPORT void rgbm_model_free(rgbm_model* m) { delete m; }

RGBM_EXPORT int rgbm_model_info(const rgbm_model* m, int32_t* info) {
    if (!m || !info) return fail(RGBM_ERR_ARG, "rgbm_model_info: bad argument");
    info[0] = m->objective; info[1] = m->num_class; info[2] = m->K; info[3] = m->n_iter; info[4] = m->F;
    return RGBM_OK;
}

RGBM_EXPORT int rgbm_model_importance(const rgbm_model* m, int32_t type, double* out) {
    if (!m || !out) return fail(RGBM_ERR_ARG, "rgbm_model_importance: bad argument");
    for (int f = 0; f < m->F; ++f) out[f] = 0.0;
    for (const Tree& t : m->trees) for (int j = 0; j < t.L - 1; ++j) out[t.feat[j]] += type == 0 ? 1.0 : t.gain[j];
    return RGBM_OK;
}

}  // extern "C"
