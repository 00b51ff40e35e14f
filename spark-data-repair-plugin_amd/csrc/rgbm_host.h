// rgbm_host.h -- host-side plumbing shared by the translation units of librepairgbm.so:
// thread-local error text, HIP error checks, device buffers, the HBM-resident table object.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rgbm.h"

#define RGBM_EXPORT __attribute__((visibility("default")))

namespace rgh {

std::string& last_error();                       // thread-local (defined in rgbm.hip)
inline int fail(int code, const std::string& msg) { last_error() = msg; return code; }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            char b_[512];                                                                              \
            snprintf(b_, sizeof(b_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            throw std::runtime_error(b_);                                                              \
        }                                                                                              \
    } while (0)

// Device memory comes from a small caching pool: a training call on K class trees allocates ~1.2 KB per row and class
// tree (scores, gradients, node ids, ...) and hipMalloc/hipFree of 12 GB cost 0.3 s -- more than 15 boosting iterations.
// Freed blocks are kept (best fit: the smallest cached block that is large enough) up to RGBM_POOL_MB (default 65536;
// 0 = plain hipMalloc/hipFree); callers release buffers only after synchronising the stream that used them.
void* pool_alloc(size_t bytes, size_t* block_bytes, int* device);     // defined in rgbm.hip; throws std::runtime_error
void pool_free(void* p, size_t block_bytes, int device);
void pool_trim(size_t keep_bytes);

template <typename T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    size_t block_bytes = 0; int device = 0;
    DevBuf() {}
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    void alloc(size_t count) {
        release(); n = count;
        if (count) p = static_cast<T*>(pool_alloc(count * sizeof(T), &block_bytes, &device));
        // RGBM_POISON=1 (tests): every buffer starts out as 0xA5 bytes instead of whatever the pool block held -- a kernel that reads a
        // location nobody wrote then fails the bit-exact comparison with the oracle every time, not once in a few processes
        static const bool poison = getenv("RGBM_POISON") != nullptr;
        if (poison && count) { HIPCHK(hipMemset(p, 0xA5, count * sizeof(T))); HIPCHK(hipStreamSynchronize(nullptr)); }   // (our streams do not wait for the null stream)
    }
    void release() { if (p) { pool_free(p, block_bytes, device); p = nullptr; } n = 0; block_bytes = 0; }
    ~DevBuf() { release(); }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(block_bytes, o.block_bytes); std::swap(device, o.device); }
    void upload(const T* h, size_t count, hipStream_t s) { if (count) HIPCHK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s)); }
    void download(T* h, size_t count, hipStream_t s) const { if (count) HIPCHK(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, s)); }
    void zero(hipStream_t s) { if (n) HIPCHK(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
};

inline void use_device(int device_id) {
    int nd = 0;
    hipError_t e = hipGetDeviceCount(&nd);
    if (e != hipSuccess || nd <= 0) throw std::domain_error("no HIP device available (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= nd) throw std::domain_error("device_id out of range: there are " + std::to_string(nd) + " HIP device(s)");
    HIPCHK(hipSetDevice(device_id));
}

struct StreamGuard {
    hipStream_t s = nullptr;
    StreamGuard() { HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~StreamGuard() { if (s) (void)hipStreamDestroy(s); }
};

// C++ exceptions never cross the C ABI: every entry point runs its body through this
template <typename Fn>
int guarded(Fn&& fn) {
    try { return fn(); }
    catch (const std::invalid_argument& e) { return fail(RGBM_ERR_PARAM, e.what()); }
    catch (const std::out_of_range& e) { return fail(RGBM_ERR_LABEL, e.what()); }
    catch (const std::domain_error& e) { return fail(RGBM_ERR_NO_DEVICE, e.what()); }
    catch (const std::bad_alloc&) { return fail(RGBM_ERR_NOMEM, "out of host memory"); }
    catch (const std::exception& e) { return fail(RGBM_ERR_HIP, e.what()); }
}

// class probabilities [n][num_class] (row-major, device) of the n rows of a device code block [c][n]; defined in rgbm.hip
void predict_proba_device(const rgbm_model* m, int device, hipStream_t s, const int32_t* d_codes, long long n, const int32_t* d_feat_cols,
                          double* d_proba);
void model_shape(const rgbm_model* m, int32_t* objective, int32_t* num_class, int32_t* n_features);

}  // namespace rgh

// the label-encoded table, resident in HBM: int32 codes [c][n], column-major, -1 = NULL
struct rgbm_table {
    int device = 0; int64_t n = 0; int32_t c = 0;
    std::vector<int32_t> n_codes;
    std::vector<std::vector<double>> col_values;   // NUMERIC columns: the ascending distinct values behind the codes (rgbm_table_set_column_values)
    std::vector<uint8_t> col_kind;                 // 1 = CATEGORICAL column (rgbm_table_set_column_kind): unseen categories are missing for a model
    rgh::DevBuf<int32_t> codes;
    // rows with MULTIPLICITIES (rgbm_table_set_row_multiplicity): row i stands for mult[i] (1..255) identical rows of a larger table -- every count,
    // every gradient sum and every coarse magnitude of a training call weighs it so; the model is byte for byte the one the expanded table gives
    rgh::DevBuf<uint8_t> mult; bool has_mult = false; int64_t mult_total = 0;
    // result of the last rgbm_table_detect_* call (rgbm_prep.hip): cells (row, column), device resident
    rgh::DevBuf<long long> cell_rows; rgh::DevBuf<int32_t> cell_cols; int64_t n_cells = 0;
    // stream + scratch of the relational steps (rgbm_prep.hip), kept with the table: a hipMalloc / hipFree / stream
    // creation per call costs more than the kernels (one call at a time per table, as the header says)
    // Python threads share one table (the folds of a hyper-parameter search gather from the training table, ctypes drops the
    // GIL): every entry point that touches the stream / scratch / cell list holds prep_mu from the first use to its final synchronise.
    mutable std::mutex prep_mu;
    mutable hipStream_t stream = nullptr;
    mutable rgh::DevBuf<unsigned char> scratch[12];
    ~rgbm_table() { if (stream) (void)hipStreamDestroy(stream); }
};
