/*
 * rgbm.h -- C-ABI of librepairgbm.so: the MI355X-native (HIP, gfx950) repair-model engine.
 *
 * The reference (maropu/spark-data-repair-plugin) has no FFI of its own: its seam for this hot
 * path is the scikit-learn estimator protocol that python/repair/train.py and
 * python/repair/model.py drive against LightGBM 3.3.1.  Each entry point below names the
 * reference interface it replaces (paths relative to /root/reference).  The Python side binds
 * these with ctypes (spark-data-repair-plugin_amd/repair/gbm.py); INTEGRATION.md shows the stub
 * a maintainer of the reference would add.
 *
 * Conventions
 *   - every array is caller-owned, plain pointers + sizes, never retained after return;
 *   - feature matrices are int32 label codes, COLUMN-MAJOR ([column][row]); code -1 (or any code
 *     outside [0, n_codes)) means NULL / unknown category == LightGBM's NaN;
 *   - return 0 on success, negative on error; rgbm_last_error() gives the (thread-local) message;
 *   - device_id >= 0 selects the HIP device; there is NO CPU fallback: without a usable GPU every
 *     compute entry point fails with RGBM_ERR_NO_DEVICE;
 *   - threads: every call is re-entrant.  Training / prediction calls own a HIP stream each, so any number of threads may
 *     train from ONE table at once.  The relational table calls (detect / null / gather / count / read / write / pmf) share
 *     the table's stream and scratch and are serialised per table by the library; the two-step detect -> rgbm_table_cells_fetch
 *     protocol keeps its result in the table, so one thread at a time should drive that pair.
 */
#ifndef RGBM_H_
#define RGBM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGBM_OK 0
#define RGBM_ERR_ARG (-1)
#define RGBM_ERR_PARAM (-2)
#define RGBM_ERR_LABEL (-3)
#define RGBM_ERR_NO_DEVICE (-10)
#define RGBM_ERR_HIP (-11)
#define RGBM_ERR_NOMEM (-12)
#define RGBM_ERR_FORMAT (-20)

#define RGBM_OBJ_BINARY 0
#define RGBM_OBJ_MULTICLASS 1
#define RGBM_OBJ_REGRESSION 2

typedef struct rgbm_model rgbm_model; /* opaque; host-resident trees, lazily mirrored on a device */
typedef struct rgbm_table rgbm_table; /* opaque; an int32 code table resident in HBM */

/* LightGBM parameters as set by python/repair/train.py:102-115 (fixed) and :148-156 (searched),
 * under their LightGBM core names (sklearn aliases in comments). */
typedef struct {
    int32_t objective;    /* train.py:97-100  binary | multiclass | regression */
    int32_t num_class;    /* train.py:118-119 (multiclass only; binary uses 2) */
    int32_t n_estimators; /* model.lgb.n_estimators, train.py:53-55 */
    int32_t num_leaves;   /* searched, train.py:149 */
    int32_t max_depth;    /* model.lgb.max_depth, train.py:43-44 */
    int32_t max_bin;      /* model.lgb.max_bin, train.py:45-46 (2..255) */
    int32_t min_data_in_leaf; /* min_child_samples, train.py:153 */
    int32_t min_data_in_bin;  /* LightGBM default 3 */
    int32_t bagging_freq;     /* subsample_freq, train.py:151 */
    int32_t seed;             /* random_state=42, train.py:113 */
    int32_t device_id;        /* HIP device ordinal */
    int32_t reserved;         /* flags: RGBM_FLAG_ROW_SHARDED */
    double learning_rate;           /* model.lgb.learning_rate, train.py:41-42 */
    double lambda_l1;               /* reg_alpha, train.py:47-49 */
    double lambda_l2;               /* reg_lambda, train.py:155 */
    double min_gain_to_split;       /* min_split_gain, train.py:50-52 */
    double min_sum_hessian_in_leaf; /* min_child_weight, train.py:154 */
    double bagging_fraction;        /* subsample, train.py:150 */
    double feature_fraction;        /* colsample_bytree, train.py:152 */
} rgbm_params;

/* Per-call measurements filled by the training entry points (bench.py roofline inputs). */
typedef struct {
    double hist_ms;          /* sum of hist_build kernel time (HIP events on the launch stream) */
    double total_ms;         /* whole training call, device side */
    int64_t hist_launches;   /* number of hist_build launches */
    int64_t hist_rows;       /* rows scanned by hist_build over all launches and class trees */
    int64_t hist_bytes;      /* algorithmic bytes: rows * (F + 8) (+4 per row via an index list) */
    int64_t root_rows;       /* of which full-table (root) scans */
    double root_ms;          /* hist_build time spent in root scans */
    int64_t trees;           /* trees grown */
    double route_ms;         /* always 0 since numerics version 200: DataPartition::Split of a level is part of the level pass (k_level_mt), */
    int64_t route_launches;  /* whose time is in hist_ms; the fields keep the struct layout of version 102 */
    /* version 220: what the LDS-atomic floor of the histogram build is priced with (bench.py roofline.classes.*.atomic_floor_us) */
    int64_t root_atomics_per_row;   /* 64-bit LDS atomics a root pass issues per accumulated row: 2 per joint-bin group (or per feature without joint bins) */
    int64_t level_atomics_per_row;  /* ... a level pass issues per built row: 2 per feature */
} rgbm_train_stats;

/* Number of usable HIP devices (0 if none). */
int rgbm_device_count(void);
/* The library parks freed device blocks in a caching pool (at most RGBM_POOL_MB, default 65536; 0 disables it).
 * rgbm_release_cache gives all of them back to the driver, e.g. before another allocator needs the memory, together with the
 * page-locked host blocks the library keeps for the tree harvest of batched fits and for the outputs of repair chains. */
int rgbm_release_cache(void);
/* Message of the last failing call on this thread. */
const char* rgbm_last_error(void);
/* Library / numerics-spec version (bumped whenever results could change). */
int rgbm_version(void);

/* ---- fit --------------------------------------------------------------------------------
 * Replaces `lgb.LGBMClassifier(**p).fit(X, y)` / `lgb.LGBMRegressor(**p).fit(X, y)`:
 * python/repair/train.py:121-131 (construction), :171-172 (inside cross_val_score), :215-216
 * (final fit).
 *   X_colmajor [f][n]   feature codes;  n_codes[f] dictionary sizes
 *   y_code [n]          class index (0..n_y_codes-1) or, for regression, index into y_value
 *   y_value [n_y_codes] regression only: the distinct target values, ascending
 *   class_weight [n_y_codes] or NULL: per-label weight (sklearn class_weight='balanced' is
 *                       n / (n_labels * count_label), train.py:39-40,105); sample_weight [n] or NULL
 */
int rgbm_train(const int32_t* X_colmajor, int64_t n, int32_t f, const int32_t* n_codes,
               const int32_t* y_code, int32_t n_y_codes, const double* y_value,
               const double* class_weight, const double* sample_weight,
               const rgbm_params* p, rgbm_model** out, rgbm_train_stats* stats /* may be NULL */);

/* ---- predict_proba / predict ------------------------------------------------------------
 * Replaces `model.predict_proba(X)` / `model.predict(X)`: python/repair/model.py:1120,1130.
 * out: binary [n][2] = {1-p, p}; multiclass [n][num_class]; regression [n]. */
int rgbm_predict(const rgbm_model* m, const int32_t* X_colmajor, int64_t n, int32_t f,
                 int32_t device_id, double* out);

/* ---- chained repair ---------------------------------------------------------------------
 * Replaces the body of the grouped-map UDF `repair(pdf)`: python/repair/model.py:1107-1133.
 * For each model in order: score every row from the listed feature columns, arg-max (first
 * maximum, like numpy), then overwrite ONLY the NULL cells of the target column with
 * class_code[label] so that later models see the repair.
 *   table [c][n] codes, modified in place; feat_cols/feat_off[T+1]; class_code/class_off[T+1]
 *   (an empty class list == regression target: column left untouched);
 *   out_label [T][n] class index (-1 for regression), out_prob [T][n] its probability / raw value. */
int rgbm_repair_chain(const rgbm_model* const* models, int32_t T, const int32_t* target_col,
                      const int32_t* feat_cols, const int32_t* feat_off,
                      const int32_t* class_code, const int32_t* class_off,
                      int32_t* table_colmajor, int64_t n, int32_t c, int32_t device_id,
                      int32_t* out_label, double* out_prob);

/* ---- HBM-resident table (the batch path: python/repair/model.py:768-815 without toPandas) ---
 * Upload the label-encoded table once; train one model per target attribute on the rows whose
 * target cell is non-NULL (model.py:776), all other listed columns being the features. */
int rgbm_table_create(const int32_t* codes_colmajor, int64_t n, int32_t c, const int32_t* n_codes,
                      int32_t device_id, rgbm_table** out);
/* CATEGORICAL columns (kind 1; default 0 = ordered codes): a model trained from this table records which codes of the column no
 * training row held, and treats them as MISSING when it predicts -- what the reference's per-model encoders do with a category they
 * have not seen (python/repair/model.py:701-729: the encoders are fitted on the training frame of each model).  Such models are
 * serialised as format version 2 (version 1 + one bitmap per feature); row gathers inherit the kinds. */
int rgbm_table_set_column_kind(rgbm_table* t, int32_t col, int32_t kind);
/* NUMERIC columns: the ascending distinct values behind the codes 0..n_codes[col]-1 (one per code; n = 0 clears).  The trainer then
 * places bin bounds at the midpoints of the VALUES, as LightGBM does on raw numbers (bin.cpp GreedyFindBin), instead of at the
 * midpoints of the rank codes: it only matters for values that no training row of the model holds.  Row gathers inherit it. */
int rgbm_table_set_column_values(rgbm_table* t, int32_t col, const double* values, int32_t n);
/* Pinned host memory for the code matrix (north_star: "pinned int32 column-major feature matrix"): a block from
 * rgbm_host_alloc is page-locked, so rgbm_table_create copies it at PCIe DMA speed without staging.  Any other block is
 * page-locked in place for the duration of the copy (and copied the plain way if that fails). */
int rgbm_host_alloc(size_t bytes, void** out);
void rgbm_host_free(void* p);
void rgbm_table_free(rgbm_table* t);
int rgbm_table_train(const rgbm_table* t, int32_t target_col, const int32_t* feat_cols, int32_t f,
                     const double* y_value /* regression dictionary or NULL */,
                     const double* class_weight /* [n_codes[target]] or NULL */,
                     const rgbm_params* p, rgbm_model** out, rgbm_train_stats* stats);
/* Rows with multiplicities (round 6; a VARIANT of the workload, reported apart from the row-for-row line): row i stands for mult[i] (1..255)
 * identical rows of a larger table.  The reference has no counterpart -- its frames (python/repair/model.py:768-815) hold every row; on the
 * categorical tables it repairs a quarter of the rows are distinct -- but the result is defined by it: every later rgbm_table_train on this
 * table returns byte for byte the model the EXPANDED table gives (identical rows take identical paths and gradients; all sums are exact
 * integers).  NULL clears.  Level grower only (1 <= max_depth <= 7), at most 32 features with a free byte in the last 16-feature record, no
 * bagging, no per-row weights: a training call that cannot honour it fails with RGBM_ERR_PARAM. */
int rgbm_table_set_row_multiplicity(rgbm_table* t, const uint8_t* mult /* [n] or NULL */);
/* ---- many small fits in one go ------------------------------------------------------------------
 * Replaces the LOOP over fits of python/repair/train.py:158-209 (every hyper-parameter trial is `cross_val_score`: n_splits fits of
 * the same estimator on row subsets of one frame, train.py:171-172) and of python/repair/model.py:768-815 on the reference's default
 * <= 10 000-row training samples (model.py:755-766).  Each spec is one rgbm_table_train call -- its own table (e.g. a fold gathered
 * with rgbm_table_gather_rows), target, features, class weights and parameters -- and every model is the one that call returns, bit
 * for bit; what changes is the schedule: all fits of the batch advance through their boosting iterations TOGETHER, three kernel
 * launches per iteration for the whole batch (csrc/rgbm_small.h: one workgroup grows one class tree of one fit).  Fits the fused
 * kernels do not cover (tables above RGBM_SMALL_ROWS = 65536 rows, more than 256 leaves, row-sharded calls) run one by one inside.
 * A fit with a valid_table always takes the batched path's scoring (its table must then be small enough for the fused kernels).
 * out_status[i] = RGBM_OK or fit i's error code (out_models[i] = NULL then): one failing fit does not fail the batch, which is the
 * reference's contract (train.py:227-229: a failing build yields PoorModel).  All tables must live on one device. */
typedef struct {
    const rgbm_table* table;
    int32_t target_col, n_features;
    const int32_t* feat_cols;
    const double* y_value;        /* regression dictionary or NULL */
    const double* class_weight;   /* [n_codes[target]] or NULL */
    const rgbm_params* params;
    /* optional: score a validation table while the fit trains (cross_val_score, train.py:171-172, without building a predictor): the
     * rows of valid_table get the label / value model.predict would give them -- the same bits, the trees are added to their scores
     * in the predictor's order.  valid_label_out [rows of valid_table]: arg-max class index (-1 for regression);
     * valid_value_out [rows]: its probability / the regression value.  NULL valid_table: nothing is scored. */
    const rgbm_table* valid_table;
    int32_t* valid_label_out;
    double* valid_value_out;
} rgbm_fit_spec;
int rgbm_table_train_batch(const rgbm_fit_spec* fits, int32_t n_fits, rgbm_model** out_models /* [n_fits] */, int32_t* out_status /* [n_fits] */);
/* Chained repair of rows [row_begin, row_begin+n_rows) of the resident table, in place in HBM.
 * out_label/out_prob [T][n_rows] are copied back to the host (either may be NULL). */
int rgbm_table_repair_chain(rgbm_table* t, const rgbm_model* const* models, int32_t T,
                            const int32_t* target_col, const int32_t* feat_cols, const int32_t* feat_off,
                            int64_t row_begin, int64_t n_rows, int32_t* out_label, double* out_prob);
/* Copy one column of the resident table back to the host. */
int rgbm_table_read_column(const rgbm_table* t, int32_t col, int32_t* out /* [n] */);

/* ---- the relational steps either side of the models, on the resident table -------
 * Cells are (row position, column index) pairs; the host maps row ids <-> positions.  Result lists are
 * ordered -- by position in the given column list, then ascending row -- so they are a deterministic function
 * of the table.  A detect / rows_of_cells call leaves its result in the table object (device memory) and
 * returns the count; rgbm_table_cells_fetch copies it out.  One such call at a time per table. */
/* NullErrorDetector (src/main/scala/.../python/ErrorDetectorApi.scala:128-157): the NULL cells of `cols`. */
int rgbm_table_detect_nulls(rgbm_table* t, const int32_t* cols, int32_t n_cols, int64_t* n_cells_out);
/* ConstraintErrorDetector (ErrorDetectorApi.scala:189-244) for two-tuple denial constraints of the form
 * t1&t2&EQ(t1.X1,t2.X1)&..&EQ(t1.Xm,t2.Xm)&IQ(t1.Y,t2.Y)  (i.e. X1..Xm -> Y): a row violates iff another row
 * agrees with it on every X (NULL-safe, `<=>`) and differs on Y (NULL is a value of its own).  n_eq == 0 is allowed
 * (IQ alone: every row violates as soon as Y takes two values).  Result: the violating rows x cell_cols
 * (n_cell_cols == 0: just the rows). */
int rgbm_table_detect_constraint(rgbm_table* t, const int32_t* eq_cols, int32_t n_eq, int32_t iq_col,
                                 const int32_t* cell_cols, int32_t n_cell_cols,
                                 int64_t* n_rows_out /* may be NULL */, int64_t* n_cells_out);
/* Ascending positions of the rows that hold at least one of the given cells: the dirty rows of
 * python/repair/model.py:549-553 (left-semi join on the row id). */
int rgbm_table_rows_of_cells(rgbm_table* t, const int64_t* rows, int64_t n_cells, int64_t* n_rows_out);
int rgbm_table_cells_fetch(const rgbm_table* t, int64_t* rows_out /* [n] or NULL */, int32_t* cols_out /* [n] or NULL */);
/* convertErrorCellsToNull (src/main/scala/.../python/RepairApi.scala:171-211): NULL the listed cells whose
 * column is one of target_cols; cells outside the table are ignored (join semantics). */
int rgbm_table_null_cells(rgbm_table* t, const int64_t* rows, const int32_t* cols, int64_t n_cells,
                          const int32_t* target_cols, int32_t n_targets);
/* The codes the given cells hold now (cells outside the table read as NULL): the `current_value` of the error cells. */
int rgbm_table_read_cells(const rgbm_table* t, const int64_t* rows, const int32_t* cols, int64_t n_cells, int32_t* codes_out);
/* Store codes into the given cells (cells outside the table are ignored).  The chained repair uses it for CONTINUOUS target
 * attributes: the regressor's prediction (python/repair/model.py:1130-1133, rounded first for integral attributes) is written
 * back as the code of the nearest dictionary value so that later models of the chain see the repaired cell. */
int rgbm_table_write_cells(rgbm_table* t, const int64_t* rows, const int32_t* cols, const int32_t* codes, int64_t n_cells);
/* New resident table made of the given rows (the dirty-row frame the chained repair runs on). */
int rgbm_table_gather_rows(const rgbm_table* t, const int64_t* rows, int64_t n_rows, rgbm_table** out);
/* Rows per code of one column (+ NULL count): class weights (train.py:39-40,105), domain statistics. */
int rgbm_table_count_codes(const rgbm_table* t, int32_t col, int64_t* counts_out /* [n_codes[col]] */,
                           int64_t* n_null_out /* may be NULL */);
/* Encoding on the device (replaces the pandas encoders of python/repair/model.py:701-729): per column, Arrow-style
 * dictionary indices (idx < 0 = NULL) are mapped through remap[col][idx] (the rank of the dictionary value in
 * sorted order, or -1) into the code table. */
int rgbm_table_create_dict(const int32_t* idx_colmajor, int64_t n, int32_t c, const int32_t* const* remap,
                           const int32_t* dict_size /* [c] */, int32_t device_id, rgbm_table** out);
/* Candidate distributions of the NULL cells of one target attribute (python/repair/model.py:1196-1212): for every row
 * whose target cell is NULL (ascending), the model's classes by descending probability (ties keep class order), those
 * with prob > threshold, at most top_k; class_out is padded with -1, prob_out with 0.  cur_code (optional, per cell: the
 * code the cell held before it was NULLed, -1 = none) -> cur_prob_out = the model's probability of that value.
 * More cells than `cap` is an error that still reports the count in n_cells_out (call again with enough room). */
int rgbm_table_repair_pmf(rgbm_table* t, const rgbm_model* m, int32_t target_col, const int32_t* feat_cols, int32_t f,
                          int32_t top_k, double threshold, const int32_t* cur_code, int64_t cap, int64_t* n_cells_out,
                          int64_t* rows_out /* [cap] */, int32_t* class_out /* [cap][top_k] */,
                          double* prob_out /* [cap][top_k] */, double* cur_prob_out /* [cap] or NULL */);
int rgbm_table_shape(const rgbm_table* t, int64_t* n_out, int32_t* c_out, int32_t* n_codes_out /* [c] or NULL */);

/* ---- row-sharded multi-GPU training ------------------------------------------------------------
 * The reference parallelises training per target attribute (python/repair/model.py:817-926); a single
 * multiclass target cannot be split that way.  Here every rank (one process per GPU) uploads a ROW SHARD
 * of the table with rgbm_table_create and calls rgbm_table_train for the same targets in the same
 * order; inside, the code counts, the per-level histograms and the child row counts are summed with
 * integer all-reduces (RCCL over xGMI, enqueued on the training stream).  All sums are exact integers,
 * so every rank ends with the same model, bit-identical to single-GPU training on the whole table.
 * The communicator belongs to the calling thread.  Level grower only, no per-row weights; bagging draws per global training-row
 * position (the shards must hold consecutive row ranges in rank order), so the model is the single-device one. */
#define RGBM_COMM_ID_BYTES 128
#define RGBM_FLAG_ROW_SHARDED 1 /* rgbm_params.reserved: this call is one rank of a row-sharded training */
#define RGBM_FLAG_NO_MODEL 2    /* rgbm_params.reserved, rgbm_table_train_batch only: the fit is wanted for its validation scores (a CV fold); no model
                                   is built, out_models[i] stays NULL with status RGBM_OK */
int rgbm_comm_unique_id(void* id_out /* [RGBM_COMM_ID_BYTES], call on one rank, hand to all */);
int rgbm_comm_init(const void* id, int32_t rank, int32_t nranks, int32_t device_id);
int rgbm_comm_finalize(void);
int rgbm_comm_info(int32_t* info /* [3] = {0 none | 1 RCCL | 2 thread group | 3 member of a fusion group, rank, nranks} */);
int rgbm_comm_count(int32_t* n_out /* ranks the communicator really spans (ncclCommCount); 1 without one */);
/* Test transport: the ranks are host threads of one process sharing one device. */
/* ---- C1 / C2 of a multi-GPU job on THIS communicator (device buffers, ncclAllGather over xGMI; a rank holds one RCCL communicator) -------
 * Replace `sparkContext.broadcast(models)` (python/repair/model.py:1069: every worker gets every serialised model) and the union of the
 * grouped-map UDF outputs (model.py:1142: the repaired cells of all row groups).  Blocks are equal-sized: callers exchange their sizes with
 * rgbm_comm_all_gather_sizes first and pad to the largest.  Thread group / no communicator: the same calls, device copies.  Not from inside a
 * fusion group. */
int rgbm_comm_all_gather_sizes(const int64_t* mine, int32_t n, int64_t* all /* [nranks][n] */);
int rgbm_comm_all_gather_bytes(const void* send, int64_t my_bytes, int64_t block_bytes /* >= every rank's my_bytes */, void* recv /* [nranks][block_bytes] */);
/* rgbm_table_repair_chain of this rank's rows with the outputs left on the device, all-gathered there and copied out once:
 * out_label / out_prob [nranks][T][max_rows] (a rank's block: its n_rows rows of every model, zero-padded to max_rows). */
int rgbm_table_repair_chain_gather(rgbm_table* t, const rgbm_model* const* models, int32_t T, const int32_t* target_col,
                                   const int32_t* feat_cols, const int32_t* feat_off, int64_t row_begin, int64_t n_rows, int64_t max_rows,
                                   int32_t* out_label, double* out_prob);
int rgbm_comm_gather_stats(int64_t* out /* [3] = {bytes received, nanoseconds, collectives} of this process's all-gathers */);
int rgbm_local_group_create(int32_t nranks, int32_t device_id, void** group_out);
void rgbm_local_group_free(void* group);
int rgbm_comm_init_local(void* group, int32_t rank);

/* Fusion group: the row-sharded training calls a rank makes AT THE SAME TIME (one host thread + one stream each; the reference trains its
 * targets in parallel, python/repair/model.py:817-926).  The calling thread's communicator moves into the group; every member thread joins
 * with its index (the same indices on every rank), trains its targets with RGBM_FLAG_ROW_SHARDED as usual and leaves.  Inside, the i-th
 * collective of every member still taking part is carried by ONE all-reduce per element type of the rank's communicator over the members'
 * buffers laid side by side (member order): a rank issues 8 collectives per boosting iteration of ALL its row-sharded targets instead of 8
 * per target, and the targets overlap on the device.  Models are unchanged (integer sums).  A member that fails breaks the group: the other
 * members' calls fail at their next collective.  rgbm_fusion_free hands the communicator back to the calling thread. */
int rgbm_fusion_create(int32_t n_members, void** fusion_out);
int rgbm_fusion_join(void* fusion, int32_t member /* 0 .. n_members-1, on the member's own thread */);
int rgbm_fusion_leave(int32_t failed /* 0: done; 1: this member failed -- the group is broken */);
int rgbm_fusion_info(void* fusion, int64_t* info /* [4] = {collectives issued, member parts carried, members still in, broken} */);
int rgbm_fusion_free(void* fusion);

/* ---- model handle: pickling (python/repair/model.py:910,921,1069) -------------------------- */
int rgbm_model_save(const rgbm_model* m, void* buf, size_t* len); /* buf==NULL: length query */
int rgbm_model_load(const void* buf, size_t len, rgbm_model** out);
void rgbm_model_free(rgbm_model* m);
/* info[5] = {objective, num_class, trees_per_iteration, n_iterations, n_features} */
int rgbm_model_info(const rgbm_model* m, int32_t* info);
/* LightGBM feature_importances_ (train.py:219): type 0 = split counts, 1 = total gain. */
int rgbm_model_importance(const rgbm_model* m, int32_t type, double* out /* [n_features] */);

#ifdef __cplusplus
}
#endif
#endif /* RGBM_H_ */
