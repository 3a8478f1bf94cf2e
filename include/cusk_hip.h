/*
 * cusk_hip.h -- C ABI of libcusk_hip.so, the MI355X-native `cusk` PC-skeleton engine.
 *
 * Plain pointers and sizes only.  Two groups of entry points:
 *
 *  (1) the reference's own operator interface for this path, with identical
 *      names, argument order and argument meaning, so that the reference's
 *      callers (cusk/src/cli.cpp:45,72,550,581,670) link against this library
 *      unchanged -- host buffers in, host buffers out, blocking;
 *  (2) a device-resident engine API (cusk_*) that the `mps` host program and
 *      bench.py use: matrices stay in HBM between the correlation build and
 *      the level sweep, adjacency is a bitmap, separation sets are sparse.
 *
 * Citations are to /root/reference/cusk.
 */
#ifndef CUSK_HIP_H_
#define CUSK_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUSK_ML 14 /* include/mps/cuPC-S.h:49 (ML) */

/* ---------------------------------------------------------------------------
 * (1) reference-compatible entry points
 * ------------------------------------------------------------------------ */

/* Replaces `extern "C" void Skeleton(...)`, include/mps/cuPC-S.h:196-198,
 * src/cuPC-S.cu:61-450.  All pointers are caller-owned HOST memory.
 *   C        in   n*n fp32 row-major correlation matrix
 *   P        in   *P = n
 *   G        out  n*n int32 adjacency (fully overwritten)
 *   Th       in   thresholds Th[0..min(14,*maxlevel)]
 *   l        out  level counter as the reference leaves it
 *   maxlevel in   maximal conditioning-set size (<= 14)
 *   pMax     out  n*n fp32: -100000 on surviving edges, max Fisher z of the
 *                 two directions on removed ones, 1 on the diagonal
 *   SepSet   out  n*n*14 int32, -1 padded
 * Errors (HIP failure, combinatorial overflow): message on stderr, exit(EXIT_FAILURE),
 * as include/mps/gpuerrors.h:6-15 does. */
void Skeleton(float *C, int *P, int *G, float *Th, int *l, const int *maxlevel, float *pMax,
              int *SepSet);

/* Replaces `extern "C" void hetcor_skeleton(...)`, include/mps/hetcor-cuPC-S.h:46,
 * src/hetcor-cuPC-S.cu:75-341.  G is IN/OUT (zeros are respected, level 0 only
 * removes), N is the n*n effective-sample-size matrix, *Th the single
 * alpha/2 quantile, time_index has n entries (markers 0). */
void hetcor_skeleton(float *C, int *P, int *G, float *N, float *Th, int *l, const int *maxlevel,
                     const int *time_index);

/* Replaces threshold_array / hetcor_threshold, include/mps/cuPC_call_prep.h:7-15,
 * src/cuPC_call_prep.cpp:13-27 (the reference returns std::vector<float>; here
 * the caller passes room for 15 floats). */
void cusk_threshold_array(int n, float alpha, float *thr15);
float cusk_hetcor_threshold(float alpha);

/* Replace cu_marker_phen_corr_pearson / cu_corr_pearson_npn,
 * include/mps/corr_host.h:38-47,92-103, src/corr_host.cu:1023-1197.  Host
 * buffers; marker_vals is SNP-major packed .bed without the 3 magic bytes,
 * phen_vals column-major with NaN = missing.  Outputs: marker_corrs upper
 * triangle without diagonal (row-major linear), marker_phen_corrs m*p
 * row-major, phen_corrs upper triangle without diagonal. */
void cu_marker_phen_corr_pearson(const unsigned char *marker_vals, const float *phen_vals,
                                 const size_t num_markers, const size_t num_individuals,
                                 const size_t num_phen, const float *marker_mean,
                                 const float *marker_std, float *marker_phen_corrs);
void cu_corr_pearson_npn(const unsigned char *marker_vals, const float *phen_vals,
                         const size_t num_markers, const size_t num_individuals,
                         const size_t num_phen, const float *marker_mean, const float *marker_std,
                         float *marker_corrs, float *marker_phen_corrs, float *phen_corrs);

/* ---------------------------------------------------------------------------
 * (2) device-resident engine
 * ------------------------------------------------------------------------ */

typedef struct cusk_engine cusk_engine;

enum {
    CUSK_OK = 0,
    CUSK_ERR_HIP = 1,      /* a HIP runtime call failed */
    CUSK_ERR_ARG = 2,      /* bad argument */
    CUSK_ERR_OVERFLOW = 3, /* C(degree, level) does not fit 62 bits */
    CUSK_ERR_STATE = 4     /* call order (no result to fetch, ...) */
};

/* per-run counters; index = level */
typedef struct cusk_stats {
    int level;                        /* what the reference leaves in *l */
    int levels_run;                   /* number of levels that launched tests (incl. level 0) */
    int max_degree[CUSK_ML + 1];      /* max degree at the start of the level (level 0: n-1) */
    long long edges[CUSK_ML + 1];     /* directed edges at the start of the level */
    long long tests[CUSK_ML + 1];     /* CI tests evaluated (SURVEY.md 8d definition) */
    long long subsets[CUSK_ML + 1];   /* conditioning sets whose inverse was formed */
    long long removed[CUSK_ML + 1];   /* ordered pairs (X,Y) for which a separating set was found */
    float kernel_ms[CUSK_ML + 1];     /* HIP-event time of the level's sweep kernels */
    float level_ms[CUSK_ML + 1];      /* HIP-event time of the whole level (compaction + sweep + finalise) */
    float total_ms;                   /* whole run, events on the engine stream */
    long long rechecks[CUSK_ML + 1];  /* tests the fast filter could not certify (re-evaluated on the exact path) */
    long long violations;             /* validate mode only: certified verdicts contradicted by the exact path */
    long long exact_fallbacks;        /* levels redone entirely on the exact path (recheck queue overflow) */
    float main_kernel_ms[CUSK_ML + 1]; /* HIP-event time of the level's dominant kernel alone (level 1: the rows
                                         kernel without its prep / count passes; other levels = kernel_ms) */
    long long canonical_tests[CUSK_ML + 1]; /* cusk_run_skeleton: CI tests of the CANONICAL schedule -- the reference
                                         algorithm run sequentially per row: a neighbour is tested with every conditioning
                                         set up to its lowest passing one -- computed on the device from
                                         the selected ranks; `tests` counts what the parallel sweep executed (more: lanes
                                         cannot see each other's fresh verdicts).  cusk_run_hetcor: level 0 and -- on a
                                         symmetric matrix with one sample size and no time index (the row-streaming
                                         kernel) -- level 1; 0 at the other levels. */
} cusk_stats;

/* device = HIP device ordinal; stream = a hipStream_t to run on, or NULL for a
 * private non-blocking stream.  Returns CUSK_OK or an error code. */
int cusk_engine_create(cusk_engine **out, int device, void *stream);
void cusk_engine_destroy(cusk_engine *e);
const char *cusk_last_error(const cusk_engine *e);
/* options: "fast" (default 1: register-Cholesky filter + exact recheck for levels >= 2; 0: exact
 * arithmetic for every test), "validate" (default 0; 1: also run the exact path on every certified
 * verdict and count contradictions in cusk_stats.violations), "pair" (default 1: level-1 kernels that exploit a
 * symmetric matrix; 0: generic staged kernel), "rows" (default 1: the row-streaming level-1 kernel that reads C
 * exactly once; 0: the pair-gather kernel), "vec" (default 1: vectorised
 * four-tests-per-ds_read_b128 sweep kernel; 0: scalar fast kernel), "overlap" (default 1: independent degree
 * classes and the winners' exact z run on an auxiliary stream), "corr_fp4" (default 1: the SNP x SNP contingency GEMMs of
 * cusk_corr_build on the FP4 matrix pipe; 0: the int8 MFMA form), "corr_popcount" (default 0; 1: bit-plane AND/popcount
 * cross-check kernels instead of the matrix cores), "corr_mxp_f32" (default 0: SNP x trait sums on the bf16 matrix pipe with every
 * trait value split exactly into three bf16 pieces; 1: the f32 matrix instructions of rounds 1-2), "assume_symmetric" (default 0: level 0 verifies C == C^T bitwise;
 * 1: the caller guarantees it, e.g. a matrix written by cusk_corr_build), "queue_capacity" (recheck queue entries, default 4Mi), "chunk" (combination ranks per work item, default 2048), "timing" (HIP events for cusk_stats: 0 total_ms only; 1, the default, around every level's sweep: kernel_ms, and level_ms = end of the previous level's sweep to the end of this one's; 2 also level start / end: level_ms = plan to finaliser; 3 only the pair around the level-1 row kernel: main_kernel_ms[1] -- every event costs a few microseconds of device time), "chunk0" (conditioning sets per work item of the first degree class, default 512), "tmaj_min_level" (first level swept by unions T = S + Y, one inverse per l + 1 tests: default 6, 99 = never; single threshold and symmetric matrix only), "tmaj_validate_stride" (with "validate": the union-major sweep checks the unions whose per-lane count is a multiple of this power of two against double precision; default 1 = all), "hostprof" (1: host-side phase marks of every run on stderr), "max_staged_classes" (test hook: at most this many degree classes keep their sub-matrix in LDS; 0 sends every row through the kernels of the unstaged class), "chunk0_low" (work-item size of the first degree class at levels 2-4, default 256), "vec_threads" (workgroup size of the vectorised sweep for the first degree class: 64, 128 or 256; default 64), "lookahead" (levels the host enqueues ahead of the level counters it has seen, default 2; every kernel checks its level's gate on the device), "sync2" (default 1: the host reads level 2's gate record -- class counts, maximum degree -- before it enqueues that level's sweeps, so that degree classes that turn out empty are not launched at levels >= 2; 0: enqueue ahead on the level-1 degree bound), "item_capacity" (work items per degree class and level the buffers hold before the engine grows them and takes the level up again, default 1Mi), "sepselect_ws_bytes" (HBM work space of
 * cusk_sepselect_greedy for candidate lists too long for LDS, default 4 GiB; such pairs run in batches of what fits). */
int cusk_engine_set_option(cusk_engine *e, const char *key, long long value);
void *cusk_engine_stream(const cusk_engine *e);
/* Makes the engine's device the calling thread's current HIP device, so that the cusk_dev_* helpers below act on it.
 * Needed on multi-GPU nodes by threads other than the one that created the engine (HIP's current device is per thread
 * and starts at 0). */
int cusk_engine_bind_thread(cusk_engine *e);
int cusk_engine_device(const cusk_engine *e); /* the HIP device ordinal the engine was created on */

/* Row-sharded sweep of ONE block over several engines, normally one per GPU (a single matrix too
 * large or too slow for one device; the reference has no counterpart, its Skeleton is single-GPU, cuPC-S.cu:42-190).
 * Every engine of the group holds the whole matrix and calls cusk_run_skeleton with identical arguments; engine
 * `rank` of `world` runs the conditional-independence tests of the rows X with X % world == rank only.  After each
 * level's sweep the engines join their per-edge selection state -- the lowest passing conditioning-set rank of every
 * ordered pair, "none" = all ones -- through `fn`, which must perform an element-wise UNSIGNED MIN all-reduce of
 * `count` elements of `elem_bytes` (4 at level 1, 8 beyond) across the group, in place, and return 0 once the
 * result is in the buffer.  Removal of edges, separating-set records and the next level's neighbour lists are then
 * derived identically on every engine, so all of them finish with the complete result (bit-identical to a
 * single-engine run).  host_staging = 1: `buf` is a pinned HOST copy (on_device = 0; for CPU collectives such as
 * gloo); 0: `buf` is the DEVICE buffer itself (on_device = 1; RCCL), the engine's stream is idle during the call and
 * the callee must have finished with the buffer when it returns.  world = 1 switches sharding off.  cusk_run_hetcor
 * shards the same way: what its engines join is one 32-bit mark per directed edge, 0 = removed at this level (elem_bytes
 * 4 at every level), and the removal of the marked edges is then applied identically on every engine. */
typedef int (*cusk_exchange_fn)(void *user, int level, void *buf, size_t count, int elem_bytes, int on_device, void *stream);
int cusk_engine_set_row_shard(cusk_engine *e, int rank, int world, cusk_exchange_fn fn, void *user, int host_staging);

/* Level sweep on a matrix already resident in HBM (C_dev: n*n fp32 row-major).
 * cusk_run_skeleton   : `Skeleton` semantics (fixed per-level thresholds Th[0..14] on
 *                       the host, sepsets + pMax recorded sparsely).
 * cusk_run_hetcor     : `hetcor_skeleton` semantics.  N_dev may be NULL when every
 *                       pair has the same effective sample size `ess_uniform`;
 *                       G_init_dev (n*n int32, device) may be NULL for the
 *                       complete graph; time_index is a HOST array of n ints or
 *                       NULL (all zero).
 * Both are asynchronous with respect to other streams but return after the
 * level loop has finished (the loop needs the max degree on the host each level). */
int cusk_run_skeleton(cusk_engine *e, const float *C_dev, int n, const float *Th, int maxlevel,
                      cusk_stats *stats);
int cusk_run_hetcor(cusk_engine *e, const float *C_dev, const float *N_dev, float ess_uniform,
                    const int *G_init_dev, int n, float th, int maxlevel, const int *time_index,
                    cusk_stats *stats);

/* Batched Skeleton run: `nblk` independent blocks (the LD blocks of one GPU's share of a chromosome job, or their reduced
 * stage-two sets) swept in ONE run.  The reference has no counterpart: it runs one block per process (src/cli.cpp:507-512,
 * README.md:62); per block the result is that of cusk_run_skeleton on the block alone (adjacency, separating sets).
 * Layout: the blocks lie along the diagonal of one n x n allocation C_dev (leading dimension n): block b holds the
 * variables [lo[b], hi[b]), its matrix at C_dev[i * n + j] for i, j in that range, each bitwise symmetric (what
 * cusk_corr_build_batch and cusk_gather_rows write); lo[b] are ascending multiples of 64, the ranges disjoint; variables
 * outside every range are padding and elements outside the diagonal blocks are never read (they need not be initialised).
 * Th / maxlevel as cusk_run_skeleton (one threshold array: the blocks share the sample size).  stats are those of the
 * whole batch; `level` is the last level at which ANY block still had a row with more neighbours than the level (a block
 * that ends earlier has no test left at the later levels, so its result does not depend on the others).  Results:
 * cusk_result_adj_bits_blocks, cusk_result_sepsets (x, y and set members are variable indices of the batch). */
int cusk_run_skeleton_batch(cusk_engine *e, const float *C_dev, int n, int nblk, const int *lo, const int *hi,
                            const float *Th, int maxlevel, cusk_stats *stats);
/* adjacency of the last batched run, block by block: rows lo..hi-1 of block b, each cut to the (hi - lo + 63) / 64
 * words of the block's own columns (bit j of a row = local variable j), blocks back to back.  out_host: room for
 * sum_b (hi[b] - lo[b]) * ((hi[b] - lo[b] + 63) / 64) words. */
int cusk_result_adj_bits_blocks(cusk_engine *e, uint64_t *out_host);
/* the same for the LAST tail_rows rows of every block only (the traits: parent_set.cpp:8-53 at depth 1 looks at nothing
 * else); a block with fewer rows gives all of them */
int cusk_result_adj_bits_blocks_tail(cusk_engine *e, int tail_rows, uint64_t *out_host);
/* rows [row0, row0 + nrows) of the last run's bitmap, cusk_result_words() words each, to host memory */
int cusk_result_adj_rows(cusk_engine *e, int row0, int nrows, uint64_t *out_host);
/* Many sub-matrices in one launch: out[row_out[t] + c] = M_dev[row_src[t] * n + idx[row_first[t] + c]] for c < row_k[t],
 * t < nrows (parent_set.cpp:84-238 for a batch: the stage-two matrices straight onto the diagonal of the next batch
 * allocation, out_on_device = 1; the retained sub-matrices of the results, out_on_device = 0, out_count floats).  Index
 * arrays are host memory. */
int cusk_gather_rows(cusk_engine *e, const float *M_dev, int n, const int *idx_host, long long nidx, const int *row_src,
                     const int *row_k, const long long *row_first, const long long *row_out, long long nrows, float *out,
                     long long out_count, int out_on_device);

/* Results of the last run (valid until the next run / destroy). */
int cusk_result_n(const cusk_engine *e);
/* adjacency bitmap on the device: n rows of cusk_result_words() uint64 words, bit j of row i */
const uint64_t *cusk_result_adj_bits_dev(const cusk_engine *e);
int cusk_result_words(const cusk_engine *e);
/* expand into the reference's layouts; dst is HOST memory unless the name says _dev */
int cusk_result_adj_i32(cusk_engine *e, int *G_host);
int cusk_result_adj_i32_dev(cusk_engine *e, int *G_dev);
int cusk_result_pmax(cusk_engine *e, const float *C_dev, float *pMax_host);
int cusk_result_sepset_dense(cusk_engine *e, int *SepSet_host /* n*n*14 */);
/* sparse separation sets: one record per ordered pair with a non-empty set, ordered by (x, y).
 * Returns the count; arrays may be NULL to query it.  x,y: count ints;
 * level: count ints; z: count floats; S: count*14 ints (-1 padded).  The winners' Fisher z is computed on the first
 * request that asks for it (z != NULL, or cusk_result_pmax) from the matrix of the run, which must still be resident. */
long long cusk_result_sepsets(cusk_engine *e, int *x, int *y, int *level, float *z, int *S);
/* the same records (x, y, S[count * 14]) in engine-owned PINNED host memory -- one synchronisation, no staging through
 * pageable memory; the pointers stay valid until the next run or result call on this engine.  Returns the count. */
long long cusk_result_sepsets_view(cusk_engine *e, const int **x, const int **y, const int **S);

/* Correlation build on the device (a2-a5).  bed/phen/mean/std
 * are HOST buffers as in cu_corr_pearson_npn; the n*n (n = m + p) square
 * matrix (markers first, then traits, unit diagonal, symmetric;
 * src/cli.cpp:597-649) is written to C_dev.  If mxp_host is not NULL it also
 * receives the m*p marker-trait correlations (for the prefilter, cli.cpp:561-576). */
int cusk_corr_build(cusk_engine *e, const unsigned char *bed, const float *phen, size_t m,
                    size_t N, size_t p, const float *mean, const float *std, float *C_dev,
                    float *mxp_host);
/* The same build for the NEXT block of a job while the current one is swept: begin enqueues it on a stream of its own
 * (device-resident inputs only: cusk_blockset_stage) and returns at once, end waits for it and copies the marker x trait
 * correlations out.  One build in flight per engine; the matrix must not be read before end returns. */
int cusk_corr_build_begin(cusk_engine *e, const unsigned char *bed_dev, const float *phen_dev, size_t m, size_t N, size_t p,
                          const float *mean_dev, const float *std_dev, float *C_dev);
int cusk_corr_build_end(cusk_engine *e, float *mxp_host);
int cusk_corr_build_pending(const cusk_engine *e); /* 1 while a cusk_corr_build_begin has not been ended */
/* The correlation matrices of MANY LD blocks in one set of launches, written onto the diagonal of the n x n batch
 * allocation C_dev that cusk_run_skeleton_batch sweeps (cli.cpp:543-649 once per block in the reference).  Inputs are the
 * device-resident arrays of cusk_blockset_stage: bed_dev = marker 0 of the file set (ceil(N/4) bytes per marker),
 * mean_dev / std_dev indexed by global marker, phen_dev column-major p x N.  Block b = markers first_marker[b] ..
 * first_marker[b] + markers[b] - 1, variables base[b] .. base[b] + markers[b] + p - 1 of the allocation (markers, then the
 * p traits; base[b] a multiple of 64).  Host arrays of nblk entries.
 *   _mxp : marker x trait correlations of every block, into C_dev (both triangles) and, block after block, into the host
 *          array mxp_host (sum markers[b] x p floats, row-major) for the prefilter (cli.cpp:561-576); returns when they
 *          are there.
 *   _mxm : marker x marker (Kendall-npn on the FP4 matrix pipe), trait x trait and the unit diagonal of the blocks with
 *          keep[b] != 0 (keep = NULL: all); asynchronous on the engine's stream. */
/* both in one call, _mxm speculatively for EVERY block: returns when the marker x trait correlations are in mxp_host,
 * with the marker x marker part still running on the engine's stream (the caller's prefilter overlaps it) */
int cusk_corr_build_batch(cusk_engine *e, const unsigned char *bed_dev, const float *phen_dev, const float *mean_dev,
                          const float *std_dev, size_t N, size_t p, int nblk, const long long *first_marker,
                          const int *markers, const int *base, int n, float *C_dev, float *mxp_host);
int cusk_corr_build_batch_mxp(cusk_engine *e, const unsigned char *bed_dev, const float *phen_dev, const float *mean_dev,
                              const float *std_dev, size_t N, size_t p, int nblk, const long long *first_marker,
                              const int *markers, const int *base, int n, float *C_dev, float *mxp_host);
int cusk_corr_build_batch_mxm(cusk_engine *e, const unsigned char *bed_dev, const float *phen_dev, const float *mean_dev,
                              const float *std_dev, size_t N, size_t p, int nblk, const long long *first_marker,
                              const int *markers, const int *base, const unsigned char *keep, int n, float *C_dev);
/* timing of the last cusk_corr_build: [0] decode, [1] count GEMM, [2] mxp/pxp, [3] total (ms) */
void cusk_corr_timing(const cusk_engine *e, float *ms4);

/* `mps block` (cli.cpp:362-411): cal_mcorrk_banded + marker_corr_banded_mat_row_abs_sums, corr_host.cu:65-128.
 * bed: packed genotypes of the m markers of ONE chromosome, SNP-major, ceil(N/4) bytes each (host).  Computes the
 * banded Kendall-npn correlations band[row * width + col] = corr(row, row + 1 + col) (0 where row + 1 + col >= m) on
 * the device and their forward row sums of absolute values (float accumulation in column order, as the reference's
 * host loop).  rowsums_host: m floats out.  band_host: m * width floats out, or NULL. */
int cusk_corr_banded(cusk_engine *e, const unsigned char *bed, size_t m, size_t N, size_t width, float *rowsums_host,
                     float *band_host);
/* hanning_smoothing, blocking.cpp:13-35: out[c] = sum_i weight[i] * v[c - window/2 + i] for window/2 <= c < n - window/2,
 * 0 elsewhere; double accumulation in window order on the device (bit-identical to the host loop).  The caller supplies
 * the window weights (0.5 - 0.5 cosf(2 pi i / (window - 1)), blocking.cpp:8-11).  All pointers are host memory. */
int cusk_hanning_smooth(cusk_engine *e, const float *v_host, size_t n, const double *weight_host, int window,
                        double *out_host);

/* `sepselect` / `orient-v-structs` hot loop: cusk_postprocessing/sepselect.py:262-329
 * (MergedCuskResults.find_maximal_and_min_pcorr_sepsets_incr) for a batch of outer pairs, one wavefront per pair.
 * All pointers are HOST memory.
 *   trait_corr  n x p doubles, row-major: corr[v, t] of every variable v with every trait t (merged layout: the
 *               traits are variables 0 .. p-1); the matrix must be symmetric on these entries
 *   pair_i/j    the outer pairs (variable indices), pair_corr[k] = corr[pair_i[k], pair_j[k]]
 *   cand_off    npairs + 1 offsets into cand; cand = the trait neighbours of pair_i[k] in the order the
 *               reference's loop visits them (iteration order of its Python set); ties of the minimum go to the
 *               LAST candidate in that order, as `<=` does at sepselect.py:286
 *   thr         thr[l] = norm.ppf(1 - alpha/2) / sqrt(num_samples - l - 3) (sepselect.py:25-26), l = 0 .. nthr-1,
 *               nthr > longest candidate list
 * Out: sel (cand_off layout) = the accepted traits of pair k in order, sel_len[k] of them (the maximal separating
 * set, :311); flags[k] bit 0 = the round-wise minimum of the partial correlation was passed (:292-294: the pair
 * gets an entry in min_pcorr_sepsets), flags[k] >> 8 = 0 ok, 1 = no candidate had a comparable z (the reference
 * fails on remove(None) there), 2 = a sub-matrix was exactly singular (the reference exits, :12-18).
 * kernel_ms (may be NULL): device time of the selection kernels alone. */
int cusk_sepselect_greedy(cusk_engine *e, const double *trait_corr, long long n, int p, long long npairs,
                          const int *pair_i, const int *pair_j, const double *pair_corr, const long long *cand_off,
                          const int *cand, const double *thr, int nthr, int *sel, int *sel_len, int *flags,
                          float *kernel_ms);

/* out_host[a*k + b] = M_dev[idx[a]*n + idx[b]]: the retained sub-matrix of parent_set.cpp:84-238
 * (reduce_gc / reduce_gcs) without copying the n*n matrix to the host; idx_host has k entries. */
int cusk_gather_submatrix(cusk_engine *e, const float *M_dev, int n, const int *idx_host, int k, float *out_host);
/* the same with the k*k result left on the device (the stage-two matrix of reduced_gcs_cusk, cli.cpp:62-87, never
 * visits the host) */
int cusk_gather_submatrix_dev(cusk_engine *e, const float *M_dev, int n, const int *idx_host, int k, float *out_dev);

/* ---------------------------------------------------------------------------
 * (3) block driver: many LD blocks from one process
 * ------------------------------------------------------------------------ */

/* The reference processes one LD block per `mps cusk` invocation (src/cli.cpp:507-512, README.md:62) and leaves the
 * loop over blocks to the job scheduler.  A block set holds what every such invocation loads again -- .phen, the
 * <bfiles>.dim/.bim/.means/.stds, the .blocks file, the thresholds (cli.cpp:458-497) -- once, maps the .bed, and runs
 * any of its blocks through exactly the code `mps cusk` runs for that block (host/block_pipeline.h), on whichever
 * engine (= GPU) the caller hands in.  This is what the multi-GPU driver (ci-gwas_amd/run_blocks.py: one process per
 * GPU, longest-processing-time assignment of blocks to ranks, one gather of the per-block results) is built on.
 * A block set is read-only after open: several threads may run different blocks at once, each with its own engine. */
typedef struct cusk_blockset cusk_blockset;
typedef struct cusk_block_result cusk_block_result;

typedef struct cusk_block_stats {
    int skipped;            /* 1: no marginally significant marker-trait correlation (cli.cpp:561-576), no result */
    int num_sig;            /* marker-trait correlations at or above Th[0] */
    long long markers;      /* markers of the block */
    long long retained;     /* markers in the result */
    long long tests[2];     /* CI tests of stage one / stage two */
    double ms_inputs, ms_corr, ms_stage1, ms_prune, ms_stage2, ms_reduce; /* wall-clock phases of this block */
    cusk_stats stage[2];    /* engine counters of both stages */
} cusk_block_stats;

/* Same arguments as `mps cusk` without outdir and block index (cli.cpp:432-456).  On failure returns an error code
 * and writes the message (what `mps cusk` would have printed before exit(1)) to err (may be NULL). */
int cusk_blockset_open(cusk_blockset **out, const char *phen_path, const char *bfiles, const char *blocks_path,
                       float alpha, int max_level, int max_level_two, int depth, char *err, size_t err_len);
void cusk_blockset_close(cusk_blockset *bs);
int cusk_blockset_num_blocks(const cusk_blockset *bs);
long long cusk_blockset_num_samples(const cusk_blockset *bs);
int cusk_blockset_num_phen(const cusk_blockset *bs);
/* markers of block i and its output file stem "<chr>_<first>_<last>" (marker_block.h:36-60) */
long long cusk_blockset_block_markers(const cusk_blockset *bs, int block_index);
int cusk_blockset_block_stem(const cusk_blockset *bs, int block_index, char *stem, size_t stem_len);
/* Copies the inputs every block reads -- the packed genotypes of the whole .bed, the phenotypes, the marker means and
 * standard deviations -- to e's device once (288 GB of HBM hold a whole-genome .bed); blocks run on any engine of that
 * device then build their correlations straight from HBM (cusk_corr_build accepts device pointers) instead of copying
 * their slice over PCIe every time.  Optional: without it every block uploads its own slice.  Returns
 * CUSK_ERR_HIP (and stages nothing) when the device memory is not there.  Call before running blocks on that device. */
int cusk_blockset_stage(cusk_blockset *bs, cusk_engine *e);
/* cli.cpp:521-677 for one block on e's device.  *out receives the result (NULL when the block is skipped);
 * stats may be NULL.  Error text: cusk_blockset_last_error (per calling thread). */
int cusk_blockset_run_block(cusk_blockset *bs, cusk_engine *e, int block_index, cusk_block_result **out,
                            cusk_block_stats *stats);
/* the same, naming the block this engine runs next: its correlation build (cusk_corr_build_begin) runs beside this block's
 * sweeps when the inputs are staged in HBM; next_index = -1: none */
int cusk_blockset_run_block_next(cusk_blockset *bs, cusk_engine *e, int block_index, int next_index,
                                 cusk_block_result **out, cusk_block_stats *stats);
const char *cusk_blockset_last_error(void);
/* Forgets what the block set keeps for engine e -- its device scratch (block matrices) and the state of a correlation
 * build started ahead -- and releases that memory.  Call before destroying an engine that ran blocks of this set when the
 * set outlives it (cusk_blockset_close releases everything anyway). */
void cusk_blockset_release_engine(cusk_blockset *bs, cusk_engine *e);

/* MANY blocks in one set of device runs (host/batch_pipeline.h): the correlation matrices of all `nblocks` blocks are built
 * by one set of launches, both skeleton stages sweep them in one level loop each (cusk_run_skeleton_batch), one read-out
 * brings the reduced results back.  The reference runs one block per process (cli.cpp:507-512, README.md:62); per block
 * the result -- and every file written from it -- is the one of cusk_blockset_run_block (= `mps cusk` on that block).
 * Needs the inputs on e's device (cusk_blockset_stage; done here when it has not been).  The padded variable count of
 * the batch, sum over blocks of (markers + traits) rounded up to 64, squared, times 4 bytes is the device memory of its
 * matrix: the caller chooses the batches (ci-gwas_amd/run_blocks.py: --batch-vars).  stats may be NULL. */
typedef struct cusk_batch_result cusk_batch_result;
typedef struct cusk_batch_stats {
    int blocks, skipped;                 /* blocks asked for / skipped by the prefilter (cli.cpp:561-576) */
    long long markers, retained;         /* markers of all blocks / markers in the results */
    long long vars_stage1, vars_stage2;  /* padded variable counts of the two batch allocations */
    long long tests[2], canonical[2];    /* executed / canonical CI tests of stage one and two */
    double ms_corr, ms_stage1, ms_prune, ms_stage2, ms_reduce; /* wall-clock phases of the batch */
    cusk_stats stage[2];                 /* engine counters of both stages (whole batch) */
} cusk_batch_stats;
int cusk_blockset_run_batch(cusk_blockset *bs, cusk_engine *e, const int *block_indices, int nblocks,
                            cusk_batch_result **out, cusk_batch_stats *stats);
int cusk_batch_result_count(const cusk_batch_result *r);                       /* blocks with a result */
int cusk_batch_result_block_index(const cusk_batch_result *r, int i);          /* index of result i in the .blocks file */
const cusk_block_result *cusk_batch_result_block(const cusk_batch_result *r, int i); /* borrowed; accessors below */
int cusk_batch_result_write(const cusk_batch_result *r, const char *outdir);   /* every block's five files */
/* the results as one byte string (the multi-GPU job gathers these to rank 0): per block six int32 {block index, num_var,
 * num_phen, max_level, 1, length of the stem} + stem + .ixs + .adj + .corr + .sep contents */
size_t cusk_batch_result_packed_bytes(const cusk_batch_result *r);
int cusk_batch_result_pack(const cusk_batch_result *r, void *buf, size_t bytes);
int cusk_packed_results_write(const void *buf, size_t bytes, const char *outdir, int *blocks_written);
/* the same byte string without the .sep arrays (nine tenths of the bytes; with_sep = 0): what the merge below needs.
 * with_sep = 2: the separating sets as a list instead of the dense num_var^2 x max_level array -- head word 4 = 2, then an int32
 * count and count records {int32 row, column, length, int32 members[14]} in ascending (row, column) order; understood by
 * cusk_packed_results_write and cusk_merge_packed (not by the reference-side Python unpacker: use it between library calls). */
size_t cusk_batch_result_packed_bytes_ex(const cusk_batch_result *r, int with_sep);
int cusk_batch_result_pack_ex(const cusk_batch_result *r, void *buf, size_t bytes, int with_sep);
/* `merge-block-outputs` (cusk_postprocessing/merge_blocks.py:361-395 + write_mm :298-325) on the packed results of a whole
 * job, in memory: writes <basepath>_sam.mtx, <basepath>_scm.mtx, <basepath>.mdim, <basepath>.ixs -- the files the
 * reference's merge writes from the per-block files, byte for byte.  blockfile = the job's .blocks file (order of the
 * blocks; a listed block without a result counts as skipped, as a block without files does there). */
int cusk_merge_packed(const char *blockfile, const void *buf, size_t bytes, const char *basepath);
void cusk_batch_result_free(cusk_batch_result *r);
/* the reduced result of one block: what ReducedGCS::to_file writes (include/mps/parent_set.h:42-52) */
void cusk_block_result_dims(const cusk_block_result *r, long long *num_var, long long *num_phen, long long *max_level);
const char *cusk_block_result_stem(const cusk_block_result *r);
const int *cusk_block_result_ixs(const cusk_block_result *r);    /* num_var */
const int *cusk_block_result_adj(const cusk_block_result *r);    /* num_var^2 */
const float *cusk_block_result_corr(const cusk_block_result *r); /* num_var^2 */
const int *cusk_block_result_sep(const cusk_block_result *r);    /* num_var^2 * max_level */
/* <outdir>/<stem>.mdim .ixs .adj .corr .sep */
int cusk_block_result_write(const cusk_block_result *r, const char *outdir);
void cusk_block_result_free(cusk_block_result *r);

/* device -> host copy ordered after the engine's work; waits for the engine's stream only */
int cusk_engine_download(cusk_engine *e, void *dst_host, const void *src_dev, size_t bytes);

/* device memory helpers so that C hosts need no HIP headers */
void *cusk_dev_alloc(size_t bytes);
void cusk_dev_free(void *p);
int cusk_dev_upload(void *dst_dev, const void *src_host, size_t bytes);
int cusk_dev_download(void *dst_host, const void *src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* CUSK_HIP_H_ */
