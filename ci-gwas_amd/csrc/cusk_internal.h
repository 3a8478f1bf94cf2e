// cusk_internal.h -- engine state shared by the translation units of libcusk_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/cusk_hip.h"

namespace cusk {

constexpr int kML = CUSK_ML;
constexpr int kLevels = CUSK_ML + 2;  // per-level device records, index = level
constexpr unsigned long long kNone = ~0ull;
// degree classes of the level sweep: rows whose (d+1)^2 sub-matrix fits the class
// capacity are staged in LDS; the last class reads C from global memory.
// The first class (up to 39 neighbours: every row of an LD block from level 2 on) is swept with ONE WAVEFRONT per work
// item (sweep_vec.hip) and a smaller work item: its sub-matrix takes 8 KB of LDS, so a CU holds 16-20 items at once
// instead of 4-8 and the per-item chain of dependent loads (item -> offsets -> list -> operands) is hidden by other
// items rather than paid serially.
constexpr int kNumClasses = 5;
constexpr int kClassCap[kNumClasses] = {39, 63, 127, 191, 1 << 30};
constexpr int kThreads = 256;
// first level swept by unions T = S + Y (sweep_tmaj.hip); measured on stage two of the 10k block (d = 39): level 5
// 0.32 ms against 0.21 ms for the set-major float2 kernel, level 6 0.83 against 0.94, level 7 2.1 against 4.2, level 8
// 6.7 against 16.8, level 14 1.3 s against 3.2 s
constexpr int kTmajMinLevelDefault = 6;
constexpr int kCounterSlots = 512;  // per-workgroup counter atomics are spread over this many slots
constexpr size_t kLdsLimit = 160 * 1024;

struct DevBuf
{
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T *as() const
    {
        return reinterpret_cast<T *>(p);
    }
};

struct LevelCounters;  // sweep_common.h
struct HostGate;       // sweep_common.h

}  // namespace cusk

struct cusk_engine
{
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // auxiliary stream: independent degree classes, off-critical-path kernels
    hipStream_t stream3 = nullptr;  // correlation build of the NEXT block while this one is swept (cusk_corr_build_begin / _end)
    float *mxp_pinned = nullptr;    // pinned landing buffer of the prefetched marker x trait correlations
    size_t mxp_pinned_cap = 0, mxp_pending = 0;  // floats; mxp_pending > 0: a build is in flight on stream3
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_z = nullptr;
    hipEvent_t ev_mxp = nullptr;  // cusk_corr_build_batch: the marker x trait correlations have landed in mxp_pinned
    bool own_stream = false;
    std::string err;

    // result / working state of the last run
    int n = 0;
    int words = 0;
    int mode = -1;  // 0 Skeleton, 1 hetcor
    bool have_result = false;
    cusk::DevBuf adj, adj0;  // uint64 n*words: live adjacency, adjacency after level 0
    cusk::DevBuf deg, binom, counters, slots, rec_base, ti, queue, symflag;
    // per-level working sets, ping-pong by level parity
    cusk::DevBuf off[2], nbr[2], best[2];
    cusk::DevBuf planblk;  // published block totals of the plan kernel
    unsigned plan_seq = 0;
    cusk::DevBuf off1;  // CSR offsets of level 1 (kept for the whole run: records live at level-1 slots)
    cusk::DevBuf items[2][cusk::kNumClasses];
    // sparse record store, indexed by level-1 CSR slot (rec_l = 0: empty); rec_s is member-major with stride rec_cap
    cusk::DevBuf rec_x, rec_y, rec_l, rec_s;
    long long rec_cap = 0;    // slots allocated
    long long rec_slots = 0;  // slots of the last run (= directed edges after level 0)
    // dense record list, produced on the first result request after a run (materialize_records)
    cusk::DevBuf den_x, den_y, den_l, den_z, den_s, den_counts, den_off;
    long long den_stride = 0;
    bool records_ready = false, z_ready = false;
    cusk::DevBuf scratch_a, scratch_b;  // result read-out / gather scratch (no hipMalloc / hipFree per call)
    const float *last_C = nullptr;  // matrix of the last run (the winners' z is computed from it on request)
    int last_levels = 0;
    // batched runs (cusk_run_skeleton_batch): block table of the last run and its per-row view on the device
    std::vector<int> batch_lo, batch_hi;
    cusk::DevBuf row_range, row_blk, blk_woff;
    void *batch_pinned = nullptr;  // staging of the per-row tables
    size_t batch_pinned_cap = 0;
    void *res_pinned = nullptr;    // cusk_result_sepsets_view: x, y, S of the records
    size_t res_pinned_cap = 0;
    cusk::DevBuf rv, rpos, sel, wpre;  // level 1, row-streaming kernel: C[X, adj(X)] and {Y, reverse position, off, deg} per CSR slot
    long long nrec = 0;
    // pinned host mirrors
    cusk::LevelCounters *hcnt = nullptr;   // kLevels entries
    unsigned long long *hslots = nullptr;  // kLevels * kCounterSlots * 4
    long long *hrec_base = nullptr;        // kLevels + 1
    unsigned long long *hcanon = nullptr;  // kLevels * kCounterSlots: canonical test counts
    int *hflag = nullptr;
    std::vector<unsigned long long> binom_host;
    long long binom_rows = 0;  // rows of the device-resident binomial table
    // per level: what the level's plan kernel tells the host, written straight into pinned host memory (the host follows
    // the device at a distance, engine.hip); run_seq tags the entries of the current run
    cusk::HostGate *hgate = nullptr, *hgate_dev = nullptr;
    int run_seq = 0;
    long long item_cap_cur = 0;         // capacity (entries) of the per-class work-item buffers
    hipEvent_t ev_main[2] = {nullptr, nullptr};  // around the level-1 rows kernel alone
    hipEvent_t ev_run[2] = {nullptr, nullptr};
    hipEvent_t ev_k0[cusk::kLevels], ev_k1[cusk::kLevels], ev_l0[cusk::kLevels], ev_l1[cusk::kLevels];

    // options (cusk_engine_set_option)
    int opt_fast = 1;
    int opt_validate = 0;
    int opt_corr_fp4 = 1;
    int opt_pair = 1;
    int opt_rows = 1;
    int opt_vec = 1;
    int opt_overlap = 1;
    int opt_corr_popcount = 0;
    int opt_corr_mxp_f32 = 0;  // 1: SNP x trait on the f32 matrix instructions (round 1-2 form) instead of the bf16 split
    int opt_assume_symmetric = 0;
    long long opt_queue_cap = 4ll << 20;
    int opt_hostprof = 0;
    int opt_tmaj_validate_stride = 1;  // validating builds of sweep_tmaj check the unions whose per-lane count is a multiple of this (power of two)
    int opt_max_staged_classes = -1;  // >= 0: at most this many degree classes are staged in LDS (test hook for the unstaged kernels)
    int opt_tmaj_min_level = cusk::kTmajMinLevelDefault;  // first level swept by unions T = S + Y (sweep_tmaj.hip); 99 = never
    long long opt_chunk = 2048;
    long long opt_chunk0 = 512;   // conditioning sets per work item of the first degree class
    long long opt_chunk0_low = 256;  // ... at levels 2-4
    int opt_vec_threads = 64;     // workgroup size of sweep_vec_kernel for the first degree class (64 / 128 / 256)
    long long opt_item_cap = 1ll << 20;  // work items per degree class and level the buffers hold before they are grown
    int opt_lookahead = 2;               // levels the host may enqueue ahead of the counters it has seen
    int opt_sync2 = 1;                   // the host reads level 2's gate record before it enqueues that level's sweeps (engine.hip)
    int opt_l1_exp = 0;                  // level-1 row kernel experiment bits (sweep_level.hip: RowsParams::exp)
    int opt_timing = 1;                  // per-level HIP events for cusk_stats' kernel_ms / level_ms (0: total only)
    long long opt_sep_ws_budget = 4ll << 30;  // HBM work space of cusk_sepselect_greedy for candidate lists beyond LDS

    // row-sharded sweep of ONE block over several engines: this engine runs the tests of rows
    // X with X % shard_world == shard_rank and joins the others through an element-wise unsigned MIN of the
    // per-slot selection state after every level's sweep
    int shard_rank = 0, shard_world = 1;
    int shard_host_staging = 0;
    cusk_exchange_fn shard_fn = nullptr;
    void *shard_user = nullptr;
    void *shard_host = nullptr;  // pinned staging buffer
    size_t shard_host_cap = 0;

    // correlation build scratch
    cusk::DevBuf bed_dev, phen_dev, mean_dev, std_dev, planes, mxp_dev, pxp_dev, mxp_bq;
    cusk::DevBuf corr_tab[2];  // batched build: block / tile tables of phase one (marker x trait) and two (marker x marker)
    void *corr_tab_pinned[2] = {nullptr, nullptr};
    size_t corr_tab_pinned_cap[2] = {0, 0};
    hipEvent_t ev_corr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float corr_ms[4] = {0, 0, 0, 0};
};

namespace cusk {

inline int fail(cusk_engine *e, int code, const std::string &msg)
{
    if (e) e->err = msg;
    return code;
}

#define CUSK_HIP(e, call)                                                                        \
    do                                                                                           \
    {                                                                                            \
        hipError_t _st = (call);                                                                 \
        if (_st != hipSuccess)                                                                   \
            return cusk::fail((e), CUSK_ERR_HIP,                                                 \
                              std::string(#call) + ": " + hipGetErrorString(_st) + " (" +        \
                                  __FILE__ + ":" + std::to_string(__LINE__) + ")");              \
    } while (0)

// host-side helpers shared by the entry points
void threshold_array_host(int n, float alpha, float *thr15);
float hetcor_threshold_host(float alpha);

// corr_build.hip
int corr_build_impl(cusk_engine *e, const unsigned char *bed, const float *phen, size_t m, size_t N,
                    size_t p, const float *mean, const float *std, float *C_dev, float *mxp_host,
                    float *mxm_tri_host, float *pxp_tri_host, bool ahead = false);

// reference-named correlation entry points (corr_build.hip); C linkage in compat_api.hip, C++ linkage in compat_cxx.cpp
void compat_marker_phen_corr_pearson(const unsigned char *marker_vals, const float *phen_vals, const size_t num_markers,
                                     const size_t num_individuals, const size_t num_phen, const float *marker_mean,
                                     const float *marker_std, float *marker_phen_corrs);
void compat_corr_pearson_npn(const unsigned char *marker_vals, const float *phen_vals, const size_t num_markers,
                             const size_t num_individuals, const size_t num_phen, const float *marker_mean,
                             const float *marker_std, float *marker_corrs, float *marker_phen_corrs, float *phen_corrs);
double qnorm_host(double p);

}  // namespace cusk
