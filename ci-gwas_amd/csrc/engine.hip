// engine.hip -- host side of the level sweep: the level loop and the cusk_* C ABI.
//
// Level loop of /root/reference/cusk/src/cuPC-S.cu:99-415 and hetcor-cuPC-S.cu:115-332,
// restructured for one host synchronisation per level:
//   level 0 : bitmap build (ballots), symmetry flag
//   level l : scan of the live degrees -> CSR offsets, wave-per-row compaction + work-item
//             count, per-class item scan, ONE sync (max degree for the termination test,
//             item counts for the grids, previous level's recheck-queue fill),
//             sweeps (pair / fast+recheck / exact), separating-set finalisation.
// Degrees are maintained incrementally (every cleared adjacency bit decrements one), the
// recheck pass reads its entry count on the device, separating-set records are placed by
// scans (deterministic order), per-level counters live in per-level device slots and are read
// back once at the end.  Two CSR working sets alternate by level parity so that a level
// whose recheck queue overflowed can be redone on the exact path after the fact.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "sweep_common.h"

namespace cusk {

static unsigned long long binom_sat(int n, int k)
{
    if (k < 0 || k > n) return 0;
    if (k > n - k) k = n - k;
    unsigned long long r = 1;
    const unsigned long long cap = 1ull << 62;
    for (int i = 1; i <= k; i++)
    {
        unsigned long long f = (unsigned long long)(n - k + i);
        if (r > cap / f) return cap;
        r = r * f / (unsigned long long)i;
    }
    return r;
}

// loc_th of the hetcor engine when every pair has the same effective sample size:
// the float running sum of (int)ess over (l+2)(l+1)/2 pairs, as mean_ess forms it
// (hetcor-cuPC-S.cu:3068-3088), then th / sqrt(mean_ess - l - 3) in double (:471,:612).
static float uniform_ess_threshold(float th, float ess, int l)
{
    int t;
    if (ess != ess)
        t = 0;
    else if (ess >= 2147483648.0f)
        t = 2147483647;
    else if (ess <= -2147483648.0f)
        t = (-2147483647 - 1);
    else
        t = (int)ess;
    const int pairs = (l + 2) * (l + 1) / 2;
    float s = 0.0f;
    for (int i = 0; i < pairs; i++) s += (float)t;
    const float me = s / (float)pairs;
    return (float)((double)th / std::sqrt((double)me - (double)l - 3.0));
}

struct RunArgs
{
    int mode;  // 0 Skeleton, 1 hetcor
    const float *C;
    const float *Ness;      // hetcor, may be null (uniform)
    float ess_uniform;      // hetcor uniform ESS
    const int *Ginit;       // hetcor, device n*n or null
    const float *Th;        // mode 0: host thresholds ; mode 1: Th[0] = alpha/2 quantile
    const int *time_index;  // host, n entries or null
    int n;
    int maxlevel;
    // batched run: per row the column range of its block (device, n entries) and the largest range; nullptr = one block
    const int2 *row_range = nullptr;
    int max_span = 0;
    long long level0_pairs = 0;  // unordered pairs inside the blocks (the level-0 tests of a batched run)
};

// layout of the per-run control block (device) and of its pinned mirror (host)
constexpr size_t kCtlCnt = 0;
constexpr size_t kCtlSlots = (kCtlCnt + sizeof(LevelCounters) * kLevels + 15) & ~(size_t)15;
constexpr size_t kCtlRecBase = (kCtlSlots + sizeof(unsigned long long) * kLevels * kCounterSlots * 4 + 15) & ~(size_t)15;
constexpr size_t kCtlCanon = (kCtlRecBase + sizeof(long long) * (kLevels + 1) + 15) & ~(size_t)15;
constexpr size_t kCtlSym = (kCtlCanon + sizeof(unsigned long long) * kLevels * kCounterSlots + 15) & ~(size_t)15;
constexpr size_t kCtlBytes = kCtlSym + 16;

struct LevelPlan
{
    SweepParams sp;
    FinalizeParams fp;
    long long nitems[kNumClasses];
    bool use_pair = false, use_fast = false, use_rows = false, filter_ok = true;
    size_t pair_lds = 0;
    bool redone = false;
    bool known_items = false;  // nitems[] holds the level's real class counts (row-sharded runs wait for them)
    int maxdeg_bound = 0;      // no row has more neighbours than this (newest maximum degree the host has seen)
    int staged_classes = 0;
    unsigned long long chunk0 = 0;  // conditioning sets per work item of the first degree class at this level
    bool tmaj = false;         // swept by unions T = S + Y (sweep_tmaj.hip): the work items count (l + 1)-subsets
    bool force_exact = false;  // the level is taken up again on the exact arithmetic (recheck queue overflow of a tmaj level)
};

static int run_levels(cusk_engine *e, const RunArgs &a, cusk_stats *st)
{
    const int n = a.n;
    if (n <= 0 || a.C == nullptr || a.Th == nullptr) return fail(e, CUSK_ERR_ARG, "bad arguments");
    // option hostprof: host-side phase marks of the run (microseconds since entry) on stderr
    const auto hp_t0 = std::chrono::steady_clock::now();
    std::string hp_log;
    auto hp_mark = [&](const char *what) {
        if (!e->opt_hostprof) return;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - hp_t0).count();
        char buf[64];
        std::snprintf(buf, sizeof(buf), " %s=%.1f", what, us);
        hp_log += buf;
    };
    CUSK_HIP(e, hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const int words = (n + 63) / 64;
    e->n = n;
    e->words = words;
    e->mode = a.mode;
    e->have_result = false;
    e->nrec = 0;
    cusk_stats local;
    std::memset(&local, 0, sizeof(local));
    const bool het = (a.mode == 1 && a.Ness != nullptr);
    const int last_level = std::min(kML, a.maxlevel);
    const bool sharded = e->shard_world > 1;
    if (sharded && !e->shard_fn) return fail(e, CUSK_ERR_ARG, "row sharding needs an exchange function");

    const size_t bm = sizeof(unsigned long long) * (size_t)n * words;
    CUSK_HIP(e, e->adj.ensure(bm));
    if (a.mode == 0) CUSK_HIP(e, e->adj0.ensure(bm));
    CUSK_HIP(e, e->deg.ensure(sizeof(int) * (size_t)n));
    // per-run control block (level counters, counter slots, record bases, symmetry flag): one allocation, one
    // memset, one read-back at the end (pinned mirror with the same layout)
    const size_t ctl_cnt = kCtlCnt, ctl_slots = kCtlSlots, ctl_recbase = kCtlRecBase, ctl_sym = kCtlSym;
    const size_t ctl_bytes = kCtlBytes;
    CUSK_HIP(e, e->counters.ensure(ctl_bytes));
    long long &item_cap = e->item_cap_cur;
    item_cap = std::max<long long>(item_cap, std::max<long long>(e->opt_item_cap, 1024));
    for (int k = 0; k < 2; k++)
    {
        CUSK_HIP(e, e->off[k].ensure(sizeof(int) * ((size_t)n + 1)));
        for (int c = 0; c < kNumClasses; c++) CUSK_HIP(e, e->items[k][c].ensure(sizeof(int2) * (size_t)item_cap));
    }
    CUSK_HIP(e, e->off1.ensure(sizeof(int) * ((size_t)n + 1)));
    {  // plan kernel: 8 words per 256-row block, validated by a 24-bit launch sequence number (cleared when it wraps or the buffer is new)
        const size_t need = sizeof(unsigned long long) * 8 * (((size_t)n + 255) / 256);
        const bool fresh = need > e->planblk.cap;
        CUSK_HIP(e, e->planblk.ensure(need));
        if (fresh || e->plan_seq >= 0xfffff0u)
        {
            CUSK_HIP(e, hipMemsetAsync(e->planblk.p, 0, e->planblk.cap, s));
            e->plan_seq = 0;
        }
    }
    const int run_seq = ++e->run_seq;
    e->records_ready = e->z_ready = false;
    e->rec_slots = 0;
    e->last_C = a.C;
    char *ctl = e->counters.as<char>();
    LevelCounters *dcnt = reinterpret_cast<LevelCounters *>(ctl + ctl_cnt);
    unsigned long long *dslots = reinterpret_cast<unsigned long long *>(ctl + ctl_slots);
    (void)ctl_recbase;
    unsigned long long *dcanon = reinterpret_cast<unsigned long long *>(ctl + kCtlCanon);
    int *dsym = reinterpret_cast<int *>(ctl + ctl_sym);
    CUSK_HIP(e, hipMemsetAsync(ctl, 0, ctl_bytes, s));
    if (a.mode == 1)
    {
        CUSK_HIP(e, e->ti.ensure(sizeof(int) * (size_t)n));
        if (a.time_index)
            CUSK_HIP(e, hipMemcpyAsync(e->ti.p, a.time_index, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
        else
            CUSK_HIP(e, hipMemsetAsync(e->ti.p, 0, sizeof(int) * (size_t)n, s));
    }

    CUSK_HIP(e, hipEventRecord(e->ev_run[0], s));
    // ---- level 0 ----
    CUSK_HIP(e, hipEventRecord(e->ev_l0[0], s));
    {
        float th0 = a.Th[0];
        if (a.mode == 1 && !het) th0 = (float)((double)a.Th[0] / std::sqrt((double)a.ess_uniform - 3.0));
        if (a.row_range)
            CUSK_HIP(e, launch_level0_batch(a.C, e->adj.as<unsigned long long>(), e->adj0.as<unsigned long long>(), e->deg.as<int>(), n,
                                            words, a.row_range, th0, s));
        else
        {
            CUSK_HIP(e, launch_level0(a.C, het ? a.Ness : nullptr, a.Ginit, e->adj.as<unsigned long long>(), n, words, th0,
                                      e->opt_assume_symmetric ? nullptr : dsym, s));
            // (Skeleton mode keeps the bitmap of level 0: the degree pass below writes the copy; without a level 1 nobody reads it
            // but the pMax read-out, which then needs the copy made here)
            if (a.mode == 0 && last_level < 1) CUSK_HIP(e, hipMemcpyAsync(e->adj0.p, e->adj.p, bm, hipMemcpyDeviceToDevice, s));
        }
    }
    CUSK_HIP(e, hipEventRecord(e->ev_l1[0], s));
    hp_mark("l0_enq");
    local.max_degree[0] = n - 1;
    local.edges[0] = (long long)n * (n - 1);
    local.tests[0] = a.row_range ? a.level0_pairs : (long long)n * (n - 1) / 2;
    local.canonical_tests[0] = local.tests[0];
    local.levels_run = 1;

    LevelPlan plan[kLevels];
    bool symmetric = false;
    long long cap_edges = 0;  // allocation bound for the CSR arrays (edges only shrink)
    int level_out = (a.maxlevel < 0) ? 0 : last_level + 1;
    int levels_swept = 0;
    bool rows_timed = false;
    // Gate of a level's finalisation and of the next level's plan: "the recheck queue held every uncertain test".  A
    // level that is run (or redone) on the exact path has no queue: ~0.
    unsigned long long qcap_gate[kLevels];
    for (auto &q : qcap_gate) q = ~0ull;

    const unsigned long long chunk = (unsigned long long)std::max<long long>(e->opt_chunk, 256);
    const unsigned long long chunk0 = (unsigned long long)std::max<long long>(e->opt_chunk0, 64);
    auto launch_level_sweeps = [&](int l, bool exact_only) -> int {
        LevelPlan &pl = plan[l];
        SweepParams sp = pl.sp;
        // The degree classes of a level are independent launches: the first class runs on the engine stream, the
        // others on the auxiliary stream (many tiny rows next to a few hub rows fill the chip better together than one
        // after the other); the streams join before the recheck pass.  Every launch is persistent: the number of work
        // items of its class is read on the device, so nothing here waits for the host (empty classes return at once).
        if (pl.use_rows && !exact_only)
        {
            CUSK_HIP(e, launch_level1_rows(a.mode, e->opt_validate != 0, pl.filter_ok && e->opt_fast != 0, sp, e->rv.as<float>(),
                                           e->rpos.p, e->sel.as<unsigned>(), e->wpre.as<int>(), e->opt_timing ? e->ev_main[0] : nullptr,
                                           e->opt_timing ? e->ev_main[1] : nullptr, e->shard_rank, e->shard_world, e->opt_l1_exp, sharded,
                                           a.time_index != nullptr, (a.mode == 1) ? dcanon + (size_t)l * kCounterSlots : nullptr, s));
            rows_timed = true;
            return CUSK_OK;
        }
        // classes that can hold work at this level: a row of degree d belongs to the first class with d <= cap, and no
        // degree exceeds the level-1 maximum
        int nonempty = 0;
        bool may[kNumClasses];
        for (int c = 0; c < kNumClasses; c++)
        {
            if (pl.known_items)
                may[c] = pl.nitems[c] > 0;
            else if (c < pl.staged_classes)
                may[c] = (c == 0 ? 0 : kClassCap[c - 1] + 1) <= pl.maxdeg_bound;
            else  // rows too large to stage all go to the last class
                may[c] = (c == kNumClasses - 1) && (pl.staged_classes == 0 || pl.maxdeg_bound > kClassCap[pl.staged_classes - 1]);
            nonempty += may[c];
        }
        const bool fork = (nonempty > 1) && (e->opt_overlap != 0);  // (a stale degree bound forks for classes that turn out empty)
        if (fork)
        {
            CUSK_HIP(e, hipEventRecord(e->ev_fork, s));
            CUSK_HIP(e, hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
        }
        bool first = true;
        for (int c = 0; c < kNumClasses; c++)
        {
            if (!may[c]) continue;
            hipStream_t cs = (fork && !first) ? e->stream2 : s;
            first = false;
            sp.items = e->items[l & 1][c].as<int2>();
            sp.cap = kClassCap[c];
            sp.cls = c;
            sp.chunk = (c == 0) ? pl.chunk0 : chunk;
            sp.item_cap = item_cap;
            sp.grid_cap = item_cap;
            if (l >= 2 && !pl.use_pair && !pl.tmaj)
            {
                // small blocks: no more workgroups than the class can have items -- at most n rows of at most dmax neighbours
                // with ceil(C(dmax, l) / chunk) items each (the bound of the degrees the host has seen, the class capacity)
                const int dmax = std::min(pl.maxdeg_bound, kClassCap[c]);
                const unsigned long long sets = binom_sat(dmax, l);
                const unsigned long long per_row = sets / sp.chunk + 1ull;
                if (per_row < (1ull << 40)) sp.grid_cap = std::min<long long>(item_cap, (long long)(per_row * (unsigned long long)n));
            }
            sp.validate = e->opt_validate ? std::max(1, e->opt_tmaj_validate_stride) : 0;
            if (pl.use_pair && !exact_only)
                CUSK_HIP(e, launch_pair(a.mode, sp, pl.pair_lds, cs));
            else if (pl.tmaj && pl.use_fast && !exact_only)
                CUSK_HIP(e, launch_sweep_tmaj(a.mode, l, sp, c, cs));
            else if (pl.use_fast && !exact_only && !het && e->opt_vec && !e->opt_validate && c < kNumClasses - 1 &&
                     sweep_vec_lds_bytes(c) <= kLdsLimit && l < kVecMaxLevel)
                CUSK_HIP(e, launch_sweep_vec(a.mode, l, sp, c, c == 0 ? e->opt_vec_threads : kThreads, cs));
            else if (pl.use_fast && !exact_only)
                CUSK_HIP(e, launch_sweep_fast(a.mode, het, l, e->opt_validate != 0, sp, c, cs));
            else
                CUSK_HIP(e, launch_sweep_exact(a.mode, het, l, sp, c, cs));
        }
        if (fork)
        {
            CUSK_HIP(e, hipEventRecord(e->ev_join, e->stream2));
            CUSK_HIP(e, hipStreamWaitEvent(s, e->ev_join, 0));
        }
        if (pl.use_fast && !exact_only) CUSK_HIP(e, launch_recheck(a.mode, het, l, sp, s));
        return CUSK_OK;
    };
    auto launch_level_finalize = [&](int l) -> int {
        if (a.mode != 0) return CUSK_OK;
        LevelPlan &pl = plan[l];
        pl.fp.qcap = qcap_gate[l];
        // separating-set records in place (level-1 slot of the pair), bitmap rows and degrees; the winners' exact z is
        // computed when somebody asks for it
        CUSK_HIP(e, launch_finalize(l, pl.fp, s));
        return CUSK_OK;
    };

    // row-sharded runs: join the engines' selection state after a level's sweep (unsigned MIN), then derive what the
    // sweep kernels would have left behind on a single engine
    auto shard_join = [&](int l) -> int {
        LevelPlan &pl = plan[l];
        if (pl.use_fast && l >= 2)
        {  // a recheck queue that overflowed on this engine: finish its rows on the exact path now (local decision)
            CUSK_HIP(e, hipMemcpyAsync(e->hcnt + l, dcnt + l, sizeof(LevelCounters), hipMemcpyDeviceToHost, s));
            CUSK_HIP(e, hipStreamSynchronize(s));
            if (e->hcnt[l].qcount > (unsigned long long)e->opt_queue_cap)
            {
                pl.redone = true;
                local.exact_fallbacks++;
                qcap_gate[l] = ~0ull;
                const int rc = launch_level_sweeps(l, true);
                if (rc != CUSK_OK) return rc;
            }
        }
        const size_t count = (size_t)e->hgate[l].total_edges;
        // Skeleton: the lowest passing rank per slot (64 bits; the row kernel's 32-bit form at level 1).  hetcor: one
        // 32-bit mark per slot, 0 = the edge is gone (the row kernel leaves exactly that; the other kernels clear
        // adjacency bits, which are turned into marks here): in both cases the join is an unsigned MIN
        if (a.mode == 1 && !pl.use_rows) CUSK_HIP(e, launch_marks_from_bitmap(pl.sp, e->sel.as<unsigned>(), s));
        const int elem = (pl.use_rows || a.mode == 1) ? 4 : 8;
        void *dev = (pl.use_rows || a.mode == 1) ? e->sel.p : e->best[l & 1].p;
        int rc = 0;
        if (e->shard_host_staging)
        {
            const size_t bytes = count * (size_t)elem;
            if (bytes > e->shard_host_cap)
            {
                if (e->shard_host) (void)hipHostFree(e->shard_host);
                e->shard_host = nullptr;
                e->shard_host_cap = 0;
                CUSK_HIP(e, hipHostMalloc(&e->shard_host, std::max<size_t>(bytes, 64)));
                e->shard_host_cap = std::max<size_t>(bytes, 64);
            }
            CUSK_HIP(e, hipMemcpyAsync(e->shard_host, dev, bytes, hipMemcpyDeviceToHost, s));
            CUSK_HIP(e, hipStreamSynchronize(s));
            rc = e->shard_fn(e->shard_user, l, e->shard_host, count, elem, 0, (void *)s);
            if (rc == 0) CUSK_HIP(e, hipMemcpyAsync(dev, e->shard_host, bytes, hipMemcpyHostToDevice, s));
        }
        else
        {
            CUSK_HIP(e, hipStreamSynchronize(s));
            rc = e->shard_fn(e->shard_user, l, dev, count, elem, 1, (void *)s);
        }
        if (rc != 0) return fail(e, CUSK_ERR_ARG, "row-shard exchange failed at level " + std::to_string(l));
        // hetcor: every engine now drops the edges any engine removed (bitmap rows and degrees, identically everywhere)
        if (a.mode == 1)
            CUSK_HIP(e, launch_level1_apply(pl.sp, e->sel.as<unsigned>(), (l == 1 && pl.use_rows) ? e->rpos.p : nullptr, false, s));
        return CUSK_OK;
    };

    if (last_level >= 1 && !a.row_range)
        CUSK_HIP(e, launch_degree(e->adj.as<unsigned long long>(), e->deg.as<int>(), n, words,
                                  a.mode == 0 ? e->adj0.as<unsigned long long>() : nullptr, s));

    // which classes can be staged in LDS in this mode
    int staged_classes = 0;
    while (staged_classes < kNumClasses - 1 && lds_layout(kClassCap[staged_classes], het).total <= kLdsLimit) staged_classes++;
    if (e->opt_max_staged_classes >= 0) staged_classes = std::min(staged_classes, e->opt_max_staged_classes);  // test hook: 0 = every row through L2
    // The host enqueues levels AHEAD of the device: every kernel of a level checks the level's gate on the device
    // (LevelCounters::active, set by the level's plan), sweeps are persistent launches that read their work-item counts
    // on the device, so no launch needs a number from the host.  The host only follows `lookahead` levels behind (it
    // waits for the counters of level l - lookahead before it enqueues level l + 1) to stop enqueuing once the loop has
    // ended; that wait overlaps the device's work on the levels in between.  Row-sharded runs call back into the host
    // every level and follow at distance 0.
    const int lookahead = sharded ? 0 : std::max(0, e->opt_lookahead);
    // the level's plan kernel writes its gate record into pinned host memory; spin until this run's record is there
    auto wait_gate = [&](int l) -> int {
        volatile int *seq = &e->hgate[l].seq;
        for (unsigned long it = 1; *seq != run_seq; it++)
        {
            if ((it & 0x3fff) == 0)
            {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess)
                {
                    if (*seq == run_seq) break;
                    return fail(e, CUSK_ERR_HIP, "level " + std::to_string(l) + ": the stream drained without the level's gate record");
                }
                if (q != hipErrorNotReady) return fail(e, CUSK_ERR_HIP, std::string("stream error while waiting for a level gate: ") + hipGetErrorString(q));
            }
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return CUSK_OK;
    };
    int maxdeg1 = 0;
    int maxdeg2 = -1;  // maximum degree at the start of level 2, once the host has seen it
    int start = 1;   // first level to enqueue in this pass
    int redo = 0;    // level whose recheck queue overflowed: its sweeps run again on the exact path before `start`
    bool first_pass = true;

    for (;;)
    {
        if (redo > 0)
        {  // everything the fast pass recorded is a certified verdict and stays valid
            plan[redo].redone = true;
            local.exact_fallbacks++;
            qcap_gate[redo] = ~0ull;
            int rc = launch_level_sweeps(redo, true);
            if (rc != CUSK_OK) return rc;
            rc = launch_level_finalize(redo);
            if (rc != CUSK_OK) return rc;
            if (e->opt_timing == 2) CUSK_HIP(e, hipEventRecord(e->ev_l1[redo], s));
            redo = 0;
        }
        int enq_last = start - 1;
        int planned_last = start - 1;  // last level whose plan kernel was enqueued (>= enq_last)
        for (int l = start; l <= last_level; l++)
        {
            const int cs = l & 1;
            {  // follow the device at a distance: has the loop ended `lookahead` levels ago?
                const int k = l - 1 - lookahead;
                if (k >= start)
                {
                    const int rc = wait_gate(k);
                    if (rc != CUSK_OK) return rc;
                    if (!e->hgate[k].active) break;
                }
            }
            if (e->opt_timing == 2) CUSK_HIP(e, hipEventRecord(e->ev_l0[l], s));
            LevelPlan &pl = plan[l];
            pl.redone = false;
            pl.known_items = false;
            const bool first_build = (l == 1 && first_pass);
            int *off_l = (l == 1) ? e->off1.as<int>() : e->off[cs].as<int>();
            // 1. the plan of the level from the degrees alone: offsets, work items, totals, gate (level 1 is planned for
            //    the generic kernels first: whether the matrix is symmetric is only known with the first read-back, and the
            //    row-streaming kernel does not use work items)
            PlanArgs pa;
            pa.deg = e->deg.as<int>();
            pa.off = off_l;
            for (int c = 0; c < kNumClasses; c++) pa.items[c] = e->items[cs][c].as<int2>();
            pa.n = n;
            pa.L = l;
            pa.binom = e->binom.as<unsigned long long>();
            pa.chunk = chunk;
            // the small work item only where the one-wavefront kernel runs (sweep_vec.hip); the 256-thread kernels of the
            // same class (exact path, heterogeneous thresholds, levels >= kVecMaxLevel) want long runs of consecutive ranks
            {
                float th_l;
                if (a.mode == 0)
                    th_l = a.Th[l];
                else if (het)
                    th_l = a.Th[0];
                else
                    th_l = uniform_ess_threshold(a.Th[0], a.ess_uniform, l);
                const bool vec0 = (e->opt_fast != 0) && l >= 2 && l < kVecMaxLevel && th_l >= kThMinFilter && !het && e->opt_vec &&
                                  !e->opt_validate && staged_classes > 0;
                // levels 2-4 of an LD block hold a few thousand sets per row at most: a smaller item (fewer sets per lane in
                // turn) shortens the serial tail of every item; from level 5 on the staging of an item has to be amortised
                const unsigned long long c0 = (l <= 4) ? (unsigned long long)std::max<long long>(e->opt_chunk0_low, 64) : chunk0;
                plan[l].chunk0 = vec0 ? c0 : chunk;
                // deep levels by unions T = S + Y: single threshold, symmetric matrix (the inverse-based form has no
                // meaning for the two orientations of an asymmetric input), filter certified for this threshold
                plan[l].tmaj = (e->opt_fast != 0) && !plan[l].force_exact && l >= std::max(2, e->opt_tmaj_min_level) && l <= kML &&
                               th_l >= kThMinFilter && !het && symmetric && !sharded;
                if (plan[l].tmaj) plan[l].chunk0 = chunk;
            }
            pa.chunk0 = plan[l].chunk0;
            pa.Lsets = plan[l].tmaj ? l + 1 : l;
            pa.staged_classes = staged_classes;
            pa.pair_mode = ((l == 1) && !first_build && plan[l].use_pair && !plan[l].use_rows) ? 1 : 0;
            pa.cnt = dcnt + l;
            pa.prev = (l >= 2) ? dcnt + (l - 1) : nullptr;
            pa.prev_qcap = (l >= 2) ? qcap_gate[l - 1] : ~0ull;
            pa.item_cap = item_cap;
            pa.shard_rank = e->shard_rank;
            pa.shard_world = e->shard_world;
            pa.gate = e->hgate_dev + l;
            pa.seq = run_seq;
            pa.sym = (l == 1 && !e->opt_assume_symmetric && !a.row_range) ? dsym : nullptr;
            pa.blocks = e->planblk.as<unsigned long long>();
            pa.blk_seq = ++e->plan_seq;
            if (first_build && e->binom_rows <= 0)
            {  // the plan reads C(d, 1) = d only at level 1, but wants a valid table pointer
                CUSK_HIP(e, e->binom.ensure(sizeof(unsigned long long) * kBinomStride));
                pa.binom = e->binom.as<unsigned long long>();
            }
            CUSK_HIP(e, launch_plan(pa, s));
            planned_last = l;
            if (first_build)
            {
                // the run's one mandatory round trip: sizes of the CSR arrays and of the binomial table come from the
                // level-1 degrees
                hp_mark("plan1_enq");
                // (the record store of a previous run is cleared while the gate record travels: the clear needs no number
                // from the device unless the store has to grow)
                int *const rec_l_before = (a.mode == 0) ? e->rec_l.as<int>() : nullptr;
                const size_t rec_l_cleared = rec_l_before ? e->rec_l.cap : 0;
                if (rec_l_cleared) CUSK_HIP(e, hipMemsetAsync(rec_l_before, 0, rec_l_cleared, s));
                int rc = wait_gate(1);
                if (rc != CUSK_OK) return rc;
                hp_mark("gate1");
                symmetric = (e->hgate[1].sym == 0) || (e->opt_assume_symmetric != 0) || (a.row_range != nullptr);
                cap_edges = std::max<long long>(e->hgate[1].total_edges, 1);
                maxdeg1 = e->hgate[1].maxdeg;
                for (int k = 0; k < 2; k++)
                {
                    CUSK_HIP(e, e->nbr[k].ensure(sizeof(int) * ((size_t)cap_edges + 4)));  // + room: the row kernel's 8-byte requests
                    if (a.mode == 0) CUSK_HIP(e, e->best[k].ensure(sizeof(unsigned long long) * (size_t)cap_edges));
                }
                CUSK_HIP(e, e->rv.ensure(sizeof(float) * ((size_t)cap_edges + 4)));
                CUSK_HIP(e, e->rpos.ensure(sizeof(int) * 4 * (size_t)cap_edges));
                CUSK_HIP(e, e->sel.ensure(sizeof(unsigned) * (size_t)cap_edges));
                if (a.mode == 0)
                {
                    CUSK_HIP(e, e->rec_x.ensure(sizeof(int) * (size_t)cap_edges));
                    CUSK_HIP(e, e->rec_y.ensure(sizeof(int) * (size_t)cap_edges));
                    CUSK_HIP(e, e->rec_l.ensure(sizeof(int) * (size_t)cap_edges));
                    CUSK_HIP(e, e->rec_s.ensure(sizeof(int) * kML * (size_t)cap_edges));
                    e->rec_cap = (long long)(e->rec_l.cap / sizeof(int));
                    if ((long long)(e->rec_s.cap / (sizeof(int) * kML)) < e->rec_cap) e->rec_cap = (long long)(e->rec_s.cap / (sizeof(int) * kML));
                    e->rec_slots = cap_edges;
                    if (e->rec_l.as<int>() != rec_l_before || rec_l_cleared < sizeof(int) * (size_t)cap_edges)
                        CUSK_HIP(e, hipMemsetAsync(e->rec_l.p, 0, sizeof(int) * (size_t)cap_edges, s));  // no records yet
                    CUSK_HIP(e, e->wpre.ensure(sizeof(int) * (size_t)n * words));
                }
                CUSK_HIP(e, e->queue.ensure(sizeof(RecheckEntry) * (size_t)e->opt_queue_cap));
                // binomial table C(a, b), a <= max degree: kept on the device across runs
                if ((long long)maxdeg1 >= e->binom_rows)
                {
                    const int rows = std::max(maxdeg1 + 1, 1024);
                    e->binom_host.assign((size_t)rows * kBinomStride, 0ull);
                    for (int aa = 0; aa < rows; aa++)
                        for (int b = 0; b < kBinomStride; b++)
                            e->binom_host[(size_t)aa * kBinomStride + b] = binom_sat(aa, b);
                    CUSK_HIP(e, e->binom.ensure(e->binom_host.size() * sizeof(unsigned long long)));
                    CUSK_HIP(e, hipMemcpyAsync(e->binom.p, e->binom_host.data(),
                                               e->binom_host.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
                    e->binom_rows = rows;
                }
                // level 1 on a symmetric matrix with a single threshold: row-streaming kernel (or the pair kernel)
                pl.pair_lds = (size_t)maxdeg1 * 20 + 16;
                pl.use_pair = !het && symmetric && (e->opt_pair != 0) && pl.pair_lds <= 64 * 1024;
                pl.use_rows = !het && symmetric && (e->opt_pair != 0) && (e->opt_rows != 0);
                if (pl.use_rows) CUSK_HIP(e, e->wpre.ensure(sizeof(int) * (size_t)n * words));
                if (pl.use_pair && !pl.use_rows)
                {  // the pair kernel counts its work items differently: plan again (option rows = 0 only)
                    CUSK_HIP(e, hipMemsetAsync(dcnt + l, 0, sizeof(LevelCounters), s));
                    pa.pair_mode = 1;
                    pa.binom = e->binom.as<unsigned long long>();
                    pa.seq = run_seq;
                    e->hgate[1].seq = 0;
                    pa.blk_seq = ++e->plan_seq;
                    CUSK_HIP(e, launch_plan(pa, s));
                }
                first_pass = false;
            }
            if (sharded)
            {  // the exchange below is a collective: every engine must know now whether the level runs
                const int rc = wait_gate(l);
                if (rc != CUSK_OK) return rc;
                if (!e->hgate[l].active) break;
                for (int c = 0; c < kNumClasses; c++) pl.nitems[c] = e->hgate[l].class_items[c];
                pl.known_items = true;
            }
            pl.use_fast = (e->opt_fast != 0) && (l >= 2) && !pl.force_exact;
            // 2. the neighbour lists (no host dependency)
            CUSK_HIP(e, launch_fill_nbr(e->adj.as<unsigned long long>(), off_l, e->nbr[cs].as<int>(),
                                        (a.mode == 0 && !pl.use_rows) ? e->best[cs].as<unsigned long long>() : nullptr, n, words,
                                        (l == 1 && (pl.use_rows || a.mode == 0)) ? e->wpre.as<int>() : nullptr, dcnt + l, a.row_range, s));
            // Level 2 is the level at which the degrees collapse (on LD data level 1 removes nine edges in ten): the host
            // looks at its gate record -- which travels while the device compacts the lists just enqueued -- and knows the
            // level's class counts and a degree bound for every later level.  Without it the stale level-1 bound made
            // levels 2-4 fork their sweeps over degree classes that turn out empty (an event fork / join per level, ~20 us).
            if (l == 2 && !sharded && e->opt_sync2)
            {
                const int rc2 = wait_gate(2);
                if (rc2 != CUSK_OK) return rc2;
                if (!e->hgate[2].active) break;
                for (int c = 0; c < kNumClasses; c++) pl.nitems[c] = e->hgate[2].class_items[c];
                pl.known_items = !e->hgate[2].item_overflow;
                maxdeg2 = e->hgate[2].maxdeg;
            }
            {  // degrees only shrink: the newest level whose counters have arrived bounds every later one
                pl.maxdeg_bound = maxdeg1;
                if (maxdeg2 >= 0 && l >= 2) pl.maxdeg_bound = std::min(pl.maxdeg_bound, maxdeg2);
                const int k = l - 1 - lookahead;
                if (k >= 1 && e->hgate[k].seq == run_seq && e->hgate[k].active) pl.maxdeg_bound = std::min(pl.maxdeg_bound, e->hgate[k].maxdeg);
                pl.staged_classes = staged_classes;
            }

            SweepParams &sp = pl.sp;
            sp.C = a.C;
            sp.Ness = a.Ness;
            sp.n = n;
            sp.level = l;
            sp.off = off_l;
            sp.nbr = e->nbr[cs].as<int>();
            sp.best = e->best[cs].as<unsigned long long>();
            sp.adj = e->adj.as<unsigned long long>();
            sp.deg = e->deg.as<int>();
            sp.words = words;
            sp.items = nullptr;
            sp.binom = e->binom.as<unsigned long long>();
            sp.time_index = e->ti.as<int>();
            sp.has_ti = (a.mode == 1 && a.time_index != nullptr) ? 1 : 0;
            sp.chunk = chunk;
            sp.cap = 0;
            sp.cls = 0;
            sp.item_cap = item_cap;
            sp.grid_cap = item_cap;
            sp.cnt = dcnt + l;
            sp.row_range = a.row_range;
            sp.max_span = a.row_range ? a.max_span : n;
            sp.slots = dslots + (size_t)l * kCounterSlots * 4;
            if (a.mode == 0)
                sp.th = a.Th[l];
            else if (het)
                sp.th = a.Th[0];
            else
                sp.th = uniform_ess_threshold(a.Th[0], a.ess_uniform, l);
            {
                const double tq = std::tanh((double)sp.th);
                sp.t2 = (float)(tq * tq);
                // The guard band of the filter (ci_fast.h) is relative; the fp32 Fisher z of the exact path carries an
                // absolute error of ~1e-7.  Below this threshold the margin between the two gets thin, so such levels
                // (N beyond ~4 million samples at alpha 1e-4) run entirely on the exact arithmetic.
                pl.filter_ok = (sp.th >= kThMinFilter);
                if (!pl.filter_ok) pl.use_fast = false;
            }
            sp.queue = nullptr;
            sp.qcap = 0;
            qcap_gate[l] = ~0ull;
            if (pl.use_fast)
            {
                sp.queue = e->queue.as<RecheckEntry>();
                sp.qcap = (unsigned long long)e->opt_queue_cap;
                qcap_gate[l] = sp.qcap;
            }
            FinalizeParams &fp = pl.fp;
            fp.C = a.C;
            fp.n = n;
            fp.off = sp.off;
            fp.nbr = sp.nbr;
            fp.best = sp.best;
            fp.sel = pl.use_rows ? e->sel.as<unsigned>() : nullptr;
            fp.level = l;
            fp.adj = sp.adj;
            fp.deg = sp.deg;
            fp.words = words;
            fp.binom = sp.binom;
            fp.off1 = e->off1.as<int>();
            fp.wpre1 = e->wpre.as<int>();
            fp.adj0 = e->adj0.as<unsigned long long>();
            fp.rec_x = e->rec_x.as<int>();
            fp.rec_y = e->rec_y.as<int>();
            fp.rec_l = e->rec_l.as<int>();
            fp.rec_s = e->rec_s.as<int>();
            fp.rec_cap = e->rec_cap;
            fp.meta = pl.use_rows ? e->rpos.as<int4>() : nullptr;
            fp.cnt = dcnt + l;
            fp.qcap = qcap_gate[l];
            fp.slots = sp.slots;
            fp.canon = dcanon + (size_t)l * kCounterSlots;

            if (e->opt_timing == 1 || e->opt_timing == 2) CUSK_HIP(e, hipEventRecord(e->ev_k0[l], s));
            int rc = launch_level_sweeps(l, false);
            if (rc != CUSK_OK) return rc;
            if (e->opt_timing == 1 || e->opt_timing == 2) CUSK_HIP(e, hipEventRecord(e->ev_k1[l], s));
            if (sharded)
            {
                rc = shard_join(l);
                if (rc != CUSK_OK) return rc;
            }
            rc = launch_level_finalize(l);
            if (rc != CUSK_OK) return rc;
            if (e->opt_timing == 2) CUSK_HIP(e, hipEventRecord(e->ev_l1[l], s));
            enq_last = l;
        }

        // read-back of the whole control block, then: did every enqueued level run to completion?
        CUSK_HIP(e, hipEventRecord(e->ev_join, e->stream2));
        CUSK_HIP(e, hipStreamWaitEvent(s, e->ev_join, 0));
        // counters, slots and record bases in one copy (hcnt, hslots, hrec_base point into the pinned mirror)
        CUSK_HIP(e, hipMemcpyAsync(e->hcnt, ctl, ctl_sym, hipMemcpyDeviceToHost, s));
        CUSK_HIP(e, hipEventRecord(e->ev_run[1], s));
        hp_mark("all_enq");
        CUSK_HIP(e, hipStreamSynchronize(s));
        hp_mark("synced");
        int ended = 0;  // first level whose gate stayed closed (a level whose plan ran but whose sweeps were not enqueued included)
        for (int l = 1; l <= planned_last && !ended; l++)
            if (!e->hcnt[l].active) ended = l;
        const int last_ran = ended ? ended - 1 : enq_last;
        // a recheck queue that overflowed: that level was not finalised and nothing after it ran
        if (!sharded && last_ran >= 2 && plan[last_ran].use_fast && !plan[last_ran].redone &&
            e->hcnt[last_ran].qcount > (unsigned long long)e->opt_queue_cap)
        {
            if (plan[last_ran].tmaj)
            {  // its work items count unions, not conditioning sets: plan the level again for the exact kernels
                plan[last_ran].force_exact = true;
                local.exact_fallbacks++;
                start = last_ran;
            }
            else
            {
                redo = last_ran;
                start = last_ran + 1;
            }
        }
        else if (ended && e->hcnt[ended].overflow)
            return fail(e, CUSK_ERR_OVERFLOW,
                        "C(degree, level) exceeds 2^62 at level " + std::to_string(ended) + " (max degree " +
                            std::to_string(e->hcnt[ended].maxdeg) + ")");
        else if (ended && e->hcnt[ended].item_overflow)
        {  // more work items than the buffers hold: grow them and take the level up again
            long long need = item_cap;
            for (int c = 0; c < kNumClasses; c++) need = std::max(need, e->hcnt[ended].class_items[c]);
            item_cap = need + need / 4;
            for (int k = 0; k < 2; k++)
                for (int c = 0; c < kNumClasses; c++) CUSK_HIP(e, e->items[k][c].ensure(sizeof(int2) * (size_t)item_cap));
            start = ended;
        }
        else
        {
            levels_swept = last_ran;
            if (ended) level_out = ended - 1;  // cuPC-S.cu:154-159
            break;
        }
        // resume: the counters of the levels that are enqueued again start from zero, their gate records are void
        for (int l = start; l < kLevels; l++) e->hgate[l].seq = 0;
        if (start <= 2) maxdeg2 = -1;
        if (start <= last_level)
        {
            CUSK_HIP(e, hipMemsetAsync(dcnt + start, 0, sizeof(LevelCounters) * (size_t)(kLevels - start), s));
            CUSK_HIP(e, hipMemsetAsync(dslots + (size_t)start * kCounterSlots * 4, 0,
                                       sizeof(unsigned long long) * (size_t)(kLevels - start) * kCounterSlots * 4, s));
            CUSK_HIP(e, hipMemsetAsync(dcanon + (size_t)start * kCounterSlots, 0,
                                       sizeof(unsigned long long) * (size_t)(kLevels - start) * kCounterSlots, s));
        }
    }
    // degrees and edge counts at the start of every level that was planned (the one at which the loop ended included)
    for (int l = 1; l <= std::min(levels_swept + 1, last_level); l++)
    {
        local.max_degree[l] = e->hcnt[l].maxdeg;
        local.edges[l] = e->hcnt[l].total_edges;
    }
    local.levels_run += levels_swept;
    float ms = 0.0f;
    CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev_l0[0], e->ev_l1[0]));
    local.kernel_ms[0] = local.level_ms[0] = ms;
    const bool timed = (e->opt_timing == 1 || e->opt_timing == 2);
    for (int l = 1; l <= levels_swept; l++)
    {
        for (int k = 0; k < kCounterSlots; k++)
        {
            const unsigned long long *sl = e->hslots + ((size_t)l * kCounterSlots + k) * 4;
            local.tests[l] += (long long)sl[0];
            local.subsets[l] += (long long)sl[1];
            local.removed[l] += (long long)sl[2];
            local.violations += (long long)sl[3];
            // (hetcor: counted at level 1 by level1_apply_kernel when no time index is given; 0 elsewhere)
            if (a.mode == 0 || l == 1) local.canonical_tests[l] += (long long)e->hcanon[(size_t)l * kCounterSlots + k];
        }
        local.rechecks[l] = (long long)e->hcnt[l].qcount;
        if (!timed)
        {  // timing = 3: the events around the level-1 row kernel only
            if (e->opt_timing == 3 && l == 1 && rows_timed)
            {
                CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev_main[0], e->ev_main[1]));
                local.main_kernel_ms[l] = ms;
            }
            continue;
        }
        CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev_k0[l], e->ev_k1[l]));
        local.kernel_ms[l] = ms;
        local.main_kernel_ms[l] = ms;
        if (l == 1 && rows_timed)
        {
            CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev_main[0], e->ev_main[1]));
            local.main_kernel_ms[l] = ms;
        }
        // level time: plan to finaliser with timing = 2; otherwise from the end of the previous level's sweep to the
        // end of this one's (the previous finaliser, this level's plan, lists and sweep)
        if (e->opt_timing == 2)
            CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev_l0[l], e->ev_l1[l]));
        else
            CUSK_HIP(e, hipEventElapsedTime(&ms, l == 1 ? e->ev_l1[0] : e->ev_k1[l - 1], e->ev_k1[l]));
        local.level_ms[l] = ms;
    }
    CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev_run[0], e->ev_run[1]));
    local.total_ms = ms;
    local.level = level_out;
    e->last_levels = levels_swept;
    e->nrec = 0;  // the dense record list is produced on request (materialize_records)
    e->have_result = true;
    if (st) *st = local;
    hp_mark("done");
    if (e->opt_hostprof) std::fprintf(stderr, "[hostprof]%s\n", hp_log.c_str());
    return CUSK_OK;
}

}  // namespace cusk

using namespace cusk;

// ---------------------------------------------------------------------------
// C ABI: engine
// ---------------------------------------------------------------------------

extern "C" int cusk_engine_create(cusk_engine **out, int device, void *stream)
{
    if (!out) return CUSK_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        std::fprintf(stderr, "libcusk_hip: no HIP device available (this library has no CPU fallback)\n");
        return CUSK_ERR_HIP;
    }
    if (device < 0 || device >= ndev) return CUSK_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return CUSK_ERR_HIP;
    cusk_engine *e = new cusk_engine();
    e->device = device;
    for (int l = 0; l < kLevels; l++) e->ev_k0[l] = e->ev_k1[l] = e->ev_l0[l] = e->ev_l1[l] = nullptr;
    bool ok = true;
    if (stream)
    {
        e->stream = reinterpret_cast<hipStream_t>(stream);
        e->own_stream = false;
    }
    else
    {
        ok = ok && hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) == hipSuccess;
        e->own_stream = ok;
    }
    ok = ok && hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&e->stream3, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&e->ev_z, hipEventDisableTiming) == hipSuccess;
    {
        char *hctl = nullptr;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&hctl), kCtlBytes) == hipSuccess;
        if (ok)
        {
            e->hcnt = reinterpret_cast<LevelCounters *>(hctl + kCtlCnt);
            e->hslots = reinterpret_cast<unsigned long long *>(hctl + kCtlSlots);
            e->hrec_base = reinterpret_cast<long long *>(hctl + kCtlRecBase);
            e->hcanon = reinterpret_cast<unsigned long long *>(hctl + kCtlCanon);
        }
    }
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&e->hflag), sizeof(int)) == hipSuccess;
    // gate records: the plan kernels store into this pinned, coherent host memory directly
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&e->hgate), sizeof(HostGate) * kLevels, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    if (ok)
    {
        std::memset(e->hgate, 0, sizeof(HostGate) * kLevels);
        ok = hipHostGetDevicePointer(reinterpret_cast<void **>(&e->hgate_dev), e->hgate, 0) == hipSuccess;
    }
    for (auto &ev : e->ev_run) ok = ok && hipEventCreate(&ev) == hipSuccess;
    for (auto &ev : e->ev_main) ok = ok && hipEventCreate(&ev) == hipSuccess;
    for (auto &ev : e->ev_corr) ok = ok && hipEventCreate(&ev) == hipSuccess;
    for (int l = 0; l < kLevels; l++)
    {
        ok = ok && hipEventCreate(&e->ev_k0[l]) == hipSuccess && hipEventCreate(&e->ev_k1[l]) == hipSuccess;
        ok = ok && hipEventCreate(&e->ev_l0[l]) == hipSuccess && hipEventCreate(&e->ev_l1[l]) == hipSuccess;
    }
    if (!ok)
    {
        cusk_engine_destroy(e);
        return CUSK_ERR_HIP;
    }
    *out = e;
    // kernel experiments without touching the caller: CUSK_OPTIONS="key=value,key=value" (cusk_engine_set_option pairs)
    if (const char *env = std::getenv("CUSK_OPTIONS"))
    {
        std::string all(env);
        size_t pos = 0;
        while (pos < all.size())
        {
            const size_t end = std::min(all.find(',', pos), all.size());
            const std::string kv = all.substr(pos, end - pos);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos)
                (void)cusk_engine_set_option(e, kv.substr(0, eq).c_str(), std::atoll(kv.c_str() + eq + 1));
            pos = end + 1;
        }
    }
    return CUSK_OK;
}

extern "C" void cusk_engine_destroy(cusk_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    // every stream is idle before anything they may still use is released (an ahead correlation build on stream3 reads the
    // planes / mxp buffers and writes the pinned landing buffer)
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->stream2) (void)hipStreamSynchronize(e->stream2);
    if (e->stream3) (void)hipStreamSynchronize(e->stream3);
    for (DevBuf *b : {&e->row_range, &e->row_blk, &e->blk_woff, &e->pxp_dev, &e->corr_tab[0], &e->corr_tab[1]}) b->release();
    if (e->res_pinned) (void)hipHostFree(e->res_pinned);
    for (void *ph : e->corr_tab_pinned)
        if (ph) (void)hipHostFree(ph);
    if (e->batch_pinned) (void)hipHostFree(e->batch_pinned);
    for (DevBuf *b : {&e->adj, &e->adj0, &e->deg, &e->binom, &e->counters, &e->ti, &e->queue,
                      &e->rv, &e->rpos, &e->sel, &e->wpre, &e->rec_x, &e->rec_y, &e->rec_l, &e->rec_s, &e->den_x, &e->den_y, &e->den_l,
                      &e->den_z, &e->den_s, &e->den_counts, &e->den_off, &e->off1, &e->planblk, &e->scratch_a, &e->scratch_b, &e->bed_dev, &e->phen_dev,
                      &e->mean_dev, &e->std_dev, &e->planes, &e->mxp_dev, &e->mxp_bq})
        b->release();
    for (int k = 0; k < 2; k++)
    {
        for (DevBuf *b : {&e->off[k], &e->nbr[k], &e->best[k]}) b->release();
        for (auto &b : e->items[k]) b.release();
    }
    if (e->shard_host) (void)hipHostFree(e->shard_host);
    if (e->hcnt) (void)hipHostFree(e->hcnt);
    if (e->hflag) (void)hipHostFree(e->hflag);
    if (e->hgate) (void)hipHostFree(e->hgate);
    for (auto &ev : e->ev_run)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : e->ev_main)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : e->ev_corr)
        if (ev) (void)hipEventDestroy(ev);
    for (int l = 0; l < kLevels; l++)
        for (hipEvent_t ev : {e->ev_k0[l], e->ev_k1[l], e->ev_l0[l], e->ev_l1[l]})
            if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {e->ev_fork, e->ev_join, e->ev_z, e->ev_mxp})
        if (ev) (void)hipEventDestroy(ev);
    if (e->stream2) (void)hipStreamDestroy(e->stream2);
    if (e->stream3) (void)hipStreamDestroy(e->stream3);
    if (e->mxp_pinned) (void)hipHostFree(e->mxp_pinned);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int cusk_engine_set_option(cusk_engine *e, const char *key, long long value)
{
    if (!e || !key) return CUSK_ERR_ARG;
    const std::string k(key);
    if (k == "fast")
        e->opt_fast = (int)value;
    else if (k == "validate")
        e->opt_validate = (int)value;
    else if (k == "corr_fp4")
        e->opt_corr_fp4 = (int)value;
    else if (k == "pair")
        e->opt_pair = (int)value;
    else if (k == "rows")
        e->opt_rows = (int)value;
    else if (k == "vec")
        e->opt_vec = (int)value;
    else if (k == "overlap")
        e->opt_overlap = (int)value;
    else if (k == "corr_popcount")
        e->opt_corr_popcount = (int)value;
    else if (k == "corr_mxp_f32")
        e->opt_corr_mxp_f32 = (int)value;
    else if (k == "assume_symmetric")
        e->opt_assume_symmetric = (int)value;
    else if (k == "queue_capacity" && value > 0)
        e->opt_queue_cap = value;
    else if (k == "chunk" && value >= 256)
        e->opt_chunk = value;
    else if (k == "max_staged_classes")
        e->opt_max_staged_classes = (int)value;
    else if (k == "tmaj_validate_stride" && value >= 1 && (value & (value - 1)) == 0)
        e->opt_tmaj_validate_stride = (int)value;
    else if (k == "tmaj_min_level")
        e->opt_tmaj_min_level = (int)value;
    else if (k == "hostprof")
        e->opt_hostprof = (int)value;
    else if (k == "chunk0_low" && value >= 64)
        e->opt_chunk0_low = value;
    else if (k == "chunk0" && value >= 64)
        e->opt_chunk0 = value;
    else if (k == "vec_threads" && (value == 64 || value == 128 || value == 256))
        e->opt_vec_threads = (int)value;
    else if (k == "item_capacity" && value > 0)
    {
        e->opt_item_cap = value;
        e->item_cap_cur = 0;  // takes effect with the next run (buffers only ever grow)
    }
    else if (k == "lookahead" && value >= 0)
        e->opt_lookahead = (int)value;
    else if (k == "timing")
        e->opt_timing = (int)value;
    else if (k == "l1_exp")
        e->opt_l1_exp = (int)value;
    else if (k == "sync2")
        e->opt_sync2 = (int)value;
    else if (k == "sepselect_ws_bytes" && value > 0)
        e->opt_sep_ws_budget = value;
    else
        return fail(e, CUSK_ERR_ARG, "unknown option " + k);
    return CUSK_OK;
}

extern "C" int cusk_engine_set_row_shard(cusk_engine *e, int rank, int world, cusk_exchange_fn fn, void *user, int host_staging)
{
    if (!e) return CUSK_ERR_ARG;
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !fn))
        return fail(e, CUSK_ERR_ARG, "cusk_engine_set_row_shard: need 0 <= rank < world and an exchange function");
    e->shard_rank = rank;
    e->shard_world = world;
    e->shard_fn = fn;
    e->shard_user = user;
    e->shard_host_staging = host_staging ? 1 : 0;
    return CUSK_OK;
}

extern "C" const char *cusk_last_error(const cusk_engine *e) { return e ? e->err.c_str() : "no engine"; }
extern "C" void *cusk_engine_stream(const cusk_engine *e) { return e ? (void *)e->stream : nullptr; }
extern "C" int cusk_engine_device(const cusk_engine *e) { return e ? e->device : -1; }
extern "C" int cusk_engine_bind_thread(cusk_engine *e)
{
    if (!e) return CUSK_ERR_ARG;
    CUSK_HIP(e, hipSetDevice(e->device));
    return CUSK_OK;
}

extern "C" int cusk_run_skeleton(cusk_engine *e, const float *C_dev, int n, const float *Th, int maxlevel,
                                 cusk_stats *stats)
{
    if (!e) return CUSK_ERR_ARG;
    e->batch_lo.clear();
    RunArgs a{};
    a.mode = 0;
    a.C = C_dev;
    a.Th = Th;
    a.n = n;
    a.maxlevel = maxlevel;
    return run_levels(e, a, stats);
}

// Batched Skeleton run: `nblk` independent blocks laid out along the diagonal of one n x n allocation (include/cusk_hip.h).
// One plan / fill / sweep / finalise chain per level serves every block: after level 0 the engine only works on CSR rows,
// and a row never meets a column outside its block, so nothing but level 0 and the level-1 row staging knows about blocks.
extern "C" int cusk_run_skeleton_batch(cusk_engine *e, const float *C_dev, int n, int nblk, const int *lo, const int *hi,
                                       const float *Th, int maxlevel, cusk_stats *stats)
{
    if (!e) return CUSK_ERR_ARG;
    if (!C_dev || !lo || !hi || !Th || n <= 0 || nblk <= 0) return fail(e, CUSK_ERR_ARG, "bad arguments");
    if (e->shard_world > 1) return fail(e, CUSK_ERR_ARG, "batched runs are not row-sharded");
    int prev = 0, span = 0;
    long long pairs0 = 0;
    for (int b = 0; b < nblk; b++)
    {
        if ((lo[b] & 63) != 0 || lo[b] < prev || hi[b] < lo[b] || hi[b] > n)
            return fail(e, CUSK_ERR_ARG, "cusk_run_skeleton_batch: block bases must be ascending multiples of 64 inside the matrix");
        prev = hi[b];
        span = std::max(span, hi[b] - lo[b]);
        pairs0 += (long long)(hi[b] - lo[b]) * (hi[b] - lo[b] - 1) / 2;
    }
    CUSK_HIP(e, hipSetDevice(e->device));
    // per-row tables: column range of the row's block, block number (-1: padding), and per block the offset of its packed
    // bitmap (cusk_result_adj_bits_blocks)
    const size_t bytes_rr = sizeof(int2) * (size_t)n, bytes_rb = sizeof(int) * (size_t)n, bytes_wo = sizeof(long long) * (size_t)nblk;
    const size_t need = bytes_rr + bytes_rb + bytes_wo;
    if (need > e->batch_pinned_cap)
    {
        if (e->batch_pinned) (void)hipHostFree(e->batch_pinned);
        e->batch_pinned = nullptr;
        e->batch_pinned_cap = 0;
        CUSK_HIP(e, hipHostMalloc(&e->batch_pinned, need + need / 2));
        e->batch_pinned_cap = need + need / 2;
    }
    // the previous run's copies out of this staging buffer have completed: every run ends with a stream synchronisation
    int2 *rr = static_cast<int2 *>(e->batch_pinned);
    int *rb = reinterpret_cast<int *>(static_cast<char *>(e->batch_pinned) + bytes_rr);
    long long *wo = reinterpret_cast<long long *>(static_cast<char *>(e->batch_pinned) + bytes_rr + bytes_rb);
    for (int r = 0; r < n; r++)
    {
        rr[r] = make_int2(0, 0);
        rb[r] = -1;
    }
    long long woff = 0;
    for (int b = 0; b < nblk; b++)
    {
        for (int r = lo[b]; r < hi[b]; r++)
        {
            rr[r] = make_int2(lo[b], hi[b]);
            rb[r] = b;
        }
        wo[b] = woff;
        woff += (long long)(hi[b] - lo[b]) * ((hi[b] - lo[b] + 63) / 64);
    }
    CUSK_HIP(e, e->row_range.ensure(bytes_rr));
    CUSK_HIP(e, e->row_blk.ensure(bytes_rb));
    CUSK_HIP(e, e->blk_woff.ensure(bytes_wo));
    CUSK_HIP(e, hipMemcpyAsync(e->row_range.p, rr, bytes_rr, hipMemcpyHostToDevice, e->stream));
    CUSK_HIP(e, hipMemcpyAsync(e->row_blk.p, rb, bytes_rb, hipMemcpyHostToDevice, e->stream));
    CUSK_HIP(e, hipMemcpyAsync(e->blk_woff.p, wo, bytes_wo, hipMemcpyHostToDevice, e->stream));
    e->batch_lo.assign(lo, lo + nblk);
    e->batch_hi.assign(hi, hi + nblk);
    RunArgs a{};
    a.mode = 0;
    a.C = C_dev;
    a.Th = Th;
    a.n = n;
    a.maxlevel = maxlevel;
    a.row_range = e->row_range.as<int2>();
    a.max_span = span;
    a.level0_pairs = pairs0;
    const int rc = run_levels(e, a, stats);
    if (rc != CUSK_OK) e->batch_lo.clear();
    return rc;
}

// the adjacency of the last batched run, block by block: rows lo..hi-1 of block b, each as the (hi - lo + 63) / 64 words of
// the block's own columns (bit j = local variable j; bases are multiples of 64), blocks back to back
static int pack_blocks(cusk_engine *e, uint64_t *out_host, int tail)
{
    if (!e || !e->have_result || e->batch_lo.empty() || !out_host || tail < 0) return fail(e, CUSK_ERR_STATE, "no batched result");
    CUSK_HIP(e, hipSetDevice(e->device));
    const size_t nblk = e->batch_lo.size();
    std::vector<long long> woff(nblk);
    long long total = 0;
    for (size_t b = 0; b < nblk; b++)
    {
        const int k = e->batch_hi[b] - e->batch_lo[b];
        woff[b] = total;
        total += (long long)(tail > 0 ? std::min(tail, k) : k) * ((k + 63) / 64);
    }
    if (total == 0) return CUSK_OK;
    CUSK_HIP(e, e->scratch_b.ensure(sizeof(unsigned long long) * (size_t)total));
    const long long *woff_d = e->blk_woff.as<long long>();  // the full packing's offsets were uploaded with the run
    if (tail > 0)
    {
        CUSK_HIP(e, e->scratch_a.ensure(sizeof(long long) * nblk));
        CUSK_HIP(e, hipMemcpyAsync(e->scratch_a.p, woff.data(), sizeof(long long) * nblk, hipMemcpyHostToDevice, e->stream));
        woff_d = e->scratch_a.as<long long>();
    }
    CUSK_HIP(e, launch_pack_block_bits(e->adj.as<unsigned long long>(), e->n, e->words, e->row_range.as<int2>(), e->row_blk.as<int>(),
                                       woff_d, e->scratch_b.as<unsigned long long>(), tail, e->stream));
    CUSK_HIP(e, hipMemcpyAsync(out_host, e->scratch_b.p, sizeof(unsigned long long) * (size_t)total, hipMemcpyDeviceToHost, e->stream));
    CUSK_HIP(e, hipStreamSynchronize(e->stream));  // (also covers the copy out of `woff`)
    return CUSK_OK;
}

extern "C" int cusk_result_adj_bits_blocks(cusk_engine *e, uint64_t *out_host) { return pack_blocks(e, out_host, 0); }
extern "C" int cusk_result_adj_bits_blocks_tail(cusk_engine *e, int tail_rows, uint64_t *out_host)
{
    if (tail_rows <= 0) return fail(e, CUSK_ERR_ARG, "tail_rows must be positive");
    return pack_blocks(e, out_host, tail_rows);
}

// rows [row0, row0 + nrows) of the last run's adjacency bitmap (cusk_result_words() words each): what the pruning of depth 1
// needs of a 10k-variable block are its trait rows, 25 KB instead of the 12.6 MB bitmap
extern "C" int cusk_result_adj_rows(cusk_engine *e, int row0, int nrows, uint64_t *out_host)
{
    if (!e || !e->have_result || !out_host || row0 < 0 || nrows < 0 || row0 + nrows > e->n) return fail(e, CUSK_ERR_ARG, "bad rows");
    if (nrows == 0) return CUSK_OK;
    CUSK_HIP(e, hipSetDevice(e->device));
    CUSK_HIP(e, hipMemcpyAsync(out_host, e->adj.as<unsigned long long>() + (size_t)row0 * e->words,
                               sizeof(unsigned long long) * (size_t)nrows * e->words, hipMemcpyDeviceToHost, e->stream));
    CUSK_HIP(e, hipStreamSynchronize(e->stream));
    return CUSK_OK;
}

// out[row_out[t] + c] = M[row_src[t] * n + idx[row_first[t] + c]], c < row_k[t], for nrows rows t: the sub-matrices of many
// blocks in one launch (stage-two matrices on the diagonal of a batch allocation: out on the device; the retained
// sub-matrices of the results: out on the host).  All index arrays are host memory.
extern "C" int cusk_gather_rows(cusk_engine *e, const float *M_dev, int n, const int *idx_host, long long nidx, const int *row_src,
                                const int *row_k, const long long *row_first, const long long *row_out, long long nrows,
                                float *out, long long out_count, int out_on_device)
{
    if (!e || !M_dev || !idx_host || !row_src || !row_k || !row_first || !row_out || !out || nrows < 0 || nidx < 0)
        return fail(e, CUSK_ERR_ARG, "bad arguments");
    if (nrows == 0) return CUSK_OK;
    CUSK_HIP(e, hipSetDevice(e->device));
    const size_t b_idx = sizeof(int) * (size_t)nidx, b_src = sizeof(int) * (size_t)nrows, b_ll = sizeof(long long) * (size_t)nrows;
    const size_t o_src = (b_idx + 15) & ~(size_t)15, o_k = o_src + ((b_src + 15) & ~(size_t)15), o_first = o_k + ((b_src + 15) & ~(size_t)15),
                 o_out = o_first + b_ll, total = o_out + b_ll;
    CUSK_HIP(e, e->scratch_a.ensure(total));
    char *d = e->scratch_a.as<char>();
    hipStream_t s = e->stream;
    CUSK_HIP(e, hipMemcpyAsync(d, idx_host, b_idx, hipMemcpyHostToDevice, s));
    CUSK_HIP(e, hipMemcpyAsync(d + o_src, row_src, b_src, hipMemcpyHostToDevice, s));
    CUSK_HIP(e, hipMemcpyAsync(d + o_k, row_k, b_src, hipMemcpyHostToDevice, s));
    CUSK_HIP(e, hipMemcpyAsync(d + o_first, row_first, b_ll, hipMemcpyHostToDevice, s));
    CUSK_HIP(e, hipMemcpyAsync(d + o_out, row_out, b_ll, hipMemcpyHostToDevice, s));
    float *dst = out;
    if (!out_on_device)
    {
        CUSK_HIP(e, e->scratch_b.ensure(sizeof(float) * (size_t)out_count));
        dst = e->scratch_b.as<float>();
    }
    CUSK_HIP(e, launch_gather_rows(M_dev, n, reinterpret_cast<const int *>(d), reinterpret_cast<const int *>(d + o_src),
                                   reinterpret_cast<const int *>(d + o_k), reinterpret_cast<const long long *>(d + o_first),
                                   reinterpret_cast<const long long *>(d + o_out), nrows, dst, s));
    if (!out_on_device) CUSK_HIP(e, hipMemcpyAsync(out, dst, sizeof(float) * (size_t)out_count, hipMemcpyDeviceToHost, s));
    CUSK_HIP(e, hipStreamSynchronize(s));  // the host arrays may go away; the scratch is reused by the next call
    return CUSK_OK;
}

extern "C" int cusk_run_hetcor(cusk_engine *e, const float *C_dev, const float *N_dev, float ess_uniform,
                               const int *G_init_dev, int n, float th, int maxlevel, const int *time_index,
                               cusk_stats *stats)
{
    if (!e) return CUSK_ERR_ARG;
    e->batch_lo.clear();
    RunArgs a{};
    a.mode = 1;
    a.C = C_dev;
    a.Ness = N_dev;
    a.ess_uniform = ess_uniform;
    a.Ginit = G_init_dev;
    float thv[1] = {th};
    a.Th = thv;
    a.time_index = time_index;
    a.n = n;
    a.maxlevel = maxlevel;
    return run_levels(e, a, stats);
}

extern "C" int cusk_result_n(const cusk_engine *e) { return (e && e->have_result) ? e->n : 0; }
extern "C" int cusk_result_words(const cusk_engine *e) { return (e && e->have_result) ? e->words : 0; }
extern "C" const uint64_t *cusk_result_adj_bits_dev(const cusk_engine *e)
{
    return (e && e->have_result) ? reinterpret_cast<const uint64_t *>(e->adj.p) : nullptr;
}

extern "C" int cusk_result_adj_i32_dev(cusk_engine *e, int *G_dev)
{
    if (!e || !e->have_result) return fail(e, CUSK_ERR_STATE, "no result");
    CUSK_HIP(e, hipSetDevice(e->device));
    CUSK_HIP(e, launch_expand_adj(e->adj.as<unsigned long long>(), G_dev, e->n, e->words, e->stream));
    CUSK_HIP(e, hipStreamSynchronize(e->stream));
    return CUSK_OK;
}

extern "C" int cusk_result_adj_i32(cusk_engine *e, int *G_host)
{
    if (!e || !e->have_result) return fail(e, CUSK_ERR_STATE, "no result");
    CUSK_HIP(e, hipSetDevice(e->device));
    const size_t bytes = sizeof(int) * (size_t)e->n * e->n;
    int *tmp = nullptr;
    CUSK_HIP(e, hipMalloc(reinterpret_cast<void **>(&tmp), bytes));
    int rc = cusk_result_adj_i32_dev(e, tmp);
    if (rc == CUSK_OK)
    {
        hipError_t st = hipMemcpy(G_host, tmp, bytes, hipMemcpyDeviceToHost);
        if (st != hipSuccess) rc = fail(e, CUSK_ERR_HIP, hipGetErrorString(st));
    }
    (void)hipFree(tmp);
    return rc;
}

// Result read-out, off the hot path: the sparse record store (level-1 slots) becomes a dense list ordered by slot,
// i.e. by (x, y); then, when asked for, the winners' exact Fisher z level by level.
static int materialize_records(cusk_engine *e, bool want_z)
{
    hipStream_t s = e->stream;
    if (!e->records_ready)
    {
        e->nrec = 0;
        const long long slots = e->rec_slots;
        if (slots > 0)
        {
            const long long nb = (slots + kRecBlock - 1) / kRecBlock;
            CUSK_HIP(e, e->den_counts.ensure(sizeof(int) * (size_t)nb));
            CUSK_HIP(e, e->den_off.ensure(sizeof(long long) * (size_t)nb));
            CUSK_HIP(e, launch_rec_count(e->rec_l.as<int>(), slots, e->den_counts.as<int>(), s));
            std::vector<int> counts((size_t)nb);
            CUSK_HIP(e, hipMemcpyAsync(counts.data(), e->den_counts.p, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, s));
            CUSK_HIP(e, hipStreamSynchronize(s));
            std::vector<long long> off((size_t)nb);
            long long tot = 0;
            for (long long b = 0; b < nb; b++)
            {
                off[(size_t)b] = tot;
                tot += counts[(size_t)b];
            }
            e->nrec = tot;
            if (tot > 0)
            {
                CUSK_HIP(e, e->den_x.ensure(sizeof(int) * (size_t)tot));
                CUSK_HIP(e, e->den_y.ensure(sizeof(int) * (size_t)tot));
                CUSK_HIP(e, e->den_l.ensure(sizeof(int) * (size_t)tot));
                CUSK_HIP(e, e->den_z.ensure(sizeof(float) * (size_t)tot));
                CUSK_HIP(e, e->den_s.ensure(sizeof(int) * kML * (size_t)tot));
                e->den_stride = tot;
                CUSK_HIP(e, hipMemcpyAsync(e->den_off.p, off.data(), sizeof(long long) * (size_t)nb, hipMemcpyHostToDevice, s));
                CUSK_HIP(e, launch_rec_compact(e->rec_l.as<int>(), e->rec_x.as<int>(), e->rec_y.as<int>(), e->rec_s.as<int>(),
                                               e->rec_cap, slots, e->den_off.as<long long>(), e->den_x.as<int>(), e->den_y.as<int>(),
                                               e->den_l.as<int>(), e->den_s.as<int>(), e->den_stride, s));
                CUSK_HIP(e, hipStreamSynchronize(s));  // `off` lives on this stack frame
            }
        }
        e->records_ready = true;
        e->z_ready = false;
    }
    if (want_z && !e->z_ready)
    {
        if (e->nrec > 0)
        {
            if (!e->last_C) return fail(e, CUSK_ERR_STATE, "the matrix of the last run is needed for the separating sets' z");
            for (int l = 1; l <= e->last_levels; l++)
                CUSK_HIP(e, launch_record_z(l, e->last_C, e->n, e->den_x.as<int>(), e->den_y.as<int>(), e->den_l.as<int>(),
                                            e->den_s.as<int>(), e->den_stride, e->nrec, e->den_z.as<float>(), s));
            CUSK_HIP(e, hipStreamSynchronize(s));
        }
        e->z_ready = true;
    }
    return CUSK_OK;
}

extern "C" int cusk_result_pmax(cusk_engine *e, const float *C_dev, float *pMax_host)
{
    if (!e || !e->have_result || e->mode != 0) return fail(e, CUSK_ERR_STATE, "no Skeleton result");
    CUSK_HIP(e, hipSetDevice(e->device));
    if (C_dev) e->last_C = C_dev;
    int rc = materialize_records(e, true);
    if (rc != CUSK_OK) return rc;
    const size_t bytes = sizeof(float) * (size_t)e->n * e->n;
    float *tmp = nullptr;
    CUSK_HIP(e, hipMalloc(reinterpret_cast<void **>(&tmp), bytes));
    hipError_t st = launch_expand_pmax(e->adj.as<unsigned long long>(), e->adj0.as<unsigned long long>(), C_dev, tmp, e->n,
                                       e->words, e->den_x.as<int>(), e->den_y.as<int>(), e->den_z.as<float>(), e->nrec,
                                       e->stream);
    if (st == hipSuccess) st = hipMemcpyAsync(pMax_host, tmp, bytes, hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    (void)hipFree(tmp);
    if (st != hipSuccess) return fail(e, CUSK_ERR_HIP, hipGetErrorString(st));
    return CUSK_OK;
}

// separating sets of all records in the ABI's [nrec x 14] layout
static int records_to_host(cusk_engine *e, int *S_host)
{
    // engine-owned scratch: hipMalloc / hipFree per call would synchronise the whole device (several engines share it)
    const size_t bytes = sizeof(int) * kML * (size_t)e->nrec;
    CUSK_HIP(e, e->scratch_a.ensure(bytes));
    int *tmp = e->scratch_a.as<int>();
    hipError_t st = launch_expand_records(e->den_s.as<int>(), e->den_l.as<int>(), e->den_stride, e->nrec, tmp, e->stream);
    if (st == hipSuccess) st = hipMemcpyAsync(S_host, tmp, bytes, hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    if (st != hipSuccess) return fail(e, CUSK_ERR_HIP, hipGetErrorString(st));
    return CUSK_OK;
}

// device -> host on the engine's stream (waits for that stream only)
static hipError_t fetch(cusk_engine *e, void *dst, const void *src, size_t bytes)
{
    hipError_t st = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    return st;
}

extern "C" int cusk_result_sepset_dense(cusk_engine *e, int *SepSet_host)
{
    if (!e || !e->have_result || e->mode != 0) return fail(e, CUSK_ERR_STATE, "no Skeleton result");
    CUSK_HIP(e, hipSetDevice(e->device));
    int rc = materialize_records(e, false);
    if (rc != CUSK_OK) return rc;
    const size_t count = (size_t)e->n * e->n * kML;
    // the dense n*n*14 array exists only for the reference's ABI; it is filled on the host
    std::fill(SepSet_host, SepSet_host + count, -1);
    if (e->nrec > 0)
    {
        std::vector<int> x(e->nrec), y(e->nrec), S((size_t)e->nrec * kML);
        CUSK_HIP(e, fetch(e, x.data(), e->den_x.p, sizeof(int) * e->nrec));
        CUSK_HIP(e, fetch(e, y.data(), e->den_y.p, sizeof(int) * e->nrec));
        rc = records_to_host(e, S.data());
        if (rc != CUSK_OK) return rc;
        for (long long r = 0; r < e->nrec; r++)
            std::memcpy(SepSet_host + ((size_t)x[r] * e->n + y[r]) * kML, S.data() + (size_t)r * kML, sizeof(int) * kML);
    }
    return CUSK_OK;
}

extern "C" long long cusk_result_sepsets(cusk_engine *e, int *x, int *y, int *level, float *z, int *S)
{
    if (!e || !e->have_result || e->mode != 0) return -1;
    if (hipSetDevice(e->device) != hipSuccess) return -1;
    if (materialize_records(e, z != nullptr) != CUSK_OK) return -1;
    const long long c = e->nrec;
    if (c > 0)
    {
        if (x && fetch(e, x, e->den_x.p, sizeof(int) * c) != hipSuccess) return -1;
        if (y && fetch(e, y, e->den_y.p, sizeof(int) * c) != hipSuccess) return -1;
        if (level && fetch(e, level, e->den_l.p, sizeof(int) * c) != hipSuccess) return -1;
        if (z && fetch(e, z, e->den_z.p, sizeof(float) * c) != hipSuccess) return -1;
        if (S && records_to_host(e, S) != CUSK_OK) return -1;
    }
    return c;
}

// x, y and the members of every record in ENGINE-OWNED PINNED host memory: one expand kernel, three copies, one
// synchronisation (cusk_result_sepsets copies into the caller's pageable arrays, each copy staged and synchronised by the
// runtime); valid until the next run or result call on this engine
extern "C" long long cusk_result_sepsets_view(cusk_engine *e, const int **x, const int **y, const int **S)
{
    if (!e || !e->have_result || e->mode != 0 || !x || !y || !S) return -1;
    if (hipSetDevice(e->device) != hipSuccess) return -1;
    if (materialize_records(e, false) != CUSK_OK) return -1;
    const long long c = e->nrec;
    *x = *y = *S = nullptr;
    if (c <= 0) return c;
    const size_t bx = sizeof(int) * (size_t)c, bs = sizeof(int) * kML * (size_t)c, need = 2 * bx + bs;
    if (need > e->res_pinned_cap)
    {
        if (e->res_pinned) (void)hipHostFree(e->res_pinned);
        e->res_pinned = nullptr;
        e->res_pinned_cap = 0;
        if (hipHostMalloc(&e->res_pinned, need + need / 2) != hipSuccess) return -1;
        e->res_pinned_cap = need + need / 2;
    }
    if (e->scratch_a.ensure(bs) != hipSuccess) return -1;
    int *hx = static_cast<int *>(e->res_pinned), *hy = hx + c, *hs = hy + c;
    hipError_t st = launch_expand_records(e->den_s.as<int>(), e->den_l.as<int>(), e->den_stride, c, e->scratch_a.as<int>(), e->stream);
    if (st == hipSuccess) st = hipMemcpyAsync(hx, e->den_x.p, bx, hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipMemcpyAsync(hy, e->den_y.p, bx, hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipMemcpyAsync(hs, e->scratch_a.p, bs, hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    if (st != hipSuccess)
    {
        (void)fail(e, CUSK_ERR_HIP, hipGetErrorString(st));
        return -1;
    }
    *x = hx;
    *y = hy;
    *S = hs;
    return c;
}

extern "C" int cusk_gather_submatrix(cusk_engine *e, const float *M_dev, int n, const int *idx_host, int k, float *out_host)
{
    if (!e || !M_dev || !idx_host || !out_host || k <= 0) return fail(e, CUSK_ERR_ARG, "bad arguments");
    CUSK_HIP(e, hipSetDevice(e->device));
    CUSK_HIP(e, e->scratch_a.ensure(sizeof(int) * (size_t)k));
    CUSK_HIP(e, e->scratch_b.ensure(sizeof(float) * (size_t)k * k));
    int *idx_d = e->scratch_a.as<int>();
    float *out_d = e->scratch_b.as<float>();
    hipError_t st = hipMemcpyAsync(idx_d, idx_host, sizeof(int) * (size_t)k, hipMemcpyHostToDevice, e->stream);
    if (st == hipSuccess) st = launch_gather_sub(M_dev, n, idx_d, k, out_d, e->stream);
    if (st == hipSuccess)
        st = hipMemcpyAsync(out_host, out_d, sizeof(float) * (size_t)k * k, hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    if (st != hipSuccess) return fail(e, CUSK_ERR_HIP, hipGetErrorString(st));
    return CUSK_OK;
}

extern "C" int cusk_gather_submatrix_dev(cusk_engine *e, const float *M_dev, int n, const int *idx_host, int k, float *out_dev)
{
    if (!e || !M_dev || !idx_host || !out_dev || k <= 0) return fail(e, CUSK_ERR_ARG, "bad arguments");
    CUSK_HIP(e, hipSetDevice(e->device));
    CUSK_HIP(e, e->scratch_a.ensure(sizeof(int) * (size_t)k));
    int *idx_d = e->scratch_a.as<int>();
    hipError_t st = hipMemcpyAsync(idx_d, idx_host, sizeof(int) * (size_t)k, hipMemcpyHostToDevice, e->stream);
    if (st == hipSuccess) st = launch_gather_sub(M_dev, n, idx_d, k, out_dev, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);  // idx_host may go away; the next run reuses the scratch
    if (st != hipSuccess) return fail(e, CUSK_ERR_HIP, hipGetErrorString(st));
    return CUSK_OK;
}

// device -> host copy ordered after the engine's work and waiting for the engine's stream only (cusk_dev_download is a
// device-wide blocking copy: with several engines on one GPU it serialises them)
extern "C" int cusk_engine_download(cusk_engine *e, void *dst_host, const void *src_dev, size_t bytes)
{
    if (!e || !dst_host || !src_dev) return fail(e, CUSK_ERR_ARG, "bad arguments");
    CUSK_HIP(e, hipSetDevice(e->device));
    CUSK_HIP(e, fetch(e, dst_host, src_dev, bytes));
    return CUSK_OK;
}

extern "C" void *cusk_dev_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    return p;
}
extern "C" void cusk_dev_free(void *p)
{
    if (p) (void)hipFree(p);
}
extern "C" int cusk_dev_upload(void *dst, const void *src, size_t bytes)
{
    return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? CUSK_OK : CUSK_ERR_HIP;
}
extern "C" int cusk_dev_download(void *dst, const void *src, size_t bytes)
{
    return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? CUSK_OK : CUSK_ERR_HIP;
}
