// sweep_common.h -- structures and device helpers shared by the level-sweep translation units.
#pragma once
#include "cusk_internal.h"

namespace cusk {

constexpr int kBinomStride = 16;        // binom[a * 16 + b] = C(a, b), b <= 15, saturating at 2^62
constexpr int kVecMaxLevel = 9;  // measured crossover (10k block, stage two, d = 39): the float2 kernel wins up to level 8, the scalar one (fewer registers, 2-3 waves per SIMD) from level 9; at l = 14 the float2 form spills
constexpr float kThMinFilter = 2e-3f;  // smallest Fisher-z threshold the fast filters are certified for

struct RowInfo
{
    int cls;      // degree class, -1 = no work this level
    int base;     // first work item of the row inside its class list
    int nchunks;  // number of work items
    int pad;
};

struct RecheckEntry
{
    int x;
    int k2;
    unsigned long long rank;
};

// One per level, device resident; the whole array is mirrored to pinned host memory at each sync.
struct LevelCounters
{
    int maxdeg;
    int overflow;  // C(d, l) >= 2^62 somewhere
    long long total_edges;
    long long class_items[kNumClasses];
    unsigned long long qcount;     // tests queued for the exact path (may exceed the capacity)
    unsigned long long rec_total;  // separating-set records produced by this level
    // The level's gate, decided on the device so that the host can enqueue levels ahead of their counters: set by the
    // last workgroup of plan_kernel to "the previous level ran to completion (its recheck queue did not overflow), a row
    // with more than `level` neighbours exists (cuPC-S.cu:154-159), no binomial overflowed, the work items fit their
    // buffers".  Every other kernel of the level returns at once when it is 0.
    int active;
    int item_overflow;         // a degree class has more work items than its buffer holds (host grows it and resumes)
    unsigned int done_blocks;  // plan_kernel workgroups that have finished (ticket of the last one)
    int pad;
    // long levels (deep stage-two levels: millions of work items per class): workgroups of a persistent launch draw
    // batches of items from here instead of taking a fixed stride, which evens out what early exits leave uneven
    unsigned long long next_item[kNumClasses];
};

// What the host needs to know about a level to follow the device, written by the level's plan kernel straight into
// pinned host memory (no copy, no event on the stream); seq = the run's sequence number, stored last.
struct HostGate
{
    int seq;
    int active, maxdeg, overflow, item_overflow, sym;
    long long total_edges;
    long long class_items[kNumClasses];
};

struct SweepParams
{
    const float *C;
    const float *Ness;  // per-pair effective sample sizes (HET) or nullptr
    int n;
    int level;
    const int *off;
    const int *nbr;
    unsigned long long *best;  // MODE 0: lowest passing rank per CSR slot
    unsigned long long *adj;   // live adjacency bitmap
    int *deg;                  // live degrees (decremented whenever an adjacency bit is cleared)
    int words;
    const int2 *items;
    const unsigned long long *binom;
    const int *time_index;  // MODE 1, device, n entries
    int has_ti;             // MODE 1: a time index was given (else all entries are 0 and the rule excludes nothing)
    float th;               // MODE 0: Th[l]; MODE 1 uniform ESS: th/sqrt(mean_ess-l-3); HET: alpha/2 quantile
    float t2;               // tanh(th)^2 for the fixed-threshold fast filter
    unsigned long long chunk;
    int cap;  // class capacity (LDS carve), ignored when !STAGED
    int cls;  // degree class of this launch: its work-item count is cnt->class_items[cls] (read on the device)
    long long item_cap;         // capacity of the class's work-item buffer
    long long grid_cap;         // host-side upper bound of the class's work items (<= item_cap): workgroups of the persistent launch
    LevelCounters *cnt;         // this level's counters
    unsigned long long *slots;  // this level's kCounterSlots x 4 spread counters: tests, subsets, removed, violations
    RecheckEntry *queue;
    unsigned long long qcap;
    int validate;  // sweep_tmaj: > 0: check the certified verdicts of every validate-th union of a lane (power of two; 1 = all)
                   // against double precision (cusk_stats.violations)
    // Batched runs (cusk_run_skeleton_batch): the matrix is block diagonal -- several independent LD blocks laid out along
    // the diagonal of one n x n allocation, block bases multiples of 64 -- and row r only ever meets the columns
    // [row_range[r].x, row_range[r].y) of its own block (padding rows: an empty range).  nullptr: one block, [0, n).
    const int2 *row_range;
    int max_span;  // largest y - x of row_range (n without it): sizes the LDS row of the level-1 kernel
};

struct FinalizeParams
{
    const float *C;
    int n;
    const int *off;
    const int *nbr;
    const unsigned long long *best;
    const unsigned *sel;  // level 1 after the row-streaming kernel: its 32-bit selection state instead of best
    int level;
    unsigned long long *adj;
    int *deg;
    int words;
    const unsigned long long *binom;
    // Separating-set records live at the LEVEL-1 CSR slot of their ordered pair (a pair is removed at most once per
    // run, so no placement scan is needed): slot of (X, Y) = off1[X] + rank of Y among X's neighbours after level 0 =
    // off1[X] + wpre1[X, Y / 64] + popcount(adj0[X, Y / 64] below bit Y % 64).  rec_l = 0 marks an empty slot; the dense
    // record list (and the winners' exact z) is produced when results are fetched (engine.hip: materialize_records).
    const int *off1;
    const int *wpre1;
    const unsigned long long *adj0;
    int *rec_x, *rec_y, *rec_l, *rec_s;  // rec_s is member-major: member a of slot r at rec_s[a * rec_cap + r]
    long long rec_cap;
    const int4 *meta;  // level 1 (row-streaming kernel ran): per CSR slot {Y, position of X in Y's list, off[Y], deg[Y]}
    const LevelCounters *cnt;  // this level's counters (gate)
    unsigned long long qcap;   // capacity of the level's recheck queue: an overflowed level is not finalised (it is redone)
    unsigned long long *slots; // this level's spread counters ([2] = ordered pairs that received a separating set)
    unsigned long long *canon; // this level's kCounterSlots spread counters of CANONICAL tests (see gather_records_kernel)
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// LDS carve of the staged sweep kernels for a class capacity
struct LdsLayout
{
    size_t nbr, best, ti, sub, ess, total;
};
__host__ __device__ inline LdsLayout lds_layout(int cap, bool het)
{
    LdsLayout l;
    size_t ld = (size_t)((cap + 1) | 1);
    l.nbr = 0;
    l.best = align16(l.nbr + sizeof(int) * (cap + 1));
    l.ti = align16(l.best + sizeof(unsigned long long) * cap);
    l.sub = align16(l.ti + sizeof(int) * (cap + 1));
    l.ess = align16(l.sub + sizeof(float) * (cap + 1) * ld);
    l.total = het ? align16(l.ess + sizeof(float) * (cap + 1) * ld) : l.ess;
    return l;
}

#if defined(__HIPCC__)
// work items of this launch's degree class; 0 when the level's gate is closed
__device__ __forceinline__ long long level_items(const SweepParams &p)
{
    if (!p.cnt->active) return 0;
    const long long c = p.cnt->class_items[p.cls];
    return c < p.item_cap ? c : p.item_cap;
}
// a level is finalised when it ran and its recheck queue held every uncertain test
__device__ __forceinline__ bool level_complete(const LevelCounters *cnt, unsigned long long qcap)
{
    return cnt->active && cnt->qcount <= qcap;
}

// mean_ess of hetcor-cuPC-S.cu:3068-3088: entries truncated to int (v_cvt_i32_f32
// saturates and maps NaN to 0, as the reference's GPU does)
__device__ __forceinline__ float ess_term(float e) { return (float)(int)e; }

// unrank a 0-based lexicographic combination rank into ascending positions idx[0..L) out of d
template <int L>
__device__ __forceinline__ void unrank_comb(unsigned long long rem, int d, const unsigned long long *__restrict__ binom,
                                            int *idx)
{
    int c = 0;
#pragma unroll
    for (int i = 0; i < L; i++)
    {
        if (i == L - 1)
        {  // C(., 0) = 1: the last member follows without a search
            idx[i] = c + (int)rem;
            break;
        }
        while (true)
        {
            unsigned long long b = binom[(size_t)(d - 1 - c) * kBinomStride + (L - 1 - i)];
            if (rem < b) break;
            rem -= b;
            c++;
        }
        idx[i] = c;
        c++;
    }
}

// Number of L-combinations of d positions, in lexicographic order, up to and including S (ascending positions) that
// CONTAIN position k.  Used to count the tests of the canonical (sequential) schedule: a slot k whose lowest passing set
// has rank r was tested at the ranks 0..r whose set does not contain k, i.e. r + 1 - combos_upto_containing(S_r, k) times.
// O(L) binomial look-ups (hockey-stick sums over the values a member can take below S[i]).
template <int L>
__device__ __forceinline__ unsigned long long combos_upto_containing(const int *S, int k, int d,
                                                                       const unsigned long long *__restrict__ binom)
{
    auto C = [&](int a, int b) -> unsigned long long {
        return (a < 0 || b < 0 || b > a) ? 0ull : binom[(size_t)a * kBinomStride + b];
    };
    unsigned long long cnt = 0;
    bool in_prefix = false;  // k among S[0..i)
    int prev = -1;
#pragma unroll
    for (int i = 0; i < L; i++)
    {
        const int lo = prev + 1, hi = S[i] - 1;  // values v < S[i] the i-th member can take behind the common prefix
        const int t = L - i - 1;                 // members still to choose behind v
        if (lo <= hi)
        {
            if (in_prefix)
                cnt += C(d - lo, t + 1) - C(d - 1 - hi, t + 1);  // every completion: sum_v C(d-1-v, t)
            else if (k > prev)
            {
                const int h2 = min(hi, k - 1);  // v < k: k is one of the t completions, sum_v C(d-2-v, t-1)
                if (t >= 1 && lo <= h2) cnt += C(d - 1 - lo, t) - C(d - 2 - h2, t);
                if (lo <= k && k <= hi) cnt += C(d - 1 - k, t);  // v == k
            }
        }
        if (S[i] == k) in_prefix = true;
        prev = S[i];
    }
    return cnt + (in_prefix ? 1ull : 0ull);
}

// clear the edge X - Y in both directions; every bit that this call flips lowers a degree once
__device__ __forceinline__ bool clear_edge(unsigned long long *adj, int *deg, int words, int X, int Y)
{
    const unsigned long long o1 = atomicAnd(&adj[(size_t)X * words + (Y >> 6)], ~(1ull << (Y & 63)));
    const unsigned long long o2 = atomicAnd(&adj[(size_t)Y * words + (X >> 6)], ~(1ull << (X & 63)));
    const bool f1 = (o1 >> (Y & 63)) & 1ull, f2 = (o2 >> (X & 63)) & 1ull;
    if (f1) atomicSub(&deg[X], 1);
    if (f2) atomicSub(&deg[Y], 1);
    return f1;
}
#endif

// ---- launchers (one translation unit each) ----
// sweep_exact.hip
hipError_t launch_sweep_exact(int mode, bool het, int L, const SweepParams &p, int cls, hipStream_t st);
hipError_t launch_recheck(int mode, bool het, int L, const SweepParams &p, hipStream_t st);
hipError_t launch_finalize(int L, const FinalizeParams &p, hipStream_t st);
// dense record list -> exact Fisher z of every record of level L (result read-out)
hipError_t launch_record_z(int L, const float *C, int n, const int *x, const int *y, const int *l, const int *s, long long stride,
                           long long count, float *z, hipStream_t st);
// sparse record store -> per-block counts / dense list (result read-out); blocks of kRecBlock slots
constexpr int kRecBlock = 1024;
hipError_t launch_rec_count(const int *rec_l, long long slots, int *counts, hipStream_t st);
hipError_t launch_rec_compact(const int *rec_l, const int *rec_x, const int *rec_y, const int *rec_s, long long rec_cap,
                              long long slots, const long long *block_off, int *out_x, int *out_y, int *out_l, int *out_s,
                              long long out_stride, hipStream_t st);
// sweep_fast.hip
hipError_t launch_sweep_fast(int mode, bool het, int L, bool validate, const SweepParams &p, int cls, hipStream_t st);
// sweep_tmaj.hip: deep levels by the union T = S + Y (one inverse per l + 1 tests); work items count (l + 1)-subsets
hipError_t launch_sweep_tmaj(int mode, int L, const SweepParams &p, int cls, hipStream_t st);
// union-major levels, work decomposition shared by the plan kernel and the sweep: prefixes (the first l - 1 members of a
// union, the last of them at list position s) per lane of a work item, so that an item holds about `chunk` unions
__host__ __device__ inline int tmaj_prefixes_per_lane(int d, int s, unsigned long long chunk)
{
    const unsigned long long r = (unsigned long long)(d - 1 - s);
    const unsigned long long pairs = r * (r - 1ull) / 2ull;  // s < c1 < c2 < d
    const unsigned long long k = chunk / ((unsigned long long)kThreads * (pairs ? pairs : 1ull));
    return (int)(k < 1ull ? 1ull : (k > 64ull ? 64ull : k));
}
// sweep_vec.hip: vectorised fast sweep (l >= 2, single threshold, staged classes only)
hipError_t launch_sweep_vec(int mode, int L, const SweepParams &p, int cls, int threads, hipStream_t st);
// workgroups of a persistent sweep launch: what the chip holds at once for this kernel (occupancy x CUs), cached
unsigned persistent_grid(const void *kernel, int threads, size_t lds);
size_t sweep_vec_lds_bytes(int cls);
// sweep_level.hip: level 0, compaction, level-1 pair kernel, result expansion
hipError_t launch_level0(const float *C, const float *Ness, const int *Ginit, unsigned long long *adj, int n, int words,
                         float th, int *asym_flag, hipStream_t st);
// block-diagonal level 0 of a batched Skeleton run: adjacency words of every row inside its block's column range, zeros
// elsewhere, written to adj and adj0; also the level-1 degrees
hipError_t launch_level0_batch(const float *C, unsigned long long *adj, unsigned long long *adj0, int *deg, int n, int words,
                               const int2 *row_range, float th, hipStream_t st);
hipError_t launch_degree(const unsigned long long *adj, int *deg, int n, int words, unsigned long long *adj0, hipStream_t st);
hipError_t launch_fill_nbr(const unsigned long long *adj, const int *off, int *nbr, unsigned long long *best, int n, int words,
                           int *wpre, const LevelCounters *cnt, const int2 *row_range, hipStream_t st);
// The level's plan from the degrees alone: CSR offsets, the work items of every degree class (written straight into the
// class buffers), totals, and the level's gate.  cnt = counters of level L; prev = counters of level L-1 (nullptr at
// level 1) with its recheck-queue capacity; gate / seq: pinned host record of the level; sym: level 0's asymmetry flag
// (level 1 only, else nullptr).
struct PlanArgs
{
    const int *deg;
    int *off;
    int2 *items[kNumClasses];
    int n, L;
    int Lsets;  // size of the subsets the work items count: L, or L + 1 where the level is swept by unions T = S + Y (sweep_tmaj.hip)
    const unsigned long long *binom;
    unsigned long long chunk, chunk0;  // conditioning sets per work item: classes >= 1, class 0
    int staged_classes, pair_mode;
    LevelCounters *cnt;
    const LevelCounters *prev;
    unsigned long long prev_qcap;
    long long item_cap;
    int shard_rank, shard_world;
    HostGate *gate;
    int seq;
    const int *sym;
    unsigned long long *blocks;  // ceil(n / 256) x 8 published block totals (plan_kernel)
    unsigned blk_seq;            // 24-bit sequence number of this launch, never 0, different from every launch that wrote `blocks` before
};
hipError_t launch_plan(const PlanArgs &a, hipStream_t st);
hipError_t launch_expand_records(const int *rec_s, const int *rec_l, long long rec_cap, long long count, int *out,
                                 hipStream_t st);
hipError_t launch_level1_rows(int mode, bool validate, bool use_filter, const SweepParams &p, float *rv, void *meta,
                              unsigned *sel, const int *wpre, hipEvent_t ev_begin, hipEvent_t ev_end, int shard_rank,
                              int shard_world, int exp, bool defer_apply, bool has_ti, unsigned long long *canon, hipStream_t st);
// hetcor mode, row-sharded runs: adjacency bitmap -> per-slot marks (0 gone / all ones alive), and back after the join
hipError_t launch_marks_from_bitmap(const SweepParams &p, unsigned *sel, hipStream_t st);
hipError_t launch_level1_apply(const SweepParams &p, const unsigned *sel, const void *meta, bool count_removed, hipStream_t st);
hipError_t launch_pair(int mode, const SweepParams &p, size_t lds, hipStream_t st);
hipError_t launch_gather_sub(const float *M, int n, const int *idx, int k, float *out, hipStream_t st);
hipError_t launch_gather_rows(const float *M, int n, const int *idx, const int *row_src, const int *row_k, const long long *row_first,
                              const long long *row_out, long long nrows, float *out, hipStream_t st);
hipError_t launch_pack_block_bits(const unsigned long long *adj, int n, int words, const int2 *row_range, const int *row_blk,
                                  const long long *blk_woff, unsigned long long *out, int tail, hipStream_t st);
hipError_t launch_expand_adj(const unsigned long long *adj, int *G, int n, int words, hipStream_t st);
hipError_t launch_expand_pmax(const unsigned long long *adj, const unsigned long long *adj0, const float *C, float *pmax,
                              int n, int words, const int *x, const int *y, const float *z, long long nrec, hipStream_t st);

}  // namespace cusk
