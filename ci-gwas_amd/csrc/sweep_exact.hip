// sweep_exact.hip -- the level sweep, the recheck pass and the finaliser on the EXACT path
// (ci_exact.h: the reference's fp32 operation order).
//
// sweep_kernel is the complete level-l kernel (cuPC-S.cu cal_Indepl1..14,
// hetcor-cuPC-S.cu cal_Indepl1_ess..14_ess): it is what runs when the fast filter is switched
// off, for level 1 on asymmetric / heterogeneous-ESS inputs, and as the fallback when the
// recheck queue overflows.  recheck_kernel evaluates the tests the fast filter could not
// certify; finalize_kernel turns the selected ranks into sparse separating-set records.
#include <algorithm>

#include "ci_exact.h"
#include "sweep_stage.h"

namespace cusk {

template <int L, int MODE, bool HET, bool STAGED>
__global__ void __launch_bounds__(kThreads) sweep_kernel(SweepParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0ull;
    unsigned long long ntests = 0, nsub = 0, nrem = 0;
    // persistent launch: the class's work items are counted on the device (sweep_common.h: level_items)
    const long long nitems = level_items(p);
    for (long long it = blockIdx.x; it < nitems; it += gridDim.x)
    {
    if (it != (long long)blockIdx.x) __syncthreads();  // the previous item's readers are done with the staged copy
    const int2 item = p.items[it];
    RowView<MODE, HET, STAGED> rv(p, item.x, smem);
    rv.stage();
    const int d = rv.d;
    const RankRange rr = lane_ranks(p.binom[(size_t)d * kBinomStride + L], item.y, p.chunk);

    if (rr.lo < rr.hi)
    {
        int idx[L];
        unrank_comb<L>(rr.lo, d, p.binom, idx);
        for (unsigned long long rank = rr.lo; rank < rr.hi; rank++)
        {
            float m2[(L > 1) ? L * L : 1], m1x[L];
#pragma unroll
            for (int a = 0; a < L; a++) m1x[a] = rv.cval(d, idx[a]);
            if constexpr (L >= 2)
            {
#pragma unroll
                for (int a = 0; a < L; a++)
#pragma unroll
                    for (int b = 0; b < L; b++)
                        m2[a * L + b] = (a == b) ? 1.0f : (a < b ? rv.cval(idx[a], idx[b]) : rv.cval(idx[b], idx[a]));
            }
            SubsetExact<L> cx;
            cx.prepare(m2, m1x);
            nsub++;
            [[maybe_unused]] int tmaxS = 0;
            if constexpr (MODE == 1)
            {
                tmaxS = rv.tix(idx[0]);
#pragma unroll
                for (int a = 1; a < L; a++) tmaxS = max(tmaxS, rv.tix(idx[a]));
            }
            bool anyalive = false;
            for (int k2 = 0; k2 < d; k2++)
            {
                const bool live = rv.live(k2, rank);
                anyalive |= live;
                if (!live) continue;
                bool inS = false;
#pragma unroll
                for (int a = 0; a < L; a++) inS |= (idx[a] == k2);
                if (inS) continue;
                if constexpr (MODE == 1)
                {
                    if (tmaxS > max(rv.tix(d), rv.tix(k2))) continue;
                }
                float m1y[L];
#pragma unroll
                for (int a = 0; a < L; a++) m1y[a] = rv.cval(k2, idx[a]);
                const float rho = cx.rho(rv.cval(d, k2), m1y);
                ntests++;
                float lth = p.th;
                if constexpr (HET) lth = rv.template ess_threshold_exact<L>(k2, idx);
                if (z_below<L == 1>(rho, lth))
                    if (rv.separate(k2, rank)) nrem++;
            }
            if (!anyalive) break;
            if (!next_comb<L>(idx, d)) break;
        }
    }
    }  // work items
    __syncthreads();
    flush_counters(s_cnt, p.slots, ntests, nsub, MODE == 1 ? nrem : 0ull, 0ull);
}

template <int L, int MODE, bool HET>
static hipError_t launch_exact_L(const SweepParams &p, int cls, hipStream_t st)
{
    if (cls < kNumClasses - 1)
    {
        const size_t lds = lds_layout(kClassCap[cls], HET).total;
        auto kfn = sweep_kernel<L, MODE, HET, true>;
        if (lds > 64 * 1024)
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const unsigned grid = (unsigned)std::min<long long>(persistent_grid(reinterpret_cast<const void *>(kfn), kThreads, lds),
                                                            std::max<long long>(p.grid_cap, 1));
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kThreads), lds, st, p);
    }
    else
    {
        auto kfn = sweep_kernel<L, MODE, HET, false>;
        const unsigned grid = (unsigned)std::min<long long>(persistent_grid(reinterpret_cast<const void *>(kfn), kThreads, 16),
                                                            std::max<long long>(p.grid_cap, 1));
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kThreads), 16, st, p);
    }
    return hipGetLastError();
}

#define CUSK_FOR_LEVELS(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14)

hipError_t launch_sweep_exact(int mode, bool het, int L, const SweepParams &p, int cls, hipStream_t st)
{
    switch (L)
    {
#define CUSK_CASE(LL)                                                               \
    case LL:                                                                        \
        if (mode == 0) return launch_exact_L<LL, 0, false>(p, cls, st);             \
        return het ? launch_exact_L<LL, 1, true>(p, cls, st)                        \
                   : launch_exact_L<LL, 1, false>(p, cls, st);
        CUSK_FOR_LEVELS(CUSK_CASE)
#undef CUSK_CASE
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// exact evaluation of the queued tests: one lane per entry, operands straight from HBM/L2;
// the entry count is read on the device (no host round trip), grid-stride loop
// ---------------------------------------------------------------------------
constexpr int kExactThreads = 64;  // lanes per workgroup of the rare-entry exact kernels (LDS work space per lane)

template <int L, int MODE, bool HET>
__global__ void __launch_bounds__(kExactThreads) recheck_kernel(SweepParams p)
{
    extern __shared__ __attribute__((aligned(16))) float ws_lds[];
    const unsigned long long count = min(p.cnt->qcount, p.qcap);
    const int n = p.n;
    // Entries go to WORKGROUPS first, to the lanes of a workgroup only when there are more entries than workgroups: the
    // deep levels queue a few dozen tests, and 33 of them side by side in one wavefront run the data-dependent loops of the
    // SVD in lockstep (the wave executes the union of every lane's path); one entry per wave is one undisturbed chain
    for (unsigned long long e = (unsigned long long)blockIdx.x + (unsigned long long)gridDim.x * threadIdx.x; e < count;
         e += (unsigned long long)gridDim.x * blockDim.x)
    {
        const RecheckEntry en = p.queue[e];
        const int X = en.x, k2 = en.k2;
        const int o0 = p.off[X];
        const int d = p.off[X + 1] - o0;
        int idx[L];
        unrank_comb<L>(en.rank, d, p.binom, idx);
        const int Y = p.nbr[o0 + k2];
        int S[L];
#pragma unroll
        for (int a = 0; a < L; a++) S[a] = p.nbr[o0 + idx[a]];
        float m2[(L > 1) ? L * L : 1], m1x[L], m1y[L];
#pragma unroll
        for (int a = 0; a < L; a++)
        {
            m1x[a] = p.C[(size_t)X * n + S[a]];
            m1y[a] = p.C[(size_t)Y * n + S[a]];
        }
        if constexpr (L >= 2)
        {
#pragma unroll
            for (int a = 0; a < L; a++)
#pragma unroll
                for (int b = 0; b < L; b++)
                    m2[a * L + b] =
                        (a == b) ? 1.0f : (a < b ? p.C[(size_t)S[a] * n + S[b]] : p.C[(size_t)S[b] * n + S[a]]);
        }
        SubsetExact<L> cx;
        cx.prepare_ws(m2, m1x, ws_lds + threadIdx.x, kExactThreads);
        const float rho = cx.rho(p.C[(size_t)X * n + Y], m1y);
        float lth = p.th;
        if constexpr (HET)
        {
            float s = 0.0f;
            s += ess_term(p.Ness[(size_t)Y * n + X]);
#pragma unroll
            for (int a = 0; a < L; a++)
            {
                s += ess_term(p.Ness[(size_t)S[a] * n + X]);
                s += ess_term(p.Ness[(size_t)S[a] * n + Y]);
#pragma unroll
                for (int b = 0; b < a; b++) s += ess_term(p.Ness[(size_t)S[a] * n + S[b]]);
            }
            const float me = s / (float)((L + 2) * (L + 1) / 2);
            lth = (float)((double)p.th / sqrt((double)me - (double)L - 3.0));
        }
        if (z_below<L == 1>(rho, lth))
        {
            bool first;
            if constexpr (MODE == 0)
            {
                (void)atomicMin(&p.best[o0 + k2], en.rank);
                first = false;  // Skeleton mode counts removals when it finalises the level
            }
            else
            {
                first = clear_edge(p.adj, p.deg, p.words, X, Y);
            }
            if (first) atomicAdd(&p.slots[(size_t)(blockIdx.x & (kCounterSlots - 1)) * 4 + 2], 1ull);
        }
    }
}

hipError_t launch_recheck(int mode, bool het, int L, const SweepParams &p, hipStream_t st)
{
    // level 2 queues ~1e3 tests on a 10k block, deeper levels tens: a small grid-stride launch
    const dim3 grid(L <= 3 ? 512 : 128), block(kExactThreads);
    const size_t lds = sizeof(float) * (size_t)exact_ws_floats(L) * kExactThreads;
    switch (L)
    {
#define CUSK_CASE(LL)                                                                             \
    case LL:                                                                                      \
    {                                                                                             \
        auto k0 = recheck_kernel<LL, 0, false>;                                                   \
        auto k1 = recheck_kernel<LL, 1, true>;                                                    \
        auto k2 = recheck_kernel<LL, 1, false>;                                                   \
        auto kf = (mode == 0) ? k0 : (het ? k1 : k2);                                             \
        if (lds > 64 * 1024)                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kf),                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        hipLaunchKernelGGL(kf, grid, block, lds, st, p);                                          \
        break;                                                                                    \
    }
        CUSK_FOR_LEVELS(CUSK_CASE)
#undef CUSK_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Skeleton mode: selected ranks -> separating-set records.
//  gather_records_kernel (one wave per row, no heavy arithmetic): every slot with a selected rank writes x, y, level
//    and the L members of S (member-major, so a level writes L dense streams) at the LEVEL-1 slot of its ordered pair
//    (no placement scan, no atomics), and the row clears the removed edges from its bitmap row and sets its degree.
//  The dense record list and the winners' Fisher z (recomputed on the exact path, so pMax never depends on which lane
//  won) are produced when results are fetched: rec_count / rec_compact / record_z kernels below.
// ---------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(256) gather_records_kernel(FinalizeParams p)
{
    __shared__ int s_found[4];
    __shared__ unsigned long long s_canon[4];
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (!level_complete(p.cnt, p.qcap)) return;
    int found = 0;
    // Tests of the canonical schedule (the reference algorithm run sequentially per row, as the oracle counts them): a
    // slot is tested with every set up to its lowest passing one that does not contain it; a slot that is never
    // separated with every set that does not contain it.  Independent of how many tests the parallel sweep executed.
    unsigned long long canon = 0;
    if (row < p.n)
    {
        const int o0 = p.off[row];
        const int d = p.off[row + 1] - o0;
        int removed = 0, dec = 0;
        // sets without a given member: C(d, L) - C(d - 1, L - 1) (0 for rows with fewer than L + 1 neighbours)
        unsigned long long never = 0;
        if (d > L) never = p.binom[(size_t)d * kBinomStride + L] - p.binom[(size_t)(d - 1) * kBinomStride + (L - 1)];
        for (int k0 = 0; k0 < d; k0 += 64)
        {
            const int k = k0 + lane;
            const bool valid = k < d;
            unsigned long long r = kNone;
            if (valid)
            {
                if (p.sel != nullptr)
                {
                    const unsigned v = p.sel[o0 + k];
                    r = (v == 0xffffffffu) ? kNone : (unsigned long long)v;
                }
                else
                    r = p.best[o0 + k];
            }
            const int Y = valid ? p.nbr[o0 + k] : 0;
            // The edge row - Y goes when EITHER side found a separating set.
            bool gone = (r != kNone);
            if (p.meta != nullptr)
            {
                // level 1 behind the row-streaming kernel: the other side's verdict is one look-up away (meta holds the
                // position of `row` in Y's list), every wave only touches its own bitmap row and its own degree, no atomics
                if (valid && !gone)
                {
                    const int4 m = p.meta[o0 + k];
                    gone = p.sel != nullptr ? (p.sel[m.z + m.y] != 0xffffffffu) : (p.best[m.z + m.y] != kNone);
                }
                // clear the bits: the Y of a list ascend, so the lanes that share a 64-bit word are contiguous ->
                // segmented OR over the wave, one plain read-modify-write per word
                const int w = valid ? (Y >> 6) : -1 - lane;
                unsigned long long bits = gone ? (1ull << (Y & 63)) : 0ull;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1)
                {
                    const unsigned long long ob = __shfl_down(bits, o);
                    const int ow = __shfl_down(w, o);
                    if (lane + o < 64 && ow == w) bits |= ob;
                }
                const int pw = __shfl_up(w, 1);
                if (valid && bits != 0ull && (lane == 0 || pw != w)) p.adj[(size_t)row * p.words + w] &= ~bits;
            }
            else if (__ballot(gone) != 0ull)
            {
                // levels >= 2 (a few removals per level): the slot that found a set clears BOTH bits of the edge with
                // returning atomics, and whoever actually cleared a bit takes the edge off that row's degree -- exactly once
                // per row and edge when both sides found a set, and nothing on a redo of the level (the bits are gone).
                // (Round 2 had every surviving slot look the other side's verdict up with a binary search in Y's list:
                // eight dependent loads per wave at every level.)
                int mine = 0;
                if (gone)
                {
                    const unsigned long long bY = 1ull << (Y & 63), bR = 1ull << (row & 63);
                    const unsigned long long o1 = atomicAnd(&p.adj[(size_t)row * p.words + (Y >> 6)], ~bY);
                    const unsigned long long o2 = atomicAnd(&p.adj[(size_t)Y * p.words + (row >> 6)], ~bR);
                    mine = (o1 & bY) ? 1 : 0;
                    if (o2 & bR) atomicSub(&p.deg[Y], 1);
                }
                dec += __popcll(__ballot(mine != 0));
            }
            removed += __popcll(__ballot(gone));
            found += __popcll(__ballot(r != kNone));
            if (valid && r == kNone) canon += never;
            if (r == kNone) continue;
            // the pair's level-1 slot (at level 1 that is the slot itself)
            long long slot = (long long)o0 + k;
            if constexpr (L > 1)
            {
                const size_t wi = (size_t)row * p.words + (Y >> 6);
                slot = (long long)p.off1[row] + p.wpre1[wi] + __popcll(p.adj0[wi] & ((1ull << (Y & 63)) - 1ull));
            }
            int idx[L];
            unrank_comb<L>(r, d, p.binom, idx);
            canon += r + 1ull - combos_upto_containing<L>(idx, k, d, p.binom);
            p.rec_x[slot] = row;
            p.rec_y[slot] = Y;
            p.rec_l[slot] = L;
#pragma unroll
            for (int a = 0; a < L; a++) p.rec_s[(size_t)a * p.rec_cap + slot] = p.nbr[o0 + idx[a]];
        }
        if (p.meta != nullptr)
        {
            if (lane == 0) p.deg[row] = d - removed;  // absolute, so a redo of the level is idempotent
        }
        else if (lane == 0 && dec)
            atomicSub(&p.deg[row], dec);
    }
    // cusk_stats.removed: ordered pairs that received a separating set at this level; cusk_stats.canonical_tests
    for (int o = 32; o > 0; o >>= 1) canon += __shfl_xor(canon, o);
    if (lane == 0)
    {
        s_found[threadIdx.x >> 6] = found;
        s_canon[threadIdx.x >> 6] = canon;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const int t = s_found[0] + s_found[1] + s_found[2] + s_found[3];
        if (t) atomicAdd(&p.slots[(size_t)(blockIdx.x & (kCounterSlots - 1)) * 4 + 2], (unsigned long long)t);
        const unsigned long long c = s_canon[0] + s_canon[1] + s_canon[2] + s_canon[3];
        if (c) atomicAdd(&p.canon[blockIdx.x & (kCounterSlots - 1)], c);
    }
}

// ---- result read-out: sparse record store -> dense list (ordered by level-1 slot = by (x, y)) -> exact z ----
__global__ void __launch_bounds__(256) rec_count_kernel(const int *__restrict__ rec_l, long long slots, int *counts)
{
    __shared__ int s_c[4];
    const long long base = (long long)blockIdx.x * kRecBlock;
    int c = 0;
    for (int i = threadIdx.x; i < kRecBlock; i += 256)
        if (base + i < slots && rec_l[base + i] != 0) c++;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

hipError_t launch_rec_count(const int *rec_l, long long slots, int *counts, hipStream_t st)
{
    if (slots <= 0) return hipSuccess;
    hipLaunchKernelGGL(rec_count_kernel, dim3((unsigned)((slots + kRecBlock - 1) / kRecBlock)), dim3(256), 0, st, rec_l, slots, counts);
    return hipGetLastError();
}

// one wave per block of kRecBlock slots, in slot order (ballot ranks): deterministic dense order
__global__ void __launch_bounds__(64) rec_compact_kernel(const int *__restrict__ rec_l, const int *__restrict__ rec_x,
                                                         const int *__restrict__ rec_y, const int *__restrict__ rec_s,
                                                         long long rec_cap, long long slots, const long long *__restrict__ block_off,
                                                         int *out_x, int *out_y, int *out_l, int *out_s, long long out_stride)
{
    const long long base = (long long)blockIdx.x * kRecBlock;
    const int lane = threadIdx.x;
    long long next = block_off[blockIdx.x];
    for (int i0 = 0; i0 < kRecBlock; i0 += 64)
    {
        const long long sl = base + i0 + lane;
        const int lv = (sl < slots) ? rec_l[sl] : 0;
        const unsigned long long has = __ballot(lv != 0);
        if (has == 0ull) continue;
        const long long dst = next + __popcll(has & ((1ull << lane) - 1ull));
        next += __popcll(has);
        if (lv == 0) continue;
        out_x[dst] = rec_x[sl];
        out_y[dst] = rec_y[sl];
        out_l[dst] = lv;
        for (int a = 0; a < lv; a++) out_s[(size_t)a * out_stride + dst] = rec_s[(size_t)a * rec_cap + sl];
    }
}

hipError_t launch_rec_compact(const int *rec_l, const int *rec_x, const int *rec_y, const int *rec_s, long long rec_cap,
                              long long slots, const long long *block_off, int *out_x, int *out_y, int *out_l, int *out_s,
                              long long out_stride, hipStream_t st)
{
    if (slots <= 0) return hipSuccess;
    hipLaunchKernelGGL(rec_compact_kernel, dim3((unsigned)((slots + kRecBlock - 1) / kRecBlock)), dim3(64), 0, st, rec_l, rec_x,
                       rec_y, rec_s, rec_cap, slots, block_off, out_x, out_y, out_l, out_s, out_stride);
    return hipGetLastError();
}

template <int L>
__global__ void __launch_bounds__(kExactThreads) record_z_kernel(const float *__restrict__ C, int n, const int *__restrict__ rx,
                                                                  const int *__restrict__ ry, const int *__restrict__ rl,
                                                                  const int *__restrict__ rs, long long stride, long long count,
                                                                  float *rz)
{
    extern __shared__ __attribute__((aligned(16))) float ws_lds[];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
    {
        if (rl[i] != L) continue;
        const int X = rx[i], Y = ry[i];
        int S[L];
#pragma unroll
        for (int a = 0; a < L; a++) S[a] = rs[(size_t)a * stride + i];
        float m2[(L > 1) ? L * L : 1], m1x[L], m1y[L];
#pragma unroll
        for (int a = 0; a < L; a++)
        {
            m1x[a] = C[(size_t)X * n + S[a]];
            m1y[a] = C[(size_t)Y * n + S[a]];
        }
        if constexpr (L >= 2)
        {
#pragma unroll
            for (int a = 0; a < L; a++)
#pragma unroll
                for (int b = 0; b < L; b++)
                    m2[a * L + b] = (a == b) ? 1.0f : (a < b ? C[(size_t)S[a] * n + S[b]] : C[(size_t)S[b] * n + S[a]]);
        }
        SubsetExact<L> cx;
        cx.prepare_ws(m2, m1x, ws_lds + threadIdx.x, kExactThreads);
        const float rho = cx.rho(C[(size_t)X * n + Y], m1y);
        float z;
        (void)z_below<L == 1>(rho, 0.0f, &z);
        rz[i] = z;
    }
}

hipError_t launch_record_z(int L, const float *C, int n, const int *x, const int *y, const int *l, const int *s, long long stride,
                           long long count, float *z, hipStream_t st)
{
    if (count <= 0) return hipSuccess;
    const dim3 zgrid((unsigned)std::min<long long>((count + kExactThreads - 1) / kExactThreads, 8192)), zblock(kExactThreads);
    const size_t zlds = sizeof(float) * (size_t)exact_ws_floats(L) * kExactThreads;
    switch (L)
    {
#define CUSK_CASE(LL)                                                                                     \
    case LL:                                                                                              \
        if (zlds > 64 * 1024)                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(record_z_kernel<LL>),                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)zlds);             \
        hipLaunchKernelGGL(record_z_kernel<LL>, zgrid, zblock, zlds, st, C, n, x, y, l, s, stride, count, z); \
        break;
        CUSK_FOR_LEVELS(CUSK_CASE)
#undef CUSK_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// records -> the ABI's [count x 14] layout, -1 beyond each record's level (result read-out, not on the hot path)
__global__ void expand_records_kernel(const int *__restrict__ rec_s, const int *__restrict__ rec_l, long long rec_cap,
                                      long long count, int *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * kML) return;
    const long long r = i / kML;
    const int a = (int)(i - r * kML);
    out[i] = (a < rec_l[r]) ? rec_s[(size_t)a * rec_cap + r] : -1;
}

hipError_t launch_expand_records(const int *rec_s, const int *rec_l, long long rec_cap, long long count, int *out,
                                 hipStream_t st)
{
    if (count <= 0) return hipSuccess;
    const long long total = count * kML;
    hipLaunchKernelGGL(expand_records_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, rec_s, rec_l, rec_cap,
                       count, out);
    return hipGetLastError();
}

hipError_t launch_finalize(int L, const FinalizeParams &p, hipStream_t st)
{
    const dim3 grid((p.n + 3) / 4), block(256);
    switch (L)
    {
#define CUSK_CASE(LL) \
    case LL: hipLaunchKernelGGL(gather_records_kernel<LL>, grid, block, 0, st, p); break;
        CUSK_FOR_LEVELS(CUSK_CASE)
#undef CUSK_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace cusk
