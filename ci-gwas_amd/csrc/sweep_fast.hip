// sweep_fast.hip -- level sweep for l >= 2 through the register-Cholesky filter (ci_fast.h).
//
// Same tiling, LDS staging, lane <-> conditioning-set mapping and selection rule as the exact
// sweep_kernel; every test is first judged by the filter, certified verdicts are applied at
// once and the uncertain remainder is queued for recheck_kernel (exact path).  VALIDATE also
// runs the exact arithmetic on every certified verdict and counts contradictions.
#include <algorithm>

#include "ci_exact.h"
#include "ci_fast.h"
#include "sweep_stage.h"

namespace cusk {

// Register budget by measurement (10k block, stage two): four waves per SIMD (<= 128 VGPRs) up to level 12, three
// (<= 168) for 13 and 14 -- the deepest level sits just above that line on its own and the third wave is worth more
// than the handful of spilled values (3.1 s instead of 4.4 s), a fourth would spill the factor (16.7 s).
template <int L, int MODE, bool HET, bool STAGED, bool VALIDATE>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(L <= 12 ? 4 : 3))) sweep_fast_kernel(SweepParams p)
{
    static_assert(L >= 2, "level 1 has its own kernels");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0ull;
    unsigned long long ntests = 0, nsub = 0, nrem = 0, nbad = 0;
    // persistent launch: the class's work items are counted on the device (sweep_common.h: level_items)
    const long long nitems = level_items(p);
    // Few items per workgroup: fixed stride (an atomic per item would cost more than it evens out).  Many (the deep
    // levels of stage two run millions): batches drawn from a counter, so that workgroups whose items end early
    // (every pair decided) take more of them.
    constexpr long long kBatch = 8;
    __shared__ long long s_next;
    const bool dynamic = nitems > (long long)gridDim.x * 64;
    long long it = dynamic ? 0 : (long long)blockIdx.x, batch_end = 0;
    bool first = true;
    for (;;)
    {
    if (dynamic && it >= batch_end)
    {
        __syncthreads();
        if (threadIdx.x == 0) s_next = (long long)atomicAdd(&p.cnt->next_item[p.cls], (unsigned long long)kBatch);
        __syncthreads();
        it = s_next;
        batch_end = it + kBatch;
        first = true;  // the barriers above already separate the items
    }
    if (it >= nitems) break;
    if (!first) __syncthreads();  // the previous item's readers are done with the staged copy
    first = false;
    const long long it_cur = it;
    it = dynamic ? it + 1 : it + gridDim.x;
    const int2 item = p.items[it_cur];
    RowView<MODE, HET, STAGED> rv(p, item.x, smem);
    rv.stage();
    const int d = rv.d;
    const RankRange rr = lane_ranks(p.binom[(size_t)d * kBinomStride + L], item.y, p.chunk);

    if (rr.lo < rr.hi)
    {
        int idx[L];
        unrank_comb<L>(rr.lo, d, p.binom, idx);
        SubsetFast<L> fx;
        int changed = 0;  // first member that differs from the previous set of this lane: rows before it are kept
        for (unsigned long long rank = rr.lo; rank < rr.hi; rank++)
        {
            [[maybe_unused]] float m1x[L];
            if constexpr (VALIDATE)
            {
#pragma unroll
                for (int a = 0; a < L; a++) m1x[a] = rv.cval(d, idx[a]);
            }
            fx.prepare_rows(changed, [&](int a, int b) { return rv.cval(idx[b], idx[a]); }, [&](int a) { return rv.cval(d, idx[a]); });
            nsub++;
            [[maybe_unused]] SubsetExact<L> cx;
            if constexpr (VALIDATE)
            {
                float m2[L * L];
#pragma unroll
                for (int a = 0; a < L; a++)
#pragma unroll
                    for (int b = 0; b < L; b++)
                        m2[a * L + b] = (a == b) ? 1.0f : (a < b ? rv.cval(idx[a], idx[b]) : rv.cval(idx[b], idx[a]));
                cx.prepare(m2, m1x);
            }
            [[maybe_unused]] int tmaxS = 0;
            [[maybe_unused]] float essS = 0.0f;
            if constexpr (MODE == 1)
            {
                tmaxS = rv.tix(idx[0]);
#pragma unroll
                for (int a = 1; a < L; a++) tmaxS = max(tmaxS, rv.tix(idx[a]));
            }
            if constexpr (HET)
            {
                // filter-only estimate of the subset's share of mean_ess (summation order is free here)
#pragma unroll
                for (int a = 0; a < L; a++)
                {
                    essS += ess_term(rv.eval(idx[a], d));
#pragma unroll
                    for (int b = 0; b < a; b++) essS += ess_term(rv.eval(idx[a], idx[b]));
                }
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
                ntests++;
                int v = kUnsure;
                float m1y[L];
#pragma unroll
                for (int a = 0; a < L; a++) m1y[a] = rv.cval(k2, idx[a]);
                const float m0 = rv.cval(d, k2);
                if (!fx.ill)
                {
                    if constexpr (HET)
                    {
                        float s = essS + ess_term(rv.eval(k2, d));
#pragma unroll
                        for (int a = 0; a < L; a++) s += ess_term(rv.eval(idx[a], k2));
                        const float me = s / (float)((L + 2) * (L + 1) / 2);
                        v = fx.verdict_z(m0, m1y, p.th * __frsqrt_rn(me - (float)(L + 3)));
                    }
                    else
                    {
                        v = fx.verdict_fixed(m0, m1y, p.t2);
                    }
                }
                if constexpr (VALIDATE)
                {
                    if (v != kUnsure)
                    {
                        float lth = p.th;
                        if constexpr (HET) lth = rv.template ess_threshold_exact<L>(k2, idx);
                        const bool ex = z_below<false>(cx.rho(m0, m1y), lth);
                        if (ex != (v == kPass)) nbad++;
                    }
                }
                if (v == kUnsure)
                {
                    const unsigned long long qi = atomicAdd(&p.cnt->qcount, 1ull);
                    if (qi < p.qcap)
                    {
                        RecheckEntry en;
                        en.x = rv.X;
                        en.k2 = k2;
                        en.rank = rank;
                        p.queue[qi] = en;
                    }
                }
                else if (v == kPass)
                {
                    if (rv.separate(k2, rank)) nrem++;
                }
            }
            if (!anyalive) break;
            changed = next_comb_pos<L>(idx, d);
            if (changed < 0) break;
        }
    }
    }  // work items
    __syncthreads();
    flush_counters(s_cnt, p.slots, ntests, nsub, MODE == 1 ? nrem : 0ull, nbad);
}

template <int L, int MODE, bool HET, bool VALIDATE>
static hipError_t launch_fast_L(const SweepParams &p, int cls, hipStream_t st)
{
    if (cls < kNumClasses - 1)
    {
        const size_t lds = lds_layout(kClassCap[cls], HET).total;
        auto kfn = sweep_fast_kernel<L, MODE, HET, true, VALIDATE>;
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
        auto kfn = sweep_fast_kernel<L, MODE, HET, false, VALIDATE>;
        const unsigned grid = (unsigned)std::min<long long>(persistent_grid(reinterpret_cast<const void *>(kfn), kThreads, 16),
                                                            std::max<long long>(p.grid_cap, 1));
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kThreads), 16, st, p);
    }
    return hipGetLastError();
}

template <int L>
static hipError_t launch_fast_level(int mode, bool het, bool validate, const SweepParams &p, int cls, hipStream_t st)
{
    if (validate)
    {
        if (mode == 0) return launch_fast_L<L, 0, false, true>(p, cls, st);
        return het ? launch_fast_L<L, 1, true, true>(p, cls, st) : launch_fast_L<L, 1, false, true>(p, cls, st);
    }
    if (mode == 0) return launch_fast_L<L, 0, false, false>(p, cls, st);
    return het ? launch_fast_L<L, 1, true, false>(p, cls, st) : launch_fast_L<L, 1, false, false>(p, cls, st);
}

hipError_t launch_sweep_fast(int mode, bool het, int L, bool validate, const SweepParams &p, int cls, hipStream_t st)
{
    switch (L)
    {
#define CUSK_CASE(LL) \
    case LL: return launch_fast_level<LL>(mode, het, validate, p, cls, st);
        CUSK_CASE(2)
        CUSK_CASE(3)
        CUSK_CASE(4)
        CUSK_CASE(5)
        CUSK_CASE(6)
        CUSK_CASE(7)
        CUSK_CASE(8)
        CUSK_CASE(9)
        CUSK_CASE(10)
        CUSK_CASE(11)
        CUSK_CASE(12)
        CUSK_CASE(13)
        CUSK_CASE(14)
#undef CUSK_CASE
    }
    return hipErrorInvalidValue;
}

}  // namespace cusk
