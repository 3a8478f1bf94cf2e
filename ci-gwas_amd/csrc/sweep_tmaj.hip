// sweep_tmaj.hip -- deep levels of the sweep (l >= kTmajMinLevelDefault = 6, single threshold): enumeration by the UNION
// T = S + {Y} instead of by the conditioning set S.
//
// The S-major kernels (sweep_vec / sweep_fast) pay one forward substitution of O(l^2 / 2) per test (X, Y | S).  All
// l + 1 tests whose variables are the same l + 1 neighbours T of X -- (X, t | T \ t) for t in T -- are entries of ONE
// inverse, P = C[T + X, T + X]^-1:   rho(X, t | T \ t)^2 = P_tX^2 / (P_tt P_XX).   A lane keeps M = C[Q, Q]^-1 of the
// first l - 1 members Q of T in registers (explicit symmetric inverse, built by bordering once per Q) together
// with M C[Q, X] and M C[Q, c1]; per set it forms M C[Q, c2] (one symmetric matrix-vector product, (l-1)^2 FMAs),
// the 3 x 3 Schur complement of Q in {c1, c2, X} and its inverse W, and reads the l + 1 tests off
//     P_qX = -(U W)_qX,   P_qq = M_qq + u_q W u_q^T   (q in Q; U = M C[Q, {c1, c2, X}]),     P_cX = W_cX, P_cc = W_cc.
// About 50 fp32 operations per test at l = 14 instead of ~150.  Every (S, Y) pair is visited exactly once (T = S + Y), the
// selection rule is unchanged: the lowest full-list rank of S per slot of Y (atomic minimum), computed only when a
// test passes.
//
// As in ci_fast.h the result is a FILTER: a verdict is certified only outside the guard band rho^2 vs t^2 (1 +- kBeta)
// and only when every pivot met on the way (bordering pivots of M, the three pivots of the Schur complement) and the
// two conditional variances 1 / P_tt, 1 / P_XX are at least kCondMin; everything else is queued for recheck_kernel
// (the reference's arithmetic).  An explicit inverse is less forgiving than a Cholesky solve (forward error ~ kappa eps
// on the entries of M), so the `validate` option checks every certified verdict of this kernel against a
// double-precision evaluation of the same test with three quarters of the band as margin (cusk_stats.violations).
#include <algorithm>

#include "ci_fast.h"
#include "sweep_stage.h"

namespace cusk {

// lexicographic rank of the ascending positions s[0..L) among the L-subsets of d positions
template <int L>
__device__ __forceinline__ unsigned long long rank_comb(const int *s, int d, const unsigned long long *__restrict__ binom)
{
    unsigned long long r = binom[(size_t)d * kBinomStride + L] - 1ull;
#pragma unroll
    for (int i = 0; i < L; i++)
    {
        const int a = d - 1 - s[i], b = L - i;
        if (a >= b) r -= binom[(size_t)a * kBinomStride + b];
    }
    return r;
}

// double-precision rho^2 of (X ; Y | S) from the staged sub-matrix (validate option only): Cholesky of C[S, S],
// forward substitution of the two right-hand sides
template <int L, typename RV>
__device__ __noinline__ double rho2_f64(const RV &rv, int d, const int *S, int k2)
{
    double F[L * L];
    double a[L], b[L];
    for (int i = 0; i < L; i++)
    {
        double dii = 1.0;
        for (int j = 0; j < i; j++)
        {
            double s = (double)rv.cval(S[i], S[j]);
            for (int k = 0; k < j; k++) s -= F[i * L + k] * F[j * L + k];
            F[i * L + j] = s / F[j * L + j];
            dii -= F[i * L + j] * F[i * L + j];
        }
        F[i * L + i] = sqrt(dii);
        double sa = (double)rv.cval(d, S[i]), sb = (double)rv.cval(k2, S[i]);
        for (int k = 0; k < i; k++)
        {
            sa -= F[i * L + k] * a[k];
            sb -= F[i * L + k] * b[k];
        }
        a[i] = sa / F[i * L + i];
        b[i] = sb / F[i * L + i];
    }
    double h00 = 1.0, h11 = 1.0, h01 = (double)rv.cval(d, k2);
    for (int i = 0; i < L; i++)
    {
        h00 -= a[i] * a[i];
        h11 -= b[i] * b[i];
        h01 -= a[i] * b[i];
    }
    return h01 * h01 / (h00 * h11);
}

// Work decomposition (second form).  A lane that walks its own lexicographic stream of unions rebuilds M whenever ITS
// prefix Q changes -- every seventh step at l = 14, but at a different step in every lane, so a wavefront paid the rebuild
// on every step (measured: 2,800 instructions per set instead of ~800).  The control flow is made uniform instead:
// a work item holds prefixes Q = P + {s} that all END AT THE SAME LIST POSITION s (P = an (l - 2)-subset of the positions
// below s, one per lane, k of them in turn); every lane then runs the same two loops over the pairs s < c1 < c2 < d.
// M is built once per prefix, by all lanes at the same time, and serves C(d - 1 - s, 2) unions; no branch of the
// arithmetic depends on the lane.  Items of a row: sum over s of ceil(C(s, l - 2) / (256 k(s))), k(s) chosen so that an
// item holds about `chunk` unions (tmaj_prefixes_per_lane, shared with the plan kernel).
//
// Register budget: the inverse of C[Q, Q] is (l - 1) l / 2 registers; four waves per SIMD up to level 7, three up to
// level 10, two beyond (at three, levels 11 and 12 spilled 60 - 85 values into the loop and ran slower than level 13).
// Occupancy beats spills here: two waves with 20-60 spilled values (a dozen scratch accesses per union in the hot loop)
// ran 18.5 / 45.6 / 585 / 1,310 ms at levels 9 / 10 / 13 / 14, one wave less without spills 21.7 / 53.1 / 885 / 1,406.
template <int L, int MODE, bool STAGED, bool VALIDATE>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(L <= 7 ? 4 : (L <= 10 ? 3 : 2))))
sweep_tmaj_kernel(SweepParams p)
{
    static_assert(L >= 2 && L + 1 < kBinomStride, "T = S + Y has l + 1 members");
    constexpr int NT = L + 1;  // members of T
    constexpr int NQ = L - 1;  // members of Q = the first l - 1 of T; c1 = T[NQ], c2 = T[NQ + 1]
    constexpr int NP = NQ - 1; // members of Q below its last one
    constexpr int NM = NQ * (NQ + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0ull;
    unsigned long long ntests = 0, nsub = 0, nrem = 0, nbad = 0;
    const long long nitems = level_items(p);
    constexpr long long kBatch = 8;
    __shared__ long long s_next;
    const bool dynamic = nitems > (long long)gridDim.x * 64;
    long long it = dynamic ? 0 : (long long)blockIdx.x, batch_end = 0;
    bool first = true;
    const float t2 = p.t2;
    const float t2lo = t2 * (1.0f - kBeta), t2hi = t2 * (1.0f + kBeta);
    constexpr float kVarMax = 1.0f / kCondMin;  // P_tt <= 64  <=>  conditional variance >= 1/64
    for (;;)
    {
    if (dynamic && it >= batch_end)
    {
        __syncthreads();
        if (threadIdx.x == 0) s_next = (long long)atomicAdd(&p.cnt->next_item[p.cls], (unsigned long long)kBatch);
        __syncthreads();
        it = s_next;
        batch_end = it + kBatch;
        first = true;
    }
    if (it >= nitems) break;
    if (!first) __syncthreads();  // the previous item's readers are done with the staged copy
    first = false;
    const long long it_cur = it;
    it = dynamic ? it + 1 : it + gridDim.x;
    const int2 item = p.items[it_cur];
    RowView<MODE, false, STAGED> rv(p, item.x, smem);
    rv.stage();
    const int d = rv.d;
    [[maybe_unused]] const int tiX = (MODE == 1) ? rv.tix(d) : 0;
    // ---- which s, which prefixes: item.y counts the row's items in the order s = NP, NP + 1, ... ----
    int s = NP, kper = 1;
    unsigned long long nP = 1ull;
    long long rel = item.y;
    for (;; s++)
    {
        nP = (NP == 0) ? 1ull : p.binom[(size_t)s * kBinomStride + NP];
        kper = tmaj_prefixes_per_lane(d, s, p.chunk);
        const long long ni = (long long)((nP + (unsigned long long)kThreads * kper - 1ull) / ((unsigned long long)kThreads * kper));
        if (rel < ni || s >= d - 3) break;
        rel -= ni;
    }
    const unsigned long long r_lo = ((unsigned long long)rel * kThreads + threadIdx.x) * (unsigned long long)kper;

    int idx[NT];
    idx[NQ - 1] = s;
    [[maybe_unused]] int pidx[NP > 0 ? NP : 1];
    for (int kk = 0; kk < kper; kk++)
    {
        const bool act = (r_lo + kk < nP);  // lanes past the last prefix repeat prefix 0 with their results masked
        if constexpr (NP > 0)
        {
            if (kk == 0 || !act)
                unrank_comb<NP>(act ? r_lo + kk : 0ull, s, p.binom, pidx);
            else
                (void)next_comb_pos<NP>(pidx, s);
#pragma unroll
            for (int a = 0; a < NP; a++) idx[a] = pidx[a];
        }
        // ---- M = C[Q, Q]^-1 by bordering, pivot sigma_k = 1 - h^T M_k h = Var(q_k | q_0 .. q_k-1) ----
        float M[NM];  // symmetric, packed rows: (i, j), j <= i at i (i + 1) / 2 + j
        auto Mat = [&](int i, int j) -> float & { return (i >= j) ? M[i * (i + 1) / 2 + j] : M[j * (j + 1) / 2 + i]; };
        bool illQ = false;
#pragma unroll
        for (int k = 0; k < NQ; k++)
        {
            float h[NQ], v[NQ];
#pragma unroll
            for (int j = 0; j < k; j++) h[j] = rv.cval(idx[k], idx[j]);
            float sig = 1.0f;
#pragma unroll
            for (int i = 0; i < k; i++)
            {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < k; j++) acc = __builtin_fmaf(Mat(i, j), h[j], acc);
                v[i] = acc;
                sig = __builtin_fmaf(-h[i], acc, sig);
            }
            illQ = illQ || !(sig >= kCondMin);
            const float inv = __frcp_rn(sig);
#pragma unroll
            for (int i = 0; i < k; i++)
            {
                const float vi = v[i] * inv;
#pragma unroll
                for (int j = 0; j <= i; j++) Mat(i, j) = __builtin_fmaf(vi, v[j], Mat(i, j));
                Mat(k, i) = -vi;
            }
            Mat(k, k) = inv;
        }
        float uX[NQ];  // M C[Q, X]
        float sXX = 1.0f;
        {
            float gx[NQ];
#pragma unroll
            for (int j = 0; j < NQ; j++) gx[j] = rv.cval(d, idx[j]);
#pragma unroll
            for (int i = 0; i < NQ; i++)
            {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < NQ; j++) acc = __builtin_fmaf(Mat(i, j), gx[j], acc);
                uX[i] = acc;
                sXX = __builtin_fmaf(-gx[i], acc, sXX);
            }
        }
        [[maybe_unused]] int tiQ_top = 0, tiQ_second = 0, tiQ_count = 0;
        if constexpr (MODE == 1)
        {  // the two largest time indices among Q (with the multiplicity of the largest)
            tiQ_top = -2147483647;
            tiQ_second = -2147483647;
#pragma unroll
            for (int a = 0; a < NQ; a++)
            {
                const int t = rv.tix(idx[a]);
                if (t > tiQ_top)
                {
                    tiQ_second = tiQ_top;
                    tiQ_top = t;
                    tiQ_count = 1;
                }
                else if (t == tiQ_top)
                    tiQ_count++;
                else if (t > tiQ_second)
                    tiQ_second = t;
            }
        }
        for (int c1 = s + 1; c1 < d - 1; c1++)
        {
            idx[NQ] = c1;
            float u1[NQ];  // M C[Q, c1]
            float s11 = 1.0f, s1X = rv.cval(d, c1);
            {
                float g1[NQ];
#pragma unroll
                for (int j = 0; j < NQ; j++) g1[j] = rv.cval(c1, idx[j]);
#pragma unroll
                for (int i = 0; i < NQ; i++)
                {
                    float acc = 0.0f;
#pragma unroll
                    for (int j = 0; j < NQ; j++) acc = __builtin_fmaf(Mat(i, j), g1[j], acc);
                    u1[i] = acc;
                    s11 = __builtin_fmaf(-g1[i], acc, s11);
                    s1X = __builtin_fmaf(-g1[i], uX[i], s1X);
                }
            }
            const float d1 = s11, r1 = __frcp_rn(d1), l31 = s1X * r1;
            const float sXX1 = __builtin_fmaf(-l31, s1X, sXX);  // Var(X | Q, c1)
            const bool ill1 = illQ || !(d1 >= kCondMin);
            for (int c2 = c1 + 1; c2 < d; c2++)
            {
                idx[NQ + 1] = c2;
                if (act) nsub++;
                // ---- this union: u2 = M C[Q, c2], Schur complement of Q in {c1, c2, X} = L D L^T ----
                float u2[NQ];
                float s12 = rv.cval(c2, c1), s22 = 1.0f, s2X = rv.cval(d, c2);
                {
                    float g2[NQ];
#pragma unroll
                    for (int j = 0; j < NQ; j++) g2[j] = rv.cval(c2, idx[j]);
#pragma unroll
                    for (int i = 0; i < NQ; i++)
                    {
                        float acc = 0.0f;
#pragma unroll
                        for (int j = 0; j < NQ; j++) acc = __builtin_fmaf(Mat(i, j), g2[j], acc);
                        u2[i] = acc;
                        s22 = __builtin_fmaf(-g2[i], acc, s22);
                        s12 = __builtin_fmaf(-g2[i], u1[i], s12);
                        s2X = __builtin_fmaf(-g2[i], uX[i], s2X);
                    }
                }
                const float l21 = s12 * r1;
                const float d2 = __builtin_fmaf(-l21, s12, s22), r2 = __frcp_rn(d2);
                const float tt = __builtin_fmaf(-l21, s1X, s2X);
                const float l32 = tt * r2;
                const float d3 = __builtin_fmaf(-l32, tt, sXX1), r3 = __frcp_rn(d3);  // Var(X | T) = 1 / P_XX
                const bool ill = ill1 || !(d2 >= kCondMin) || !(d3 >= kCondMin);
                // P_tX and P_tt of every member t through z = L^-1 u_t:  u^T W u = sum z_i^2 / d_i,  (W u)_X = z_3 / d_3
                const float hiX = t2hi * r3, loX = t2lo * r3;
                bool anytodo = ill;
                float numv[NT], pyyv[NT];
#pragma unroll
                for (int j = 0; j < NT; j++)
                {
                    float z1, z2, z3, base;
                    if (j < NQ)
                    {
                        z1 = u1[j];
                        z2 = __builtin_fmaf(-l21, z1, u2[j]);
                        z3 = __builtin_fmaf(-l32, z2, __builtin_fmaf(-l31, z1, uX[j]));
                        base = Mat(j, j);
                    }
                    else if (j == NQ)
                    {  // t = c1: u = e_1
                        z1 = 1.0f;
                        z2 = -l21;
                        z3 = __builtin_fmaf(l32, l21, -l31);
                        base = 0.0f;
                    }
                    else
                    {  // t = c2: u = e_2
                        z1 = 0.0f;
                        z2 = 1.0f;
                        z3 = -l32;
                        base = 0.0f;
                    }
                    const float num = z3 * r3;
                    const float pyy = __builtin_fmaf(z3, num, __builtin_fmaf(z2 * z2, r2, __builtin_fmaf(z1 * z1, r1, base)));
                    numv[j] = num;
                    pyyv[j] = pyy;
                    // certainly not separated: rho^2 above the band, conditional variance of t in range (a NaN anywhere
                    // reads as "not certain")
                    const bool fail = (num * num > hiX * pyy) && (pyy <= kVarMax) && (pyy > 0.0f);
                    anytodo = anytodo || !fail;
                }
                if constexpr (MODE == 0)
                {
                    if (act) ntests += NT;
                }
                if (!(anytodo || VALIDATE || MODE == 1)) continue;
                if (!act) continue;
                // ---- rare (Skeleton engine): some test of the union passes or is uncertain ----
                unsigned passm = 0u, unsurem = 0u;
#pragma unroll
                for (int j = 0; j < NT; j++)
                {
                    const float lhs = numv[j] * numv[j];
                    const bool okc = (pyyv[j] <= kVarMax) && (r3 <= kVarMax) && (pyyv[j] > 0.0f);
                    const bool pass = okc && (lhs < loX * pyyv[j]);
                    const bool fail = okc && (lhs > hiX * pyyv[j]);
                    if (pass) passm |= 1u << j;
                    if (!pass && !fail) unsurem |= 1u << j;
                }
                if (ill)
                {
                    passm = 0u;
                    unsurem = (1u << NT) - 1u;
                }
                if constexpr (MODE == 1)
                {
                    // time-index rule of the hetcor engine: a test is skipped when its conditioning set holds a variable
                    // later than both X and Y; the two largest time indices of T settle that for every choice of Y
                    int top = tiQ_top, second = tiQ_second, cnt = tiQ_count;
#pragma unroll
                    for (int a = NQ; a < NT; a++)
                    {
                        const int t = rv.tix(idx[a]);
                        if (t > top)
                        {
                            second = top;
                            top = t;
                            cnt = 1;
                        }
                        else if (t == top)
                            cnt++;
                        else if (t > second)
                            second = t;
                    }
                    unsigned skip = 0u;
#pragma unroll
                    for (int j = 0; j < NT; j++)
                    {
                        const int tj = rv.tix(idx[j]);
                        const int tmaxS = (tj == top && cnt == 1) ? second : top;
                        if (tmaxS > max(tiX, tj)) skip |= 1u << j;
                    }
                    passm &= ~skip;
                    unsurem &= ~skip;
                    ntests += NT - __popc(skip);
                }
                if (VALIDATE && !ill && ((nsub & (unsigned long long)(p.validate - 1)) == 0ull))
                {  // certified verdicts against double precision, three quarters of the band as margin
#pragma unroll
                    for (int j = 0; j < NT; j++)
                    {
                        if ((unsurem >> j) & 1u) continue;
                        int S[L];
#pragma unroll
                        for (int a = 0; a < L; a++) S[a] = idx[a < j ? a : a + 1];
                        const double r2d = rho2_f64<L>(rv, d, S, idx[j]);
                        const bool pj = (passm >> j) & 1u;
                        if (pj ? !(r2d < (double)t2 * (1.0 - 0.25 * (double)kBeta)) : !(r2d > (double)t2 * (1.0 + 0.25 * (double)kBeta))) nbad++;
                    }
                }
                if ((passm | unsurem) != 0u)
                {  // (static member indices: the set stays in registers)
#pragma unroll
                    for (int j = 0; j < NT; j++)
                    {
                        if (!(((passm | unsurem) >> j) & 1u)) continue;
                        int S[L];
#pragma unroll
                        for (int a = 0; a < L; a++) S[a] = idx[a < j ? a : a + 1];
                        const int k2 = idx[j];
                        const unsigned long long srank = rank_comb<L>(S, d, p.binom);
                        if ((passm >> j) & 1u)
                        {
                            if (rv.separate(k2, srank)) nrem++;
                        }
                        else if (rv.live(k2, srank))
                        {  // uncertain, and not yet decided by a lower set: the exact path
                            const unsigned long long qi = atomicAdd(&p.cnt->qcount, 1ull);
                            if (qi < p.qcap)
                            {
                                RecheckEntry en;
                                en.x = rv.X;
                                en.k2 = k2;
                                en.rank = srank;
                                p.queue[qi] = en;
                            }
                        }
                    }
                }
            }
        }
    }
    }  // work items
    __syncthreads();
    flush_counters(s_cnt, p.slots, ntests, nsub, MODE == 1 ? nrem : 0ull, nbad);
}

template <int L, int MODE>
static hipError_t launch_tmaj_L(const SweepParams &p, int cls, hipStream_t st)
{
    if (cls < kNumClasses - 1)
    {
        const size_t lds = lds_layout(kClassCap[cls], false).total;
        // the validating build exists for the Skeleton engine's staged classes (a device function call in the kernel
        // costs the other builds their registers)
        auto kfn = (MODE == 0 && p.validate) ? sweep_tmaj_kernel<L, MODE, true, (MODE == 0)> : sweep_tmaj_kernel<L, MODE, true, false>;
        if (lds > 64 * 1024)
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const unsigned grid = (unsigned)std::min<long long>(persistent_grid(reinterpret_cast<const void *>(kfn), kThreads, lds),
                                                            std::max<long long>(p.grid_cap, 1));
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kThreads), lds, st, p);
    }
    else
    {
        auto kfn = sweep_tmaj_kernel<L, MODE, false, false>;
        const unsigned grid = (unsigned)std::min<long long>(persistent_grid(reinterpret_cast<const void *>(kfn), kThreads, 16),
                                                            std::max<long long>(p.grid_cap, 1));
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kThreads), 16, st, p);
    }
    return hipGetLastError();
}

hipError_t launch_sweep_tmaj(int mode, int L, const SweepParams &p, int cls, hipStream_t st)
{
    switch (L)
    {
#define CUSK_CASE(LL) \
    case LL: return mode == 0 ? launch_tmaj_L<LL, 0>(p, cls, st) : launch_tmaj_L<LL, 1>(p, cls, st);
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
