// sweep_vec.hip -- the flagship level-sweep kernel: levels >= 2, single threshold per level
// (Skeleton, or hetcor with a uniform effective sample size), rows whose sub-matrix fits LDS.
//
// Same algorithm and certification rule as sweep_fast_kernel (register Cholesky per
// conditioning set, one forward substitution per test, guard-band filter, uncertain tests
// queued for the exact path), restructured around the CDNA4 issue model:
//   * the (d+1)^2 sub-matrix is staged TRANSPOSED with a leading dimension ld4 = 4 (mod 8):
//     the l operands C[Y, S_a] of four consecutive Y are 16 contiguous bytes, so one
//     ds_read_b128 per conditioning variable feeds four tests, and consecutive conditioning
//     columns land on distinct 16-byte slots of the 256-byte bank row (conflict-free);
//   * the Y loop runs in groups of four with all four tests evaluated branch-free as float2
//     pairs (v_pk_fma_f32 / v_pk_mul_f32); liveness, membership of S and the verdicts are
//     bit masks, and only the rare "separated" / "uncertain" lanes leave the straight line;
//   * selection state is kept chunk-relative in 32 bits (0 = decided below this chunk,
//     0xffffffff = undecided), so four states are one ds_read_b128 as well.
// Roughly 25 issued instructions per test at l = 5 instead of ~100.
#include <algorithm>

#include "ci_fast.h"
#include "sweep_common.h"

namespace cusk {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int vec_ld4(int d)
{
    int ld = (d + 1 + 3) & ~3;
    if ((ld & 7) == 0) ld += 4;
    return ld;
}

struct VecLayout
{
    size_t rel, rowx, nbr, ti, sub, bin, total;
};
constexpr int kVecBinCols = kVecMaxLevel - 2;  // binomial columns C(., 1) .. C(., L - 1) of the unranking, L < kVecMaxLevel
__host__ __device__ inline VecLayout vec_layout(int cap)
{
    VecLayout l;
    const size_t dp = (size_t)((cap + 4) & ~3);  // d rounded up to a multiple of 4 (+ room)
    l.rel = 0;
    l.rowx = align16(l.rel + sizeof(unsigned) * dp);
    l.nbr = align16(l.rowx + sizeof(float) * dp);
    l.ti = align16(l.nbr + sizeof(int) * (cap + 1));
    l.sub = align16(l.ti + sizeof(int) * (cap + 1));
    l.bin = align16(l.sub + sizeof(float) * (size_t)(cap + 1) * vec_ld4(cap));
    l.total = align16(l.bin + sizeof(unsigned long long) * (size_t)kVecBinCols * (cap + 1));
    return l;
}

// unrank_comb (sweep_common.h) against the workgroup's LDS copy of the binomial columns: s_bin[(b - 1) * ldb + a] = C(a, b)
template <int L>
__device__ __forceinline__ void unrank_comb_lds(unsigned long long rem, int d, const unsigned long long *s_bin, int ldb, int *idx)
{
    int c = 0;
#pragma unroll
    for (int i = 0; i < L; i++)
    {
        if (i == L - 1)
        {
            idx[i] = c + (int)rem;
            break;
        }
        const unsigned long long *col = s_bin + (L - 2 - i) * ldb;  // C(., L - 1 - i)
        while (true)
        {
            const unsigned long long b = col[d - 1 - c];
            if (rem < b) break;
            rem -= b;
            c++;
        }
        idx[i] = c;
        c++;
    }
}

// THREADS = 256: a workgroup per work item (rows of 40 and more neighbours).  THREADS = 64: ONE WAVEFRONT per work item,
// for the first degree class: its LDS carve is ~10 KB, so a CU holds 16 items at once and the chain of dependent loads
// at the head of every item (offsets -> neighbour list -> operands) overlaps with the arithmetic of the other items;
// with 256 threads and 4-8 items per CU that chain (~10 us) was most of levels 2 and 3 and a third of levels 4 and 5.
template <int L, int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) sweep_vec_kernel(SweepParams p)
{
    static_assert(L >= 2, "level 1 has its own kernels");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_cnt[4];

    const int n = p.n;
    const int tid = threadIdx.x;
    const VecLayout lay = vec_layout(p.cap);
    unsigned *s_rel = reinterpret_cast<unsigned *>(smem + lay.rel);
    float *s_rowx = reinterpret_cast<float *>(smem + lay.rowx);
    int *s_nbr = reinterpret_cast<int *>(smem + lay.nbr);
    int *s_ti = reinterpret_cast<int *>(smem + lay.ti);
    float *s_sub = reinterpret_cast<float *>(smem + lay.sub);
    unsigned long long *s_bin = reinterpret_cast<unsigned long long *>(smem + lay.bin);
    const int ldb = p.cap + 1;
    if (tid < 4) s_cnt[tid] = 0ull;
    // binomial columns of the unranking, once per workgroup (every row of the class has at most p.cap neighbours)
    for (int e = tid; e < (L - 1) * ldb; e += THREADS)
    {
        const int b = e / ldb, a = e - b * ldb;
        s_bin[e] = p.binom[(size_t)a * kBinomStride + (b + 1)];
    }
    unsigned long long ntests = 0, nsub = 0, nrem = 0;
    // persistent launch: the class's work items are counted on the device (sweep_common.h: level_items); a workgroup
    // takes the items blockIdx.x, blockIdx.x + gridDim.x, ... so that the chunks of one row spread over the chip
    const long long nitems = level_items(p);
    int2 item_next = ((long long)blockIdx.x < nitems) ? p.items[blockIdx.x] : make_int2(0, 0);
    for (long long it = blockIdx.x; it < nitems; it += gridDim.x)
    {
    if (it != (long long)blockIdx.x) __syncthreads();  // the previous item's readers are done with the staged copy
    const int2 item = item_next;
    if (it + gridDim.x < nitems) item_next = p.items[it + gridDim.x];  // in flight during this item
    const int X = item.x;
    const int o0 = p.off[X];
    const int d = p.off[X + 1] - o0;
    const int ld4 = vec_ld4(d);
    const int dp = (d + 3) & ~3;
    const int *g_nbr = p.nbr + o0;

    const unsigned long long ncomb = p.binom[(size_t)d * kBinomStride + L];  // (uniform: scalar load)
    const unsigned long long r0 = (unsigned long long)item.y * p.chunk;
    const unsigned long long cntr = min(p.chunk, ncomb - r0);

    for (int k = tid; k <= d; k += THREADS)
    {
        const int v = (k < d) ? g_nbr[k] : X;
        s_nbr[k] = v;
        if constexpr (MODE == 1) s_ti[k] = p.time_index[v];
    }
    for (int k = tid; k < dp; k += THREADS)
    {
        unsigned rel = 0u;  // padding entries are "decided"
        float rx = 0.0f;
        if (k < d)
        {
            const int y = g_nbr[k];
            rx = p.C[(size_t)X * n + y];
            if constexpr (MODE == 0)
            {
                // 0 = decided below this chunk; otherwise (lowest passing rank - r0 + 1), saturated
                const unsigned long long b = p.best[o0 + k];
                rel = (b == kNone) ? 0xffffffffu : ((b < r0) ? 0u : (unsigned)min(b - r0 + 1ull, 0xfffffffeull));
            }
            else
            {
                const unsigned long long wv = p.adj[(size_t)X * p.words + (y >> 6)];
                rel = ((wv >> (y & 63)) & 1ull) ? 0xffffffffu : 0u;
            }
        }
        s_rel[k] = rel;
        s_rowx[k] = rx;
    }
    {
        // element (row i, col j) of C[adj(X)]^2 at s_sub[j * ld4 + i]; rows/cols 0..d-1 (X itself is s_rowx).  The indices
        // come straight from the global list (cache hits behind the first touch) rather than from the LDS copy above, so
        // the gathers do not wait for a barrier; kU of them are in flight per lane.
        constexpr int kU = 8;
        const int total = d * dp;
        const int di = THREADS % dp, dj = THREADS / dp;  // one step of THREADS elements in (j, i) coordinates
        int j = tid / dp, i = tid - j * dp;
        for (int e0 = tid; e0 < total; e0 += THREADS * kU)
        {
            float v[kU];
            int at[kU];
#pragma unroll
            for (int u = 0; u < kU; u++)
            {
                const bool in = (e0 + u * THREADS < total);
                at[u] = in ? j * ld4 + i : -1;
                // unconditional requests (indices clamped into the list): a conditional load would make the compiler
                // drain every outstanding request at the join
                const float c = p.C[(size_t)g_nbr[min(i, d - 1)] * n + g_nbr[min(j, d - 1)]];
                v[u] = (i < d) ? c : 0.0f;
                i += di;
                j += dj;
                if (i >= dp)
                {
                    i -= dp;
                    j++;
                }
            }
#pragma unroll
            for (int u = 0; u < kU; u++)
                if (at[u] >= 0) s_sub[at[u]] = v[u];
        }
    }
    __syncthreads();

    const unsigned long long q = (cntr + THREADS - 1) / THREADS;
    const unsigned long long lo = r0 + (unsigned long long)tid * q;
    const unsigned long long hi = min(r0 + cntr, lo + q);
    const float t2lo = p.t2 * (1.0f - kBeta), t2hi = p.t2 * (1.0f + kBeta);
    [[maybe_unused]] const int tiX = (MODE == 1) ? s_ti[d] : 0;

    if (lo < hi)
    {
        int idx[L];
        unrank_comb_lds<L>(lo, d, s_bin, ldb, idx);
        for (unsigned long long rank = lo; rank < hi; rank++)
        {
            const unsigned myrel = (unsigned)(rank - r0) + 1u;  // >= 1: padding and decided entries (0) are never live
            SubsetFast<L> fx;
            {
                float cl[SubsetFast<L>::NL], m1x[L];
#pragma unroll
                for (int a = 0; a < L; a++) m1x[a] = s_rowx[idx[a]];
#pragma unroll
                for (int a = 1; a < L; a++)
#pragma unroll
                    for (int b = 0; b < a; b++) cl[a * (a - 1) / 2 + b] = s_sub[idx[a] * ld4 + idx[b]];  // C[S_b, S_a]
                fx.prepare(cl, m1x);
            }
            nsub++;
            [[maybe_unused]] int tmaxS = 0;
            if constexpr (MODE == 1)
            {
                tmaxS = s_ti[idx[0]];
#pragma unroll
                for (int a = 1; a < L; a++) tmaxS = max(tmaxS, s_ti[idx[a]]);
            }
            const float *col[L];
#pragma unroll
            for (int a = 0; a < L; a++) col[a] = s_sub + idx[a] * ld4;
            const f2 h00v = {fx.h00, fx.h00};
            bool anyalive = false;
            // members of S as a bit mask over the list positions (the first degree class has at most 39 of them)
            [[maybe_unused]] unsigned long long smask = 0ull;
            if constexpr (THREADS == 64)
            {
#pragma unroll
                for (int a = 0; a < L; a++) smask |= 1ull << idx[a];
            }
            for (int g = 0; g < dp; g += 4)
            {
                const u4 rel = *reinterpret_cast<const u4 *>(s_rel + g);
                unsigned livem = (rel.x >= myrel ? 1u : 0u) | (rel.y >= myrel ? 2u : 0u) | (rel.z >= myrel ? 4u : 0u) |
                                 (rel.w >= myrel ? 8u : 0u);
                anyalive |= (livem != 0u);
                // members of S inside this group of four are not tested
                if constexpr (THREADS == 64)
                    livem &= ~((unsigned)(smask >> g) & 15u);
                else
                {
#pragma unroll
                    for (int a = 0; a < L; a++)
                        if ((idx[a] >> 2) == (g >> 2)) livem &= ~(1u << (idx[a] & 3));
                }
                if constexpr (MODE == 1)
                {
                    if (p.has_ti)
                    {
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (tmaxS > max(tiX, s_ti[min(g + u, d)])) livem &= ~(1u << u);
                    }
                }
                if (livem == 0u) continue;
                // four tests, evaluated as two float2 pairs
                f4 my[L];
#pragma unroll
                for (int a = 0; a < L; a++) my[a] = *reinterpret_cast<const f4 *>(col[a] + g);
                const f4 m0 = *reinterpret_cast<const f4 *>(s_rowx + g);
                f2 bA[L], bB[L];
                f2 h11A = {1.0f, 1.0f}, h11B = {1.0f, 1.0f};
                f2 h01A = {m0.x, m0.y}, h01B = {m0.z, m0.w};
#pragma unroll
                for (int i = 0; i < L; i++)
                {
                    f2 sA = {my[i].x, my[i].y}, sB = {my[i].z, my[i].w};
#pragma unroll
                    for (int k = 0; k < i; k++)
                    {
                        const float fik = fx.f[i * (i - 1) / 2 + k];
                        const f2 nf = {-fik, -fik};
                        sA = __builtin_elementwise_fma(nf, bA[k], sA);
                        sB = __builtin_elementwise_fma(nf, bB[k], sB);
                    }
                    const f2 iv = {fx.invd[i], fx.invd[i]};
                    bA[i] = sA * iv;
                    bB[i] = sB * iv;
                    h11A = __builtin_elementwise_fma(-bA[i], bA[i], h11A);
                    h11B = __builtin_elementwise_fma(-bB[i], bB[i], h11B);
                    const f2 na = {-fx.a[i], -fx.a[i]};
                    h01A = __builtin_elementwise_fma(na, bA[i], h01A);
                    h01B = __builtin_elementwise_fma(na, bB[i], h01B);
                }
                const f2 prodA = h00v * h11A, prodB = h00v * h11B;
                const f2 lhsA = h01A * h01A, lhsB = h01B * h01B;
                const f2 loA = prodA * t2lo, loB = prodB * t2lo, hiA = prodA * t2hi, hiB = prodB * t2hi;
                ntests += __popc(livem);
                // The common outcome first: every live test of the group certainly fails (edge stays).  One margin per
                // test, rho^2 side and conditioning side both > 0 (a test exactly on the conditioning bound counts as "not
                // certain" here and is sorted out by the full masks below, which apply the rule of ci_fast.h literally).
                unsigned failq;
                {
                    const f2 dA = lhsA - hiA, dB = lhsB - hiB;
                    const f2 eA = h11A - kCondMin, eB = h11B - kCondMin;
                    // (two comparisons per test, not a min: a NaN on either side must read as "not certain")
                    failq = ((dA.x > 0.0f && eA.x > 0.0f) ? 1u : 0u) | ((dA.y > 0.0f && eA.y > 0.0f) ? 2u : 0u) |
                            ((dB.x > 0.0f && eB.x > 0.0f) ? 4u : 0u) | ((dB.y > 0.0f && eB.y > 0.0f) ? 8u : 0u);
                }
                if (!fx.ill && (livem & ~failq) == 0u) continue;
                // verdict masks (bit u = Y index g+u): pass = certainly separated, fail = certainly not
                unsigned passm = 0u, failm = 0u;
                passm |= (h11A.x >= kCondMin && lhsA.x < loA.x) ? 1u : 0u;
                passm |= (h11A.y >= kCondMin && lhsA.y < loA.y) ? 2u : 0u;
                passm |= (h11B.x >= kCondMin && lhsB.x < loB.x) ? 4u : 0u;
                passm |= (h11B.y >= kCondMin && lhsB.y < loB.y) ? 8u : 0u;
                failm |= (h11A.x >= kCondMin && lhsA.x > hiA.x) ? 1u : 0u;
                failm |= (h11A.y >= kCondMin && lhsA.y > hiA.y) ? 2u : 0u;
                failm |= (h11B.x >= kCondMin && lhsB.x > hiB.x) ? 4u : 0u;
                failm |= (h11B.y >= kCondMin && lhsB.y > hiB.y) ? 8u : 0u;
                if (fx.ill)
                {
                    passm = 0u;
                    failm = 0u;
                }
                unsigned todo = livem & ~failm;  // separated or uncertain: leaves the straight line
                while (todo)
                {
                    const int u = __ffs(todo) - 1;
                    todo &= todo - 1;
                    const int k2 = g + u;
                    if ((passm >> u) & 1u)
                    {
                        bool first;
                        if constexpr (MODE == 0)
                        {
                            const unsigned long long old = atomicMin(&p.best[o0 + k2], rank);
                            atomicMin(&s_rel[k2], myrel);
                            first = (old == kNone);
                        }
                        else
                        {
                            first = clear_edge(p.adj, p.deg, p.words, X, s_nbr[k2]);
                            s_rel[k2] = 0u;
                        }
                        if (first) nrem++;
                    }
                    else
                    {
                        const unsigned long long qi = atomicAdd(&p.cnt->qcount, 1ull);
                        if (qi < p.qcap)
                        {
                            RecheckEntry en;
                            en.x = X;
                            en.k2 = k2;
                            en.rank = rank;
                            p.queue[qi] = en;
                        }
                    }
                }
            }
            if (!anyalive) break;
            // next combination
            {
                int i = L - 1;
                while (i >= 0 && idx[i] == d - L + i) i--;
                if (i < 0) break;
                idx[i]++;
                for (int j2 = i + 1; j2 < L; j2++) idx[j2] = idx[j2 - 1] + 1;
            }
        }
    }
    }  // work items
    for (int o = 32; o > 0; o >>= 1)
    {
        ntests += __shfl_xor(ntests, o);
        nsub += __shfl_xor(nsub, o);
        nrem += __shfl_xor(nrem, o);
    }
    __syncthreads();
    if ((tid & 63) == 0)
    {
        if (ntests) atomicAdd(&s_cnt[0], ntests);
        if (nsub) atomicAdd(&s_cnt[1], nsub);
        if (MODE == 1 && nrem) atomicAdd(&s_cnt[2], nrem);  // Skeleton mode counts removals when it finalises the level
    }
    __syncthreads();
    if (tid == 0)
    {
        unsigned long long *sl = p.slots + (size_t)(blockIdx.x & (kCounterSlots - 1)) * 4;
        for (int i = 0; i < 3; i++)
            if (s_cnt[i]) atomicAdd(&sl[i], s_cnt[i]);
    }
}

size_t sweep_vec_lds_bytes(int cls) { return vec_layout(kClassCap[cls]).total; }

template <int L, int MODE, int THREADS>
static hipError_t launch_vec_LT(const SweepParams &p, int cls, hipStream_t st)
{
    const size_t lds = vec_layout(kClassCap[cls]).total;
    auto kfn = sweep_vec_kernel<L, MODE, THREADS>;
    if (lds > 64 * 1024)
    {
        hipError_t e =
            hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const unsigned grid = (unsigned)std::min<long long>(persistent_grid(reinterpret_cast<const void *>(kfn), THREADS, lds),
                                                        std::max<long long>(p.grid_cap, 1));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(THREADS), lds, st, p);
    return hipGetLastError();
}

template <int L, int MODE>
static hipError_t launch_vec_L(const SweepParams &p, int cls, int threads, hipStream_t st)
{
    if (threads == 64) return launch_vec_LT<L, MODE, 64>(p, cls, st);
    if (threads == 128) return launch_vec_LT<L, MODE, 128>(p, cls, st);
    return launch_vec_LT<L, MODE, 256>(p, cls, st);
}

hipError_t launch_sweep_vec(int mode, int L, const SweepParams &p, int cls, int threads, hipStream_t st)
{
    switch (L)
    {
#define CUSK_CASE(LL) \
    case LL: return mode == 0 ? launch_vec_L<LL, 0>(p, cls, threads, st) : launch_vec_L<LL, 1>(p, cls, threads, st);
        CUSK_CASE(2)
        CUSK_CASE(3)
        CUSK_CASE(4)
        CUSK_CASE(5)
        CUSK_CASE(6)
        CUSK_CASE(7)
        CUSK_CASE(8)
        // levels >= kVecMaxLevel (9) run on sweep_fast_kernel: the factor alone is l(l-1)/2 registers there, the
        // float2 form drops to one wave per SIMD (and spills at l = 14) and loses to the scalar form
#undef CUSK_CASE
        static_assert(kVecMaxLevel == 9, "instantiate the levels below kVecMaxLevel");
    }
    return hipErrorInvalidValue;
}

}  // namespace cusk
