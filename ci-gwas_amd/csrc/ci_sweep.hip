// ci_sweep.hip -- level-ordered conditional-independence sweep for gfx950.
//
// Replaces the level loop and kernels of the reference's two engines
// (/root/reference/cusk/src/cuPC-S.cu:61-450 `Skeleton`, cal_Indepl0..14;
//  src/hetcor-cuPC-S.cu:75-341 `hetcor_skeleton`, cal_Indepl0_ess..14_ess;
//  src/cuPC-S.cu:6355-6432 scan_compact) with an MI355X-first layout:
//
//   * adjacency is an n x ceil(n/64) uint64 bitmap in HBM (n^2/8 bytes instead
//     of the reference's three n^2 int32 arrays); level 0 writes it with
//     wavefront ballots,
//   * per level the bitmap is compacted into CSR neighbour lists (ascending
//     index, frozen for the level) by one wavefront per row,
//   * the sweep is tiled into work items (row X, range of combination ranks);
//     a 256-thread workgroup stages the (d+1)^2 sub-matrix C[adj(X)+X]^2 into
//     LDS once (coalesced-ish row gathers) and every test then runs from LDS:
//     lane <-> conditioning set, loop over Y shares the l x l inverse,
//   * rows are bucketed into degree classes so each launch requests only the
//     LDS its class needs; hubs that do not fit 160 KB read C through L2,
//   * separating sets are selected deterministically (lowest combination rank
//     per ORDERED pair, 64-bit atomicMin) and stored sparsely,
//   * 64-bit combination ranks; per-level test / subset counters.
//
// Arithmetic: ci_exact.h (reference operation order, -ffp-contract=off).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "ci_exact.h"
#include "ci_fast.h"
#include "cusk_internal.h"

namespace cusk {

// ---------------------------------------------------------------------------
// level 0
// ---------------------------------------------------------------------------

// adjacency bitmap <- complete graph without self loops (Skeleton) or the caller's G (hetcor)
__global__ void init_bits_kernel(unsigned long long *adj, const int *Ginit, int n, int words)
{
    const int row = blockIdx.x;
    for (int w = threadIdx.x; w < words; w += blockDim.x)
    {
        unsigned long long bits = 0;
        const int base = w * 64;
        if (Ginit == nullptr)
        {
            int valid = n - base;
            bits = (valid >= 64) ? ~0ull : ((1ull << valid) - 1ull);
        }
        else
        {
            for (int b = 0; b < 64 && base + b < n; b++)
                if (Ginit[(size_t)row * n + base + b] == 1) bits |= (1ull << b);
        }
        if (row >= base && row < base + 64) bits &= ~(1ull << (row - base));
        adj[(size_t)row * words + w] = bits;
    }
}

// One 64x64 tile of the upper triangle per workgroup (4 waves x 16 rows, lane = column).
// ESS: per-pair effective sample size (hetcor), else the fixed threshold th.
template <bool ESS>
__global__ void __launch_bounds__(256) level0_kernel(const float *__restrict__ C, const float *__restrict__ N,
                                                      unsigned long long *adj, int n, int words, float th,
                                                      int tiles, int *asym_flag)
{
    __shared__ unsigned long long s_col[64];
    __shared__ float s_t[64][65];  // mirrored tile C[bj rows][bi cols], for the symmetry check
    // linear tile id -> (bi <= bj)
    int t = blockIdx.x;
    int bi = 0;
    {
        // rows of the tile triangle have tiles-bi entries
        int rem = t;
        int len = tiles;
        while (rem >= len)
        {
            rem -= len;
            len--;
            bi++;
        }
        t = bi + rem;
    }
    const int bj = t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) s_col[threadIdx.x] = 0ull;
    for (int rr = 0; rr < 16; rr++)
    {
        const int r = wave * 16 + rr;
        const int jr = bj * 64 + r, ic = bi * 64 + lane;
        s_t[r][lane] = (jr < n && ic < n) ? C[(size_t)jr * n + ic] : 0.0f;
    }
    __syncthreads();
    const int j = bj * 64 + lane;
    unsigned long long colbits = 0ull;
    for (int rr = 0; rr < 16; rr++)
    {
        const int r = wave * 16 + rr;
        const int i = bi * 64 + r;
        bool rm = false;
        if (i < n && j < n && i < j)
        {
            float c = C[(size_t)i * n + j];
            // is the matrix bitwise symmetric?  (lets level 1 read only the upper triangle)
            float ct = s_t[lane][r];
            if (__float_as_uint(c) != __float_as_uint(ct) && !((c != c) && (ct != ct))) *asym_flag = 1;
            float lth = th;
            if constexpr (ESS) lth = (float)((double)th / sqrt((double)N[(size_t)i * n + j] - 3.0));
            rm = z_below<false>(c, lth);
        }
        unsigned long long m = __ballot(rm);
        if (lane == 0 && m != 0ull) atomicAnd(&adj[(size_t)i * words + bj], ~m);
        if (rm) colbits |= (1ull << r);
    }
    if (colbits) atomicOr(&s_col[lane], colbits);
    __syncthreads();
    if (threadIdx.x < 64)
    {
        unsigned long long m = s_col[threadIdx.x];
        const int jj = bj * 64 + threadIdx.x;
        if (m != 0ull && jj < n) atomicAnd(&adj[(size_t)jj * words + bi], ~m);
    }
}

// ---------------------------------------------------------------------------
// compaction: bitmap -> CSR neighbour lists
// ---------------------------------------------------------------------------

__global__ void degree_kernel(const unsigned long long *__restrict__ adj, int *deg, int n, int words,
                              LevelCounters *cnt)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    int d = 0;
    for (int w = lane; w < words; w += 64) d += __popcll(adj[(size_t)row * words + w]);
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if (lane == 0)
    {
        deg[row] = d;
        if (d > *reinterpret_cast<volatile int *>(&cnt->maxdeg)) atomicMax(&cnt->maxdeg, d);
    }
}

// single-workgroup exclusive scan of deg[0..n) into off[0..n]; total to counters
__global__ void __launch_bounds__(1024) scan_kernel(const int *deg, int *off, int n, LevelCounters *cnt)
{
    __shared__ long long s_part[1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    long long s = 0;
    for (int i = lo; i < hi; i++) s += deg[i];
    s_part[threadIdx.x] = s;
    __syncthreads();
    for (int step = 1; step < 1024; step <<= 1)
    {
        long long v = (threadIdx.x >= step) ? s_part[threadIdx.x - step] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    long long run = (threadIdx.x == 0) ? 0 : s_part[threadIdx.x - 1];
    for (int i = lo; i < hi; i++)
    {
        off[i] = (int)run;
        run += deg[i];
    }
    if (threadIdx.x == 1023)
    {
        off[n] = (int)s_part[1023];
        cnt->total_edges = s_part[1023];
    }
}

struct RowInfo
{
    int cls;     // degree class, -1 = no work this level
    int base;    // first item of the row inside its class list
    int nchunks;
    int pad;
};

// one wave per row: write ascending neighbour indices, count the row's work items
__global__ void fill_nbr_kernel(const unsigned long long *__restrict__ adj, const int *__restrict__ off,
                                int *nbr, int n, int words, int L, const unsigned long long *__restrict__ binom,
                                unsigned long long chunk, int staged_classes, int pair_mode, RowInfo *rowinfo,
                                LevelCounters *cnt)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const int o0 = off[row];
    int run = 0;
    for (int w0 = 0; w0 < words; w0 += 64)
    {
        const int w = w0 + lane;
        unsigned long long bits = (w < words) ? adj[(size_t)row * words + w] : 0ull;
        int c = __popcll(bits);
        int incl = c;
        for (int o = 1; o < 64; o <<= 1)
        {
            int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        int pos = o0 + run + incl - c;
        while (bits)
        {
            int b = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            nbr[pos++] = w * 64 + b;
        }
        run += __shfl(incl, 63);
    }
    if (lane == 0)
    {
        const int d = run;
        RowInfo ri;
        ri.cls = -1;
        ri.base = 0;
        ri.nchunks = 0;
        ri.pad = 0;
        if (d > L)
        {
            // work units of the row: conditioning sets, or unordered neighbour pairs for the level-1 pair kernel
            unsigned long long nc = pair_mode ? (unsigned long long)d * (d - 1) / 2 : binom[(size_t)d * (L + 1) + L];
            if (nc >= (1ull << 62))
            {
                cnt->overflow = 1;
            }
            else
            {
                unsigned long long k = (nc + chunk - 1) / chunk;
                int cls = 0;
                while (d > kClassCap[cls]) cls++;
                if (cls >= staged_classes) cls = kNumClasses - 1;
                ri.cls = cls;
                ri.nchunks = (int)k;
            }
        }
        rowinfo[row] = ri;
    }
}

// single workgroup: per-class exclusive scan of the rows' item counts -> RowInfo.base, class totals
__global__ void __launch_bounds__(1024) item_scan_kernel(RowInfo *rowinfo, int n, LevelCounters *cnt)
{
    __shared__ long long s_part[kNumClasses][1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    long long sum[kNumClasses];
#pragma unroll
    for (int c = 0; c < kNumClasses; c++) sum[c] = 0;
    for (int i = lo; i < hi; i++)
    {
        const RowInfo ri = rowinfo[i];
#pragma unroll
        for (int c = 0; c < kNumClasses; c++)
            if (ri.cls == c) sum[c] += ri.nchunks;
    }
#pragma unroll
    for (int c = 0; c < kNumClasses; c++) s_part[c][threadIdx.x] = sum[c];
    __syncthreads();
    for (int step = 1; step < 1024; step <<= 1)
    {
        long long v[kNumClasses];
#pragma unroll
        for (int c = 0; c < kNumClasses; c++) v[c] = (threadIdx.x >= step) ? s_part[c][threadIdx.x - step] : 0;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < kNumClasses; c++) s_part[c][threadIdx.x] += v[c];
        __syncthreads();
    }
    long long run[kNumClasses];
#pragma unroll
    for (int c = 0; c < kNumClasses; c++) run[c] = (threadIdx.x == 0) ? 0 : s_part[c][threadIdx.x - 1];
    for (int i = lo; i < hi; i++)
    {
        RowInfo ri = rowinfo[i];
#pragma unroll
        for (int c = 0; c < kNumClasses; c++)
            if (ri.cls == c)
            {
                ri.base = (int)run[c];
                run[c] += ri.nchunks;
            }
        rowinfo[i] = ri;
    }
    if (threadIdx.x == 1023)
    {
#pragma unroll
        for (int c = 0; c < kNumClasses; c++) cnt->class_items[c] = s_part[c][1023];
    }
}

__global__ void fill_items_kernel(const RowInfo *__restrict__ rowinfo, int n, int2 *i0, int2 *i1, int2 *i2,
                                  int2 *i3, int2 *i4)
{
    const int row = blockIdx.x;
    const RowInfo ri = rowinfo[row];
    if (ri.cls < 0) return;
    int2 *dst = ri.cls == 0 ? i0 : ri.cls == 1 ? i1 : ri.cls == 2 ? i2 : ri.cls == 3 ? i3 : i4;
    for (int c = threadIdx.x; c < ri.nchunks; c += blockDim.x) dst[ri.base + c] = make_int2(row, c);
}

__global__ void fill_u64_kernel(unsigned long long *p, size_t count, unsigned long long v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = v;
}

// ---------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------

struct RecheckEntry
{
    int x;
    int k2;
    unsigned long long rank;
};

struct SweepParams
{
    const float *C;
    const float *Ness;  // per-pair effective sample sizes (HET) or nullptr
    int n;
    const int *off;
    const int *nbr;
    unsigned long long *best;  // MODE 0: lowest passing rank per CSR slot
    unsigned long long *adj;   // MODE 1: live adjacency bitmap
    int words;
    const int2 *items;
    const unsigned long long *binom;  // [(a)*(L+1)+b] = C(a,b)
    const int *time_index;            // MODE 1, device, n entries
    float th;                         // MODE 0: Th[l]; MODE 1 uniform ESS: th/sqrt(mean_ess-l-3); HET: alpha/2 quantile
    unsigned long long chunk;
    int cap;  // class capacity (LDS carve), ignored when !STAGED
    LevelCounters *cnt;
    unsigned long long *slots;  // kCounterSlots x 4 spread counters: tests, subsets, removed, violations
    // fast path (ci_fast.h)
    float t2;                  // tanh(th)^2 for fixed-threshold modes
    RecheckEntry *queue;       // tests that need the exact path
    unsigned long long qcap;
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// LDS carve for a class capacity
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

// mean_ess of hetcor-cuPC-S.cu:3068-3088: entries truncated to int (v_cvt_i32_f32
// saturates and maps NaN to 0, as the reference's GPU does), float running sum in
// the reference's pair order, divided by the pair count.
__device__ __forceinline__ float ess_term(float e) { return (float)(int)e; }

template <int L, int MODE, bool HET, bool STAGED>
__global__ void __launch_bounds__(kThreads) sweep_kernel(SweepParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_cnt[3];

    const int2 item = p.items[blockIdx.x];
    const int X = item.x;
    const int o0 = p.off[X];
    const int d = p.off[X + 1] - o0;
    const int n = p.n;
    const int tid = threadIdx.x;
    const int ld = (d + 1) | 1;

    const LdsLayout lay = lds_layout(STAGED ? p.cap : 0, HET);
    int *s_nbr = reinterpret_cast<int *>(smem + lay.nbr);
    unsigned long long *s_best = reinterpret_cast<unsigned long long *>(smem + lay.best);
    int *s_ti = reinterpret_cast<int *>(smem + lay.ti);
    float *s_sub = reinterpret_cast<float *>(smem + lay.sub);
    float *s_ess = reinterpret_cast<float *>(smem + lay.ess);
    const int *g_nbr = p.nbr + o0;

    if (tid < 3) s_cnt[tid] = 0ull;
    if constexpr (STAGED)
    {
        for (int k = tid; k <= d; k += kThreads)
        {
            int v = (k < d) ? g_nbr[k] : X;
            s_nbr[k] = v;
            if constexpr (MODE == 1) s_ti[k] = p.time_index[v];
        }
        for (int k = tid; k < d; k += kThreads)
        {
            if constexpr (MODE == 0)
                s_best[k] = p.best[o0 + k];
            else
            {
                int y = g_nbr[k];
                unsigned long long wv = p.adj[(size_t)X * p.words + (y >> 6)];
                s_best[k] = ((wv >> (y & 63)) & 1ull) ? kNone : 0ull;  // 0 = already removed
            }
        }
        __syncthreads();
        const int dd = d + 1;
        for (int e = tid; e < dd * dd; e += kThreads)
        {
            int i = e / dd, j = e - i * dd;
            size_t g = (size_t)s_nbr[i] * n + s_nbr[j];
            s_sub[i * ld + j] = p.C[g];
            if constexpr (HET) s_ess[i * ld + j] = p.Ness[g];
        }
    }
    __syncthreads();

    // element (i,j) of the row's sub-matrix, i,j in [0,d], index d = X itself
    auto var_of = [&](int i) -> int {
        if constexpr (STAGED)
            return s_nbr[i];
        else
            return (i < d) ? g_nbr[i] : X;
    };
    auto cval = [&](int i, int j) -> float {
        if constexpr (STAGED)
            return s_sub[i * ld + j];
        else
            return p.C[(size_t)var_of(i) * n + var_of(j)];
    };
    auto eval = [&](int i, int j) -> float {
        if constexpr (STAGED)
            return s_ess[i * ld + j];
        else
            return p.Ness[(size_t)var_of(i) * n + var_of(j)];
    };
    auto tix = [&](int i) -> int {
        if constexpr (STAGED)
            return s_ti[i];
        else
            return p.time_index[var_of(i)];
    };
    auto best_of = [&](int k) -> unsigned long long {
        if constexpr (STAGED)
            return s_best[k];
        else if constexpr (MODE == 0)
            return p.best[o0 + k];
        else
        {
            int y = g_nbr[k];
            unsigned long long wv = p.adj[(size_t)X * p.words + (y >> 6)];
            return ((wv >> (y & 63)) & 1ull) ? kNone : 0ull;
        }
    };

    const unsigned long long ncomb = p.binom[(size_t)d * (L + 1) + L];
    const unsigned long long r0 = (unsigned long long)item.y * p.chunk;
    const unsigned long long cntr = min(p.chunk, ncomb - r0);
    const unsigned long long q = (cntr + kThreads - 1) / kThreads;
    unsigned long long lo = r0 + (unsigned long long)tid * q;
    unsigned long long hi = min(r0 + cntr, lo + q);

    unsigned long long ntests = 0, nsub = 0, nrem = 0;
    if (lo < hi)
    {
        // unrank lo (0-based, lexicographic) into ascending positions idx[0..L)
        int idx[L];
        {
            unsigned long long rem = lo;
            int c = 0;
#pragma unroll
            for (int i = 0; i < L; i++)
            {
                while (true)
                {
                    unsigned long long b = p.binom[(size_t)(d - 1 - c) * (L + 1) + (L - 1 - i)];
                    if (rem < b) break;
                    rem -= b;
                    c++;
                }
                idx[i] = c;
                c++;
            }
        }
        for (unsigned long long rank = lo; rank < hi; rank++)
        {
            // build the subset's matrices
            float m2[(L > 1) ? L * L : 1], m1x[L];
#pragma unroll
            for (int a = 0; a < L; a++) m1x[a] = cval(d, idx[a]);
            if constexpr (L >= 2)
            {
#pragma unroll
                for (int a = 0; a < L; a++)
#pragma unroll
                    for (int b = 0; b < L; b++)
                    {
                        if (a == b)
                            m2[a * L + b] = 1.0f;
                        else if (a < b)
                            m2[a * L + b] = cval(idx[a], idx[b]);
                        else
                            m2[a * L + b] = cval(idx[b], idx[a]);
                    }
            }
            SubsetExact<L> cx;
            cx.prepare(m2, m1x);
            nsub++;
            int tmaxS = 0;
            float essS = 0.0f;  // unused placeholder
            (void)essS;
            if constexpr (MODE == 1)
            {
                tmaxS = tix(idx[0]);
#pragma unroll
                for (int a = 1; a < L; a++) tmaxS = max(tmaxS, tix(idx[a]));
            }
            bool anyalive = false;
            for (int k2 = 0; k2 < d; k2++)
            {
                const unsigned long long bk = best_of(k2);
                const bool live = (MODE == 0) ? (bk >= rank) : (bk == kNone);
                anyalive |= live;
                if (!live) continue;
                bool inS = false;
#pragma unroll
                for (int a = 0; a < L; a++) inS |= (idx[a] == k2);
                if (inS) continue;
                if constexpr (MODE == 1)
                {
                    if (tmaxS > max(tix(d), tix(k2))) continue;
                }
                float m1y[L];
#pragma unroll
                for (int a = 0; a < L; a++) m1y[a] = cval(k2, idx[a]);
                const float rho = cx.rho(cval(d, k2), m1y);
                ntests++;
                float lth = p.th;
                if constexpr (HET)
                {
                    // pair order of mean_ess: vix = [X, Y, S0, S1, ...], i over vix, j < i
                    float s = 0.0f;
                    s += ess_term(eval(k2, d));
#pragma unroll
                    for (int a = 0; a < L; a++)
                    {
                        s += ess_term(eval(idx[a], d));
                        s += ess_term(eval(idx[a], k2));
#pragma unroll
                        for (int b = 0; b < a; b++) s += ess_term(eval(idx[a], idx[b]));
                    }
                    float me = s / (float)((L + 2) * (L + 1) / 2);
                    lth = (float)((double)p.th / sqrt((double)me - (double)L - 3.0));
                }
                const bool pass = z_below<L == 1>(rho, lth);
                if (pass)
                {
                    if constexpr (MODE == 0)
                    {
                        unsigned long long old = atomicMin(&p.best[o0 + k2], rank);
                        if constexpr (STAGED) atomicMin(&s_best[k2], rank);
                        if (old == kNone) nrem++;
                    }
                    else
                    {
                        const int Y = var_of(k2);
                        unsigned long long old =
                            atomicAnd(&p.adj[(size_t)X * p.words + (Y >> 6)], ~(1ull << (Y & 63)));
                        atomicAnd(&p.adj[(size_t)Y * p.words + (X >> 6)], ~(1ull << (X & 63)));
                        if constexpr (STAGED) s_best[k2] = 0ull;
                        if ((old >> (Y & 63)) & 1ull) nrem++;
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
    // counters: wave reduce, then LDS, then one global atomic per workgroup
    for (int o = 32; o > 0; o >>= 1)
    {
        ntests += __shfl_xor(ntests, o);
        nsub += __shfl_xor(nsub, o);
        nrem += __shfl_xor(nrem, o);
    }
    if ((tid & 63) == 0)
    {
        atomicAdd(&s_cnt[0], ntests);
        atomicAdd(&s_cnt[1], nsub);
        atomicAdd(&s_cnt[2], nrem);
    }
    __syncthreads();
    if (tid == 0)
    {
        unsigned long long *sl = p.slots + (size_t)(blockIdx.x & (kCounterSlots - 1)) * 4;
        if (s_cnt[0]) atomicAdd(&sl[0], s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&sl[1], s_cnt[1]);
        if (s_cnt[2]) atomicAdd(&sl[2], s_cnt[2]);
    }
}


// Level 1 on a symmetric matrix (fixed or uniform-ESS threshold).  A level-1 test
// (X ; Y | S) needs C[X,Y], C[X,S] and C[Y,S]; the first two live in row X and are staged once,
// the third is used by exactly two tests, (X;Y|S) and (X;S|Y), so staging a (d+1)^2 sub-matrix
// buys no reuse.  Lane <-> unordered neighbour pair {a<b}: ONE 4-byte gather of the upper-triangle
// element C[min,max] feeds both tests, pairs whose two tests are already decided are skipped
// without touching memory, and with ~20 VGPRs the kernel runs at full occupancy to hide the
// gather latency.  Arithmetic is the exact level-1 formula (cuPC-S.cu:561-566), so no recheck.
template <int MODE>
__global__ void __launch_bounds__(kThreads) level1_pair_kernel(SweepParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_cnt[3];
    const int2 item = p.items[blockIdx.x];
    const int X = item.x;
    const int o0 = p.off[X];
    const int d = p.off[X + 1] - o0;
    const int n = p.n;
    const int tid = threadIdx.x;
    unsigned long long *s_best = reinterpret_cast<unsigned long long *>(smem);
    int *s_nbr = reinterpret_cast<int *>(smem + sizeof(unsigned long long) * d);
    float *s_m1x = reinterpret_cast<float *>(s_nbr + d);
    int *s_ti = reinterpret_cast<int *>(s_m1x + d);
    const int *g_nbr = p.nbr + o0;
    if (tid < 3) s_cnt[tid] = 0ull;
    for (int k = tid; k < d; k += kThreads)
    {
        const int y = g_nbr[k];
        s_nbr[k] = y;
        s_m1x[k] = p.C[(size_t)X * n + y];
        if constexpr (MODE == 0)
            s_best[k] = p.best[o0 + k];
        else
        {
            unsigned long long wv = p.adj[(size_t)X * p.words + (y >> 6)];
            s_best[k] = ((wv >> (y & 63)) & 1ull) ? kNone : 0ull;
            s_ti[k] = p.time_index[y];
        }
    }
    __syncthreads();
    [[maybe_unused]] int tiX = 0;
    if constexpr (MODE == 1) tiX = p.time_index[X];

    const unsigned long long npairs = (unsigned long long)d * (d - 1) / 2;
    const unsigned long long r0 = (unsigned long long)item.y * p.chunk;
    const unsigned long long cntr = min(p.chunk, npairs - r0);
    const unsigned long long q = (cntr + kThreads - 1) / kThreads;
    unsigned long long lo = r0 + (unsigned long long)tid * q;
    const unsigned long long hi = min(r0 + cntr, lo + q);
    unsigned long long ntests = 0, nrem = 0;
    if (lo < hi)
    {
        // unrank the pair: rows of the strict upper triangle have d-1-a entries
        int a = 0;
        {
            unsigned long long rem = lo;
            // closed form start then fix up (exact in integers)
            double dd = (double)d - 0.5;
            int guess = (int)(dd - sqrt(dd * dd - 2.0 * (double)rem));
            if (guess < 0) guess = 0;
            if (guess > d - 2) guess = d - 2;
            a = guess;
            auto start_of = [&](int aa) -> unsigned long long {
                return (unsigned long long)aa * d - (unsigned long long)aa * (aa + 1) / 2;
            };
            while (a > 0 && start_of(a) > rem) a--;
            while (a < d - 2 && start_of(a + 1) <= rem) a++;
            lo = rem - start_of(a);  // reuse lo as offset inside row a
        }
        int b = a + 1 + (int)lo;
        const unsigned long long count = hi - (r0 + (unsigned long long)tid * q);
        auto apply = [&](int ky, int ks) {
            // edge X - nbr[ky] is separated by S = nbr[ks]
            if constexpr (MODE == 0)
            {
                unsigned long long old = atomicMin(&p.best[o0 + ky], (unsigned long long)ks);
                atomicMin(&s_best[ky], (unsigned long long)ks);
                if (old == kNone) nrem++;
            }
            else
            {
                const int Y = s_nbr[ky];
                unsigned long long old = atomicAnd(&p.adj[(size_t)X * p.words + (Y >> 6)], ~(1ull << (Y & 63)));
                atomicAnd(&p.adj[(size_t)Y * p.words + (X >> 6)], ~(1ull << (X & 63)));
                s_best[ky] = 0ull;
                if ((old >> (Y & 63)) & 1ull) nrem++;
            }
        };
        for (unsigned long long it = 0; it < count; it++)
        {
            bool needA, needB;  // A: Y = a, S = b ; B: Y = b, S = a
            if constexpr (MODE == 0)
            {
                needA = s_best[a] >= (unsigned long long)b;
                needB = s_best[b] >= (unsigned long long)a;
            }
            else
            {
                needA = (s_best[a] == kNone) && !(s_ti[b] > max(tiX, s_ti[a]));
                needB = (s_best[b] == kNone) && !(s_ti[a] > max(tiX, s_ti[b]));
            }
            if (needA || needB)
            {
                const int ya = s_nbr[a], yb = s_nbr[b];  // ascending lists: ya < yb
                const float c = p.C[(size_t)ya * n + yb];
                const float ra = s_m1x[a], rb = s_m1x[b];
                const float hc = 1.0f - (c * c);
                if (needA)
                {
                    const float H00 = 1.0f - (rb * rb);
                    const float H01 = ra - (rb * c);
                    const float rho = H01 / (sqrtf(fabsf(H00)) * sqrtf(fabsf(hc)));
                    ntests++;
                    if (z_below<true>(rho, p.th)) apply(a, b);
                }
                if (needB)
                {
                    const float H00 = 1.0f - (ra * ra);
                    const float H01 = rb - (ra * c);
                    const float rho = H01 / (sqrtf(fabsf(H00)) * sqrtf(fabsf(hc)));
                    ntests++;
                    if (z_below<true>(rho, p.th)) apply(b, a);
                }
            }
            b++;
            if (b == d)
            {
                a++;
                b = a + 1;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        ntests += __shfl_xor(ntests, o);
        nrem += __shfl_xor(nrem, o);
    }
    if ((tid & 63) == 0)
    {
        atomicAdd(&s_cnt[0], ntests);
        atomicAdd(&s_cnt[2], nrem);
    }
    __syncthreads();
    if (tid == 0)
    {
        unsigned long long *sl = p.slots + (size_t)(blockIdx.x & (kCounterSlots - 1)) * 4;
        if (s_cnt[0]) atomicAdd(&sl[0], s_cnt[0]);
        if (s_cnt[2]) atomicAdd(&sl[2], s_cnt[2]);
    }
}

// unrank a 0-based lexicographic combination rank into ascending positions idx[0..L) out of d
template <int L>
__device__ __forceinline__ void unrank_comb(unsigned long long rem, int d, const unsigned long long *__restrict__ binom,
                                            int *idx)
{
    int c = 0;
#pragma unroll
    for (int i = 0; i < L; i++)
    {
        while (true)
        {
            unsigned long long b = binom[(size_t)(d - 1 - c) * (L + 1) + (L - 1 - i)];
            if (rem < b) break;
            rem -= b;
            c++;
        }
        idx[i] = c;
        c++;
    }
}

// Fast sweep for L >= 2 (ci_fast.h): same tiling, staging, lane <-> conditioning-set mapping and
// selection rule as sweep_kernel, but every test is first judged by the register-Cholesky
// filter; only uncertain ones are queued for the exact path.  VALIDATE additionally runs the
// exact arithmetic on every certain verdict and counts disagreements (must stay 0).
template <int L, int MODE, bool HET, bool STAGED, bool VALIDATE>
__global__ void __launch_bounds__(kThreads) sweep_fast_kernel(SweepParams p)
{
    static_assert(L >= 2, "level 1 has its own kernels");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_cnt[4];

    const int2 item = p.items[blockIdx.x];
    const int X = item.x;
    const int o0 = p.off[X];
    const int d = p.off[X + 1] - o0;
    const int n = p.n;
    const int tid = threadIdx.x;
    const int ld = (d + 1) | 1;

    const LdsLayout lay = lds_layout(STAGED ? p.cap : 0, HET);
    int *s_nbr = reinterpret_cast<int *>(smem + lay.nbr);
    unsigned long long *s_best = reinterpret_cast<unsigned long long *>(smem + lay.best);
    int *s_ti = reinterpret_cast<int *>(smem + lay.ti);
    float *s_sub = reinterpret_cast<float *>(smem + lay.sub);
    float *s_ess = reinterpret_cast<float *>(smem + lay.ess);
    const int *g_nbr = p.nbr + o0;

    if (tid < 4) s_cnt[tid] = 0ull;
    if constexpr (STAGED)
    {
        for (int k = tid; k <= d; k += kThreads)
        {
            int v = (k < d) ? g_nbr[k] : X;
            s_nbr[k] = v;
            if constexpr (MODE == 1) s_ti[k] = p.time_index[v];
        }
        for (int k = tid; k < d; k += kThreads)
        {
            if constexpr (MODE == 0)
                s_best[k] = p.best[o0 + k];
            else
            {
                int y = g_nbr[k];
                unsigned long long wv = p.adj[(size_t)X * p.words + (y >> 6)];
                s_best[k] = ((wv >> (y & 63)) & 1ull) ? kNone : 0ull;
            }
        }
        __syncthreads();
        const int dd = d + 1;
        for (int e = tid; e < dd * dd; e += kThreads)
        {
            int i = e / dd, j = e - i * dd;
            size_t g = (size_t)s_nbr[i] * n + s_nbr[j];
            s_sub[i * ld + j] = p.C[g];
            if constexpr (HET) s_ess[i * ld + j] = p.Ness[g];
        }
    }
    __syncthreads();

    auto var_of = [&](int i) -> int {
        if constexpr (STAGED)
            return s_nbr[i];
        else
            return (i < d) ? g_nbr[i] : X;
    };
    auto cval = [&](int i, int j) -> float {
        if constexpr (STAGED)
            return s_sub[i * ld + j];
        else
            return p.C[(size_t)var_of(i) * n + var_of(j)];
    };
    auto eval = [&](int i, int j) -> float {
        if constexpr (STAGED)
            return s_ess[i * ld + j];
        else
            return p.Ness[(size_t)var_of(i) * n + var_of(j)];
    };
    auto tix = [&](int i) -> int {
        if constexpr (STAGED)
            return s_ti[i];
        else
            return p.time_index[var_of(i)];
    };
    auto best_of = [&](int k) -> unsigned long long {
        if constexpr (STAGED)
            return s_best[k];
        else if constexpr (MODE == 0)
            return p.best[o0 + k];
        else
        {
            int y = g_nbr[k];
            unsigned long long wv = p.adj[(size_t)X * p.words + (y >> 6)];
            return ((wv >> (y & 63)) & 1ull) ? kNone : 0ull;
        }
    };

    const unsigned long long ncomb = p.binom[(size_t)d * (L + 1) + L];
    const unsigned long long r0 = (unsigned long long)item.y * p.chunk;
    const unsigned long long cntr = min(p.chunk, ncomb - r0);
    const unsigned long long q = (cntr + kThreads - 1) / kThreads;
    unsigned long long lo = r0 + (unsigned long long)tid * q;
    unsigned long long hi = min(r0 + cntr, lo + q);

    unsigned long long ntests = 0, nsub = 0, nrem = 0, nbad = 0;
    if (lo < hi)
    {
        int idx[L];
        unrank_comb<L>(lo, d, p.binom, idx);
        for (unsigned long long rank = lo; rank < hi; rank++)
        {
            float cl[SubsetFast<L>::NL], m1x[L];
#pragma unroll
            for (int a = 0; a < L; a++) m1x[a] = cval(d, idx[a]);
#pragma unroll
            for (int a = 1; a < L; a++)
#pragma unroll
                for (int b = 0; b < a; b++) cl[a * (a - 1) / 2 + b] = cval(idx[b], idx[a]);
            SubsetFast<L> fx;
            fx.prepare(cl, m1x);
            nsub++;
            [[maybe_unused]] SubsetExact<L> cx;
            if constexpr (VALIDATE)
            {
                float m2[L * L];
#pragma unroll
                for (int a = 0; a < L; a++)
#pragma unroll
                    for (int b = 0; b < L; b++)
                        m2[a * L + b] = (a == b) ? 1.0f : (a < b ? cval(idx[a], idx[b]) : cval(idx[b], idx[a]));
                cx.prepare(m2, m1x);
            }
            int tmaxS = 0;
            [[maybe_unused]] float essS = 0.0f;
            if constexpr (MODE == 1)
            {
                tmaxS = tix(idx[0]);
#pragma unroll
                for (int a = 1; a < L; a++) tmaxS = max(tmaxS, tix(idx[a]));
            }
            if constexpr (HET)
            {
                // filter-only estimate of the subset's part of mean_ess (summation order is free here)
#pragma unroll
                for (int a = 0; a < L; a++)
                {
                    essS += ess_term(eval(idx[a], d));
#pragma unroll
                    for (int b = 0; b < a; b++) essS += ess_term(eval(idx[a], idx[b]));
                }
            }
            bool anyalive = false;
            for (int k2 = 0; k2 < d; k2++)
            {
                const unsigned long long bk = best_of(k2);
                const bool live = (MODE == 0) ? (bk >= rank) : (bk == kNone);
                anyalive |= live;
                if (!live) continue;
                bool inS = false;
#pragma unroll
                for (int a = 0; a < L; a++) inS |= (idx[a] == k2);
                if (inS) continue;
                if constexpr (MODE == 1)
                {
                    if (tmaxS > max(tix(d), tix(k2))) continue;
                }
                ntests++;
                int v = kUnsure;
                float m1y[L];
#pragma unroll
                for (int a = 0; a < L; a++) m1y[a] = cval(k2, idx[a]);
                const float m0 = cval(d, k2);
                [[maybe_unused]] float lthf = 0.0f;
                if (!fx.ill)
                {
                    if constexpr (HET)
                    {
                        float s = essS + ess_term(eval(k2, d));
#pragma unroll
                        for (int a = 0; a < L; a++) s += ess_term(eval(idx[a], k2));
                        const float me = s / (float)((L + 2) * (L + 1) / 2);
                        lthf = p.th * __frsqrt_rn(me - (float)(L + 3));
                        v = fx.verdict_z(m0, m1y, lthf);
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
                        if constexpr (HET)
                        {
                            float s = 0.0f;
                            s += ess_term(eval(k2, d));
#pragma unroll
                            for (int a = 0; a < L; a++)
                            {
                                s += ess_term(eval(idx[a], d));
                                s += ess_term(eval(idx[a], k2));
#pragma unroll
                                for (int b = 0; b < a; b++) s += ess_term(eval(idx[a], idx[b]));
                            }
                            float me = s / (float)((L + 2) * (L + 1) / 2);
                            lth = (float)((double)p.th / sqrt((double)me - (double)L - 3.0));
                        }
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
                        en.x = X;
                        en.k2 = k2;
                        en.rank = rank;
                        p.queue[qi] = en;
                    }
                }
                else if (v == kPass)
                {
                    if constexpr (MODE == 0)
                    {
                        unsigned long long old = atomicMin(&p.best[o0 + k2], rank);
                        if constexpr (STAGED) atomicMin(&s_best[k2], rank);
                        if (old == kNone) nrem++;
                    }
                    else
                    {
                        const int Y = var_of(k2);
                        unsigned long long old =
                            atomicAnd(&p.adj[(size_t)X * p.words + (Y >> 6)], ~(1ull << (Y & 63)));
                        atomicAnd(&p.adj[(size_t)Y * p.words + (X >> 6)], ~(1ull << (X & 63)));
                        if constexpr (STAGED) s_best[k2] = 0ull;
                        if ((old >> (Y & 63)) & 1ull) nrem++;
                    }
                }
            }
            if (!anyalive) break;
            {
                int i = L - 1;
                while (i >= 0 && idx[i] == d - L + i) i--;
                if (i < 0) break;
                idx[i]++;
                for (int j2 = i + 1; j2 < L; j2++) idx[j2] = idx[j2 - 1] + 1;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        ntests += __shfl_xor(ntests, o);
        nsub += __shfl_xor(nsub, o);
        nrem += __shfl_xor(nrem, o);
        nbad += __shfl_xor(nbad, o);
    }
    if ((tid & 63) == 0)
    {
        atomicAdd(&s_cnt[0], ntests);
        atomicAdd(&s_cnt[1], nsub);
        atomicAdd(&s_cnt[2], nrem);
        atomicAdd(&s_cnt[3], nbad);
    }
    __syncthreads();
    if (tid == 0)
    {
        unsigned long long *sl = p.slots + (size_t)(blockIdx.x & (kCounterSlots - 1)) * 4;
        if (s_cnt[0]) atomicAdd(&sl[0], s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&sl[1], s_cnt[1]);
        if (s_cnt[2]) atomicAdd(&sl[2], s_cnt[2]);
        if (s_cnt[3]) atomicAdd(&sl[3], s_cnt[3]);
    }
}

// exact-path evaluation of the queued tests: one lane per entry, operands straight from HBM
template <int L, int MODE, bool HET>
__global__ void __launch_bounds__(256) recheck_kernel(SweepParams p, unsigned long long count)
{
    const unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const RecheckEntry en = p.queue[e];
    const int X = en.x, k2 = en.k2;
    const int o0 = p.off[X];
    const int d = p.off[X + 1] - o0;
    const int n = p.n;
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
    cx.prepare(m2, m1x);
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
        float me = s / (float)((L + 2) * (L + 1) / 2);
        lth = (float)((double)p.th / sqrt((double)me - (double)L - 3.0));
    }
    if (z_below<L == 1>(rho, lth))
    {
        if constexpr (MODE == 0)
        {
            unsigned long long old = atomicMin(&p.best[o0 + k2], en.rank);
            if (old == kNone) atomicAdd(&p.slots[(size_t)(blockIdx.x & (kCounterSlots - 1)) * 4 + 2], 1ull);
        }
        else
        {
            unsigned long long old = atomicAnd(&p.adj[(size_t)X * p.words + (Y >> 6)], ~(1ull << (Y & 63)));
            atomicAnd(&p.adj[(size_t)Y * p.words + (X >> 6)], ~(1ull << (X & 63)));
            if ((old >> (Y & 63)) & 1ull) atomicAdd(&p.slots[(size_t)(blockIdx.x & (kCounterSlots - 1)) * 4 + 2], 1ull);
        }
    }
}

// MODE 0 only: turn the per-slot winning ranks into sparse sepset records, recompute
// the winner's Fisher z on the exact path (so pMax does not depend on which lane won)
// and clear the edge in both directions.
struct FinalizeParams
{
    const float *C;
    int n;
    const int *off;
    const int *nbr;
    const unsigned long long *best;
    unsigned long long *adj;
    int words;
    const unsigned long long *binom;
    int *rec_x, *rec_y, *rec_l, *rec_s;
    float *rec_z;
    LevelCounters *cnt;
};

template <int L>
__global__ void finalize_kernel(FinalizeParams p)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= p.n) return;
    const int o0 = p.off[row];
    const int d = p.off[row + 1] - o0;
    const int n = p.n;
    for (int k0 = 0; k0 < d; k0 += 64)
    {
        const int k = k0 + lane;
        const unsigned long long r = (k < d) ? p.best[o0 + k] : kNone;
        // one atomic per wave: the leader reserves slots for every lane that has a record
        const unsigned long long has = __ballot(r != kNone);
        if (has == 0ull) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&p.cnt->nrec, (unsigned long long)__popcll(has));
        base = __shfl(base, 0);
        if (r == kNone) continue;
        const unsigned long long slot = base + (unsigned long long)__popcll(has & ((1ull << lane) - 1ull));
        int idx[L];
        {
            unsigned long long rem = r;
            int c = 0;
#pragma unroll
            for (int i = 0; i < L; i++)
            {
                while (true)
                {
                    unsigned long long b = p.binom[(size_t)(d - 1 - c) * (L + 1) + (L - 1 - i)];
                    if (rem < b) break;
                    rem -= b;
                    c++;
                }
                idx[i] = c;
                c++;
            }
        }
        const int X = row, Y = p.nbr[o0 + k];
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
                    m2[a * L + b] = (a == b) ? 1.0f
                                             : (a < b ? p.C[(size_t)S[a] * n + S[b]] : p.C[(size_t)S[b] * n + S[a]]);
        }
        SubsetExact<L> cx;
        cx.prepare(m2, m1x);
        const float rho = cx.rho(p.C[(size_t)X * n + Y], m1y);
        float z;
        (void)z_below<L == 1>(rho, 0.0f, &z);
        p.rec_x[slot] = X;
        p.rec_y[slot] = Y;
        p.rec_l[slot] = L;
        p.rec_z[slot] = z;
#pragma unroll
        for (int a = 0; a < kML; a++) p.rec_s[slot * kML + a] = (a < L) ? S[a < L ? a : 0] : -1;
        atomicAnd(&p.adj[(size_t)X * p.words + (Y >> 6)], ~(1ull << (Y & 63)));
        atomicAnd(&p.adj[(size_t)Y * p.words + (X >> 6)], ~(1ull << (X & 63)));
    }
}

// ---------------------------------------------------------------------------
// launch tables
// ---------------------------------------------------------------------------

template <int L, int MODE, bool HET>
static hipError_t launch_sweep_L(const SweepParams &p, int cls, long long nitems, hipStream_t st)
{
    if (nitems <= 0) return hipSuccess;
    const bool staged = (cls < kNumClasses - 1);
    if (staged)
    {
        size_t lds = lds_layout(kClassCap[cls], HET).total;
        auto kfn = sweep_kernel<L, MODE, HET, true>;
        if (lds > 64 * 1024)
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kfn, dim3((unsigned)nitems), dim3(kThreads), lds, st, p);
    }
    else
    {
        hipLaunchKernelGGL((sweep_kernel<L, MODE, HET, false>), dim3((unsigned)nitems), dim3(kThreads), 16, st, p);
    }
    return hipGetLastError();
}

template <int MODE, bool HET>
static hipError_t launch_sweep(int L, const SweepParams &p, int cls, long long nitems, hipStream_t st)
{
    switch (L)
    {
#define CUSK_CASE(LL) \
    case LL: return launch_sweep_L<LL, MODE, HET>(p, cls, nitems, st);
        CUSK_CASE(1)
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

template <int L, int MODE, bool HET, bool VALIDATE>
static hipError_t launch_fast_L(const SweepParams &p, int cls, long long nitems, hipStream_t st)
{
    if (nitems <= 0) return hipSuccess;
    const bool staged = (cls < kNumClasses - 1);
    if (staged)
    {
        size_t lds = lds_layout(kClassCap[cls], HET).total;
        auto kfn = sweep_fast_kernel<L, MODE, HET, true, VALIDATE>;
        if (lds > 64 * 1024)
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kfn, dim3((unsigned)nitems), dim3(kThreads), lds, st, p);
    }
    else
    {
        hipLaunchKernelGGL((sweep_fast_kernel<L, MODE, HET, false, VALIDATE>), dim3((unsigned)nitems), dim3(kThreads),
                           16, st, p);
    }
    return hipGetLastError();
}

template <int MODE, bool HET>
static hipError_t launch_fast(int L, bool validate, const SweepParams &p, int cls, long long nitems, hipStream_t st)
{
    switch (L)
    {
#define CUSK_CASE(LL)                                                                     \
    case LL:                                                                              \
        return validate ? launch_fast_L<LL, MODE, HET, true>(p, cls, nitems, st)          \
                        : launch_fast_L<LL, MODE, HET, false>(p, cls, nitems, st);
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

template <int MODE, bool HET>
static hipError_t launch_recheck(int L, const SweepParams &p, unsigned long long count, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    dim3 grid((unsigned)((count + 255) / 256)), block(256);
    switch (L)
    {
#define CUSK_CASE(LL)                                                                             \
    case LL:                                                                                      \
        hipLaunchKernelGGL((recheck_kernel<LL, MODE, HET>), grid, block, 0, st, p, count);        \
        break;
        CUSK_CASE(1)
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
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static hipError_t launch_finalize(int L, const FinalizeParams &p, hipStream_t st)
{
    dim3 grid((p.n + 3) / 4), block(256);
    switch (L)
    {
#define CUSK_CASE(LL)                                                  \
    case LL:                                                           \
        hipLaunchKernelGGL(finalize_kernel<LL>, grid, block, 0, st, p); \
        break;
        CUSK_CASE(1)
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
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

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
// the float running sum of (int)ess over (l+2)(l+1)/2 pairs, as mean_ess forms it.
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
    float me = s / (float)pairs;
    return (float)((double)th / std::sqrt((double)me - (double)l - 3.0));
}

// ---------------------------------------------------------------------------
// the level loop (cuPC-S.cu:99-415 / hetcor-cuPC-S.cu:115-332)
// ---------------------------------------------------------------------------

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
};

static int run_levels(cusk_engine *e, const RunArgs &a, cusk_stats *st)
{
    const int n = a.n;
    if (n <= 0 || a.C == nullptr || a.Th == nullptr) return fail(e, CUSK_ERR_ARG, "bad arguments");
    CUSK_HIP(e, hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const int words = (n + 63) / 64;
    e->n = n;
    e->words = words;
    e->mode = a.mode;
    e->have_result = false;
    e->nrec = 0;
    e->exact_fallbacks = 0;
    cusk_stats local;
    std::memset(&local, 0, sizeof(local));
    const bool het = (a.mode == 1 && a.Ness != nullptr);

    CUSK_HIP(e, e->adj.ensure(sizeof(unsigned long long) * (size_t)n * words));
    CUSK_HIP(e, e->adj0.ensure(sizeof(unsigned long long) * (size_t)n * words));
    CUSK_HIP(e, e->deg.ensure(sizeof(int) * (size_t)n));
    CUSK_HIP(e, e->off.ensure(sizeof(int) * ((size_t)n + 1)));
    CUSK_HIP(e, e->rowinfo.ensure(sizeof(RowInfo) * (size_t)n));
    CUSK_HIP(e, e->counters.ensure(sizeof(LevelCounters)));
    LevelCounters *dcnt = e->counters.as<LevelCounters>();
    CUSK_HIP(e, e->slots.ensure(sizeof(unsigned long long) * kCounterSlots * 4));
    if (a.mode == 1)
    {
        CUSK_HIP(e, e->ti.ensure(sizeof(int) * (size_t)n));
        if (a.time_index)
            CUSK_HIP(e, hipMemcpyAsync(e->ti.p, a.time_index, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
        else
            CUSK_HIP(e, hipMemsetAsync(e->ti.p, 0, sizeof(int) * (size_t)n, s));
    }

    CUSK_HIP(e, e->symflag.ensure(sizeof(int)));
    CUSK_HIP(e, hipMemsetAsync(e->symflag.p, 0, sizeof(int), s));
    CUSK_HIP(e, hipEventRecord(e->ev[0], s));
    // ---- level 0 ----
    {
        CUSK_HIP(e, hipEventRecord(e->ev[1], s));
        hipLaunchKernelGGL(init_bits_kernel, dim3(n), dim3(64), 0, s, e->adj.as<unsigned long long>(), a.Ginit, n, words);
        const int tiles = words;
        const long long ntile = (long long)tiles * (tiles + 1) / 2;
        if (het)
            hipLaunchKernelGGL(level0_kernel<true>, dim3((unsigned)ntile), dim3(256), 0, s, a.C, a.Ness,
                               e->adj.as<unsigned long long>(), n, words, a.Th[0], tiles, e->symflag.as<int>());
        else
        {
            float th0 = a.Th[0];
            if (a.mode == 1) th0 = (float)((double)a.Th[0] / std::sqrt((double)a.ess_uniform - 3.0));
            hipLaunchKernelGGL(level0_kernel<false>, dim3((unsigned)ntile), dim3(256), 0, s, a.C, nullptr,
                               e->adj.as<unsigned long long>(), n, words, th0, tiles, e->symflag.as<int>());
        }
        CUSK_HIP(e, hipGetLastError());
        if (a.mode == 0)
            CUSK_HIP(e, hipMemcpyAsync(e->adj0.p, e->adj.p, sizeof(unsigned long long) * (size_t)n * words,
                                       hipMemcpyDeviceToDevice, s));
        CUSK_HIP(e, hipEventRecord(e->ev[2], s));
        local.max_degree[0] = n - 1;
        local.edges[0] = (long long)n * (n - 1);
        local.tests[0] = (long long)n * (n - 1) / 2;
        local.levels_run = 1;
    }
    float ms = 0.0f;
    int l = 0;
    bool finished = false;
    bool symmetric = false;
    for (l = 0; l <= kML && !finished && l <= a.maxlevel; l++)
    {
        if (l == 0)
        {
            CUSK_HIP(e, hipEventSynchronize(e->ev[2]));
            CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev[1], e->ev[2]));
            local.kernel_ms[0] = local.level_ms[0] = ms;
            continue;
        }
        CUSK_HIP(e, hipEventRecord(e->ev[1], s));
        CUSK_HIP(e, hipMemsetAsync(dcnt, 0, sizeof(LevelCounters), s));
        CUSK_HIP(e, hipMemsetAsync(e->slots.p, 0, sizeof(unsigned long long) * kCounterSlots * 4, s));
        hipLaunchKernelGGL(degree_kernel, dim3((n + 3) / 4), dim3(256), 0, s, e->adj.as<unsigned long long>(),
                           e->deg.as<int>(), n, words, dcnt);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, e->deg.as<int>(), e->off.as<int>(), n, dcnt);
        CUSK_HIP(e, hipMemcpyAsync(e->hcnt, dcnt, sizeof(LevelCounters), hipMemcpyDeviceToHost, s));
        if (l == 1) CUSK_HIP(e, hipMemcpyAsync(e->hflag, e->symflag.p, sizeof(int), hipMemcpyDeviceToHost, s));
        CUSK_HIP(e, hipStreamSynchronize(s));
        if (l == 1) symmetric = (*e->hflag == 0);
        const int maxdeg = e->hcnt->maxdeg;
        const long long total = e->hcnt->total_edges;
        local.max_degree[l] = maxdeg;
        local.edges[l] = total;
        if (maxdeg - 1 < l)
        {  // cuPC-S.cu:154-159
            l = l - 1;
            finished = true;
            break;
        }
        // binomial table for this level, a <= maxdeg
        {
            std::vector<unsigned long long> tab((size_t)(maxdeg + 1) * (l + 1));
            for (int aa = 0; aa <= maxdeg; aa++)
                for (int b = 0; b <= l; b++) tab[(size_t)aa * (l + 1) + b] = binom_sat(aa, b);
            CUSK_HIP(e, e->binom.ensure(tab.size() * sizeof(unsigned long long)));
            CUSK_HIP(e, hipMemcpyAsync(e->binom.p, tab.data(), tab.size() * sizeof(unsigned long long),
                                       hipMemcpyHostToDevice, s));
            CUSK_HIP(e, hipStreamSynchronize(s));  // tab is a stack-lifetime vector
        }
        CUSK_HIP(e, e->nbr.ensure(sizeof(int) * (size_t)std::max<long long>(total, 1)));
        if (a.mode == 0)
        {
            CUSK_HIP(e, e->best.ensure(sizeof(unsigned long long) * (size_t)std::max<long long>(total, 1)));
            if (l == 1)
            {
                e->rec_cap = total;
                CUSK_HIP(e, e->rec_x.ensure(sizeof(int) * (size_t)std::max<long long>(total, 1)));
                CUSK_HIP(e, e->rec_y.ensure(sizeof(int) * (size_t)std::max<long long>(total, 1)));
                CUSK_HIP(e, e->rec_l.ensure(sizeof(int) * (size_t)std::max<long long>(total, 1)));
                CUSK_HIP(e, e->rec_z.ensure(sizeof(float) * (size_t)std::max<long long>(total, 1)));
                CUSK_HIP(e, e->rec_s.ensure(sizeof(int) * kML * (size_t)std::max<long long>(total, 1)));
            }
            if (total > 0)
                hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                                   e->best.as<unsigned long long>(), (size_t)total, kNone);
        }
        // which classes can be staged in LDS in this mode
        int staged_classes = 0;
        while (staged_classes < kNumClasses - 1 && lds_layout(kClassCap[staged_classes], het).total <= kLdsLimit)
            staged_classes++;
        const unsigned long long chunk = (unsigned long long)std::max<long long>(e->opt_chunk, 256);
        // level 1 on a symmetric matrix with a single threshold: pair kernel (one gather feeds two tests)
        const size_t pair_lds = (size_t)maxdeg * 20 + 16;
        const bool use_pair = (l == 1) && !het && symmetric && (e->opt_pair != 0) && pair_lds <= 64 * 1024;
        hipLaunchKernelGGL(fill_nbr_kernel, dim3((n + 3) / 4), dim3(256), 0, s, e->adj.as<unsigned long long>(),
                           e->off.as<int>(), e->nbr.as<int>(), n, words, l, e->binom.as<unsigned long long>(), chunk,
                           staged_classes, use_pair ? 1 : 0, e->rowinfo.as<RowInfo>(), dcnt);
        hipLaunchKernelGGL(item_scan_kernel, dim3(1), dim3(1024), 0, s, e->rowinfo.as<RowInfo>(), n, dcnt);
        CUSK_HIP(e, hipMemcpyAsync(e->hcnt, dcnt, sizeof(LevelCounters), hipMemcpyDeviceToHost, s));
        CUSK_HIP(e, hipStreamSynchronize(s));
        if (e->hcnt->overflow)
            return fail(e, CUSK_ERR_OVERFLOW,
                        "C(degree, level) exceeds 2^62 at level " + std::to_string(l) + " (max degree " +
                            std::to_string(maxdeg) + ")");
        long long nitems[kNumClasses];
        for (int c = 0; c < kNumClasses; c++)
        {
            nitems[c] = e->hcnt->class_items[c];
            CUSK_HIP(e, e->items[c].ensure(sizeof(int2) * (size_t)std::max<long long>(nitems[c], 1)));
        }
        hipLaunchKernelGGL(fill_items_kernel, dim3(n), dim3(64), 0, s, e->rowinfo.as<RowInfo>(), n,
                           e->items[0].as<int2>(), e->items[1].as<int2>(), e->items[2].as<int2>(),
                           e->items[3].as<int2>(), e->items[4].as<int2>());
        CUSK_HIP(e, hipGetLastError());

        SweepParams sp;
        sp.C = a.C;
        sp.Ness = a.Ness;
        sp.n = n;
        sp.off = e->off.as<int>();
        sp.nbr = e->nbr.as<int>();
        sp.best = e->best.as<unsigned long long>();
        sp.adj = e->adj.as<unsigned long long>();
        sp.words = words;
        sp.binom = e->binom.as<unsigned long long>();
        sp.time_index = e->ti.as<int>();
        sp.chunk = chunk;
        sp.cnt = dcnt;
        sp.slots = e->slots.as<unsigned long long>();
        if (a.mode == 0)
            sp.th = a.Th[l];
        else if (het)
            sp.th = a.Th[0];
        else
            sp.th = uniform_ess_threshold(a.Th[0], a.ess_uniform, l);

        const bool use_fast = (e->opt_fast != 0) && (l >= 2);
        if (use_fast)
        {
            const double tq = std::tanh((double)sp.th);
            sp.t2 = (float)(tq * tq);
            CUSK_HIP(e, e->queue.ensure(sizeof(RecheckEntry) * (size_t)e->opt_queue_cap));
            sp.queue = e->queue.as<RecheckEntry>();
            sp.qcap = (unsigned long long)e->opt_queue_cap;
        }
        auto run_exact_sweeps = [&]() -> int {
            for (int c = 0; c < kNumClasses; c++)
            {
                if (nitems[c] <= 0) continue;
                sp.items = e->items[c].as<int2>();
                sp.cap = kClassCap[c];
                hipError_t le;
                if (a.mode == 0)
                    le = launch_sweep<0, false>(l, sp, c, nitems[c], s);
                else if (het)
                    le = launch_sweep<1, true>(l, sp, c, nitems[c], s);
                else
                    le = launch_sweep<1, false>(l, sp, c, nitems[c], s);
                CUSK_HIP(e, le);
            }
            return CUSK_OK;
        };
        CUSK_HIP(e, hipEventRecord(e->ev[2], s));
        if (use_pair)
        {
            for (int c = 0; c < kNumClasses; c++)
            {
                if (nitems[c] <= 0) continue;
                sp.items = e->items[c].as<int2>();
                if (a.mode == 0)
                    hipLaunchKernelGGL(level1_pair_kernel<0>, dim3((unsigned)nitems[c]), dim3(kThreads), pair_lds, s, sp);
                else
                    hipLaunchKernelGGL(level1_pair_kernel<1>, dim3((unsigned)nitems[c]), dim3(kThreads), pair_lds, s, sp);
                CUSK_HIP(e, hipGetLastError());
            }
        }
        else if (!use_fast)
        {
            int rc = run_exact_sweeps();
            if (rc != CUSK_OK) return rc;
        }
        else
        {
            for (int c = 0; c < kNumClasses; c++)
            {
                if (nitems[c] <= 0) continue;
                sp.items = e->items[c].as<int2>();
                sp.cap = kClassCap[c];
                hipError_t le;
                if (a.mode == 0)
                    le = launch_fast<0, false>(l, e->opt_validate != 0, sp, c, nitems[c], s);
                else if (het)
                    le = launch_fast<1, true>(l, e->opt_validate != 0, sp, c, nitems[c], s);
                else
                    le = launch_fast<1, false>(l, e->opt_validate != 0, sp, c, nitems[c], s);
                CUSK_HIP(e, le);
            }
            // how many tests need the exact path?
            CUSK_HIP(e, hipMemcpyAsync(e->hcnt, dcnt, sizeof(LevelCounters), hipMemcpyDeviceToHost, s));
            CUSK_HIP(e, hipStreamSynchronize(s));
            const unsigned long long qn = e->hcnt->qcount;
            local.rechecks[l] = (long long)qn;
            if (qn > sp.qcap)
            {
                // queue overflow (pathologically ill-conditioned input): redo the level on the exact path.
                // Everything the fast pass already recorded is a certified verdict, so it stays valid.
                e->exact_fallbacks++;
                int rc = run_exact_sweeps();
                if (rc != CUSK_OK) return rc;
            }
            else if (qn > 0)
            {
                hipError_t le;
                if (a.mode == 0)
                    le = launch_recheck<0, false>(l, sp, qn, s);
                else if (het)
                    le = launch_recheck<1, true>(l, sp, qn, s);
                else
                    le = launch_recheck<1, false>(l, sp, qn, s);
                CUSK_HIP(e, le);
            }
        }
        CUSK_HIP(e, hipEventRecord(e->ev[3], s));
        if (a.mode == 0)
        {
            FinalizeParams fp;
            fp.C = a.C;
            fp.n = n;
            fp.off = e->off.as<int>();
            fp.nbr = e->nbr.as<int>();
            fp.best = e->best.as<unsigned long long>();
            fp.adj = e->adj.as<unsigned long long>();
            fp.words = words;
            fp.binom = e->binom.as<unsigned long long>();
            fp.rec_x = e->rec_x.as<int>();
            fp.rec_y = e->rec_y.as<int>();
            fp.rec_l = e->rec_l.as<int>();
            fp.rec_s = e->rec_s.as<int>();
            fp.rec_z = e->rec_z.as<float>();
            fp.cnt = dcnt;
            // nrec continues across levels: restore the running count before the launch
            CUSK_HIP(e, hipMemcpyAsync(&dcnt->nrec, &e->nrec, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
            CUSK_HIP(e, launch_finalize(l, fp, s));
        }
        CUSK_HIP(e, hipMemcpyAsync(e->hcnt, dcnt, sizeof(LevelCounters), hipMemcpyDeviceToHost, s));
        CUSK_HIP(e, hipMemcpyAsync(e->hslots, e->slots.p, sizeof(unsigned long long) * kCounterSlots * 4,
                                   hipMemcpyDeviceToHost, s));
        CUSK_HIP(e, hipEventRecord(e->ev[4], s));
        CUSK_HIP(e, hipStreamSynchronize(s));
        CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev[1], e->ev[4]));
        local.level_ms[l] = ms;
        for (int k = 0; k < kCounterSlots; k++)
        {
            local.tests[l] += (long long)e->hslots[k * 4 + 0];
            local.subsets[l] += (long long)e->hslots[k * 4 + 1];
            local.removed[l] += (long long)e->hslots[k * 4 + 2];
            local.violations += (long long)e->hslots[k * 4 + 3];
        }
        if (a.mode == 0) e->nrec = (long long)e->hcnt->nrec;
        CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev[2], e->ev[3]));
        local.kernel_ms[l] = ms;
        local.levels_run++;
    }
    local.level = l;
    local.exact_fallbacks = e->exact_fallbacks;
    CUSK_HIP(e, hipEventRecord(e->ev[5], s));
    CUSK_HIP(e, hipEventSynchronize(e->ev[5]));
    CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev[0], e->ev[5]));
    local.total_ms = ms;
    e->have_result = true;
    if (st) *st = local;
    return CUSK_OK;
}

// ---------------------------------------------------------------------------
// result expansion
// ---------------------------------------------------------------------------

__global__ void expand_adj_kernel(const unsigned long long *__restrict__ adj, int *G, int n, int words)
{
    const int row = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n) return;
    unsigned long long w = adj[(size_t)row * words + (col >> 6)];
    G[(size_t)row * n + col] = (int)((w >> (col & 63)) & 1ull);
}

// pMax before the sparse records are applied (cuPC-S.cu:424-442 semantics): -100000 on
// surviving edges, 1 on the diagonal, level-0 z where level 0 removed the pair, else 0.
__global__ void expand_pmax_kernel(const unsigned long long *__restrict__ adj,
                                   const unsigned long long *__restrict__ adj0, const float *__restrict__ C,
                                   float *pmax, int n, int words)
{
    const int row = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n) return;
    const bool live = (adj[(size_t)row * words + (col >> 6)] >> (col & 63)) & 1ull;
    const bool live0 = (adj0[(size_t)row * words + (col >> 6)] >> (col & 63)) & 1ull;
    float v;
    if (row == col)
        v = 1.0f;
    else if (live)
        v = -100000.0f;
    else if (!live0)
    {
        const int i = min(row, col), j = max(row, col);
        v = fisher_z_ratio(C[(size_t)i * n + j]);
    }
    else
        v = 0.0f;
    pmax[(size_t)row * n + col] = v;
}

__global__ void scatter_pmax_kernel(const int *x, const int *y, const float *z, long long nrec, float *pmax, int n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    // z >= 0 or NaN never occurs for a stored record (NaN fails z < th); order as ints
    int zi = __float_as_int(z[i]);
    atomicMax(reinterpret_cast<int *>(&pmax[(size_t)x[i] * n + y[i]]), zi);
    atomicMax(reinterpret_cast<int *>(&pmax[(size_t)y[i] * n + x[i]]), zi);
}

__global__ void scatter_sepset_kernel(const int *x, const int *y, const int *S, long long nrec, int *sep, int n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec * kML) return;
    long long r = i / kML;
    int a = (int)(i - r * kML);
    sep[((size_t)x[r] * n + y[r]) * kML + a] = S[i];
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
    if (stream)
    {
        e->stream = reinterpret_cast<hipStream_t>(stream);
        e->own_stream = false;
    }
    else
    {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
        {
            delete e;
            return CUSK_ERR_HIP;
        }
        e->own_stream = true;
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&e->hcnt), sizeof(LevelCounters)) != hipSuccess)
    {
        delete e;
        return CUSK_ERR_HIP;
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&e->hslots), sizeof(unsigned long long) * kCounterSlots * 4) != hipSuccess)
    {
        delete e;
        return CUSK_ERR_HIP;
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&e->hflag), sizeof(int)) != hipSuccess)
    {
        delete e;
        return CUSK_ERR_HIP;
    }
    for (auto &ev : e->ev)
        if (hipEventCreate(&ev) != hipSuccess)
        {
            delete e;
            return CUSK_ERR_HIP;
        }
    *out = e;
    return CUSK_OK;
}

extern "C" void cusk_engine_destroy(cusk_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (DevBuf *b : {&e->adj, &e->adj0, &e->deg, &e->off, &e->nbr, &e->best, &e->rowinfo, &e->binom, &e->counters,
                      &e->ti, &e->rec_x, &e->rec_y, &e->rec_l, &e->rec_z, &e->rec_s, &e->bed_dev, &e->phen_dev,
                      &e->mean_dev, &e->std_dev, &e->planes, &e->mxp_dev, &e->queue, &e->symflag, &e->slots})
        b->release();
    for (auto &b : e->items) b.release();
    if (e->hcnt) (void)hipHostFree(e->hcnt);
    if (e->hflag) (void)hipHostFree(e->hflag);
    if (e->hslots) (void)hipHostFree(e->hslots);
    for (auto &ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
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
    else if (k == "pair")
        e->opt_pair = (int)value;
    else if (k == "queue_capacity" && value > 0)
        e->opt_queue_cap = value;
    else if (k == "chunk" && value >= 256)
        e->opt_chunk = value;
    else
        return fail(e, CUSK_ERR_ARG, "unknown option " + k);
    return CUSK_OK;
}

extern "C" const char *cusk_last_error(const cusk_engine *e) { return e ? e->err.c_str() : "no engine"; }
extern "C" void *cusk_engine_stream(const cusk_engine *e) { return e ? (void *)e->stream : nullptr; }

extern "C" int cusk_run_skeleton(cusk_engine *e, const float *C_dev, int n, const float *Th, int maxlevel,
                                 cusk_stats *stats)
{
    if (!e) return CUSK_ERR_ARG;
    RunArgs a{};
    a.mode = 0;
    a.C = C_dev;
    a.Th = Th;
    a.n = n;
    a.maxlevel = maxlevel;
    return run_levels(e, a, stats);
}

extern "C" int cusk_run_hetcor(cusk_engine *e, const float *C_dev, const float *N_dev, float ess_uniform,
                               const int *G_init_dev, int n, float th, int maxlevel, const int *time_index,
                               cusk_stats *stats)
{
    if (!e) return CUSK_ERR_ARG;
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
    hipLaunchKernelGGL(expand_adj_kernel, dim3((e->n + 255) / 256, e->n), dim3(256), 0, e->stream,
                       e->adj.as<unsigned long long>(), G_dev, e->n, e->words);
    CUSK_HIP(e, hipGetLastError());
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

extern "C" int cusk_result_pmax(cusk_engine *e, const float *C_dev, float *pMax_host)
{
    if (!e || !e->have_result || e->mode != 0) return fail(e, CUSK_ERR_STATE, "no Skeleton result");
    CUSK_HIP(e, hipSetDevice(e->device));
    const size_t bytes = sizeof(float) * (size_t)e->n * e->n;
    float *tmp = nullptr;
    CUSK_HIP(e, hipMalloc(reinterpret_cast<void **>(&tmp), bytes));
    hipLaunchKernelGGL(expand_pmax_kernel, dim3((e->n + 255) / 256, e->n), dim3(256), 0, e->stream,
                       e->adj.as<unsigned long long>(), e->adj0.as<unsigned long long>(), C_dev, tmp, e->n, e->words);
    if (e->nrec > 0)
        hipLaunchKernelGGL(scatter_pmax_kernel, dim3((unsigned)((e->nrec + 255) / 256)), dim3(256), 0, e->stream,
                           e->rec_x.as<int>(), e->rec_y.as<int>(), e->rec_z.as<float>(), e->nrec, tmp, e->n);
    hipError_t st = hipGetLastError();
    if (st == hipSuccess) st = hipMemcpyAsync(pMax_host, tmp, bytes, hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    (void)hipFree(tmp);
    if (st != hipSuccess) return fail(e, CUSK_ERR_HIP, hipGetErrorString(st));
    return CUSK_OK;
}

extern "C" int cusk_result_sepset_dense(cusk_engine *e, int *SepSet_host)
{
    if (!e || !e->have_result || e->mode != 0) return fail(e, CUSK_ERR_STATE, "no Skeleton result");
    CUSK_HIP(e, hipSetDevice(e->device));
    const size_t count = (size_t)e->n * e->n * kML;
    // the dense n*n*14 array is only needed for the reference's ABI; build it on the host
    std::fill(SepSet_host, SepSet_host + count, -1);
    if (e->nrec > 0)
    {
        std::vector<int> x(e->nrec), y(e->nrec), S((size_t)e->nrec * kML);
        CUSK_HIP(e, hipMemcpy(x.data(), e->rec_x.p, sizeof(int) * e->nrec, hipMemcpyDeviceToHost));
        CUSK_HIP(e, hipMemcpy(y.data(), e->rec_y.p, sizeof(int) * e->nrec, hipMemcpyDeviceToHost));
        CUSK_HIP(e, hipMemcpy(S.data(), e->rec_s.p, sizeof(int) * kML * e->nrec, hipMemcpyDeviceToHost));
        for (long long r = 0; r < e->nrec; r++)
            std::memcpy(SepSet_host + ((size_t)x[r] * e->n + y[r]) * kML, S.data() + (size_t)r * kML, sizeof(int) * kML);
    }
    return CUSK_OK;
}

extern "C" long long cusk_result_sepsets(cusk_engine *e, int *x, int *y, int *level, float *z, int *S)
{
    if (!e || !e->have_result || e->mode != 0) return -1;
    if (hipSetDevice(e->device) != hipSuccess) return -1;
    const long long c = e->nrec;
    if (c > 0)
    {
        if (x && hipMemcpy(x, e->rec_x.p, sizeof(int) * c, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (y && hipMemcpy(y, e->rec_y.p, sizeof(int) * c, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (level && hipMemcpy(level, e->rec_l.p, sizeof(int) * c, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (z && hipMemcpy(z, e->rec_z.p, sizeof(float) * c, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (S && hipMemcpy(S, e->rec_s.p, sizeof(int) * kML * c, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    return c;
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
