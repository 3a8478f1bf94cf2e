// sweep_stage.h -- per-workgroup view of one row X of the level sweep.
//
// STAGED: the neighbour list, the row's selection state and the (d+1)^2 sub-matrix
// C[adj(X)+X]^2 (plus the matching effective-sample-size block when HET) are copied into
// LDS once per workgroup; every operand of every test then comes from LDS.  Index d of the
// sub-matrix is X itself.  Rows of the sub-matrix keep the exact [row][col] orientation of C,
// so asymmetric inputs read the same elements as the reference does.
// !STAGED (hubs whose sub-matrix exceeds 160 KB): the same accessors read through L2.
#pragma once
#include "sweep_common.h"

namespace cusk {

template <int MODE, bool HET, bool STAGED>
struct RowView
{
    const SweepParams &p;
    int X, o0, d, n, ld;
    const int *g_nbr;
    int *s_nbr;
    unsigned long long *s_best;
    int *s_ti;
    float *s_sub;
    float *s_ess;

    __device__ __forceinline__ RowView(const SweepParams &pp, int x, unsigned char *smem) : p(pp)
    {
        X = x;
        o0 = p.off[X];
        d = p.off[X + 1] - o0;
        n = p.n;
        ld = (d + 1) | 1;
        g_nbr = p.nbr + o0;
        const LdsLayout lay = lds_layout(STAGED ? p.cap : 0, HET);
        s_nbr = reinterpret_cast<int *>(smem + lay.nbr);
        s_best = reinterpret_cast<unsigned long long *>(smem + lay.best);
        s_ti = reinterpret_cast<int *>(smem + lay.ti);
        s_sub = reinterpret_cast<float *>(smem + lay.sub);
        s_ess = reinterpret_cast<float *>(smem + lay.ess);
    }

    // all threads of the workgroup; ends with a barrier
    __device__ __forceinline__ void stage()
    {
        if constexpr (STAGED)
        {
            const int tid = threadIdx.x;
            for (int k = tid; k <= d; k += kThreads)
            {
                const int v = (k < d) ? g_nbr[k] : X;
                s_nbr[k] = v;
                if constexpr (MODE == 1) s_ti[k] = p.time_index[v];
            }
            for (int k = tid; k < d; k += kThreads) s_best[k] = global_state(k);
            __syncthreads();
            const int dd = d + 1;
            for (int e = tid; e < dd * dd; e += kThreads)
            {
                const int i = e / dd, j = e - i * dd;
                const size_t g = (size_t)s_nbr[i] * n + s_nbr[j];
                s_sub[i * ld + j] = p.C[g];
                if constexpr (HET) s_ess[i * ld + j] = p.Ness[g];
            }
        }
        __syncthreads();
    }

    // selection state of slot k in global memory: MODE 0 lowest passing rank (kNone = none yet),
    // MODE 1 kNone while the edge is alive, 0 once removed
    __device__ __forceinline__ unsigned long long global_state(int k) const
    {
        if constexpr (MODE == 0)
            return p.best[o0 + k];
        else
        {
            const int y = g_nbr[k];
            const unsigned long long wv = p.adj[(size_t)X * p.words + (y >> 6)];
            return ((wv >> (y & 63)) & 1ull) ? kNone : 0ull;
        }
    }
    __device__ __forceinline__ unsigned long long state(int k) const
    {
        if constexpr (STAGED)
            return s_best[k];
        else
            return global_state(k);
    }
    __device__ __forceinline__ bool live(int k, unsigned long long rank) const
    {
        const unsigned long long b = state(k);
        return (MODE == 0) ? (b >= rank) : (b == kNone);
    }
    __device__ __forceinline__ int var_of(int i) const
    {
        if constexpr (STAGED)
            return s_nbr[i];
        else
            return (i < d) ? g_nbr[i] : X;
    }
    __device__ __forceinline__ float cval(int i, int j) const
    {
        if constexpr (STAGED)
            return s_sub[i * ld + j];
        else
            return p.C[(size_t)var_of(i) * n + var_of(j)];
    }
    __device__ __forceinline__ float eval(int i, int j) const
    {
        if constexpr (STAGED)
            return s_ess[i * ld + j];
        else
            return p.Ness[(size_t)var_of(i) * n + var_of(j)];
    }
    __device__ __forceinline__ int tix(int i) const
    {
        if constexpr (STAGED)
            return s_ti[i];
        else
            return p.time_index[var_of(i)];
    }

    // record "S (combination `rank`) separates X from its k2-th neighbour"; returns true when
    // this call is the first to decide the pair
    __device__ __forceinline__ bool separate(int k2, unsigned long long rank) const
    {
        if constexpr (MODE == 0)
        {
            const unsigned long long old = atomicMin(&p.best[o0 + k2], rank);
            if constexpr (STAGED) atomicMin(&s_best[k2], rank);
            return old == kNone;
        }
        else
        {
            const bool first = clear_edge(p.adj, p.deg, p.words, X, var_of(k2));
            if constexpr (STAGED) s_best[k2] = 0ull;
            return first;
        }
    }

    // exact mean_ess threshold of hetcor-cuPC-S.cu:471,612,3068-3088 for the test (X, Y=k2 | idx)
    template <int L>
    __device__ __forceinline__ float ess_threshold_exact(int k2, const int *idx) const
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
        const float me = s / (float)((L + 2) * (L + 1) / 2);
        return (float)((double)p.th / sqrt((double)me - (double)L - 3.0));
    }
};

// lane-contiguous range of combination ranks of a work item
struct RankRange
{
    unsigned long long lo, hi;
};
__device__ __forceinline__ RankRange lane_ranks(unsigned long long ncomb, int chunk_index, unsigned long long chunk)
{
    const unsigned long long r0 = (unsigned long long)chunk_index * chunk;
    const unsigned long long cnt = min(chunk, ncomb - r0);
    const unsigned long long q = (cnt + kThreads - 1) / kThreads;
    RankRange r;
    r.lo = r0 + (unsigned long long)threadIdx.x * q;
    r.hi = min(r0 + cnt, r.lo + q);
    return r;
}

// advance ascending positions idx[0..L) out of d to the next combination; false at the end
template <int L>
__device__ __forceinline__ bool next_comb(int *idx, int d)
{
    int i = L - 1;
    while (i >= 0 && idx[i] == d - L + i) i--;
    if (i < 0) return false;
    idx[i]++;
    for (int j = i + 1; j < L; j++) idx[j] = idx[j - 1] + 1;
    return true;
}

// the same step with static register indexing only, returning the first position that changed (-1 at the end):
// consecutive combinations share everything before that position (the incremental factorisation relies on it)
template <int L>
__device__ __forceinline__ int next_comb_pos(int (&idx)[L], int d)
{
    int pos = -1;
#pragma unroll
    for (int i = 0; i < L; i++)
        if (idx[i] < d - L + i) pos = i;
    if (pos < 0) return -1;
    int base = 0;
#pragma unroll
    for (int i = 0; i < L; i++)
    {
        if (i == pos)
        {
            idx[i]++;
            base = idx[i];
        }
        else if (i > pos)
            idx[i] = base + (i - pos);
    }
    return pos;
}

// workgroup-level flush of the per-lane counters into the level's spread slots
__device__ __forceinline__ void flush_counters(unsigned long long *s_cnt, unsigned long long *slots, unsigned long long a,
                                               unsigned long long b, unsigned long long c, unsigned long long dd)
{
    for (int o = 32; o > 0; o >>= 1)
    {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
        c += __shfl_xor(c, o);
        dd += __shfl_xor(dd, o);
    }
    if ((threadIdx.x & 63) == 0)
    {
        if (a) atomicAdd(&s_cnt[0], a);
        if (b) atomicAdd(&s_cnt[1], b);
        if (c) atomicAdd(&s_cnt[2], c);
        if (dd) atomicAdd(&s_cnt[3], dd);
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        unsigned long long *sl = slots + (size_t)(blockIdx.x & (kCounterSlots - 1)) * 4;
        for (int i = 0; i < 4; i++)
            if (s_cnt[i]) atomicAdd(&sl[i], s_cnt[i]);
    }
}

}  // namespace cusk
