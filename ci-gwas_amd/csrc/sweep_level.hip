// sweep_level.hip -- level 0, bitmap -> CSR compaction, the level-1 pair kernel and result expansion.
//
// Replaces /root/reference/cusk/src/cuPC-S.cu:458-484 (cal_Indepl0), :6355-6432 (scan_compact),
// :486-582 (cal_Indepl1) and their hetcor twins (src/hetcor-cuPC-S.cu:343-486).
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <type_traits>

#include "ci_exact.h"
#include "ci_fast.h"
#include "sweep_common.h"

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

// One 64x64 tile of the upper triangle per workgroup (4 waves x 16 rows, lane = column):
// coalesced 256-byte row segments in, wavefront ballots out (one 64-bit adjacency word per
// tile row, plus the mirrored word through LDS).  The mirrored tile is loaded as well so that
// level 0 also answers "is C bitwise symmetric?" (level 1 then reads only the upper triangle).
// Without per-pair sample sizes the level-0 verdict z(|c|) < th is a comparison of |c| with tanh(th): outside
// a guard band around that value (c_lo, c_hi from the host, +-5e-4 relative, far above the fp32 error of the
// reference's Fisher z) two compares settle the element, inside it (and for NaN or |c| > 1, where the reference's
// formula is not monotone) the reference's arithmetic decides.  All 16 rows of a wave are requested before the
// first is evaluated.
template <bool ESS, bool SYMCHECK>
__global__ void __launch_bounds__(256) level0_kernel(const float *__restrict__ C, const float *__restrict__ N,
                                                      unsigned long long *adj, int n, int words, float th, float c_lo,
                                                      float c_hi, int tiles, int *asym_flag)
{
    __shared__ unsigned long long s_col[64];
    __shared__ float s_t[64][65];
    int t = blockIdx.x, bi = 0;
    {
        int rem = t, len = tiles;
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
    if constexpr (SYMCHECK)
    {
        for (int rr = 0; rr < 16; rr++)
        {
            const int r = wave * 16 + rr;
            const int jr = bj * 64 + r, ic = bi * 64 + lane;
            s_t[r][lane] = (jr < n && ic < n) ? C[(size_t)jr * n + ic] : 0.0f;
        }
    }
    __syncthreads();
    const int j = bj * 64 + lane;
    float cv[16];
    [[maybe_unused]] float nv[16];
#pragma unroll
    for (int rr = 0; rr < 16; rr++)
    {
        const int i = bi * 64 + wave * 16 + rr;
        const bool in = (i < n && j < n && i < j);
        cv[rr] = in ? C[(size_t)i * n + j] : 0.0f;
        if constexpr (ESS) nv[rr] = in ? N[(size_t)i * n + j] : 4.0f;
    }
    unsigned long long colbits = 0ull;
    bool asym = false;
#pragma unroll
    for (int rr = 0; rr < 16; rr++)
    {
        const int r = wave * 16 + rr;
        const int i = bi * 64 + r;
        bool rm = false;
        if (i < n && j < n && i < j)
        {
            const float c = cv[rr];
            if constexpr (SYMCHECK)
            {
                const float ct = s_t[lane][r];
                asym |= (__float_as_uint(c) != __float_as_uint(ct)) && !((c != c) && (ct != ct));
            }
            if constexpr (ESS)
            {
                // per-pair threshold th / sqrt(N_ij - 3): a single-precision estimate of z sqrt(N_ij - 3) settles the
                // element unless it falls within 1e-3 of th (or the threshold is too small for that band, or an
                // operand is unusual); only then the reference's double-precision threshold and Fisher z are formed
                const float nm3 = nv[rr] - 3.0f;
                const float ac = fabsf(c);
                int fastv = 2;
                if (nm3 > 0.0f && nm3 < 3.0e38f && ac < 1.0f && th * __frsqrt_rn(nm3) >= kThMinFilter)
                {
                    const float sest = 0.5f * fabsf(__logf((1.0f + ac) / (1.0f - ac))) * __fsqrt_rn(nm3);
                    if (sest < th * (1.0f - 1e-3f))
                        fastv = 1;
                    else if (sest > th * (1.0f + 1e-3f))
                        fastv = 0;
                }
                if (fastv == 2)
                {
                    const float lth = (float)((double)th / sqrt((double)nv[rr] - 3.0));
                    rm = z_below<false>(c, lth);
                }
                else
                    rm = (fastv == 1);
            }
            else
            {
                const float ac = fabsf(c);
                if (ac < c_lo)
                    rm = true;
                else if (ac > c_hi && ac <= 1.0f)
                    rm = false;
                else
                    rm = z_below<false>(c, th);
            }
        }
        const unsigned long long m = __ballot(rm);
        if (lane == 0 && m != 0ull) atomicAnd(&adj[(size_t)i * words + bj], ~m);
        if (rm) colbits |= (1ull << r);
    }
    if constexpr (SYMCHECK)
    {
        if (__ballot(asym) != 0ull && lane == 0) *asym_flag = 1;
    }
    if (colbits) atomicOr(&s_col[lane], colbits);
    __syncthreads();
    if (threadIdx.x < 64)
    {
        const unsigned long long m = s_col[threadIdx.x];
        const int jj = bj * 64 + threadIdx.x;
        if (m != 0ull && jj < n) atomicAnd(&adj[(size_t)jj * words + bi], ~m);
    }
}

// Wide-tile form for the common case (one threshold, symmetry not checked): 64 rows x 256 columns per workgroup,
// so every row contributes a contiguous 1 KB to the stream instead of 256 B (DRAM pages are opened for a useful
// amount of data).  Lane l of a wave takes columns l, l+64, l+128, l+192 of its 16 rows: four coalesced 256-byte
// loads per row whose ballots are directly the four bitmap words; the mirrored words go through LDS.
constexpr int kL0Cols = 256;
__global__ void __launch_bounds__(256) level0_wide_kernel(const float *__restrict__ C, unsigned long long *adj, int n, int words,
                                                           float th, float c_lo, float c_hi, int col_tiles, int complete_graph)
{
    __shared__ unsigned long long s_col[kL0Cols];
    // tile (bi, bj): rows [64 bi, +64), columns [256 bj, +256); only tiles that reach right of the diagonal
    int bi = 0, t = blockIdx.x;
    {
        // row block bi has col_tiles - (bi / 4) column tiles (its first one contains the diagonal)
        for (;;)
        {
            const int len = col_tiles - (bi >> 2);
            if (t < len) break;
            t -= len;
            bi++;
        }
    }
    const int bj = (bi >> 2) + t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_col[threadIdx.x] = 0ull;
    __syncthreads();
    // Straight-line fast pass, rare slow pass.  (Round 1 evaluated the in-band elements where they were met: 64 inlined
    // copies of the Fisher-z comparison made the kernel 90 KB of code, more than the instruction cache holds, and a
    // conditional request per element; the launch streamed at 2.1 TB/s.)  All 64 requests of a lane are unconditional
    // (indices clamped into the matrix); the fast pass only compares |c| with the band and remembers which of its
    // elements fell inside it (about one in a million); those are fetched again and decided by the reference's form
    // in one rolled loop afterwards.
    const int i0 = bi * 64 + wave * 16, j0 = bj * kL0Cols + lane;
    float cv[16][4];
#pragma unroll
    for (int rr = 0; rr < 16; rr++)
    {
        const size_t ro = (size_t)min(i0 + rr, n - 1) * n;
#pragma unroll
        for (int q = 0; q < 4; q++) cv[rr][q] = C[ro + min(j0 + q * 64, n - 1)];
    }
    unsigned long long colbits[4] = {0ull, 0ull, 0ull, 0ull};
    unsigned need_lo = 0u, need_hi = 0u;  // bit rr * 4 + q: the element needs the exact comparison
#pragma unroll
    for (int rr = 0; rr < 16; rr++)
    {
        const int r = wave * 16 + rr;
        const int i = i0 + rr;
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int j = j0 + q * 64;
            const bool valid = (i < n && j < n && i < j);
            const float ac = fabsf(cv[rr][q]);
            const bool rm = valid && (ac < c_lo);
            const bool need = valid && !(ac < c_lo) && !(ac > c_hi && ac <= 1.0f);  // in the band, |c| > 1, NaN
            const unsigned long long m = __ballot(rm);
            if (lane == 0 && m != 0ull)
            {
                // Every bitmap word has one owner except those that straddle the diagonal: word (row, w) with
                // w > row / 64 is decided entirely here (plain store of the complete word), w < row / 64 entirely by
                // a mirrored tile; only w == row / 64 collects bits from both sides and needs the atomic.  (With a
                // caller-supplied starting graph the words are not all ones: atomics throughout.)
                const int w = bj * 4 + q;
                unsigned long long *dst = &adj[(size_t)i * words + w];
                if (complete_graph && w != (i >> 6))
                {
                    const int nv = n - w * 64;
                    *dst = ((nv >= 64) ? ~0ull : ((1ull << nv) - 1ull)) & ~m;
                }
                else
                    atomicAnd(dst, ~m);
            }
            if (rm) colbits[q] |= (1ull << r);
            if (need)
            {
                if (rr * 4 + q < 32)
                    need_lo |= 1u << ((rr * 4 + q) & 31);
                else
                    need_hi |= 1u << ((rr * 4 + q) & 31);
            }
        }
    }
    if (__ballot((need_lo | need_hi) != 0u) != 0ull)
    {
        // this wave's plain stores above are performed before the read-modify-writes below touch the same words
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        unsigned long long nm = ((unsigned long long)need_hi << 32) | need_lo;
        while (nm != 0ull)
        {
            const int e = __builtin_ctzll(nm);
            nm &= nm - 1ull;
            const int rr = e >> 2, q = e & 3;
            const int i = i0 + rr, j = j0 + q * 64;
            const float c = C[(size_t)i * n + j];
            if (z_below<false>(c, th))
            {
                atomicAnd(&adj[(size_t)i * words + bj * 4 + q], ~(1ull << lane));  // j % 64 == lane
                atomicOr(&s_col[q * 64 + lane], 1ull << (wave * 16 + rr));
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (colbits[q]) atomicOr(&s_col[q * 64 + lane], colbits[q]);
    __syncthreads();
    {
        const unsigned long long m = s_col[threadIdx.x];
        const int jj = bj * kL0Cols + threadIdx.x;
        if (m != 0ull && jj < n)
        {
            unsigned long long *dst = &adj[(size_t)jj * words + bi];
            if (complete_graph && bi != (jj >> 6))
                *dst = ~m;  // all 64 rows of block bi lie above row jj and exist
            else
                atomicAnd(dst, ~m);
        }
    }
}

// The same tile with a fifth of the instructions (round 3).  The form above spends ~3,900 instructions per wave on its 64
// elements (PMC: 1,950 vector + 2,000 scalar; the launch is bound by instruction issue at 2.3 TB/s, not by the stream):
// per element a conditional single-lane store with its own address, the word-ownership logic, a need-bit.  Here an
// element costs a compare (its ballot IS the bitmap word), three instructions that park the word in lane 4 rr + q, two
// for the mirrored bit and three for "does anything of this lane need the exact comparison": the 64 words
// of the wave leave in ONE store instruction (lane L owns word (rr, q) = (L / 4, L % 4), ownership logic evaluated
// once per lane), and the exact pass -- entered by about one wave in 250 -- walks the flagged lanes' elements again.
// EDGE = the tile touches the diagonal or the matrix border (index clamps and per-element validity); interior tiles
// (nine in ten) run without either.
template <bool EDGE>
__device__ __forceinline__ void level0_tile(const float *__restrict__ C, unsigned long long *adj, int n, int words, float th,
                                            float c_lo, float c_hi, int bi, int bj, int complete_graph,
                                            unsigned long long *s_col)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = bi * 64 + wave * 16, j0 = bj * kL0Cols + lane;
    float cv[16][4];
    if constexpr (EDGE)
    {
#pragma unroll
        for (int rr = 0; rr < 16; rr++)
        {
            const size_t ro = (size_t)min(i0 + rr, n - 1) * n;
#pragma unroll
            for (int q = 0; q < 4; q++) cv[rr][q] = C[ro + min(j0 + q * 64, n - 1)];
        }
    }
    else
    {
        const float *base = C + (size_t)i0 * n + j0;
#pragma unroll
        for (int rr = 0; rr < 16; rr++)
#pragma unroll
            for (int q = 0; q < 4; q++) cv[rr][q] = base[(size_t)rr * n + q * 64];
    }
    // |c| within (slightly more than) the guard band <=> | |c| - mid | <= hw: the lane keeps the smallest such distance and
    // the largest |c| it saw (|c| > 1 is where the reference's formula is not monotone) and is "flagged" by either -- a
    // superset of the elements the exact pass decides (it applies the band test of the form above again, element by
    // element).  NaN is never flagged: it fails `|c| < c_lo` here and every comparison of the exact form, so the edge
    // stays either way.  (The accumulations are kept dependent chains by empty asm statements: OR-ing 64 independent
    // terms lets the compiler build a tree at the end, which keeps all 64 ballots alive -- 700 scalar-register spills.)
    const float mid = 0.5f * (c_lo + c_hi), hw = (c_hi - c_lo) * 0.51f + 1e-30f;
    float dmin = 3.0e38f, amax = 0.0f;
    unsigned wlo = 0u, whi = 0u;
    unsigned cb[4] = {0u, 0u, 0u, 0u};  // bit 15 - rr: element (rr, q) of this lane's column goes
#pragma unroll
    for (int rr = 0; rr < 16; rr++)
    {
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const float ac = fabsf(cv[rr][q]);
            bool rm = ac < c_lo;
            if constexpr (EDGE)
            {
                const int i = i0 + rr, j = j0 + q * 64;
                rm = rm && (i < n && j < n && i < j);
            }
            const unsigned long long m = __ballot(rm);
            const bool mine = lane == rr * 4 + q;
            wlo = mine ? (unsigned)m : wlo;
            whi = mine ? (unsigned)(m >> 32) : whi;
            cb[q] = (cb[q] << 1) | (rm ? 1u : 0u);
            asm volatile("" : "+v"(cb[q]), "+v"(wlo), "+v"(whi));  // (opaque: chains stay chains, element by element)
            dmin = fminf(dmin, fabsf(ac - mid));
            amax = fmaxf(amax, ac);
        }
        asm volatile("" : "+v"(dmin), "+v"(amax));
    }
    {
        // Every bitmap word has one owner except those that straddle the diagonal: word (row, w) with w > row / 64 is
        // decided entirely here (plain store of the complete word), w < row / 64 entirely by a mirrored tile; only
        // w == row / 64 collects bits from both sides and needs the atomic.  (With a caller-supplied starting graph
        // the words are not all ones: atomics throughout.)
        const int i = i0 + (lane >> 2), w = bj * 4 + (lane & 3);
        const unsigned long long m = ((unsigned long long)whi << 32) | wlo;
        if (m != 0ull)  // (EDGE: m is empty for rows and columns outside the matrix)
        {
            unsigned long long *dst = &adj[(size_t)i * words + w];
            if (complete_graph && w != (i >> 6))
            {
                const int nv = n - w * 64;
                *dst = ((nv >= 64) ? ~0ull : ((1ull << nv) - 1ull)) & ~m;
            }
            else
                atomicAnd(dst, ~m);
        }
    }
    const bool flagged = (dmin <= hw) || (amax > 1.0f);
    if (__ballot(flagged) != 0ull)
    {
        // exact pass (uniform branch, a fraction of a percent of the waves on unrelated markers, more along the diagonal
        // of an LD block where |c| reaches 1 + 1 ulp): which of the 64 elements -- still in registers -- need the
        // reference's arithmetic, then only those are decided, one rolled loop as in the form above
        unsigned need_lo = 0u, need_hi = 0u;  // bit rr * 4 + q
        [[maybe_unused]] int j0s = j0;
        asm volatile("" : "+v"(j0s));  // (likewise: the validity masks of the fast pass are not kept either)
#pragma unroll
        for (int rr = 0; rr < 16; rr++)
        {
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                float cq = cv[rr][q];
                asm volatile("" : "+v"(cq));  // (a value of its own: the fast pass's 64 compare masks are not kept for this)
                const float ac = fabsf(cq);
                bool need = !(ac < c_lo) && !(ac > c_hi && ac <= 1.0f);  // in the band, |c| > 1, NaN
                if constexpr (EDGE)
                {
                    const int i = i0 + rr, j = j0s + q * 64;
                    need = need && (i < n && j < n && i < j);
                }
                if (rr * 4 + q < 32)
                    need_lo |= need ? (1u << ((rr * 4 + q) & 31)) : 0u;
                else
                    need_hi |= need ? (1u << ((rr * 4 + q) & 31)) : 0u;
                asm volatile("" : "+v"(need_lo), "+v"(need_hi));
            }
        }
        // this wave's plain stores above are performed before the read-modify-writes below touch the same words
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        unsigned long long nm = ((unsigned long long)need_hi << 32) | need_lo;
        while (nm != 0ull)
        {
            const int e = __builtin_ctzll(nm);
            nm &= nm - 1ull;
            const int rr = e >> 2, q = e & 3;
            const int i = i0 + rr, j = j0 + q * 64;
            const float c = C[(size_t)i * n + j];
            if (z_below<false>(c, th))
            {
                atomicAnd(&adj[(size_t)i * words + bj * 4 + q], ~(1ull << lane));  // j % 64 == lane
                atomicOr(&s_col[q * 64 + lane], 1ull << (wave * 16 + rr));
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (cb[q]) atomicOr(&s_col[q * 64 + lane], (unsigned long long)(__brev(cb[q]) >> 16) << (wave * 16));
}

__global__ void __launch_bounds__(256) level0_wide2_kernel(const float *__restrict__ C, unsigned long long *adj, int n, int words,
                                                            float th, float c_lo, float c_hi, int col_tiles, int complete_graph)
{
    __shared__ unsigned long long s_col[kL0Cols];
    // tile (bi, bj): rows [64 bi, +64), columns [256 bj, +256); only tiles that reach right of the diagonal
    int bi = 0, t = blockIdx.x;
    for (;;)
    {
        // row block bi has col_tiles - (bi / 4) column tiles (its first one contains the diagonal)
        const int len = col_tiles - (bi >> 2);
        if (t < len) break;
        t -= len;
        bi++;
    }
    const int bj = (bi >> 2) + t;
    s_col[threadIdx.x] = 0ull;
    __syncthreads();
    const bool interior = (bj * kL0Cols > bi * 64 + 63) && (bi * 64 + 64 <= n) && (bj * kL0Cols + kL0Cols <= n);
    if (interior)
        level0_tile<false>(C, adj, n, words, th, c_lo, c_hi, bi, bj, complete_graph, s_col);
    else
        level0_tile<true>(C, adj, n, words, th, c_lo, c_hi, bi, bj, complete_graph, s_col);
    __syncthreads();
    {
        const unsigned long long m = s_col[threadIdx.x];
        const int jj = bj * kL0Cols + threadIdx.x;
        if (m != 0ull && jj < n)
        {
            unsigned long long *dst = &adj[(size_t)jj * words + bi];
            if (complete_graph && bi != (jj >> 6))
                *dst = ~m;  // all 64 rows of block bi lie above row jj and exist
            else
                atomicAnd(dst, ~m);
        }
    }
}

hipError_t launch_level0(const float *C, const float *Ness, const int *Ginit, unsigned long long *adj, int n, int words,
                         float th, int *asym_flag, hipStream_t st)
{
    hipLaunchKernelGGL(init_bits_kernel, dim3(n), dim3(64), 0, st, adj, Ginit, n, words);
    const int tiles = words;
    const long long ntile = (long long)tiles * (tiles + 1) / 2;
    const dim3 grid((unsigned)ntile), block(256);
    // guard band of the |c| comparison; thresholds below kThMinFilter get an empty fast range (exact everywhere)
    float c_lo = 0.0f, c_hi = 2.0f;
    if (th >= kThMinFilter)
    {
        const double tq = std::tanh((double)th);
        c_lo = (float)(tq * (1.0 - 5e-4));
        c_hi = (float)(tq * (1.0 + 5e-4));
    }
    if (!Ness && !asym_flag)
    {
        const int col_tiles = (n + kL0Cols - 1) / kL0Cols;
        long long nt = 0;
        for (int bi = 0; bi < tiles; bi++) nt += col_tiles - (bi >> 2);
        static const bool old_form = std::getenv("CUSK_L0_OLD") != nullptr;  // (A/B timing of the round-2 form)
        if (old_form)
            hipLaunchKernelGGL(level0_wide_kernel, dim3((unsigned)nt), block, 0, st, C, adj, n, words, th, c_lo, c_hi, col_tiles,
                               Ginit == nullptr ? 1 : 0);
        else
            hipLaunchKernelGGL(level0_wide2_kernel, dim3((unsigned)nt), block, 0, st, C, adj, n, words, th, c_lo, c_hi, col_tiles,
                               Ginit == nullptr ? 1 : 0);
    }
    else if (Ness && asym_flag)
        hipLaunchKernelGGL((level0_kernel<true, true>), grid, block, 0, st, C, Ness, adj, n, words, th, c_lo, c_hi, tiles,
                           asym_flag);
    else if (Ness)
        hipLaunchKernelGGL((level0_kernel<true, false>), grid, block, 0, st, C, Ness, adj, n, words, th, c_lo, c_hi, tiles,
                           asym_flag);
    else
        hipLaunchKernelGGL((level0_kernel<false, true>), grid, block, 0, st, C, Ness, adj, n, words, th, c_lo, c_hi, tiles,
                           asym_flag);
    return hipGetLastError();
}

// Block-diagonal level 0 (batched runs: many small LD blocks along the diagonal of one allocation, bases multiples of 64,
// so no bitmap word straddles two blocks).  One wave per row: the words of the row's own block come from C[row, lo..hi)
// (coalesced 256-byte pieces, ballots = bitmap words), every other word of the row is zero -- the cross-block pairs do not
// exist.  The verdict is cal_Indepl0's (cuPC-S.cu:458-484) evaluated per ordered pair; the reference evaluates i < j and
// mirrors, which is the same thing on a bitwise symmetric matrix (the batched correlation build and the device gather of
// a symmetric matrix write both triangles from one value).  Writes the live bitmap, its level-0 copy and the degrees.
__global__ void __launch_bounds__(256) level0_batch_kernel(const float *__restrict__ C, unsigned long long *adj,
                                                            unsigned long long *adj0, int *deg, int n, int words,
                                                            const int2 *__restrict__ row_range, float th, float c_lo, float c_hi)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const int2 rg = row_range[row];
    const int w_lo = rg.x >> 6, w_hi = (rg.y + 63) >> 6;  // empty range: w_lo >= w_hi
    unsigned long long *arow = adj + (size_t)row * words, *arow0 = adj0 ? adj0 + (size_t)row * words : nullptr;
    for (int w = lane; w < words; w += 64)
        if (w < w_lo || w >= w_hi || rg.y <= rg.x)
        {
            arow[w] = 0ull;
            if (arow0) arow0[w] = 0ull;
        }
    int d = 0;
    if (rg.y > rg.x)
    {
        const float *crow = C + (size_t)row * n;
        for (int w = w_lo; w < w_hi; w++)
        {
            const int col = w * 64 + lane;
            const bool valid = col < rg.y && col != row;
            const float c = crow[valid ? col : row];
            const float ac = fabsf(c);
            bool rm;
            if (ac < c_lo)
                rm = true;
            else if (ac > c_hi && ac <= 1.0f)
                rm = false;
            else
                rm = z_below<false>(c, th);
            const unsigned long long m = __ballot(valid && !rm);
            if (lane == 0)
            {
                arow[w] = m;
                if (arow0) arow0[w] = m;
            }
            d += __popcll(m);
        }
    }
    if (lane == 0) deg[row] = d;
}

hipError_t launch_level0_batch(const float *C, unsigned long long *adj, unsigned long long *adj0, int *deg, int n, int words,
                               const int2 *row_range, float th, hipStream_t st)
{
    float c_lo = 0.0f, c_hi = 2.0f;
    if (th >= kThMinFilter)
    {
        const double tq = std::tanh((double)th);
        c_lo = (float)(tq * (1.0 - 5e-4));
        c_hi = (float)(tq * (1.0 + 5e-4));
    }
    hipLaunchKernelGGL(level0_batch_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, C, adj, adj0, deg, n, words, row_range,
                       th, c_lo, c_hi);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// compaction: bitmap -> CSR neighbour lists + work list
// ---------------------------------------------------------------------------

// degrees after level 0; the same pass leaves the level-0 copy of the bitmap (adj0: record slots, pMax) when asked to
__global__ void degree_kernel(const unsigned long long *__restrict__ adj, int *deg, int n, int words, unsigned long long *adj0)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    int d = 0;
    for (int w = lane; w < words; w += 64)
    {
        const unsigned long long v = adj[(size_t)row * words + w];
        if (adj0) adj0[(size_t)row * words + w] = v;
        d += __popcll(v);
    }
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if (lane == 0) deg[row] = d;
}

hipError_t launch_degree(const unsigned long long *adj, int *deg, int n, int words, unsigned long long *adj0, hipStream_t st)
{
    hipLaunchKernelGGL(degree_kernel, dim3((n + 3) / 4), dim3(256), 0, st, adj, deg, n, words, adj0);
    return hipGetLastError();
}

// one wave per row: ascending neighbour indices, reset of the row's selection state
__global__ void fill_nbr_kernel(const unsigned long long *__restrict__ adj, const int *__restrict__ off, int *nbr,
                                unsigned long long *best, int n, int words, int *wpre, const LevelCounters *cnt,
                                const int2 *__restrict__ row_range)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n || !cnt->active) return;
    const int o0 = off[row];
    int run = 0;
    // batched runs: only the words of the row's own block can hold neighbours
    int w_begin = 0, w_end = words;
    if (row_range)
    {
        const int2 rg = row_range[row];
        w_begin = rg.x >> 6;
        w_end = (rg.y > rg.x) ? ((rg.y + 63) >> 6) : w_begin;
    }
    for (int w0 = w_begin; w0 < w_end; w0 += 64)
    {
        const int w = w0 + lane;
        unsigned long long bits = (w < w_end) ? adj[(size_t)row * words + w] : 0ull;
        const int c = __popcll(bits);
        int incl = c;
        for (int o = 1; o < 64; o <<= 1)
        {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        int pos = o0 + run + incl - c;
        // list position of the word's first neighbour: lets anyone turn (row, column) into a list position with
        // one more popcount (level1_prep_kernel)
        if (wpre && w < w_end) wpre[(size_t)row * words + w] = run + incl - c;
        while (bits)
        {
            const int b = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            if (best) best[pos] = kNone;
            nbr[pos++] = w * 64 + b;
        }
        run += __shfl(incl, 63);
    }
}

hipError_t launch_fill_nbr(const unsigned long long *adj, const int *off, int *nbr, unsigned long long *best, int n, int words,
                           int *wpre, const LevelCounters *cnt, const int2 *row_range, hipStream_t st)
{
    hipLaunchKernelGGL(fill_nbr_kernel, dim3((n + 3) / 4), dim3(256), 0, st, adj, off, nbr, best, n, words, wpre, cnt, row_range);
    return hipGetLastError();
}

// Exclusive prefix of in[0..n) for the 256 rows of this workgroup, without a second kernel and without atomics:
// the workgroup first sums everything that lies before its block (redundantly with its peers: block b reads 256 b
// values, 40 loads per thread at 10k rows), then scans its own 256 values.  Returns the thread's exclusive prefix;
// *block_total receives the sum over the workgroup's own rows.  s_red: 8 long longs of LDS.
__device__ __forceinline__ long long prefix_256(const int *__restrict__ in, int n, int v, long long *s_red, long long *block_total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = blockIdx.x * 256;
    long long before = 0;
    for (int i0 = 0; i0 < first; i0 += 256 * 16)
    {  // sixteen independent loads in flight per thread
        int t[16];
#pragma unroll
        for (int u = 0; u < 16; u++)
        {
            const int i = i0 + u * 256 + tid;
            t[u] = (i < first) ? in[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < 16; u++) before += t[u];
    }
    int incl = v;
    for (int o = 1; o < 64; o <<= 1)
    {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if (lane == 63) s_red[wave] = incl;
    if (lane == 0) s_red[4 + wave] = before;
    __syncthreads();
    long long pre = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    for (int w = 0; w < wave; w++) pre += s_red[w];
    *block_total = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    (void)n;
    return pre + incl - v;
}

// The level's work plan, from the degrees alone (so it runs before the neighbour lists are even written): CSR offsets,
// class and work-item count of every row, the work items themselves, level totals, and the level's gate.  256 rows per
// workgroup.
//
// Round 2, second form: no same-address atomics.  The first form placed a workgroup's items with one returning atomicAdd
// per class, took the maximum degree with an atomicMax and found the last workgroup with a ticket counter: three to seven
// device-scope atomics on the same few addresses from every workgroup, which serialise at ~0.3-0.5 us each across the
// XCDs -- 18-23 us per level for a kernel that moves 40 KB.  Now every workgroup PUBLISHES its block totals (degree sum,
// items per class, maximum degree) as self-validating 64-bit words, (sequence number << 40) | value, written with
// agent-scope stores, and reads the words of the blocks before it (spinning on a word until it carries this launch's
// sequence number: no fence, no flag).  Exclusive sums over the earlier blocks give the CSR offsets and the placement
// of the items -- in row order, so the item lists are deterministic -- and the LAST block, which has seen every other
// block's totals, closes or opens the gate.  Forward progress: a workgroup only waits for workgroups with a smaller
// index, which the dispatcher starts first and which never wait for a larger one.
constexpr int kPlanWords = 8;  // per block: [0] degree sum, [1 .. kNumClasses] items per class, [6] maximum degree, [7] overflow
static_assert(kNumClasses + 1 < 7, "plan words");
constexpr unsigned long long kPlanMask = (1ull << 40) - 1ull;

__global__ void __launch_bounds__(256) plan_kernel(PlanArgs a)
{
    __shared__ long long s_wave[kNumClasses][4];  // (64-bit: 64 hub rows of a deep level can hold more than 2^31 items together)
    __shared__ int s_wdeg[4], s_wmax[4];
    __shared__ long long s_part[4][kPlanWords];
    __shared__ long long s_base[kNumClasses];
    __shared__ int s_cls[256], s_nch[256], s_pos[256];
    __shared__ unsigned long long s_big[4];
    LevelCounters *cnt = a.cnt;
    const int n = a.n, L = a.L;
    // the previous level did not run to completion (the loop ended there, or its recheck queue overflowed and it is
    // going to be redone): nothing of this level may touch the working sets; the gate stays closed (counters are zeroed
    // at run start)
    if (a.prev != nullptr && !level_complete(a.prev, a.prev_qcap))
    {
        if (blockIdx.x == 0 && threadIdx.x == 0)
        {
            HostGate *g = a.gate;
            g->active = 0;
            g->maxdeg = 0;
            g->overflow = 0;
            g->item_overflow = 0;
            g->sym = 0;
            g->total_edges = 0;
            for (int c = 0; c < kNumClasses; c++) g->class_items[c] = 0;
            __threadfence_system();
            *(volatile int *)&g->seq = a.seq;
        }
        return;
    }
    const int tid = threadIdx.x;
    const int row = blockIdx.x * 256 + tid;
    const int lane = tid & 63, wave = tid >> 6;
    int cls = -1, nchunks = 0;
    const int d = (row < n) ? a.deg[row] : 0;
    bool ovf = false;
    if (d > L)
    {
        // work units of the row: conditioning sets, or unordered neighbour pairs for the pair kernel
        const unsigned long long nc = a.pair_mode ? (unsigned long long)d * (d - 1) / 2
                                                  : (L == 1 ? (unsigned long long)d : a.binom[(size_t)d * kBinomStride + a.Lsets]);
        if (nc >= (1ull << 62))
            ovf = true;
        else
        {
            cls = 0;
            while (d > kClassCap[cls]) cls++;
            if (cls >= a.staged_classes) cls = kNumClasses - 1;
            const unsigned long long ch = (cls == 0) ? a.chunk0 : a.chunk;
            if (a.Lsets == L + 1 && L >= 2)
            {  // union-major level (sweep_tmaj.hip): items per end position s of the prefix
                const int np = L - 2;
                long long tot = 0;
                for (int s = np; s <= d - 3; s++)
                {
                    const unsigned long long nP = (np == 0) ? 1ull : a.binom[(size_t)s * kBinomStride + np];
                    const unsigned long long per = (unsigned long long)kThreads * tmaj_prefixes_per_lane(d, s, ch);
                    tot += (long long)((nP + per - 1ull) / per);
                }
                if (tot > (long long)0x7fffffff)
                    ovf = true;  // more work items than a row may have: reported like a binomial overflow
                else
                    nchunks = (int)tot;
            }
            else
            {
                const unsigned long long units = (nc + ch - 1) / ch;
                if (units > 0x7fffffffull)
                    ovf = true;
                else
                    nchunks = (int)units;
            }
            // row-sharded runs: only the owner of a row enumerates it (offsets, totals and the overflow checks are global)
            if (ovf || !(a.shard_world == 1 || row % a.shard_world == a.shard_rank))
            {
                cls = -1;
                nchunks = 0;
            }
        }
    }
    const bool wave_ovf = __ballot(ovf) != 0ull;
    // ---- scans inside the workgroup: degrees, items per class, maximum degree ----
    int dincl = d, dmax = d;
    for (int o = 1; o < 64; o <<= 1)
    {
        const int t = __shfl_up(dincl, o);
        if (lane >= o) dincl += t;
    }
    for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, __shfl_xor(dmax, o));
    if (lane == 63) s_wdeg[wave] = dincl;
    if (lane == 0) s_wmax[wave] = dmax | (wave_ovf ? (1 << 30) : 0);  // bit 30: a row's C(d, l) does not fit 62 bits
    long long excl[kNumClasses];
#pragma unroll
    for (int c = 0; c < kNumClasses; c++)
    {
        const long long mine = (cls == c) ? nchunks : 0;
        long long v = mine;
        for (int o = 1; o < 64; o <<= 1)
        {
            const long long t = __shfl_up(v, o);
            if (lane >= o) v += t;
        }
        excl[c] = v - mine;
        if (lane == 63) s_wave[c][wave] = v;
    }
    __syncthreads();
    // ---- publish this block's totals ----
    unsigned long long *blk = a.blocks + (size_t)blockIdx.x * kPlanWords;
    const unsigned long long tag = (unsigned long long)a.blk_seq << 40;
    if (tid < kPlanWords)
    {
        unsigned long long v = 0;
        if (tid == 0)
            v = (unsigned long long)((long long)s_wdeg[0] + s_wdeg[1] + s_wdeg[2] + s_wdeg[3]);
        else if (tid <= kNumClasses)
            v = (unsigned long long)(s_wave[tid - 1][0] + s_wave[tid - 1][1] + s_wave[tid - 1][2] + s_wave[tid - 1][3]);
        else if (tid == 6)
            v = (unsigned long long)(max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])) & ~(1 << 30));
        else if (tid == 7)
            v = (unsigned long long)(((s_wmax[0] | s_wmax[1] | s_wmax[2] | s_wmax[3]) >> 30) & 1);
        __hip_atomic_store(&blk[tid], tag | (v & kPlanMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- totals of the blocks before this one: thread <-> (block j, word w), 32 blocks per round ----
    long long acc = 0;
    {
        const int w = tid & 7;
        for (int j = tid >> 3; j < (int)blockIdx.x; j += 32)
        {
            const unsigned long long *src = a.blocks + (size_t)j * kPlanWords + w;
            unsigned long long x;
            while (((x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 40) != (unsigned long long)a.blk_seq)
                __builtin_amdgcn_s_sleep(1);
            const long long v = (long long)(x & kPlanMask);
            acc = (w >= 6) ? max(acc, v) : acc + v;
        }
        for (int o = 8; o < 64; o <<= 1)
        {
            const long long t = __shfl_xor(acc, o);
            acc = (w >= 6) ? max(acc, t) : acc + t;
        }
        if (lane < kPlanWords) s_part[wave][lane] = acc;
    }
    __syncthreads();
    long long before[kPlanWords];
#pragma unroll
    for (int w = 0; w < kPlanWords; w++)
    {
        const long long p0 = s_part[0][w], p1 = s_part[1][w], p2 = s_part[2][w], p3 = s_part[3][w];
        before[w] = (w >= 6) ? max(max(p0, p1), max(p2, p3)) : p0 + p1 + p2 + p3;
    }
    // CSR offsets of the level (exclusive prefix of the degrees)
    {
        long long o0 = before[0] + dincl - d;
        for (int w = 0; w < wave; w++) o0 += s_wdeg[w];
        if (row < n) a.off[row] = (int)o0;
    }
    const bool last = (blockIdx.x == gridDim.x - 1);
    long long tot_items[kNumClasses];
#pragma unroll
    for (int c = 0; c < kNumClasses; c++)
    {
        const long long mine = s_wave[c][0] + s_wave[c][1] + s_wave[c][2] + s_wave[c][3];
        tot_items[c] = before[1 + c] + mine;  // through this block (the last block: the level's total)
        if (tid == c) s_base[c] = (tot_items[c] > a.item_cap) ? -1 : before[1 + c];  // would not fit: the host grows the buffers
    }
    __syncthreads();
    // the work items of the workgroup's rows, written cooperatively (a hub row has hundreds of them)
    {
        int pos = -1;
        if (cls >= 0 && s_base[cls] >= 0)
        {  // (the class total fits the buffer here, so every position does)
            long long add = 0;
            for (int w = 0; w < wave; w++) add += s_wave[cls][w];
            long long e = 0;
#pragma unroll
            for (int c = 0; c < kNumClasses; c++)
                if (cls == c) e = excl[c];
            pos = (int)(s_base[cls] + add + e);
        }
        // rows with a single item write it themselves; the few rows with several (hubs: hundreds) are handled by the
        // whole workgroup, found through ballots instead of a walk over all 256 rows
        const bool big = (pos >= 0 && nchunks > 1);
        if (pos >= 0 && nchunks == 1) a.items[cls][pos] = make_int2(row, 0);
        const unsigned long long bm = __ballot(big);
        if (lane == 0) s_big[wave] = bm;
        if (big)
        {
            s_cls[tid] = cls;
            s_nch[tid] = nchunks;
            s_pos[tid] = pos;
        }
        __syncthreads();
        const int row0 = blockIdx.x * 256;
        for (int w = 0; w < 4; w++)
        {
            unsigned long long m = s_big[w];
            while (m)
            {
                const int r = w * 64 + __builtin_ctzll(m);
                m &= m - 1;
                const int nch = s_nch[r];
                int2 *dst = a.items[s_cls[r]] + s_pos[r];
                for (int c = tid; c < nch; c += 256) dst[c] = make_int2(row0 + r, c);
            }
        }
    }
    // ---- the level's totals and gate: the last block has seen every block's words ----
    if (last && tid == 0)
    {
        const long long edges = before[0] + s_wdeg[0] + s_wdeg[1] + s_wdeg[2] + s_wdeg[3];
        const int maxdeg = (int)max(before[6], (long long)(max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])) & ~(1 << 30)));
        a.off[n] = (int)edges;
        const int ovf_all = (before[7] != 0 || (((s_wmax[0] | s_wmax[1] | s_wmax[2] | s_wmax[3]) >> 30) & 1)) ? 1 : 0;
        cnt->overflow = ovf_all;
        bool fits = true;
        HostGate *g = a.gate;
        for (int c = 0; c < kNumClasses; c++)
        {
            fits = fits && (tot_items[c] <= a.item_cap);
            cnt->class_items[c] = tot_items[c];
            g->class_items[c] = tot_items[c];
        }
        const int active = (maxdeg - 1 >= L && ovf_all == 0 && fits) ? 1 : 0;
        cnt->maxdeg = maxdeg;
        cnt->total_edges = edges;
        cnt->item_overflow = fits ? 0 : 1;
        cnt->active = active;
        g->active = active;
        g->maxdeg = maxdeg;
        g->overflow = ovf_all;
        g->item_overflow = fits ? 0 : 1;
        g->sym = a.sym ? *a.sym : 0;
        g->total_edges = edges;
        __threadfence_system();
        *(volatile int *)&g->seq = a.seq;
    }
}

hipError_t launch_plan(const PlanArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(plan_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// level 1 on a symmetric matrix with a single threshold
// ---------------------------------------------------------------------------
// A level-1 test (X ; Y | S) needs C[X,Y], C[X,S] and C[Y,S]; the first two live in row X and
// are staged once, the third is used by exactly two tests, (X;Y|S) and (X;S|Y), so staging a
// (d+1)^2 sub-matrix buys no reuse.  Lane <-> unordered neighbour pair {a<b}: ONE 4-byte gather
// of the upper-triangle element C[min,max] feeds both tests, pairs whose two tests are already
// decided are skipped without touching memory, and with ~20 VGPRs the kernel runs at full
// occupancy to hide the gather latency.  Arithmetic is the exact level-1 formula
// (cuPC-S.cu:561-566), so nothing needs rechecking.
template <int MODE>
__global__ void __launch_bounds__(kThreads) level1_pair_kernel(SweepParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_cnt[3];
    const int n = p.n;
    const int tid = threadIdx.x;
    if (tid < 3) s_cnt[tid] = 0ull;
    unsigned long long ntests = 0, nrem = 0;
    // persistent launch: the work items of the class are read on the device (the host does not know their number)
    const long long nitems = level_items(p);
    for (long long it = blockIdx.x; it < nitems; it += gridDim.x)
    {
    if (it != (long long)blockIdx.x) __syncthreads();  // the previous item's readers are done with the LDS copy
    const int2 item = p.items[it];
    const int X = item.x;
    const int o0 = p.off[X];
    const int d = p.off[X + 1] - o0;
    unsigned long long *s_best = reinterpret_cast<unsigned long long *>(smem);
    int *s_nbr = reinterpret_cast<int *>(smem + sizeof(unsigned long long) * d);
    float *s_m1x = reinterpret_cast<float *>(s_nbr + d);
    int *s_ti = reinterpret_cast<int *>(s_m1x + d);
    const int *g_nbr = p.nbr + o0;
    for (int k = tid; k < d; k += kThreads)
    {
        const int y = g_nbr[k];
        s_nbr[k] = y;
        s_m1x[k] = p.C[(size_t)X * n + y];
        if constexpr (MODE == 0)
            s_best[k] = p.best[o0 + k];
        else
        {
            const unsigned long long wv = p.adj[(size_t)X * p.words + (y >> 6)];
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
    const unsigned long long lo = r0 + (unsigned long long)tid * q;
    const unsigned long long hi = min(r0 + cntr, lo + q);
    if (lo < hi)
    {
        // unrank the pair: row a of the strict upper triangle starts at a*d - a(a+1)/2
        auto start_of = [&](int aa) -> unsigned long long {
            return (unsigned long long)aa * d - (unsigned long long)aa * (aa + 1) / 2;
        };
        const double dd = (double)d - 0.5;
        int a = (int)(dd - sqrt(fmax(dd * dd - 2.0 * (double)lo, 0.0)));
        a = max(0, min(a, d - 2));
        while (a > 0 && start_of(a) > lo) a--;
        while (a < d - 2 && start_of(a + 1) <= lo) a++;
        int b = a + 1 + (int)(lo - start_of(a));
        auto apply = [&](int ky, int ks) {
            // edge X - nbr[ky] is separated by S = nbr[ks]
            if constexpr (MODE == 0)
            {
                const unsigned long long old = atomicMin(&p.best[o0 + ky], (unsigned long long)ks);
                atomicMin(&s_best[ky], (unsigned long long)ks);
                if (old == kNone) nrem++;
            }
            else
            {
                if (clear_edge(p.adj, p.deg, p.words, X, s_nbr[ky])) nrem++;
                s_best[ky] = 0ull;
            }
        };
        for (unsigned long long it = lo; it < hi; it++)
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
    }  // work items
    for (int o = 32; o > 0; o >>= 1)
    {
        ntests += __shfl_xor(ntests, o);
        nrem += __shfl_xor(nrem, o);
    }
    __syncthreads();
    if ((tid & 63) == 0)
    {
        atomicAdd(&s_cnt[0], ntests);
        if (MODE == 1) atomicAdd(&s_cnt[2], nrem);  // Skeleton mode counts removals when it finalises the level
    }
    __syncthreads();
    if (tid == 0)
    {
        unsigned long long *sl = p.slots + (size_t)(blockIdx.x & (kCounterSlots - 1)) * 4;
        if (s_cnt[0]) atomicAdd(&sl[0], s_cnt[0]);
        if (s_cnt[2]) atomicAdd(&sl[2], s_cnt[2]);
    }
}

// Workgroups of a persistent launch: as many as the chip holds at once for this kernel (twice that, so that a
// workgroup that drew short items does not leave its slot empty), never more than the work-item buffer holds.
unsigned persistent_grid(const void *kernel, int threads, size_t lds)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, size_t>, unsigned> cache;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(kernel, lds);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu <= 0)
    {
        (void)hipGetLastError();
        per_cu = 1;
    }
    if (hipGetDevice(&dev) == hipSuccess)
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const unsigned g = (unsigned)std::max(1, per_cu) * (unsigned)cus * 2u;
    cache[key] = g;
    return g;
}

hipError_t launch_pair(int mode, const SweepParams &p, size_t lds, hipStream_t st)
{
    const void *kf = mode == 0 ? reinterpret_cast<const void *>(level1_pair_kernel<0>) : reinterpret_cast<const void *>(level1_pair_kernel<1>);
    const unsigned grid = (unsigned)std::min<long long>(persistent_grid(kf, kThreads, lds), std::max<long long>(p.grid_cap, 1));
    if (mode == 0)
        hipLaunchKernelGGL(level1_pair_kernel<0>, dim3(grid), dim3(kThreads), lds, st, p);
    else
        hipLaunchKernelGGL(level1_pair_kernel<1>, dim3(grid), dim3(kThreads), lds, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// level 1, row-streaming form (symmetric C, single threshold): HBM traffic = C once
// ---------------------------------------------------------------------------
// The pair kernel above pays one 64-byte HBM sector for every 4-byte operand C[Y,S], because a workgroup owns an
// X and its operands C[ya, yb] are scattered over as many rows as X has neighbours.  Here the loop nest is turned
// inside out: a workgroup owns one ROW ya of C and runs every level-1 test that needs an element of that row:
// for each X adjacent to ya and each later neighbour yb of X, the element C[ya,yb] feeds the two tests
// (X; ya | yb) and (X; yb | ya).  All of the workgroup's reads of C fall into that one row (and, with LD, mostly
// into a narrow window behind the diagonal), so HBM sees each touched sector once and L1/L2 serve the rest.
// Everything else those tests need is per-edge data that was compacted at level start and is read contiguously:
// the neighbour list of X, the gathered row values rv = C[X, adj(X)], and meta (position of ya inside X's list); the
// selection state sel is only written (fire-and-forget minima).

// Per CSR slot (row, k) with Y = nbr_k:  rv = C[row, Y];  meta = {Y, position of row inside Y's ascending
// list, start of Y's list, degree of Y};  sel = kNone32 (no separating set yet).
constexpr unsigned kNone32 = 0xffffffffu;

__global__ void level1_prep_kernel(const float *__restrict__ C, const int *__restrict__ off, const int *__restrict__ nbr,
                                   const unsigned long long *__restrict__ adj, const int *__restrict__ wpre, int words,
                                   float *rv, int4 *meta, unsigned *sel, int n, const LevelCounters *cnt)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n || !cnt->active) return;
    const int o0 = off[row], d = off[row + 1] - o0;
    const int rw = row >> 6;
    const unsigned long long below = (1ull << (row & 63)) - 1ull;
    for (int k = lane; k < d; k += 64)
    {
        const int y = nbr[o0 + k];
        rv[o0 + k] = C[(size_t)row * n + y];
        sel[o0 + k] = kNone32;
        // position of `row` in y's ascending list = neighbours of y below `row`: the word prefix written by
        // fill_nbr plus one popcount -- two independent loads instead of a binary search
        const int pos = wpre[(size_t)y * words + rw] + __popcll(adj[(size_t)y * words + rw] & below);
        meta[o0 + k] = make_int4(y, pos, off[y], off[y + 1] - off[y]);
    }
}

struct RowsParams
{
    const float *rv;
    const int4 *meta;
    unsigned *sel;   // level-1 selection state per CSR slot: lowest passing position (Skeleton) / 0 = edge gone (hetcor)
    int use_filter;  // 0: every test on the exact arithmetic (thresholds too small for the guard band)
    int has_ti;      // hetcor engine: a time index was given (else every index is 0 and the rule excludes nothing)
    float beta;      // half-width of the level-1 guard band on rho^2 (level1_beta)
    int exp;         // experiment bits: 2 = rows contiguous per XCD, 4 = count the atomics, 8 = non-temporal meta loads
    int shard_rank, shard_world;  // row-sharded runs: this engine streams the rows ya with ya % world == rank
};

constexpr int kRowsThreads = 256;
constexpr int kRowsChunk = 256;    // neighbours X of the row handled per staging round

// Half-width of the level-1 guard band on rho^2.  Unlike the deeper levels (Cholesky against SVD, conditioning-dependent:
// kBeta) the two forms of the level-1 test share their operands, so the band only has to cover rounding: the exact form's
// rho carries <= 4 roundings (2.4e-7 relative), its Fisher z (two correctly rounded logs of 1 +- rho) <= 5.6e-8 absolute,
// i.e. <= 5.6e-8 / t relative in rho at the decision point |rho| = t = tanh(th); the squared form adds <= 5 roundings
// (3e-7 relative on rho^2).  Sixteen times that sum, never more than kBeta: 7.6e-5 at the headline threshold
// (t = 0.0304) instead of 2e-3 -- with the wide band a quarter of all 256-test wave steps had a lane in the band and
// went through the exact form (division, two square roots, log), which was 40 % of the kernel's vector instructions.
inline float level1_beta(float t2)
{
    const double t = std::sqrt(std::max((double)t2, 1e-30));
    return (float)std::min((double)kBeta, 16.0 * (2.0 * (2.4e-7 + 5.6e-8 / t) + 3.0e-7));
}

// Level-1 test, rho = h01 / (sqrt|h00| sqrt|hc|) against th, in the squared form with the guard band of
// ci_fast.h (level1_beta wide).  It starts from the SAME fp32 h00, h01, hc as the reference's form (identical operations), so
// unlike the deeper levels no conditioning margin is needed: the two differ by a few ulp whatever the
// operands are, as long as they are positive.  Returns pass; sure = false (inside the band, operand <= 0, NaN,
// filter off) sends the lane to level1_exact.
__device__ __forceinline__ bool level1_filter(float h00, float h01, float hc, float t2, float beta, bool ok, bool &sure)
{
    const float lhs = h01 * h01;
    const float rhs = t2 * (h00 * hc);
    const bool pass = lhs < rhs * (1.0f - beta);
    const bool fail = lhs > rhs * (1.0f + beta);
    sure = ok && (h00 > 0.0f) && (pass || fail);
    return pass;
}

__device__ __forceinline__ bool level1_exact(float h00, float h01, float hc, float th)
{
    const float rho = h01 / (sqrtf(fabsf(h00)) * sqrtf(fabsf(hc)));
    return z_below<true>(rho, th);
}

// Work of a row: for every neighbour X (list position a of the row inside X's list) the segment of later
// positions b in (a, deg X).  The segments of one staging round are laid end to end (block scan, empty ones
// dropped) and every wave takes a contiguous quarter of that flat range, 64 entries per step, so lanes stay
// busy whatever the segment lengths are; a lane finds its segment by walking the LDS prefix array from the
// wave's current segment.  A step is a three-stage software pipeline over three operand sets:
//   stage A (step i+2): two coalesced loads (nbr and rv of X at b);
//   stage B (step i+1): C[row, yb] -- consecutive lanes hold ascending, mostly adjacent columns of ONE row of C,
//                       the row every wave of the workgroup reads, so it is served by L1/L2;
//   stage C (step i):   both tests through the branch-free filter, fire-and-forget minima.
template <typename T>
__device__ __forceinline__ T ld32(const T *base, unsigned idx)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (idx << 2));
}

struct RowsStep
{
    unsigned ia, ib;
    int a, b, X, first_lane;
    bool in;
    int yb;
    float ra, rb, c;
};

template <int MODE, bool VALIDATE>
__global__ void __launch_bounds__(kRowsThreads) level1_rows_kernel(SweepParams p, RowsParams rp)
{
    __shared__ int4 s_seg[kRowsChunk];  // {slot of (X, a), a, C[X, row] bits, X}
    __shared__ int s_pre[kRowsChunk + 1];
    __shared__ int s_wtot[2][kRowsThreads / 64];
    __shared__ unsigned long long s_cnt[4];
    const int n = p.n;
    int ya = blockIdx.x;
    if (rp.exp & 2)
    {  // workgroups are dealt to the 8 XCDs round robin: give every XCD one contiguous band of rows
        const int per = (n + 7) >> 3;
        ya = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (ya >= n) return;
    }
    if (!p.cnt->active) return;
    const int o0 = p.off[ya];
    const int d = p.off[ya + 1] - o0;
    if (d == 0 || ya + 1 >= n) return;
    if (rp.shard_world > 1 && ya % rp.shard_world != rp.shard_rank) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kRowsThreads >> 6;
    if (tid < 4) s_cnt[tid] = 0ull;
    int4 m = make_int4(0, 0, 0, 0);
    float mra = 0.0f;
    if (tid < d)
    {
        // read once per (row, X): streamed past the caches (exp bit 8) so that the X lists every row of the band re-reads
        // keep their place in L2
        if (rp.exp & 8)
        {
            const int4 *mp = rp.meta + o0 + tid;
            m.x = __builtin_nontemporal_load(&mp->x);
            m.y = __builtin_nontemporal_load(&mp->y);
            m.z = __builtin_nontemporal_load(&mp->z);
            m.w = __builtin_nontemporal_load(&mp->w);
        }
        else
            m = rp.meta[o0 + tid];
        mra = rp.rv[m.z + m.y];  // C[X, row]
    }
    const float *crow = p.C + (size_t)ya * n;
    [[maybe_unused]] int tiA = 0;
    if constexpr (MODE == 1) tiA = p.time_index[ya];
    const bool use_filter = rp.use_filter != 0;
    const float th = p.th, t2 = p.t2, beta = rp.beta;
    unsigned ntests = 0, nrem = 0, viol = 0;
    for (int kc = 0; kc < d; kc += kRowsChunk)
    {
        // ---- lay the non-empty segments of this round end to end ----
        const int len = (kc + tid < d) ? max(0, m.w - m.y - 1) : 0;
        int pos = (len > 0) ? 1 : 0, pre = len;
        for (int o = 1; o < 64; o <<= 1)
        {
            const int v1 = __shfl_up(pos, o), v2 = __shfl_up(pre, o);
            if (lane >= o)
            {
                pos += v1;
                pre += v2;
            }
        }
        if (kc > 0) __syncthreads();  // the previous round's readers are done with s_seg / s_pre / s_wtot
        if (lane == 63)
        {
            s_wtot[0][wave] = pos;
            s_wtot[1][wave] = pre;
        }
        __syncthreads();
        int nseg = 0, total = 0, pos0 = 0, pre0 = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++)
        {
            const int c1 = s_wtot[0][w], c2 = s_wtot[1][w];
            if (w < wave)
            {
                pos0 += c1;
                pre0 += c2;
            }
            nseg += c1;
            total += c2;
        }
        if (len > 0)
        {
            s_seg[pos0 + pos - 1] = make_int4(m.z + m.y, m.y, __float_as_int(mra), m.x);
            s_pre[pos0 + pos - 1] = pre0 + pre - len;
        }
        if (tid == 0) s_pre[nseg] = total;
        if (kc + kRowsChunk + tid < d)
        {  // next round, in flight meanwhile
            m = rp.meta[o0 + kc + kRowsChunk + tid];
            mra = rp.rv[m.z + m.y];
        }
        __syncthreads();
        // ---- this wave's contiguous share of the flat range ----
        const int per = ((total + kWaves * 64 - 1) / (kWaves * 64)) * 64;
        const int f_begin = wave * per, f_end = min(total, f_begin + per);
        if (f_begin >= f_end) continue;
        int kw = 0;
        {
            int hi = nseg;  // largest kw with s_pre[kw] <= f_begin
            while (hi - kw > 1)
            {
                const int mid = (kw + hi) >> 1;
                if (s_pre[mid] <= f_begin)
                    kw = mid;
                else
                    hi = mid;
            }
        }
        // stage A: locate the lanes of a step inside the segments and request the per-slot operands.  Lanes past
        // the end of the wave's range idle on slot a of the last segment, so every request is unconditional (a
        // conditional one would make the compiler drain ALL outstanding loads at the join).
        auto stage_a = [&](int base, RowsStep &st) {
            const int f = base + lane;
            st.in = f < f_end;
            int kk = kw;
            if (st.in)
                while (s_pre[kk + 1] <= f) kk++;
            kw = __shfl(kk, 63);
            const int seg0 = s_pre[kk];
            const int4 e = s_seg[kk];
            st.a = e.y;
            st.b = e.y + 1 + (f - seg0);
            st.ra = __int_as_float(e.z);
            st.X = e.w;
            st.first_lane = max(0, seg0 - base);
            st.ia = (unsigned)e.x;
            st.ib = st.in ? (unsigned)(e.x + 1 + (f - seg0)) : st.ia;
            // 32-bit byte offsets off uniform bases (the CSR arrays stay below 4 GB): scalar-base addressing, no
            // 64-bit vector address arithmetic
            st.yb = ld32<int>(p.nbr, st.ib);
            st.rb = ld32<float>(rp.rv, st.ib);
            // No selection-state reads: every pair is evaluated.  The minima do not depend on them, the reads were a third
            // of the kernel's list traffic, and a state that other XCDs update past this XCD's L2 skipped little
            // (measured: 0.49 -> 0.40 ms on the 10k block).
        };
        auto stage_b = [&](RowsStep &st) { st.c = ld32<float>(crow, (unsigned)st.yb); };
        auto stage_c = [&](const RowsStep &cur) {
            const int yb = cur.yb, a = cur.a, b = cur.b;
            const float ra = cur.ra, rb = cur.rb, c = cur.c;
            const bool act = cur.in;
            bool needA, needB;  // A: Y = ya, S = yb ; B: Y = yb, S = ya
            if constexpr (MODE == 0)
            {
                needA = act;
                needB = act;
            }
            else if (rp.has_ti)
            {
                const int tiX = p.time_index[cur.X], tiB = p.time_index[yb];
                needA = act && !(tiB > max(tiX, tiA));
                needB = act && !(tiA > max(tiX, tiB));
            }
            else
            {
                needA = act;
                needB = act;
            }
            const float hc = 1.0f - (c * c);
            const float h00a = 1.0f - (rb * rb), h01a = ra - (rb * c);
            const float h00b = 1.0f - (ra * ra), h01b = rb - (ra * c);
            const bool ok = use_filter && (hc > 0.0f);
            bool sureA, sureB;
            bool passA = level1_filter(h00a, h01a, hc, t2, beta, ok, sureA);
            bool passB = level1_filter(h00b, h01b, hc, t2, beta, ok, sureB);
            ntests += (needA ? 1u : 0u) + (needB ? 1u : 0u);
            const bool slowA = needA && (VALIDATE || !sureA), slowB = needB && (VALIDATE || !sureB);
            if (__ballot(slowA || slowB) != 0ull)
            {  // rare: the reference's operation order
                if (slowA)
                {
                    const bool ex = level1_exact(h00a, h01a, hc, th);
                    if (VALIDATE && sureA && ex != passA) viol++;
                    passA = ex;
                }
                if (slowB)
                {
                    const bool ex = level1_exact(h00b, h01b, hc, th);
                    if (VALIDATE && sureB && ex != passB) viol++;
                    passB = ex;
                }
            }
            passA = passA && needA;
            passB = passB && needB;
            // Y = ya: positions ascend with the lane inside a segment, so the lowest passing lane of a segment
            // carries the segment's minimum; only that lane speaks
            const unsigned long long pa = __ballot(passA);
            const unsigned long long below = pa & ((1ull << lane) - 1ull) & ~((1ull << cur.first_lane) - 1ull);
            const bool headA = passA && (below == 0ull);
            if (rp.exp & 4) nrem += (passB ? 1u : 0u) + (headA ? 1u : 0u);
            if constexpr (MODE == 0)
            {
                // fire-and-forget minima: nobody waits for the L2 round trip; which slots got a separating
                // set is counted once afterwards (level1_count_kernel)
                if (passB)
                    (void)__hip_atomic_fetch_min(&rp.sel[cur.ib], (unsigned)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (headA)
                    (void)__hip_atomic_fetch_min(&rp.sel[cur.ia], (unsigned)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            else
            {
                // hetcor: the edge goes in both directions; both directed slots are marked with plain stores (racing
                // writers store the same 0) and the bitmap / degrees are updated once afterwards (level1_apply_kernel),
                // so the sweep carries no returning atomics
                if (passB) rp.sel[cur.ib] = 0u;
                if (headA) rp.sel[cur.ia] = 0u;
            }
        };
        // three operand sets rotate through the stages (no register copies, so the compiler can wait for exactly
        // the set it needs); the scheduling barriers keep the requests ahead of the evaluation
        RowsStep s0, s1, s2;
        stage_a(f_begin, s0);
        stage_a(f_begin + 64, s1);
        stage_b(s0);
        for (int base = f_begin;; base += 192)
        {
            stage_a(base + 128, s2);
            stage_b(s1);
            __builtin_amdgcn_sched_barrier(0);
            stage_c(s0);
            if (base + 64 >= f_end) break;
            stage_a(base + 192, s0);
            stage_b(s2);
            __builtin_amdgcn_sched_barrier(0);
            stage_c(s1);
            if (base + 128 >= f_end) break;
            stage_a(base + 256, s1);
            stage_b(s0);
            __builtin_amdgcn_sched_barrier(0);
            stage_c(s2);
            if (base + 192 >= f_end) break;
        }
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        ntests += __shfl_xor(ntests, o);  // < 2^32 per wave: 64 lanes x (work of a quarter row)
        nrem += __shfl_xor(nrem, o);
        if (VALIDATE) viol += __shfl_xor(viol, o);
    }
    if (lane == 0)
    {
        if (ntests) atomicAdd(&s_cnt[0], (unsigned long long)ntests);
        if (nrem) atomicAdd(&s_cnt[2], (unsigned long long)nrem);
        if (VALIDATE && viol) atomicAdd(&s_cnt[3], (unsigned long long)viol);
    }
    __syncthreads();
    if (tid == 0)
    {
        unsigned long long *sl = p.slots + (size_t)(blockIdx.x & (kCounterSlots - 1)) * 4;
        if (s_cnt[0]) atomicAdd(&sl[0], s_cnt[0]);
        if (s_cnt[2]) atomicAdd(&sl[2], s_cnt[2]);
        if (s_cnt[3]) atomicAdd(&sl[3], s_cnt[3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same sweep with TWO consecutive list positions per lane (round 2).  The one-position form above is bound by
// vector-instruction issue, not by memory (136 vector instructions per 64 pairs, 67 % of the SIMDs' issue cycles at
// 0.40 ms on the 10k block; staging the row of C in LDS instead of gathering it through L1/L2 made it slower), and
// two thirds of those instructions are bookkeeping of the flattened iteration space that does not depend on how many
// positions a lane carries: the walk over the segment prefix array, slot arithmetic, segment-head detection, loop
// control.  Here the flat range counts PAIRS of positions (b, b + 1) of a segment (odd segments are padded by one masked
// position, so a pair never straddles two segments); a lane requests both list entries with one 8-byte load each
// (nbr, rv: 4-byte aligned dwordx2), gathers two elements of the row of C, and evaluates its four tests as float2
// pairs (v_pk_mul_f32 / v_pk_add_f32): the same fp32 operations in the same order per element as level1_filter, so
// verdicts, rechecks and minima are identical to the one-position form (option l1_exp bit 1 selects that form).
#ifndef CUSK_ROWS_SETS
#define CUSK_ROWS_SETS 5
#endif
typedef float rows_f2 __attribute__((ext_vector_type(2)));
typedef float rows_f4 __attribute__((ext_vector_type(4)));
typedef int rows_i2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float rows_f2u __attribute__((ext_vector_type(2), aligned(4)));

struct RowsStep2
{
    unsigned ia, ib;  // slots of (X, a) and of (X, b)
    int a, b, X, first_lane;
    bool in0, in1;  // position b / b + 1 is a real position of this wave's range
    rows_i2u nb;    // list entries at b, b + 1 (the second one is only meaningful when in1)
    rows_f2u rbu;
    int yb0, yb1;
    float ra;
    rows_f2 c;
};

// LDSROW: the columns [ya, n) of the workgroup's row of C are staged in LDS once (coalesced 16-byte loads) and the
// C[ya, yb] gathers of stage B become LDS reads: with the gathers going through L1/L2 the rows of all resident workgroups
// (40 KB each at n = 10,020, five per CU) evict each other and the per-edge streams -- 2.3 GB of L2 fills per launch for
// a 401 MB matrix, which is what bounds the kernel once the instruction count is down (37 % vector issue).  THREADS grows
// with n so that the CU keeps its waves when fewer rows fit into its 160 KB (launch_level1_rows).
template <int MODE, bool VALIDATE, int THREADS, bool LDSROW>
__global__ void __launch_bounds__(THREADS) level1_rows2_kernel(SweepParams p, RowsParams rp)
{
    constexpr int kRowsThreads = THREADS, kRowsChunk = THREADS;
    __shared__ int4 s_seg[kRowsChunk];  // {slot of (X, a), a, C[X, row] bits, X}
    __shared__ int s_dx[kRowsChunk];    // degree of X
    __shared__ int s_pre[kRowsChunk + 1];
    __shared__ int s_wtot[2][kRowsThreads / 64];
    __shared__ unsigned long long s_cnt[4];
    extern __shared__ __attribute__((aligned(16))) float s_row[];  // LDSROW: s_row[col + sh] = C[ya, col], col >= ya
    const int n = p.n;
    // (Rows to XCDs, tried in round 3: XCD x takes the x-th and (15 - x)-th sixteenth of the rows instead of every eighth
    // row, so that an L2 keeps re-reading the lists of one region: 0.387 ms against 0.255.  Consecutive rows running at the
    // same time on all XCDs already share their lists in time, and whole regions per XCD unbalance dense and sparse ones.)
    const int ya = blockIdx.x;
    if (ya + 1 >= n) return;
    if (rp.shard_world > 1 && ya % rp.shard_world != rp.shard_rank) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kRowsThreads >> 6;
    const float *crow = p.C + (size_t)ya * n;
    [[maybe_unused]] int sh = 0;
    // LDS index = column + sh with sh = (element offset of the row) mod 4: 16-byte aligned global loads land on
    // 16-byte aligned LDS addresses whatever n is.  Only columns >= ya are ever asked for (yb follows ya in an
    // ascending list; idle lanes ask for ya itself).  The rounded-down head and rounded-up tail read at most three
    // elements of the neighbouring rows (ya >= 1 whenever the head reaches back, ya <= n - 2 always).
    // Batched runs stage the columns of the row's own block only, [ya, hi), at LDS index column - lo + sh0 (lo, the
    // block's base, is a multiple of 64 and keeps the alignment).
    // Every request is unconditional (a start past the end is clamped to the last 16-byte piece of the range: those lanes
    // repeat that piece, load and store, same bytes to the same place) and the four pieces of a pass live in four named
    // registers: with `if (iu < i_end) v[u] = ...` on an array the compiler kept the array in scratch memory and waited
    // for each load before the next (rounds 2 and 3a: 0.31 ms for this kernel instead of 0.27).  Non-temporal: the row
    // is read once, by this workgroup only, and should not push the neighbour lists and the selection words of the other
    // rows out of the L2.  The first pass (all of the row up to 16 x THREADS columns) is requested right behind the
    // per-neighbour records and before the gather that depends on them: the row streams in beside those two round trips.
    constexpr int kStep = THREADS * 4;
    [[maybe_unused]] const float *gbase = nullptr;
    [[maybe_unused]] int i_first = 0, i_end = 0, i_last = 0, i0 = 0, i1 = 0, i2 = 0, i3 = 0;
    [[maybe_unused]] rows_f4 v0, v1, v2, v3;
    // (the activity flag and the list bounds in ONE scalar round trip: left alone, the compiler waits for the flag before it
    // asks for the bounds)
    const int act = p.cnt->active;
    const int o0 = p.off[ya], o1 = p.off[ya + 1];
    asm volatile("" ::"s"(act), "s"(o0), "s"(o1));
    if (!act) return;
    const int d = o1 - o0;
    if (d == 0) return;
    if (tid < 4) s_cnt[tid] = 0ull;
    int4 m = make_int4(0, 0, 0, 0);
    float mra = 0.0f;
    if (tid < d) m = rp.meta[o0 + tid];
    if constexpr (LDSROW)
    {
        const size_t g0 = (size_t)ya * n;
        const int sh0 = (int)(g0 & 3);
        int lo = 0, hi = n;
        if (p.row_range)
        {
            const int2 rg = p.row_range[ya];
            lo = rg.x;
            hi = rg.y;
        }
        sh = sh0 - lo;
        gbase = p.C + (g0 - sh0) + lo;  // 16-byte aligned (p.C is: checked by the launcher)
        i_end = hi - lo + sh0;
        i_last = (i_end - 1) & ~3;
        i_first = ((ya - lo + sh0) & ~3) + tid * 4;
        i0 = min(i_first, i_last), i1 = min(i_first + kStep, i_last), i2 = min(i_first + 2 * kStep, i_last), i3 = min(i_first + 3 * kStep, i_last);
        v0 = __builtin_nontemporal_load(reinterpret_cast<const rows_f4 *>(gbase + i0));
        v1 = __builtin_nontemporal_load(reinterpret_cast<const rows_f4 *>(gbase + i1));
        v2 = __builtin_nontemporal_load(reinterpret_cast<const rows_f4 *>(gbase + i2));
        v3 = __builtin_nontemporal_load(reinterpret_cast<const rows_f4 *>(gbase + i3));
        asm volatile("" ::: "memory");  // the row's requests stay between the record load and its dependent gather
    }
    if (tid < d) mra = rp.rv[m.z + m.y];  // C[X, row]
    if constexpr (LDSROW)
    {
        *reinterpret_cast<rows_f4 *>(s_row + i0) = v0;
        *reinterpret_cast<rows_f4 *>(s_row + i1) = v1;
        *reinterpret_cast<rows_f4 *>(s_row + i2) = v2;
        *reinterpret_cast<rows_f4 *>(s_row + i3) = v3;
        for (int i = i_first + kStep * 4; i < i_end; i += kStep * 4)
        {
            i0 = min(i, i_last), i1 = min(i + kStep, i_last), i2 = min(i + 2 * kStep, i_last), i3 = min(i + 3 * kStep, i_last);
            v0 = __builtin_nontemporal_load(reinterpret_cast<const rows_f4 *>(gbase + i0));
            v1 = __builtin_nontemporal_load(reinterpret_cast<const rows_f4 *>(gbase + i1));
            v2 = __builtin_nontemporal_load(reinterpret_cast<const rows_f4 *>(gbase + i2));
            v3 = __builtin_nontemporal_load(reinterpret_cast<const rows_f4 *>(gbase + i3));
            *reinterpret_cast<rows_f4 *>(s_row + i0) = v0;
            *reinterpret_cast<rows_f4 *>(s_row + i1) = v1;
            *reinterpret_cast<rows_f4 *>(s_row + i2) = v2;
            *reinterpret_cast<rows_f4 *>(s_row + i3) = v3;
        }
        // visible to every wave after the first barrier of the staging round below
    }
    [[maybe_unused]] int tiA = 0;
    if constexpr (MODE == 1) tiA = p.time_index[ya];
    const bool use_filter = rp.use_filter != 0;
    const float th = p.th, t2 = p.t2, beta = rp.beta;
    unsigned ntests = 0, nrem = 0, viol = 0;
    const bool count_by_segment = (MODE == 0) || !rp.has_ti;
    for (int kc = 0; kc < d; kc += kRowsChunk)
    {
        // ---- lay the non-empty segments of this round end to end, counted in pairs of positions ----
        const int len = (kc + tid < d) ? max(0, m.w - m.y - 1) : 0;
        const int plen = (len + 1) >> 1;
        // executed tests: without a time-index rule every position of a segment is tested in both directions -- counted
        // here once per segment instead of four flag additions per lane and step
        if (count_by_segment) ntests += 2u * (unsigned)len;
        int pos = (plen > 0) ? 1 : 0, pre = plen;
        for (int o = 1; o < 64; o <<= 1)
        {
            const int v1 = __shfl_up(pos, o), v2 = __shfl_up(pre, o);
            if (lane >= o)
            {
                pos += v1;
                pre += v2;
            }
        }
        if (kc > 0) __syncthreads();  // the previous round's readers are done with s_seg / s_pre / s_wtot
        if (lane == 63)
        {
            s_wtot[0][wave] = pos;
            s_wtot[1][wave] = pre;
        }
        __syncthreads();
        int nseg = 0, total = 0, pos0 = 0, pre0 = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++)
        {
            const int c1 = s_wtot[0][w], c2 = s_wtot[1][w];
            if (w < wave)
            {
                pos0 += c1;
                pre0 += c2;
            }
            nseg += c1;
            total += c2;
        }
        if (plen > 0)
        {
            s_seg[pos0 + pos - 1] = make_int4(m.z + m.y, m.y, __float_as_int(mra), m.x);
            s_dx[pos0 + pos - 1] = m.w;
            s_pre[pos0 + pos - 1] = pre0 + pre - plen;
        }
        if (tid == 0) s_pre[nseg] = total;
        if (kc + kRowsChunk + tid < d)
        {  // next round, in flight meanwhile
            m = rp.meta[o0 + kc + kRowsChunk + tid];
            mra = rp.rv[m.z + m.y];
        }
        __syncthreads();
        // ---- this wave's contiguous share of the flat range ----
        const int per = ((total + kWaves * 64 - 1) / (kWaves * 64)) * 64;
        const int f_begin = wave * per, f_end = min(total, f_begin + per);
        if (f_begin >= f_end) continue;
        int kw = 0;
        {
            int hi = nseg;  // largest kw with s_pre[kw] <= f_begin
            while (hi - kw > 1)
            {
                const int mid = (kw + hi) >> 1;
                if (s_pre[mid] <= f_begin)
                    kw = mid;
                else
                    hi = mid;
            }
        }
        // stage A: locate the lanes of a step inside the segments and request the per-slot operands.  Lanes past the
        // end of the wave's range idle on slot a of the last segment (their "neighbour" is the row itself), so every
        // request is unconditional.
        auto stage_a = [&](int base, RowsStep2 &st) {
            const int f = base + lane;
            st.in0 = f < f_end;
            int kk = kw;
            if (st.in0)
                while (s_pre[kk + 1] <= f) kk++;
            kw = __shfl(kk, 63);
            const int seg0 = s_pre[kk];
            const int4 e = s_seg[kk];
            const int dX = s_dx[kk];
            const int r = st.in0 ? 2 * (f - seg0) + 1 : 0;
            st.a = e.y;
            st.b = e.y + r;
            st.in1 = st.in0 && (st.b + 1 < dX);
            st.ra = __int_as_float(e.z);
            st.X = e.w;
            st.first_lane = max(0, seg0 - base);
            st.ia = (unsigned)e.x;
            st.ib = (unsigned)(e.x + r);
            st.nb = *reinterpret_cast<const rows_i2u *>(reinterpret_cast<const char *>(p.nbr) + (st.ib << 2));
            st.rbu = *reinterpret_cast<const rows_f2u *>(reinterpret_cast<const char *>(rp.rv) + (st.ib << 2));
        };
        auto stage_b = [&](RowsStep2 &st) {
            st.yb0 = st.nb.x;
            st.yb1 = st.in1 ? st.nb.y : st.nb.x;  // the entry behind an odd segment belongs to another list
            if constexpr (LDSROW)
            {
                st.c.x = s_row[st.yb0 + sh];
                st.c.y = s_row[st.yb1 + sh];
            }
            else
            {
                st.c.x = ld32<float>(crow, (unsigned)st.yb0);
                st.c.y = ld32<float>(crow, (unsigned)st.yb1);
            }
        };
        auto stage_c = [&](const RowsStep2 &cur) {
            const int a = cur.a, b = cur.b;
            const float ra = cur.ra;
            const rows_f2 c = cur.c;
            const rows_f2 rb = {cur.rbu.x, cur.rbu.y};
            bool needA0, needB0, needA1, needB1;  // A: Y = ya, S = yb ; B: Y = yb, S = ya
            if constexpr (MODE == 0)
            {
                needA0 = needB0 = cur.in0;
                needA1 = needB1 = cur.in1;
            }
            else if (rp.has_ti)
            {
                const int tiX = p.time_index[cur.X], tiB0 = p.time_index[cur.yb0], tiB1 = p.time_index[cur.yb1];
                needA0 = cur.in0 && !(tiB0 > max(tiX, tiA));
                needB0 = cur.in0 && !(tiA > max(tiX, tiB0));
                needA1 = cur.in1 && !(tiB1 > max(tiX, tiA));
                needB1 = cur.in1 && !(tiA > max(tiX, tiB1));
            }
            else
            {  // no time index given (all equal): the rule excludes nothing, and its three gathers per lane and step go
                needA0 = needB0 = cur.in0;
                needA1 = needB1 = cur.in1;
            }
            // per element exactly the operations of the one-position form: 1 - (c c), ra - (rb c), t2 ((h00) (hc)), ...
            const rows_f2 one = {1.0f, 1.0f}, rav = {ra, ra};
            const rows_f2 hc = one - (c * c);
            const rows_f2 h00a = one - (rb * rb), h01a = rav - (rb * c);
            const float h00b = 1.0f - (ra * ra);
            const rows_f2 h00bv = {h00b, h00b}, h01b = rb - (rav * c);
            const rows_f2 lhsA = h01a * h01a, rhsA = t2 * (h00a * hc);
            const rows_f2 lhsB = h01b * h01b, rhsB = t2 * (h00bv * hc);
            const rows_f2 loA = rhsA * (1.0f - beta), hiA = rhsA * (1.0f + beta);
            const rows_f2 loB = rhsB * (1.0f - beta), hiB = rhsB * (1.0f + beta);
            const bool ok0 = use_filter && (hc.x > 0.0f), ok1 = use_filter && (hc.y > 0.0f), okb = h00b > 0.0f;
            bool passA0 = lhsA.x < loA.x, passA1 = lhsA.y < loA.y, passB0 = lhsB.x < loB.x, passB1 = lhsB.y < loB.y;
            const bool sureA0 = ok0 && (h00a.x > 0.0f) && (passA0 || lhsA.x > hiA.x);
            const bool sureA1 = ok1 && (h00a.y > 0.0f) && (passA1 || lhsA.y > hiA.y);
            const bool sureB0 = ok0 && okb && (passB0 || lhsB.x > hiB.x);
            const bool sureB1 = ok1 && okb && (passB1 || lhsB.y > hiB.y);
            if (!count_by_segment) ntests += (needA0 ? 1u : 0u) + (needB0 ? 1u : 0u) + (needA1 ? 1u : 0u) + (needB1 ? 1u : 0u);
            const bool slowA0 = needA0 && (VALIDATE || !sureA0), slowB0 = needB0 && (VALIDATE || !sureB0);
            const bool slowA1 = needA1 && (VALIDATE || !sureA1), slowB1 = needB1 && (VALIDATE || !sureB1);
            if (__ballot(slowA0 || slowB0 || slowA1 || slowB1) != 0ull)
            {  // rare: the reference's operation order
                if (slowA0)
                {
                    const bool ex = level1_exact(h00a.x, h01a.x, hc.x, th);
                    if (VALIDATE && sureA0 && ex != passA0) viol++;
                    passA0 = ex;
                }
                if (slowB0)
                {
                    const bool ex = level1_exact(h00b, h01b.x, hc.x, th);
                    if (VALIDATE && sureB0 && ex != passB0) viol++;
                    passB0 = ex;
                }
                if (slowA1)
                {
                    const bool ex = level1_exact(h00a.y, h01a.y, hc.y, th);
                    if (VALIDATE && sureA1 && ex != passA1) viol++;
                    passA1 = ex;
                }
                if (slowB1)
                {
                    const bool ex = level1_exact(h00b, h01b.y, hc.y, th);
                    if (VALIDATE && sureB1 && ex != passB1) viol++;
                    passB1 = ex;
                }
            }
            passA0 = passA0 && needA0;
            passB0 = passB0 && needB0;
            passA1 = passA1 && needA1;
            passB1 = passB1 && needB1;
            // Y = ya: positions ascend with the lane inside a segment (and b before b + 1 inside a lane), so the lowest
            // passing lane of a segment carries the segment's minimum; only that lane speaks
            const bool anyA = passA0 || passA1;
            const unsigned long long pa = __ballot(anyA);
            const unsigned long long below = pa & ((1ull << lane) - 1ull) & ~((1ull << cur.first_lane) - 1ull);
            const bool headA = anyA && (below == 0ull);
            const int bsel = passA0 ? b : b + 1;
            if (rp.exp & 4) nrem += (passB0 ? 1u : 0u) + (passB1 ? 1u : 0u) + (headA ? 1u : 0u);
            if constexpr (MODE == 0)
            {
                // fire-and-forget minima: nobody waits for the L2 round trip; which slots got a separating set is
                // counted once afterwards
                if (passB0) (void)__hip_atomic_fetch_min(&rp.sel[cur.ib], (unsigned)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (passB1) (void)__hip_atomic_fetch_min(&rp.sel[cur.ib + 1], (unsigned)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (headA) (void)__hip_atomic_fetch_min(&rp.sel[cur.ia], (unsigned)bsel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            else
            {
                // hetcor: the edge goes in both directions; only the slot of the ordered pair that was tested is marked
                // (level1_apply_kernel removes the edge when either direction is marked, updates bitmap and degrees once).
                // The mark is the lowest passing position, as in Skeleton mode: anything but kNone32 means "gone", and the
                // position lets level1_apply_kernel count the tests of the canonical schedule (round 2 stored plain zeros)
                if (passB0) (void)__hip_atomic_fetch_min(&rp.sel[cur.ib], (unsigned)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (passB1) (void)__hip_atomic_fetch_min(&rp.sel[cur.ib + 1], (unsigned)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (headA) (void)__hip_atomic_fetch_min(&rp.sel[cur.ia], (unsigned)bsel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        // Operand sets rotate through the stages (no register copies: every index below is a compile-time constant after
        // unrolling, so the compiler can wait for exactly the set it needs); the scheduling barriers keep the requests
        // ahead of the evaluation.  Round 3: kSets sets, the list entries requested kSets - 1 steps ahead of their
        // evaluation (rounds 1-2: three sets, two steps ahead -- a step's ~90 vector instructions cover 300 ns, the entries
        // take longer than two of those to arrive from L2, and the three waves of a SIMD, all the LDS rows allow, do not
        // cover the rest: 0.254 ms; five sets 0.242; launch_level1_rows has the other depths).
        constexpr int kSets = CUSK_ROWS_SETS;
        RowsStep2 st[kSets];
#pragma unroll
        for (int j = 0; j < kSets - 1; j++) stage_a(f_begin + 64 * j, st[j]);
        stage_b(st[0]);
        for (int base = f_begin;; base += 64 * kSets)
        {
            bool done = false;
#pragma unroll
            for (int j = 0; j < kSets; j++)
            {
                stage_a(base + 64 * (kSets - 1 + j), st[(kSets - 1 + j) % kSets]);
                stage_b(st[(j + 1) % kSets]);
                __builtin_amdgcn_sched_barrier(0);
                stage_c(st[j]);
                if (base + 64 * (j + 1) >= f_end)
                {
                    done = true;
                    break;
                }
            }
            if (done) break;
        }
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        ntests += __shfl_xor(ntests, o);  // < 2^32 per wave
        nrem += __shfl_xor(nrem, o);
        if (VALIDATE) viol += __shfl_xor(viol, o);
    }
    if (lane == 0)
    {
        if (ntests) atomicAdd(&s_cnt[0], (unsigned long long)ntests);
        if (nrem) atomicAdd(&s_cnt[2], (unsigned long long)nrem);
        if (VALIDATE && viol) atomicAdd(&s_cnt[3], (unsigned long long)viol);
    }
    __syncthreads();
    if (tid == 0)
    {
        unsigned long long *sl = p.slots + (size_t)(blockIdx.x & (kCounterSlots - 1)) * 4;
        if (s_cnt[0]) atomicAdd(&sl[0], s_cnt[0]);
        if (s_cnt[2]) atomicAdd(&sl[2], s_cnt[2]);
        if (s_cnt[3]) atomicAdd(&sl[3], s_cnt[3]);
    }
}

// hetcor mode, after the sweep: every row drops the neighbours whose slot was marked (the wave owns its bitmap row:
// word-aggregated plain read-modify-writes as in gather_records), sets its degree, and the removed directed edges
// are counted
// meta != nullptr (level 1 behind the row-streaming kernel): a slot is marked by the test of ITS ordered pair only, so the
// slot of the reverse pair is looked at as well (the edge goes when either direction found a separating variable)
__global__ void __launch_bounds__(256) level1_apply_kernel(const int *__restrict__ off, const int *__restrict__ nbr,
                                                           const unsigned *__restrict__ sel, unsigned long long *adj, int *deg,
                                                           int n, int words, unsigned long long *slots, const LevelCounters *cnt,
                                                           const int4 *__restrict__ meta, unsigned long long *canon)
{
    __shared__ int s_sum[4];
    __shared__ unsigned long long s_can[4];
    if (!cnt->active) return;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int removed = 0;
    unsigned long long ctests = 0;
    if (row < n)
    {
        const int o0 = off[row], d = off[row + 1] - o0;
        for (int k0 = 0; k0 < d; k0 += 64)
        {
            const int k = k0 + lane;
            const bool valid = k < d;
            const int Y = valid ? nbr[o0 + k] : 0;
            const unsigned mark = valid ? sel[o0 + k] : kNone32;
            bool gone = valid && (mark != kNone32);
            // canonical (sequential) schedule of this ordered pair at level 1: the neighbour at position k is tested with the
            // positions 0, 1, ... (without k itself) up to its lowest passing one, or with all d - 1 of them
            if (canon != nullptr && valid)
                ctests += gone ? (unsigned long long)(mark + 1u - (mark > (unsigned)k ? 1u : 0u)) : (unsigned long long)(d - 1);
            if (valid && !gone && meta != nullptr)
            {
                const int4 m = meta[o0 + k];
                gone = (sel[m.z + m.y] != kNone32);
            }
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
            if (valid && bits != 0ull && (lane == 0 || pw != w)) adj[(size_t)row * words + w] &= ~bits;
            removed += __popcll(__ballot(gone));
        }
        if (lane == 0 && removed) deg[row] = d - removed;
    }
    if (canon != nullptr)
        for (int o = 32; o > 0; o >>= 1) ctests += __shfl_xor(ctests, o);
    if (lane == 0)
    {
        s_sum[threadIdx.x >> 6] = removed;
        s_can[threadIdx.x >> 6] = ctests;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const int t = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        // cusk_stats.removed: ordered pairs, as in Skeleton mode (both directions of an edge go)
        if (t && slots) atomicAdd(&slots[(size_t)(blockIdx.x & (kCounterSlots - 1)) * 4 + 2], (unsigned long long)t);
        const unsigned long long c = s_can[0] + s_can[1] + s_can[2] + s_can[3];
        if (c && canon) atomicAdd(&canon[blockIdx.x & (kCounterSlots - 1)], c);
    }
}

// hetcor mode, row-sharded runs: the sweeps of levels that do not use the row-streaming kernel clear adjacency bits
// directly; turn the bitmap back into per-slot marks (0 = the edge is gone, all ones = alive) so that the engines can
// join them with the same unsigned MIN as the Skeleton engine's selection state
__global__ void __launch_bounds__(256) marks_from_bitmap_kernel(const int *__restrict__ off, const int *__restrict__ nbr,
                                                                const unsigned long long *__restrict__ adj, unsigned *sel, int n,
                                                                int words, const LevelCounters *cnt)
{
    if (!cnt->active) return;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const int o0 = off[row], d = off[row + 1] - o0;
    for (int k = lane; k < d; k += 64)
    {
        const int Y = nbr[o0 + k];
        const bool alive = (adj[(size_t)row * words + (Y >> 6)] >> (Y & 63)) & 1ull;
        sel[o0 + k] = alive ? kNone32 : 0u;
    }
}

hipError_t launch_marks_from_bitmap(const SweepParams &p, unsigned *sel, hipStream_t st)
{
    hipLaunchKernelGGL(marks_from_bitmap_kernel, dim3((p.n + 3) / 4), dim3(256), 0, st, p.off, p.nbr, p.adj, sel, p.n, p.words, p.cnt);
    return hipGetLastError();
}

hipError_t launch_level1_apply(const SweepParams &p, const unsigned *sel, const void *meta, bool count_removed, hipStream_t st)
{
    hipLaunchKernelGGL(level1_apply_kernel, dim3((p.n + 3) / 4), dim3(256), 0, st, p.off, p.nbr, sel, p.adj, p.deg, p.n, p.words,
                       count_removed ? p.slots : nullptr, p.cnt, static_cast<const int4 *>(meta), (unsigned long long *)nullptr);
    return hipGetLastError();
}

hipError_t launch_level1_rows(int mode, bool validate, bool use_filter, const SweepParams &p, float *rv, void *meta,
                              unsigned *sel, const int *wpre, hipEvent_t ev_begin, hipEvent_t ev_end, int shard_rank,
                              int shard_world, int exp, bool defer_apply, bool has_ti, unsigned long long *canon, hipStream_t st)
{
    const int n = p.n;
    hipLaunchKernelGGL(level1_prep_kernel, dim3((n + 3) / 4), dim3(256), 0, st, p.C, p.off, p.nbr, p.adj, wpre, p.words, rv,
                       static_cast<int4 *>(meta), sel, n, p.cnt);
    RowsParams rp;
    rp.rv = rv;
    rp.meta = static_cast<const int4 *>(meta);
    rp.sel = sel;
    rp.use_filter = use_filter ? 1 : 0;
    rp.has_ti = has_ti ? 1 : 0;
    rp.beta = (exp & 512) ? kBeta : level1_beta(p.t2);  // bit 512: the wide band of the deeper levels
    rp.exp = exp;
    rp.shard_rank = shard_rank;
    rp.shard_world = shard_world;
    const bool two = !(exp & 1) && !(exp & 2);  // two list positions per lane (default); bit 1: the one-position form
    const dim3 grid((unsigned)((exp & 2) ? ((n + 7) / 8) * 8 : n));
    const dim3 blk(kRowsThreads);
    if (ev_begin) (void)hipEventRecord(ev_begin, st);
    // Row of C in LDS (4 (n + 8) bytes per workgroup).  Of 256 / 512 threads the size that puts most ROWS on a CU (at most
    // 16 waves, 12 for the hetcor / validating forms: 125 / 131-137 VGPRs with five operand sets), the larger one on a tie: a row's prologue is a chain of four dependent round trips during which
    // its waves have nothing to do, and only other rows on the CU fill that time.  When two rows do not fit (n > ~16,000)
    // or the matrix is not 16-byte aligned the row is gathered through L1/L2.  exp bit 32 forces the gather form, bits
    // 64 / 128 / 256 force 512 / 1,024 / 256 threads.
    const size_t row_lds = sizeof(float) * ((size_t)(p.row_range ? p.max_span : n) + 8);
    constexpr size_t kLdsCu = 160 * 1024;
    int threads = 0;
    if (two && !(exp & 32) && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0)
    {
        int best_wgs = 0, best_waves = 0;
        // measured at n = 10,020, round 3 with three operand sets (gather form 0.336 ms): 256 threads (three rows per CU)
        // 0.255, 512 threads (two rows) 0.263, 384 threads (three rows, six waves each) 0.312, 1,024 threads (one row per
        // CU) 0.383; operand sets at 256 threads: 3: 0.254, 4: 0.244, 5: 0.242, 6: 0.243, 7: 0.246, 9 (two waves per SIMD): 0.328
        for (int t : {256, 512, 1024})
        {
            const size_t fixed = sizeof(int4) * t + sizeof(int) * (2 * t + 1) + sizeof(int) * 2 * (t / 64) + 64;
            const int wgs = std::min((int)(kLdsCu / (row_lds + fixed)), ((mode == 0 && !validate) ? 16 : 12) / (t / 64));
            const int waves = wgs * (t / 64);
            const bool forced = ((exp & 64) && t == 512) || ((exp & 128) && t == 1024) || ((exp & 256) && t == 256);
            if (t == 1024 && !forced) continue;
            if (wgs >= (forced ? 1 : 2) && (wgs > best_wgs || (wgs == best_wgs && waves > best_waves) || forced))
            {
                best_wgs = forced ? 1000 : wgs;
                best_waves = waves;
                threads = t;
            }
        }
    }
#define CUSK_ROWS2_T(M, V, T)                                                                                     \
    do                                                                                                            \
    {                                                                                                             \
        static size_t have = 0;                                                                                   \
        if (row_lds > have)                                                                                       \
        {                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(level1_rows2_kernel<M, V, T, true>),         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds);                  \
            have = row_lds;                                                                                       \
        }                                                                                                         \
        hipLaunchKernelGGL((level1_rows2_kernel<M, V, T, true>), grid, dim3(T), row_lds, st, p, rp);              \
    } while (0)
#define CUSK_ROWS2(M, V)                                                                                          \
    do                                                                                                            \
    {                                                                                                             \
        if (threads == 256)                                                                                       \
            CUSK_ROWS2_T(M, V, 256);                                                                              \
        else if (threads == 512)                                                                                  \
            CUSK_ROWS2_T(M, V, 512);                                                                              \
        else if (threads == 1024)                                                                                 \
            CUSK_ROWS2_T(M, V, 1024);                                                                             \
        else                                                                                                      \
            hipLaunchKernelGGL((level1_rows2_kernel<M, V, 256, false>), grid, dim3(256), 0, st, p, rp);           \
    } while (0)
    if (mode == 0 && !validate)
    {
        if (two)
            CUSK_ROWS2(0, false);
        else
            hipLaunchKernelGGL((level1_rows_kernel<0, false>), grid, blk, 0, st, p, rp);
    }
    else if (mode == 0)
    {
        if (two)
            CUSK_ROWS2(0, true);
        else
            hipLaunchKernelGGL((level1_rows_kernel<0, true>), grid, blk, 0, st, p, rp);
    }
    else if (!validate)
    {
        if (two)
            CUSK_ROWS2(1, false);
        else
            hipLaunchKernelGGL((level1_rows_kernel<1, false>), grid, blk, 0, st, p, rp);
    }
    else
    {
        if (two)
            CUSK_ROWS2(1, true);
        else
            hipLaunchKernelGGL((level1_rows_kernel<1, true>), grid, blk, 0, st, p, rp);
    }
#undef CUSK_ROWS2
#undef CUSK_ROWS2_T
    if (ev_end) (void)hipEventRecord(ev_end, st);
    if (mode != 0 && !defer_apply)
        hipLaunchKernelGGL(level1_apply_kernel, dim3((n + 3) / 4), dim3(256), 0, st, p.off, p.nbr, sel, p.adj, p.deg, n,
                           p.words, p.slots, p.cnt, static_cast<const int4 *>(meta), (two && !has_ti) ? canon : nullptr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// result expansion into the reference's dense layouts
// ---------------------------------------------------------------------------

__global__ void expand_adj_kernel(const unsigned long long *__restrict__ adj, int *G, int n, int words)
{
    const int row = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n) return;
    const unsigned long long w = adj[(size_t)row * words + (col >> 6)];
    G[(size_t)row * n + col] = (int)((w >> (col & 63)) & 1ull);
}

hipError_t launch_expand_adj(const unsigned long long *adj, int *G, int n, int words, hipStream_t st)
{
    hipLaunchKernelGGL(expand_adj_kernel, dim3((n + 255) / 256, n), dim3(256), 0, st, adj, G, n, words);
    return hipGetLastError();
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
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    // a stored z is never NaN (NaN fails z < th) and never negative: integer order == float order
    const int zi = __float_as_int(z[i]);
    atomicMax(reinterpret_cast<int *>(&pmax[(size_t)x[i] * n + y[i]]), zi);
    atomicMax(reinterpret_cast<int *>(&pmax[(size_t)y[i] * n + x[i]]), zi);
}

hipError_t launch_expand_pmax(const unsigned long long *adj, const unsigned long long *adj0, const float *C, float *pmax,
                              int n, int words, const int *x, const int *y, const float *z, long long nrec, hipStream_t st)
{
    hipLaunchKernelGGL(expand_pmax_kernel, dim3((n + 255) / 256, n), dim3(256), 0, st, adj, adj0, C, pmax, n, words);
    if (nrec > 0)
        hipLaunchKernelGGL(scatter_pmax_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st, x, y, z, nrec, pmax,
                           n);
    return hipGetLastError();
}

// out[a * k + b] = M[idx[a] * n + idx[b]]
__global__ void gather_sub_kernel(const float *__restrict__ M, int n, const int *__restrict__ idx, int k, float *out)
{
    const int a = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < k) out[(size_t)a * k + b] = M[(size_t)idx[a] * n + idx[b]];
}

hipError_t launch_gather_sub(const float *M, int n, const int *idx, int k, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(gather_sub_kernel, dim3((k + 255) / 256, k), dim3(256), 0, st, M, n, idx, k, out);
    return hipGetLastError();
}

// many sub-matrices in one launch (cusk_gather_rows): one wave per output row
__global__ void __launch_bounds__(256) gather_rows_kernel(const float *__restrict__ M, int n, const int *__restrict__ idx,
                                                           const int *__restrict__ row_src, const int *__restrict__ row_k,
                                                           const long long *__restrict__ row_first,
                                                           const long long *__restrict__ row_out, long long nrows, float *out)
{
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= nrows) return;
    const float *src = M + (size_t)row_src[t] * n;
    const int k = row_k[t];
    const int *cols = idx + row_first[t];
    float *dst = out + row_out[t];
    for (int c = lane; c < k; c += 64) dst[c] = src[cols[c]];
}

hipError_t launch_gather_rows(const float *M, int n, const int *idx, const int *row_src, const int *row_k, const long long *row_first,
                              const long long *row_out, long long nrows, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, st, M, n, idx, row_src, row_k, row_first,
                       row_out, nrows, out);
    return hipGetLastError();
}

// batched runs: the bitmap rows of every block, cut to the block's own words, packed back to back
// tail > 0: only the last `tail` rows of every block (the traits: all the pruning of depth 1 looks at)
__global__ void __launch_bounds__(256) pack_block_bits_kernel(const unsigned long long *__restrict__ adj, int n, int words,
                                                               const int2 *__restrict__ row_range, const int *__restrict__ row_blk,
                                                               const long long *__restrict__ blk_woff, unsigned long long *out,
                                                               int tail)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const int b = row_blk[row];
    if (b < 0) return;
    const int2 rg = row_range[row];
    const int first = (tail > 0) ? max(rg.x, rg.y - tail) : rg.x;
    if (row < first) return;
    const int wb = (rg.y - rg.x + 63) >> 6, w0 = rg.x >> 6;
    unsigned long long *dst = out + blk_woff[b] + (long long)(row - first) * wb;
    for (int w = lane; w < wb; w += 64) dst[w] = adj[(size_t)row * words + w0 + w];
}

hipError_t launch_pack_block_bits(const unsigned long long *adj, int n, int words, const int2 *row_range, const int *row_blk,
                                  const long long *blk_woff, unsigned long long *out, int tail, hipStream_t st)
{
    hipLaunchKernelGGL(pack_block_bits_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, adj, n, words, row_range, row_blk,
                       blk_woff, out, tail);
    return hipGetLastError();
}

}  // namespace cusk
