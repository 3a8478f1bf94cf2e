// sepselect.hip -- greedy forward selection of separating sets for a batch of outer pairs.
//
// Reference: cusk_postprocessing/sepselect.py:262-329 (find_maximal_and_min_pcorr_sepsets_incr).  For every outer
// pair (i, j) of an unshielded triple it grows a set S out of the trait neighbours of i, one variable per round:
// the candidate t that minimises |Fisher z| of the partial correlation of (i, j | S + t) joins S; the round
// also decides "independent" against norm.ppf(1 - alpha/2) / sqrt(n - |S + t| - 3), notes whether the minimum of
// that z over the rounds has been passed, and stops once a separating set exists and the next one is none.  The
// reference evaluates every candidate of every round with a fresh np.linalg.inv of the (|S| + 3)-variable
// correlation sub-matrix (:8-18, :162-164): O(t^2 * t^3) flops per pair in Python.
//
// Here one wavefront owns a pair and keeps the residual covariance of {i, j} + remaining candidates GIVEN S in LDS
// (t x t block plus the two border rows).  A candidate's z needs only its own row of that state (the 3 x 3 Schur
// complement), and accepting it is one rank-1 downdate, so a pair costs O(t^3) flops in total, all in double
// precision.  Mathematically the same partial correlations as the inverse gives; rounding differs in the last
// bits, which can only matter for exact ties (documented in DESIGN.md; the parity tests compare every output
// file with the reference's).
#include "cusk_internal.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace cusk {

namespace {

constexpr int kSepCaps[] = {8, 16, 32, 64, 84};  // candidate-count classes served from LDS (84: 59 KB)
constexpr int kSepLdsClasses = 5;

struct SepBatch
{
    const double *tc;       // n x p: corr[v, t] for every variable v and trait t
    int p;
    const int *pair_i, *pair_j;
    const double *pair_c;   // corr[i, j]
    const long long *cand_off;
    const int *cand;        // trait ids in the reference's iteration order
    const double *thr;      // thr[l] = quantile / sqrt(num_samples - l - 3), l = 0 .. max candidates
    int *sel;               // cand_off layout: accepted traits in order
    int *sel_len;
    int *flags;             // bit 0: the minimum was passed; bits 8..: status (0 ok, 1 no comparable candidate, 2 singular)
    const int *list;        // pair ids of this launch
    double *ws;             // global work space for pairs beyond the LDS classes
    long long ws_stride;    // doubles per pair in ws
};

// e = a * w + c with 0 <= c < w: float reciprocal while it is exact (w <= 512, e < 2^18), integer division beyond
__device__ __forceinline__ void split_index(int e, int w, float inv_w, int &a, int &c)
{
    a = (w <= 512) ? (int)(((float)e + 0.5f) * inv_w) : e / w;
    c = e - a * w;
}

__device__ __forceinline__ double abs_fisher_z(double r) { return fabs(0.5 * log(fabs((1.0 + r) / (1.0 - r)))); }

template <bool IN_LDS>
__global__ void __launch_bounds__(64) sepselect_kernel(SepBatch b, int cap)
{
    extern __shared__ double s_mem[];
    const int lane = threadIdx.x;
    const int pid = b.list[blockIdx.x];
    const long long c0 = b.cand_off[pid];
    const int t = (int)(b.cand_off[pid + 1] - c0);
    const int ld = cap | 1;
    // carve: M[cap * ld], ui[cap], uj[cap], col[cap], fcl[cap], rem[cap] (ints)
    double *M = IN_LDS ? s_mem : b.ws + (size_t)blockIdx.x * b.ws_stride;
    double *ui = M + (size_t)cap * ld;
    double *uj = ui + cap;
    double *col = uj + cap;
    double *fcl = col + cap;
    int *rem = reinterpret_cast<int *>(fcl + cap);  // remaining candidates, iteration order kept
    const int vi = b.pair_i[pid], vj = b.pair_j[pid];
    const int p = b.p;
    for (int k = lane; k < t; k += 64)
    {
        const int tk = b.cand[c0 + k];
        ui[k] = b.tc[(size_t)vi * p + tk];
        uj[k] = b.tc[(size_t)vj * p + tk];
        rem[k] = k;
    }
    {
        const float inv_t = 1.0f / (float)max(t, 1);
        for (int e = lane; e < t * t; e += 64)
        {
            int a, c;
            split_index(e, t, inv_t, a, c);
            M[a * ld + c] = b.tc[(size_t)b.cand[c0 + a] * p + b.cand[c0 + c]];
        }
    }
    __syncthreads();
    double rii = 1.0, rjj = 1.0, rij = b.pair_c[pid];
    int status = 0;
    if (rii * rjj - rij * rij == 0.0) status = 2;  // the 2 x 2 matrix itself is singular
    bool separated = abs_fisher_z(rij / sqrt(fabs(rii * rjj))) < b.thr[0];
    bool seen_minimum = false;
    double previous = INFINITY;
    int len = 0;
    for (int r = t; r > 0 && status == 0; r--)
    {
        const int size = t - r + 1;
        // ---- every remaining candidate through its 3 x 3 Schur complement ----
        double best = INFINITY;
        int pick = -1;  // position in rem[]
        bool singular = false;
        for (int q = lane; q < r; q += 64)
        {
            const int k = rem[q];
            const double mkk = M[k * ld + k], a = ui[k], c = uj[k];
            const double va = rii - a * a / mkk, vb = rjj - c * c / mkk, vab = rij - a * c / mkk;
            if (mkk == 0.0 || va * vb == 0.0) singular = true;
            const double z = abs_fisher_z(vab / sqrt(fabs(va * vb)));
            if (z <= best)
            {  // ties go to the later candidate of the iteration order, as `<=` does in the reference's loop
                best = z;
                pick = q;
            }
        }
        for (int o = 32; o > 0; o >>= 1)
        {
            const double oz = __shfl_xor(best, o);
            const int ok = __shfl_xor(pick, o);
            if (ok >= 0 && (pick < 0 || oz < best || (oz == best && ok > pick)))
            {
                best = oz;
                pick = ok;
            }
        }
        if (__any(singular))
        {
            status = 2;
            break;
        }
        if (pick < 0)
        {  // nothing comparable (NaN everywhere): the reference fails on remove(None)
            status = 1;
            break;
        }
        if (best > previous && separated && !seen_minimum) seen_minimum = true;
        const bool indep = best < b.thr[size];
        if (separated && !indep) break;
        separated = separated || indep;
        previous = best;
        const int kp = rem[pick];
        if (lane == 0) b.sel[c0 + len] = b.cand[c0 + kp];
        len++;
        if (r == 1) break;
        // ---- drop it from the list (order kept), then the rank-1 downdate of what remains ----
        const double piv = M[kp * ld + kp], ai = ui[kp], aj = uj[kp];
        for (int base = 0; base < r; base += 64)
        {  // chunks ascend, so a chunk reads its upper neighbour's first entry before that chunk overwrites it
            const int q = base + lane;
            const int moved = (q >= pick && q + 1 < r) ? rem[q + 1] : -1;
            __syncthreads();
            if (moved >= 0) rem[q] = moved;
            __syncthreads();
        }
        const int r1 = r - 1;
        rii -= ai * ai / piv;
        rjj -= aj * aj / piv;
        rij -= ai * aj / piv;
        for (int q = lane; q < r1; q += 64)
        {
            const int k = rem[q];
            const double cv = M[k * ld + kp];
            const double f = cv / piv;
            col[q] = cv;
            fcl[q] = f;
            ui[k] -= f * ai;
            uj[k] -= f * aj;
        }
        __syncthreads();
        const float inv_r = 1.0f / (float)r1;
        for (int e = lane; e < r1 * r1; e += 64)
        {
            int qa, qc;
            split_index(e, r1, inv_r, qa, qc);
            M[rem[qa] * ld + rem[qc]] -= fcl[qa] * col[qc];
        }
        __syncthreads();
    }
    if (lane == 0)
    {
        b.sel_len[pid] = len;
        b.flags[pid] = (seen_minimum ? 1 : 0) | (status << 8);
    }
}

size_t sep_bytes(int cap) { return sizeof(double) * ((size_t)cap * (cap | 1) + 4 * (size_t)cap) + sizeof(int) * (size_t)cap; }

}  // namespace

int sepselect_greedy_impl(cusk_engine *e, const double *trait_corr, long long n, int p, long long npairs, const int *pair_i,
                          const int *pair_j, const double *pair_corr, const long long *cand_off, const int *cand,
                          const double *thr, int nthr, int *sel, int *sel_len, int *flags, float *kernel_ms)
{
    if (!e || !trait_corr || n <= 0 || p <= 0 || npairs < 0 || !cand_off || !thr || !sel_len || !flags)
        return fail(e, CUSK_ERR_ARG, "cusk_sepselect_greedy: bad arguments");
    if (kernel_ms) *kernel_ms = 0.0f;
    if (npairs == 0) return CUSK_OK;
    if (!pair_i || !pair_j || !pair_corr) return fail(e, CUSK_ERR_ARG, "cusk_sepselect_greedy: bad arguments");
    const long long total = cand_off[npairs];
    if (total > 0 && (!cand || !sel)) return fail(e, CUSK_ERR_ARG, "cusk_sepselect_greedy: bad arguments");
    // shape checks on the host: every index the kernel forms stays inside the buffers it was given
    int max_t = 0;
    std::vector<std::vector<int>> lists(kSepLdsClasses + 1);
    for (long long k = 0; k < npairs; k++)
    {
        const long long t = cand_off[k + 1] - cand_off[k];
        if (t < 0 || t > p || cand_off[k] < 0) return fail(e, CUSK_ERR_ARG, "cusk_sepselect_greedy: bad candidate offsets");
        if (pair_i[k] < 0 || pair_i[k] >= n || pair_j[k] < 0 || pair_j[k] >= n)
            return fail(e, CUSK_ERR_ARG, "cusk_sepselect_greedy: pair index out of range");
        max_t = std::max(max_t, (int)t);
        int c = 0;
        while (c < kSepLdsClasses && t > kSepCaps[c]) c++;
        lists[c].push_back((int)k);
    }
    if (nthr < max_t + 1) return fail(e, CUSK_ERR_ARG, "cusk_sepselect_greedy: threshold table shorter than the longest candidate list");
    for (long long k = 0; k < total; k++)
        if (cand[k] < 0 || cand[k] >= p) return fail(e, CUSK_ERR_ARG, "cusk_sepselect_greedy: candidate is not a trait index");

    CUSK_HIP(e, hipSetDevice(e->device));
    hipStream_t s = e->stream;
    DevBuf d_tc, d_pi, d_pj, d_pc, d_off, d_cand, d_thr, d_sel, d_len, d_flags, d_list, d_ws;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> hipError_t {
        hipError_t st = b.ensure(std::max<size_t>(bytes, 8));
        if (st != hipSuccess || bytes == 0) return st;
        return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s);
    };
    CUSK_HIP(e, up(d_tc, trait_corr, sizeof(double) * (size_t)n * p));
    CUSK_HIP(e, up(d_pi, pair_i, sizeof(int) * (size_t)npairs));
    CUSK_HIP(e, up(d_pj, pair_j, sizeof(int) * (size_t)npairs));
    CUSK_HIP(e, up(d_pc, pair_corr, sizeof(double) * (size_t)npairs));
    CUSK_HIP(e, up(d_off, cand_off, sizeof(long long) * (size_t)(npairs + 1)));
    CUSK_HIP(e, up(d_cand, cand, sizeof(int) * (size_t)total));
    CUSK_HIP(e, up(d_thr, thr, sizeof(double) * (size_t)nthr));
    CUSK_HIP(e, d_sel.ensure(std::max<size_t>(sizeof(int) * (size_t)total, 8)));
    CUSK_HIP(e, d_len.ensure(sizeof(int) * (size_t)npairs));
    CUSK_HIP(e, d_flags.ensure(sizeof(int) * (size_t)npairs));
    std::vector<int> flat;
    flat.reserve((size_t)npairs);
    size_t first[kSepLdsClasses + 2];
    for (int c = 0; c <= kSepLdsClasses; c++)
    {
        first[c] = flat.size();
        flat.insert(flat.end(), lists[c].begin(), lists[c].end());
    }
    first[kSepLdsClasses + 1] = flat.size();
    CUSK_HIP(e, up(d_list, flat.data(), sizeof(int) * flat.size()));
    const size_t nbig = lists[kSepLdsClasses].size();
    const size_t big_bytes = (sep_bytes(max_t) + 15) & ~(size_t)15;
    // the HBM work space of the long-list pairs is bounded: they run in batches of what fits into the budget (option
    // "sepselect_ws_bytes", default 4 GiB)
    const size_t big_batch = std::max<size_t>(1, std::min<size_t>(nbig, (size_t)std::max<long long>(e->opt_sep_ws_budget, 0) / big_bytes));
    if (nbig) CUSK_HIP(e, d_ws.ensure(big_bytes * big_batch));

    SepBatch b;
    b.tc = d_tc.as<double>();
    b.p = p;
    b.pair_i = d_pi.as<int>();
    b.pair_j = d_pj.as<int>();
    b.pair_c = d_pc.as<double>();
    b.cand_off = d_off.as<long long>();
    b.cand = d_cand.as<int>();
    b.thr = d_thr.as<double>();
    b.sel = d_sel.as<int>();
    b.sel_len = d_len.as<int>();
    b.flags = d_flags.as<int>();
    b.ws = d_ws.as<double>();
    b.ws_stride = (long long)(big_bytes / sizeof(double));
    hipEvent_t ev0, ev1;
    CUSK_HIP(e, hipEventCreate(&ev0));
    CUSK_HIP(e, hipEventCreate(&ev1));
    CUSK_HIP(e, hipEventRecord(ev0, s));
    for (int c = 0; c <= kSepLdsClasses; c++)
    {
        const size_t cnt = first[c + 1] - first[c];
        if (cnt == 0) continue;
        b.list = d_list.as<int>() + first[c];
        if (c < kSepLdsClasses)
        {
            const size_t lds = sep_bytes(kSepCaps[c]);
            if (lds > 48 * 1024)
                CUSK_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void *>(&sepselect_kernel<true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(sepselect_kernel<true>, dim3((unsigned)cnt), dim3(64), lds, s, b, kSepCaps[c]);
        }
        else
        {
            for (size_t done = 0; done < cnt; done += big_batch)
            {  // same stream: a batch reuses the work space after the previous one has finished
                b.list = d_list.as<int>() + first[c] + done;
                hipLaunchKernelGGL(sepselect_kernel<false>, dim3((unsigned)std::min(big_batch, cnt - done)), dim3(64), 0, s, b,
                                   max_t);
            }
        }
        CUSK_HIP(e, hipGetLastError());
    }
    CUSK_HIP(e, hipEventRecord(ev1, s));
    if (total) CUSK_HIP(e, hipMemcpyAsync(sel, d_sel.p, sizeof(int) * (size_t)total, hipMemcpyDeviceToHost, s));
    CUSK_HIP(e, hipMemcpyAsync(sel_len, d_len.p, sizeof(int) * (size_t)npairs, hipMemcpyDeviceToHost, s));
    CUSK_HIP(e, hipMemcpyAsync(flags, d_flags.p, sizeof(int) * (size_t)npairs, hipMemcpyDeviceToHost, s));
    CUSK_HIP(e, hipStreamSynchronize(s));
    float ms = 0.0f;
    CUSK_HIP(e, hipEventElapsedTime(&ms, ev0, ev1));
    if (kernel_ms) *kernel_ms = ms;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    for (DevBuf *d : {&d_tc, &d_pi, &d_pj, &d_pc, &d_off, &d_cand, &d_thr, &d_sel, &d_len, &d_flags, &d_list, &d_ws}) d->release();
    return CUSK_OK;
}

}  // namespace cusk

extern "C" int cusk_sepselect_greedy(cusk_engine *e, const double *trait_corr, long long n, int p, long long npairs,
                                     const int *pair_i, const int *pair_j, const double *pair_corr, const long long *cand_off,
                                     const int *cand, const double *thr, int nthr, int *sel, int *sel_len, int *flags,
                                     float *kernel_ms)
{
    return cusk::sepselect_greedy_impl(e, trait_corr, n, p, npairs, pair_i, pair_j, pair_corr, cand_off, cand, thr, nthr, sel,
                                       sel_len, flags, kernel_ms);
}
