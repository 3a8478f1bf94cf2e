// corr_build.hip -- SNP x SNP / SNP x trait / trait x trait correlations from packed .bed.
//
// Replaces /root/reference/cusk/src/corr_host.cu:1023-1197 (cu_marker_phen_corr_pearson,
// cu_corr_pearson_npn) and the kernels they launch (src/corr_kernels.cu:157-238,
// :285-343, :478-565; decode tables include/mps/bed_lut_gpu.h):
//   mxm : Kendall tau-b from the 3x3 genotype contingency table over jointly
//         non-missing individuals, mapped by sin(pi/2 tau)  (exact integer counts,
//         the reference's fp32 epilogue order)
//   mxp : Pearson with precomputed marker mean/std, NaN traits skipped
//   pxp : sum(y_a y_b) / #jointly-valid
// and writes the n x n square matrix of src/cli.cpp:597-649 straight into HBM
// (markers first, traits last, unit diagonal, symmetric) so that the level
// sweep can start without a host round trip.
//
// .bed 2-bit codes, low bits first: 00 -> 2, 01 -> missing, 10 -> 1, 11 -> 0.
#include <type_traits>
#include <cmath>
#include <cstring>
#include <vector>

#include "cusk_internal.h"

namespace cusk {

// ---------------------------------------------------------------------------
// decode: .bed bytes -> three bit planes per marker (1 bit / individual)
//   plane 0: genotype == 1, plane 1: genotype == 2, plane 2: non-missing
// ---------------------------------------------------------------------------
__global__ void bed_to_bitplanes_kernel(const unsigned char *__restrict__ bed, unsigned long long *planes, size_t m,
                                        size_t N, size_t clb, size_t w64)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= m * w64) return;
    const size_t mk = gid / w64, w = gid - mk * w64;
    unsigned long long b1 = 0, b2 = 0, bv = 0;
    const unsigned char *src = bed + mk * clb + w * 16;
    for (int byte = 0; byte < 16; byte++)
    {
        const size_t bi = w * 16 + byte;
        if (bi >= clb) break;
        const unsigned v = src[byte];
        for (int j = 0; j < 4; j++)
        {
            const size_t smp = bi * 4 + j;
            if (smp >= N) break;
            const unsigned code = (v >> (2 * j)) & 3u;
            const unsigned long long bit = 1ull << (byte * 4 + j);
            if (code == 2u) b1 |= bit;
            if (code == 0u) b2 |= bit;
            if (code != 1u) bv |= bit;
        }
    }
    planes[(0 * m + mk) * w64 + w] = b1;
    planes[(1 * m + mk) * w64 + w] = b2;
    planes[(2 * m + mk) * w64 + w] = bv;
}

// sin(x) in double precision for the arguments this file has, |x| <= pi/2 (x = pi/2 tau): the two kernel polynomials of
// fdlibm (k_sin.c / k_cos.c, error < 1 ulp) and sin x = sign(x) cos(pi/2 - |x|) above pi/4, the difference formed from the
// two-part pi/2 of e_rem_pio2.c.  The library routine spends ~300 double-precision instructions per call on the general
// argument reduction; sixteen calls per lane made the epilogue of the SNP x SNP kernels as long as 64 K-blocks of matrix
// products.  The result is rounded to float by the caller: it can differ from the library's only where the two doubles
// straddle a float rounding boundary, as the library's own last bit already may against the host's libm (the oracle).
// NaN stays NaN; anything outside the range (rounding can put |tau| a few ulps above 1) goes to the library.
__device__ __forceinline__ double sin_quarter_turn(double x)
{
    const double ax = fabs(x);
    if (!(ax <= 1.6)) return sin(x);
    if (ax <= 0.78539816339744830962)
    {
        const double z = x * x, v = z * x;
        const double r = 8.33333333332248946124e-03 +
                         z * (-1.98412698298579493134e-04 +
                              z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
        return x + v * (-1.66666666666666324348e-01 + z * r);
    }
    const double y = (1.57079632679489655800e+00 - ax) + 6.12323399573676603587e-17;
    const double z = y * y;
    const double r = z * (4.16666666666666019037e-02 +
                          z * (-1.38888888888741095749e-03 +
                               z * (2.48015872894767294178e-05 +
                                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double c = 1.0 - (0.5 * z - z * r);
    return x < 0.0 ? -c : c;
}

// tau-b -> sin(pi/2 tau) in the reference's fp32 operation order (corr_kernels.cu:544-564)
__device__ __forceinline__ float npn_from_counts(const float *s)
{
    float p = ((s[0] * (s[4] + s[5] + s[7] + s[8])) + (s[1] * (s[5] + s[8])) + (s[3] * (s[7] + s[8])) + (s[4] * s[8]));
    float q = ((s[1] * (s[3] + s[6])) + (s[2] * (s[3] + s[4] + s[6] + s[7])) + (s[4] * s[6]) + (s[5] * (s[6] + s[7])));
    float t = ((s[0] * (s[1] + s[2])) + (s[1] * s[2]) + (s[3] * (s[4] + s[5])) + (s[4] * s[5]) + (s[6] * (s[7] + s[8])) +
               (s[7] * s[8]));
    float u = ((s[0] * (s[3] + s[6])) + (s[1] * (s[4] + s[7])) + (s[2] * (s[5] + s[8])) + (s[3] * s[6]) + (s[4] * s[7]) +
               (s[5] * s[8]));
    float kendall = (p - q) / sqrtf((p + q + t) * (p + q + u));
    return (float)sin_quarter_turn(M_PI / 2 * (double)kendall);
}

// counts for a 32 x 32 tile of marker pairs (upper triangle of tiles), 2 x 2 pairs per thread
constexpr int kTile = 32;
constexpr int kKW = 8;  // 64-bit words per staged K chunk
__global__ void __launch_bounds__(256) mxm_popcount_kernel(const unsigned long long *__restrict__ planes, float *C,
                                                            size_t m, size_t w64, size_t n, int tiles)
{
    __shared__ unsigned long long sa[3][kTile][kKW + 1];
    __shared__ unsigned long long sb[3][kTile][kKW + 1];
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
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // pair sub-tile (2 rows x 2 cols)
    unsigned cnt[2][2][9];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int c = 0; c < 9; c++) cnt[a][b][c] = 0;

    for (size_t w0 = 0; w0 < w64; w0 += kKW)
    {
        for (int e = threadIdx.x; e < 3 * kTile * kKW; e += 256)
        {
            const int pl = e / (kTile * kKW), r = (e / kKW) % kTile, k = e % kKW;
            const size_t w = w0 + k;
            const size_t ma = (size_t)bi * kTile + r, mb = (size_t)bj * kTile + r;
            sa[pl][r][k] = (ma < m && w < w64) ? planes[((size_t)pl * m + ma) * w64 + w] : 0ull;
            sb[pl][r][k] = (mb < m && w < w64) ? planes[((size_t)pl * m + mb) * w64 + w] : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kKW; k++)
        {
            unsigned long long a1[2], a2[2], av[2], b1[2], b2[2], bv[2];
#pragma unroll
            for (int u = 0; u < 2; u++)
            {
                a1[u] = sa[0][ty * 2 + u][k];
                a2[u] = sa[1][ty * 2 + u][k];
                av[u] = sa[2][ty * 2 + u][k];
                b1[u] = sb[0][tx * 2 + u][k];
                b2[u] = sb[1][tx * 2 + u][k];
                bv[u] = sb[2][tx * 2 + u][k];
            }
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
                {
                    // raw products; the table cells are formed from them in the epilogue
                    cnt[a][b][0] += __popcll(a1[a] & b1[b]);
                    cnt[a][b][1] += __popcll(a1[a] & b2[b]);
                    cnt[a][b][2] += __popcll(a2[a] & b1[b]);
                    cnt[a][b][3] += __popcll(a2[a] & b2[b]);
                    cnt[a][b][4] += __popcll(a1[a] & bv[b]);
                    cnt[a][b][5] += __popcll(a2[a] & bv[b]);
                    cnt[a][b][6] += __popcll(av[a] & b1[b]);
                    cnt[a][b][7] += __popcll(av[a] & b2[b]);
                    cnt[a][b][8] += __popcll(av[a] & bv[b]);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
        {
            const size_t i = (size_t)bi * kTile + ty * 2 + a, j = (size_t)bj * kTile + tx * 2 + b;
            if (i < m && j < m && i < j)
            {
                const unsigned *c = cnt[a][b];
                const unsigned n11 = c[0], n12 = c[1], n21 = c[2], n22 = c[3], n1v = c[4], n2v = c[5], nv1 = c[6],
                               nv2 = c[7], nvv = c[8];
                float s[9];
                // s[3*va + vb], va/vb = genotype value of the row/column marker
                s[4] = (float)n11;
                s[5] = (float)n12;
                s[7] = (float)n21;
                s[8] = (float)n22;
                s[3] = (float)(n1v - n11 - n12);
                s[6] = (float)(n2v - n21 - n22);
                s[1] = (float)(nv1 - n11 - n21);
                s[2] = (float)(nv2 - n12 - n22);
                s[0] = (float)(nvv - n1v - n2v - (nv1 - n11 - n21) - (nv2 - n12 - n22));
                const float r = npn_from_counts(s);
                C[i * n + j] = r;
                C[j * n + i] = r;
            }
        }
}

// ---------------------------------------------------------------------------
// SNP x SNP contingency counts as an int8 MFMA GEMM, decoded from .bed on the fly.
//
// Every cell of the 3x3 table of a marker pair is a dot product over individuals of two 0/1
// indicator vectors.  With the planes X in {[g==1], [g==2], [non-missing]} the nine products
// X_a(rows) . X_b(cols)^T are nine int8 GEMMs with exact int32 accumulation
// (v_mfma_i32_32x32x32_i8: 32x32 outputs, 32 individuals per instruction).
//
// Workgroup = 256 threads = 4 waves (2x2), output tile 64 x 64 marker pairs, K block = 128
// individuals.  Per K block each thread loads 16 bytes of packed .bed (64 individuals of one of
// the 128 tile markers), expands them with bit logic + one multiply per 4 individuals into the
// three int8 planes and stores them to LDS in MFMA fragment order (48 KB: [plane][k-step][half]
// [row] x 16 B, so every fragment is one conflict-free ds_read_b128).  Each wave then issues
// 4 k-steps x 9 MFMAs on its 32 x 32 sub-tile (9 accumulators = 144 VGPRs).  HBM traffic is the
// packed .bed itself (2 bits per genotype); the int8 planes exist only in LDS.  Two workgroups
// per CU overlap one's decode with the other's MFMAs.  The epilogue converts the counts to
// floats and applies the reference's tau-b -> sin(pi/2 tau) formula in its fp32 operation order,
// writing both triangles of the square matrix with coalesced rows (transpose through LDS).
// Only tiles of the upper triangle are launched.
// ---------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kMT = 64;    // markers per tile side
constexpr int kKB = 128;   // individuals per K block
constexpr int kKS = kKB / 32;

// 16 genotype indicator bits at even positions of x -> 16 bytes of 0/1
__device__ __forceinline__ v4i expand16(unsigned x)
{
    v4i o;
    o.x = (int)((((x)&0x55u) * 0x00041041u) & 0x01010101u);
    o.y = (int)((((x >> 8) & 0x55u) * 0x00041041u) & 0x01010101u);
    o.z = (int)((((x >> 16) & 0x55u) * 0x00041041u) & 0x01010101u);
    o.w = (int)((((x >> 24) & 0x55u) * 0x00041041u) & 0x01010101u);
    return o;
}

__global__ void __launch_bounds__(256, 2) mxm_mfma_kernel(const unsigned char *__restrict__ bed, float *C, size_t m, size_t N,
                                                           size_t clb, size_t n, int tiles)
{
    __shared__ v4i sA[3][kKS][2][kMT];
    __shared__ v4i sB[3][kKS][2][kMT];
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // decode role: tile-local marker (0..63 rows of A, 64..127 rows of B) and 64-individual half of the K block
    const int row_l = tid >> 1, q = tid & 1;
    const size_t mk = (row_l < kMT) ? (size_t)bi * kMT + row_l : (size_t)bj * kMT + (row_l - kMT);
    const bool row_ok = mk < m;
    const unsigned char *rowp = bed + mk * clb;
    v4i(*dst)[kKS][2][kMT] = (row_l < kMT) ? sA : sB;
    const int rr = row_l & (kMT - 1);

    v16i acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0;

    const size_t nkb = (N + kKB - 1) / kKB;
    auto load16 = [&](size_t kb) -> uint4 {
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        const size_t off = kb * (kKB / 4) + (size_t)q * 16;
        if (row_ok && off < clb)
        {
            const unsigned char *src = rowp + off;
            if (off + 16 <= clb && ((reinterpret_cast<uintptr_t>(src) & 3u) == 0))
            {
                const unsigned *s4 = reinterpret_cast<const unsigned *>(src);
                w = make_uint4(s4[0], s4[1], s4[2], s4[3]);
            }
            else
            {
                unsigned tmp[4] = {0u, 0u, 0u, 0u};
                for (size_t b = 0; b < 16 && off + b < clb; b++) tmp[b >> 2] |= (unsigned)src[b] << (8 * (b & 3));
                w = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
        }
        return w;
    };
    uint4 w = load16(0);
    for (size_t kb = 0; kb < nkb; kb++)
    {
        const unsigned wu[4] = {w.x, w.y, w.z, w.w};
        const size_t base = kb * kKB + (size_t)q * 64;
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            // 16 individuals; those at or beyond N (padding bits, or past the file) count as missing
            const size_t s0 = base + 16 * u;
            const unsigned nv = (!row_ok || s0 >= N) ? 0u : (unsigned)min((size_t)16, N - s0);
            const unsigned msk = (nv >= 16u) ? 0x55555555u : (((1u << (2 * nv)) - 1u) & 0x55555555u);
            const unsigned lo = wu[u] & 0x55555555u, hi = (wu[u] >> 1) & 0x55555555u;
            const unsigned b1 = hi & ~lo & msk;         // code 10 -> genotype 1
            const unsigned b2 = ~hi & ~lo & msk;        // code 00 -> genotype 2
            const unsigned bv = (hi | ~lo) & msk;       // anything but 01 (missing)
            const int ks = 2 * q + (u >> 1), h = u & 1;
            dst[0][ks][h][rr] = expand16(b1);
            dst[1][ks][h][rr] = expand16(b2);
            dst[2][ks][h][rr] = expand16(bv);
        }
        if (kb + 1 < nkb) w = load16(kb + 1);  // in flight while the MFMAs run
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kKS; ks++)
        {
            v4i a[3], b[3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++)
            {
                a[pl] = sA[pl][ks][lane >> 5][wr * 32 + (lane & 31)];
                b[pl] = sB[pl][ks][lane >> 5][wc * 32 + (lane & 31)];
            }
#pragma unroll
            for (int pa = 0; pa < 3; pa++)
#pragma unroll
                for (int pb = 0; pb < 3; pb++)
                    acc[pa][pb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[pa], b[pb], acc[pa][pb], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float *scratch = reinterpret_cast<float *>(&sA[0][0][0][0]) + wave * (32 * 33);
    const size_t i0 = (size_t)bi * kMT + wr * 32, j0 = (size_t)bj * kMT + wc * 32;
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        const int rl = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), cl = lane & 31;
        const size_t i = i0 + rl, j = j0 + cl;
        float r = 0.0f;
        if (i < m && j < m && i < j)
        {
            const unsigned n11 = acc[0][0][e], n12 = acc[0][1][e], n21 = acc[1][0][e], n22 = acc[1][1][e],
                           n1v = acc[0][2][e], n2v = acc[1][2][e], nv1 = acc[2][0][e], nv2 = acc[2][1][e],
                           nvv = acc[2][2][e];
            float s[9];
            s[4] = (float)n11;
            s[5] = (float)n12;
            s[7] = (float)n21;
            s[8] = (float)n22;
            s[3] = (float)(n1v - n11 - n12);
            s[6] = (float)(n2v - n21 - n22);
            s[1] = (float)(nv1 - n11 - n21);
            s[2] = (float)(nv2 - n12 - n22);
            s[0] = (float)(nvv - n1v - n2v - (nv1 - n11 - n21) - (nv2 - n12 - n22));
            r = npn_from_counts(s);
            C[i * n + j] = r;
        }
        scratch[cl * 33 + rl] = r;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        const int jl = 2 * e + (lane >> 5), il = lane & 31;
        const size_t i = i0 + il, j = j0 + jl;
        if (i < m && j < m && i < j) C[j * n + i] = scratch[jl * 33 + il];
    }
}

// ---------------------------------------------------------------------------
// The same nine contingency GEMMs on the FP4 matrix pipe (v_mfma_scale_f32_32x32x64_f8f6f4, twice the int8
// rate).  The operands are 0/1 indicators, so e2m1 holds them exactly and the f32 accumulation is exact (counts
// <= 2^24).  Two observations make the decode almost free:
//   * a dot product over individuals does not care about their ORDER, as long as both operands use the same
//     one.  A plane word has its 16 indicator bits at the even bit positions; (x & 0x11111111) is already a
//     valid word of eight 4-bit elements (individuals 0,2,4,...) and ((x >> 2) & 0x11111111) the other eight:
//     two ANDs and a shift instead of a bit-spreading multiply per four individuals;
//   * the element those masks produce is 0b0001 = 0.5 in e2m1, so every product is 0.25 and the accumulator
//     holds count/4, exactly; the epilogue multiplies by 4.
// Workgroup = 4 waves (2x2), tile 64 x 64 marker pairs, K block = 256 individuals = four MFMA steps of 64; each
// thread decodes 128 individuals of one tile marker per block (32 bytes of .bed) into 3 planes x 4 fragments of
// 16 bytes, stored in fragment order ([plane][k-step][lane half][row]), 48 KB for both operands.
// ---------------------------------------------------------------------------
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// Workgroup barrier that only drains the LDS queue.  __syncthreads() also waits for every outstanding global
// load (vmcnt(0)), which would cancel the two-blocks-ahead .bed requests of the K loop below.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int kKB4 = 256;        // individuals per K block
constexpr int kKS4 = kKB4 / 64;  // MFMA steps per K block

// band_w = 0: square output C[n x n], both triangles, upper-triangle tiles enumerated along blockIdx.x.
// band_w > 0 (mps block): banded output C[i * band_w + (j - i - 1)] for 0 < j - i <= band_w, tile (blockIdx.x,
// blockIdx.x + blockIdx.y).
// btile != nullptr (cusk_corr_build_batch): many LD blocks in one launch -- workgroup w computes tile (btile[w].y, btile[w].z)
// of block btile[w].x, whose genotypes start at marker bblk[..].bed_row0 of `bed` and whose matrix lies on the diagonal of
// the n x n batch allocation at variable bblk[..].base.
struct CorrBatchBlock
{
    unsigned long long bed_row0;  // first marker of the block (row of the staged .bed; index into mean / sd)
    int m;                        // markers of the block
    int base;                     // first variable of the block in the batch allocation
    long long mxp_off;            // first marker of the block in the compact marker x trait output
};

template <bool FAST>
__global__ void __launch_bounds__(256, 2) mxm_fp4_kernel(const unsigned char *__restrict__ bed_in, float *C_in, size_t m_in, size_t N,
                                                          size_t clb, size_t n, int tiles, size_t band_w,
                                                          const int4 *__restrict__ btile, const CorrBatchBlock *__restrict__ bblk)
{
    __shared__ v4i sA[3][kKS4][2][kMT];
    __shared__ v4i sB[3][kKS4][2][kMT];
    int t = blockIdx.x, bi = 0;
    const unsigned char *__restrict__ bed = bed_in;
    float *C = C_in;
    size_t m = m_in;
    if (btile != nullptr)
    {
        const int4 bt = btile[blockIdx.x];
        const CorrBatchBlock bb = bblk[bt.x];
        bed = bed_in + bb.bed_row0 * clb;
        C = C_in + (size_t)bb.base * n + bb.base;
        m = (size_t)bb.m;
        bi = bt.y;
        t = bt.z;
    }
    else if (band_w == 0)
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
    else
    {
        bi = blockIdx.x;
        t = bi + (int)blockIdx.y;
        if (t >= tiles) return;
    }
    const int bj = t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // decode role: tile-local marker (0..63 rows of A, 64..127 rows of B) and 128-individual half of the K block
    const int row_l = tid >> 1, q = tid & 1;
    const size_t mk = (row_l < kMT) ? (size_t)bi * kMT + row_l : (size_t)bj * kMT + (row_l - kMT);
    const bool row_ok = mk < m;
    const unsigned char *rowp = bed + (row_ok ? mk : 0) * clb;
    v4i(*dst)[kKS4][2][kMT] = (row_l < kMT) ? sA : sB;
    const int rr = row_l & (kMT - 1);

    v16f acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    const size_t nkb = (N + kKB4 - 1) / kKB4;
    // 32 bytes of the marker's row: individuals [kb*256 + q*128, +128)
    auto load32 = [&](size_t kb, unsigned (&w)[8]) {
        const size_t off = kb * (kKB4 / 4) + (size_t)q * 32;
#pragma unroll
        for (int u = 0; u < 8; u++) w[u] = 0u;
        if (row_ok && off < clb)
        {
            const unsigned char *src = rowp + off;
            if (off + 32 <= clb && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0))
            {
                const uint4 lo = reinterpret_cast<const uint4 *>(src)[0], hi = reinterpret_cast<const uint4 *>(src)[1];
                w[0] = lo.x, w[1] = lo.y, w[2] = lo.z, w[3] = lo.w;
                w[4] = hi.x, w[5] = hi.y, w[6] = hi.z, w[7] = hi.w;
            }
            else
            {
                for (size_t b = 0; b < 32 && off + b < clb; b++) w[b >> 2] |= (unsigned)src[b] << (8 * (b & 3));
            }
        }
    };
    const int scale1 = 0x7f7f7f7f;  // E8M0 127 = 2^0 for every block
    // one K block: decode 128 individuals of this thread's marker into LDS fragments ...
    auto decode = [&](size_t kb, const unsigned (&w)[8], auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;  // every individual of the block exists (kb < N / 256)
        const size_t base = kb * kKB4 + (size_t)q * 128;
#pragma unroll
        for (int u2 = 0; u2 < 4; u2++)
        {
            // two code words = 32 individuals = one fragment (k-step 2q + (u2 >> 1), lane half u2 & 1) per plane
            unsigned pe[3][2], po[3][2];
#pragma unroll
            for (int z = 0; z < 2; z++)
            {
                const int u = 2 * u2 + z;
                // individuals at or beyond N (padding bits, or past the file) count as missing
                unsigned msk;
                if constexpr (FULL)
                    msk = row_ok ? 0x55555555u : 0u;
                else
                {
                    const size_t s0 = base + 16 * u;
                    const unsigned nv = (!row_ok || s0 >= N) ? 0u : (unsigned)min((size_t)16, N - s0);
                    msk = (nv >= 16u) ? 0x55555555u : (((1u << (2 * nv)) - 1u) & 0x55555555u);
                }
                const unsigned lo = w[u] & 0x55555555u, hi = (w[u] >> 1) & 0x55555555u;
                const unsigned pl[3] = {hi & ~lo & msk,     // code 10 -> genotype 1
                                        ~hi & ~lo & msk,    // code 00 -> genotype 2
                                        (hi | ~lo) & msk};  // anything but 01 (missing)
#pragma unroll
                for (int k = 0; k < 3; k++)
                {
                    pe[k][z] = pl[k] & 0x11111111u;
                    po[k][z] = (pl[k] >> 2) & 0x11111111u;
                }
            }
            const int ks = 2 * q + (u2 >> 1), h = u2 & 1;
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
                v4i f;
                f.x = (int)pe[k][0];
                f.y = (int)po[k][0];
                f.z = (int)pe[k][1];
                f.w = (int)po[k][1];
                dst[k][ks][h][rr] = f;
            }
        }
    };
    // ... and the 4 x 9 products of this wave's 32 x 32 sub-tile
    auto products = [&]() {
#pragma unroll
        for (int ks = 0; ks < kKS4; ks++)
        {
            v8i a[3], b[3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++)
            {
                const v4i fa = sA[pl][ks][lane >> 5][wr * 32 + (lane & 31)];
                const v4i fb = sB[pl][ks][lane >> 5][wc * 32 + (lane & 31)];
                a[pl] = v8i{fa.x, fa.y, fa.z, fa.w, 0, 0, 0, 0};
                b[pl] = v8i{fb.x, fb.y, fb.z, fb.w, 0, 0, 0, 0};
            }
#pragma unroll
            for (int pa = 0; pa < 3; pa++)
#pragma unroll
                for (int pb = 0; pb < 3; pb++)
                    acc[pa][pb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[pa], b[pb], acc[pa][pb], 4, 4, 0, scale1, 0,
                                                                                  scale1);
        }
    };
    // Full K blocks of 16-byte aligned rows (FAST, checked by the launcher) are requested two blocks ahead with
    // unconditional loads into two alternating register sets: .bed comes from L2/HBM with a latency of several
    // K blocks' worth of MFMAs.  Rows without a marker read row 0 and are masked in the decode.
    size_t kb0 = 0;
    if constexpr (FAST)
    {
        const size_t nfull = N / kKB4;
        const uint4 *row4 = reinterpret_cast<const uint4 *>(rowp) + 2 * q;
        // The requests are issued as inline assembly so that NO compiler-generated s_waitcnt covers them: hipcc
        // drains the whole load queue (vmcnt(0)) at the first use inside a loop, which would serialise request and
        // use again.  Each register set is waited for explicitly, with exactly the other set's two loads allowed
        // to stay in flight, and everything is drained before the registers can be reused after the loop.
#define CUSK_BED_REQUEST(kb_, lo_, hi_)                                                                          \
    {                                                                                                          \
        const uint4 *src_ = row4 + (kb_) * (kKB4 / 64);                                                        \
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"            \
                     : "=&v"(lo_), "=&v"(hi_)                                                                  \
                     : "v"(src_)                                                                               \
                     : "memory");                                                                              \
    }
#define CUSK_BED_WAIT_OLDER(lo_, hi_) asm volatile("s_waitcnt vmcnt(2)" : "+v"(lo_), "+v"(hi_)::"memory")
        if (nfull > 0)
        {
            v4u alo, ahi, blo, bhi;  // native vectors: register operands of the assembly
            unsigned w[8];
            CUSK_BED_REQUEST(0, alo, ahi);
            CUSK_BED_REQUEST(min((size_t)1, nfull - 1), blo, bhi);
            for (size_t kb = 0; kb < nfull; kb += 2)
            {
                CUSK_BED_WAIT_OLDER(alo, ahi);
                w[0] = alo.x, w[1] = alo.y, w[2] = alo.z, w[3] = alo.w;
                w[4] = ahi.x, w[5] = ahi.y, w[6] = ahi.z, w[7] = ahi.w;
                decode(kb, w, std::true_type{});
                CUSK_BED_REQUEST(min(kb + 2, nfull - 1), alo, ahi);
                lds_barrier();
                products();
                lds_barrier();
                if (kb + 1 >= nfull) break;
                CUSK_BED_WAIT_OLDER(blo, bhi);
                w[0] = blo.x, w[1] = blo.y, w[2] = blo.z, w[3] = blo.w;
                w[4] = bhi.x, w[5] = bhi.y, w[6] = bhi.z, w[7] = bhi.w;
                decode(kb + 1, w, std::true_type{});
                CUSK_BED_REQUEST(min(kb + 3, nfull - 1), blo, bhi);
                lds_barrier();
                products();
                lds_barrier();
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi)::"memory");
#undef CUSK_BED_REQUEST
#undef CUSK_BED_WAIT_OLDER
            kb0 = nfull;
        }
    }
    for (size_t kb = kb0; kb < nkb; kb++)
    {
        unsigned w[8];
        load32(kb, w);
        decode(kb, w, std::false_type{});
        __syncthreads();
        products();
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float *scratch = reinterpret_cast<float *>(&sA[0][0][0][0]) + wave * (32 * 33);
    const size_t i0 = (size_t)bi * kMT + wr * 32, j0 = (size_t)bj * kMT + wc * 32;
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        const int rl = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), cl = lane & 31;
        const size_t i = i0 + rl, j = j0 + cl;
        float r = 0.0f;
        if (i < m && j < m && i < j)
        {
            // every product was 0.5 * 0.5: the accumulators hold count / 4 exactly
            const unsigned n11 = (unsigned)(acc[0][0][e] * 4.0f), n12 = (unsigned)(acc[0][1][e] * 4.0f),
                           n21 = (unsigned)(acc[1][0][e] * 4.0f), n22 = (unsigned)(acc[1][1][e] * 4.0f),
                           n1v = (unsigned)(acc[0][2][e] * 4.0f), n2v = (unsigned)(acc[1][2][e] * 4.0f),
                           nv1 = (unsigned)(acc[2][0][e] * 4.0f), nv2 = (unsigned)(acc[2][1][e] * 4.0f),
                           nvv = (unsigned)(acc[2][2][e] * 4.0f);
            float s[9];
            s[4] = (float)n11;
            s[5] = (float)n12;
            s[7] = (float)n21;
            s[8] = (float)n22;
            s[3] = (float)(n1v - n11 - n12);
            s[6] = (float)(n2v - n21 - n22);
            s[1] = (float)(nv1 - n11 - n21);
            s[2] = (float)(nv2 - n12 - n22);
            s[0] = (float)(nvv - n1v - n2v - (nv1 - n11 - n21) - (nv2 - n12 - n22));
            r = npn_from_counts(s);
            if (band_w == 0)
                C[i * n + j] = r;
            else if (j - i - 1 < band_w)
                C[i * band_w + (j - i - 1)] = r;
        }
        scratch[cl * 33 + rl] = r;
    }
    if (band_w != 0) return;  // no mirrored triangle in the banded layout (uniform per launch)
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        const int jl = 2 * e + (lane >> 5), il = lane & 31;
        const size_t i = i0 + il, j = j0 + jl;
        if (i < m && j < m && i < j) C[j * n + i] = scratch[jl * 33 + il];
    }
}

// one wave per marker, all traits: Pearson of corr_kernels.cu:157-238
constexpr int kMaxPhenRegs = 32;
__global__ void __launch_bounds__(256) mxp_kernel(const unsigned char *__restrict__ bed, const float *__restrict__ phen,
                                                   const float *__restrict__ mean, const float *__restrict__ sd, float *C,
                                                   float *mxp, size_t m, size_t N, size_t p, size_t clb, size_t n,
                                                   size_t p0, size_t pcount)
{
    const size_t mk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (mk >= m) return;
    float sgy[kMaxPhenRegs], sy[kMaxPhenRegs], sn[kMaxPhenRegs];
#pragma unroll
    for (int k = 0; k < kMaxPhenRegs; k++) sgy[k] = sy[k] = sn[k] = 0.0f;
    for (size_t bi = lane; bi < clb; bi += 64)
    {
        const unsigned v = bed[mk * clb + bi];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const size_t smp = bi * 4 + j;
            if (smp >= N) break;
            const unsigned code = (v >> (2 * j)) & 3u;
            const float valid = (code != 1u) ? 1.0f : 0.0f;
            const float g = (code == 0u) ? 2.0f : ((code == 2u) ? 1.0f : 0.0f);
#pragma unroll
            for (int k = 0; k < kMaxPhenRegs; k++)
            {
                if ((size_t)k < pcount)
                {
                    const float y = phen[(p0 + k) * N + smp];
                    if (!(y != y))
                    {
                        sgy[k] += valid * g * y;
                        sy[k] += valid * y;
                        sn[k] += valid;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kMaxPhenRegs; k++)
    {
        if ((size_t)k < pcount)
        {
            float a = sgy[k], b = sy[k], c = sn[k];
            for (int o = 32; o > 0; o >>= 1)
            {
                a += __shfl_xor(a, o);
                b += __shfl_xor(b, o);
                c += __shfl_xor(c, o);
            }
            if (lane == 0)
            {
                const float r = (a - mean[mk] * b) / (c * sd[mk]);
                const size_t t = p0 + k;
                if (C)
                {
                    C[mk * n + m + t] = r;
                    C[(m + t) * n + mk] = r;
                }
                if (mxp) mxp[mk * p + t] = r;
            }
        }
    }
}

// SNP x trait Pearson as an exact-f32 MFMA contraction over individuals (corr_kernels.cu:157-238 gives the
// formula: r = (sum g y - mean_g sum y) / (n sd_g) over the individuals where genotype and trait are both
// present).  With A = [genotype dosage | non-missing flag] (markers x individuals, decoded from .bed in
// registers) and B = [trait, NaN -> 0 | NaN flag] (individuals x traits) the three sums are A_g B_y, A_v B_y and
// A_v B_nan: v_mfma_f32_32x32x2_f32 accumulates each as a k-ordered chain of f32 FMAs (exact f32, the matrix
// pipe runs at the f32 vector rate but leaves the vector ALU to the decoding and needs 48 accumulator registers
// instead of 3 per marker-trait pair).
//
// Workgroup = 8 waves, output tile 32 markers x 32 traits; wave w owns the w-th eighth of the individuals
// (split-K) and the eight partial tiles are added in a fixed order through LDS, so the result is deterministic.
// Operand lane maps of the 32x32x2 form: lane l holds A[row l&31][k = l>>5] and B[k = l>>5][col l&31], one f32
// each: lane (i, k) walks marker i and takes individuals 2q+k of each byte (two MFMA steps per .bed byte), lane
// (j, k) walks trait row j with float4 loads.  The NaN product is only issued for 16-individual groups that
// contain a NaN.
constexpr int kMxpWaves = 8;

// FAST: every marker row of the .bed block is dword-aligned and every trait row 16-byte aligned (N % 4 == 0 and
// aligned bases, checked by the launcher), so full 16-individual groups are fetched with one dword + four
// float4 loads, unconditionally, two groups in flight.  Anything else (odd N, the last partial group) goes
// through the bounds-checked fetch.
template <bool FAST>
__global__ void __launch_bounds__(64 * kMxpWaves) mxp_mfma_kernel(const unsigned char *__restrict__ bed,
                                                                  const float *__restrict__ phen,
                                                                  const float *__restrict__ mean,
                                                                  const float *__restrict__ sd, float *C, float *mxp, size_t m,
                                                                  size_t N, size_t p, size_t clb, size_t n, size_t p0,
                                                                  int pcount, const int2 *__restrict__ mtile,
                                                                  const CorrBatchBlock *__restrict__ bblk)
{
    __shared__ float s_part[kMxpWaves / 2][3][16][64];  // partial accumulators of the upper half of the waves
    __shared__ float s_sv[kMxpWaves][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, kh = lane >> 5;
    size_t tile32 = blockIdx.x;
    if (mtile != nullptr)
    {  // batched build: tile mtile[w].y of block mtile[w].x (see mxm_fp4_kernel)
        const int2 mt = mtile[blockIdx.x];
        const CorrBatchBlock bb = bblk[mt.x];
        bed += bb.bed_row0 * clb;
        mean += bb.bed_row0;
        sd += bb.bed_row0;
        if (C) C += (size_t)bb.base * n + bb.base;
        if (mxp) mxp += (size_t)bb.mxp_off * p;
        m = (size_t)bb.m;
        tile32 = (size_t)mt.y;
    }
    const size_t mk = tile32 * 32 + r32;  // A side: this lane's marker
    const bool mok = mk < m;
    const unsigned char *rowp = bed + (mok ? mk : 0) * clb;
    const bool tok = r32 < pcount;  // B side: this lane's trait
    const float *yrow = phen + (p0 + (tok ? r32 : 0)) * N;
    // Lanes without a marker / trait still feed the MFMA (row i of A only reaches row i of the product, column j
    // of B only column j, and those are never written); their genotypes are forced to "missing".

    v16f acc_gy, acc_vy, acc_vn;
#pragma unroll
    for (int r = 0; r < 16; r++) acc_gy[r] = acc_vy[r] = acc_vn[r] = 0.0f;
    float sv = 0.0f;

    // bounds-checked operands of group g: individuals at or beyond N count as missing / zero
    auto fetch_checked = [&](size_t g, unsigned &w, float (&y)[16]) {
        const size_t s0 = g * 16, off = g * 4;
        w = 0u;
        for (size_t bb = 0; bb < 4; bb++) w |= (unsigned)((mok && off + bb < clb) ? rowp[off + bb] : 0x55u) << (8 * bb);
#pragma unroll
        for (int q = 0; q < 16; q++)
        {
            if (s0 + q >= N) w = (w & ~(3u << (2 * q))) | (1u << (2 * q));
            y[q] = (tok && s0 + q < N) ? yrow[s0 + q] : 0.0f;
        }
    };
    auto fetch_fast = [&](size_t g, unsigned &w, float (&y)[16]) {
        w = reinterpret_cast<const unsigned *>(rowp)[g];
        const float4 *y4 = reinterpret_cast<const float4 *>(yrow) + g * 4;
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const float4 v = y4[q];
            y[4 * q] = v.x;
            y[4 * q + 1] = v.y;
            y[4 * q + 2] = v.z;
            y[4 * q + 3] = v.w;
        }
    };
    // 8 MFMA steps of one group; lane (row r32, half kh) supplies individual 2 st + kh in step st
    auto multiply = [&](unsigned w, const float (&y)[16]) {
        const unsigned wk = (mok ? w : 0x55555555u) >> (2 * kh);
        bool has_nan = false;
#pragma unroll
        for (int q = 0; q < 16; q++) has_nan = has_nan || (y[q] != y[q]);
        const bool any_nan = __ballot(has_nan && tok) != 0ull;
#pragma unroll
        for (int st = 0; st < 8; st++)
        {
            const unsigned code = (wk >> (4 * st)) & 3u;           // 00 -> 2, 01 -> missing, 10 -> 1, 11 -> 0
            const float av = (float)((0xdu >> code) & 1u);         // non-missing flag
            const float ag = (float)((0x12u >> (2 * code)) & 3u);  // dosage, 0 when missing
            const float yq = kh ? y[2 * st + 1] : y[2 * st];
            const float by = (yq != yq) ? 0.0f : yq;
            sv += av;
            acc_gy = __builtin_amdgcn_mfma_f32_32x32x2f32(ag, by, acc_gy, 0, 0, 0);
            acc_vy = __builtin_amdgcn_mfma_f32_32x32x2f32(av, by, acc_vy, 0, 0, 0);
        }
        // the NaN count product only for groups that contain a NaN, in its own loop: a branch around the
        // products above would make the compiler copy the 32 accumulator registers at every join
        if (any_nan)
        {
#pragma unroll
            for (int st = 0; st < 8; st++)
            {
                const unsigned code = (wk >> (4 * st)) & 3u;
                const float av = (float)((0xdu >> code) & 1u);
                const float yq = kh ? y[2 * st + 1] : y[2 * st];
                acc_vn = __builtin_amdgcn_mfma_f32_32x32x2f32(av, (yq != yq) ? 1.0f : 0.0f, acc_vn, 0, 0, 0);
            }
        }
    };

    // the wave's range of 16-individual groups
    const size_t groups = (N + 15) / 16;
    const size_t gper = (groups + kMxpWaves - 1) / kMxpWaves;
    const size_t g_begin = min(groups, (size_t)wave * gper), g_end = min(groups, g_begin + gper);
    size_t g_fast_end = g_begin;
    if constexpr (FAST)
    {
        g_fast_end = max(g_begin, min(g_end, N / 16));  // full groups of this wave
        if (g_begin < g_fast_end)
        {
            unsigned w0, w1;
            float y0[16], y1[16];
            fetch_fast(g_begin, w0, y0);
            for (size_t g = g_begin; g < g_fast_end; g += 2)
            {
                // unconditional request of the following group (clamped), then the products of the older set
                fetch_fast(min(g + 1, g_fast_end - 1), w1, y1);
                __builtin_amdgcn_sched_barrier(0);
                multiply(w0, y0);
                if (g + 1 >= g_fast_end) break;
                fetch_fast(min(g + 2, g_fast_end - 1), w0, y0);
                __builtin_amdgcn_sched_barrier(0);
                multiply(w1, y1);
            }
        }
    }
    for (size_t g = g_fast_end; g < g_end; g++)
    {
        unsigned w;
        float y[16];
        fetch_checked(g, w, y);
        multiply(w, y);
    }
    // ---- split-K reduction in a fixed order: waves 4..7 -> 0..3, then 2,3 -> 0,1, then 1 -> 0 ----
    s_sv[wave][lane] = sv;
    for (int half = kMxpWaves / 2; half >= 1; half >>= 1)
    {
        __syncthreads();
        if (wave >= half && wave < 2 * half)
        {
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                s_part[wave - half][0][r][lane] = acc_gy[r];
                s_part[wave - half][1][r][lane] = acc_vy[r];
                s_part[wave - half][2][r][lane] = acc_vn[r];
            }
        }
        __syncthreads();
        if (wave < half)
        {
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                acc_gy[r] += s_part[wave][0][r][lane];
                acc_vy[r] += s_part[wave][1][r][lane];
                acc_vn[r] += s_part[wave][2][r][lane];
            }
        }
    }
    if (wave != 0) return;
    // non-missing count per marker: both k halves of all waves, fixed order
    float svm = 0.0f;
#pragma unroll
    for (int wv = 0; wv < kMxpWaves; wv++) svm += s_sv[wv][r32] + s_sv[wv][r32 + 32];
    // C/D layout: col (trait) = lane & 31, row (marker) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    s_sv[0][lane] = svm;  // lanes i and i+32 hold the same value for marker i
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const size_t t = p0 + r32;
#pragma unroll
    for (int r = 0; r < 16; r++)
    {
        const int il = (r & 3) + 8 * (r >> 2) + 4 * kh;
        const size_t mi = tile32 * 32 + il;
        if (mi < m && tok)
        {
            const float cnt = s_sv[0][il] - acc_vn[r];
            const float rr = (acc_gy[r] - mean[mi] * acc_vy[r]) / (cnt * sd[mi]);
            if (C)
            {
                C[mi * n + m + t] = rr;
                C[(m + t) * n + mi] = rr;
            }
            if (mxp) mxp[mi * p + t] = rr;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the same three sums on the bf16 matrix pipe (16x the rate of the f32 form above, whose 8 + 8 K = 2 products
// per 16 individuals bound the kernel at 0.41 ms for 13,700 markers x 20 traits -- as long as the 64 x more work of the
// SNP x SNP kernel of a batch).  A trait value is split EXACTLY into three bf16 pieces, y = hi + mid + lo (8 + 8 + 8
// significant bits), the dosage {0, 1, 2} and the flags {0, 1} are bf16 numbers anyway, so every product is exact in f32
// and only the order of the f32 additions differs from the chain above (the reference's own order is that of its block
// reduction, corr_kernels.cu:157-238; the bar is its test tolerance, 1e-5: tests/test_gpu_parity.py).
//   B fragments (mxp_split_kernel, once per build): bf16 columns c = q * 21 + t for piece q of trait t (columns 0..62),
//     64 + t = NaN flag of trait t, 85 = 1 for every individual < N; zero beyond N (padded to 64 individuals), so the
//     padding bits of a .bed row never count whatever they decode to; stored fragment-major (see the kernel).
//   Workgroup = 4 waves (one per SIMD, so that a wave may hold 160 accumulator registers) = split-K over the individuals,
//     TWO marker tiles of 32 per wave (they share the B fragments:
//     half the L2 traffic; with a tile table the two tiles may belong to different blocks of a batch -- the traits are the
//     same for all).  Per 16 individuals and tile: G x P0, G x P1, V x P0, V x P1, V x F = five
//     v_mfma_f32_32x32x16_bf16.  A fragment of a lane (marker r32, half kh): individuals 8 kh .. 8 kh + 7 = two bytes of
//     the row; a nibble (two 2-bit codes) becomes one dword of two bf16 through v_perm_b32 with the code as byte selector
//     into a 4-entry table (low bytes in one source, high bytes in the other): 6 vector instructions per dword pair (g, v).
//   The four partial tiles are added in a fixed order through LDS (deterministic), then every thread finishes a few
//   (marker, trait) pairs: pieces summed (hi + mid) + lo, r = (sum g y - mean_g sum y) / (n sd_g).
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 mxb_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned mxb_u4 __attribute__((ext_vector_type(4)));
constexpr int kMxbTraits = 21;  // traits per launch: 3 x 21 piece columns <= 64, 21 flags + the ones column <= 32
constexpr int kMxbCols = 96;
constexpr int kMxbOnes = 64 + kMxbTraits;
constexpr int kMxbWaves = 4;  // one wave per SIMD: two marker tiles x five products = 160 accumulator registers per wave

// One thread per (column tile k, group g, lane): the 16 bytes that lane (column k * 32 + (lane & 31), half lane >> 5) of a
// wave feeds to the MFMA of group g -- eight consecutive individuals of one column -- stored at ((k * G + g) * 64 + lane) * 8,
// so a wave's fragment load is one contiguous kilobyte.  G = groups, a multiple of four (the kernel requests four at a time).
__global__ void __launch_bounds__(256) mxp_split_kernel(const float *__restrict__ phen, size_t N, size_t G, size_t p0, int pc,
                                                         unsigned short *__restrict__ Bq)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= 3 * G * 64) return;
    const int lane = (int)(id & 63);
    const size_t g = (id >> 6) % G;
    const int k = (int)((id >> 6) / G);
    const int c = k * 32 + (lane & 31);
    const size_t i0 = g * 16 + 8 * (size_t)(lane >> 5);
    // column c: piece q of trait t (c = q * 21 + t < 63), NaN flag of trait c - 64, the ones column, or nothing
    const int q = c < 63 ? c / kMxbTraits : -1;
    const int t = c < 63 ? c % kMxbTraits : c - 64;
    unsigned short out[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
    {
        const size_t i = i0 + j;
        unsigned short v = 0;
        if (i < N)
        {
            if (c == kMxbOnes)
                v = 0x3f80;  // 1.0
            else if (c != 63 && c < kMxbOnes && t < pc)
            {
                const float y = phen[(p0 + t) * N + i];
                if (q < 0)
                    v = (y != y) ? 0x3f80 : 0;
                else if (y == y)
                {
                    const float hi = __uint_as_float(__float_as_uint(y) & 0xffff0000u);
                    const float r1 = (hi - hi == 0.0f) ? y - hi : 0.0f;  // exact; an infinite value stays in the first piece
                    const float mid = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
                    const float r2 = r1 - mid;  // exact, at most 8 significant bits
                    v = (unsigned short)(__float_as_uint(q == 0 ? hi : (q == 1 ? mid : r2)) >> 16);
                }
            }
        }
        out[j] = v;
    }
    mxb_u4 o;
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = (unsigned)out[2 * j] | ((unsigned)out[2 * j + 1] << 16);
    *reinterpret_cast<mxb_u4 *>(Bq + id * 8) = o;
}

struct MxbTile
{
    long long c_off;    // element offset of the block's matrix inside C
    long long mxp_off;  // element offset of the block's rows inside mxp
    long long row0;     // first marker row of the block in the .bed / mean / sd arrays
    int m, tile32, valid, pad;
};

template <bool FAST>
__global__ void __launch_bounds__(64 * kMxbWaves) mxp_bf16_kernel(const unsigned char *__restrict__ bed, const unsigned short *__restrict__ Bq,
                                                        const float *__restrict__ mean, const float *__restrict__ sd, float *C,
                                                        float *mxp, size_t m_in, size_t G, size_t p, size_t clb, size_t n,
                                                        size_t p0, int pcount, int ntiles, const int2 *__restrict__ mtile,
                                                        const CorrBatchBlock *__restrict__ bblk)
{
    __shared__ float s_red[kMxbWaves][16][64];  // the partial copies of one accumulator tile
    __shared__ float s_out[2][5][32][33];  // reduced tiles: [marker tile][product][marker][column]
    __shared__ MxbTile s_tile[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, kh = lane >> 5;
    if (tid < 2)
    {
        MxbTile t;
        const int T = 2 * (int)blockIdx.x + tid;
        t.valid = T < ntiles ? 1 : 0;
        t.c_off = 0;
        t.mxp_off = 0;
        t.row0 = 0;
        t.m = (int)m_in;
        t.tile32 = T;
        t.pad = 0;
        if (t.valid && mtile != nullptr)
        {  // batched build: tile mtile[T].y of block mtile[T].x
            const int2 mt = mtile[T];
            const CorrBatchBlock bb = bblk[mt.x];
            t.row0 = (long long)bb.bed_row0;
            t.c_off = (long long)bb.base * (long long)n + bb.base;
            t.mxp_off = (long long)bb.mxp_off * (long long)p;
            t.m = bb.m;
            t.tile32 = mt.y;
        }
        s_tile[tid] = t;
    }
    __syncthreads();
    const unsigned char *rowp[2];
    bool mok[2];
#pragma unroll
    for (int u = 0; u < 2; u++)
    {
        const MxbTile t = s_tile[u];
        const long long mk = (long long)t.tile32 * 32 + r32;
        mok[u] = t.valid && mk < t.m;
        rowp[u] = bed + (size_t)(t.row0 + (mok[u] ? mk : 0)) * clb;
    }
    v16f acc[2][5];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[u][a][r] = 0.0f;

    // the wave's range of groups, in fours (G is a multiple of four)
    const size_t gper = ((G / 4 + kMxbWaves - 1) / kMxbWaves) * 4;
    const size_t g_begin = min(G, (size_t)wave * gper), g_end = min(G, g_begin + gper);
    const unsigned short *bq = Bq + (size_t)lane * 8;
    // FAST (rows of the .bed block 16-byte aligned): the four dwords of four consecutive groups in one request per marker
    // row -- a request of 64 lanes in 64 different rows costs the address unit one cycle per lane whatever its width
    auto fetch_a4 = [&](int u, size_t g0, unsigned (&w)[4]) {
        if constexpr (FAST)
        {
            const mxb_u4 v = reinterpret_cast<const mxb_u4 *>(rowp[u])[g0 >> 2];
            w[0] = v[0], w[1] = v[1], w[2] = v[2], w[3] = v[3];
        }
        else
        {
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                unsigned x = 0u;
                for (size_t bb = 0; bb < 4; bb++) x |= (unsigned)(((g0 + j) * 4 + bb < clb) ? rowp[u][(g0 + j) * 4 + bb] : 0x55u) << (8 * bb);
                w[j] = x;
            }
        }
    };
    auto fetch_b = [&](size_t g, mxb_u4 (&b)[3]) {
#pragma unroll
        for (int k = 0; k < 3; k++) b[k] = *reinterpret_cast<const mxb_u4 *>(bq + ((size_t)k * G + g) * 512);
    };
    // eight 2-bit codes -> eight bf16 dosages and eight bf16 non-missing flags
    // code 00 -> dosage 2 (0x4000), 01 -> missing (0, flag 0), 10 -> 1 (0x3f80), 11 -> 0; flag 1.0 = 0x3f80
    auto decode = [&](unsigned x, mxb_u4 &ag, mxb_u4 &av) {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const unsigned nib = x >> (4 * j);
            const unsigned t = (nib & 3u) | ((nib & 0xcu) << 14);  // c0 | c1 << 16
            const unsigned sel = t * 0x0101u + 0x04000400u;        // bytes [c0, 4 + c0, c1, 4 + c1]: low byte, high byte per value
            ag[j] = __builtin_amdgcn_perm(0x003f0040u, 0x00800000u, sel);
            av[j] = __builtin_amdgcn_perm(0x3f3f003fu, 0x80800080u, sel);
        }
    };
    auto multiply = [&](const unsigned (&w)[2], const mxb_u4 (&b)[3]) {
        const mxb_bf8 b0 = __builtin_bit_cast(mxb_bf8, b[0]), b1 = __builtin_bit_cast(mxb_bf8, b[1]), b2 = __builtin_bit_cast(mxb_bf8, b[2]);
#pragma unroll
        for (int u = 0; u < 2; u++)
        {
            const unsigned x = mok[u] ? ((w[u] >> (16 * kh)) & 0xffffu) : 0x5555u;
            mxb_u4 ag, av;
            decode(x, ag, av);
            const mxb_bf8 g8 = __builtin_bit_cast(mxb_bf8, ag), v8 = __builtin_bit_cast(mxb_bf8, av);
            acc[u][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g8, b0, acc[u][0], 0, 0, 0);
            acc[u][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g8, b1, acc[u][1], 0, 0, 0);
            acc[u][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v8, b0, acc[u][2], 0, 0, 0);
            acc[u][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v8, b1, acc[u][3], 0, 0, 0);
            acc[u][4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v8, b2, acc[u][4], 0, 0, 0);
        }
    };
    if (g_begin < g_end)
    {
        // One loop body, four groups (64 individuals) per trip: the operands of the following four are requested
        // (clamped, unconditional) before the forty products of the current four and handed over by register moves.  (An
        // exit in the middle of an unrolled pair makes the compiler keep two copies of the 160 accumulator registers and
        // move them at every join.)  The first form asked for one dword per lane and group and for B columns N apart: five
        // requests of 64 scattered addresses per group and wave kept the address unit busier than the products kept the
        // matrix pipe (184 us); now two scattered 16-byte requests and twelve contiguous kilobytes per four groups.
        unsigned wc[2][4], wn[2][4];
        mxb_u4 bc[4][3], bn[4][3];
        auto fetch4 = [&](size_t g0, unsigned (&w)[2][4], mxb_u4 (&bb)[4][3]) {
            fetch_a4(0, g0, w[0]);
            fetch_a4(1, g0, w[1]);
#pragma unroll
            for (int j = 0; j < 4; j++) fetch_b(g0 + j, bb[j]);
        };
        fetch4(g_begin, wc, bc);
        for (size_t g = g_begin; g < g_end; g += 4)
        {
            fetch4(min(g + 4, g_end - 4), wn, bn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const unsigned wj[2] = {wc[0][j], wc[1][j]};
                multiply(wj, bc[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                wc[0][j] = wn[0][j];
                wc[1][j] = wn[1][j];
#pragma unroll
                for (int kk = 0; kk < 3; kk++) bc[j][kk] = bn[j][kk];
            }
        }
    }
    // ---- split-K reduction, tile by tile, waves 0, 1, ... in that order ----
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int a = 0; a < 5; a++)
        {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; r++) s_red[wave][r][lane] = acc[u][a][r];
            __syncthreads();
            for (int e = tid; e < 16 * 64; e += 64 * kMxbWaves)
            {
                const int r = e >> 6, l = e & 63;
                float v = s_red[0][r][l];
#pragma unroll
                for (int wv = 1; wv < kMxbWaves; wv++) v += s_red[wv][r][l];
                // C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
                s_out[u][a][(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31] = v;
            }
        }
    __syncthreads();
    // ---- every thread finishes a few (tile, marker, trait) triples ----
    for (int idx = tid; idx < 2 * 32 * pcount; idx += 64 * kMxbWaves)
    {
        const int t = idx % pcount, i = (idx / pcount) & 31, u = idx / (pcount * 32);
        const MxbTile tt = s_tile[u];
        const long long mi = (long long)tt.tile32 * 32 + i;
        if (!tt.valid || mi >= tt.m) continue;
        float sgy[3], svy[3];
#pragma unroll
        for (int q = 0; q < 3; q++)
        {
            const int c = q * kMxbTraits + t;
            sgy[q] = s_out[u][c >> 5][i][c & 31];
            svy[q] = s_out[u][2 + (c >> 5)][i][c & 31];
        }
        const float gy = (sgy[0] + sgy[1]) + sgy[2], vy = (svy[0] + svy[1]) + svy[2];
        const float cnt = s_out[u][4][i][kMxbOnes - 64] - s_out[u][4][i][t];
        const size_t mg = (size_t)(tt.row0 + mi);
        const float rr = (gy - mean[mg] * vy) / (cnt * sd[mg]);
        const size_t tg = p0 + (size_t)t;
        if (C)
        {
            float *Cb = C + tt.c_off;
            Cb[(size_t)mi * n + (size_t)tt.m + tg] = rr;
            Cb[((size_t)tt.m + tg) * n + (size_t)mi] = rr;
        }
        if (mxp) mxp[(size_t)tt.mxp_off + (size_t)mi * p + tg] = rr;
    }
}

// one workgroup per trait pair (a < b): corr_kernels.cu:285-343
__global__ void __launch_bounds__(256) pxp_kernel(const float *__restrict__ phen, float *C, size_t m, size_t N, size_t p,
                                                   size_t n)
{
    __shared__ float ss[256], sc[256];
    size_t lin = blockIdx.x, a = 0, len = p - 1;
    while (lin >= len)
    {
        lin -= len;
        len--;
        a++;
    }
    const size_t b = a + 1 + lin;
    float s = 0.0f, c = 0.0f;
    for (size_t i = threadIdx.x; i < N; i += 256)
    {
        const float va = phen[a * N + i], vb = phen[b * N + i];
        if (!((va != va) || (vb != vb)))
        {
            s += va * vb;
            c += 1.0f;
        }
    }
    ss[threadIdx.x] = s;
    sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1)
    {
        if ((int)threadIdx.x < o)
        {
            ss[threadIdx.x] += ss[threadIdx.x + o];
            sc[threadIdx.x] += sc[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        const float r = ss[0] / sc[0];
        C[(m + a) * n + m + b] = r;
        C[(m + b) * n + m + a] = r;
    }
}

__global__ void unit_diag_kernel(float *C, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) C[i * n + i] = 1.0f;
}

// upper triangle (without diagonal) of the rows/cols [lo, lo+cnt) of C into a linear array
__global__ void extract_tri_kernel(const float *__restrict__ C, float *out, size_t lo, size_t cnt, size_t n)
{
    const size_t i = blockIdx.y;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt || j >= cnt || j <= i) return;
    const size_t lin = i * (cnt - 1) - (i * (i - 1)) / 2 + (j - i - 1);
    out[lin] = C[(lo + i) * n + lo + j];
}

// true when p is a HIP device allocation (hipPointerGetAttributes fails for plain host memory: not an error here)
static bool is_device_pointer(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess)
    {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice;
}

// `ahead`: the build of the NEXT block, enqueued on the engine's third stream while the current block is swept
// (cusk_corr_build_begin): device-resident inputs only, no events, no synchronisation; the marker x trait correlations
// land in `mxp_host` (pinned) when the stream gets there.
int corr_build_impl(cusk_engine *e, const unsigned char *bed, const float *phen, size_t m, size_t N, size_t p,
                    const float *mean, const float *std, float *C_dev, float *mxp_host, float *mxm_tri_host,
                    float *pxp_tri_host, bool ahead)
{
    if (!e || !bed || !phen || !mean || !std || m == 0 || N == 0) return fail(e, CUSK_ERR_ARG, "bad arguments");
    CUSK_HIP(e, hipSetDevice(e->device));
    hipStream_t s = ahead ? e->stream3 : e->stream;
    if (ahead && !(is_device_pointer(bed) && is_device_pointer(phen) && is_device_pointer(mean) && is_device_pointer(std)))
        return fail(e, CUSK_ERR_ARG, "cusk_corr_build_begin needs device-resident inputs");
    const size_t clb = (N + 3) / 4, w64 = (N + 63) / 64, n = m + p;
    // Inputs that already live in HBM (a whole .bed staged once per GPU by the block driver, cusk_blockset_stage)
    // are used where they are; host inputs are copied to engine scratch first.
    const unsigned char *bed_d = nullptr;
    const float *phen_d = nullptr, *mean_d = nullptr, *std_d = nullptr;
    if (!ahead) CUSK_HIP(e, hipEventRecord(e->ev_corr[0], s));
    if (is_device_pointer(bed))
        bed_d = bed;
    else
    {
        CUSK_HIP(e, e->bed_dev.ensure(m * clb));
        CUSK_HIP(e, hipMemcpyAsync(e->bed_dev.p, bed, m * clb, hipMemcpyHostToDevice, s));
        bed_d = e->bed_dev.as<unsigned char>();
    }
    if (is_device_pointer(phen))
        phen_d = phen;
    else
    {
        CUSK_HIP(e, e->phen_dev.ensure(sizeof(float) * std::max<size_t>(p * N, 1)));
        if (p) CUSK_HIP(e, hipMemcpyAsync(e->phen_dev.p, phen, sizeof(float) * p * N, hipMemcpyHostToDevice, s));
        phen_d = e->phen_dev.as<float>();
    }
    if (is_device_pointer(mean))
        mean_d = mean;
    else
    {
        CUSK_HIP(e, e->mean_dev.ensure(sizeof(float) * m));
        CUSK_HIP(e, hipMemcpyAsync(e->mean_dev.p, mean, sizeof(float) * m, hipMemcpyHostToDevice, s));
        mean_d = e->mean_dev.as<float>();
    }
    if (is_device_pointer(std))
        std_d = std;
    else
    {
        CUSK_HIP(e, e->std_dev.ensure(sizeof(float) * m));
        CUSK_HIP(e, hipMemcpyAsync(e->std_dev.p, std, sizeof(float) * m, hipMemcpyHostToDevice, s));
        std_d = e->std_dev.as<float>();
    }
    float *mxp_d = nullptr;
    if (mxp_host && p)
    {
        CUSK_HIP(e, e->mxp_dev.ensure(sizeof(float) * m * p));
        mxp_d = e->mxp_dev.as<float>();
    }
    if (!ahead) CUSK_HIP(e, hipEventRecord(e->ev_corr[1], s));
    if (C_dev && e->opt_corr_popcount)
    {
        // cross-check path: bit planes + AND/popcount (no matrix cores)
        CUSK_HIP(e, e->planes.ensure(sizeof(unsigned long long) * 3 * m * w64));
        hipLaunchKernelGGL(bed_to_bitplanes_kernel, dim3((unsigned)((m * w64 + 255) / 256)), dim3(256), 0, s,
                           bed_d, e->planes.as<unsigned long long>(), m, N, clb, w64);
        if (!ahead) CUSK_HIP(e, hipEventRecord(e->ev_corr[2], s));
        const int tiles = (int)((m + kTile - 1) / kTile);
        const long long nt = (long long)tiles * (tiles + 1) / 2;
        hipLaunchKernelGGL(mxm_popcount_kernel, dim3((unsigned)nt), dim3(256), 0, s, e->planes.as<unsigned long long>(),
                           C_dev, m, w64, n, tiles);
        hipLaunchKernelGGL(unit_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, C_dev, n);
    }
    else if (C_dev)
    {
        if (!ahead) CUSK_HIP(e, hipEventRecord(e->ev_corr[2], s));
        const int tiles = (int)((m + kMT - 1) / kMT);
        const long long nt = (long long)tiles * (tiles + 1) / 2;
        const bool rows16 = (clb % 16 == 0) && ((reinterpret_cast<uintptr_t>(bed_d) & 15u) == 0);
        if (e->opt_corr_fp4 && rows16)
            hipLaunchKernelGGL(mxm_fp4_kernel<true>, dim3((unsigned)nt), dim3(256), 0, s, bed_d, C_dev,
                               m, N, clb, n, tiles, (size_t)0, (const int4 *)nullptr, (const CorrBatchBlock *)nullptr);
        else if (e->opt_corr_fp4)
            hipLaunchKernelGGL(mxm_fp4_kernel<false>, dim3((unsigned)nt), dim3(256), 0, s, bed_d, C_dev,
                               m, N, clb, n, tiles, (size_t)0, (const int4 *)nullptr, (const CorrBatchBlock *)nullptr);
        else
            hipLaunchKernelGGL(mxm_mfma_kernel, dim3((unsigned)nt), dim3(256), 0, s, bed_d, C_dev, m,
                               N, clb, n, tiles);
        hipLaunchKernelGGL(unit_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, C_dev, n);
    }
    else
    {
        if (!ahead) CUSK_HIP(e, hipEventRecord(e->ev_corr[2], s));
    }
    if (!ahead) CUSK_HIP(e, hipEventRecord(e->ev_corr[3], s));
    const bool mxp_bf16 = !e->opt_corr_popcount && !e->opt_corr_mxp_f32 && p > 0;
    if (mxp_bf16)
    {
        // the bf16 form: exact three-piece split of the traits, 21 traits per launch
        const size_t G = (N + 63) / 64 * 4;  // groups of 16 individuals, in fours
        CUSK_HIP(e, e->mxp_bq.ensure(sizeof(unsigned short) * kMxbCols * G * 16));
        const bool fast = (clb % 16 == 0) && ((reinterpret_cast<uintptr_t>(bed_d) & 15u) == 0);
        const int ntiles = (int)((m + 31) / 32);
        for (size_t p0 = 0; p0 < p; p0 += kMxbTraits)
        {
            const int pc = (int)std::min<size_t>(kMxbTraits, p - p0);
            hipLaunchKernelGGL(mxp_split_kernel, dim3((unsigned)((3 * G * 64 + 255) / 256)), dim3(256), 0, s, phen_d, N, G, p0, pc,
                               e->mxp_bq.as<unsigned short>());
            if (fast)
                hipLaunchKernelGGL(mxp_bf16_kernel<true>, dim3((unsigned)((ntiles + 1) / 2)), dim3(64 * kMxbWaves), 0, s, bed_d,
                                   e->mxp_bq.as<unsigned short>(), mean_d, std_d, C_dev, mxp_d, m, G, p, clb, n, p0, pc, ntiles,
                                   (const int2 *)nullptr, (const CorrBatchBlock *)nullptr);
            else
                hipLaunchKernelGGL(mxp_bf16_kernel<false>, dim3((unsigned)((ntiles + 1) / 2)), dim3(64 * kMxbWaves), 0, s, bed_d,
                                   e->mxp_bq.as<unsigned short>(), mean_d, std_d, C_dev, mxp_d, m, G, p, clb, n, p0, pc, ntiles,
                                   (const int2 *)nullptr, (const CorrBatchBlock *)nullptr);
        }
    }
    for (size_t p0 = 0; p0 < p && !mxp_bf16; p0 += kMaxPhenRegs)
    {
        const size_t pc = std::min<size_t>(kMaxPhenRegs, p - p0);
        if (e->opt_corr_popcount)
            hipLaunchKernelGGL(mxp_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s, bed_d,
                               phen_d, mean_d, std_d, C_dev, mxp_d, m, N,
                               p, clb, n, p0, pc);
        else
        {
            const bool fast = (clb % 4 == 0) && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(bed_d) & 3u) == 0) &&
                              ((reinterpret_cast<uintptr_t>(phen_d) & 15u) == 0);
            if (fast)
                hipLaunchKernelGGL(mxp_mfma_kernel<true>, dim3((unsigned)((m + 31) / 32)), dim3(64 * kMxpWaves), 0, s,
                                   bed_d, phen_d, mean_d,
                                   std_d, C_dev, mxp_d, m, N, p, clb, n, p0, (int)pc, (const int2 *)nullptr,
                                   (const CorrBatchBlock *)nullptr);
            else
                hipLaunchKernelGGL(mxp_mfma_kernel<false>, dim3((unsigned)((m + 31) / 32)), dim3(64 * kMxpWaves), 0, s,
                                   bed_d, phen_d, mean_d,
                                   std_d, C_dev, mxp_d, m, N, p, clb, n, p0, (int)pc, (const int2 *)nullptr,
                                   (const CorrBatchBlock *)nullptr);
        }
    }
    if (C_dev && p > 1)
        hipLaunchKernelGGL(pxp_kernel, dim3((unsigned)(p * (p - 1) / 2)), dim3(256), 0, s, phen_d, C_dev, m,
                           N, p, n);
    CUSK_HIP(e, hipGetLastError());
    if (!ahead) CUSK_HIP(e, hipEventRecord(e->ev_corr[4], s));
    if (mxp_d) CUSK_HIP(e, hipMemcpyAsync(mxp_host, mxp_d, sizeof(float) * m * p, hipMemcpyDeviceToHost, s));
    if (ahead) return CUSK_OK;  // cusk_corr_build_end waits for the stream
    if (C_dev && (mxm_tri_host || pxp_tri_host))
    {
        DevBuf tri;
        const size_t cm = m * (m - 1) / 2, cp = p ? p * (p - 1) / 2 : 0;
        CUSK_HIP(e, tri.ensure(sizeof(float) * std::max<size_t>(std::max(cm, cp), 1)));
        if (mxm_tri_host && cm)
        {
            hipLaunchKernelGGL(extract_tri_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, s, C_dev,
                               tri.as<float>(), (size_t)0, m, n);
            CUSK_HIP(e, hipMemcpyAsync(mxm_tri_host, tri.p, sizeof(float) * cm, hipMemcpyDeviceToHost, s));
            CUSK_HIP(e, hipStreamSynchronize(s));
        }
        if (pxp_tri_host && cp)
        {
            hipLaunchKernelGGL(extract_tri_kernel, dim3((unsigned)((p + 255) / 256), (unsigned)p), dim3(256), 0, s, C_dev,
                               tri.as<float>(), m, p, n);
            CUSK_HIP(e, hipMemcpyAsync(pxp_tri_host, tri.p, sizeof(float) * cp, hipMemcpyDeviceToHost, s));
            CUSK_HIP(e, hipStreamSynchronize(s));
        }
        tri.release();
    }
    CUSK_HIP(e, hipStreamSynchronize(s));
    float ms = 0;
    CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev_corr[1], e->ev_corr[2]));
    e->corr_ms[0] = ms;
    CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev_corr[2], e->ev_corr[3]));
    e->corr_ms[1] = ms;
    CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev_corr[3], e->ev_corr[4]));
    e->corr_ms[2] = ms;
    CUSK_HIP(e, hipEventElapsedTime(&ms, e->ev_corr[0], e->ev_corr[4]));
    e->corr_ms[3] = ms;
    return CUSK_OK;
}

// ---------------------------------------------------------------------------
// batched build: the correlation matrices of many LD blocks in one set of launches (cusk_corr_build_batch_*)
// ---------------------------------------------------------------------------

// per block: the trait x trait correlations (computed once, the same for every block: cli.cpp:597-649 writes them into
// every block's matrix) and the unit diagonal
__global__ void __launch_bounds__(256) batch_finish_kernel(float *C, size_t n, const CorrBatchBlock *__restrict__ bblk,
                                                            const int *__restrict__ kept, const float *__restrict__ pxp, int p)
{
    const CorrBatchBlock bb = bblk[kept[blockIdx.x]];
    float *Cb = C + (size_t)bb.base * n + bb.base;
    const int nb = bb.m + p;
    for (int i = threadIdx.x; i < nb; i += 256) Cb[(size_t)i * n + i] = 1.0f;
    for (int e = threadIdx.x; e < p * p; e += 256)
    {
        const int a = e / p, b = e % p;
        if (a != b) Cb[(size_t)(bb.m + a) * n + bb.m + b] = pxp[(size_t)a * p + b];
    }
}

// Phase one: marker x trait correlations of every block (into the batch allocation and into a compact host array, block
// after block: the prefilter of cli.cpp:561-576 is the caller's).  Waits for the result.
// Phase two: marker x marker tiles, trait x trait, diagonal of the blocks the caller keeps; asynchronous on the engine's
// stream (the sweep that follows is ordered behind it).
int corr_build_batch_impl(cusk_engine *e, int phase, const unsigned char *bed_dev, const float *phen_dev, const float *mean_dev,
                          const float *std_dev, size_t N, size_t p, int nblk, const long long *first_marker, const int *markers,
                          const int *base, const unsigned char *keep, size_t n, float *C_dev, float *mxp_host, bool async_mxp = false)
{
    if (!e || !bed_dev || !phen_dev || !mean_dev || !std_dev || !first_marker || !markers || !base || !C_dev || nblk <= 0 || N == 0)
        return fail(e, CUSK_ERR_ARG, "bad arguments");
    if (!(is_device_pointer(bed_dev) && is_device_pointer(phen_dev) && is_device_pointer(mean_dev) && is_device_pointer(std_dev)))
        return fail(e, CUSK_ERR_ARG, "cusk_corr_build_batch needs device-resident inputs (cusk_blockset_stage)");
    CUSK_HIP(e, hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const size_t clb = (N + 3) / 4;
    // tables: blocks, then the tiles of this phase
    std::vector<CorrBatchBlock> blk((size_t)nblk);
    long long moff = 0;
    for (int b = 0; b < nblk; b++)
    {
        if (markers[b] <= 0 || base[b] < 0 || (size_t)base[b] + (size_t)markers[b] + p > n)
            return fail(e, CUSK_ERR_ARG, "cusk_corr_build_batch: block outside the batch allocation");
        blk[(size_t)b] = CorrBatchBlock{(unsigned long long)first_marker[b], markers[b], base[b], moff};
        moff += markers[b];
    }
    const size_t b_blk = sizeof(CorrBatchBlock) * (size_t)nblk;
    std::vector<int4> tiles4;
    std::vector<int2> tiles2;
    std::vector<int> kept;
    if (phase == 1)
    {
        for (int b = 0; b < nblk; b++)
            for (int t = 0; t < (markers[b] + 31) / 32; t++) tiles2.push_back(make_int2(b, t));
    }
    else
    {
        for (int b = 0; b < nblk; b++)
        {
            if (keep && !keep[b]) continue;
            kept.push_back(b);
            const int tl = (markers[b] + kMT - 1) / kMT;
            for (int bi = 0; bi < tl; bi++)
                for (int bj = bi; bj < tl; bj++) tiles4.push_back(make_int4(b, bi, bj, 0));
        }
        if (kept.empty()) return CUSK_OK;
    }
    const size_t b_t = (phase == 1) ? sizeof(int2) * tiles2.size() : sizeof(int4) * tiles4.size();
    const size_t b_k = sizeof(int) * kept.size();
    const size_t o_t = (b_blk + 15) & ~(size_t)15, o_k = o_t + ((b_t + 15) & ~(size_t)15), total = o_k + ((b_k + 15) & ~(size_t)15) + 16;
    // one pinned staging area per phase: phase one waits for its stream at the end, phase two is followed by a run that does
    const int slot = phase == 1 ? 0 : 1;
    if (total > e->corr_tab_pinned_cap[slot])
    {
        if (e->corr_tab_pinned[slot]) (void)hipHostFree(e->corr_tab_pinned[slot]);
        e->corr_tab_pinned[slot] = nullptr;
        e->corr_tab_pinned_cap[slot] = 0;
        CUSK_HIP(e, hipHostMalloc(&e->corr_tab_pinned[slot], total * 2));
        e->corr_tab_pinned_cap[slot] = total * 2;
    }
    char *h = static_cast<char *>(e->corr_tab_pinned[slot]);
    std::memcpy(h, blk.data(), b_blk);
    if (phase == 1)
        std::memcpy(h + o_t, tiles2.data(), b_t);
    else
    {
        std::memcpy(h + o_t, tiles4.data(), b_t);
        std::memcpy(h + o_k, kept.data(), b_k);
    }
    CUSK_HIP(e, e->corr_tab[slot].ensure(total));
    CUSK_HIP(e, hipMemcpyAsync(e->corr_tab[slot].p, h, total, hipMemcpyHostToDevice, s));
    const char *d = e->corr_tab[slot].as<char>();
    const CorrBatchBlock *blk_d = reinterpret_cast<const CorrBatchBlock *>(d);
    if (phase == 1)
    {
        if (p == 0) return CUSK_OK;
        float *mxp_d = nullptr;
        if (mxp_host)
        {
            CUSK_HIP(e, e->mxp_dev.ensure(sizeof(float) * (size_t)moff * p));
            mxp_d = e->mxp_dev.as<float>();
        }
        const bool fast = (clb % 4 == 0) && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(bed_dev) & 3u) == 0) &&
                          ((reinterpret_cast<uintptr_t>(phen_dev) & 15u) == 0);
        const int2 *mt_d = reinterpret_cast<const int2 *>(d + o_t);
        const bool mxp_bf16 = !e->opt_corr_mxp_f32;
        if (mxp_bf16)
        {
            const size_t G = (N + 63) / 64 * 4;
            CUSK_HIP(e, e->mxp_bq.ensure(sizeof(unsigned short) * kMxbCols * G * 16));
            const bool fastb = (clb % 16 == 0) && ((reinterpret_cast<uintptr_t>(bed_dev) & 15u) == 0);
            const int ntiles = (int)tiles2.size();
            for (size_t p0 = 0; p0 < p; p0 += kMxbTraits)
            {
                const int pc = (int)std::min<size_t>(kMxbTraits, p - p0);
                hipLaunchKernelGGL(mxp_split_kernel, dim3((unsigned)((3 * G * 64 + 255) / 256)), dim3(256), 0, s, phen_dev, N, G, p0, pc,
                                   e->mxp_bq.as<unsigned short>());
                if (fastb)
                    hipLaunchKernelGGL(mxp_bf16_kernel<true>, dim3((unsigned)((ntiles + 1) / 2)), dim3(64 * kMxbWaves), 0, s, bed_dev,
                                       e->mxp_bq.as<unsigned short>(), mean_dev, std_dev, C_dev, mxp_d, (size_t)0, G, p, clb, n, p0, pc,
                                       ntiles, mt_d, blk_d);
                else
                    hipLaunchKernelGGL(mxp_bf16_kernel<false>, dim3((unsigned)((ntiles + 1) / 2)), dim3(64 * kMxbWaves), 0, s, bed_dev,
                                       e->mxp_bq.as<unsigned short>(), mean_dev, std_dev, C_dev, mxp_d, (size_t)0, G, p, clb, n, p0, pc,
                                       ntiles, mt_d, blk_d);
            }
        }
        for (size_t p0 = 0; p0 < p && !mxp_bf16; p0 += kMaxPhenRegs)
        {
            const size_t pc = std::min<size_t>(kMaxPhenRegs, p - p0);
            if (fast)
                hipLaunchKernelGGL(mxp_mfma_kernel<true>, dim3((unsigned)tiles2.size()), dim3(64 * kMxpWaves), 0, s, bed_dev, phen_dev,
                                   mean_dev, std_dev, C_dev, mxp_d, (size_t)0, N, p, clb, n, p0, (int)pc, mt_d, blk_d);
            else
                hipLaunchKernelGGL(mxp_mfma_kernel<false>, dim3((unsigned)tiles2.size()), dim3(64 * kMxpWaves), 0, s, bed_dev, phen_dev,
                                   mean_dev, std_dev, C_dev, mxp_d, (size_t)0, N, p, clb, n, p0, (int)pc, mt_d, blk_d);
        }
        CUSK_HIP(e, hipGetLastError());
        if (async_mxp)
        {  // cusk_corr_build_batch: the correlations land in pinned memory, an event marks their arrival, nobody waits here
            const size_t cnt = (size_t)moff * p;
            if (cnt > e->mxp_pinned_cap)
            {
                if (e->mxp_pending) return fail(e, CUSK_ERR_STATE, "a correlation build is in flight");
                if (e->mxp_pinned) (void)hipHostFree(e->mxp_pinned);
                e->mxp_pinned = nullptr;
                e->mxp_pinned_cap = 0;
                CUSK_HIP(e, hipHostMalloc(reinterpret_cast<void **>(&e->mxp_pinned), sizeof(float) * (cnt + cnt / 4 + 64)));
                e->mxp_pinned_cap = cnt + cnt / 4 + 64;
            }
            if (mxp_d) CUSK_HIP(e, hipMemcpyAsync(e->mxp_pinned, mxp_d, sizeof(float) * cnt, hipMemcpyDeviceToHost, s));
            if (!e->ev_mxp) CUSK_HIP(e, hipEventCreateWithFlags(&e->ev_mxp, hipEventDisableTiming));
            CUSK_HIP(e, hipEventRecord(e->ev_mxp, s));
            return CUSK_OK;
        }
        if (mxp_d) CUSK_HIP(e, hipMemcpyAsync(mxp_host, mxp_d, sizeof(float) * (size_t)moff * p, hipMemcpyDeviceToHost, s));
        CUSK_HIP(e, hipStreamSynchronize(s));
        return CUSK_OK;
    }
    const bool rows16 = (clb % 16 == 0) && ((reinterpret_cast<uintptr_t>(bed_dev) & 15u) == 0);
    const int4 *bt_d = reinterpret_cast<const int4 *>(d + o_t);
    if (rows16)
        hipLaunchKernelGGL(mxm_fp4_kernel<true>, dim3((unsigned)tiles4.size()), dim3(256), 0, s, bed_dev, C_dev, (size_t)0, N, clb, n, 0,
                           (size_t)0, bt_d, blk_d);
    else
        hipLaunchKernelGGL(mxm_fp4_kernel<false>, dim3((unsigned)tiles4.size()), dim3(256), 0, s, bed_dev, C_dev, (size_t)0, N, clb, n, 0,
                           (size_t)0, bt_d, blk_d);
    // trait x trait once (a p x p matrix of its own: m = 0, leading dimension p), then into every kept block with the diagonal
    CUSK_HIP(e, e->pxp_dev.ensure(sizeof(float) * std::max<size_t>(p * p, 1)));
    if (p > 1)
        hipLaunchKernelGGL(pxp_kernel, dim3((unsigned)(p * (p - 1) / 2)), dim3(256), 0, s, phen_dev, e->pxp_dev.as<float>(), (size_t)0, N, p,
                           p);
    hipLaunchKernelGGL(batch_finish_kernel, dim3((unsigned)kept.size()), dim3(256), 0, s, C_dev, n, blk_d,
                       reinterpret_cast<const int *>(d + o_k), e->pxp_dev.as<float>(), (int)p);
    CUSK_HIP(e, hipGetLastError());
    return CUSK_OK;
}

// forward row sums of |band|: one thread per row, float accumulation in column order (the reference's host loop,
// corr_host.cu:112-128, so the sums are bit-identical)
__global__ void band_row_abs_sums_kernel(const float *__restrict__ band, size_t m, size_t w, float *sums)
{
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    float acc = 0.0f;
    for (size_t c = 0; c < w; c++) acc += fabsf(band[row * w + c]);
    sums[row] = acc;
}

// cal_mcorrk_banded + marker_corr_banded_mat_row_abs_sums (corr_host.cu:65-128) for one chromosome
int corr_banded_impl(cusk_engine *e, const unsigned char *bed, size_t m, size_t N, size_t width, float *rowsums_host,
                     float *band_host)
{
    if (!e || !bed || !rowsums_host || m == 0 || N == 0 || width == 0) return fail(e, CUSK_ERR_ARG, "bad arguments");
    CUSK_HIP(e, hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const size_t clb = (N + 3) / 4;
    CUSK_HIP(e, e->bed_dev.ensure(m * clb));
    DevBuf band, sums;
    CUSK_HIP(e, band.ensure(sizeof(float) * m * width));
    CUSK_HIP(e, sums.ensure(sizeof(float) * m));
    CUSK_HIP(e, hipMemcpyAsync(e->bed_dev.p, bed, m * clb, hipMemcpyHostToDevice, s));
    CUSK_HIP(e, hipMemsetAsync(band.p, 0, sizeof(float) * m * width, s));  // pairs past the last marker stay 0
    const int tiles = (int)((m + kMT - 1) / kMT);
    const unsigned ndj = (unsigned)((width + kMT - 1) / kMT + 1);
    const bool rows16 = (clb % 16 == 0) && ((reinterpret_cast<uintptr_t>(e->bed_dev.p) & 15u) == 0);
    if (rows16)
        hipLaunchKernelGGL(mxm_fp4_kernel<true>, dim3((unsigned)tiles, ndj), dim3(256), 0, s, e->bed_dev.as<unsigned char>(),
                           band.as<float>(), m, N, clb, m, tiles, width, (const int4 *)nullptr, (const CorrBatchBlock *)nullptr);
    else
        hipLaunchKernelGGL(mxm_fp4_kernel<false>, dim3((unsigned)tiles, ndj), dim3(256), 0, s, e->bed_dev.as<unsigned char>(),
                           band.as<float>(), m, N, clb, m, tiles, width, (const int4 *)nullptr, (const CorrBatchBlock *)nullptr);
    hipLaunchKernelGGL(band_row_abs_sums_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, band.as<float>(), m, width,
                       sums.as<float>());
    CUSK_HIP(e, hipGetLastError());
    CUSK_HIP(e, hipMemcpyAsync(rowsums_host, sums.p, sizeof(float) * m, hipMemcpyDeviceToHost, s));
    if (band_host) CUSK_HIP(e, hipMemcpyAsync(band_host, band.p, sizeof(float) * m * width, hipMemcpyDeviceToHost, s));
    CUSK_HIP(e, hipStreamSynchronize(s));
    band.release();
    sums.release();
    return CUSK_OK;
}

// Hanning smoothing of the row-sum profile (blocking.cpp:13-35): one thread per centre, the window walked in
// order with separate double multiply and add (this file is compiled -ffp-contract=off), so every value equals
// the reference's host loop bit for bit; the bisection of `mps block` calls it a dozen times per chromosome and
// the O(n * window) loop is the one host hot spot the reference has there.
__global__ void hanning_smooth_kernel(const float *__restrict__ v, const double *__restrict__ weight, long long n, int window,
                                      double *out)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const long long margin = window / 2;
    double acc = 0.0;
    if (c >= margin && c < n - margin)
    {
        const float *src = v + (c - margin);
        for (int i = 0; i < window; i++) acc += weight[i] * (double)src[i];
    }
    out[c] = acc;
}

int hanning_smooth_impl(cusk_engine *e, const float *v_host, size_t n, const double *weight_host, int window, double *out_host)
{
    if (!e || !v_host || !weight_host || !out_host || n == 0 || window <= 0) return fail(e, CUSK_ERR_ARG, "bad arguments");
    CUSK_HIP(e, hipSetDevice(e->device));
    hipStream_t s = e->stream;
    DevBuf dv, dw, dout;
    CUSK_HIP(e, dv.ensure(sizeof(float) * n));
    CUSK_HIP(e, dw.ensure(sizeof(double) * (size_t)window));
    CUSK_HIP(e, dout.ensure(sizeof(double) * n));
    CUSK_HIP(e, hipMemcpyAsync(dv.p, v_host, sizeof(float) * n, hipMemcpyHostToDevice, s));
    CUSK_HIP(e, hipMemcpyAsync(dw.p, weight_host, sizeof(double) * (size_t)window, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(hanning_smooth_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, dv.as<float>(), dw.as<double>(),
                       (long long)n, window, dout.as<double>());
    CUSK_HIP(e, hipGetLastError());
    CUSK_HIP(e, hipMemcpyAsync(out_host, dout.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    CUSK_HIP(e, hipStreamSynchronize(s));
    dv.release();
    dw.release();
    dout.release();
    return CUSK_OK;
}

}  // namespace cusk

using namespace cusk;

extern "C" int cusk_hanning_smooth(cusk_engine *e, const float *v_host, size_t n, const double *weight_host, int window,
                                   double *out_host)
{
    return hanning_smooth_impl(e, v_host, n, weight_host, window, out_host);
}

extern "C" int cusk_corr_banded(cusk_engine *e, const unsigned char *bed, size_t m, size_t N, size_t width, float *rowsums_host,
                                float *band_host)
{
    return corr_banded_impl(e, bed, m, N, width, rowsums_host, band_host);
}

extern "C" int cusk_corr_build(cusk_engine *e, const unsigned char *bed, const float *phen, size_t m, size_t N, size_t p,
                               const float *mean, const float *std, float *C_dev, float *mxp_host)
{
    return corr_build_impl(e, bed, phen, m, N, p, mean, std, C_dev, mxp_host, nullptr, nullptr);
}

// The build of the next block while the current one is swept: begin enqueues everything on the engine's third stream and
// returns; end waits for it and hands over the marker x trait correlations.  One build in flight per engine.
extern "C" int cusk_corr_build_batch_mxp(cusk_engine *e, const unsigned char *bed_dev, const float *phen_dev, const float *mean_dev,
                                         const float *std_dev, size_t N, size_t p, int nblk, const long long *first_marker,
                                         const int *markers, const int *base, int n, float *C_dev, float *mxp_host)
{
    return corr_build_batch_impl(e, 1, bed_dev, phen_dev, mean_dev, std_dev, N, p, nblk, first_marker, markers, base, nullptr, (size_t)n,
                                 C_dev, mxp_host);
}

extern "C" int cusk_corr_build_batch_mxm(cusk_engine *e, const unsigned char *bed_dev, const float *phen_dev, const float *mean_dev,
                                         const float *std_dev, size_t N, size_t p, int nblk, const long long *first_marker,
                                         const int *markers, const int *base, const unsigned char *keep, int n, float *C_dev)
{
    return corr_build_batch_impl(e, 2, bed_dev, phen_dev, mean_dev, std_dev, N, p, nblk, first_marker, markers, base, keep, (size_t)n,
                                 C_dev, nullptr);
}

// Everything in one call, the marker x marker part speculatively for EVERY block: the host's prefilter (cli.cpp:561-576)
// then runs beside the contingency GEMMs instead of between two synchronisations (a block that is skipped after all had
// its matrix built for nothing: rare -- one false positive among its m x p marginal tests is enough to keep it)
extern "C" int cusk_corr_build_batch(cusk_engine *e, const unsigned char *bed_dev, const float *phen_dev, const float *mean_dev,
                                     const float *std_dev, size_t N, size_t p, int nblk, const long long *first_marker,
                                     const int *markers, const int *base, int n, float *C_dev, float *mxp_host)
{
    if (!e || !mxp_host) return e ? fail(e, CUSK_ERR_ARG, "bad arguments") : CUSK_ERR_ARG;
    int rc = corr_build_batch_impl(e, 1, bed_dev, phen_dev, mean_dev, std_dev, N, p, nblk, first_marker, markers, base, nullptr, (size_t)n,
                                   C_dev, mxp_host, true);
    if (rc != CUSK_OK) return rc;
    rc = corr_build_batch_impl(e, 2, bed_dev, phen_dev, mean_dev, std_dev, N, p, nblk, first_marker, markers, base, nullptr, (size_t)n, C_dev,
                               nullptr);
    if (rc != CUSK_OK) return rc;
    if (p == 0) return CUSK_OK;
    CUSK_HIP(e, hipEventSynchronize(e->ev_mxp));
    size_t cnt = 0;
    for (int b = 0; b < nblk; b++) cnt += (size_t)markers[b] * p;
    std::memcpy(mxp_host, e->mxp_pinned, sizeof(float) * cnt);
    return CUSK_OK;
}

extern "C" int cusk_corr_build_begin(cusk_engine *e, const unsigned char *bed_dev, const float *phen_dev, size_t m, size_t N,
                                     size_t p, const float *mean_dev, const float *std_dev, float *C_dev)
{
    if (!e || !C_dev || p == 0) return e ? fail(e, CUSK_ERR_ARG, "bad arguments") : CUSK_ERR_ARG;
    if (e->mxp_pending) return fail(e, CUSK_ERR_ARG, "a correlation build is already in flight");
    if (m * p > e->mxp_pinned_cap)
    {
        if (e->mxp_pinned) (void)hipHostFree(e->mxp_pinned);
        e->mxp_pinned = nullptr;
        e->mxp_pinned_cap = 0;
        CUSK_HIP(e, hipHostMalloc(reinterpret_cast<void **>(&e->mxp_pinned), sizeof(float) * (m * p + m * p / 4 + 64)));
        e->mxp_pinned_cap = m * p + m * p / 4 + 64;
    }
    const int rc = corr_build_impl(e, bed_dev, phen_dev, m, N, p, mean_dev, std_dev, C_dev, e->mxp_pinned, nullptr, nullptr, true);
    if (rc == CUSK_OK) e->mxp_pending = m * p;
    return rc;
}

extern "C" int cusk_corr_build_pending(const cusk_engine *e) { return (e && e->mxp_pending) ? 1 : 0; }

extern "C" int cusk_corr_build_end(cusk_engine *e, float *mxp_host)
{
    if (!e) return CUSK_ERR_ARG;
    if (!e->mxp_pending) return fail(e, CUSK_ERR_ARG, "no correlation build in flight");
    CUSK_HIP(e, hipSetDevice(e->device));
    const size_t count = e->mxp_pending;
    e->mxp_pending = 0;
    CUSK_HIP(e, hipStreamSynchronize(e->stream3));
    if (mxp_host) std::memcpy(mxp_host, e->mxp_pinned, sizeof(float) * count);
    return CUSK_OK;
}

extern "C" void cusk_corr_timing(const cusk_engine *e, float *ms4)
{
    for (int i = 0; i < 4; i++) ms4[i] = e ? e->corr_ms[i] : 0.0f;
}

namespace {
cusk_engine *corr_engine()
{
    cusk_engine *e = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (cusk_engine_create(&e, dev, nullptr) != CUSK_OK)
    {
        std::fprintf(stderr, "libcusk_hip: cannot create engine (no HIP device?)\n");
        std::exit(EXIT_FAILURE);
    }
    return e;
}
}  // namespace

namespace cusk {
// bodies of the reference-named correlation entry points (host buffers in and out, engine per call); exported with C
// linkage by compat_api.hip and with the reference's own C++ linkage by compat_cxx.cpp
void compat_marker_phen_corr_pearson(const unsigned char *marker_vals, const float *phen_vals, const size_t num_markers,
                                     const size_t num_individuals, const size_t num_phen, const float *marker_mean,
                                     const float *marker_std, float *marker_phen_corrs)
{
    cusk_engine *e = corr_engine();
    if (corr_build_impl(e, marker_vals, phen_vals, num_markers, num_individuals, num_phen, marker_mean, marker_std, nullptr,
                        marker_phen_corrs, nullptr, nullptr) != CUSK_OK)
    {
        std::fprintf(stderr, "libcusk_hip: cu_marker_phen_corr_pearson: %s\n", cusk_last_error(e));
        std::exit(EXIT_FAILURE);
    }
    cusk_engine_destroy(e);
}

void compat_corr_pearson_npn(const unsigned char *marker_vals, const float *phen_vals, const size_t num_markers,
                             const size_t num_individuals, const size_t num_phen, const float *marker_mean,
                             const float *marker_std, float *marker_corrs, float *marker_phen_corrs, float *phen_corrs)
{
    cusk_engine *e = corr_engine();
    const size_t n = num_markers + num_phen;
    float *Cd = static_cast<float *>(cusk_dev_alloc(sizeof(float) * n * n));
    if (!Cd || corr_build_impl(e, marker_vals, phen_vals, num_markers, num_individuals, num_phen, marker_mean, marker_std, Cd,
                               marker_phen_corrs, marker_corrs, phen_corrs) != CUSK_OK)
    {
        std::fprintf(stderr, "libcusk_hip: cu_corr_pearson_npn: %s\n", cusk_last_error(e));
        std::exit(EXIT_FAILURE);
    }
    cusk_dev_free(Cd);
    cusk_engine_destroy(e);
}
}  // namespace cusk
