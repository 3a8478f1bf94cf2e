// ci_exact.h -- device arithmetic of one conditional-independence test in the
// REFERENCE's fp32 operation order (no FMA contraction: this header must be
// compiled with -ffp-contract=off).  It is the "exact path" of the engine:
// every decision whose fast evaluation lands inside the guard band, and every
// value that is reported (pMax), goes through these functions, so that
// adjacency and separation sets agree bit for bit with a CPU evaluation of
// the same formulas.
//
// Formulas follow /root/reference/cusk/src/cuPC-S.cu:
//   Fisher z ............ :465 (level 0), :565-566 (level 1), :698-699/:852-853 (level >= 2)
//   pseudo-inverse l=2,3  :3084-3461, :6434-6451 (Courrieu full-rank Cholesky)
//   pseudo-inverse l>=4 . :3063-3082, :3463-3724 (svdcmp, no singular-value cutoff)
//   Schur complement .... :685-696, :829-850
// Float `log` is the correctly rounded one, (float)log((double)x).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

namespace cusk {

__device__ __forceinline__ float logf_cr(float x) { return (float)log((double)x); }

// |0.5 * log(|(1+r)/(1-r)|)|
__device__ __forceinline__ float fisher_z_ratio(float r)
{
    float q = (1.0f + r) / (1.0f - r);
    float lg = logf_cr(fabsf(q));
    return fabsf(0.5f * lg);
}

// |0.5 * (log|1+r| - log|1-r|)|
__device__ __forceinline__ float fisher_z_diff(float r)
{
    float d = logf_cr(fabsf(1.0f + r)) - logf_cr(fabsf(1.0f - r));
    return fabsf(0.5f * d);
}

// Decide Z(r) < th without evaluating the double-precision log unless the
// cheap estimate is within the guard band of the threshold.  The estimate
// uses the hardware log2 (absolute error of a few 1e-7 on the quantities that
// occur here); the band is 2e-4 relative + 2e-6 absolute, far above that, so
// the fast verdicts can never disagree with the exact comparison.  NaN falls
// through to the exact path, where every comparison with NaN is false.
template <bool DIFF_FORM>
__device__ __forceinline__ bool z_below(float r, float th, float *z_out = nullptr)
{
    float q = (1.0f + r) / (1.0f - r);
    float zf = 0.5f * fabsf(__logf(fabsf(q)));
    float band = th * 2e-4f + 2e-6f;
    if (z_out == nullptr)
    {
        if (zf < th - band) return true;
        if (zf > th + band) return false;
    }
    float z = DIFF_FORM ? fisher_z_diff(r) : fisher_z_ratio(r);
    if (z_out) *z_out = z;
    return z < th;
}

__device__ __forceinline__ float sgn_of(float a, float b) { return (b >= 0.0f) ? fabsf(a) : -fabsf(a); }

// sqrt(a^2+b^2) with a double radicand, cuPC-S.cu:3063-3082
__device__ __forceinline__ float pythag(float a, float b)
{
    float at = fabsf(a), bt = fabsf(b), ct;
    if (at > bt)
    {
        ct = bt / at;
        return (float)((double)at * sqrt(1.0 + (double)(ct * ct)));
    }
    else if (bt > 0.0f)
    {
        ct = at / bt;
        return (float)((double)bt * sqrt(1.0 + (double)(ct * ct)));
    }
    return 0.0f;
}

__device__ __forceinline__ void inverse3(const float (&A)[3][3], float (&B)[3][3])
{
    float det = A[0][0] * (A[2][2] * A[1][1]) - A[0][0] * (A[2][1] * A[1][2]) -
                A[1][0] * (A[2][2] * A[0][1]) + A[1][0] * (A[2][1] * A[0][2]) +
                A[2][0] * (A[1][2] * A[0][1]) - A[2][0] * (A[1][1] * A[0][2]);
    float tmp = (float)(1.0 / (double)det);
    B[0][0] = tmp * (A[1][1] * A[2][2] - A[1][2] * A[2][1]);
    B[0][1] = tmp * (A[0][2] * A[2][1] - A[0][1] * A[2][2]);
    B[0][2] = tmp * (A[0][1] * A[1][2] - A[0][2] * A[1][1]);
    B[1][0] = tmp * (A[1][2] * A[2][0] - A[1][0] * A[2][2]);
    B[1][1] = tmp * (A[0][0] * A[2][2] - A[0][2] * A[2][0]);
    B[1][2] = tmp * (A[0][2] * A[1][0] - A[0][0] * A[1][2]);
    B[2][0] = tmp * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    B[2][1] = tmp * (A[0][1] * A[2][0] - A[0][0] * A[2][1]);
    B[2][2] = tmp * (A[0][0] * A[1][1] - A[0][1] * A[1][0]);
}

// Courrieu pseudo-inverse for SZ = 2 or 3.  M2 and Inv are SZ x SZ row-major.
template <int SZ>
__device__ __forceinline__ void pinv_courrieu(const float *M2, float *Inv)
{
    float A[3][3], M[3][3], Gm[3][3], L[3][3], nL[3][3], t0[3][3], t1[3][3], t2[3][3], t3[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            A[i][j] = M[i][j] = Gm[i][j] = L[i][j] = nL[i][j] = t0[i][j] = t1[i][j] = t2[i][j] = t3[i][j] = 0.0f;
#pragma unroll
    for (int i = 0; i < SZ; i++)
#pragma unroll
        for (int j = 0; j < SZ; j++)
        {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < SZ; k++) acc += M2[i * SZ + k] * M2[k * SZ + j];
            A[i][j] = acc;
        }
    float tol = 999.99f;
#pragma unroll
    for (int i = 0; i < SZ; i++)
        if (tol > A[i][i] && A[i][i] > 0) tol = A[i][i];
    tol = (float)((double)tol * (1e-20));

    int r = 0;
#pragma unroll
    for (int k = 0; k < SZ; k++)
    {
        if (r == 0)
        {
#pragma unroll
            for (int i = 0; i < SZ; i++)
                if (i >= k) L[i][0] = A[i][k];
        }
        else
        {
#pragma unroll
            for (int i = 0; i < SZ; i++)
                if (i >= k)
                {
#pragma unroll
                    for (int l = 0; l < SZ; l++)
                        if (l < r) t0[i][k] += L[i][l] * L[k][l];
                }
#pragma unroll
            for (int i = 0; i < SZ; i++)
                if (i >= k)
                {
#pragma unroll
                    for (int c = 0; c < SZ; c++)
                        if (c == r) L[i][c] = A[i][k] - t0[i][k];
                }
        }
        float piv = 0.0f;
#pragma unroll
        for (int c = 0; c < SZ; c++)
            if (c == r) piv = L[k][c];
        if (piv > tol)
        {
            float sq = sqrtf(piv);
#pragma unroll
            for (int c = 0; c < SZ; c++)
                if (c == r)
                {
                    L[k][c] = sq;
#pragma unroll
                    for (int i = 0; i < SZ; i++)
                        if (i > k) L[i][c] = L[i][c] / sq;
                }
            r++;
        }
    }
#pragma unroll
    for (int i = 0; i < SZ; i++)
#pragma unroll
        for (int j = 0; j < SZ; j++)
            if (j < r) nL[i][j] = L[i][j];
#pragma unroll
    for (int i = 0; i < SZ; i++)
#pragma unroll
        for (int j = 0; j < SZ; j++)
            if (i < r && j < r)
            {
#pragma unroll
                for (int k = 0; k < SZ; k++) Gm[i][j] += nL[k][i] * nL[k][j];
            }
    if (r == 1)
    {
        M[0][0] = 1 / Gm[0][0];
    }
    else if (r == 2)
    {
        float det = 1 / (Gm[0][0] * Gm[1][1] - Gm[0][1] * Gm[1][0]);
        M[0][0] = det * Gm[1][1];
        M[1][1] = det * Gm[0][0];
        M[0][1] = (-1 * det) * Gm[0][1];
        M[1][0] = (-1 * det) * Gm[1][0];
    }
    else if (SZ == 3)
    {
        inverse3(Gm, M);
    }
#pragma unroll
    for (int i = 0; i < SZ; i++)
#pragma unroll
        for (int j = 0; j < SZ; j++)
            if (j < r)
            {
#pragma unroll
                for (int k = 0; k < SZ; k++)
                    if (k < r) t1[i][j] += nL[i][k] * M[k][j];
            }
#pragma unroll
    for (int i = 0; i < SZ; i++)
        if (i < r)
        {
#pragma unroll
            for (int j = 0; j < SZ; j++)
#pragma unroll
                for (int k = 0; k < SZ; k++) t2[i][j] += nL[k][i] * M2[k * SZ + j];
        }
#pragma unroll
    for (int i = 0; i < SZ; i++)
        if (i < r)
        {
#pragma unroll
            for (int j = 0; j < SZ; j++)
#pragma unroll
                for (int k = 0; k < SZ; k++) t3[i][j] += M[i][k] * t2[k][j];
        }
#pragma unroll
    for (int i = 0; i < SZ; i++)
#pragma unroll
        for (int j = 0; j < SZ; j++)
        {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < SZ; k++) acc += t1[i][k] * t3[k][j];
            Inv[i * SZ + j] = acc;
        }
}

// Work arrays of the SVD kept in LDS, one element per lane at a stride of the workgroup size:
// dynamic indexing then costs an LDS access (~64 cycles) instead of a scratch-memory round trip.
struct StridedArr
{
    float *base;
    int stride;
    __device__ __forceinline__ float &operator[](int i) const { return base[i * stride]; }
};

// svdcmp-based pseudo-inverse for M >= 4.  A (M x M row-major) is destroyed; V (M x M), w and
// rv1 (M each) are work arrays.  Arr is float* (private memory) or StridedArr (LDS).
template <int M, typename Arr>
__device__ __forceinline__ void pinv_svd_core(Arr A, Arr V, Arr w, Arr rv1, float *Inv)
{
    int flag, its, i, j, jj, k, l = 0, nm = 0;
    float c, f, h, s, x, y, z;
    float anorm = 0.0f, g = 0.0f, scale = 0.0f;
#define AA(i_, j_) A[(i_)*M + (j_)]
#define VV(i_, j_) V[(i_)*M + (j_)]
    for (i = 0; i < M; i++)
    {
        l = i + 1;
        rv1[i] = scale * g;
        g = s = scale = 0.0f;
        for (k = i; k < M; k++) scale += fabsf(AA(k, i));
        if (scale != 0.0f)
        {
            for (k = i; k < M; k++)
            {
                AA(k, i) = (AA(k, i) / scale);
                s += (AA(k, i) * AA(k, i));
            }
            f = AA(i, i);
            g = -sgn_of(sqrtf(s), f);
            h = f * g - s;
            AA(i, i) = f - g;
            if (i != M - 1)
            {
                for (j = l; j < M; j++)
                {
                    for (s = 0.0f, k = i; k < M; k++) s += (AA(k, i) * AA(k, j));
                    f = s / h;
                    for (k = i; k < M; k++) AA(k, j) += (f * AA(k, i));
                }
            }
            for (k = i; k < M; k++) AA(k, i) = (AA(k, i) * scale);
        }
        w[i] = scale * g;
        g = s = scale = 0.0f;
        if (i != M - 1)
        {
            for (k = l; k < M; k++) scale += fabsf(AA(i, k));
            if (scale != 0.0f)
            {
                for (k = l; k < M; k++)
                {
                    AA(i, k) = (AA(i, k) / scale);
                    s += (AA(i, k) * AA(i, k));
                }
                f = AA(i, l);
                g = -sgn_of(sqrtf(s), f);
                h = f * g - s;
                AA(i, l) = f - g;
                for (k = l; k < M; k++) rv1[k] = AA(i, k) / h;
                for (j = l; j < M; j++)
                {
                    for (s = 0.0f, k = l; k < M; k++) s += (AA(j, k) * AA(i, k));
                    for (k = l; k < M; k++) AA(j, k) += (s * rv1[k]);
                }
                for (k = l; k < M; k++) AA(i, k) = AA(i, k) * scale;
            }
        }
        float cand = fabsf(w[i]) + fabsf(rv1[i]);
        anorm = (anorm > cand) ? anorm : cand;
    }
    for (i = M - 1; i >= 0; i--)
    {
        if (i < M - 1)
        {
            if (g != 0.0f)
            {
                for (j = l; j < M; j++) VV(j, i) = (AA(i, j) / AA(i, l)) / g;
                for (j = l; j < M; j++)
                {
                    for (s = 0.0f, k = l; k < M; k++) s += (AA(i, k) * VV(k, j));
                    for (k = l; k < M; k++) VV(k, j) += (s * VV(k, i));
                }
            }
            for (j = l; j < M; j++) VV(i, j) = VV(j, i) = 0.0f;
        }
        VV(i, i) = 1.0f;
        g = rv1[i];
        l = i;
    }
    for (i = M - 1; i >= 0; i--)
    {
        l = i + 1;
        g = w[i];
        if (i < M - 1)
            for (j = l; j < M; j++) AA(i, j) = 0.0f;
        if (g != 0.0f)
        {
            g = (float)(1.0 / (double)g);
            if (i != M - 1)
            {
                for (j = l; j < M; j++)
                {
                    for (s = 0.0f, k = l; k < M; k++) s += (AA(k, i) * AA(k, j));
                    f = (s / AA(i, i)) * g;
                    for (k = i; k < M; k++) AA(k, j) += (f * AA(k, i));
                }
            }
            for (j = i; j < M; j++) AA(j, i) = (AA(j, i) * g);
        }
        else
        {
            for (j = i; j < M; j++) AA(j, i) = 0.0f;
        }
        AA(i, i) = AA(i, i) + 1.0f;
    }
    for (k = M - 1; k >= 0; k--)
    {
        for (its = 0; its < 30; its++)
        {
            flag = 1;
            for (l = k; l >= 0; l--)
            {
                nm = l - 1;
                if (fabsf(rv1[l]) + anorm == anorm)
                {
                    flag = 0;
                    break;
                }
                if (fabsf(w[nm]) + anorm == anorm) break;
            }
            if (flag)
            {
                c = 0.0f;
                s = 1.0f;
                for (i = l; i <= k; i++)
                {
                    f = s * rv1[i];
                    if (fabsf(f) + anorm != anorm)
                    {
                        g = w[i];
                        h = pythag(f, g);
                        w[i] = h;
                        h = (float)(1.0 / (double)h);
                        c = g * h;
                        s = (-f * h);
                        for (j = 0; j < M; j++)
                        {
                            y = AA(j, nm);
                            z = AA(j, i);
                            AA(j, nm) = (y * c + z * s);
                            AA(j, i) = (z * c - y * s);
                        }
                    }
                }
            }
            z = w[k];
            if (l == k)
            {
                if (z < 0.0f)
                {
                    w[k] = (-z);
                    for (j = 0; j < M; j++) VV(j, k) = (-VV(j, k));
                }
                break;
            }
            x = w[l];
            nm = k - 1;
            y = w[nm];
            g = rv1[nm];
            h = rv1[k];
            f = (float)((double)((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * (double)h * (double)y));
            g = pythag(f, 1.0f);
            f = ((x - z) * (x + z) + h * ((y / (f + sgn_of(g, f))) - h)) / x;
            c = s = 1.0f;
            for (j = l; j <= nm; j++)
            {
                i = j + 1;
                g = rv1[i];
                y = w[i];
                h = s * g;
                g = c * g;
                z = pythag(f, h);
                rv1[j] = z;
                c = f / z;
                s = h / z;
                f = x * c + g * s;
                g = g * c - x * s;
                h = y * s;
                y = y * c;
                for (jj = 0; jj < M; jj++)
                {
                    x = VV(jj, j);
                    z = VV(jj, i);
                    VV(jj, j) = (x * c + z * s);
                    VV(jj, i) = (z * c - x * s);
                }
                z = pythag(f, h);
                w[j] = z;
                if (z != 0.0f)
                {
                    z = (float)(1.0 / (double)z);
                    c = f * z;
                    s = h * z;
                }
                f = (c * g) + (s * y);
                x = (c * y) - (s * g);
                for (jj = 0; jj < M; jj++)
                {
                    y = AA(jj, j);
                    z = AA(jj, i);
                    AA(jj, j) = (y * c + z * s);
                    AA(jj, i) = (z * c - y * s);
                }
            }
            rv1[l] = 0.0f;
            rv1[k] = f;
            w[k] = x;
        }
    }
    // Inv = (V / w) * U^T ; V(i,k)/w[k] is formed first, as the reference does
#pragma unroll
    for (int ii = 0; ii < M; ii++)
#pragma unroll
        for (int jj2 = 0; jj2 < M; jj2++)
        {
            float acc = 0.0f;
#pragma unroll
            for (int kk = 0; kk < M; kk++) acc = acc + (VV(ii, kk) / w[kk]) * AA(jj2, kk);
            Inv[ii * M + jj2] = acc;
        }
#undef AA
#undef VV
}

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N)
    {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// The same svdcmp for small M (4, 5: the levels of the headline workload) with every array index known at compile
// time, so that the matrices live in registers: the LDS form above pays ~64 cycles of LDS latency for almost every
// one of its few thousand dependent operations, which made a single pseudo-inverse ~40 us -- and a level's recheck
// pass or its winners' z is exactly one such chain long.  Loops whose bounds depend on data (the search for l, the
// rotations from l to k) run over their full static range under a predicate; the operations that execute, and
// their order, are those of pinv_svd_core, so the result is bit-identical.
template <int M>
__device__ __forceinline__ void pinv_svd_static(const float *m2, float *Inv)
{
    float a[M][M], v[M][M], w[M], rv1[M];
#pragma unroll
    for (int r = 0; r < M; r++)
#pragma unroll
        for (int q = 0; q < M; q++)
        {
            a[r][q] = m2[r * M + q];
            v[r][q] = 0.0f;
        }
    float anorm = 0.0f, g = 0.0f, scale = 0.0f, s = 0.0f, f, h;
    // Householder reduction to bidiagonal form
    static_for<0, M>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int l = i + 1;
        rv1[i] = scale * g;
        g = s = scale = 0.0f;
#pragma unroll
        for (int k = i; k < M; k++) scale += fabsf(a[k][i]);
        if (scale != 0.0f)
        {
#pragma unroll
            for (int k = i; k < M; k++)
            {
                a[k][i] = (a[k][i] / scale);
                s += (a[k][i] * a[k][i]);
            }
            f = a[i][i];
            g = -sgn_of(sqrtf(s), f);
            h = f * g - s;
            a[i][i] = f - g;
            if constexpr (i != M - 1)
            {
#pragma unroll
                for (int j = l; j < M; j++)
                {
                    s = 0.0f;
#pragma unroll
                    for (int k = i; k < M; k++) s += (a[k][i] * a[k][j]);
                    f = s / h;
#pragma unroll
                    for (int k = i; k < M; k++) a[k][j] += (f * a[k][i]);
                }
            }
#pragma unroll
            for (int k = i; k < M; k++) a[k][i] = (a[k][i] * scale);
        }
        w[i] = scale * g;
        g = s = scale = 0.0f;
        if constexpr (i != M - 1)
        {
#pragma unroll
            for (int k = l; k < M; k++) scale += fabsf(a[i][k]);
            if (scale != 0.0f)
            {
#pragma unroll
                for (int k = l; k < M; k++)
                {
                    a[i][k] = (a[i][k] / scale);
                    s += (a[i][k] * a[i][k]);
                }
                f = a[i][l];
                g = -sgn_of(sqrtf(s), f);
                h = f * g - s;
                a[i][l] = f - g;
#pragma unroll
                for (int k = l; k < M; k++) rv1[k] = a[i][k] / h;
#pragma unroll
                for (int j = l; j < M; j++)
                {
                    s = 0.0f;
#pragma unroll
                    for (int k = l; k < M; k++) s += (a[j][k] * a[i][k]);
#pragma unroll
                    for (int k = l; k < M; k++) a[j][k] += (s * rv1[k]);
                }
#pragma unroll
                for (int k = l; k < M; k++) a[i][k] = a[i][k] * scale;
            }
        }
        const float cand = fabsf(w[i]) + fabsf(rv1[i]);
        anorm = (anorm > cand) ? anorm : cand;
    });
    // accumulation of the right-hand transformations (i = M-1 .. 0; l = i + 1 whenever it is used)
    static_for<0, M>([&](auto rc) {
        constexpr int i = M - 1 - decltype(rc)::value;
        constexpr int l = i + 1;
        if constexpr (i < M - 1)
        {
            if (g != 0.0f)
            {
#pragma unroll
                for (int j = l; j < M; j++) v[j][i] = (a[i][j] / a[i][l]) / g;
#pragma unroll
                for (int j = l; j < M; j++)
                {
                    s = 0.0f;
#pragma unroll
                    for (int k = l; k < M; k++) s += (a[i][k] * v[k][j]);
#pragma unroll
                    for (int k = l; k < M; k++) v[k][j] += (s * v[k][i]);
                }
            }
#pragma unroll
            for (int j = l; j < M; j++) v[i][j] = v[j][i] = 0.0f;
        }
        v[i][i] = 1.0f;
        g = rv1[i];
    });
    // accumulation of the left-hand transformations
    static_for<0, M>([&](auto rc) {
        constexpr int i = M - 1 - decltype(rc)::value;
        constexpr int l = i + 1;
        g = w[i];
#pragma unroll
        for (int j = l; j < M; j++) a[i][j] = 0.0f;
        if (g != 0.0f)
        {
            g = (float)(1.0 / (double)g);
            if constexpr (i != M - 1)
            {
#pragma unroll
                for (int j = l; j < M; j++)
                {
                    s = 0.0f;
#pragma unroll
                    for (int k = l; k < M; k++) s += (a[k][i] * a[k][j]);
                    f = (s / a[i][i]) * g;
#pragma unroll
                    for (int k = i; k < M; k++) a[k][j] += (f * a[k][i]);
                }
            }
#pragma unroll
            for (int j = i; j < M; j++) a[j][i] = (a[j][i] * g);
        }
        else
        {
#pragma unroll
            for (int j = i; j < M; j++) a[j][i] = 0.0f;
        }
        a[i][i] = a[i][i] + 1.0f;
    });
    // diagonalisation of the bidiagonal form, k = M-1 .. 0
    static_for<0, M>([&](auto rc) {
        constexpr int k = M - 1 - decltype(rc)::value;
        for (int its = 0; its < 30; its++)
        {
            int flag = 1, l = 0;
            {  // for (l = k; l >= 0; l--) with its two exits; rv1[0] is always 0, so the search ends at l = 0 at the latest
                bool found = false;
#pragma unroll
                for (int ll = k; ll >= 0; ll--)
                {
                    if (!found)
                    {
                        l = ll;
                        if (fabsf(rv1[ll]) + anorm == anorm)
                        {
                            flag = 0;
                            found = true;
                        }
                        else if (ll > 0 && fabsf(w[ll > 0 ? ll - 1 : 0]) + anorm == anorm)
                            found = true;
                    }
                }
            }
            float c, x, y, z;
            if (flag)
            {  // cancellation of rv1[l]; nm = l - 1 >= 0 here
                const int nm = l - 1;
                c = 0.0f;
                s = 1.0f;
#pragma unroll
                for (int i = 0; i <= k; i++)
                {
                    if (i >= l)
                    {
                        f = s * rv1[i];
                        if (fabsf(f) + anorm != anorm)
                        {
                            g = w[i];
                            h = pythag(f, g);
                            w[i] = h;
                            h = (float)(1.0 / (double)h);
                            c = g * h;
                            s = (-f * h);
#pragma unroll
                            for (int j = 0; j < M; j++)
                            {
                                y = 0.0f;
#pragma unroll
                                for (int q = 0; q < M; q++)
                                    if (q == nm) y = a[j][q];
                                z = a[j][i];
                                const float ynew = (y * c + z * s);
#pragma unroll
                                for (int q = 0; q < M; q++)
                                    if (q == nm) a[j][q] = ynew;
                                a[j][i] = (z * c - y * s);
                            }
                        }
                    }
                }
            }
            z = w[k];
            if (l == k)
            {
                if (z < 0.0f)
                {
                    w[k] = (-z);
#pragma unroll
                    for (int j = 0; j < M; j++) v[j][k] = (-v[j][k]);
                }
                break;
            }
            if constexpr (k > 0)
            {
                constexpr int nm = k - 1;
                x = 0.0f;
#pragma unroll
                for (int q = 0; q <= k; q++)
                    if (q == l) x = w[q];
                y = w[nm];
                g = rv1[nm];
                h = rv1[k];
                f = (float)((double)((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * (double)h * (double)y));
                g = pythag(f, 1.0f);
                f = ((x - z) * (x + z) + h * ((y / (f + sgn_of(g, f))) - h)) / x;
                c = s = 1.0f;
#pragma unroll
                for (int j = 0; j <= nm; j++)
                {
                    if (j >= l)
                    {
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int i = j + 1;
                        g = rv1[i];
                        y = w[i];
                        h = s * g;
                        g = c * g;
                        z = pythag(f, h);
                        rv1[j] = z;
                        c = f / z;
                        s = h / z;
                        f = x * c + g * s;
                        g = g * c - x * s;
                        h = y * s;
                        y = y * c;
#pragma unroll
                        for (int jj = 0; jj < M; jj++)
                        {
                            x = v[jj][j];
                            z = v[jj][i];
                            v[jj][j] = (x * c + z * s);
                            v[jj][i] = (z * c - x * s);
                        }
                        z = pythag(f, h);
                        w[j] = z;
                        if (z != 0.0f)
                        {
                            z = (float)(1.0 / (double)z);
                            c = f * z;
                            s = h * z;
                        }
                        f = (c * g) + (s * y);
                        x = (c * y) - (s * g);
#pragma unroll
                        for (int jj = 0; jj < M; jj++)
                        {
                            y = a[jj][j];
                            z = a[jj][i];
                            a[jj][j] = (y * c + z * s);
                            a[jj][i] = (z * c - y * s);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q <= k; q++)
                    if (q == l) rv1[q] = 0.0f;
                rv1[k] = f;
                w[k] = x;
            }
        }
    });
    // Inv = (V / w) * U^T ; V(i,k)/w[k] is formed first, as the reference does
#pragma unroll
    for (int ii = 0; ii < M; ii++)
#pragma unroll
        for (int jj2 = 0; jj2 < M; jj2++)
        {
            float acc = 0.0f;
#pragma unroll
            for (int kk = 0; kk < M; kk++) acc = acc + (v[ii][kk] / w[kk]) * a[jj2][kk];
            Inv[ii * M + jj2] = acc;
        }
}

template <int M>
__device__ __noinline__ void pinv_svd(float *A, float *Inv)
{
    float V[M * M], w[M], rv1[M];
    pinv_svd_core<M, float *>(A, V, w, rv1, Inv);
}

// floats of LDS work space per lane for the exact path at conditioning-set size L
__host__ __device__ constexpr int exact_ws_floats(int L) { return (L >= 4) ? (2 * L * L + 2 * L) : 0; }

// M2 is L x L row-major and may be destroyed.
template <int L>
__device__ __forceinline__ void pinv_ref_order(float *M2, float *Inv)
{
    if constexpr (L == 2 || L == 3)
        pinv_courrieu<L>(M2, Inv);
    else
        pinv_svd<L>(M2, Inv);
}

// Per-subset state: inverse of C[S,S] and the X-side products.
constexpr int kSvdStaticMax = 5;  // conditioning-set sizes whose pseudo-inverse runs from registers

template <int L>
struct SubsetExact
{
    float m1x[L];
    float inv[(L > 1) ? L * L : 1];
    float mmx[L];
    float h00;

    // m2: L x L row-major with unit diagonal (destroyed); m1x_in: C[X,S]
    __device__ __forceinline__ void prepare(float *m2, const float *m1x_in)
    {
#pragma unroll
        for (int a = 0; a < L; a++) m1x[a] = m1x_in[a];
        if constexpr (L >= 2)
        {
            pinv_ref_order<L>(m2, inv);
#pragma unroll
            for (int c2 = 0; c2 < L; c2++)
            {
                float acc = 0.0f;
#pragma unroll
                for (int c3 = 0; c3 < L; c3++) acc += m1x[c3] * inv[c3 * L + c2];
                mmx[c2] = acc;
            }
            float h = 0.0f;
#pragma unroll
            for (int c3 = 0; c3 < L; c3++) h += mmx[c3] * m1x[c3];
            h00 = h;
        }
    }

    // same as prepare(), with the SVD work arrays in LDS: ws points at this lane's first float,
    // consecutive floats of one lane are `stride` apart (exact_ws_floats(L) of them)
    __device__ __forceinline__ void prepare_ws(const float *m2, const float *m1x_in, float *ws, int stride)
    {
        if constexpr (L < 4)
        {
            float tmp[(L > 1) ? L * L : 1];
#pragma unroll
            for (int a = 0; a < ((L > 1) ? L * L : 1); a++) tmp[a] = m2[a];
            prepare(tmp, m1x_in);
        }
        else if constexpr (L <= kSvdStaticMax)
        {  // registers instead of the LDS work space (which stays allocated but unused)
#pragma unroll
            for (int a = 0; a < L; a++) m1x[a] = m1x_in[a];
            pinv_svd_static<L>(m2, inv);
#pragma unroll
            for (int c2 = 0; c2 < L; c2++)
            {
                float acc = 0.0f;
#pragma unroll
                for (int c3 = 0; c3 < L; c3++) acc += m1x[c3] * inv[c3 * L + c2];
                mmx[c2] = acc;
            }
            float h = 0.0f;
#pragma unroll
            for (int c3 = 0; c3 < L; c3++) h += mmx[c3] * m1x[c3];
            h00 = h;
            (void)ws;
            (void)stride;
        }
        else
        {
#pragma unroll
            for (int a = 0; a < L; a++) m1x[a] = m1x_in[a];
            StridedArr A{ws, stride}, V{ws + (size_t)L * L * stride, stride}, w{ws + (size_t)2 * L * L * stride, stride},
                rv1{ws + (size_t)(2 * L * L + L) * stride, stride};
#pragma unroll
            for (int a = 0; a < L * L; a++) A[a] = m2[a];
            pinv_svd_core<L, StridedArr>(A, V, w, rv1, inv);
#pragma unroll
            for (int c2 = 0; c2 < L; c2++)
            {
                float acc = 0.0f;
#pragma unroll
                for (int c3 = 0; c3 < L; c3++) acc += m1x[c3] * inv[c3 * L + c2];
                mmx[c2] = acc;
            }
            float h = 0.0f;
#pragma unroll
            for (int c3 = 0; c3 < L; c3++) h += mmx[c3] * m1x[c3];
            h00 = h;
        }
    }

    // partial correlation of X and Y given S; m0 = C[X,Y], m1y = C[Y,S]
    __device__ __forceinline__ float rho(float m0, const float *m1y) const
    {
        if constexpr (L == 1)
        {
            float a = m1x[0], b = m1y[0];
            float H00 = 1.0f - (a * a);
            float H01 = m0 - (a * b);
            float H11 = 1.0f - (b * b);
            return H01 / (sqrtf(fabsf(H00)) * sqrtf(fabsf(H11)));
        }
        else
        {
            float mmy[L];
#pragma unroll
            for (int c2 = 0; c2 < L; c2++)
            {
                float acc = 0.0f;
#pragma unroll
                for (int c3 = 0; c3 < L; c3++) acc += m1y[c3] * inv[c3 * L + c2];
                mmy[c2] = acc;
            }
            float h01 = 0.0f, h11 = 0.0f;
#pragma unroll
            for (int c3 = 0; c3 < L; c3++) h01 += mmx[c3] * m1y[c3];
#pragma unroll
            for (int c3 = 0; c3 < L; c3++) h11 += mmy[c3] * m1y[c3];
            float H00 = 1.0f - h00;
            float H01 = m0 - h01;
            float H11 = 1.0f - h11;
            return H01 / (sqrtf(fabsf(H00 * H11)));
        }
    }
};

}  // namespace cusk
