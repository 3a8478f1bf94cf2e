/*
 * cusk_oracle.c -- CPU restatement of the ci-gwas `cusk` hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the shipped
 * library (ci-gwas_amd/csrc) never links, includes or calls anything here.
 *
 * It restates, in plain C with IEEE fp32/fp64 arithmetic and no FMA
 * contraction (build with -ffp-contract=off), the algorithm of the
 * reference's CUDA engines.  Citations are to /root/reference/cusk:
 *
 *   thresholds ............ src/cuPC_call_prep.cpp:7-27
 *   level 0 ............... src/cuPC-S.cu:458-484, src/hetcor-cuPC-S.cu:343-377
 *   neighbour lists ....... src/cuPC-S.cu:6355-6432 (ascending index, frozen per level)
 *   level 1 ............... src/cuPC-S.cu:486-582, src/hetcor-cuPC-S.cu:379-486
 *   level l>=2 ............ src/cuPC-S.cu:584-871 (l=2,3), :873-3020 (l=4..14)
 *   pseudo-inverse l=2,3 .. src/cuPC-S.cu:3084-3461, :6434-6451
 *   pseudo-inverse l>=4 ... src/cuPC-S.cu:3063-3082, :3463-3724 (same text for every l)
 *   combinations .......... src/cuPC-S.cu:6453-6506 (lexicographic, 1-based rank)
 *   level loop ............ src/cuPC-S.cu:99-159, :418-442
 *   hetcor extras ......... src/hetcor-cuPC-S.cu:3055-3088 (time index, mean_ess)
 *   correlation build ..... src/corr_kernels.cu:157-238, :285-343, :478-565
 *
 * Where the reference is racy or undefined the oracle fixes ONE answer that
 * the reference can produce (documented in DESIGN.md, "canonical semantics"):
 *   - every subset of every frozen neighbour list is enumerated in rank order
 *     (the reference's NoEdgeFlag early exit only skips work);
 *   - the separating set stored for the ORDERED pair (X,Y) is the lowest-rank
 *     subset of adj(X)\{Y} that passes the test; (X,Y) and (Y,X) are decided
 *     independently (the reference locks on the ordered pair, cuPC-S.cu:570);
 *   - device pMax starts at 0 (the reference leaves it uninitialised);
 *   - binomials are 64-bit (the reference's int overflows);
 *   - float `log` is the correctly rounded one, (float)log((double)x): the
 *     reference's is CUDA's --use_fast_math __logf, which no CPU reproduces;
 *   - NaN -> int in mean_ess is 0, as on the GPU.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_ML 14

/* ------------------------------------------------------------------ */
/* thresholds: cuPC_call_prep.cpp:7-27                                 */
/* ------------------------------------------------------------------ */

/* Standard normal quantile in double.  The reference calls
 * boost::math::quantile(normal(0,1), p) (boost is not vendored, version
 * unpinned); any double-accurate inverse gives the same float.  Acklam's
 * rational approximation followed by two Halley steps on erfc. */
double orc_qnorm(double p)
{
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02,
                               -2.759285104469687e+02, 1.383577518672690e+02,
                               -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02,
                               -1.556989798598866e+02, 6.680131188771972e+01,
                               -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01,
                               -2.400758277161838e+00, -2.549732539343734e+00,
                               4.374664141464968e+00,  2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01,
                               2.445134137142996e+00, 3.754408661907416e+00};
    if (!(p > 0.0 && p < 1.0)) return p == 0.0 ? -INFINITY : (p == 1.0 ? INFINITY : NAN);
    double x, q, r;
    if (p < 0.02425) {
        q = sqrt(-2.0 * log(p));
        x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
            ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    } else if (p > 1.0 - 0.02425) {
        q = sqrt(-2.0 * log(1.0 - p));
        x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
            ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    } else {
        q = p - 0.5;
        r = q * q;
        x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
            (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
    }
    for (int it = 0; it < 2; it++) {
        double e = 0.5 * erfc(-x / sqrt(2.0)) - p;
        double u = e * sqrt(2.0 * M_PI) * exp(x * x / 2.0);
        x = x - u / (1.0 + x * u / 2.0);
    }
    return x;
}

/* std_normal_qnorm takes and returns float (cuPC_call_prep.cpp:7-11). */
static float qnorm_f(float p) { return (float)orc_qnorm((double)p); }

/* threshold_array: thr[i] = abs(qnorm(0.5f*alpha)) / sqrt(n - i - 3), i = 0..14;
 * the divisor is size_t -> double, the quotient is rounded to float. */
void orc_threshold_array(int n, float alpha, float *thr)
{
    const float half = 0.5f;
    for (size_t i = 0; i < ORC_ML + 1; i++) {
        float q = fabsf(qnorm_f(half * alpha));
        size_t dof = (size_t)n - i - 3;
        thr[i] = (float)((double)q / sqrt((double)dof));
    }
}

/* hetcor_threshold: abs(qnorm(0.5 * alpha)); 0.5*alpha is double, narrowed to
 * the float parameter. */
float orc_hetcor_threshold(float alpha) { return fabsf(qnorm_f((float)(0.5 * (double)alpha))); }

/* ------------------------------------------------------------------ */
/* scalar helpers                                                      */
/* ------------------------------------------------------------------ */

static inline float logf_cr(float x) { return (float)log((double)x); }

/* abs(0.5 * log(abs((1 + r) / (1 - r))))  -- cuPC-S.cu:465, :853 */
static inline float fisher_z_ratio(float r)
{
    float q = (1 + r) / (1 - r);
    float lg = logf_cr(fabsf(q));
    return (float)fabs(0.5 * (double)lg);
}

/* fabs(0.5 * (log(fabs(1 + r)) - log(fabs(1 - r))))  -- cuPC-S.cu:566 */
static inline float fisher_z_diff(float r)
{
    float d = logf_cr(fabsf(1 + r)) - logf_cr(fabsf(1 - r));
    return (float)fabs(0.5 * (double)d);
}

float orc_fisher_z_ratio(float r) { return fisher_z_ratio(r); }
float orc_fisher_z_diff(float r) { return fisher_z_diff(r); }

/* ------------------------------------------------------------------ */
/* combinations: cuPC-S.cu:6453-6506, in 64-bit                        */
/* ------------------------------------------------------------------ */

/* C(n,k), saturating at UINT64_MAX/4. */
uint64_t orc_binom(int n, int k)
{
    if (k < 0 || k > n) return 0;
    if (k > n - k) k = n - k;
    uint64_t r = 1;
    const uint64_t cap = UINT64_MAX / 4;
    for (int i = 1; i <= k; i++) {
        uint64_t f = (uint64_t)(n - k + i);
        if (r > cap / f) return cap;
        r = r * f / (uint64_t)i;
    }
    return r;
}

/* rank is 1-based, out[] holds 1-based ascending positions (IthCombination). */
void orc_ith_combination(int *out, int N, int P, uint64_t L)
{
    uint64_t k = 0, R = 0;
    for (int i = 0; i < P - 1; i++) {
        out[i] = (i > 0) ? out[i - 1] : 0;
        while (k < L) {
            out[i] = out[i] + 1;
            R = orc_binom(N - out[i], P - (i + 1));
            k = k + R;
        }
        k = k - R;
    }
    out[P - 1] = (P > 1 ? out[P - 2] : 0) + (int)(L - k);
}

/* next combination of 0-based positions idx[0..l-1] out of d; returns 0 at end */
static int next_comb(int *idx, int l, int d)
{
    int i = l - 1;
    while (i >= 0 && idx[i] == d - l + i) i--;
    if (i < 0) return 0;
    idx[i]++;
    for (int j = i + 1; j < l; j++) idx[j] = idx[j - 1] + 1;
    return 1;
}

/* ------------------------------------------------------------------ */
/* pseudo-inverses                                                     */
/* ------------------------------------------------------------------ */

#define SGN(a, b) ((b) >= 0.0 ? fabsf(a) : -fabsf(a))

/* PYTHAG, cuPC-S.cu:3063-3082: the 1.0 literal makes the radicand double. */
static float pythag(float a, float b)
{
    float at = fabsf(a), bt = fabsf(b), ct, result;
    if (at > bt) {
        ct = bt / at;
        result = (float)((double)at * sqrt(1.0 + (double)(ct * ct)));
    } else if (bt > 0.0) {
        ct = at / bt;
        result = (float)((double)bt * sqrt(1.0 + (double)(ct * ct)));
    } else {
        result = 0.0f;
    }
    return result;
}

/* 3x3 adjugate inverse, cuPC-S.cu:6434-6451 */
static void inverse3(float A[3][3], float B[3][3])
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

/* Courrieu full-rank-Cholesky pseudo-inverse for size 2 and 3
 * (cuPC-S.cu:3084-3270 and :3272-3461; the two differ only in how the r x r
 * Gram matrix is inverted).  m2 and inv are size x size, row-major, stride 3. */
static void pinv_courrieu(int size, float M2[3][3], float M2Inv[3][3])
{
    float A[3][3] = {{0}}, M[3][3] = {{0}}, G[3][3] = {{0}}, L[3][3] = {{0}}, nL[3][3] = {{0}};
    float t0[3][3] = {{0}}, t1[3][3] = {{0}}, t2[3][3] = {{0}}, t3[3][3] = {{0}};
    float tol = 999.99f;
    int r = 0;

    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++) M2Inv[i][j] = 0.0f;

    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++)
            for (int k = 0; k < size; k++) A[i][j] += M2[i][k] * M2[k][j];

    for (int i = 0; i < size; i++)
        if (tol > A[i][i] && A[i][i] > 0) tol = A[i][i];
    tol = (float)((double)tol * (1e-20));

    for (int k = 0; k < size; k++) {
        if (r == 0) {
            for (int i = k; i < size; i++) L[i][r] = A[i][k];
        } else {
            for (int i = k; i < size; ++i)
                for (int l = 0; l < r; l++) t0[i][k] += L[i][l] * L[k][l];
            for (int i = k; i < size; i++) L[i][r] = A[i][k] - t0[i][k];
        }
        if (L[k][r] > tol) {
            L[k][r] = sqrtf(L[k][r]);
            for (int i = k + 1; i < size; i++) L[i][r] = L[i][r] / L[k][r];
        } else {
            r--;
        }
        r++;
    }

    for (int i = 0; i < size; i++)
        for (int j = 0; j < r; j++) nL[i][j] = L[i][j];

    /* Gram matrix of the retained columns */
    for (int i = 0; i < r; i++)
        for (int j = 0; j < r; j++)
            for (int k = 0; k < size; k++) G[i][j] += nL[k][i] * nL[k][j];

    if (r == 1) {
        M[0][0] = 1 / G[0][0];
    } else if (r == 2) {
        float det = 1 / (G[0][0] * G[1][1] - G[0][1] * G[1][0]);
        M[0][0] = det * G[1][1];
        M[1][1] = det * G[0][0];
        M[0][1] = (-1 * det) * G[0][1];
        M[1][0] = (-1 * det) * G[1][0];
    } else if (size == 3) {
        inverse3(G, M); /* r == 3, and also r == 0 (yields NaN), as in the reference */
    }

    for (int i = 0; i < size; i++)
        for (int j = 0; j < r; j++)
            for (int k = 0; k < r; k++) t1[i][j] += nL[i][k] * M[k][j];
    for (int i = 0; i < r; i++)
        for (int j = 0; j < size; j++)
            for (int k = 0; k < size; k++) t2[i][j] += nL[k][i] * M2[k][j];
    for (int i = 0; i < r; i++)
        for (int j = 0; j < size; j++)
            for (int k = 0; k < size; k++) t3[i][j] += M[i][k] * t2[k][j];
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++)
            for (int k = 0; k < size; k++) M2Inv[i][j] += t1[i][k] * t3[k][j];
}

/* Numerical-Recipes svdcmp pseudo-inverse used for l >= 4
 * (cuPC-S.cu:3463-3724; identical text for l = 5..14).  A is m x m row-major
 * (destroyed: becomes U), Inv receives V diag(1/w) U^T with NO cutoff. */
#define AA(i, j) A[(i) * m + (j)]
#define VV(i, j) V[(i) * m + (j)]
static void pinv_svd(int m, float *A, float *Inv)
{
    float V[ORC_ML * ORC_ML], R[ORC_ML * ORC_ML], w[ORC_ML], rv1[ORC_ML];
    int flag, its, i, j, jj, k, l = 0, nm = 0;
    float c, f, h, s, x, y, z;
    float anorm = 0.0f, g = 0.0f, scale = 0.0f;

    for (i = 0; i < m; i++) {
        l = i + 1;
        rv1[i] = scale * g;
        g = s = scale = 0.0f;
        for (k = i; k < m; k++) scale += fabsf(AA(k, i));
        if (scale) {
            for (k = i; k < m; k++) {
                AA(k, i) = (AA(k, i) / scale);
                s += (AA(k, i) * AA(k, i));
            }
            f = AA(i, i);
            g = -SGN(sqrtf(s), f);
            h = f * g - s;
            AA(i, i) = f - g;
            if (i != m - 1) {
                for (j = l; j < m; j++) {
                    for (s = 0.0f, k = i; k < m; k++) s += (AA(k, i) * AA(k, j));
                    f = s / h;
                    for (k = i; k < m; k++) AA(k, j) += (f * AA(k, i));
                }
            }
            for (k = i; k < m; k++) AA(k, i) = (AA(k, i) * scale);
        }
        w[i] = scale * g;

        g = s = scale = 0.0f;
        if (i != m - 1) {
            for (k = l; k < m; k++) scale += fabsf(AA(i, k));
            if (scale) {
                for (k = l; k < m; k++) {
                    AA(i, k) = (AA(i, k) / scale);
                    s += (AA(i, k) * AA(i, k));
                }
                f = AA(i, l);
                g = -SGN(sqrtf(s), f);
                h = f * g - s;
                AA(i, l) = f - g;
                for (k = l; k < m; k++) rv1[k] = AA(i, k) / h;
                for (j = l; j < m; j++) {
                    for (s = 0.0f, k = l; k < m; k++) s += (AA(j, k) * AA(i, k));
                    for (k = l; k < m; k++) AA(j, k) += (s * rv1[k]);
                }
                for (k = l; k < m; k++) AA(i, k) = AA(i, k) * scale;
            }
        }
        {
            float cand = fabsf(w[i]) + fabsf(rv1[i]);
            anorm = (anorm > cand) ? anorm : cand;
        }
    }

    /* right-hand transformations */
    for (i = m - 1; i >= 0; i--) {
        if (i < m - 1) {
            if (g) {
                for (j = l; j < m; j++) VV(j, i) = (AA(i, j) / AA(i, l)) / g;
                for (j = l; j < m; j++) {
                    for (s = 0.0f, k = l; k < m; k++) s += (AA(i, k) * VV(k, j));
                    for (k = l; k < m; k++) VV(k, j) += (s * VV(k, i));
                }
            }
            for (j = l; j < m; j++) VV(i, j) = VV(j, i) = 0.0f;
        }
        VV(i, i) = 1.0f;
        g = rv1[i];
        l = i;
    }

    /* left-hand transformations */
    for (i = m - 1; i >= 0; i--) {
        l = i + 1;
        g = w[i];
        if (i < m - 1)
            for (j = l; j < m; j++) AA(i, j) = 0.0f;
        if (g) {
            g = (float)(1.0 / (double)g);
            if (i != m - 1) {
                for (j = l; j < m; j++) {
                    for (s = 0.0f, k = l; k < m; k++) s += (AA(k, i) * AA(k, j));
                    f = (s / AA(i, i)) * g;
                    for (k = i; k < m; k++) AA(k, j) += (f * AA(k, i));
                }
            }
            for (j = i; j < m; j++) AA(j, i) = (AA(j, i) * g);
        } else {
            for (j = i; j < m; j++) AA(j, i) = 0.0f;
        }
        AA(i, i) = AA(i, i) + 1.0f;
    }

    /* diagonalisation of the bidiagonal form */
    for (k = m - 1; k >= 0; k--) {
        for (its = 0; its < 30; its++) {
            flag = 1;
            for (l = k; l >= 0; l--) {
                nm = l - 1;
                if (fabsf(rv1[l]) + anorm == anorm) {
                    flag = 0;
                    break;
                }
                if (fabsf(w[nm]) + anorm == anorm) break;
            }
            if (flag) {
                c = 0.0f;
                s = 1.0f;
                for (i = l; i <= k; i++) {
                    f = s * rv1[i];
                    if (fabsf(f) + anorm != anorm) {
                        g = w[i];
                        h = pythag(f, g);
                        w[i] = h;
                        h = (float)(1.0 / (double)h);
                        c = g * h;
                        s = (-f * h);
                        for (j = 0; j < m; j++) {
                            y = AA(j, nm);
                            z = AA(j, i);
                            AA(j, nm) = (y * c + z * s);
                            AA(j, i) = (z * c - y * s);
                        }
                    }
                }
            }
            z = w[k];
            if (l == k) {
                if (z < 0.0) {
                    w[k] = (-z);
                    for (j = 0; j < m; j++) VV(j, k) = (-VV(j, k));
                }
                break;
            }
            x = w[l];
            nm = k - 1;
            y = w[nm];
            g = rv1[nm];
            h = rv1[k];
            f = (float)((double)((y - z) * (y + z) + (g - h) * (g + h)) /
                        (2.0 * (double)h * (double)y));
            g = pythag(f, 1.0f);
            f = ((x - z) * (x + z) + h * ((y / (f + SGN(g, f))) - h)) / x;

            c = s = 1.0f;
            for (j = l; j <= nm; j++) {
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
                for (jj = 0; jj < m; jj++) {
                    x = VV(jj, j);
                    z = VV(jj, i);
                    VV(jj, j) = (x * c + z * s);
                    VV(jj, i) = (z * c - x * s);
                }
                z = pythag(f, h);
                w[j] = z;
                if (z) {
                    z = (float)(1.0 / (double)z);
                    c = f * z;
                    s = h * z;
                }
                f = (c * g) + (s * y);
                x = (c * y) - (s * g);
                for (jj = 0; jj < m; jj++) {
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

    for (i = 0; i < m; i++)
        for (j = 0; j < m; j++) R[i * m + j] = VV(i, j) / w[j];
    for (i = 0; i < m; i++)
        for (j = 0; j < m; j++) {
            float acc = 0;
            for (k = 0; k < m; k++) acc = acc + R[i * m + k] * AA(j, k);
            Inv[i * m + j] = acc;
        }
}
#undef AA
#undef VV

/* dispatcher: M2 (l x l, row-major, destroyed) -> Inv (l x l, row-major) */
static void pinv_ref_order(int l, float *M2, float *Inv)
{
    if (l == 2 || l == 3) {
        float a[3][3] = {{0}}, b[3][3];
        for (int i = 0; i < l; i++)
            for (int j = 0; j < l; j++) a[i][j] = M2[i * l + j];
        pinv_courrieu(l, a, b);
        for (int i = 0; i < l; i++)
            for (int j = 0; j < l; j++) Inv[i * l + j] = b[i][j];
    } else {
        pinv_svd(l, M2, Inv);
    }
}

/* exported for unit tests */
void orc_pinv(int l, const float *M2, float *Inv)
{
    float tmp[ORC_ML * ORC_ML];
    memcpy(tmp, M2, sizeof(float) * (size_t)l * (size_t)l);
    pinv_ref_order(l, tmp, Inv);
}

/* ------------------------------------------------------------------ */
/* one CI test, reference operation order                              */
/* ------------------------------------------------------------------ */

typedef struct {
    int l;
    float m1x[ORC_ML];          /* C[X, S] */
    float inv[ORC_ML * ORC_ML]; /* pinv(C[S,S]) */
    float mmx[ORC_ML];          /* m1x * inv (row 0 of M1MulM2Inv) */
    float h00;                  /* sum_c mmx[c]*m1x[c] (before "1 -") */
} subset_ctx;

static void subset_prepare(subset_ctx *cx, const float *C, size_t n, int X, const int *S, int l)
{
    cx->l = l;
    if (l == 1) {
        cx->m1x[0] = C[(size_t)X * n + S[0]];
        return;
    }
    float m2[ORC_ML * ORC_ML];
    for (int a = 0; a < l; a++)
        for (int b = 0; b < l; b++) {
            if (a == b)
                m2[a * l + b] = 1;
            else if (a < b)
                m2[a * l + b] = C[(size_t)S[a] * n + S[b]];
            else
                m2[a * l + b] = C[(size_t)S[b] * n + S[a]];
        }
    for (int a = 0; a < l; a++) cx->m1x[a] = C[(size_t)X * n + S[a]];
    pinv_ref_order(l, m2, cx->inv);
    for (int c2 = 0; c2 < l; c2++) {
        float acc = 0;
        for (int c3 = 0; c3 < l; c3++) acc += cx->m1x[c3] * cx->inv[c3 * l + c2];
        cx->mmx[c2] = acc;
    }
    float h = 0;
    for (int c3 = 0; c3 < l; c3++) h += cx->mmx[c3] * cx->m1x[c3];
    cx->h00 = h;
}

/* returns Z; *rho_out gets the partial correlation */
static float subset_test(const subset_ctx *cx, const float *C, size_t n, int X, int Y,
                         const int *S, float *rho_out)
{
    int l = cx->l;
    float M0 = C[(size_t)X * n + Y];
    float rho, Z;
    if (l == 1) {
        float a = cx->m1x[0], b = C[(size_t)Y * n + S[0]];
        float H00 = 1 - (a * a);
        float H01 = M0 - (a * b);
        float H11 = 1 - (b * b);
        rho = H01 / (sqrtf(fabsf(H00)) * sqrtf(fabsf(H11)));
        Z = fisher_z_diff(rho);
    } else {
        float m1y[ORC_ML], mmy[ORC_ML];
        for (int a = 0; a < l; a++) m1y[a] = C[(size_t)Y * n + S[a]];
        for (int c2 = 0; c2 < l; c2++) {
            float acc = 0;
            for (int c3 = 0; c3 < l; c3++) acc += m1y[c3] * cx->inv[c3 * l + c2];
            mmy[c2] = acc;
        }
        float h01 = 0, h11 = 0;
        for (int c3 = 0; c3 < l; c3++) h01 += cx->mmx[c3] * m1y[c3];
        for (int c3 = 0; c3 < l; c3++) h11 += mmy[c3] * m1y[c3];
        float H00 = 1 - cx->h00;
        float H01 = M0 - h01;
        float H11 = 1 - h11;
        rho = H01 / (sqrtf(fabsf(H00 * H11)));
        Z = fisher_z_ratio(rho);
    }
    if (rho_out) *rho_out = rho;
    return Z;
}

/* single test, exported: S holds variable indices (not positions) */
float orc_ci_test(const float *C, int n, int X, int Y, const int *S, int l, float *rho_out)
{
    if (l == 0) {
        float r = C[(size_t)X * n + Y];
        if (rho_out) *rho_out = r;
        return fisher_z_ratio(r);
    }
    subset_ctx cx;
    subset_prepare(&cx, C, (size_t)n, X, S, l);
    return subset_test(&cx, C, (size_t)n, X, Y, S, rho_out);
}

/* mean_ess, hetcor-cuPC-S.cu:3068-3088: entries truncated to int (NaN -> 0,
 * never skipped because isnan(int) is false), float running sum. */
static float mean_ess(const float *N, const int *v, int cnt, size_t n)
{
    float s = 0.0f;
    int num = 0;
    for (int i = 0; i < cnt; i++)
        for (int j = 0; j < i; j++) {
            float e = N[(size_t)v[i] * n + v[j]];
            int t;
            if (e != e)
                t = 0;
            else if (e >= 2147483648.0f)
                t = 2147483647;
            else if (e <= -2147483648.0f)
                t = (-2147483647 - 1);
            else
                t = (int)e;
            s += (float)t;
            num += 1;
        }
    return s / (float)num;
}

/* ------------------------------------------------------------------ */
/* level loop                                                          */
/* ------------------------------------------------------------------ */

typedef struct {
    int *off; /* n+1 */
    int *nbr; /* off[n] */
    int maxdeg;
} nbrlist;

static void build_nbr(const int *G, int n, nbrlist *nl)
{
    nl->off = (int *)malloc(sizeof(int) * ((size_t)n + 1));
    size_t tot = 0;
    nl->maxdeg = 0;
    for (int i = 0; i < n; i++) {
        nl->off[i] = (int)tot;
        int d = 0;
        for (int j = 0; j < n; j++) d += (G[(size_t)i * n + j] == 1);
        tot += (size_t)d;
        if (d > nl->maxdeg) nl->maxdeg = d;
    }
    nl->off[n] = (int)tot;
    nl->nbr = (int *)malloc(sizeof(int) * (tot ? tot : 1));
    for (int i = 0; i < n; i++) {
        int k = nl->off[i];
        for (int j = 0; j < n; j++)
            if (G[(size_t)i * n + j] == 1) nl->nbr[k++] = j;
    }
}

/* Common level >= 1 sweep.  mode 0 = Skeleton (fixed th, sepsets, pMax),
 * mode 1 = hetcor (ESS thresholds, time index, G only). */
static void sweep_level(int mode, const float *C, int n, int *G, int l, float th, const float *N,
                        const int *time_index, float *pMax, int *SepSet, long long *tests_out,
                        long long *subsets_out, const nbrlist *nl)
{
    long long tests = 0, subsets = 0;
    size_t nn = (size_t)n;
    unsigned char *found = (unsigned char *)calloc((size_t)nl->off[n] + 1, 1);

#pragma omp parallel for schedule(dynamic, 8) reduction(+ : tests, subsets)
    for (int X = 0; X < n; X++) {
        const int d = nl->off[X + 1] - nl->off[X];
        const int *adj = nl->nbr + nl->off[X];
        unsigned char *fx = found + nl->off[X];
        if (d <= l) continue; /* cuPC-S.cu:611; no (S,Y) pair exists */
        int idx[ORC_ML], S[ORC_ML], vix[ORC_ML + 2];
        for (int i = 0; i < l; i++) idx[i] = i;
        int nfound = 0;
        do {
            for (int i = 0; i < l; i++) S[i] = adj[idx[i]];
            subset_ctx cx;
            subset_prepare(&cx, C, nn, X, S, l);
            subsets++;
            int tmax = 0;
            if (mode == 1) {
                tmax = time_index[S[0]];
                for (int i = 1; i < l; i++)
                    if (time_index[S[i]] > tmax) tmax = time_index[S[i]];
            }
            for (int k2 = 0; k2 < d; k2++) {
                int inS = 0;
                for (int i = 0; i < l; i++) inS |= (idx[i] == k2);
                if (inS) continue;
                if (fx[k2]) continue;
                int Y = adj[k2];
                if (mode == 1) {
                    int tm = time_index[X] > time_index[Y] ? time_index[X] : time_index[Y];
                    if (tmax > tm) continue; /* hetcor-cuPC-S.cu:451,585,3055-3066 */
                }
                tests++;
                float rho;
                float Z = subset_test(&cx, C, nn, X, Y, S, &rho);
                float loc_th = th;
                if (mode == 1) {
                    vix[0] = X;
                    vix[1] = Y;
                    for (int i = 0; i < l; i++) vix[2 + i] = S[i];
                    float me = mean_ess(N, vix, l + 2, nn);
                    loc_th = (float)((double)th / sqrt((double)me - (double)l - 3.0));
                }
                if (Z < loc_th) {
                    fx[k2] = 1;
                    nfound++;
                    if (mode == 0) {
                        pMax[(size_t)X * nn + Y] = Z;
                        for (int i = 0; i < l; i++) SepSet[((size_t)X * nn + Y) * ORC_ML + i] = S[i];
                    }
                }
            }
            if (nfound == d) break; /* nothing left to test from X's side */
        } while (next_comb(idx, l, d));
    }

    for (int X = 0; X < n; X++)
        for (int k = nl->off[X]; k < nl->off[X + 1]; k++)
            if (found[k]) {
                int Y = nl->nbr[k];
                G[(size_t)X * nn + Y] = 0;
                G[(size_t)Y * nn + X] = 0;
            }
    free(found);
    *tests_out = tests;
    *subsets_out = subsets;
}

/* Skeleton(), cuPC-S.cu:61-450.  tests/subsets: 15 counters each (may be NULL).
 * G is fully overwritten, pMax and SepSet as the reference returns them. */
void orc_skeleton(const float *C, int n, int *G, const float *Th, int *l_out, int maxlevel,
                  float *pMax, int *SepSet, long long *tests, long long *subsets)
{
    size_t nn = (size_t)n;
    long long t_dummy[ORC_ML + 1], s_dummy[ORC_ML + 1];
    if (!tests) tests = t_dummy;
    if (!subsets) subsets = s_dummy;
    for (int i = 0; i <= ORC_ML; i++) tests[i] = subsets[i] = 0;
    for (size_t i = 0; i < nn * nn; i++) pMax[i] = 0.0f;
    for (size_t i = 0; i < nn * nn * ORC_ML; i++) SepSet[i] = -1;

    int l;
    for (l = 0; l <= ORC_ML && l <= maxlevel; l++) {
        if (l == 0) {
            for (int row = 0; row < n; row++) {
                for (int col = row + 1; col < n; col++) {
                    float res = fisher_z_ratio(C[row * nn + col]);
                    if (res < Th[0]) {
                        pMax[row * nn + col] = res;
                        pMax[col * nn + row] = res;
                        G[row * nn + col] = 0;
                        G[col * nn + row] = 0;
                    } else {
                        G[row * nn + col] = 1;
                        G[col * nn + row] = 1;
                    }
                }
                G[row * nn + row] = 0;
            }
            tests[0] = (long long)nn * (long long)(nn - 1) / 2;
        } else {
            nbrlist nl;
            build_nbr(G, n, &nl);
            if (nl.maxdeg - 1 < l) {
                free(nl.off);
                free(nl.nbr);
                l = l - 1;
                break;
            }
            sweep_level(0, C, n, G, l, Th[l], NULL, NULL, pMax, SepSet, &tests[l], &subsets[l], &nl);
            free(nl.off);
            free(nl.nbr);
        }
    }
    *l_out = l;

    for (size_t i = 0; i < nn; i++) {
        pMax[i * nn + i] = 1;
        for (size_t j = i + 1; j < nn; j++) {
            if (G[i * nn + j] == 0) {
                float t = fmaxf(pMax[j * nn + i], pMax[i * nn + j]);
                pMax[j * nn + i] = t;
                pMax[i * nn + j] = t;
            } else {
                pMax[j * nn + i] = -100000;
                pMax[i * nn + j] = -100000;
            }
        }
    }
}

/* hetcor_skeleton(), hetcor-cuPC-S.cu:75-341.  G is in/out (level 0 only removes). */
void orc_hetcor_skeleton(const float *C, int n, int *G, const float *N, float th, int *l_out,
                         int maxlevel, const int *time_index, long long *tests,
                         long long *subsets)
{
    size_t nn = (size_t)n;
    long long t_dummy[ORC_ML + 1], s_dummy[ORC_ML + 1];
    if (!tests) tests = t_dummy;
    if (!subsets) subsets = s_dummy;
    for (int i = 0; i <= ORC_ML; i++) tests[i] = subsets[i] = 0;

    int l;
    for (l = 0; l <= ORC_ML && l <= maxlevel; l++) {
        if (l == 0) {
            for (int row = 0; row < n; row++) {
                for (int col = row + 1; col < n; col++) {
                    float res = fisher_z_ratio(C[row * nn + col]);
                    float loc_th = (float)((double)th / sqrt((double)N[row * nn + col] - 3.0));
                    if (res < loc_th) {
                        G[row * nn + col] = 0;
                        G[col * nn + row] = 0;
                    }
                }
                G[row * nn + row] = 0;
            }
            tests[0] = (long long)nn * (long long)(nn - 1) / 2;
        } else {
            nbrlist nl;
            build_nbr(G, n, &nl);
            if (nl.maxdeg - 1 < l) {
                free(nl.off);
                free(nl.nbr);
                l = l - 1;
                break;
            }
            sweep_level(1, C, n, G, l, th, N, time_index, NULL, NULL, &tests[l], &subsets[l], &nl);
            free(nl.off);
            free(nl.nbr);
        }
    }
    *l_out = l;
}

/* ------------------------------------------------------------------ */
/* CPU baseline variant: PC-stable in double precision with the test   */
/* of pcalg::gaussCItest                 */
/* ------------------------------------------------------------------ */

/* The named baseline of BASELINE.json is pcalg::skeleton(method = "stable", indepTest = gaussCItest); pcalg is not
 * in the reference tree and R is not in this image, so this is the same algorithm from pcalg's documented
 * behaviour: order-independent level sweep on frozen neighbour lists, partial correlation of (X, Y | S) in double
 * precision (Cholesky of C[S,S], Schur complement), edge removed when sqrt(N - |S| - 3) * |atanh r| <= qnorm(1 -
 * alpha/2).  OpenMP over rows.  G (n*n, out) and tests[0..14]; returns the last level run.  Not a parity oracle:
 * at threshold-borderline tests and singular sets it may differ from the fp32 reference arithmetic. */
int orc_pcstable_f64(const float *C, int n, int *G, double nsamples, double alpha, int maxlevel, long long *tests)
{
    size_t nn = (size_t)n;
    const double q = fabs(orc_qnorm(alpha / 2.0));
    for (int i = 0; i <= ORC_ML; i++) tests[i] = 0;
    int l;
    for (l = 0; l <= ORC_ML && l <= maxlevel; l++) {
        const double cut = q / sqrt(nsamples - (double)l - 3.0);
        if (l == 0) {
#pragma omp parallel for schedule(static)
            for (int row = 0; row < n; row++) {
                for (int col = 0; col < n; col++) {
                    double r = (double)C[row * nn + col];
                    G[row * nn + col] = (row != col) && !(fabs(atanh(r)) <= cut);
                }
            }
            tests[0] = (long long)nn * (long long)(nn - 1) / 2;
            continue;
        }
        nbrlist nl;
        build_nbr(G, n, &nl);
        if (nl.maxdeg - 1 < l) {
            free(nl.off);
            free(nl.nbr);
            l = l - 1;
            break;
        }
        unsigned char *found = (unsigned char *)calloc((size_t)nl.off[n] + 1, 1);
        long long cnt = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : cnt)
        for (int X = 0; X < n; X++) {
            const int d = nl.off[X + 1] - nl.off[X];
            const int *adj = nl.nbr + nl.off[X];
            unsigned char *fx = found + nl.off[X];
            if (d <= l) continue;
            int idx[ORC_ML], S[ORC_ML];
            double L[ORC_ML][ORC_ML], wx[ORC_ML], wy[ORC_ML];
            for (int i = 0; i < l; i++) idx[i] = i;
            int nfound = 0;
            do {
                for (int i = 0; i < l; i++) S[i] = adj[idx[i]];
                /* Cholesky of C[S,S] (unit diagonal), then w_x = L^-1 C[S,X] */
                int ok = 1;
                for (int i = 0; i < l && ok; i++) {
                    for (int j = 0; j <= i; j++) {
                        double a = (i == j) ? 1.0 : (double)C[(size_t)S[i] * nn + S[j]];
                        for (int k = 0; k < j; k++) a -= L[i][k] * L[j][k];
                        if (i == j) {
                            if (!(a > 1e-14)) { ok = 0; break; }
                            L[i][i] = sqrt(a);
                        } else
                            L[i][j] = a / L[j][j];
                    }
                }
                double hxx = 1.0;
                if (ok) {
                    for (int i = 0; i < l; i++) {
                        double a = (double)C[(size_t)S[i] * nn + X];
                        for (int k = 0; k < i; k++) a -= L[i][k] * wx[k];
                        wx[i] = a / L[i][i];
                        hxx -= wx[i] * wx[i];
                    }
                }
                for (int k2 = 0; k2 < d; k2++) {
                    int inS = 0;
                    for (int i = 0; i < l; i++) inS |= (idx[i] == k2);
                    if (inS || fx[k2]) continue;
                    int Y = adj[k2];
                    cnt++;
                    if (!ok) continue; /* singular conditioning set: NA test, edge kept (the reference's behaviour) */
                    double hyy = 1.0, hxy = (double)C[(size_t)X * nn + Y];
                    for (int i = 0; i < l; i++) {
                        double a = (double)C[(size_t)S[i] * nn + Y];
                        for (int k = 0; k < i; k++) a -= L[i][k] * wy[k];
                        wy[i] = a / L[i][i];
                        hyy -= wy[i] * wy[i];
                        hxy -= wx[i] * wy[i];
                    }
                    double r = hxy / sqrt(hxx * hyy);
                    if (r > 1.0) r = 1.0;
                    if (r < -1.0) r = -1.0;
                    if (fabs(atanh(r)) <= cut) {
                        fx[k2] = 1;
                        nfound++;
                    }
                }
                if (nfound == d) break;
            } while (next_comb(idx, l, d));
        }
        tests[l] = cnt;
        for (int X = 0; X < n; X++)
            for (int k = nl.off[X]; k < nl.off[X + 1]; k++)
                if (found[k]) {
                    int Y = nl.nbr[k];
                    G[(size_t)X * nn + Y] = 0;
                    G[(size_t)Y * nn + X] = 0;
                }
        free(found);
        free(nl.off);
        free(nl.nbr);
    }
    return l;
}

/* ------------------------------------------------------------------ */
/* correlation build: corr_kernels.cu                                  */
/* ------------------------------------------------------------------ */

/* .bed 2-bit codes, low bits first (bed_lut_gpu.h): 00 -> 2, 01 -> missing,
 * 10 -> 1, 11 -> 0. */
static inline void bed_decode(unsigned char byte, int j, float *val, float *valid)
{
    int code = (byte >> (2 * j)) & 3;
    switch (code) {
        case 0: *val = 2.0f; *valid = 1.0f; break;
        case 1: *val = 0.0f; *valid = 0.0f; break;
        case 2: *val = 1.0f; *valid = 1.0f; break;
        default: *val = 0.0f; *valid = 1.0f; break;
    }
}

#define NUMTHREADS 512 /* corr_kernels.h:3 */

/* value that the reference's Hillis-Steele scan leaves in slot NUMTHREADS-1 */
static float scan_last(float *x)
{
    float tmp[NUMTHREADS];
    for (int step = 1; step < NUMTHREADS; step *= 2) {
        for (int t = 0; t < NUMTHREADS; t++) tmp[t] = (t < step) ? x[t] : x[t] + x[t - step];
        memcpy(x, tmp, sizeof(tmp));
    }
    return x[NUMTHREADS - 1];
}

/* bed_marker_phen_corr_pearson_scan, corr_kernels.cu:157-238 */
void orc_marker_phen_corr_pearson(const unsigned char *bed, const float *phen, size_t m, size_t N,
                                  size_t p, const float *mean, const float *std, float *out)
{
    size_t clb = (N + 3) / 4;
#pragma omp parallel for schedule(dynamic, 4)
    for (long long lin = 0; lin < (long long)(m * p); lin++) {
        size_t mv = (size_t)lin / p, ph = (size_t)lin - p * mv;
        float s_gy[NUMTHREADS], s_y[NUMTHREADS], s_n[NUMTHREADS];
        for (int t = 0; t < NUMTHREADS; t++) {
            float a = 0.0f, b = 0.0f, c = 0.0f;
            for (size_t i = (size_t)t; i < clb; i += NUMTHREADS) {
                unsigned char byte = bed[mv * clb + i];
                for (size_t j = 0; (j < 4) && (i * 4 + j < N); j++) {
                    float val, valid;
                    bed_decode(byte, (int)j, &val, &valid);
                    float y = phen[ph * N + 4 * i + j];
                    if (!(y != y)) {
                        a += valid * val * y;
                        b += valid * y;
                        c += valid;
                    }
                }
            }
            s_gy[t] = a;
            s_y[t] = b;
            s_n[t] = c;
        }
        float sgy = scan_last(s_gy), sy = scan_last(s_y), nv = scan_last(s_n);
        out[lin] = (sgy - mean[mv] * sy) / (nv * std[mv]);
    }
}

/* row/col of the linear index into the upper triangle without diagonal */
static void tri_rowcol(size_t lin, size_t n, size_t *row, size_t *col)
{
    size_t r = 0, rem = lin, len = n - 1;
    while (rem >= len) {
        rem -= len;
        len--;
        r++;
    }
    *row = r;
    *col = r + 1 + rem;
}

/* phen_corr_pearson_scan, corr_kernels.cu:285-343 */
void orc_phen_corr_pearson(const float *phen, size_t N, size_t p, float *out)
{
    size_t npairs = p * (p - 1) / 2;
    for (size_t lin = 0; lin < npairs; lin++) {
        size_t ra, cb;
        tri_rowcol(lin, p, &ra, &cb);
        float s[NUMTHREADS], c[NUMTHREADS];
        for (int t = 0; t < NUMTHREADS; t++) {
            float a = 0.0f, b = 0.0f;
            for (size_t i = (size_t)t; i < N; i += NUMTHREADS) {
                float va = phen[ra * N + i], vb = phen[cb * N + i];
                if (!((va != va) || (vb != vb))) {
                    a += va * vb;
                    b++;
                }
            }
            s[t] = a;
            c[t] = b;
        }
        float ss = scan_last(s), nn = scan_last(c);
        out[lin] = ss / nn;
    }
}

/* Kendall tau-b -> sin(pi/2 tau) from the 3x3 table, corr_kernels.cu:544-564.
 * s[] are the nine exact counts as floats. */
float orc_npn_from_counts(const float *s)
{
    float p = ((s[0] * (s[4] + s[5] + s[7] + s[8])) + (s[1] * (s[5] + s[8])) +
               (s[3] * (s[7] + s[8])) + (s[4] * s[8]));
    float q = ((s[1] * (s[3] + s[6])) + (s[2] * (s[3] + s[4] + s[6] + s[7])) + (s[4] * s[6]) +
               (s[5] * (s[6] + s[7])));
    float t = ((s[0] * (s[1] + s[2])) + (s[1] * s[2]) + (s[3] * (s[4] + s[5])) + (s[4] * s[5]) +
               (s[6] * (s[7] + s[8])) + (s[7] * s[8]));
    float u = ((s[0] * (s[3] + s[6])) + (s[1] * (s[4] + s[7])) + (s[2] * (s[5] + s[8])) +
               (s[3] * s[6]) + (s[4] * s[7]) + (s[5] * s[8]));
    float kendall = (p - q) / sqrtf((p + q + t) * (p + q + u));
    return (float)sin(M_PI / 2 * (double)kendall);
}

/* bed_marker_corr_pearson_npn_scan, corr_kernels.cu:478-565.  The nine sums
 * are integer counts < 2^24, so any summation order gives the same floats. */
void orc_marker_corr_npn(const unsigned char *bed, size_t m, size_t N, float *out)
{
    size_t clb = (N + 3) / 4;
    long long npairs = (long long)(m * (m - 1) / 2);
#pragma omp parallel for schedule(dynamic, 64)
    for (long long lin = 0; lin < npairs; lin++) {
        size_t ra, cb;
        tri_rowcol((size_t)lin, m, &ra, &cb);
        unsigned cnt[9] = {0};
        for (size_t i = 0; i < clb; i++) {
            unsigned char ba = bed[ra * clb + i], bb = bed[cb * clb + i];
            for (size_t j = 0; (j < 4) && (i * 4 + j < N); j++) {
                float va, vb, oka, okb;
                bed_decode(ba, (int)j, &va, &oka);
                bed_decode(bb, (int)j, &vb, &okb);
                if (oka * okb != 0.0f) cnt[(int)(3 * va + vb)]++;
            }
        }
        float s[9];
        for (int i = 0; i < 9; i++) s[i] = (float)cnt[i];
        out[lin] = orc_npn_from_counts(s);
    }
}

/* cu_corr_pearson_npn, corr_host.cu:1094-1197: mxm (npn), mxp, pxp */
void orc_corr_pearson_npn(const unsigned char *bed, const float *phen, size_t m, size_t N, size_t p,
                          const float *mean, const float *std, float *mxm, float *mxp, float *pxp)
{
    orc_marker_corr_npn(bed, m, N, mxm);
    orc_marker_phen_corr_pearson(bed, phen, m, N, p, mean, std, mxp);
    orc_phen_corr_pearson(phen, N, p, pxp);
}

/* ------------------------------------------------------------------------- */
/* mps block: banded Kendall-npn correlations, forward row sums,  */
/* Hanning smoothing, local minima, bisection on the window size                */
/* ------------------------------------------------------------------------- */

/* cal_mcorrk_banded / cu_marker_corr_pearson_npn_batched_sparse, corr_host.cu:65-110,1199-1319 (single batch):
 * out[row * width + col] = npn(row, row + 1 + col), 0 where row + 1 + col >= m */
void orc_marker_corr_banded(const unsigned char *bed, size_t m, size_t N, size_t width, float *out)
{
    size_t clb = (N + 3) / 4;
#pragma omp parallel for schedule(dynamic, 16)
    for (long long row = 0; row < (long long)m; row++) {
        for (size_t col = 0; col < width; col++) {
            size_t cb = (size_t)row + 1 + col;
            float r = 0.0f;
            if (cb < m) {
                unsigned cnt[9] = {0};
                for (size_t i = 0; i < clb; i++) {
                    unsigned char ba = bed[(size_t)row * clb + i], bb = bed[cb * clb + i];
                    for (size_t j = 0; (j < 4) && (i * 4 + j < N); j++) {
                        float va, vb, oka, okb;
                        bed_decode(ba, (int)j, &va, &oka);
                        bed_decode(bb, (int)j, &vb, &okb);
                        if (oka * okb != 0.0f) cnt[(int)(3 * va + vb)]++;
                    }
                }
                float s9[9];
                for (int i = 0; i < 9; i++) s9[i] = (float)cnt[i];
                r = orc_npn_from_counts(s9);
            }
            out[(size_t)row * width + col] = r;
        }
    }
}

/* marker_corr_banded_mat_row_abs_sums, corr_host.cu:112-128: float accumulation, columns in order */
void orc_banded_row_abs_sums(const float *band, size_t m, size_t width, float *sums)
{
    for (size_t row = 0; row < m; row++) {
        float acc = 0.0f;
        for (size_t col = 0; col < width; col++) acc += fabsf(band[row * width + col]);
        sums[row] = acc;
    }
}

/* blocking.cpp:8-11: the cosine is the single-precision one */
static double hann(int n, int m) { return 0.5 - 0.5 * cosf(2.0 * M_PI * (double)n / ((double)m - 1.0)); }

/* blocking.cpp:13-35 */
void orc_hanning_smoothing(const float *v, int n, int window_size, double *res)
{
    double *win = (double *)malloc(sizeof(double) * (size_t)(window_size > 0 ? window_size : 1));
    for (int i = 0; i < window_size; i++) win[i] = hann(i, window_size);
    int margin = window_size / 2;
    for (int i = 0; i < n; i++) res[i] = 0.0;
    for (int center = margin; center < n - margin; center++)
        for (int i = 0; i < window_size; i++) res[center] += win[i] * (double)v[center - margin + i];
    free(win);
}

/* blocking.cpp:37-56,58-70,72-83,85-136.  first/last receive the blocks (chromosome-local marker indices);
 * returns their number, or -1 when cap is too small */
int orc_block_chr(const float *v, int n, int max_block_size, long long *first, long long *last, int cap)
{
    const int tol = 100; /* MAX_BLOCK_SIZE_TOL */
    double *smooth = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int nb = 0, lbs = 0;
    int too_large = n, too_small = 3;
    int window_size = (too_large + too_small) / 2;
    if (window_size % 2 == 0) window_size -= 1;
    for (;;) {
        orc_hanning_smoothing(v, n, window_size, smooth);
        /* local minima -> blocks */
        nb = 0;
        lbs = 0;
        long long prev = 0;
        double left = 0.0;
        int overflow = 0;
        for (long long i = 1; i < (long long)n - 1; i++) {
            if ((left > smooth[i]) && (smooth[i] < smooth[i + 1])) {
                if (nb < cap) { first[nb] = prev; last[nb] = i; } else overflow = 1;
                if ((int)(i - prev + 1) > lbs) lbs = (int)(i - prev + 1);
                nb++;
                prev = i + 1;
                left = 0.0;
            } else if (smooth[i] > left) {
                left = smooth[i];
            }
        }
        if (nb < cap) { first[nb] = prev; last[nb] = (long long)n - 1; } else overflow = 1;
        if ((int)((long long)n - 1 - prev + 1) > lbs) lbs = (int)((long long)n - prev);
        nb++;
        if (overflow) { free(smooth); return -1; }
        if (!((abs(lbs - max_block_size) > tol) || (lbs > max_block_size))) break;
        if (lbs > max_block_size) { if (window_size < too_large) too_large = window_size; }
        else { if (window_size > too_small) too_small = window_size; }
        int nw = (too_large + too_small) / 2;
        if (nw % 2 == 0) nw -= 1;
        if (nw == window_size) break;
        window_size = nw;
    }
    free(smooth);
    return nb;
}

