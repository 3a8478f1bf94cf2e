// ci_fast.h -- fast evaluation of conditional-independence tests with a certified verdict.
//
// The reference inverts C[S,S] with an iterative SVD per conditioning set
// (/root/reference/cusk/src/cuPC-S.cu:3463-3724) and multiplies the l x l inverse into
// every test (:1002-1023).  Here the conditioning block is factored once per set by an
// unrolled register Cholesky (C[S,S] = F F^T, unit diagonal) and every test costs one
// forward substitution:
//     a = F^-1 C[S,X]  (per set)      b = F^-1 C[S,Y]  (per test)
//     H00 = 1 - a.a    H11 = 1 - b.b   H01 = C[X,Y] - a.b     rho = H01 / sqrt(H00 H11)
// The result is NOT used as the answer.  It is a filter: with t = tanh(th),
//     rho^2 < t^2 (1-beta)  -> certainly  z < th   (edge separated by S)
//     rho^2 > t^2 (1+beta)  -> certainly  z >= th
// and everything in between, plus every test whose factorisation is not comfortably
// conditioned (min pivot / H00 / H11 below kCondMin) or produced a non-finite value, is
// queued and re-evaluated by recheck_kernel on the exact path (ci_exact.h: the
// reference's operation order).  Adjacency and separating sets are therefore those of
// the exact arithmetic; the fast path only decides which tests need it.
//
// beta = 2^-9 on rho^2 (i.e. +-1e-3 relative on |rho|) against fp32 evaluation errors of
// <= ~1e-5 relative on well-conditioned sets leaves two orders of
// magnitude of margin; `validate` runs (cusk_engine_set_option) evaluate both paths for
// every test and count disagreements outside the band (must be zero; tests assert it).
#pragma once
#include <hip/hip_runtime.h>

namespace cusk {

constexpr float kBeta = 1.0f / 512.0f;
constexpr float kCondMin = 1.0f / 64.0f;

enum Verdict : int
{
    kFail = 0,    // certainly z >= th: edge stays (for this S)
    kPass = 1,    // certainly z <  th
    kUnsure = 2   // needs the exact path
};

template <int L>
struct SubsetFast
{
    static constexpr int NL = (L > 1) ? L * (L - 1) / 2 : 1;
    float f[NL];    // strict lower triangle of F, row-major packed: (i,j), j<i at i(i-1)/2+j
    float invd[L];  // 1 / F_ii
    float a[L];     // F^-1 C[S,X]
    float h00;
    bool ill;

    // c: strict lower triangle of C[S,S] in the same packing; m1x: C[X,S]
    __device__ __forceinline__ void prepare(const float *c, const float *m1x)
    {
        float dmin = 1.0f;
#pragma unroll
        for (int i = 0; i < L; i++)
        {
            float dii = 1.0f;
#pragma unroll
            for (int j = 0; j < i; j++)
            {
                float s = c[i * (i - 1) / 2 + j];
#pragma unroll
                for (int k = 0; k < j; k++) s = __builtin_fmaf(-f[i * (i - 1) / 2 + k], f[j * (j - 1) / 2 + k], s);
                const float lij = s * invd[j];
                f[i * (i - 1) / 2 + j] = lij;
                dii = __builtin_fmaf(-lij, lij, dii);
            }
            dmin = fminf(dmin, dii);
            invd[i] = __frsqrt_rn(dii);
        }
        float hh = 1.0f;
#pragma unroll
        for (int i = 0; i < L; i++)
        {
            float s = m1x[i];
#pragma unroll
            for (int k = 0; k < i; k++) s = __builtin_fmaf(-f[i * (i - 1) / 2 + k], a[k], s);
            a[i] = s * invd[i];
            hh = __builtin_fmaf(-a[i], a[i], hh);
        }
        h00 = hh;
        ill = !(dmin >= kCondMin) || !(hh >= kCondMin);
    }

    // Incremental form: rows >= first of the factor (and of a) are recomputed, the others are kept from the
    // previous conditioning set, which shares its first `first` members with this one (lexicographic successor).
    // Row i of F and a[i] depend on S_0..S_i only, so the kept rows are exactly what a full factorisation would
    // give; operands are fetched where they are consumed (c_of(i, j) = C[S_i, S_j], j < i; x_of(i) = C[X, S_i]).
    // The pivot guard is read off the stored reciprocal square roots (d_ii >= 1/64  <=>  1/sqrt(d_ii) <= 8).
    template <typename FC, typename FX>
    __device__ __forceinline__ void prepare_rows(int first, FC c_of, FX x_of)
    {
#pragma unroll
        for (int i = 0; i < L; i++)
        {
            if (i < first) continue;
            float dii = 1.0f;
#pragma unroll
            for (int j = 0; j < i; j++)
            {
                float s = c_of(i, j);
#pragma unroll
                for (int k = 0; k < j; k++) s = __builtin_fmaf(-f[i * (i - 1) / 2 + k], f[j * (j - 1) / 2 + k], s);
                const float lij = s * invd[j];
                f[i * (i - 1) / 2 + j] = lij;
                dii = __builtin_fmaf(-lij, lij, dii);
            }
            invd[i] = __frsqrt_rn(dii);
            float s = x_of(i);
#pragma unroll
            for (int k = 0; k < i; k++) s = __builtin_fmaf(-f[i * (i - 1) / 2 + k], a[k], s);
            a[i] = s * invd[i];
        }
        float hh = 1.0f;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < L; i++)
        {
            hh = __builtin_fmaf(-a[i], a[i], hh);
            bad = bad || !(invd[i] <= 8.0f);
        }
        h00 = hh;
        ill = bad || !(hh >= kCondMin);
    }

    // m0 = C[X,Y], m1y = C[Y,S]; returns H01 and H11
    __device__ __forceinline__ void schur(float m0, const float *m1y, float &h01, float &h11) const
    {
        float b[L];
        float s11 = 1.0f, s01 = m0;
#pragma unroll
        for (int i = 0; i < L; i++)
        {
            float s = m1y[i];
#pragma unroll
            for (int k = 0; k < i; k++) s = __builtin_fmaf(-f[i * (i - 1) / 2 + k], b[k], s);
            b[i] = s * invd[i];
            s11 = __builtin_fmaf(-b[i], b[i], s11);
            s01 = __builtin_fmaf(-a[i], b[i], s01);
        }
        h01 = s01;
        h11 = s11;
    }

    // verdict against a fixed threshold given as t2 = tanh(th)^2
    __device__ __forceinline__ int verdict_fixed(float m0, const float *m1y, float t2) const
    {
        float h01, h11;
        schur(m0, m1y, h01, h11);
        const float prod = h00 * h11;
        const float lhs = h01 * h01;
        const float rhs = t2 * prod;
        if (!(h11 >= kCondMin)) return kUnsure;
        if (lhs < rhs * (1.0f - kBeta)) return kPass;
        if (lhs > rhs * (1.0f + kBeta)) return kFail;
        return kUnsure;
    }

    // verdict against a per-test Fisher-z threshold lth (approximate, float)
    __device__ __forceinline__ int verdict_z(float m0, const float *m1y, float lth) const
    {
        float h01, h11;
        schur(m0, m1y, h01, h11);
        if (!(h11 >= kCondMin)) return kUnsure;
        const float rho = h01 * __frsqrt_rn(h00 * h11);
        const float z = 0.5f * fabsf(__logf(fabsf((1.0f + rho) / (1.0f - rho))));
        const float band = lth * (0.5f * kBeta) + 2e-6f;
        if (z < lth - band) return kPass;
        if (z > lth + band) return kFail;
        return kUnsure;
    }
};

}  // namespace cusk
