// Zero-edit link boundary: this translation unit sees ONLY the reference's own headers for the correlation and
// threshold entry points (compiled with -I/root/reference/cusk/include in the build container) and links against
// libcusk_hip.so.  It resolves iff the library exports the reference's C++-linkage (mangled) names with the reference's
// signatures (csrc/compat_cxx.cpp).  Running it needs no GPU: only the threshold functions are called; the
// correlation entry points are referenced through their addresses.
#include <mps/corr_host.h>
#include <mps/cuPC_call_prep.h>

#include <cstdio>

// the two engine entry points are extern "C" in the reference (include/mps/cuPC-S.h:196-198, hetcor-cuPC-S.h:46; those
// headers carry CUDA __global__ declarations and are not host-includable): same declarations, verbatim signatures
extern "C" void Skeleton(float *C, int *P, int *G, float *Th, int *l, const int *maxlevel, float *pMax, int *SepSet);
extern "C" void hetcor_skeleton(float *C, int *P, int *G, float *N, float *Th, int *l, const int *maxlevel,
                                const int *time_index);

int main()
{
    const std::vector<float> thr = threshold_array(500000, 1e-8f);  // cupc_tests.cpp:10-15
    std::printf("%zu %.9g %.9g %.9g\n", thr.size(), thr[0], hetcor_threshold(1e-4f), std_normal_qnorm(0.025f));
    void (*f1)(const unsigned char *, const float *, const size_t, const size_t, const size_t, const float *, const float *,
               float *) = &cu_marker_phen_corr_pearson;
    void (*f2)(const unsigned char *, const float *, const size_t, const size_t, const size_t, const float *, const float *,
               float *, float *, float *) = &cu_corr_pearson_npn;
    void (*f3)(float *, int *, int *, float *, int *, const int *, float *, int *) = &Skeleton;
    void (*f4)(float *, int *, int *, float *, float *, int *, const int *, const int *) = &hetcor_skeleton;
    std::printf("%d\n", (f1 != nullptr) + (f2 != nullptr) + (f3 != nullptr) + (f4 != nullptr));
    return 0;
}
