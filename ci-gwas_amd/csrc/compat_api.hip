// compat_api.hip -- the reference's own entry points on top of the device engine.
// Same names, argument order and host-buffer ownership as
// /root/reference/cusk/include/mps/cuPC-S.h:196-198 (Skeleton) and
// include/mps/hetcor-cuPC-S.h:46 (hetcor_skeleton); errors print and exit like
// include/mps/gpuerrors.h:6-15.
#include <cstdio>
#include <cstdlib>

#include "cusk_internal.h"

namespace {

[[noreturn]] void die(const char *what, const cusk_engine *e)
{
    std::fprintf(stderr, "libcusk_hip: %s: %s\n", what, e ? cusk_last_error(e) : "no HIP device / engine");
    std::exit(EXIT_FAILURE);
}

cusk_engine *make_engine()
{
    cusk_engine *e = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (cusk_engine_create(&e, dev, nullptr) != CUSK_OK) die("engine create", nullptr);
    return e;
}

template <typename T>
T *upload(const T *host, size_t count, cusk_engine *e)
{
    T *d = static_cast<T *>(cusk_dev_alloc(sizeof(T) * count));
    if (!d || cusk_dev_upload(d, host, sizeof(T) * count) != CUSK_OK) die("upload", e);
    return d;
}

}  // namespace

extern "C" void Skeleton(float *C, int *P, int *G, float *Th, int *l, const int *maxlevel, float *pMax, int *SepSet)
{
    const int n = *P;
    cusk_engine *e = make_engine();
    float *Cd = upload(C, (size_t)n * n, e);
    cusk_stats st;
    if (cusk_run_skeleton(e, Cd, n, Th, *maxlevel, &st) != CUSK_OK) die("Skeleton", e);
    *l = st.level;
    if (cusk_result_adj_i32(e, G) != CUSK_OK) die("Skeleton adjacency", e);
    if (pMax && cusk_result_pmax(e, Cd, pMax) != CUSK_OK) die("Skeleton pMax", e);
    if (SepSet && cusk_result_sepset_dense(e, SepSet) != CUSK_OK) die("Skeleton sepsets", e);
    cusk_dev_free(Cd);
    cusk_engine_destroy(e);
}

extern "C" void hetcor_skeleton(float *C, int *P, int *G, float *N, float *Th, int *l, const int *maxlevel,
                                const int *time_index)
{
    const int n = *P;
    cusk_engine *e = make_engine();
    float *Cd = upload(C, (size_t)n * n, e);
    float *Nd = upload(N, (size_t)n * n, e);
    int *Gd = upload(G, (size_t)n * n, e);
    cusk_stats st;
    if (cusk_run_hetcor(e, Cd, Nd, 0.0f, Gd, n, *Th, *maxlevel, time_index, &st) != CUSK_OK) die("hetcor_skeleton", e);
    *l = st.level;
    if (cusk_result_adj_i32(e, G) != CUSK_OK) die("hetcor_skeleton adjacency", e);
    cusk_dev_free(Gd);
    cusk_dev_free(Nd);
    cusk_dev_free(Cd);
    cusk_engine_destroy(e);
}

// corr_host.h:38-47, 92-103 with C linkage (ctypes, C hosts); the C++-linkage twins the reference's own callers bind to
// live in compat_cxx.cpp
extern "C" void cu_marker_phen_corr_pearson(const unsigned char *marker_vals, const float *phen_vals, const size_t num_markers,
                                            const size_t num_individuals, const size_t num_phen, const float *marker_mean,
                                            const float *marker_std, float *marker_phen_corrs)
{
    cusk::compat_marker_phen_corr_pearson(marker_vals, phen_vals, num_markers, num_individuals, num_phen, marker_mean, marker_std,
                                          marker_phen_corrs);
}

extern "C" void cu_corr_pearson_npn(const unsigned char *marker_vals, const float *phen_vals, const size_t num_markers,
                                    const size_t num_individuals, const size_t num_phen, const float *marker_mean,
                                    const float *marker_std, float *marker_corrs, float *marker_phen_corrs, float *phen_corrs)
{
    cusk::compat_corr_pearson_npn(marker_vals, phen_vals, num_markers, num_individuals, num_phen, marker_mean, marker_std,
                                  marker_corrs, marker_phen_corrs, phen_corrs);
}
