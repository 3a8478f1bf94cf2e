// compat_cxx.cpp -- the reference's C++-linkage entry points for this path, so that its own translation units link
// against libcusk_hip.so without an edit: a TU that includes <mps/corr_host.h> / <mps/cuPC_call_prep.h>
// (/root/reference/cusk/src/cli.cpp:1-25) refers to the MANGLED names
//   cu_marker_phen_corr_pearson, cu_corr_pearson_npn        include/mps/corr_host.h:38-47, 92-103
//   threshold_array, hetcor_threshold, std_normal_qnorm     include/mps/cuPC_call_prep.h:5-15
// (Skeleton / hetcor_skeleton are extern "C" in the reference as well: compat_api.hip).  This file must not see
// include/cusk_hip.h: there the two correlation functions are declared with C linkage, and C++ forbids both linkages
// for one signature in one translation unit -- the same bodies (cusk::compat_*) serve both.
#include <cmath>
#include <cstddef>
#include <vector>

namespace cusk {
void compat_marker_phen_corr_pearson(const unsigned char *marker_vals, const float *phen_vals, const size_t num_markers,
                                     const size_t num_individuals, const size_t num_phen, const float *marker_mean,
                                     const float *marker_std, float *marker_phen_corrs);
void compat_corr_pearson_npn(const unsigned char *marker_vals, const float *phen_vals, const size_t num_markers,
                             const size_t num_individuals, const size_t num_phen, const float *marker_mean,
                             const float *marker_std, float *marker_corrs, float *marker_phen_corrs, float *phen_corrs);
void threshold_array_host(int n, float alpha, float *thr15);
float hetcor_threshold_host(float alpha);
double qnorm_host(double p);
}  // namespace cusk

#define CUSK_EXPORT __attribute__((visibility("default")))

CUSK_EXPORT void cu_marker_phen_corr_pearson(const unsigned char *marker_vals, const float *phen_vals, const size_t num_markers,
                                             const size_t num_individuals, const size_t num_phen, const float *marker_mean,
                                             const float *marker_std, float *marker_phen_corrs)
{
    cusk::compat_marker_phen_corr_pearson(marker_vals, phen_vals, num_markers, num_individuals, num_phen, marker_mean, marker_std,
                                          marker_phen_corrs);
}

CUSK_EXPORT void cu_corr_pearson_npn(const unsigned char *marker_vals, const float *phen_vals, const size_t num_markers,
                                     const size_t num_individuals, const size_t num_phen, const float *marker_mean,
                                     const float *marker_std, float *marker_corrs, float *marker_phen_corrs, float *phen_corrs)
{
    cusk::compat_corr_pearson_npn(marker_vals, phen_vals, num_markers, num_individuals, num_phen, marker_mean, marker_std,
                                  marker_corrs, marker_phen_corrs, phen_corrs);
}

// cuPC_call_prep.cpp:7-11 (boost::math::quantile of the standard normal, evaluated in double, returned as float)
CUSK_EXPORT float std_normal_qnorm(const float p) { return (float)cusk::qnorm_host((double)p); }

// cuPC_call_prep.cpp:13-23
CUSK_EXPORT std::vector<float> threshold_array(const int n, const float alpha)
{
    std::vector<float> thr(15);
    cusk::threshold_array_host(n, alpha, thr.data());
    return thr;
}

// cuPC_call_prep.cpp:25-27
CUSK_EXPORT float hetcor_threshold(const float alpha) { return cusk::hetcor_threshold_host(alpha); }
