// thresholds.cpp -- significance thresholds of the level sweep.
// Replaces /root/reference/cusk/src/cuPC_call_prep.cpp:7-27, which calls
// boost::math::quantile(normal(0,1), p) (boost is not available here and is
// not vendored by the reference).  The quantile is Wichura's AS 241 PPND16
// (relative accuracy ~1e-16) plus one Newton step on erfc, i.e. a
// double-accurate inverse like boost's; the float narrowing points are the
// reference's: float p in, float quantile out, double division by sqrt(dof).
#include <cmath>
#include <cstddef>

#include "cusk_internal.h"

namespace cusk {

static double ppnd16(double p)
{
    const double q = p - 0.5;
    double r, val;
    if (std::fabs(q) <= 0.425)
    {
        r = 0.180625 - q * q;
        val = q *
              (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r +
                   4.5921953931549871457e4) * r + 1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r +
                1.3314166789178437745e2) * r + 3.3871328727963666080e0) /
              (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r +
                   2.1213794301586595867e4) * r + 5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r +
                4.2313330701600911252e1) * r + 1.0);
        return val;
    }
    r = (q < 0.0) ? p : 1.0 - p;
    if (r <= 0.0) return (q < 0.0) ? -INFINITY : INFINITY;
    r = std::sqrt(-std::log(r));
    if (r <= 5.0)
    {
        r -= 1.6;
        val = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r +
                   1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r +
                4.63033784615654529590e0) * r + 1.42343711074968357734e0) /
              (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r +
                   1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r +
                2.05319162663775882187e0) * r + 1.0);
    }
    else
    {
        r -= 5.0;
        val = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r +
                   2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r +
                5.46378491116411436990e0) * r + 6.65790464350110377720e0) /
              (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r +
                   7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                5.99832206555887937690e-1) * r + 1.0);
    }
    return (q < 0.0) ? -val : val;
}

static double qnorm(double p)
{
    double x = ppnd16(p);
    if (std::isfinite(x))
    {
        const double e = 0.5 * std::erfc(-x / std::sqrt(2.0)) - p;
        x -= e * std::sqrt(2.0 * M_PI) * std::exp(0.5 * x * x);
    }
    return x;
}

double qnorm_host(double p) { return qnorm(p); }

void threshold_array_host(int n, float alpha, float *thr15)
{
    const float half = 0.5f;
    for (size_t i = 0; i < (size_t)kML + 1; i++)
    {
        const float q = std::fabs((float)qnorm((double)(half * alpha)));
        const size_t dof = (size_t)n - i - 3;
        thr15[i] = (float)((double)q / std::sqrt((double)dof));
    }
}

float hetcor_threshold_host(float alpha) { return std::fabs((float)qnorm((double)(float)(0.5 * (double)alpha))); }

}  // namespace cusk

extern "C" void cusk_threshold_array(int n, float alpha, float *thr15) { cusk::threshold_array_host(n, alpha, thr15); }
extern "C" float cusk_hetcor_threshold(float alpha) { return cusk::hetcor_threshold_host(alpha); }
