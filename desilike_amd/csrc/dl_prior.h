// Log-prior of one parameter (row a9: desilike/parameter.py:1994-2017), maximum removed ("zero lag", 2003-2016), shared by every finalize path.
//   Table row pr[5] = (kind, lo, hi, loc, scale) (include/desilike_amd.h, key "priors").  kind 0: uniform (0 inside the closed limits), 1: norm
//   (-1/2 ((x - loc) / scale)^2, line 2007); kinds >= 2: the location-scale families of scipy.stats whose density is finite and non-zero at loc, evaluated as the
//   reference does through `rv.logpdf(x) - rv.logpdf(loc)` (2012-2016) -- the normalisation cancels, what is left is u(y) - u(0) with y = (x - loc) / scale:
//     2 expon      -y (y >= 0)                     3 laplace    -|y|                      4 cauchy     -log1p(y^2)
//     5 logistic   -|y| - 2 log1p(e^-|y|) + 2 ln 2 6 halfnorm   -y^2 / 2 (y >= 0)         7 halfcauchy -log1p(y^2) (y >= 0)
//     8 gumbel_r   -(y + e^-y) + 1                 9 gumbel_l   y - e^y + 1
//   -inf outside the limits or the support.
#pragma once
#include <hip/hip_runtime.h>

#define DL_PRIOR_MAX_KIND 9

__device__ __forceinline__ double dl_prior_family(int kind, double y) {
    const double inf = __builtin_huge_val();
    switch (kind) {
        case 2: return y >= 0. ? -y : -inf;
        case 3: return -fabs(y);
        case 4: return -log1p(y * y);
        case 5: { const double a = fabs(y); return (-a - 2. * log1p(exp(-a))) + 1.3862943611198906; }
        case 6: return y >= 0. ? -0.5 * (y * y) : -inf;
        case 7: return y >= 0. ? -log1p(y * y) : -inf;
        case 8: return 1. - (y + exp(-y));
        case 9: return (y - exp(y)) + 1.;
        default: return 0.;
    }
}

// value added to the log-prior by parameter value x (NaN x: the caller flags it; the comparison chain then yields -inf)
__device__ __forceinline__ double dl_prior_logpdf(const double* __restrict__ pr, double x) {
    const double inf = __builtin_huge_val();
    // the five entries of the row are read together, ahead of the comparisons (read where they are used -- behind the short-circuit of the limits, behind the test
    // of the kind -- a row in LDS was three dependent round trips per parameter: 0.21 us per parameter in the ensemble step kernel)
    const double kind = pr[0], lo = pr[1], hi = pr[2], loc = pr[3], scale = pr[4];
    const bool isin = (lo <= x) & (x <= hi);
    double v = 0.;
    if (kind == 1.) { const double t = x - loc; v = -0.5 * (t * t) / (scale * scale); }   // parameter.py:2007
    else if (kind >= 2.) v = dl_prior_family((int)kind, (x - loc) / scale);
    return isin ? v : -inf;
}
