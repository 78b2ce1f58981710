// chi2 / prior / status of ONE point from the partial chi2 of the column blocks of the chi2 GEMM (dl_chi2_gemm.h): shared by dl_finalize_part_kernel and by the
// ensemble step kernel (dl_ensemble.hip), which finishes the pending half-step's proposals itself instead of waiting for a separate finalize launch.
// Priors: parameter.py:1994-2017 (uniform / norm, maximum removed, closed limits); status rules of include/desilike_amd.h.
#pragma once
#include "dl_prior.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ double dl_chi2_of_parts(const double* __restrict__ part_row, int n_tiles) {
    double chi2 = 0.;
    for (int t = 0; t < n_tiles; ++t) chi2 += part_row[t];   // fixed order: deterministic
    return chi2;
}

__device__ __forceinline__ void dl_finalize_from_chi2(double chi2, const double* __restrict__ theta_row, int n_params, const double* __restrict__ priors, double& ll, double& lp,
                                                      int& st) {
    lp = 0.;
    bool nan_in = false;
    const double inf = __builtin_huge_val();
    for (int p = 0; p < n_params; ++p) {
        double x = theta_row[p];
        const double* pr = priors + 5 * p;
        if (x != x) nan_in = true;
        lp += dl_prior_logpdf(pr, x);
    }
    ll = -0.5 * chi2;
    st = 0;                                                                   // DL_STATUS_OK
    if (nan_in) st = 3;                                                       // DL_STATUS_NAN_INPUT
    else if (lp == -inf) st = 1;                                              // DL_STATUS_OUT_OF_PRIOR
    else if (!(ll == ll) || ll == inf || ll == -inf) st = 2;                  // DL_STATUS_NONFINITE
}

__device__ __forceinline__ void dl_finalize_point(const double* __restrict__ part_row, int n_tiles, const double* __restrict__ theta_row, int n_params,
                                                  const double* __restrict__ priors, double& ll, double& lp, int& st) {
    dl_finalize_from_chi2(dl_chi2_of_parts(part_row, n_tiles), theta_row, n_params, priors, ll, lp, st);
}
