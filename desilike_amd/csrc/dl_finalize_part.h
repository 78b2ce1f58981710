// chi2 / prior / status of ONE point from the partial chi2 of the column blocks of the chi2 GEMM (dl_chi2_gemm.h): shared by dl_finalize_part_kernel and by the
// ensemble step kernel (dl_ensemble.hip), which finishes the pending half-step's proposals itself instead of waiting for a separate finalize launch.
// Priors: parameter.py:1994-2017 (uniform / norm, maximum removed, closed limits); status rules of include/desilike_amd.h.
#pragma once
#include "dl_prior.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

// Loads are issued in batches and consumed in the original order: a load-wait-add loop over a run-time count is one round trip PER ELEMENT on the dependent
// chain (the first version: 16 + n_params serialized round trips per point, 4.7 us for a kernel that moves 200 bytes per thread).  Lanes beyond the count re-read
// the last element and add nothing: the sums are those of the plain loops, bit for bit.
__device__ __forceinline__ double dl_chi2_of_parts(const double* __restrict__ part_row, int n_tiles) {
    double chi2 = 0.;
    for (int t0 = 0; t0 < n_tiles; t0 += 16) {   // fixed order: deterministic
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = part_row[t0 + t < n_tiles ? t0 + t : n_tiles - 1];
#pragma unroll
        for (int t = 0; t < 16; ++t) chi2 = t0 + t < n_tiles ? chi2 + v[t] : chi2;
    }
    return chi2;
}

// log-likelihood and status of a point from its chi2, its summed log-prior and the NaN flag of its parameters (status rules of include/desilike_amd.h)
__device__ __forceinline__ void dl_finalize_status(double chi2, double lp, bool nan_in, double& ll, int& st) {
    const double inf = __builtin_huge_val();
    ll = -0.5 * chi2;
    st = 0;                                                                   // DL_STATUS_OK
    if (nan_in) st = 3;                                                       // DL_STATUS_NAN_INPUT
    else if (lp == -inf) st = 1;                                              // DL_STATUS_OUT_OF_PRIOR
    else if (!(ll == ll) || ll == inf || ll == -inf) st = 2;                  // DL_STATUS_NONFINITE
}

// eight parameter values of a point from p0 on (beyond n_params: the last one again)
__device__ __forceinline__ void dl_load_theta8(const double* __restrict__ theta_row, int n_params, int p0, double (&x)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = theta_row[p0 + q < n_params ? p0 + q : n_params - 1];
}

// priors: table [n_params, 5] (global memory or LDS); x0: the first eight parameter values (dl_load_theta8 at p0 = 0: the caller issues these loads together with
// its others, ahead of whatever it has to wait for)
template <int NB = 8>   // parameters whose prior rows are read in one batch (registers: 5 NB doubles)
__device__ __forceinline__ void dl_finalize_from_chi2(double chi2, const double (&x0)[8], const double* __restrict__ theta_row, int n_params, const double* __restrict__ priors,
                                                      double& ll, double& lp, int& st) {
    lp = 0.;
    bool nan_in = false;
    for (int p0 = 0; p0 < n_params; p0 += 8) {
        double x[8];
        if (p0 == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = x0[q];
        } else dl_load_theta8(theta_row, n_params, p0, x);
#pragma unroll
        for (int q0 = 0; q0 < 8; q0 += NB) {
            double pr[NB][5];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int p = p0 + q0 + q < n_params ? p0 + q0 + q : n_params - 1;
#pragma unroll
                for (int c = 0; c < 5; ++c) pr[q][c] = priors[5 * p + c];
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                if (p0 + q0 + q < n_params) {
                    if (x[q0 + q] != x[q0 + q]) nan_in = true;
                    lp += dl_prior_logpdf(pr[q], x[q0 + q]);
                }
            }
        }
    }
    dl_finalize_status(chi2, lp, nan_in, ll, st);
}

