// Analytic marginalisation / best fit of the n_s linear parameters of ONE point by ONE lane, from the Gram matrix G = X X^T of X = [dt; Tt_1 .. Tt_ns]
// (likelihoods/base.py:129-200, 314-413; in whitened variables dt = L^T Delta, Tt_s = L^T dDelta/dx_s):
//   H_L = -Tt Tt^T, g_L = -Tt dt, H = H_L - diag(prec), g = g_L - (x0 - loc) prec, dx = -H^-1 g,
//   loglike = -1/2 |dt|^2 + 1/2 dx H_L dx + g_L dx - 1/2 logdet(-H[marg, marg]),  logprior += sum -1/2 (x0 + dx - loc)^2 prec.
// Shared by dl_finalize_marg_gram_kernel (dl_kernels.hip: G from global memory, 64 points per wavefront) and by the tail of dl_emulated_feature_gram_kernel
// (dl_emu_batch.h: G of the workgroup's 16 points from LDS).  NS is a compile-time size: the lower triangle lives in registers, nothing is indexed at run time, no
// LDS, no cross-lane traffic -- the solve of a point is a few hundred flops on one dependent chain, and lanes are what a GPU has most of.
#pragma once
#include "dl_kernels.h"
#include "dl_prior.h"
#include <hip/hip_runtime.h>

// 1 / sqrt(d), d > 0 normal: v_rsq_f64 (about 26 bits) and two Newton steps -- a dozen instructions on the pivot chain instead of a square root and a division
__device__ __forceinline__ double dl_rsqrt_pos(double d) {
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double h = 0.5 * y, e = fma(-d * y, h, 0.5);
        y = fma(y, e, y);
    }
    return y;
}

// sum of the log-priors of the sampled parameters of one point (parameter.py:1994-2017), NaN flag of its parameters
__device__ __forceinline__ void dl_marg_priors_lane(const double* __restrict__ theta_row, int n_params, const double* __restrict__ priors, double& lp, int& nan_in) {
    lp = 0.;
    nan_in = 0;
    for (int p0 = 0; p0 < n_params; p0 += 4) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = p0 + u < n_params ? theta_row[p0 + u] : 0.;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (p0 + u < n_params) {
                if (x[u] != x[u]) nan_in = 1;
                lp += dl_prior_logpdf(priors + 5 * (p0 + u), x[u]);
            }
    }
}

// G(i, j), 0 <= j <= i <= NS: entry of the point's Gram matrix.  solved_row [NS] / hessian_blk [NS, NS]: outputs of this point, or null.
// Returns the log-likelihood; lps = log-prior of the solved parameters; ok = false if a pivot is not positive.
// Partially marginalised sets (n_marg < NS): rows / columns of the parameters that are only solved become unit vectors in the second factorisation (the determinant of
// the marginalised block is unchanged) instead of being compacted away.
template <int NS, class Entry>
__device__ __forceinline__ double dl_marg_solve_lane(Entry&& G, const DlMargDev& mg, double* __restrict__ solved_row, double* __restrict__ hessian_blk, double& lps, bool& ok) {
    const double inf = __builtin_huge_val();
    double H[NS][NS], gd[NS];
    const double g00 = G(0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        gd[s] = G(1 + s, 0);
#pragma unroll
        for (int t = 0; t <= s; ++t) H[s][t] = G(1 + s, 1 + t);
    }
    // A = -H = Tt Tt^T + diag(prec) (SPD); rhs = g = -(Tt dt) - (x0 - loc) prec; dx = A^-1 g
    double C[NS][NS], inv[NS], dx[NS];
    ok = true;
    // log det = log of the product of the pivots: mantissas and exponents apart (the product of NS <= 8 mantissas in [1/2, 1) cannot underflow), ONE logarithm
    double mant_all = 1.;
    int exp_all = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        double d = H[j][j] + mg.prec[j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= C[j][k] * C[j][k];
        if (!(d > 0.) || d == inf) { ok = false; d = 1.; }
        mant_all *= __builtin_amdgcn_frexp_mant(d); exp_all += __builtin_amdgcn_frexp_exp(d);
        inv[j] = dl_rsqrt_pos(d);
        C[j][j] = d * inv[j];
#pragma unroll
        for (int i = j + 1; i < NS; ++i) {
            double sum = H[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) sum -= C[i][k] * C[j][k];
            C[i][j] = sum * inv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {   // forward, backward substitution
        double sum = -gd[i] - (mg.x0[i] - mg.loc[i]) * mg.prec[i];
#pragma unroll
        for (int k = 0; k < i; ++k) sum -= C[i][k] * dx[k];
        dx[i] = sum * inv[i];
    }
#pragma unroll
    for (int i = NS - 1; i >= 0; --i) {
        double sum = dx[i];
#pragma unroll
        for (int k = i + 1; k < NS; ++k) sum -= C[k][i] * dx[k];
        dx[i] = sum * inv[i];
    }
    // 1/2 dx H_L dx + g_L dx  (likelihoods/base.py:385-386), H_L = -Tt Tt^T, g_L = -Tt dt
    double quad = 0., lin = 0.;
    lps = 0.;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double rowsum = 0.;
#pragma unroll
        for (int t = 0; t < NS; ++t) rowsum += (s >= t ? H[s][t] : H[t][s]) * dx[t];
        quad += dx[s] * rowsum;
        lin += gd[s] * dx[s];
        const double xs = mg.x0[s] + dx[s];
        lps += -0.5 * (xs - mg.loc[s]) * (xs - mg.loc[s]) * mg.prec[s];   // 363-364 with parameter.py:2007 (0 for flat priors: prec = 0)
        if (solved_row) solved_row[s] = xs;
        if (hessian_blk) {   // likelihood Hessian H_L = -Tt Tt^T w.r.t. the solved parameters (derived output, likelihoods/base.py:388-390)
#pragma unroll
            for (int t = 0; t < NS; ++t) hessian_blk[s * NS + t] = -(s >= t ? H[s][t] : H[t][s]);
        }
    }
    double ll = -0.5 * g00 - 0.5 * quad - lin;
    // -1/2 logdet(-H[marg, marg]) (394-404); all-marg: the factorisation above
    if (mg.n_marg == NS) ll -= 0.5 * (log(mant_all) + (double)exp_all * 0.693147180559945309417);
    else if (mg.n_marg > 0) {
        double S[NS][NS], mant2 = 1.;
        int exp2 = 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const bool mj = mg.is_marg[j] != 0;
            double d = mj ? H[j][j] + mg.prec[j] : 1.;
#pragma unroll
            for (int k = 0; k < j; ++k) d -= S[j][k] * S[j][k];
            if (!(d > 0.) || d == inf) { ok = false; d = 1.; }
            mant2 *= __builtin_amdgcn_frexp_mant(d); exp2 += __builtin_amdgcn_frexp_exp(d);
            const double invj = dl_rsqrt_pos(d);
#pragma unroll
            for (int i = j + 1; i < NS; ++i) {
                double sum = (mj && mg.is_marg[i] != 0) ? H[i][j] : 0.;
#pragma unroll
                for (int k = 0; k < j; ++k) sum -= S[i][k] * S[j][k];
                S[i][j] = sum * invj;
            }
        }
        ll -= 0.5 * (log(mant2) + (double)exp2 * 0.693147180559945309417);
    }
    return ll;
}

// status rules of include/desilike_amd.h; post_mode: log-posterior with the samplers' conventions (samplers/base.py:185-191)
__device__ __forceinline__ void dl_marg_store_lane(double ll, double lps, bool ok, double lp, bool nan_in, int post_mode, int64_t b, double* __restrict__ loglike,
                                                   double* __restrict__ logprior, int32_t* __restrict__ status) {
    const double inf = __builtin_huge_val();
    const double lptot = lp + lps;
    int st = DL_ST_OK;
    if (nan_in) st = DL_ST_NAN_INPUT;
    else if (lp == -inf) st = DL_ST_OUT_OF_PRIOR;
    else if (!ok || !(ll == ll) || ll == inf || ll == -inf) st = DL_ST_NONFINITE;
    if (loglike) loglike[b] = post_mode ? (st == DL_ST_OK ? ll + lptot : -inf) : ll;
    if (logprior) logprior[b] = lptot;
    if (status) status[b] = st;
}
