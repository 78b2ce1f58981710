// dl_cov.hip -- Gaussian covariance of power-spectrum / correlation-function multipoles on the device (include/desilike_amd.h: dl_cov_*).
//
// Reference: desilike/observables/galaxy_clustering/covariance.py:355-456 (ObservablesCovarianceMatrix._run).  Every matrix element is
//
//     C[r, c] = front / den * sum_q (sigma(k_q) w_q) w2_q + const,     sigma(k) = prefactor * sum_{la, lb} (P1_la(k) P2_lb(k) - zero lag) I(la, lb, l1, l2),
//
// with P_l(k) the theory multipoles (+ shot noise on the monopole) interpolated linearly in k like np.interp (covariance.py:361-371), I the integral of the product
// of four Legendre polynomials, and -- per kind of block -- P x P: q = integration points of the intersected k-bin, w = k^2, den = sum k^2, front = (2 pi)^3 V(bin) /
// (V(bin1) V(bin2)) (397-406); xi x P: w = k^2, w2 = s-bin average of j_l(s k), front = i^l (408-416); xi x xi: q = the theories' common k grid, w = shell volume,
// w2 = product of the two s-bin averages, front = i^(l1 + l2) / (2 pi)^3, const = the shot-noise term of overlapping bins (423-446).  The lists of cells and points
// depend on the binning only: the host builds them once (desilike_amd/observables/galaxy_clustering/covariance.py); what changes from one parameter point to the next
// is the theory power, which stays on the device (dl_eval_theory).  One thread per (cell, parameter point); diagonal blocks are then symmetrised like the reference
// does, (C + C^T) / 2 (covariance.py:349-351).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/desilike_amd.h"
#include "dl_kernels.h"

#define DL_COV_MAX_THEORIES 8
#define DL_COV_MAX_ELLS 5

struct DlCovTheories {
    const double* power[DL_COV_MAX_THEORIES];     // [B, n_ell, n_k]
    int32_t n_ell[DL_COV_MAX_THEORIES], n_k[DL_COV_MAX_THEORIES], ell0[DL_COV_MAX_THEORIES];   // ell0: index of the monopole in the theory's multipoles, or -1
    double shotnoise[DL_COV_MAX_THEORIES];
};

struct dl_cov {
    int device = 0, n = 0, n_theories = 0;
    int64_t n_cells = 0, n_points = 0, n_sym = 0;
    int32_t n_ell[DL_COV_MAX_THEORIES], n_k[DL_COV_MAX_THEORIES], ell0[DL_COV_MAX_THEORIES];
    double shotnoise[DL_COV_MAX_THEORIES];
    int32_t *cell_i = nullptr, *pt_i = nullptr, *sym = nullptr;
    double *cell_d = nullptr, *pt_d = nullptr, *gtab = nullptr;
};

namespace {
int cov_fail(const std::string& msg) { dl_set_last_error(msg.c_str()); return 1; }
}

// cell_i [n_cells, 8]: row, col, theory 1, theory 2, index of the Legendre table, zero-lag flag, first point, number of points
// cell_d [n_cells, 4]: prefactor, front, den, const;  pt_i [n_points, 2]: interval of k_q in the k grid of theory 1 / 2;  pt_d [n_points, 6]: (k_q - k_j, k_j+1 - k_j) for
// theory 1, for theory 2, w, w2;  gtab [n_gtab, 5, 5]
__global__ __launch_bounds__(256) void dl_cov_kernel(DlCovTheories th, const int32_t* __restrict__ cell_i, const double* __restrict__ cell_d, const int32_t* __restrict__ pt_i,
                                                     const double* __restrict__ pt_d, const double* __restrict__ gtab, int64_t n_cells, int n, double* __restrict__ cov) {
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= n_cells) return;
    const int64_t b = blockIdx.y;
    const int32_t* ci = cell_i + cell * 8;
    const double* cd = cell_d + cell * 4;
    const int row = ci[0], col = ci[1], t1 = ci[2], t2 = ci[3];
    const bool zero_lag = ci[5] != 0;
    const double* G = gtab + (size_t)ci[4] * (DL_COV_MAX_ELLS * DL_COV_MAX_ELLS);
    const int n1 = th.n_ell[t1], n2 = th.n_ell[t2], nk1 = th.n_k[t1], nk2 = th.n_k[t2], e1 = th.ell0[t1], e2 = th.ell0[t2];
    const double sn1 = th.shotnoise[t1], sn2 = th.shotnoise[t2];
    const double* P1 = th.power[t1] + (size_t)b * n1 * nk1;
    const double* P2 = th.power[t2] + (size_t)b * n2 * nk2;
    double g[DL_COV_MAX_ELLS * DL_COV_MAX_ELLS];
#pragma unroll
    for (int e = 0; e < DL_COV_MAX_ELLS * DL_COV_MAX_ELLS; ++e) g[e] = G[e];
    double sum = 0.;
    for (int q = ci[6]; q < ci[6] + ci[7]; ++q) {
        const int j1 = pt_i[2 * (size_t)q], j2 = pt_i[2 * (size_t)q + 1];
        const double* pd = pt_d + 6 * (size_t)q;
        double p2[DL_COV_MAX_ELLS];
#pragma unroll
        for (int lb = 0; lb < DL_COV_MAX_ELLS; ++lb) {
            p2[lb] = 0.;
            if (lb < n2) {
                const double f0 = P2[(size_t)lb * nk2 + j2] + (lb == e2 ? sn2 : 0.), f1 = P2[(size_t)lb * nk2 + j2 + 1] + (lb == e2 ? sn2 : 0.);
                p2[lb] = ((f1 - f0) / pd[3]) * pd[2] + f0;           // np.interp: slope (x - x_j) + f_j
            }
        }
        double sigma = 0.;
#pragma unroll
        for (int la = 0; la < DL_COV_MAX_ELLS; ++la) {
            if (la < n1) {
                const double f0 = P1[(size_t)la * nk1 + j1] + (la == e1 ? sn1 : 0.), f1 = P1[(size_t)la * nk1 + j1 + 1] + (la == e1 ? sn1 : 0.);
                const double p1 = ((f1 - f0) / pd[1]) * pd[0] + f0;
#pragma unroll
                for (int lb = 0; lb < DL_COV_MAX_ELLS; ++lb)
                    if (lb < n2) sigma += (p1 * p2[lb] - ((zero_lag && la == e1 && lb == e2) ? sn1 * sn2 : 0.)) * g[la * DL_COV_MAX_ELLS + lb];   // covariance.py:378-381
            }
        }
        sigma = cd[0] * sigma;
        sum += (sigma * pd[4]) * pd[5];
    }
    cov[((size_t)b * n + row) * n + col] = cd[1] * sum / cd[2] + cd[3];
}

// (C + C^T) / 2 on the listed pairs (r, c), r < c, of the diagonal blocks
__global__ __launch_bounds__(256) void dl_cov_sym_kernel(const int32_t* __restrict__ sym, int64_t n_sym, int n, double* __restrict__ cov) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_sym) return;
    double* c = cov + (size_t)blockIdx.y * n * n;
    const int r = sym[2 * e], col = sym[2 * e + 1];
    const double v = (c[(size_t)r * n + col] + c[(size_t)col * n + r]) / 2.;
    c[(size_t)r * n + col] = v; c[(size_t)col * n + r] = v;
}

extern "C" {

void dl_cov_destroy(dl_cov* plan) {
    if (!plan) return;
    (void)hipSetDevice(plan->device);
    for (void* p : {(void*)plan->cell_i, (void*)plan->pt_i, (void*)plan->sym, (void*)plan->cell_d, (void*)plan->pt_d, (void*)plan->gtab}) if (p) (void)hipFree(p);
    delete plan;
}

int dl_cov_create(dl_cov** out, int device, int32_t n, int32_t n_theories, const int32_t* n_ell, const int32_t* n_k, const int32_t* ell0, const double* shotnoise,
                  int64_t n_cells, const int32_t* cell_i, const double* cell_d, int64_t n_points, const int32_t* pt_i, const double* pt_d, int32_t n_gtab, const double* gtab,
                  int64_t n_sym, const int32_t* sym) {
    if (!out) return cov_fail("dl_cov_create: null argument");
    *out = nullptr;
    if (n < 1 || n_theories < 1 || n_theories > DL_COV_MAX_THEORIES || n_cells < 0 || n_points < 0 || n_gtab < 1 || !n_ell || !n_k || !ell0 || !shotnoise || !gtab)
        return cov_fail("dl_cov_create: invalid argument");
    for (int t = 0; t < n_theories; ++t)
        if (n_ell[t] < 1 || n_ell[t] > DL_COV_MAX_ELLS || n_k[t] < 2) return cov_fail("dl_cov_create: a theory needs 1 to 5 multipoles and at least 2 wavenumbers");
    for (int64_t c = 0; c < n_cells; ++c) {
        const int32_t* ci = cell_i + c * 8;
        if (ci[0] < 0 || ci[0] >= n || ci[1] < 0 || ci[1] >= n || ci[2] < 0 || ci[2] >= n_theories || ci[3] < 0 || ci[3] >= n_theories || ci[4] < 0 || ci[4] >= n_gtab ||
            ci[6] < 0 || ci[7] < 0 || (int64_t)ci[6] + ci[7] > n_points)
            return cov_fail("dl_cov_create: cell " + std::to_string(c) + " out of range");
        for (int q = ci[6]; q < ci[6] + ci[7]; ++q)
            if (pt_i[2 * (size_t)q] < 0 || pt_i[2 * (size_t)q] > n_k[ci[2]] - 2 || pt_i[2 * (size_t)q + 1] < 0 || pt_i[2 * (size_t)q + 1] > n_k[ci[3]] - 2)
                return cov_fail("dl_cov_create: interpolation interval out of range in cell " + std::to_string(c));
    }
    for (int64_t e = 0; e < n_sym; ++e) if (sym[2 * e] < 0 || sym[2 * e] >= n || sym[2 * e + 1] < 0 || sym[2 * e + 1] >= n) return cov_fail("dl_cov_create: symmetrisation pair out of range");
    dl_cov* plan = new dl_cov();
    plan->device = device; plan->n = n; plan->n_theories = n_theories; plan->n_cells = n_cells; plan->n_points = n_points; plan->n_sym = n_sym;
    for (int t = 0; t < n_theories; ++t) { plan->n_ell[t] = n_ell[t]; plan->n_k[t] = n_k[t]; plan->ell0[t] = ell0[t]; plan->shotnoise[t] = shotnoise[t]; }
    auto upload = [&](void** dst, const void* src, size_t bytes) {
        if (bytes == 0) bytes = 8;
        if (hipMalloc(dst, bytes) != hipSuccess) return false;
        return src == nullptr || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (hipSetDevice(device) != hipSuccess || !upload((void**)&plan->cell_i, cell_i, (size_t)n_cells * 8 * sizeof(int32_t)) || !upload((void**)&plan->cell_d, cell_d, (size_t)n_cells * 4 * sizeof(double)) ||
        !upload((void**)&plan->pt_i, pt_i, (size_t)n_points * 2 * sizeof(int32_t)) || !upload((void**)&plan->pt_d, pt_d, (size_t)n_points * 6 * sizeof(double)) ||
        !upload((void**)&plan->gtab, gtab, (size_t)n_gtab * DL_COV_MAX_ELLS * DL_COV_MAX_ELLS * sizeof(double)) || !upload((void**)&plan->sym, sym, (size_t)n_sym * 2 * sizeof(int32_t))) {
        dl_cov_destroy(plan);
        return cov_fail("dl_cov_create: device allocation / upload failed");
    }
    *out = plan;
    return 0;
}

int dl_cov_apply(dl_cov* plan, const double* const* power_dev, int64_t B, double* cov_dev, void* hip_stream) {
    if (!plan || !power_dev || !cov_dev || B < 0) return cov_fail("dl_cov_apply: invalid argument");
    if (B == 0) return 0;
    if (B > 65535) return cov_fail("dl_cov_apply: at most 65535 parameter points per call");
    hipStream_t stream = (hipStream_t)hip_stream;
    if (hipSetDevice(plan->device) != hipSuccess) return cov_fail("dl_cov_apply: hipSetDevice failed");
    DlCovTheories th;
    for (int t = 0; t < DL_COV_MAX_THEORIES; ++t) {
        const bool live = t < plan->n_theories;
        if (live && power_dev[t] == nullptr) return cov_fail("dl_cov_apply: null power array");
        th.power[t] = live ? power_dev[t] : nullptr; th.n_ell[t] = live ? plan->n_ell[t] : 0; th.n_k[t] = live ? plan->n_k[t] : 0; th.ell0[t] = live ? plan->ell0[t] : -1;
        th.shotnoise[t] = live ? plan->shotnoise[t] : 0.;
    }
    if (hipMemsetAsync(cov_dev, 0, (size_t)B * plan->n * plan->n * sizeof(double), stream) != hipSuccess) return cov_fail("dl_cov_apply: hipMemsetAsync failed");
    if (plan->n_cells > 0)
        hipLaunchKernelGGL(dl_cov_kernel, dim3((unsigned)((plan->n_cells + 255) / 256), (unsigned)B), dim3(256), 0, stream, th, plan->cell_i, plan->cell_d, plan->pt_i, plan->pt_d, plan->gtab,
                           plan->n_cells, plan->n, cov_dev);
    if (plan->n_sym > 0)
        hipLaunchKernelGGL(dl_cov_sym_kernel, dim3((unsigned)((plan->n_sym + 255) / 256), (unsigned)B), dim3(256), 0, stream, plan->sym, plan->n_sym, plan->n, cov_dev);
    if (hipGetLastError() != hipSuccess) return cov_fail("dl_cov_apply: launch failed");
    return 0;
}

}  // extern "C"
