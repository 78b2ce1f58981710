// dl_mlp.hip -- training of the MLP emulator on the GPU (include/desilike_amd.h: dl_mlp_*).
//
// In the reference the MLP emulator engine is third-party (``cosmoprimo.emulators.tools.MLPEmulatorEngine``, wrapped by desilike/emulators/__init__.py:510-533;
// structure documented by emulators/conversion.py:20-96: dense layers ``v @ kernel + bias``, silu / relu / tanh between them, min-max scalers outside) and trains on
// the CPU / through jax.  Here: fp64 mini-batch training with Adam (Kingma & Ba 2015) on mean-squared error of the (scaled) outputs, every step of it on the device:
//     forward   z_l = a_{l-1} W_l + b_l,  a_l = act(z_l)  (last layer linear)           -- fp64 MFMA GEMM, bias + activation in the epilogue
//     loss      L = mean (a_L - y)^2,  delta_L = 2 (a_L - y) / (B n_out)
//     backward  dW_l = a_{l-1}^T delta_l,  db_l = colsum(delta_l),  delta_{l-1} = (delta_l W_l^T) * act'(z_{l-1})   -- the same GEMM kernel with transposed operands
//     update    Adam with bias correction
// One generic GEMM kernel C = op(A) op(B) (v_mfma_f64_16x16x4_f64, one wavefront per 16 x 16 tile, operands addressed through (row, column) strides so that the
// transposes cost nothing); sizes are emulator-sized (hidden width 64, up to ~10^4 outputs, batches of ~10^3): the last layer's three products dominate.
// Training is not on the per-evaluation hot path: the kernel is written for correctness and determinism (fixed summation orders: two runs give the same weights,
// and the NumPy oracle's Adam reproduces them to rounding: tests/test_gpu_mlp_train.py), not tuned further.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/desilike_amd.h"
#include "dl_kernels.h"

namespace {

int fail(const std::string& msg) {
    dl_set_last_error(msg.c_str());
    return 1;
}

#define DL_MLP_HIP(call)                                                                              \
    do {                                                                                              \
        hipError_t err__ = (call);                                                                    \
        if (err__ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(err__));   \
    } while (0)

typedef double dl_ml_double4 __attribute__((ext_vector_type(4)));

enum { DL_ACT_SILU = 0, DL_ACT_RELU = 1, DL_ACT_TANH = 2, DL_ACT_NONE = 3 };

__device__ __forceinline__ double dl_act(double z, int act) {
    switch (act) {
        case DL_ACT_SILU: return z / (1. + exp(-z));          // conversion.py:27-28
        case DL_ACT_RELU: return z > 0. ? z : 0.;
        case DL_ACT_TANH: return tanh(z);
        default: return z;
    }
}

__device__ __forceinline__ double dl_act_prime(double z, int act) {
    switch (act) {
        case DL_ACT_SILU: { const double s = 1. / (1. + exp(-z)); return s * (1. + z * (1. - s)); }
        case DL_ACT_RELU: return z > 0. ? 1. : 0.;
        case DL_ACT_TANH: { const double t = tanh(z); return 1. - t * t; }
        default: return 1.;
    }
}

// C[M, N] (row-major, ldc) = sum_k A(i, k) B(k, j), A(i, k) = A[i sa_r + k sa_c], B(k, j) = B[k sb_r + j sb_c]; epilogue:
//   mode 0: C = acc (+ bias[j]), and if act_out != nullptr: act_out = act(C)                       (forward)
//   mode 1: C = acc * act'(zprev[i, j])                                                            (delta of the previous layer)
// One wavefront per 16 x 16 tile (4 tiles per workgroup); MFMA operand layout: A lane l = A(row l & 15, k + (l >> 4)), B lane l = B(k + (l >> 4), col l & 15),
// C register r of lane l = C[row (l >> 4) + 4 r][col l & 15].
__global__ __launch_bounds__(256) void dl_mlp_gemm_kernel(const double* __restrict__ A, int64_t sa_r, int64_t sa_c, const double* __restrict__ B, int64_t sb_r, int64_t sb_c,
                                                          double* __restrict__ C, int64_t ldc, int M, int N, int K, const double* __restrict__ bias, int mode, int act,
                                                          double* __restrict__ act_out, const double* __restrict__ zprev, int64_t ldz) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_n = (N + 15) / 16;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t tiles = (int64_t)((M + 15) / 16) * tiles_n;
    if (tile >= tiles) return;
    const int i0 = (int)(tile / tiles_n) * 16, j0 = (int)(tile % tiles_n) * 16;
    const int r = lane & 15, g = lane >> 4;
    const bool arow_ok = i0 + r < M, bcol_ok = j0 + r < N;
    const double* ap = A + (int64_t)(i0 + r) * sa_r;
    const double* bp = B + (int64_t)(j0 + r) * sb_c;
    dl_ml_double4 acc = {0., 0., 0., 0.};
    int k = 0;
    for (; k + 16 <= K; k += 16) {   // four k-steps with their loads issued together
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k + 4 * u + g;
            a[u] = arow_ok ? ap[(int64_t)kk * sa_c] : 0.;
            b[u] = bcol_ok ? bp[(int64_t)kk * sb_r] : 0.;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    for (; k < K; k += 4) {
        const int kk = k + g;
        const double a = (arow_ok && kk < K) ? ap[(int64_t)kk * sa_c] : 0.;
        const double b = (bcol_ok && kk < K) ? bp[(int64_t)kk * sb_r] : 0.;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    const int col = j0 + r;
    if (col >= N) return;
    const double bj = (mode == 0 && bias != nullptr) ? bias[col] : 0.;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = i0 + g + 4 * q;
        if (row >= M) continue;
        double v = acc[q];
        if (mode == 0) {
            v += bj;
            C[(int64_t)row * ldc + col] = v;
            if (act_out != nullptr) act_out[(int64_t)row * ldc + col] = dl_act(v, act);
        } else {
            C[(int64_t)row * ldc + col] = v * dl_act_prime(zprev[(int64_t)row * ldz + col], act);
        }
    }
}

void gemm(hipStream_t stream, const double* A, int64_t sa_r, int64_t sa_c, const double* B, int64_t sb_r, int64_t sb_c, double* C, int64_t ldc, int M, int N, int K,
          const double* bias, int mode, int act, double* act_out, const double* zprev, int64_t ldz) {
    const int64_t tiles = (int64_t)((M + 15) / 16) * ((N + 15) / 16);
    hipLaunchKernelGGL(dl_mlp_gemm_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, A, sa_r, sa_c, B, sb_r, sb_c, C, ldc, M, N, K, bias, mode, act, act_out, zprev, ldz);
}

// delta_L = 2 (out - y) / (B n_out); partial sums of squared errors per workgroup (summed in a fixed order by the caller's reduction kernel)
__global__ __launch_bounds__(256) void dl_mlp_loss_kernel(const double* __restrict__ out, const double* __restrict__ y, double* __restrict__ delta, int64_t n, double scale,
                                                          double* __restrict__ partial) {
    __shared__ double red[256];
    double sum = 0.;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double d = out[i] - y[i];
        delta[i] = scale * d;
        sum += d * d;
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void dl_mlp_loss_reduce_kernel(const double* __restrict__ partial, int n, double norm, double* __restrict__ loss) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double sum = 0.;
        for (int i = 0; i < n; ++i) sum += partial[i];
        *loss = sum * norm;
    }
}

// db[j] = sum_i delta[i, j] (fixed order)
__global__ void dl_mlp_colsum_kernel(const double* __restrict__ delta, int64_t ld, int M, int N, double* __restrict__ db) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    double sum = 0.;
    for (int i = 0; i < M; ++i) sum += delta[(int64_t)i * ld + j];
    db[j] = sum;
}

__global__ void dl_mlp_adam_kernel(double* __restrict__ w, const double* __restrict__ grad, double* __restrict__ m, double* __restrict__ v, int64_t n, double lr, double beta1,
                                   double beta2, double eps, double c1, double c2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double gi = grad[i];
    const double mi = beta1 * m[i] + (1. - beta1) * gi;
    const double vi = beta2 * v[i] + (1. - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    w[i] -= lr * (mi / c1) / (sqrt(vi / c2) + eps);
}

}  // namespace

struct dl_mlp {
    int device = 0, n_layers = 0, act = 0;
    std::vector<int> widths;          // [n_layers + 1]
    std::vector<int64_t> w_off, b_off;   // offsets of kernel [in, out] / bias [out] of each layer in the flat parameter vector
    int64_t n_weights = 0;
    long long step = 0;               // Adam time step
    double *w = nullptr, *grad = nullptr, *m = nullptr, *v = nullptr;   // [n_weights]
    // per-batch workspaces
    int64_t cap = 0;
    std::vector<double*> z, a, delta;   // z[l], a[l] [cap, widths[l + 1]]; delta[l] same shape
    double *partial = nullptr, *loss = nullptr;
};

static void dl_mlp_free_ws(dl_mlp* net) {
    for (auto* vec : {&net->z, &net->a, &net->delta})
        for (double*& p : *vec) if (p) { (void)hipFree(p); p = nullptr; }
    net->cap = 0;
}

static int dl_mlp_reserve(dl_mlp* net, int64_t rows) {
    if (rows <= net->cap) return 0;
    dl_mlp_free_ws(net);
    for (int l = 0; l < net->n_layers; ++l) {
        const size_t bytes = (size_t)rows * net->widths[l + 1] * sizeof(double);
        DL_MLP_HIP(hipMalloc((void**)&net->z[l], bytes));
        DL_MLP_HIP(hipMalloc((void**)&net->a[l], bytes));
        DL_MLP_HIP(hipMalloc((void**)&net->delta[l], bytes));
    }
    net->cap = rows;
    return 0;
}

// forward pass of ``rows`` samples; the last layer's output is z[L - 1] (linear)
static void dl_mlp_forward_ws(dl_mlp* net, const double* x, int64_t rows, hipStream_t stream) {
    const double* prev = x;
    for (int l = 0; l < net->n_layers; ++l) {
        const int in = net->widths[l], out = net->widths[l + 1];
        const bool last = l == net->n_layers - 1;
        gemm(stream, prev, in, 1, net->w + net->w_off[l], out, 1, net->z[l], out, (int)rows, out, in, net->w + net->b_off[l], 0, last ? DL_ACT_NONE : net->act,
             last ? nullptr : net->a[l], nullptr, 0);
        prev = net->a[l];
    }
}

extern "C" {

void dl_mlp_destroy(dl_mlp* net) {
    if (!net) return;
    (void)hipSetDevice(net->device);
    dl_mlp_free_ws(net);
    for (double* p : {net->w, net->grad, net->m, net->v, net->partial, net->loss}) if (p) (void)hipFree(p);
    delete net;
}

int dl_mlp_create(dl_mlp** out, int device, int32_t n_layers, const int32_t* widths, int32_t activation, const double* weights) {
    if (!out || !widths || !weights) return fail("dl_mlp_create: null argument");
    *out = nullptr;
    if (n_layers < 1 || n_layers > 16 || activation < 0 || activation > 2) return fail("dl_mlp_create: 1 to 16 layers, activation 0 (silu) / 1 (relu) / 2 (tanh)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("dl_mlp_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail("dl_mlp_create: device ordinal out of range");
    DL_MLP_HIP(hipSetDevice(device));
    dl_mlp* net = new dl_mlp();
    net->device = device; net->n_layers = n_layers; net->act = activation;
    net->widths.assign(widths, widths + n_layers + 1);
    for (int wdt : net->widths) if (wdt < 1) { delete net; return fail("dl_mlp_create: layer widths must be positive"); }
    int64_t off = 0;
    for (int l = 0; l < n_layers; ++l) {
        net->w_off.push_back(off); off += (int64_t)widths[l] * widths[l + 1];
        net->b_off.push_back(off); off += widths[l + 1];
    }
    net->n_weights = off;
    net->z.assign(n_layers, nullptr); net->a.assign(n_layers, nullptr); net->delta.assign(n_layers, nullptr);
    const size_t bytes = (size_t)off * sizeof(double);
    auto bail = [&](const std::string& msg) { dl_mlp_destroy(net); return fail(msg); };
    if (hipMalloc((void**)&net->w, bytes) != hipSuccess || hipMalloc((void**)&net->grad, bytes) != hipSuccess || hipMalloc((void**)&net->m, bytes) != hipSuccess ||
        hipMalloc((void**)&net->v, bytes) != hipSuccess || hipMalloc((void**)&net->partial, 1024 * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&net->loss, sizeof(double)) != hipSuccess)
        return bail("dl_mlp_create: device allocation failed");
    if (hipMemcpy(net->w, weights, bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemset(net->m, 0, bytes) != hipSuccess || hipMemset(net->v, 0, bytes) != hipSuccess)
        return bail("dl_mlp_create: upload failed");
    if (hipDeviceSynchronize() != hipSuccess) return bail("dl_mlp_create: hipDeviceSynchronize failed");   // (null-stream memsets vs the caller's non-blocking streams)
    *out = net;
    return 0;
}

int64_t dl_mlp_info(const dl_mlp* net, const char* key) {
    if (!net || !key) return -1;
    std::string k(key);
    if (k == "n_weights") return net->n_weights;
    if (k == "n_layers") return net->n_layers;
    if (k == "step") return net->step;
    if (k == "n_in") return net->widths.front();
    if (k == "n_out") return net->widths.back();
    return -1;
}

int dl_mlp_get_weights(dl_mlp* net, double* weights, void* hip_stream) {
    if (!net || !weights) return fail("dl_mlp_get_weights: null argument");
    DL_MLP_HIP(hipSetDevice(net->device));
    DL_MLP_HIP(hipMemcpyAsync(weights, net->w, (size_t)net->n_weights * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    DL_MLP_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    return 0;
}

int dl_mlp_forward(dl_mlp* net, const double* x_dev, int64_t rows, double* y_dev, void* hip_stream) {
    if (!net || (rows > 0 && (!x_dev || !y_dev)) || rows < 0) return fail("dl_mlp_forward: invalid argument");
    if (rows == 0) return 0;
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_MLP_HIP(hipSetDevice(net->device));
    if (dl_mlp_reserve(net, rows)) return 1;
    dl_mlp_forward_ws(net, x_dev, rows, stream);
    DL_MLP_HIP(hipMemcpyAsync(y_dev, net->z[net->n_layers - 1], (size_t)rows * net->widths.back() * sizeof(double), hipMemcpyDeviceToDevice, stream));
    DL_MLP_HIP(hipGetLastError());
    return 0;
}

// loss and gradient of one batch (no update): grad_dev [n_weights] (may be null: the internal buffer only), loss_dev [1] (may be null)
static int dl_mlp_backprop(dl_mlp* net, const double* x, const double* y, int64_t rows, hipStream_t stream) {
    const int L = net->n_layers, n_out = net->widths.back();
    dl_mlp_forward_ws(net, x, rows, stream);
    const int64_t n = rows * n_out;
    const int nblocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(dl_mlp_loss_kernel, dim3(nblocks), dim3(256), 0, stream, net->z[L - 1], y, net->delta[L - 1], n, 2. / (double)n, net->partial);
    hipLaunchKernelGGL(dl_mlp_loss_reduce_kernel, dim3(1), dim3(64), 0, stream, net->partial, nblocks, 1. / (double)n, net->loss);
    for (int l = L - 1; l >= 0; --l) {
        const int in = net->widths[l], out = net->widths[l + 1];
        const double* aprev = l > 0 ? net->a[l - 1] : x;
        // dW_l [in, out] = a_{l-1}^T [in, rows] . delta_l [rows, out]
        gemm(stream, aprev, 1, in, net->delta[l], out, 1, net->grad + net->w_off[l], out, in, out, (int)rows, nullptr, 0, DL_ACT_NONE, nullptr, nullptr, 0);
        hipLaunchKernelGGL(dl_mlp_colsum_kernel, dim3((unsigned)((out + 127) / 128)), dim3(128), 0, stream, net->delta[l], (int64_t)out, (int)rows, out, net->grad + net->b_off[l]);
        if (l > 0)   // delta_{l-1} [rows, in] = (delta_l [rows, out] . W_l^T [out, in]) * act'(z_{l-1})
            gemm(stream, net->delta[l], out, 1, net->w + net->w_off[l], 1, out, net->delta[l - 1], in, (int)rows, in, out, nullptr, 1, net->act, nullptr, net->z[l - 1], in);
    }
    return 0;
}

int dl_mlp_loss_and_grad(dl_mlp* net, const double* x_dev, const double* y_dev, int64_t rows, double* loss_host, double* grad_host, void* hip_stream) {
    if (!net || !x_dev || !y_dev || rows < 1) return fail("dl_mlp_loss_and_grad: invalid argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_MLP_HIP(hipSetDevice(net->device));
    if (dl_mlp_reserve(net, rows)) return 1;
    if (dl_mlp_backprop(net, x_dev, y_dev, rows, stream)) return 1;
    if (loss_host) DL_MLP_HIP(hipMemcpyAsync(loss_host, net->loss, sizeof(double), hipMemcpyDeviceToHost, stream));
    if (grad_host) DL_MLP_HIP(hipMemcpyAsync(grad_host, net->grad, (size_t)net->n_weights * sizeof(double), hipMemcpyDeviceToHost, stream));
    DL_MLP_HIP(hipStreamSynchronize(stream));
    DL_MLP_HIP(hipGetLastError());
    return 0;
}

int dl_mlp_train(dl_mlp* net, const double* x_dev, const double* y_dev, int64_t n_samples, int64_t batch, int64_t n_steps, double lr, double beta1, double beta2, double eps,
                 double* loss_host, void* hip_stream) {
    if (!net || !x_dev || !y_dev) return fail("dl_mlp_train: null argument");
    if (n_samples < 1 || batch < 1 || batch > n_samples || n_steps < 0) return fail("dl_mlp_train: need 1 <= batch <= n_samples and n_steps >= 0");
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_MLP_HIP(hipSetDevice(net->device));
    if (dl_mlp_reserve(net, batch)) return 1;
    const int n_in = net->widths.front(), n_out = net->widths.back();
    const int64_t chunks = n_samples / batch;      // batches are consecutive chunks of the sample set (the caller shuffles it); a remainder is not used
    double* loss_dev = nullptr;
    if (loss_host && n_steps > 0) DL_MLP_HIP(hipMalloc((void**)&loss_dev, (size_t)n_steps * sizeof(double)));
    for (int64_t it = 0; it < n_steps; ++it) {
        const int64_t c = (net->step % chunks);
        if (dl_mlp_backprop(net, x_dev + (size_t)c * batch * n_in, y_dev + (size_t)c * batch * n_out, batch, stream)) { if (loss_dev) (void)hipFree(loss_dev); return 1; }
        if (loss_dev) (void)hipMemcpyAsync(loss_dev + it, net->loss, sizeof(double), hipMemcpyDeviceToDevice, stream);
        net->step += 1;
        const double c1 = 1. - std::pow(beta1, (double)net->step), c2 = 1. - std::pow(beta2, (double)net->step);
        hipLaunchKernelGGL(dl_mlp_adam_kernel, dim3((unsigned)((net->n_weights + 255) / 256)), dim3(256), 0, stream, net->w, net->grad, net->m, net->v, net->n_weights, lr, beta1, beta2,
                           eps, c1, c2);
    }
    if (loss_dev) {
        hipError_t e = hipMemcpyAsync(loss_host, loss_dev, (size_t)n_steps * sizeof(double), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(loss_dev);
        if (e != hipSuccess) return fail(std::string("dl_mlp_train: ") + hipGetErrorString(e));
    }
    DL_MLP_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
