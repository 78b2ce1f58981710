// dl_fftlog.hip -- batched FFTLog Hankel transform on gfx950.
//
// Replaces the reference's third-party ``cosmoprimo.PowerToCorrelation(k, ell, q=0, lowring=True)`` (call sites
// theories/galaxy_clustering/base.py:76-77, 135): zero-pad a = fun * k^{3/2} to npad points, A = irfft(rfft(a) * u_ell)[::-1],
// out = prefactor_ell * s^{-3/2} * A.  The Mellin coefficients u_ell (loggamma of a complex argument, low-ringing offset) and the
// pre / post factors are constants of the grid, computed once on the host (desilike_amd/fftlog.py) and uploaded at dl_fftlog_create.
//
// One 128-thread workgroup per (point, multipole).  The padded REAL sequence is packed into N2 = npad / 2 complex numbers that live in LDS for the whole
// transform (32 KB at npad = 4096: four workgroups per CU); the twiddle table W_{npad}^m (N2 entries) stays in global memory (L1-resident, a few values per pass).
//   forward : decimation in frequency, FOUR radix-2 stages fused per pass on 16 elements held in registers (11 stages = 4 + 4 + 3: three LDS round trips),
//             natural order in -> bit-reversed order out; one table twiddle per stage and thread, the others by compile-time unit roots;
//   spectrum: real-FFT unpacking X[m] = Fe[m] + W^m Fo[m], multiplication by u[m] and re-packing for the inverse, on the pair
//             (m, N2 - m) addressed through __brev (the pair is a closed set: no barrier inside);
//   inverse : decimation in time with conjugate twiddles, 3 + 4 + 4 fused stages, bit-reversed in -> natural out (no permutation pass at all);
//   output  : reversed, scaled, coalesced store of the n un-padded points.
// Algorithmic work: 2 x 2.5 N2 log2 N2 complex-FFT FLOP + ~40 N2 (spectrum step) per (point, multipole) = 0.53 MFLOP at npad = 4096;
// bound: LDS bandwidth (6 fused passes + the spectrum step, each moving 2 x 16 B x N2: 0.45 MB per transform).
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/desilike_amd.h"
#include "dl_kernels.h"

typedef double dl_ff_c __attribute__((ext_vector_type(2)));   // (re, im)

__device__ __forceinline__ dl_ff_c dl_ff_mul(dl_ff_c a, dl_ff_c b) { return dl_ff_c{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ dl_ff_c dl_ff_mulc(dl_ff_c a, dl_ff_c b) { return dl_ff_c{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }   // a * conj(b)
__device__ __forceinline__ dl_ff_c dl_ff_conj(dl_ff_c a) { return dl_ff_c{a.x, -a.y}; }

#define DL_FF_THREADS 128
// LDS position of complex element i: one slot of padding every 16 elements (= 64 banks), so that the power-of-two strides of the fused passes (lanes 8 or 128
// elements apart) spread over all banks instead of piling onto a few (the un-skewed layout left the two small-stride passes 8- to 32-way conflicted)
#define DL_FF_P(i) ((i) + ((i) >> 4))

// exp(-i pi r / half), r < half <= 8: the twiddle of butterfly r of a fused stage relative to butterfly 0 (compile-time constants after unrolling)
__device__ __forceinline__ dl_ff_c dl_ff_unit(int r, int half) {
    // angle index on the 16-point half circle: q = r * (8 / half), exp(-i pi q / 8)
    const double c8[9] = {1., 0.92387953251128674, 0.70710678118654752, 0.38268343236508977, 0., -0.38268343236508977, -0.70710678118654752, -0.92387953251128674, -1.};
    const double s8[9] = {0., 0.38268343236508977, 0.70710678118654752, 0.92387953251128674, 1., 0.92387953251128674, 0.70710678118654752, 0.38268343236508977, 0.};
    const int q = r * (8 / half);
    return dl_ff_c{c8[q], -s8[q]};
}

// S fused radix-2 stages on the 2^S elements x[i0 + e hl] held in registers (one LDS round trip per S stages).
//   DIF (forward, natural -> bit-reversed order): stage half-sizes hl 2^(S-1), ..., hl;  DIT (inverse, conjugate twiddles): hl, ..., hl 2^(S-1).
// The twiddle of the butterfly (e, e + half) of a stage of half-size hs = hl half is W_{2 hs}^(j + r hl), r = e mod half: one table value tw[j N2 / hs]
// (exp(-i pi m / N2), from global memory: L1-resident, requested together with the LDS loads) times the constant exp(-i pi r / half).
template <int S, bool INVERSE>
__device__ __forceinline__ void dl_ff_pass(dl_ff_c* x, const dl_ff_c* __restrict__ tw, int N2, int hl, int tid) {
    constexpr int R = 1 << S;
    for (int t = tid; t < (N2 >> S); t += DL_FF_THREADS) {
        const int g = t / hl, j = t - g * hl, i0 = g * (hl << S) + j;
        dl_ff_c v[R], base[S];
#pragma unroll
        for (int e = 0; e < R; ++e) v[e] = x[DL_FF_P(i0 + e * hl)];
#pragma unroll
        for (int s = 0; s < S; ++s) {   // base twiddle of the stage with half = 2^s element steps: hs = hl << s
            base[s] = tw[j * (N2 / (hl << s))];
        }
#pragma unroll
        for (int st = 0; st < S; ++st) {
            const int ls = INVERSE ? st : S - 1 - st;     // log2 of the stage's half in element steps
            const int half = 1 << ls;
#pragma unroll
            for (int e = 0; e < R; ++e) {
                if (e & half) continue;
                const int r = e & (half - 1);
                const dl_ff_c w = (r == 0) ? base[ls] : dl_ff_mul(base[ls], dl_ff_unit(r, half));
                if (INVERSE) {
                    const dl_ff_c b = dl_ff_mulc(v[e + half], w), a = v[e];
                    v[e] = a + b;
                    v[e + half] = a - b;
                } else {
                    const dl_ff_c a = v[e], b = v[e + half];
                    v[e] = a + b;
                    v[e + half] = dl_ff_mul(a - b, w);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < R; ++e) x[DL_FF_P(i0 + e * hl)] = v[e];
    }
}

template <bool INVERSE>
__device__ __forceinline__ void dl_ff_pass_s(int S, dl_ff_c* x, const dl_ff_c* __restrict__ tw, int N2, int hl, int tid) {
    switch (S) {
        case 4: dl_ff_pass<4, INVERSE>(x, tw, N2, hl, tid); break;
        case 3: dl_ff_pass<3, INVERSE>(x, tw, N2, hl, tid); break;
        case 2: dl_ff_pass<2, INVERSE>(x, tw, N2, hl, tid); break;
        default: dl_ff_pass<1, INVERSE>(x, tw, N2, hl, tid);
    }
}

// fun [B, n_ell, n]; pre [n]; u [n_ell, N2 + 1] complex; post [n_ell, n]; tw [N2] complex = exp(-i pi m / N2); out [B, n_ell, n]
__global__ __launch_bounds__(DL_FF_THREADS) void dl_fftlog_kernel(const double* __restrict__ fun, const double* __restrict__ pre, const dl_ff_c* __restrict__ u,
                                                                  const double* __restrict__ post, const dl_ff_c* __restrict__ tw, double* __restrict__ out, int n, int pad,
                                                                  int L, int n_ell) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    const int N2 = 1 << L, tid = threadIdx.x;
    dl_ff_c* x = reinterpret_cast<dl_ff_c*>(lds_raw);
    const int ell = blockIdx.x % n_ell;
    const double* f = fun + (size_t)blockIdx.x * n;
    for (int j = tid; j < N2; j += DL_FF_THREADS) {
        const int q0 = 2 * j - pad, q1 = q0 + 1;
        dl_ff_c v;
        v.x = (q0 >= 0 && q0 < n) ? f[q0] * pre[q0] : 0.;
        v.y = (q1 >= 0 && q1 < n) ? f[q1] * pre[q1] : 0.;
        x[DL_FF_P(j)] = v;
    }
    __syncthreads();
    // ---- forward, decimation in frequency: passes of 4 fused stages from the top, the remainder last
    const int Slast = (L & 3) ? (L & 3) : 4;
    {
        int done = 0;   // stages done
        while (done < L) {
            const int S = (L - done > Slast) ? 4 : Slast;
            // the S stages have half-sizes N2 >> (done + 1), ..., N2 >> (done + S): hl = the smallest
            dl_ff_pass_s<false>(S, x, tw, N2, N2 >> (done + S), tid);
            done += S;
            __syncthreads();
        }
    }
    // ---- spectrum: bin m sits at the bit-reversed position
    const dl_ff_c* ul = u + (size_t)ell * (N2 + 1);
    for (int m = tid; m <= (N2 >> 1); m += DL_FF_THREADS) {
        if (m == 0) {
            dl_ff_c z = x[0];   // (position of element 0 is 0)
            // rfft bins 0 and N2 are real; irfft ignores the imaginary parts of both
            const double y0 = (z.x + z.y) * ul[0].x, yn = (z.x - z.y) * ul[N2].x;
            x[0] = dl_ff_c{0.5 * (y0 + yn), 0.5 * (y0 - yn)};
            continue;
        }
        const int mm = N2 - m;
        const int bm = (int)(__brev((unsigned)m) >> (32 - L)), bmm = (int)(__brev((unsigned)mm) >> (32 - L));
        const int pm = DL_FF_P(bm), pmm = DL_FF_P(bmm);
        const dl_ff_c zm = x[pm], zc = dl_ff_conj(x[pmm]), w = tw[m];
        const dl_ff_c s = zm + zc, d = zm - zc;
        const dl_ff_c fe = dl_ff_c{0.5 * s.x, 0.5 * s.y}, fo = dl_ff_c{0.5 * d.y, -0.5 * d.x};   // fo = -i d / 2
        const dl_ff_c wfo = dl_ff_mul(w, fo);
        const dl_ff_c ym = dl_ff_mul(fe + wfo, ul[m]), ymmc = dl_ff_mulc(fe - wfo, ul[mm]);      // ymmc = conj(Y[N2 - m]) = (fe - w fo) conj(u[mm])
        const dl_ff_c gs = ym + ymmc, gd = dl_ff_mulc(ym - ymmc, w);
        const dl_ff_c ge = dl_ff_c{0.5 * gs.x, 0.5 * gs.y}, go = dl_ff_c{0.5 * gd.x, 0.5 * gd.y};
        x[pm] = dl_ff_c{ge.x - go.y, ge.y + go.x};                     // ge + i go
        if (mm != m) x[pmm] = dl_ff_c{ge.x + go.y, go.x - ge.y};       // conj(ge) + i conj(go)
    }
    __syncthreads();
    // ---- inverse, decimation in time, conjugate twiddles: the remainder first, then passes of 4 fused stages
    {
        int done = 0;
        while (done < L) {
            const int S = (done == 0) ? Slast : 4;
            dl_ff_pass_s<true>(S, x, tw, N2, 1 << done, tid);
            done += S;
            __syncthreads();
        }
    }
    // ---- output: A = a'[::-1], un-padded
    const double scale = 1. / (double)N2;
    const double* po = post + (size_t)ell * n;
    double* o = out + (size_t)blockIdx.x * n;
    const double* xr = reinterpret_cast<const double*>(x);
    const int npad = 2 * N2;
    for (int i = tid; i < n; i += DL_FF_THREADS) {
        const int q = npad - 1 - pad - i;   // real sample q = component (q & 1) of complex element q >> 1
        o[i] = po[i] * (xr[2 * DL_FF_P(q >> 1) + (q & 1)] * scale);
    }
}

struct dl_fftlog {
    int device = 0, n = 0, npad = 0, n_ell = 0, L = 0, pad = 0;
    double *pre = nullptr, *u = nullptr, *post = nullptr, *tw = nullptr;
    std::string last_error;
};

static int dl_ff_fail(dl_fftlog* plan, const std::string& msg) {
    if (plan) plan->last_error = msg;
    dl_set_last_error(msg.c_str());
    return 1;
}

#define DL_FF_CHECK(plan, call)                                                                           \
    do {                                                                                                  \
        hipError_t err__ = (call);                                                                        \
        if (err__ != hipSuccess) return dl_ff_fail(plan, std::string(#call) + ": " + hipGetErrorString(err__)); \
    } while (0)

extern "C" {

int dl_fftlog_create(dl_fftlog** out, int device, int32_t n, int32_t npad, int32_t n_ell, const double* pre, const double* u, const double* post) {
    if (!out || !pre || !u || !post) return dl_ff_fail(nullptr, "dl_fftlog_create: null argument");
    *out = nullptr;
    int L = 0;
    while ((2 << L) < npad) ++L;
    if (npad < 16 || npad > 8192 || (2 << L) != npad) return dl_ff_fail(nullptr, "dl_fftlog_create: npad must be a power of two in [16, 8192]");
    if (n < 1 || n > npad || n_ell < 1) return dl_ff_fail(nullptr, "dl_fftlog_create: need 1 <= n <= npad and n_ell >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return dl_ff_fail(nullptr, "dl_fftlog_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return dl_ff_fail(nullptr, "dl_fftlog_create: invalid device ordinal");
    DL_FF_CHECK(nullptr, hipSetDevice(device));
    dl_fftlog* plan = new dl_fftlog();
    plan->device = device; plan->n = n; plan->npad = npad; plan->n_ell = n_ell; plan->L = L; plan->pad = (npad - n) / 2;
    const int N2 = npad / 2;
    std::vector<double> tw(2 * (size_t)N2);
    for (int m = 0; m < N2; ++m) {
        // exact at the multiples of a quarter turn, cos / sin of the first-octant angle elsewhere
        const double ang = -M_PI * (double)m / (double)N2;
        tw[2 * m] = (2 * m == N2) ? 0. : std::cos(ang);
        tw[2 * m + 1] = (m == 0) ? 0. : std::sin(ang);
    }
    struct Up { double** dst; const double* src; size_t count; } ups[] = {{&plan->pre, pre, (size_t)n}, {&plan->u, u, (size_t)n_ell * (N2 + 1) * 2},
                                                                         {&plan->post, post, (size_t)n_ell * n}, {&plan->tw, tw.data(), tw.size()}};
    for (auto& up : ups) {
        hipError_t err = hipMalloc((void**)up.dst, up.count * sizeof(double));
        if (err == hipSuccess) err = hipMemcpy(*up.dst, up.src, up.count * sizeof(double), hipMemcpyHostToDevice);
        if (err != hipSuccess) {
            std::string msg = std::string("dl_fftlog_create: ") + hipGetErrorString(err);
            dl_fftlog_destroy(plan);
            return dl_ff_fail(nullptr, msg);
        }
    }
    const size_t shm = (size_t)(N2 + (N2 >> 4) + 1) * sizeof(dl_ff_c);
    if (shm > 48 * 1024) DL_FF_CHECK(plan, hipFuncSetAttribute((const void*)dl_fftlog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    *out = plan;
    return 0;
}

int dl_fftlog_apply(dl_fftlog* plan, const double* fun_dev, int64_t B, double* out_dev, void* hip_stream) {
    if (!plan) return dl_ff_fail(nullptr, "dl_fftlog_apply: null plan");
    if (B < 0 || (B > 0 && (!fun_dev || !out_dev))) return dl_ff_fail(plan, "dl_fftlog_apply: invalid argument");
    if (B == 0) return 0;
    if (B * plan->n_ell > 0x7fffffffLL) return dl_ff_fail(plan, "dl_fftlog_apply: batch too large for one launch");
    DL_FF_CHECK(plan, hipSetDevice(plan->device));
    const int N2 = plan->npad / 2;
    const size_t shm = (size_t)(N2 + (N2 >> 4) + 1) * sizeof(dl_ff_c);
    hipLaunchKernelGGL(dl_fftlog_kernel, dim3((unsigned)(B * plan->n_ell)), dim3(DL_FF_THREADS), shm, (hipStream_t)hip_stream, fun_dev, plan->pre,
                       reinterpret_cast<const dl_ff_c*>(plan->u), plan->post, reinterpret_cast<const dl_ff_c*>(plan->tw), out_dev, plan->n, plan->pad, plan->L, plan->n_ell);
    DL_FF_CHECK(plan, hipGetLastError());
    return 0;
}

void dl_fftlog_destroy(dl_fftlog* plan) {
    if (!plan) return;
    for (double* p : {plan->pre, plan->u, plan->post, plan->tw})
        if (p) (void)hipFree(p);
    delete plan;
}

}  // extern "C"
