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

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <utility>
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

// ------------------------------------------------------------------------------------------------------------------------
// The reference's grid (2048 points -> npad = 4096, N2 = 2048 complex), specialised: PERSISTENT workgroups, radix-16 passes written as a constant
// 16-point transform + one twiddle per register slot, every twiddle that does not depend on the data kept for the life of the workgroup.
//   N2 = 16 x 16 x 8.  Forward (decimation in frequency, natural -> bit-reversed):
//     pass A  thread j (0..127) owns x[j + 128 e], e < 16: loaded from global memory straight into registers (only e = 4..11 carry data: the rest is the zero
//             padding), constant 16-point DIF, slot e times W_2048^(j brev4(e)) -- four resident twiddles per thread, the other eleven are products -- then LDS;
//     pass B  thread t = 8 g + j owns x[128 g + j + 8 e]: LDS -> registers, 16-point DIF, slot e times W_128^(j brev4(e)) (128 distinct values: a 2 KB LDS table);
//     pass C  two blocks of 8 consecutive positions per thread: plain 8-point DIF (no twiddles left), on registers, followed IN THE SAME REGISTERS by
//   the spectrum step (the two blocks of a thread are chosen so that they are closed under the pairing m <-> N2 - m: see the phase itself) and the inverse pass C';
//   u_ell[m], u_ell[N2 - m] come from tables laid out in work-item order at plan creation, requested right after pass A (they arrive behind pass B).
//   Inverse = the conjugate transposes in reverse order (C', B', A'); pass A' ends in registers and is stored from there, reversed and scaled (16-byte stores).
//   Three LDS round trips (the general kernel: seven), no table loads inside the passes, raw s_barrier with lgkmcnt-only waits so that the table requests stay
//   in flight across barriers.
// ------------------------------------------------------------------------------------------------------------------------
#define DL_FF4_N2 2048
#define DL_FF4_L 11
#ifndef DL_FF4_SYNC_AFTER_APRIME
#define DL_FF4_SYNC_AFTER_APRIME 1
#endif
#ifndef DL_FF4_POST_STAGE
#define DL_FF4_POST_STAGE 0   // the output factors are requested after this many stages of the last 16-point transform (0: with the input factors, before it)
#endif

// v * exp(-i pi q / 8) (forward) or v * exp(+i pi q / 8) (INV), q = 0..7 known at compile time after unrolling
template <bool INV>
__device__ __forceinline__ dl_ff_c dl_ff_rot(int q, dl_ff_c v) {
    const double h = 0.70710678118654752, c1 = 0.92387953251128674, s1 = 0.38268343236508977;
    switch (q) {
        case 0: return v;
        case 4: return INV ? dl_ff_c{-v.y, v.x} : dl_ff_c{v.y, -v.x};
        case 2: return INV ? dl_ff_c{(v.x - v.y) * h, (v.x + v.y) * h} : dl_ff_c{(v.x + v.y) * h, (v.y - v.x) * h};
        case 6: return INV ? dl_ff_c{-(v.x + v.y) * h, (v.x - v.y) * h} : dl_ff_c{(v.y - v.x) * h, -(v.x + v.y) * h};
        default: {
            const double c = (q == 1) ? c1 : (q == 3) ? s1 : (q == 5) ? -s1 : -c1;    // cos(pi q / 8)
            const double s = (q == 1 || q == 7) ? s1 : c1;                               // sin(pi q / 8)
            return INV ? dl_ff_c{v.x * c - v.y * s, v.y * c + v.x * s} : dl_ff_c{v.x * c + v.y * s, v.y * c - v.x * s};
        }
    }
}

// constant 2^S-point transform on registers: DIF forward (natural -> bit-reversed) or its conjugate transpose (bit-reversed -> natural)
template <int S, bool INV, int ST0 = 0, int ST1 = S>
__device__ __forceinline__ void dl_ff_const(dl_ff_c* v) {
    constexpr int R = 1 << S;
#pragma unroll
    for (int st = ST0; st < ST1; ++st) {
        const int half = INV ? (1 << st) : (R >> (st + 1));
#pragma unroll
        for (int e = 0; e < R; ++e) {
            if (e & half) continue;
            const int q = (e & (half - 1)) * (8 / half);
            if (INV) {
                const dl_ff_c a = v[e], b = dl_ff_rot<true>(q, v[e + half]);
                v[e] = a + b;
                v[e + half] = a - b;
            } else {
                const dl_ff_c a = v[e], b = v[e + half];
                v[e] = a + b;
                v[e + half] = dl_ff_rot<false>(q, a - b);
            }
        }
    }
}

// forward stages FIRST .. S - 1 of dl_ff_const<S, false> (the earlier ones done by the caller)
template <int S, int FIRST>
__device__ __forceinline__ void dl_ff_const_from(dl_ff_c* v) {
    constexpr int R = 1 << S;
#pragma unroll
    for (int st = FIRST; st < S; ++st) {
        const int half = R >> (st + 1);
#pragma unroll
        for (int e = 0; e < R; ++e) {
            if (e & half) continue;
            const int q = (e & (half - 1)) * (8 / half);
            const dl_ff_c a = v[e], b = v[e + half];
            v[e] = a + b;
            v[e + half] = dl_ff_rot<false>(q, a - b);
        }
    }
}

__device__ __forceinline__ constexpr int dl_ff_brev4(int e) { return ((e & 1) << 3) | ((e & 2) << 1) | ((e & 4) >> 1) | ((e & 8) >> 3); }

// T[q] = w1^(q & 1) w2^((q >> 1) & 1) w4^((q >> 2) & 1), q = 1 .. 7 (four products); the caller forms T[q + 8] = T[q] w8 where it uses them
__device__ __forceinline__ void dl_ff_twiddle_products(dl_ff_c& w1, dl_ff_c& w2, dl_ff_c& w4, dl_ff_c& w8, dl_ff_c* T) {
    // (opaque to the optimiser: the products are loop invariants, and hoisted out of the transform loop they would be fifteen resident twiddles again)
    __asm__ volatile("" : "+v"(w1.x), "+v"(w1.y), "+v"(w2.x), "+v"(w2.y), "+v"(w4.x), "+v"(w4.y), "+v"(w8.x), "+v"(w8.y));
    T[1] = w1; T[2] = w2; T[4] = w4;
    T[3] = dl_ff_mul(w1, w2); T[5] = dl_ff_mul(w1, w4); T[6] = dl_ff_mul(w2, w4); T[7] = dl_ff_mul(T[3], w4);
}

// spectrum step on the pair (X[m], X[N2 - m]) in place: real-FFT unpacking X = Fe + W^m Fo, multiplication by u[m] / u[N2 - m], re-packing for the inverse transform
__device__ __forceinline__ void dl_ff_pair_update(dl_ff_c& xm, dl_ff_c& xmm, dl_ff_c w, dl_ff_c um, dl_ff_c umm) {
    const dl_ff_c zm = xm, zc = dl_ff_conj(xmm);
    const dl_ff_c s = zm + zc, d = zm - zc;
    const dl_ff_c fe = dl_ff_c{0.5 * s.x, 0.5 * s.y}, fo = dl_ff_c{0.5 * d.y, -0.5 * d.x};   // fo = -i d / 2
    const dl_ff_c wfo = dl_ff_mul(w, fo);
    const dl_ff_c ym = dl_ff_mul(fe + wfo, um), ymmc = dl_ff_mulc(fe - wfo, umm);            // ymmc = conj(Y[N2 - m])
    const dl_ff_c gs = ym + ymmc, gd = dl_ff_mulc(ym - ymmc, w);
    const dl_ff_c ge = dl_ff_c{0.5 * gs.x, 0.5 * gs.y}, go = dl_ff_c{0.5 * gd.x, 0.5 * gd.y};
    xm = dl_ff_c{ge.x - go.y, ge.y + go.x};                     // ge + i go
    xmm = dl_ff_c{ge.x + go.y, go.x - ge.y};                    // conj(ge) + i conj(go)
}

// this wave's LDS traffic has landed, then the workgroup barrier: global requests stay in flight (no vmcnt wait)
__device__ __forceinline__ void dl_ff_lds_barrier() {
    __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The spectrum step pairs bin m (position brev11(m)) with bin N2 - m.  The bins m = 2^k (4 j + 1) occupy the positions [2^(10-k), 2^(10-k) + 2^(9-k)), their partners
// the next 2^(9-k) positions in reverse order; in blocks of 8 positions: thread t owns block A and its mirror B, slot e of A pairs with slot 7 - e of B.
#define DL_FF4_NU 1032    // coefficient table entries per multipole: item e * 128 + t (slot e of thread t's block A), then the seven pairs inside blocks 0 / 1
__host__ __device__ __forceinline__ void dl_ff4_blocks(int t, int& A, int& B) {
    if (t < 64) { A = 128 + t; B = 255 - t; }                    // k = 0: positions 1024 .. 2047
    else if (t < 96) { A = t; B = 191 - t; }                     // k = 1: 64 + j, 127 - j with j = t - 64
    else if (t < 112) { A = t - 64; B = 159 - t; }               // k = 2: 32 + j, 63 - j, j = t - 96
    else if (t < 120) { A = t - 96; B = 143 - t; }               // k = 3: 16 + j, 31 - j, j = t - 112
    else if (t < 124) { A = t - 112; B = 135 - t; }              // k = 4: 8 + j, 15 - j, j = t - 120
    else if (t < 126) { A = t - 120; B = 131 - t; }              // k = 5: 4 + j, 7 - j, j = t - 124
    else if (t == 126) { A = 2; B = 3; }                         // k = 6
    else { A = 0; B = 1; }                                       // positions 0 .. 15: pairs inside the two blocks (dl_ff4_special_pair)
}
// pair i (0 .. 6) inside blocks 0 / 1: positions (8, 15), (9, 14), (10, 13), (11, 12), (4, 7), (5, 6), (2, 3)
__host__ __device__ __forceinline__ void dl_ff4_special_pair(int i, int& p, int& pm) {
    if (i < 4) { p = 8 + i; pm = 15 - i; }
    else if (i < 6) { p = i; pm = 11 - i; }
    else { p = 2; pm = 3; }
}
__host__ __device__ __forceinline__ constexpr int dl_ff_brev3(int e) { return ((e & 1) << 2) | (e & 2) | ((e & 4) >> 2); }

// entry m (< 4096) of the full circle exp(-i pi m / 2048) from the half-circle table
__device__ __forceinline__ dl_ff_c dl_ff_tw_full(const dl_ff_c* __restrict__ tw, int m) {
    const dl_ff_c w = tw[m & (DL_FF4_N2 - 1)];
    return (m & DL_FF4_N2) ? dl_ff_c{-w.x, -w.y} : w;
}

// fun / out [total, 2048] (total = B n_ell transforms, multipole = transform % n_ell); pre [2048]; u [n_ell, 2049] complex; post [n_ell, 2048]; tw [2048] complex
__global__ __launch_bounds__(DL_FF_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void dl_fftlog4096_kernel(const double* __restrict__ fun, const double* __restrict__ pre, const dl_ff_c* __restrict__ u, const dl_ff_c* __restrict__ u1,
                          const dl_ff_c* __restrict__ u2, const double* __restrict__ post, const dl_ff_c* __restrict__ tw, double* __restrict__ out, int n_ell, int total) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    constexpr int N2 = DL_FF4_N2, n = 2048;    // (zero padding: 1024 on either side of the 2048 samples)
    const int tid = threadIdx.x;
    dl_ff_c* x = reinterpret_cast<dl_ff_c*>(lds_raw);
    dl_ff_c* twB = x + (N2 + (N2 >> 4) + 1);             // [q][j]: W_128^(j q)
    // ---- persistent twiddles
    // W_2048^(j q) = exp(-i pi 2 j q / 2048) for q = 1, 2, 4, 8; the other eleven are products formed where they are used (keeping all fifteen resident, with
    // the operands in flight on top, overflows the 256 registers a wave has at this occupancy)
    const dl_ff_c twA1 = dl_ff_tw_full(tw, 2 * tid), twA2 = dl_ff_tw_full(tw, 4 * tid), twA4 = dl_ff_tw_full(tw, 8 * tid), twA8 = dl_ff_tw_full(tw, 16 * tid);
    // spectrum step: the pair of 8-blocks of this thread and W_4096^(brev8(blkA)); lanes 64 .. 70: the pair inside blocks 0 / 1 they look after
    int blkA, blkB;
    dl_ff4_blocks(tid, blkA, blkB);
    const dl_ff_c wt = tw[(int)(__brev((unsigned)blkA) >> 24)];
    int posS = 2, posSm = 3;
    dl_ff4_special_pair(tid >= 64 && tid < 71 ? tid - 64 : 6, posS, posSm);
    const dl_ff_c wS = tw[(int)(__brev((unsigned)posS) >> (32 - DL_FF4_L))];
    { const int j = tid & 7, q = tid >> 3; twB[q * 8 + j] = dl_ff_tw_full(tw, 32 * j * q); } // W_128^(j q) = exp(-i pi 32 j q / 2048)
    const int jB = tid & 7, gB = tid >> 3, rtid = 127 - tid;
    __syncthreads();
    // the input row of a transform is requested during the inverse passes of the one before it (HBM latency off the critical path)
    dl_ff_c fv[8];
    if ((int)blockIdx.x < total) {
#pragma unroll
        for (int e = 0; e < 8; ++e) fv[e] = reinterpret_cast<const dl_ff_c*>(fun + (size_t)blockIdx.x * n + 256 * e)[tid];
    }
    // ... and the input factors during the last inverse pass (kept resident they cost 32 registers through the passes that need them most)
    dl_ff_c pr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) pr[e] = reinterpret_cast<const dl_ff_c*>(pre + 256 * e)[tid];
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
        const int ell = id % n_ell;
        const dl_ff_c* ul = u + (size_t)ell * (N2 + 1);
        const dl_ff_c *u1l = u1 + (size_t)ell * DL_FF4_NU, *u2l = u2 + (size_t)ell * DL_FF4_NU;
        dl_ff_c v[16];
        // ---- pass A: element j + 128 e = real samples q0 = 2 (j + 128 e) - pad, q0 + 1; data for e = 4 .. 11 only, so the first stage (pairs (e, e + 8)) has one
        //      zero operand everywhere: v[e] = b, v[e + 8] = rot(-b) for e < 4 (a = 0), v[e] = a, v[e + 8] = rot(a) for e >= 4 (b = 0)
        {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // samples q0 = 2 tid + 256 e (= 2 (tid + 128 (e + 4)) - 1024), q0 + 1: uniform base + one per-thread offset for every table (tid or 127 - tid)
                const dl_ff_c d = dl_ff_c{fv[e].x * pr[e].x, fv[e].y * pr[e].y};   // element e + 4
                if (e < 4) { v[e + 4] = d; v[e + 12] = dl_ff_rot<false>(e + 4, d); }                       // a = d, b = 0; slot (e + 4) + 8, q = e + 4
                else { v[e - 4] = d; v[e + 4] = dl_ff_rot<false>(e - 4, dl_ff_c{-d.x, -d.y}); }           // a = 0, b = d (element e + 4 = (e - 4) + 8); q = e - 4
            }
        }
        dl_ff_const_from<4, 1>(v);
        x[DL_FF_P(tid)] = v[0];
        {
            dl_ff_c T[8], w1 = twA1, w2 = twA2, w4 = twA4, w8 = twA8;
            dl_ff_twiddle_products(w1, w2, w4, w8, T);
#pragma unroll
            for (int e = 1; e < 16; ++e) {
                const int q = dl_ff_brev4(e);
                const dl_ff_c t = (q < 8) ? T[q] : (q == 8) ? w8 : dl_ff_mul(T[q - 8], w8);
                x[DL_FF_P(tid + 128 * e)] = dl_ff_mul(v[e], t);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the spectrum step's coefficients: requested now, used after passes B and C
        dl_ff_c um[4], umm[4];     // bins m = tid + 128 i, i < 4 now; i + 4 takes the place of i as soon as i is consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            um[i] = (u1l + 128 * i)[tid];                     // u[m] of the pair (slot i of block A, slot 7 - i of block B)
            umm[i] = (u2l + 128 * i)[tid];                    // u[N2 - m]
        }
        dl_ff_lds_barrier();
        // ---- pass B
        {
            const int i0 = 128 * gB + jB;
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = x[DL_FF_P(i0 + 8 * e)];
            dl_ff_const<4, false>(v);
            x[DL_FF_P(i0)] = v[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) x[DL_FF_P(i0 + 8 * e)] = dl_ff_mul(v[e], twB[dl_ff_brev4(e) * 8 + jB]);
        }
        dl_ff_lds_barrier();
        // ---- pass C, spectrum step and inverse pass C' in ONE register-resident phase.  Pass C works on blocks of 8 consecutive positions; the bins of the run of
        //      positions [2^(10-k), 2^(10-k) + 2^(9-k)) have their partners N2 - m in the next run of the same length in reverse order (dl_ff4_blocks), so a pair of
        //      8-blocks (A, B = the mirror of A in the partner run) is closed under the pairing: slot e of A goes with slot 7 - e of B.  A thread loads its two blocks,
        //      runs the two 8-point transforms, the 8 pair updates and the two inverse 8-point transforms on registers, and writes back: two LDS round trips and two
        //      barriers fewer than with the three phases apart.  127 block pairs cover the positions 16 .. 2047 (dl_ff4_blocks); blocks 0 and 1 (thread 127) pair
        //      within themselves: that thread parks its transformed blocks in LDS, seven lanes of its own wave do the seven pairs (positions 2 .. 15) and one the two
        //      real bins (positions 0, 1) there, and it takes the blocks back -- same wave: program order, no barrier.
        //      Bin of slot e of block A: m = brev3(e) 2^8 + brev8(A), so W_4096^m = W^(brev8(A)) (resident) x exp(-i pi brev3(e) / 8) (compile time).
        {
            dl_ff_c a[8], b[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] = x[DL_FF_P(8 * blkA + e)]; b[e] = x[DL_FF_P(8 * blkB + e)]; }
            dl_ff_const<3, false>(a);
            dl_ff_const<3, false>(b);
            if (tid == 127) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[DL_FF_P(e)] = a[e]; x[DL_FF_P(8 + e)] = b[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const dl_ff_c w = dl_ff_rot<false>(dl_ff_brev3(e), wt);
                dl_ff_pair_update(a[e], b[7 - e], w, um[e & 3], umm[e & 3]);
                if (e < 4) { um[e] = (u1l + 128 * (e + 4))[tid]; umm[e] = (u2l + 128 * (e + 4))[tid]; }
            }
            if (tid >= 64) {   // wave 1: the pairs inside blocks 0 and 1
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int ls = tid - 64;
                if (ls < 7) {
                    const int pm = DL_FF_P(posS), pmm = DL_FF_P(posSm);
                    dl_ff_c zm = x[pm], zmm = x[pmm];
                    dl_ff_pair_update(zm, zmm, wS, (u1l + 1024)[ls], (u2l + 1024)[ls]);
                    x[pm] = zm; x[pmm] = zmm;
                } else if (ls == 7) {
                    // m = 1024 = N2 - m, W^m = -i: the pair formulas collapse to x <- x conj(u[1024]) (position brev(1024) = 1)
                    x[DL_FF_P(1)] = dl_ff_mulc(x[DL_FF_P(1)], ul[1024]);
                    // m = 0: rfft bins 0 and N2 are real (packed as the two components of element 0); irfft ignores the imaginary parts of both
                    const dl_ff_c z = x[0];
                    const double y0 = (z.x + z.y) * ul[0].x, yn = (z.x - z.y) * ul[N2].x;
                    x[0] = dl_ff_c{0.5 * (y0 + yn), 0.5 * (y0 - yn)};
                }
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (tid == 127) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { a[e] = x[DL_FF_P(e)]; b[e] = x[DL_FF_P(8 + e)]; }
                }
            }
            dl_ff_const<3, true>(a);
            dl_ff_const<3, true>(b);
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[DL_FF_P(8 * blkA + e)] = a[e]; x[DL_FF_P(8 * blkB + e)] = b[e]; }
        }
        // the next transform's input row: requested now, used at the top of the next iteration
        {
            const int next = (id + (int)gridDim.x < total) ? id + (int)gridDim.x : id;
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[e] = reinterpret_cast<const dl_ff_c*>(fun + (size_t)next * n + 256 * e)[tid];
        }
        dl_ff_lds_barrier();
        // ---- inverse pass B'
        {
            const int i0 = 128 * gB + jB;
            v[0] = x[DL_FF_P(i0)];
#pragma unroll
            for (int e = 1; e < 16; ++e) v[e] = dl_ff_mulc(x[DL_FF_P(i0 + 8 * e)], twB[dl_ff_brev4(e) * 8 + jB]);
            dl_ff_const<4, true>(v);
#pragma unroll
            for (int e = 0; e < 16; ++e) x[DL_FF_P(i0 + 8 * e)] = v[e];
        }
        dl_ff_lds_barrier();
        // ---- inverse pass A' -> registers -> global memory; the output factors are requested half-way through the 16-point transform (earlier, they
        //      are 32 more live registers where the pass needs the most)
        v[0] = x[DL_FF_P(tid)];
        {
            dl_ff_c T[8], w1 = twA1, w2 = twA2, w4 = twA4, w8 = twA8;
            dl_ff_twiddle_products(w1, w2, w4, w8, T);
#pragma unroll
            for (int e = 1; e < 16; ++e) {
                const int q = dl_ff_brev4(e);
                const dl_ff_c t = (q < 8) ? T[q] : (q == 8) ? w8 : dl_ff_mul(T[q - 8], w8);
                v[e] = dl_ff_mulc(x[DL_FF_P(tid + 128 * e)], t);
                if (e == 5 || e == 10) __builtin_amdgcn_sched_barrier(0);   // (three batches of LDS reads: all sixteen in flight at once overflow the register file here)
            }
        }
#if DL_FF4_SYNC_AFTER_APRIME
        dl_ff_lds_barrier();
#else
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        {
            const double* prel = pre;
            __asm__ volatile("" : "+s"(prel));    // (not a loop invariant for the optimiser: see above)
#pragma unroll
            for (int e = 0; e < 8; ++e) pr[e] = reinterpret_cast<const dl_ff_c*>(prel + 256 * e)[tid];
        }
        dl_ff_const<4, true, 0, DL_FF4_POST_STAGE>(v);
        __builtin_amdgcn_sched_barrier(0);
        dl_ff_c pv[8];
        {
            const double* po = post + (size_t)ell * n;
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = reinterpret_cast<const dl_ff_c*>(po + (1792 - 256 * e))[rtid];     // post[2046 - 2 tid - 256 e], [.. + 1]
        }
        __builtin_amdgcn_sched_barrier(0);
        dl_ff_const<4, true, DL_FF4_POST_STAGE, 4>(v);
        {
            // A = a'[::-1], un-padded: real sample q -> out[3071 - q]; element j + 128 e (e = 4 .. 11) = samples q0 = 2 tid + 256 e (e - 4 -> e), q0 + 1 -> out[2046 - q0'], out[2047 - q0']
            const double scale = 1. / (double)N2;
            double* o = out + (size_t)id * n;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const dl_ff_c r = v[e + 4];
                reinterpret_cast<dl_ff_c*>(o + (1792 - 256 * e))[rtid] = dl_ff_c{pv[e].x * (r.y * scale), pv[e].y * (r.x * scale)};
            }
        }
    }
}

struct dl_fftlog {
    int device = 0, n = 0, npad = 0, n_ell = 0, L = 0, pad = 0, n_cu = 256;
    double *pre = nullptr, *u = nullptr, *post = nullptr, *tw = nullptr;
    double *u1 = nullptr, *u2 = nullptr;   // npad = 4096: u[m], u[N2 - m] in the order of the spectrum step's work items ([n_ell, 1024] complex each)
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
    if (npad == 4096 && n == 2048) {
        std::vector<double> u1((size_t)n_ell * DL_FF4_NU * 2), u2(u1.size());
        auto brev11 = [](int p) { int m = 0; for (int bit = 0; bit < 11; ++bit) m |= ((p >> bit) & 1) << (10 - bit); return m; };
        for (int l = 0; l < n_ell; ++l) {
            const double* ul = u + (size_t)l * (N2 + 1) * 2;
            for (int r = 0; r < DL_FF4_NU; ++r) {
                int p = 2, pm = 3;                       // (entries that no work item reads: any valid pair)
                if (r < 1024 && (r & 127) < 127) { int A, B; dl_ff4_blocks(r & 127, A, B); p = 8 * A + (r >> 7); pm = 8 * B + 7 - (r >> 7); }
                else if (r >= 1024 && r < 1031) dl_ff4_special_pair(r - 1024, p, pm);
                const int m = brev11(p);
                if (brev11(pm) != (N2 - m) % N2) { dl_fftlog_destroy(plan); return dl_ff_fail(nullptr, "dl_fftlog_create: internal error (pairing of the spectrum step)"); }
                const size_t o = ((size_t)l * DL_FF4_NU + r) * 2;
                u1[o] = ul[2 * m]; u1[o + 1] = ul[2 * m + 1];
                u2[o] = ul[2 * (N2 - m)]; u2[o + 1] = ul[2 * (N2 - m) + 1];
            }
        }
        for (auto up : {std::make_pair(&plan->u1, &u1), std::make_pair(&plan->u2, &u2)}) {
            hipError_t err = hipMalloc((void**)up.first, up.second->size() * sizeof(double));
            if (err == hipSuccess) err = hipMemcpy(*up.first, up.second->data(), up.second->size() * sizeof(double), hipMemcpyHostToDevice);
            if (err != hipSuccess) {
                std::string msg = std::string("dl_fftlog_create: ") + hipGetErrorString(err);
                dl_fftlog_destroy(plan);
                return dl_ff_fail(nullptr, msg);
            }
        }
    }
    const size_t shm = (size_t)(N2 + (N2 >> 4) + 1) * sizeof(dl_ff_c);
    if (shm > 48 * 1024) DL_FF_CHECK(plan, hipFuncSetAttribute((const void*)dl_fftlog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) plan->n_cu = prop.multiProcessorCount;
    }
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
    static const bool generic_only = getenv("DL_FFTLOG_GENERIC") && atoi(getenv("DL_FFTLOG_GENERIC")) != 0;
    if (plan->npad == 4096 && plan->n == 2048 && !generic_only) {   // the reference's grid: persistent radix-16 kernel
        const size_t shm4 = (size_t)(N2 + (N2 >> 4) + 1 + 128) * sizeof(dl_ff_c);
        const int64_t total = B * plan->n_ell;
        const int64_t resident = 4 * (int64_t)plan->n_cu;           // four workgroups per CU (LDS)
        hipLaunchKernelGGL(dl_fftlog4096_kernel, dim3((unsigned)std::min<int64_t>(total, resident)), dim3(DL_FF_THREADS), shm4, (hipStream_t)hip_stream, fun_dev, plan->pre,
                           reinterpret_cast<const dl_ff_c*>(plan->u), reinterpret_cast<const dl_ff_c*>(plan->u1), reinterpret_cast<const dl_ff_c*>(plan->u2), plan->post,
                           reinterpret_cast<const dl_ff_c*>(plan->tw), out_dev, plan->n_ell, (int)total);
        DL_FF_CHECK(plan, hipGetLastError());
        return 0;
    }
    const size_t shm = (size_t)(N2 + (N2 >> 4) + 1) * sizeof(dl_ff_c);
    hipLaunchKernelGGL(dl_fftlog_kernel, dim3((unsigned)(B * plan->n_ell)), dim3(DL_FF_THREADS), shm, (hipStream_t)hip_stream, fun_dev, plan->pre,
                       reinterpret_cast<const dl_ff_c*>(plan->u), plan->post, reinterpret_cast<const dl_ff_c*>(plan->tw), out_dev, plan->n, plan->pad, plan->L, plan->n_ell);
    DL_FF_CHECK(plan, hipGetLastError());
    return 0;
}

void dl_fftlog_destroy(dl_fftlog* plan) {
    if (!plan) return;
    for (double* p : {plan->pre, plan->u, plan->post, plan->tw, plan->u1, plan->u2})
        if (p) (void)hipFree(p);
    delete plan;
}

}  // extern "C"
