// dl_ensemble.hip -- device-resident affine-invariant ensemble sampler (include/desilike_amd.h, dl_ensemble_*).
//
// What the reference does on the host per ensemble update (desilike/samplers/emcee.py:69-111 -> emcee.EnsembleSampler(vectorize=True) with its default
// StretchMove: Goodman & Weare 2010, two half-ensemble updates; the log-posterior of a half is ONE batched call, desilike/samplers/base.py:144-200) runs here as one
// enqueued sequence per half-step with no host synchronisation:
//
//     [accept previous half | stretch proposals]  ->  dl_eval_logposterior (this rank's share)  ->  ncclAllGather (N > 1)  ->  next ...
//
// Walker positions, log-posteriors, acceptance counts and the random number generator live on the device; the host only enqueues and, at the end of a run,
// drains the chain.  Random numbers are COUNTER-BASED (Philox4x32-10, Salmon et al. 2011): the draw for (iteration, half-step, walker slot) is a pure
// function of the seed, so every rank of a sharded run holds the same ensemble without exchanging anything but log-posteriors, and the NumPy
// ``EnsembleStretchMove`` of the host package reproduces the chain bit for bit with the same generator (tests/test_gpu_sampler.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/desilike_amd.h"
#include "dl_kernels.h"
#include "dl_finalize_part.h"
#include "dl_ens_fold.h"
#include "dl_scalar_prefetch.h"

namespace {

int fail(const std::string& msg) {
    dl_set_last_error(msg.c_str());
    return 1;
}

#define DL_ENS_HIP(call)                                                                              \
    do {                                                                                              \
        hipError_t err__ = (call);                                                                    \
        if (err__ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(err__));   \
    } while (0)

#define DL_ENS_THREADS 1024   // one workgroup
struct DlEnsArgs {
    double* coords;        // [nw, P]
    double* logp;          // [nw]
    long long* nacc;       // [nw]
    double* prop;          // [half_pad, P] proposals of the pending half-step
    double* factors;       // [half] (P - 1) log z
    double* newlp;         // [half_pad] log-posteriors of the proposals (all ranks' shares after the all-gather)
    double* chain;         // record target of this launch: [nw, P] or null
    double* chain_logp;    // [nw] or null
    const double* part;    // deferred finalize (single rank, plain likelihood): partial chi2 [half, n_tiles] of the pending proposals straight from the chi2 GEMM,
    const double* priors;  // ... prior table [P, 5]: this kernel sums the partials, adds the priors and applies the status rules itself (no finalize launch)
    int32_t n_tiles;
    int32_t stage_parts;   // partial chi2 staged in LDS with the rest (0: read from global memory in the accept phase -- ensembles whose state alone fills the LDS)
    int32_t nw, P;
    double a, offset;
    uint32_t k0, k1;
    long long it_acc, it_prop;   // iteration of the half-step to accept / to propose
    int32_t half_acc, half_prop; // 0 / 1, or -1: nothing to accept / propose
    DlEnsSplit split_acc, split_prop;   // the splits of those two iterations
    unsigned long long* stamps;         // DL_ENS_STAMPS diagnostics (nullptr in production): time spent per phase, summed over the launches (100 MHz clock), [7] = launches
};

// draws of slot j of the half-step to accept: walker and log(u); of the half-step to propose: walker, partner, z and (P - 1) log z
// (emcee moves/red_blue.py, moves/stretch.py: z ~ g(z) on [1/a, a], partner from the complementary half)
__device__ __forceinline__ void dl_ens_draw_accept(const DlEnsArgs& s, int j, int half, int& i, double& logu) {
#pragma clang fp contract(off)
    i = dl_ens_split_at(s.split_acc, s.half_acc * half + j);
    const DlPhilox r = dl_philox4x32((uint32_t)s.it_acc, (uint32_t)((unsigned long long)s.it_acc >> 32), (uint32_t)j, DL_ENS_STREAM_ACCEPT + s.half_acc, s.k0, s.k1);
    logu = log(dl_uniform53(r.x[0], r.x[1]));
}

__device__ __forceinline__ void dl_ens_draw_move(const DlEnsArgs& s, int j, int half, int& is, int& ic, double& zz, double& factor) {
#pragma clang fp contract(off)
    const DlPhilox r = dl_philox4x32((uint32_t)s.it_prop, (uint32_t)((unsigned long long)s.it_prop >> 32), (uint32_t)j, DL_ENS_STREAM_MOVE + s.half_prop, s.k0, s.k1);
    const double u = dl_uniform53(r.x[0], r.x[1]);
    const double t = (s.a - 1.) * u + 1.;
    zz = (t * t) / s.a;
    factor = (s.P - 1.) * log(zz);
    ic = dl_ens_split_at(s.split_prop, (1 - s.half_prop) * half + (int)(r.x[2] % (uint32_t)half));
    is = dl_ens_split_at(s.split_prop, s.half_prop * half + j);
}

// asynchronous copy of n doubles (16-byte aligned source) into LDS: 16-byte LDS-DMA chunks (global_load_lds_dwordx4: no registers, nothing waits until the
// workgroup's barrier), lane l of a wavefront writes chunk l of the wavefront's 1 KB slot.  rot_c > 0: rows of rot_c chunks, chunk c of row r lands at position
// (c + r) % rot_c of the row (threads that walk their own row chunk by chunk then spread over the LDS banks instead of all hitting the same two).
__device__ __forceinline__ void dl_ens_stage(double* dst, const double* __restrict__ src, int n, int rot_c, int wave, int lane, int nthr) {
    const int chunks = n >> 1;
    for (int c0 = wave * 64; c0 < chunks; c0 += nthr) {
        const int slot = c0 + lane;
        if (slot < chunks) {
            int from = slot;
            if (rot_c > 0) { const int r = slot / rot_c, cp = slot - r * rot_c; int c = cp - r % rot_c; if (c < 0) c += rot_c; from = r * rot_c + c; }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2 * (size_t)from), (__attribute__((address_space(3))) void*)(dst + 2 * (size_t)c0), 16, 0, 0);
        }
    }
    if ((n & 1) && wave == 0 && lane == 0) dst[n - 1] = src[n - 1];
}

// One workgroup: the ensemble is a few hundred walkers x <= 64 parameters.  Everything the launch reads -- positions, log-posteriors, the pending proposals, their
// partial chi2 and the prior table -- is requested at the top as LDS-DMA (one round trip for all of it; with register loads in run-time loops every loop iteration
// was a round trip of its own: 6 us), the random draws and logarithms of both phases are computed in the shadow of that round trip (accept draws on the first waves,
// move draws on waves of the other half of the workgroup: different SIMD slots), and the phases after the barrier touch LDS only and end in fire-and-forget stores.
// (diagnostics: everything outstanding is waited for, then the time since the previous stamp is added to slot k by thread 0)
#define DL_ENS_STAMP(k) if (s.stamps != nullptr) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0) atomicAdd(s.stamps + (k), t_ - t_last); t_last = t_; }
template <int THREADS>
__global__ __launch_bounds__(THREADS) void dl_ensemble_step_lds_kernel(const DlEnsArgs s) {
#pragma clang fp contract(off)   // the NumPy driver rounds after every operation: no fused multiply-adds here
    extern __shared__ __attribute__((aligned(16))) double dl_ens_lds[];
    unsigned long long t_last = s.stamps != nullptr ? __builtin_amdgcn_s_memrealtime() : 0ull;
    dl_kernarg_prefetch<sizeof(DlEnsArgs)>();
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = s.nw, half = nw / 2, P = s.P, n_tiles = s.part != nullptr ? s.n_tiles : 0, n_tiles_lds = s.stage_parts ? n_tiles : 0;
    double* coords = dl_ens_lds;                                  // [nw, P]
    double* logp = coords + (size_t)nw * P;                       // [nw]
    double* prop = logp + nw;                                     // [half, P] pending proposals (the new ones go straight to global memory)
    double* parts = prop + (((size_t)half * P + 1) & ~(size_t)1); // [half, n_tiles], chunks rotated by the row index
    double* priors = parts + (size_t)half * n_tiles_lds;          // [P, 5]
    const double inf = __builtin_huge_val();
    const bool accepting = s.half_acc >= 0, proposing = s.half_prop >= 0;
    dl_ens_stage(coords, s.coords, nw * P, 0, wave, lane, nthr);
    dl_ens_stage(logp, s.logp, nw, 0, wave, lane, nthr);
    double lp0 = 0., f0 = 0.;
    if (accepting) {
        dl_ens_stage(prop, s.prop, half * P, 0, wave, lane, nthr);
        if (s.part != nullptr) {
            if (n_tiles_lds) dl_ens_stage(parts, s.part, half * n_tiles, n_tiles / 2, wave, lane, nthr);
            for (int e = nthr - 1 - tid; e < 5 * P; e += nthr) priors[e] = s.priors[e];   // (register loads: on the last wavefront, which has no draws to make meanwhile)
        } else if (tid < half) lp0 = s.newlp[tid];
        if (tid < half) f0 = s.factors[tid];
    }
    DL_ENS_STAMP(0)   // requests issued (and, with stamps on, landed: the wait is part of the stamp)
    __builtin_amdgcn_sched_barrier(0);
    // in the shadow of the round trip: the draws of this thread's first slot of either phase
    const int jp0 = (tid + nthr / 2) % nthr;      // the move phase starts half a workgroup away from the accept phase
    int i0 = 0, is0 = 0, ic0 = 0;
    double logu0 = 0., zz0 = 0., fac0 = 0.;
    if (accepting && tid < half) dl_ens_draw_accept(s, tid, half, i0, logu0);
    if (proposing && jp0 < half) dl_ens_draw_move(s, jp0, half, is0, ic0, zz0, fac0);
    __builtin_amdgcn_sched_barrier(0);
    DL_ENS_STAMP(1)   // draws
    __syncthreads();
    DL_ENS_STAMP(2)   // barrier
    if (accepting) {
        // accept / reject the pending proposals (emcee moves/red_blue.py: lnpdiff = factors + new_log_prob - log_prob; accepted = log(u) < lnpdiff)
        for (int j = tid; j < half; j += nthr) {
            int i = i0;
            double logu = logu0, lp = lp0, fj = f0;
            if (j != tid) { dl_ens_draw_accept(s, j, half, i, logu); fj = s.factors[j]; if (s.part == nullptr) lp = s.newlp[j]; }
            if (s.part != nullptr) {
                // partial chi2 of the slot, summed in the order of dl_chi2_of_parts
                const int C = n_tiles / 2;
                const double* row = parts + (size_t)j * n_tiles;
                double chi2 = 0.;
                int cp = j % C;
                if (n_tiles_lds == 0) chi2 = dl_chi2_of_parts(s.part + (size_t)j * n_tiles, n_tiles);
                else if (C == 8) {   // 16 column blocks (two tracers): the whole row in one batch of reads
                    double v[16];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { v[2 * k] = row[2 * cp]; v[2 * k + 1] = row[2 * cp + 1]; cp = (cp + 1) & 7; }
#pragma unroll
                    for (int k = 0; k < 16; ++k) chi2 += v[k];
                } else for (int k0 = 0; k0 < C; k0 += 4) {   // (n_tiles is a multiple of 8)
                    double v[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] = row[2 * cp]; v[2 * k + 1] = row[2 * cp + 1]; cp = cp + 1 == C ? 0 : cp + 1; }
#pragma unroll
                    for (int k = 0; k < 8; ++k) chi2 += v[k];
                }
                // (one copy of the prior code, walked parameter by parameter: unrolled in batches with the families of dl_prior.h inlined in every copy the accept
                //  phase was 6554 instructions for the same 3.3 us)
                double ll, lpr = 0.;
                bool nan_in = false;
                int st;
#pragma nounroll
                for (int p = 0; p < P; ++p) {
                    const double x = prop[(size_t)j * P + p];
                    if (x != x) nan_in = true;
                    lpr += dl_prior_logpdf(priors + 5 * p, x);
                }
                dl_finalize_status(chi2, lpr, nan_in, ll, st);
                lp = st == 0 ? ll + lpr : -inf;          // what dl_eval_logposterior writes (samplers/base.py:185-191)
            }
            if (lp != lp) lp = -inf;                     // NaN results count as -inf (samplers/base.py:187-189)
            lp = lp + s.offset;
            const double lnpdiff = (fj + lp) - logp[i];
            if (logu < lnpdiff) {
                for (int p0 = 0; p0 < P; p0 += 8) {   // (reads of eight values together, then the writes: a read -> write loop pays the LDS latency per parameter)
                    double v[8];
                    dl_load_theta8(prop + (size_t)j * P, P, p0, v);
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (p0 + q < P) { coords[(size_t)i * P + p0 + q] = v[q]; s.coords[(size_t)i * P + p0 + q] = v[q]; }
                }
                logp[i] = lp; s.logp[i] = lp;
                (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(s.nacc) + i, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // no return value: nothing waits
            }
        }
        __syncthreads();
    }
    DL_ENS_STAMP(3)   // accept
    if (s.chain != nullptr) {
        for (int e = tid; e < nw * P; e += nthr) s.chain[e] = coords[e];
        if (s.chain_logp != nullptr)
            for (int e = tid; e < nw; e += nthr) s.chain_logp[e] = logp[e];
    }
    if (!proposing) return;
    for (int j = jp0; j < half; j += nthr) {
        int is = is0, ic = ic0;
        double zz = zz0, fac = fac0;
        if (j != jp0) dl_ens_draw_move(s, j, half, is, ic, zz, fac);
        for (int p = 0; p < P; ++p) {
            const double c = coords[(size_t)ic * P + p], x = coords[(size_t)is * P + p];
            s.prop[(size_t)j * P + p] = c - (c - x) * zz;      // q = c - (c - s) z
        }
        s.factors[j] = fac;
    }
    DL_ENS_STAMP(4)   // record + move
    if (s.stamps != nullptr && threadIdx.x == 0) atomicAdd(s.stamps + 7, 1ull);
}
#undef DL_ENS_STAMP

// The same step with the state in global memory (ensembles whose state does not fit the LDS of one CU): phases separated by barriers (memory written before a
// barrier is visible to the workgroup after it).
__global__ __launch_bounds__(DL_ENS_THREADS) void dl_ensemble_step_kernel(const DlEnsArgs s) {
#pragma clang fp contract(off)
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nw = s.nw, half = nw / 2, P = s.P;
    const double inf = __builtin_huge_val();
    if (s.half_acc >= 0) {
        for (int j = tid; j < half; j += nthr) {
            int i;
            double logu, lp;
            dl_ens_draw_accept(s, j, half, i, logu);
            if (s.part != nullptr) {
                double ll, lpr, x0[8];
                int st;
                dl_load_theta8(s.prop + (size_t)j * P, P, 0, x0);
                dl_finalize_from_chi2<2>(dl_chi2_of_parts(s.part + (size_t)j * s.n_tiles, s.n_tiles), x0, s.prop + (size_t)j * P, P, s.priors, ll, lpr, st);
                lp = st == 0 ? ll + lpr : -inf;
            } else lp = s.newlp[j];
            if (lp != lp) lp = -inf;
            lp = lp + s.offset;
            const double lnpdiff = (s.factors[j] + lp) - s.logp[i];
            if (logu < lnpdiff) {
                for (int p = 0; p < P; ++p) s.coords[(size_t)i * P + p] = s.prop[(size_t)j * P + p];
                s.logp[i] = lp;
                (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(s.nacc) + i, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
    if (s.chain != nullptr) {
        for (int e = tid; e < nw * P; e += nthr) s.chain[e] = s.coords[e];
        if (s.chain_logp != nullptr)
            for (int e = tid; e < nw; e += nthr) s.chain_logp[e] = s.logp[e];
    }
    if (s.half_prop < 0) return;
    for (int j = tid; j < half; j += nthr) {
        int is, ic;
        double zz, fac;
        dl_ens_draw_move(s, j, half, is, ic, zz, fac);
        for (int p = 0; p < P; ++p) {
            const double c = s.coords[(size_t)ic * P + p], x = s.coords[(size_t)is * P + p];
            s.prop[(size_t)j * P + p] = c - (c - x) * zz;
        }
        s.factors[j] = fac;
    }
}

// LDS bytes of the LDS-resident step: positions, log-posteriors, pending proposals, their partial chi2 (n_tiles = 0: none), prior table
size_t dl_ens_shared_bytes(int nw, int P, int n_tiles) {
    const size_t half = (size_t)(nw / 2);
    return ((size_t)nw * P + nw + ((half * P + 1) & ~(size_t)1) + half * n_tiles + (size_t)5 * P) * 8 + 16;
}

void dl_ens_launch(const DlEnsArgs& s_in, hipStream_t stream) {
    const int n_tiles = s_in.part != nullptr ? s_in.n_tiles : 0;
    DlEnsArgs s = s_in;
    s.stage_parts = n_tiles > 0 && n_tiles % 8 == 0 && dl_ens_shared_bytes(s.nw, s.P, n_tiles) <= 144 * 1024;   // (the staged rows are walked in groups of four 16-byte chunks)
    const size_t shm = dl_ens_shared_bytes(s.nw, s.P, s.stage_parts ? n_tiles : 0);
    const bool force_global = dl_options().ens_global;   // (tests: the global-memory variant on a small ensemble)
    if (shm <= 144 * 1024 && !force_global) {
        // up to 512 walkers: 512 threads (one slot per thread in either phase); beyond: 1024 threads
        if (s.nw <= 512) {
            if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)dl_ensemble_step_lds_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            hipLaunchKernelGGL((dl_ensemble_step_lds_kernel<512>), dim3(1), dim3(512), shm, stream, s);
        } else {
            if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)dl_ensemble_step_lds_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            hipLaunchKernelGGL((dl_ensemble_step_lds_kernel<1024>), dim3(1), dim3(1024), shm, stream, s);
        }
    } else hipLaunchKernelGGL(dl_ensemble_step_kernel, dim3(1), dim3(DL_ENS_THREADS), 0, stream, s);
}

}  // namespace

struct dl_ensemble {
    dl_ctx* ctx = nullptr;
    dl_comm* comm = nullptr;
    int device = 0, nw = 0, P = 0, rank = 0, world = 1;
    int64_t count = 0;            // proposals per rank and half-step (the last ranks' shares may be shorter or empty)
    double a = 2., offset = 0.;
    uint64_t seed = 0;
    long long iteration = 0;      // ensemble updates done since creation (the counter of the random number generator)
    bool have_logp = false;
    bool deferred = true;         // finish the proposals' log-posteriors inside the step kernel (single rank, contexts on the chi2 GEMM path)
    double *coords = nullptr, *logp = nullptr, *prop = nullptr, *factors = nullptr, *newlp = nullptr;
    long long* nacc = nullptr;
    // folded update (dl_ens_fold.h): ping-pong buffers of the pending half-step's inputs -- parity 0 aliases prop / factors, parity 1 and the partial chi2 are extra
    int fold = -1;                // -1: not yet decided, 0: no (three launches per half-step), 1: yes (two)
    int fold_tiles = 0;
    const double* fold_priors = nullptr;
    double *prop1 = nullptr, *factors1 = nullptr, *part2[2] = {nullptr, nullptr};
    double *coords_alt = nullptr, *logp_alt = nullptr;   // the other state buffer (the extra workgroups of a folded theory launch write it; then the roles swap)
};

extern "C" {

void dl_ensemble_destroy(dl_ensemble* ens) {
    if (!ens) return;
    (void)hipSetDevice(ens->device);
    for (void* p : {(void*)ens->coords, (void*)ens->logp, (void*)ens->prop, (void*)ens->factors, (void*)ens->newlp, (void*)ens->nacc, (void*)ens->prop1, (void*)ens->factors1,
                    (void*)ens->part2[0], (void*)ens->part2[1], (void*)ens->coords_alt, (void*)ens->logp_alt})
        if (p) (void)hipFree(p);
    delete ens;
}

int dl_ensemble_create(dl_ensemble** out, dl_ctx* ctx, int32_t nwalkers, double a, uint64_t seed, double offset, dl_comm* comm) {
    if (!out || !ctx) return fail("dl_ensemble_create: null argument");
    *out = nullptr;
    const int P = (int)dl_info(ctx, "n_params");
    if (nwalkers < 2 || nwalkers % 2 || nwalkers > 8192) return fail("dl_ensemble_create: nwalkers must be even, in [2, 8192]");
    if (!(a > 1.)) return fail("dl_ensemble_create: stretch parameter a must be > 1");
    dl_ensemble* ens = new dl_ensemble();
    ens->ctx = ctx; ens->comm = comm; ens->nw = nwalkers; ens->P = P; ens->a = a; ens->seed = seed; ens->offset = offset;
    ens->device = (int)dl_info(ctx, "device");
    if (comm) {
        ens->rank = (int)dl_comm_info(comm, "rank"); ens->world = (int)dl_comm_info(comm, "world");
        if ((int)dl_comm_info(comm, "device") != ens->device) { delete ens; return fail("dl_ensemble_create: communicator and context live on different devices"); }
    }
    const int half = nwalkers / 2;
    ens->count = (half + ens->world - 1) / ens->world;
    const size_t half_pad = (size_t)ens->count * ens->world;
    auto bail = [&](const std::string& msg) { dl_ensemble_destroy(ens); return fail(msg); };
    if (hipSetDevice(ens->device) != hipSuccess) return bail("dl_ensemble_create: hipSetDevice failed");
    if (hipMalloc((void**)&ens->coords, (size_t)nwalkers * P * sizeof(double)) != hipSuccess || hipMalloc((void**)&ens->logp, (size_t)nwalkers * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&ens->prop, half_pad * P * sizeof(double)) != hipSuccess || hipMalloc((void**)&ens->factors, half_pad * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&ens->newlp, half_pad * sizeof(double)) != hipSuccess || hipMalloc((void**)&ens->nacc, (size_t)nwalkers * sizeof(long long)) != hipSuccess)
        return bail("dl_ensemble_create: device allocation failed");
    if (hipMemset(ens->nacc, 0, (size_t)nwalkers * sizeof(long long)) != hipSuccess || hipMemset(ens->prop, 0, half_pad * P * sizeof(double)) != hipSuccess ||
        hipMemset(ens->newlp, 0, half_pad * sizeof(double)) != hipSuccess)
        return bail("dl_ensemble_create: hipMemset failed");
    if (hipDeviceSynchronize() != hipSuccess) return bail("dl_ensemble_create: hipDeviceSynchronize failed");   // (null-stream memsets vs the caller's non-blocking streams)
    *out = ens;
    return 0;
}

// Evaluate this rank's share of ``n`` rows of ``theta_dev`` into ``out_dev[n_pad]`` and all-gather (count rows per rank)
static int dl_ens_logposterior(dl_ensemble* ens, const double* theta_dev, int n, double* out_dev, hipStream_t stream) {
    const int64_t lo = std::min<int64_t>((int64_t)ens->rank * ens->count, n), hi = std::min<int64_t>(lo + ens->count, n);
    if (hi > lo && dl_eval_logposterior(ens->ctx, theta_dev + (size_t)lo * ens->P, hi - lo, out_dev + lo, nullptr, stream)) return 1;
    // (DL_ENS_FORCE_COMM=1: the collective is issued even with a single rank -- smoke test of the in-stream RCCL call on a 1-GPU box)
    static const bool force = getenv("DL_ENS_FORCE_COMM") != nullptr;
    if (ens->comm && (ens->world > 1 || force))
        return dl_comm_allgather_f64(ens->comm, out_dev + (size_t)ens->rank * ens->count, out_dev, ens->count, stream);
    return 0;
}

int dl_ensemble_set_state(dl_ensemble* ens, const double* coords, const double* logposterior, void* hip_stream) {
    if (!ens || !coords) return fail("dl_ensemble_set_state: null argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_ENS_HIP(hipSetDevice(ens->device));
    DL_ENS_HIP(hipMemcpyAsync(ens->coords, coords, (size_t)ens->nw * ens->P * sizeof(double), hipMemcpyHostToDevice, stream));
    if (logposterior) DL_ENS_HIP(hipMemcpyAsync(ens->logp, logposterior, (size_t)ens->nw * sizeof(double), hipMemcpyHostToDevice, stream));
    DL_ENS_HIP(hipStreamSynchronize(stream));   // the host buffers may be pageable
    ens->have_logp = logposterior != nullptr;
    return 0;
}

int dl_ensemble_run(dl_ensemble* ens, int64_t niterations, int32_t thin_by, double* chain_dev, double* chain_logp_dev, void* hip_stream) {
    if (!ens) return fail("dl_ensemble_run: null ensemble");
    if (niterations < 0 || thin_by < 1) return fail("dl_ensemble_run: invalid argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_ENS_HIP(hipSetDevice(ens->device));
    const int nw = ens->nw, P = ens->P, half = nw / 2;
    if (!ens->have_logp) {
        // log-posterior of the starting positions: two half-ensemble batches through the same sharded path
        for (int h = 0; h < 2; ++h) {
            if (dl_ens_logposterior(ens, ens->coords + (size_t)h * half * P, half, ens->newlp, stream)) return 1;
            DL_ENS_HIP(hipMemcpyAsync(ens->logp + (size_t)h * half, ens->newlp, (size_t)half * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
        if (ens->offset != 0.) {
            // rare path (marginalised posteriors with a constant): add the offset on the host
            std::vector<double> tmp(nw);
            DL_ENS_HIP(hipMemcpyAsync(tmp.data(), ens->logp, (size_t)nw * sizeof(double), hipMemcpyDeviceToHost, stream));
            DL_ENS_HIP(hipStreamSynchronize(stream));
            for (double& v : tmp) v = (v != v ? -__builtin_huge_val() : v) + ens->offset;
            DL_ENS_HIP(hipMemcpyAsync(ens->logp, tmp.data(), (size_t)nw * sizeof(double), hipMemcpyHostToDevice, stream));
            DL_ENS_HIP(hipStreamSynchronize(stream));
        }
        ens->have_logp = true;
    }
    if (niterations == 0) return 0;
    DlEnsArgs s;
    std::memset(&s, 0, sizeof(s));
    static unsigned long long* stamps_dev = nullptr;   // DL_ENS_STAMPS=1: per-phase times of the step kernel, printed at the end of the run (synchronises)
    const bool stamps_on = dl_options().ens_stamps;
    if (stamps_on && !stamps_dev) DL_ENS_HIP(hipMalloc((void**)&stamps_dev, 8 * sizeof(unsigned long long)));
    if (stamps_on) DL_ENS_HIP(hipMemsetAsync(stamps_dev, 0, 8 * sizeof(unsigned long long), stream));
    s.stamps = stamps_on ? stamps_dev : nullptr;
    s.coords = ens->coords; s.logp = ens->logp; s.nacc = ens->nacc; s.prop = ens->prop; s.factors = ens->factors; s.newlp = ens->newlp;
    s.nw = nw; s.P = P; s.a = ens->a; s.offset = ens->offset;
    s.k0 = (uint32_t)ens->seed; s.k1 = (uint32_t)(ens->seed >> 32);
    s.half_acc = -1;
    const long long it0 = ens->iteration;
    // the launch that proposes half-step (it, h) also accepts the half-step before it; the accept of half-step 1 completes an iteration: recorded there
    auto set_record = [&]() {
        s.chain = nullptr; s.chain_logp = nullptr;
        if (s.half_acc != 1 || !chain_dev) return;
        const long long done = s.it_acc - it0 + 1;
        if (done % thin_by) return;
        const size_t row = (size_t)(done / thin_by - 1);
        s.chain = chain_dev + row * nw * P;
        s.chain_logp = chain_logp_dev ? chain_logp_dev + row * nw : nullptr;
    };
    // Folded update: decided once per ensemble (single rank, plain likelihood on the chi2-GEMM path with fast full-shape kernels); DL_ENS_NO_FOLD=1 keeps the
    // three-launch sequence (comparison, tests)
    if (ens->fold < 0) {
        ens->fold = 0;
        const bool comm_on = ens->comm && (ens->world > 1 || dl_options().ens_force_comm);
        if (ens->deferred && !comm_on && !dl_options().ens_no_defer && !dl_options().ens_no_fold && !dl_options().ens_global &&
            dl_internal_fold_info(ens->ctx, half, &ens->fold_tiles, &ens->fold_priors) == 0) {
            const size_t half_pad = (size_t)ens->count * ens->world;
            if (hipMalloc((void**)&ens->prop1, half_pad * P * sizeof(double)) == hipSuccess && hipMalloc((void**)&ens->factors1, half_pad * sizeof(double)) == hipSuccess &&
                hipMalloc((void**)&ens->part2[0], half_pad * ens->fold_tiles * sizeof(double)) == hipSuccess &&
                hipMalloc((void**)&ens->part2[1], half_pad * ens->fold_tiles * sizeof(double)) == hipSuccess &&
                hipMalloc((void**)&ens->coords_alt, (size_t)nw * P * sizeof(double)) == hipSuccess && hipMalloc((void**)&ens->logp_alt, (size_t)nw * sizeof(double)) == hipSuccess)
                ens->fold = 1;
        }
    }
    bool folded = ens->fold == 1;
    DlEnsFold f;
    std::memset(&f, 0, sizeof(f));
    static unsigned long long* fold_stamps_dev = nullptr;   // DL_ENS_FOLD_STAMPS=1: phase times of the theory kernel's proposal prologue, printed at the end of the run (synchronises)
    const bool fold_stamps = folded && dl_options().ens_fold_stamps;
    if (fold_stamps && !fold_stamps_dev) DL_ENS_HIP(hipMalloc((void**)&fold_stamps_dev, (size_t)8192 * 8 * sizeof(unsigned long long)));
    if (fold_stamps) f.stamps = fold_stamps_dev;
    if (folded) {
        f.nacc = ens->nacc; f.priors = ens->fold_priors; f.a = ens->a; f.offset = ens->offset;
        f.nw = nw; f.P = P; f.n_tiles = ens->fold_tiles; f.k0 = s.k0; f.k1 = s.k1;
        f.pend.half = -1;
    }
    for (long long it = it0; it < it0 + niterations; ++it)
        for (int h = 0; h < 2; ++h) {
            if (folded) {
                // half-step (it, h): its proposals / factors / partial chi2 go to the buffers of parity h; the pending half-step's sit in parity 1 - h
                double* prop_h = h ? ens->prop1 : ens->prop; double* fac_h = h ? ens->factors1 : ens->factors;
                if (h == 0) s.split_prop = dl_ens_split(it, nw, s.k0, s.k1);
                f.it_prop = it; f.half_prop = h; f.split_prop = s.split_prop; f.prop_out = prop_h; f.factors_out = fac_h;
                set_record();     // (record target of the accept that is pending: written by this half-step's chi2 GEMM)
                f.chain = s.chain; f.chain_logp = s.chain_logp;
                f.coords = ens->coords; f.logp = ens->logp; f.coords_out = ens->coords_alt; f.logp_out = ens->logp_alt;
                const int rc = dl_internal_eval_fold(ens->ctx, f, half, ens->part2[h], stream);
                if (rc == 1) return 1;
                if (rc == 2) {
                    if (f.pend.half >= 0) return fail("dl_ensemble_run: the context left the folded path in the middle of a run");
                    ens->fold = 0; folded = false;      // (first half-step of the first run: nothing pending yet -- fall through to the three-launch sequence)
                } else {
                    if (f.pend.half >= 0) { std::swap(ens->coords, ens->coords_alt); std::swap(ens->logp, ens->logp_alt); }   // the launch wrote the state after the pending accepts
                    f.pend.prop = prop_h; f.pend.factors = fac_h; f.pend.part = ens->part2[h]; f.pend.it = it; f.pend.half = h; f.pend.split = s.split_prop;
                    s.it_acc = it; s.half_acc = h; s.split_acc = s.split_prop;
                    continue;
                }
            }
            s.it_prop = it; s.half_prop = h;
            if (h == 0) s.split_prop = dl_ens_split(it, nw, s.k0, s.k1);
            set_record();
            dl_ens_launch(s, stream);
            s.part = nullptr;
            if (ens->deferred && !(ens->comm && (ens->world > 1 || dl_options().ens_force_comm)) && !dl_options().ens_no_defer) {
                int rc = dl_internal_eval_partials(ens->ctx, ens->prop, half, &s.part, &s.n_tiles, &s.priors, stream);
                if (rc == 1) return 1;
                if (rc == 2) { ens->deferred = false; s.part = nullptr; }
            }
            if (s.part == nullptr && dl_ens_logposterior(ens, ens->prop, half, ens->newlp, stream)) return 1;
            s.it_acc = it; s.half_acc = h; s.split_acc = s.split_prop;
        }
    if (folded && f.pend.half >= 0) {
        // the last half-step's accept: the step kernel (accept + record, nothing to propose) on the pending buffers
        s.prop = const_cast<double*>(f.pend.prop); s.factors = const_cast<double*>(f.pend.factors); s.part = f.pend.part; s.n_tiles = ens->fold_tiles; s.priors = ens->fold_priors;
        s.coords = ens->coords; s.logp = ens->logp;
    }
    s.half_prop = -1;
    set_record();
    dl_ens_launch(s, stream);
    DL_ENS_HIP(hipGetLastError());
    if (stamps_on) {
        unsigned long long h[8];
        DL_ENS_HIP(hipMemcpyAsync(h, stamps_dev, sizeof(h), hipMemcpyDeviceToHost, stream));
        DL_ENS_HIP(hipStreamSynchronize(stream));
        const double n = h[7] ? 100. * (double)h[7] : 1.;
        fprintf(stderr, "dl_ensemble_run: step kernel, us per launch over %llu launches: requests + landing %.2f, draws %.2f, barrier %.2f, accept %.2f, record + move %.2f\n", h[7], h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n);
    }
    if (fold_stamps) {
        std::vector<unsigned long long> h((size_t)half * 16);
        DL_ENS_HIP(hipMemcpyAsync(h.data(), fold_stamps_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        DL_ENS_HIP(hipStreamSynchronize(stream));
        double acc[6] = {0., 0., 0., 0., 0., 0.};
        for (int b = 0; b < half; ++b) for (int q = 0; q < 6; ++q) acc[q] += (double)(h[(size_t)b * 8 + q + 1] - h[(size_t)b * 8 + q]);
        fprintf(stderr, "dl_ensemble_run: theory kernel of the last half-step, mean shader-clock ticks per workgroup: kernarg prefetch %.0f, draw + splits %.0f, decisions (loads + sums) %.0f, theta %.0f, barrier %.0f, phases %.0f\n",
                acc[0] / half, acc[1] / half, acc[2] / half, acc[3] / half, acc[4] / half, acc[5] / half);
    }
    ens->iteration += niterations;
    return 0;
}

int dl_ensemble_get_state(dl_ensemble* ens, double* coords, double* logposterior, int64_t* naccepted, void* hip_stream) {
    if (!ens) return fail("dl_ensemble_get_state: null ensemble");
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_ENS_HIP(hipSetDevice(ens->device));
    if (coords) DL_ENS_HIP(hipMemcpyAsync(coords, ens->coords, (size_t)ens->nw * ens->P * sizeof(double), hipMemcpyDeviceToHost, stream));
    if (logposterior) DL_ENS_HIP(hipMemcpyAsync(logposterior, ens->logp, (size_t)ens->nw * sizeof(double), hipMemcpyDeviceToHost, stream));
    if (naccepted) DL_ENS_HIP(hipMemcpyAsync(naccepted, ens->nacc, (size_t)ens->nw * sizeof(long long), hipMemcpyDeviceToHost, stream));
    DL_ENS_HIP(hipStreamSynchronize(stream));
    return 0;
}

int dl_ensemble_set_counter(dl_ensemble* ens, int64_t iteration, const int64_t* naccepted, void* hip_stream) {
    if (!ens) return fail("dl_ensemble_set_counter: null ensemble");
    if (iteration < 0) return fail("dl_ensemble_set_counter: negative iteration");
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_ENS_HIP(hipSetDevice(ens->device));
    if (naccepted) {
        DL_ENS_HIP(hipMemcpyAsync(ens->nacc, naccepted, (size_t)ens->nw * sizeof(long long), hipMemcpyHostToDevice, stream));
        DL_ENS_HIP(hipStreamSynchronize(stream));   // the host buffer may be pageable
    }
    ens->iteration = iteration;
    return 0;
}

int64_t dl_ensemble_info(const dl_ensemble* ens, const char* key) {
    if (!ens || !key) return -1;
    std::string k(key);
    if (k == "nwalkers") return ens->nw;
    if (k == "n_params") return ens->P;
    if (k == "iteration") return ens->iteration;
    if (k == "rank") return ens->rank;
    if (k == "world") return ens->world;
    if (k == "rows_per_rank") return ens->count;
    return -1;
}

}  // extern "C"
