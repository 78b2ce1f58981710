// dl_ensemble.hip -- device-resident affine-invariant ensemble sampler (include/desilike_amd.h, dl_ensemble_*).
//
// What the reference does on the host per ensemble update (desilike/samplers/emcee.py:69-111 -> emcee.EnsembleSampler(vectorize=True) with its default
// StretchMove: Goodman & Weare 2010, two half-ensemble updates; the log-posterior of a half is ONE batched call, desilike/samplers/base.py:144-200) runs here as one
// enqueued sequence per half-step with no host synchronisation:
//
//     [accept previous half | permutation | stretch proposals]  ->  dl_eval_logposterior (this rank's share)  ->  ncclAllGather (N > 1)  ->  next ...
//
// Walker positions, log-posteriors, acceptance counts and the random number generator live on the device; the host only enqueues and, at the end of a run,
// drains the chain.  Random numbers are COUNTER-BASED (Philox4x32-10, Salmon et al. 2011): the draw for (iteration, half-step, walker slot) is a pure
// function of the seed, so every rank of a sharded run holds the same ensemble without exchanging anything but log-posteriors, and the NumPy
// ``EnsembleStretchMove`` of the host package reproduces the chain bit for bit with the same generator (tests/test_gpu_sampler.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/desilike_amd.h"
#include "dl_kernels.h"
#include "dl_finalize_part.h"

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

struct DlPhilox {
    uint32_t x[4];
};

// Philox4x32-10 (Random123): counter c[4], key k[2]
__host__ __device__ inline DlPhilox dl_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return DlPhilox{{c0, c1, c2, c3}};
}

// 53-bit uniform on [0, 1) from two 32-bit words (the construction of numpy's random_sample)
__host__ __device__ inline double dl_uniform53(uint32_t hi, uint32_t lo) { return ((double)(hi >> 5) * 67108864. + (double)(lo >> 6)) * (1. / 9007199254740992.); }

#define DL_ENS_PRE 4           // results of the pending half-step prefetched per thread (covers nwalkers <= 8 DL_ENS_THREADS)
#define DL_ENS_THREADS 1024   // one workgroup: ranking the 64-bit keys of the random split is O(nwalkers^2 / threads) LDS reads per thread
enum { DL_ENS_STREAM_PERM = 0, DL_ENS_STREAM_MOVE = 1, DL_ENS_STREAM_ACCEPT = 3 };   // + half-step for the last two

struct DlEnsArgs {
    double* coords;        // [nw, P]
    double* logp;          // [nw]
    long long* nacc;       // [nw]
    int32_t* perm;         // [nw]: first half = walkers updated in half-step 0, second half = half-step 1
    double* prop;          // [half_pad, P] proposals of the pending half-step
    double* factors;       // [half] (P - 1) log z
    double* newlp;         // [half_pad] log-posteriors of the proposals (all ranks' shares after the all-gather)
    double* chain;         // record target of this launch: [nw, P] or null
    double* chain_logp;    // [nw] or null
    const double* part;    // deferred finalize (single rank, plain likelihood): partial chi2 [half, n_tiles] of the pending proposals straight from the chi2 GEMM,
    const double* priors;  // ... prior table [P, 5]: this kernel sums the partials, adds the priors and applies the status rules itself (no finalize launch)
    int32_t n_tiles;
    int32_t nw, P;
    double a, offset;
    uint32_t k0, k1;
    long long it_acc, it_prop;   // iteration of the half-step to accept / to propose
    int32_t half_acc, half_prop; // 0 / 1, or -1: nothing to accept / propose
};

// One workgroup: the ensemble is a few hundred walkers x <= 64 parameters.  STAGED: positions, log-posteriors and the split live in LDS for the duration of the
// launch (one coalesced load at the top, one write-back at the end): the dependent global round trips of the phases (split -> positions of a walker and of its
// partner -> proposal) become LDS accesses (profiles/r02a: 15.5 us per launch on average, 30 us for the launches that draw a split, with the state in global memory).
// Phases are separated by barriers (memory written before a barrier is visible to the workgroup after it).
template <bool STAGED>
__global__ __launch_bounds__(DL_ENS_THREADS) void dl_ensemble_step_kernel(const DlEnsArgs s) {
#pragma clang fp contract(off)   // the NumPy driver rounds after every operation: no fused multiply-adds here
    extern __shared__ __attribute__((aligned(16))) unsigned long long dl_ens_lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nw = s.nw, half = nw / 2, P = s.P;
    const int parts = (DL_ENS_THREADS / nw) > 1 ? DL_ENS_THREADS / nw : 1;      // threads that share the ranking of one walker
    uint32_t* keys = reinterpret_cast<uint32_t*>(dl_ens_lds);                     // [nw] sort keys of the split (the slot is sized for 8 bytes per walker)
    int* rankpart = reinterpret_cast<int*>(dl_ens_lds + nw);                      // [parts, nw]
    double* lds_state = reinterpret_cast<double*>(rankpart + (size_t)((parts * nw + 1) & ~1));
    double* coords = STAGED ? lds_state : s.coords;                               // [nw, P]
    double* logp = STAGED ? lds_state + (size_t)nw * P : s.logp;                  // [nw]
    int32_t* perm = STAGED ? reinterpret_cast<int32_t*>(lds_state + (size_t)nw * (P + 1)) : s.perm;   // [nw]
    const double inf = __builtin_huge_val();
    // the pending half-step's results are requested first: their round trip overlaps the staging of the state
    double pre_lp[DL_ENS_PRE], pre_f[DL_ENS_PRE];
    if (s.half_acc >= 0) {
#pragma unroll
        for (int q = 0; q < DL_ENS_PRE; ++q) {
            const int j = tid + q * nthr;
            // (deferred finalize: the partial chi2 of the slot are summed now -- their round trip overlaps the staging; pre_lp then holds chi2)
            pre_lp[q] = j >= half ? 0. : (s.part == nullptr ? s.newlp[j] : dl_chi2_of_parts(s.part + (size_t)j * s.n_tiles, s.n_tiles));
            pre_f[q] = j < half ? s.factors[j] : 0.;
        }
    }
    if (STAGED) {
        for (int e = tid; e < nw * P; e += nthr) coords[e] = s.coords[e];
        for (int e = tid; e < nw; e += nthr) { logp[e] = s.logp[e]; perm[e] = s.perm[e]; }
        __syncthreads();
    }
    if (s.half_acc >= 0) {
        // accept / reject the pending proposals (emcee moves/red_blue.py: lnpdiff = factors + new_log_prob - log_prob; accepted = log(u) < lnpdiff)
        const int32_t* set = perm + s.half_acc * half;
        int q = 0;
        for (int j = tid; j < half; j += nthr, ++q) {
            const int i = set[j];
            const DlPhilox r = dl_philox4x32((uint32_t)s.it_acc, (uint32_t)((unsigned long long)s.it_acc >> 32), (uint32_t)j, DL_ENS_STREAM_ACCEPT + s.half_acc, s.k0, s.k1);
            const double u = dl_uniform53(r.x[0], r.x[1]);
            double lp;
            if (s.part != nullptr) {
                double ll, lpr;
                int st;
                const double chi2 = q < DL_ENS_PRE ? pre_lp[q] : dl_chi2_of_parts(s.part + (size_t)j * s.n_tiles, s.n_tiles);
                dl_finalize_from_chi2(chi2, s.prop + (size_t)j * P, P, s.priors, ll, lpr, st);
                lp = st == 0 ? ll + lpr : -inf;          // what dl_eval_logposterior writes (samplers/base.py:185-191)
            } else lp = q < DL_ENS_PRE ? pre_lp[q] : s.newlp[j];
            const double fj = q < DL_ENS_PRE ? pre_f[q] : s.factors[j];
            if (lp != lp) lp = -inf;                     // NaN results count as -inf (samplers/base.py:187-189)
            lp = lp + s.offset;
            const double lnpdiff = (fj + lp) - logp[i];
            const bool accepted = log(u) < lnpdiff;
            if (accepted) {
                for (int p = 0; p < P; ++p) coords[(size_t)i * P + p] = s.prop[(size_t)j * P + p];
                logp[i] = lp;
                s.nacc[i] += 1;
            }
        }
        __syncthreads();
        if (STAGED) {   // write-back of the state the accept step changed
            for (int e = tid; e < nw * P; e += nthr) s.coords[e] = coords[e];
            for (int e = tid; e < nw; e += nthr) s.logp[e] = logp[e];
        }
    }
    if (s.chain != nullptr) {
        for (int e = tid; e < nw * P; e += nthr) s.chain[e] = coords[e];
        if (s.chain_logp != nullptr)
            for (int e = tid; e < nw; e += nthr) s.chain_logp[e] = logp[e];
    }
    if (s.half_prop < 0) return;
    if (s.half_prop == 0) {
        // random split of the ensemble into two halves: walkers ranked by a 32-bit key each = 19 random bits above the walker index (13 bits: distinct keys,
        // ties of the random part fall back on the index) -- numpy: argsort((x0 & ~0x1fff) | i).  One compare + one add per pair: the O(nwalkers^2) ranking runs
        // on ONE CU (a 64-bit key with a separate tie rule cost 8.5 us of a 30 us launch at 512 walkers, profiles/r02b)
        for (int i = tid; i < nw; i += nthr) {
            const DlPhilox r = dl_philox4x32((uint32_t)s.it_prop, (uint32_t)((unsigned long long)s.it_prop >> 32), (uint32_t)i, DL_ENS_STREAM_PERM, s.k0, s.k1);
            keys[i] = (r.x[0] & ~0x1fffu) | (uint32_t)i;
        }
        __syncthreads();
        // rank of walker i = number of keys below its own: `parts` threads share the scan of one walker (nw <= DL_ENS_THREADS), partial counts meet in LDS
        const int span = (nw + parts - 1) / parts;
        for (int t = tid; t < parts * nw; t += nthr) {
            const int i = t % nw, part = t / nw;
            const uint32_t ki = keys[i];
            const int j0 = part * span, j1 = (j0 + span < nw) ? j0 + span : nw;
            int rank = 0;
#pragma unroll 16
            for (int j = j0; j < j1; ++j) rank += (int)(keys[j] < ki);
            rankpart[part * nw + i] = rank;
        }
        __syncthreads();
        for (int i = tid; i < nw; i += nthr) {
            int rank = 0;
            for (int part = 0; part < parts; ++part) rank += rankpart[part * nw + i];
            perm[rank] = i;
            if (STAGED) s.perm[rank] = i;
        }
        __syncthreads();
    }
    {
        // stretch move (emcee moves/stretch.py): z ~ g(z) on [1/a, a], partner drawn from the complementary half, q = c - (c - s) z
        const int32_t* set = perm + s.half_prop * half;
        const int32_t* comp = perm + (1 - s.half_prop) * half;
        for (int j = tid; j < half; j += nthr) {
            const DlPhilox r = dl_philox4x32((uint32_t)s.it_prop, (uint32_t)((unsigned long long)s.it_prop >> 32), (uint32_t)j, DL_ENS_STREAM_MOVE + s.half_prop, s.k0, s.k1);
            const double u = dl_uniform53(r.x[0], r.x[1]);
            const double t = (s.a - 1.) * u + 1.;
            const double zz = (t * t) / s.a;
            const int ic = comp[r.x[2] % (uint32_t)half], is = set[j];
            for (int p = 0; p < P; ++p) {
                const double c = coords[(size_t)ic * P + p], x = coords[(size_t)is * P + p];
                s.prop[(size_t)j * P + p] = c - (c - x) * zz;
            }
            s.factors[j] = (P - 1.) * log(zz);
        }
    }
}

// LDS bytes of a launch; staged = false: keys and partial ranks only
size_t dl_ens_shared_bytes(int nw, int P, bool staged) {
    const int parts = (DL_ENS_THREADS / nw) > 1 ? DL_ENS_THREADS / nw : 1;
    size_t bytes = (size_t)nw * 8 + (size_t)((parts * nw + 1) & ~1) * 4;
    if (staged) bytes += (size_t)nw * (P + 1) * 8 + (size_t)nw * 4;
    return (bytes + 15) & ~(size_t)15;
}

void dl_ens_launch(const DlEnsArgs& s, hipStream_t stream) {
    const bool staged = dl_ens_shared_bytes(s.nw, s.P, true) <= 144 * 1024;
    const size_t shm = dl_ens_shared_bytes(s.nw, s.P, staged);
    if (staged) {
        if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)dl_ensemble_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(dl_ensemble_step_kernel<true>, dim3(1), dim3(DL_ENS_THREADS), shm, stream, s);
    } else {
        if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)dl_ensemble_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(dl_ensemble_step_kernel<false>, dim3(1), dim3(DL_ENS_THREADS), shm, stream, s);
    }
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
    int32_t* perm = nullptr;
};

extern "C" {

void dl_ensemble_destroy(dl_ensemble* ens) {
    if (!ens) return;
    (void)hipSetDevice(ens->device);
    for (void* p : {(void*)ens->coords, (void*)ens->logp, (void*)ens->prop, (void*)ens->factors, (void*)ens->newlp, (void*)ens->nacc, (void*)ens->perm})
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
        hipMalloc((void**)&ens->newlp, half_pad * sizeof(double)) != hipSuccess || hipMalloc((void**)&ens->nacc, (size_t)nwalkers * sizeof(long long)) != hipSuccess ||
        hipMalloc((void**)&ens->perm, (size_t)nwalkers * sizeof(int32_t)) != hipSuccess)
        return bail("dl_ensemble_create: device allocation failed");
    if (hipMemset(ens->nacc, 0, (size_t)nwalkers * sizeof(long long)) != hipSuccess || hipMemset(ens->prop, 0, half_pad * P * sizeof(double)) != hipSuccess ||
        hipMemset(ens->newlp, 0, half_pad * sizeof(double)) != hipSuccess)
        return bail("dl_ensemble_create: hipMemset failed");
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
    s.coords = ens->coords; s.logp = ens->logp; s.nacc = ens->nacc; s.perm = ens->perm; s.prop = ens->prop; s.factors = ens->factors; s.newlp = ens->newlp;
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
    for (long long it = it0; it < it0 + niterations; ++it)
        for (int h = 0; h < 2; ++h) {
            s.it_prop = it; s.half_prop = h;
            set_record();
            dl_ens_launch(s, stream);
            s.part = nullptr;
            if (ens->deferred && !(ens->comm && (ens->world > 1 || getenv("DL_ENS_FORCE_COMM"))) && !getenv("DL_ENS_NO_DEFER")) {
                int rc = dl_internal_eval_partials(ens->ctx, ens->prop, half, &s.part, &s.n_tiles, &s.priors, stream);
                if (rc == 1) return 1;
                if (rc == 2) { ens->deferred = false; s.part = nullptr; }
            }
            if (s.part == nullptr && dl_ens_logposterior(ens, ens->prop, half, ens->newlp, stream)) return 1;
            s.it_acc = it; s.half_acc = h;
        }
    s.half_prop = -1;
    set_record();
    dl_ens_launch(s, stream);
    DL_ENS_HIP(hipGetLastError());
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
