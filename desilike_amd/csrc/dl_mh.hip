// dl_mh.hip -- device-resident blocked Metropolis-Hastings sampler (include/desilike_amd.h, dl_mh_*).
//
// The reference's own MCMC (desilike/samplers/mcmc.py: MHSampler 25-127 + BlockProposer 199-328, the CosmoMC / cobaya blocked proposal) advances ONE chain per
// group of MPI ranks and uses the ranks of the group speculatively: ``vectorize`` proposals are drawn from the current state, their log-posteriors evaluated in
// parallel, and the first one that passes the Metropolis test is taken, the rejected ones before it adding to the weight of the current state (mcmc.py:94-105).
// Here C chains x V speculative proposals are ONE batch of C V rows of dl_eval_logposterior; positions, log-posteriors, weights, the accepted samples and the
// random draws (dl_mh.h: pure functions of (seed, chain, call counter)) live on the device, and a try is two launches with no host synchronisation:
//
//     [Metropolis scan of the previous try | record the state that is left | V new proposals per chain]  ->  dl_eval_logposterior(C V rows)  ->  ...
//
// One wavefront per (chain, proposal slot), a lane per parameter; the wavefronts of a chain repeat the (cheap) Metropolis scan instead of synchronising.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/desilike_amd.h"
#include "dl_kernels.h"
#include "dl_mh.h"
#include "dl_finalize_part.h"

struct dl_mh {
    dl_ctx* ctx = nullptr;
    int device = 0;
    int C = 0, V = 0, P = 0, nblocks = 0, n_rep = 0;
    double scale = 2.4, offset = 0.;
    uint64_t seed = 0;
    int64_t max_tries = 1000, tries = 0;
    bool have_logp = false, have_cov = false;
    bool deferred = true;    // the Metropolis scan finishes the proposals from the chi2 GEMM's partial sums (until the context declines: dl_internal_eval_partials)
    // device
    // the state (positions, log-posteriors, weights, counters) and the proposals exist twice: a launch reads one copy and writes the other, so that the wavefronts
    // of a chain (one per proposal slot, spread over the chip) need no barrier between the Metropolis scan and the new proposals
    double *coords[2] = {nullptr, nullptr}, *logp[2] = {nullptr, nullptr}, *prop[2] = {nullptr, nullptr}, *newlp = nullptr, *L = nullptr;
    long long *weight[2] = {nullptr, nullptr}, *naccepted[2] = {nullptr, nullptr};
    int32_t *fails[2] = {nullptr, nullptr}, *chain_ids = nullptr, *order = nullptr, *rep_block = nullptr, *block_start = nullptr, *block_reps = nullptr;
    int cur = 0, cur_prop = 0;
    // record buffers of dl_mh_run_host (host-pointer variant for FFI callers), grown on demand
    double *rec_coords = nullptr, *rec_logp = nullptr;
    long long* rec_weight = nullptr;
    int32_t* rec_count = nullptr;
    int64_t rec_cap = 0;
};

namespace {

int fail(const std::string& msg) {
    dl_set_last_error(msg.c_str());
    return 1;
}

#define DL_MH_HIP(call)                                                                               \
    do {                                                                                              \
        hipError_t err__ = (call);                                                                    \
        if (err__ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(err__));   \
    } while (0)

struct DlMhArgs {
    const double *coords, *logp, *prop;  // state before this launch, pending proposals
    double *coords_out, *logp_out, *prop_out;
    const double *newlp, *L;
    const double *part, *priors;         // deferred finalize (plain likelihoods on the chi2-GEMM path): partial chi2 [C V, n_tiles] of the pending proposals straight from
    int32_t n_tiles, pad_;               // the chi2 GEMM, prior table [P, 5] -- the scan sums them and applies the status rules itself (no finalize launch); else null
    const long long *weight, *naccepted;
    long long *weight_out, *naccepted_out;
    const int32_t* fails;
    int32_t* fails_out;
    const int32_t *chain_ids, *order, *rep_block, *block_start, *block_reps;
    double *out_coords, *out_logp;       // records of this run: [C, cap, P], [C, cap]
    long long* out_weight;               // [C, cap]
    int32_t* out_count;                  // [C]
    int32_t C, V, P, n_rep, cap, thin_by;
    double scale, offset;
    uint32_t k0, k1;
    long long try_acc, try_prop;         // try whose proposals are pending / to draw; -1: none
    long long max_tries;
};

__device__ __forceinline__ double dl_mh_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// direction of proposer call: column j of the Haar rotation m of block ib (b parameters), lane i holds component i (lanes >= b: 0).  Householder reflections of
// Gaussian vectors applied to the unit vector e_j, last reflection first, then the signs D (Stewart 1980 as in scipy.stats.special_ortho_group; mcmc.py:170-173).
// Reflection k acts on the components >= k only: the ones with k > j leave e_j alone and contribute their sign alone.
//   wide blocks (b > DL_MH_LDS_B): every reflection draws its Gaussians when it is applied;
//   b <= DL_MH_LDS_B: all Gaussians the column needs are drawn in ONE pass over the lanes (a Philox call gives a pair) into the wavefront's LDS rows, every lane sums
//   the squares of its own row, and the loop over the reflections is one LDS read, one short reduction and two FMAs per reflection (a single wavefront runs this
//   kernel's critical path: its length is what a try pays).
#define DL_MH_LDS_B 32

__device__ __forceinline__ double dl_mh_direction_wide(int lane, int b, int j, uint64_t m, uint32_t chain, int ib, uint32_t k0, uint32_t k1) {
    double y = lane == j ? 1. : 0., dsign = 1., dprod = 1.;
    for (int k = b - 2; k >= 0; --k) {
        double x = (lane >= k && lane < b) ? dl_mh_rot_gauss(m, chain, ib, k, lane - k, k0, k1) : 0.;
        const double norm2 = dl_mh_wave_sum(x * x);
        const double x0 = __shfl(x, k, 64);
        const double dk = x0 < 0. ? -1. : 1.;
        const double x0n = x0 + dk * sqrt(norm2);
        if (lane == k) { x = x0n; dsign = dk; }
        dprod *= dk;
        const double xx = (norm2 - x0 * x0) + x0n * x0n;
        const double dot = dl_mh_wave_sum(x * y);
        y -= 2. * x * (dot / xx);
    }
    if (lane == b - 1) dsign = (((b - 1) & 1) ? -1. : 1.) * dprod;
    return y * dsign;
}

__device__ __forceinline__ double dl_mh_direction(int lane, int b, int j, uint64_t m, uint32_t chain, int ib, uint32_t k0, uint32_t k1, double* g /* LDS [DL_MH_LDS_B][DL_MH_LDS_B] of this wavefront */) {
    if (b > DL_MH_LDS_B) return dl_mh_direction_wide(lane, b, j, m, chain, ib, k0, k1);
    const int kmax = j < b - 2 ? j : b - 2, pb = (b + 1) >> 1;
    const int full = (kmax + 1) * pb, slots = full + (b - 2 - kmax);       // whole rows 0 .. kmax, the first pair of the rows above (their signs)
    for (int sl = lane; sl < slots; sl += 64) {
        const int k = sl < full ? sl / pb : kmax + 1 + (sl - full), pr = sl < full ? sl % pb : 0;
        if (2 * pr < b - k) {
            const uint32_t stream = (uint32_t)DL_MH_STREAM_ROT | ((uint32_t)ib << 8) | ((uint32_t)k << 14) | ((uint32_t)pr << 20);
            const DlPhilox r = dl_philox4x32((uint32_t)m, (uint32_t)(m >> 32), chain, stream, k0, k1);
            const double rho = sqrt(-2. * log1p(-dl_uniform53(r.x[0], r.x[1])));
            double sn, cs;
            sincospi(2. * dl_uniform53(r.x[2], r.x[3]), &sn, &cs);
            g[k * DL_MH_LDS_B + 2 * pr] = rho * cs;
            if (2 * pr + 1 < b - k) g[k * DL_MH_LDS_B + 2 * pr + 1] = rho * sn;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the rows are in LDS
    // lane k: sum of squares, sign and modified first element of row k
    double norm2 = 0., x0 = 1.;
    if (lane <= b - 2) {
        x0 = g[lane * DL_MH_LDS_B];
        if (lane <= kmax) for (int e = 0; e < b - lane; ++e) { const double t = g[lane * DL_MH_LDS_B + e]; norm2 += t * t; }
    }
    const double dk = x0 < 0. ? -1. : 1.;                                 // lanes above b - 2: + 1
    double dprod = dk;                                                     // product of the signs of all reflections
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dprod *= __shfl_xor(dprod, off, 64);
    const double x0n = x0 + dk * sqrt(norm2), xx = (norm2 - x0 * x0) + x0n * x0n;
    int width = 1;
    while (width < b) width <<= 1;
    double y = lane == j ? 1. : 0.;
    for (int k = kmax; k >= 0; --k) {
        double x = (lane > k && lane < b) ? g[k * DL_MH_LDS_B + lane - k] : 0.;
        const double x0k = __shfl(x0n, k, 64), xxk = __shfl(xx, k, 64);
        if (lane == k) x = x0k;
        double dot = x * y;
        for (int off = width >> 1; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, 64);   // (the lanes beyond the block hold zeros)
        y -= 2. * x * (dot / xxk);
    }
    const double dsign = lane == b - 1 ? (((b - 1) & 1) ? -1. : 1.) * dprod : dk;
    return y * dsign;
}

#define DL_MH_WAVES 4   // wavefronts per workgroup: independent (chain, slot) pairs

__global__ __launch_bounds__(64 * DL_MH_WAVES) void dl_mh_step_kernel(const DlMhArgs s) {
    __shared__ double gauss[DL_MH_WAVES][DL_MH_LDS_B * DL_MH_LDS_B];
    const int lane = threadIdx.x & 63;
    const int P = s.P, V = s.V;
    const long long pair = (long long)blockIdx.x * DL_MH_WAVES + (threadIdx.x >> 6);
    if (pair >= (long long)s.C * V) return;
    const int c = (int)(pair / V), v = (int)(pair % V);
    const uint32_t chain = (uint32_t)s.chain_ids[c];
    // The jump of this wavefront's slot does not depend on the chain's state: it is formed FIRST, in the shadow of the loads the Metropolis scan waits for (state,
    // log-posterior, the pending proposals' results) -- the kernel is one wavefront's critical path, and the dependent round trips are what it is made of.
    const double* __restrict__ logp_in = s.logp;
    const double* __restrict__ coords_in = s.coords;
    const double cur_lp = logp_in[c];
    double state = lane < P ? coords_in[(size_t)c * P + lane] : 0.;   // lane = parameter (context order)
    double lp = 0.;
    if (s.try_acc >= 0 && lane < V && s.part == nullptr) lp = s.newlp[(size_t)c * V + lane];
    const int sorted = lane < P ? lane : 0;                             // (every lane takes part in the exchanges below; the lanes beyond P write nothing)
    const int target = s.order[sorted];                                 // sorted index i is parameter order[i] of the context
    double delta = 0.;
    int start = 0;
    if (s.try_prop >= 0) {
        const uint64_t n = (uint64_t)s.try_prop * V + v;
        const uint64_t q = n / (uint64_t)s.n_rep;
        const uint32_t p = (uint32_t)(n % (uint64_t)s.n_rep);
        const DlMhKeys keys = dl_mh_perm_keys(q, chain, s.k0, s.k1);
        const int ib = s.rep_block[dl_mh_perm_at(keys, p, (uint32_t)s.n_rep)];
        int cnt = 0;                                                    // earlier calls of this cycle that went to the same block
        for (uint32_t p0 = 0; p0 < p; p0 += 64) {
            const uint32_t pp = p0 + lane;
            const bool same = pp < p && s.rep_block[dl_mh_perm_at(keys, pp, (uint32_t)s.n_rep)] == ib;
            cnt += __popcll(__ballot(same));
        }
        start = s.block_start[ib];
        const int b = s.block_start[ib + 1] - start;
        const uint64_t calls = q * (uint64_t)s.block_reps[ib] + (uint64_t)cnt;
        double sign = 1.;
        const double radius = dl_mh_radial(n, chain, b, s.k0, s.k1, &sign);
        double y;
        if (b == 1) y = lane == 0 ? sign : 0.;                          // mcmc.py:165-166
        else y = dl_mh_direction(lane, b, (int)(calls % (uint64_t)b), calls / (uint64_t)b, chain, ib, s.k0, s.k1, gauss[threadIdx.x >> 6]);
        y *= radius * s.scale;
        // jump of the sorted parameters start .. P - 1: L[start:, start : start + b] . y (mcmc.py:290-296, 315-327); lane i = sorted index start + i
        const int i = start + lane;
        for (int jj = 0; jj < b; ++jj) {
            const double yj = __shfl(y, jj, 64);
            if (i < P) delta += s.L[(size_t)i * P + start + jj] * yj;
        }
    }
    // ---- Metropolis scan of the pending try: every wavefront of the chain takes the same decision from the same data; the wavefront of slot 0 writes it ------------
    double new_lp = cur_lp;
    int first = -1;
    if (s.try_acc >= 0) {
        bool acc = false;
        if (lane < V) {
            if (s.part != nullptr) {
                const double* row = s.prop + ((size_t)c * V + lane) * P;
                double ll, lpr, x0[8];
                int st;
                dl_load_theta8(row, P, 0, x0);
                dl_finalize_from_chi2<2>(dl_chi2_of_parts(s.part + ((size_t)c * V + lane) * s.n_tiles, s.n_tiles), x0, row, P, s.priors, ll, lpr, st);
                lp = st == 0 ? ll + lpr : -__builtin_huge_val();
            }
            lp = (lp != lp ? -__builtin_huge_val() : lp) + s.offset;                                    // samplers/base.py:187-189
            const double e = dl_mh_accept_exp((uint64_t)s.try_acc * V + lane, chain, s.k0, s.k1);
            acc = lp > -__builtin_huge_val() && (lp > cur_lp || e > cur_lp - lp);                       // mcmc.py:107-112
        }
        const unsigned long long mask = __ballot(acc);
        if (mask) {
            first = __ffsll((long long)mask) - 1;
            new_lp = __shfl(lp, first, 64);
        }
    }
    const double old_state = state;
    if (first >= 0 && lane < P) state = s.prop[((size_t)c * V + first) * P + lane];
    if (v == 0) {
        long long weight = s.weight[c], iter = s.naccepted[c];
        int fails = s.fails[c];
        if (first >= 0) {
            if (iter > 0 && iter % s.thin_by == 0) {                    // the state that is left is recorded with its final weight; the start is skipped (mcmc.py:97-99)
                const int slot = s.out_count[c];
                if (slot < s.cap) {
                    if (lane < P) s.out_coords[((size_t)c * s.cap + slot) * P + lane] = old_state;
                    if (lane == 0) { s.out_logp[(size_t)c * s.cap + slot] = cur_lp; s.out_weight[(size_t)c * s.cap + slot] = weight + first; s.out_count[c] = slot + 1; }
                }
            }
            weight = 1; iter += 1; fails = 0;
        } else if (s.try_acc >= 0) {
            weight += V; fails += 1;
        }
        if (lane < P) s.coords_out[(size_t)c * P + lane] = state;
        if (lane == 0) { s.logp_out[c] = new_lp; s.weight_out[c] = weight; s.naccepted_out[c] = iter; s.fails_out[c] = fails; }
    }
    if (s.try_prop < 0) return;
    // ---- the proposal of slot v: the (new) state + the jump, scattered into context order -------------------------------------------------------------------------
    double* row = s.prop_out + ((size_t)c * V + v) * P;
    const double base = __shfl(state, target, 64);
    const double jump = __shfl(delta, sorted >= start ? sorted - start : 0, 64);
    if (lane < P) row[target] = base + (sorted >= start ? jump : 0.);   // the parameters of slower blocks keep their values
}

}  // namespace

extern "C" {

void dl_mh_destroy(dl_mh* mh) {
    if (!mh) return;
    (void)hipSetDevice(mh->device);
    for (void* p : {(void*)mh->coords[0], (void*)mh->coords[1], (void*)mh->logp[0], (void*)mh->logp[1], (void*)mh->prop[0], (void*)mh->prop[1], (void*)mh->newlp, (void*)mh->L,
                    (void*)mh->weight[0], (void*)mh->weight[1], (void*)mh->naccepted[0], (void*)mh->naccepted[1], (void*)mh->fails[0], (void*)mh->fails[1],
                    (void*)mh->chain_ids, (void*)mh->order, (void*)mh->rep_block, (void*)mh->block_start, (void*)mh->block_reps, (void*)mh->rec_coords, (void*)mh->rec_logp,
                    (void*)mh->rec_weight, (void*)mh->rec_count})
        if (p) (void)hipFree(p);
    delete mh;
}

int dl_mh_create(dl_mh** out, dl_ctx* ctx, int32_t nchains, int32_t vectorize, const int32_t* chain_ids, const int32_t* order, const int32_t* blocks,
                 const int32_t* oversample, int32_t nblocks, double proposal_scale, uint64_t seed, double offset, int64_t max_tries) {
    if (!out || !ctx || !blocks || nblocks < 1) return fail("dl_mh_create: null argument");
    *out = nullptr;
    const int P = (int)dl_info(ctx, "n_params");
    if (P < 1 || P > DL_MH_MAX_P) return fail("dl_mh_create: the sampler takes 1 .. 64 parameters");
    if (nchains < 1 || vectorize < 1 || vectorize > DL_MH_MAX_V) return fail("dl_mh_create: nchains >= 1 and 1 <= vectorize <= 64");
    if (!(proposal_scale > 0.) || max_tries < 1) return fail("dl_mh_create: proposal_scale and max_tries must be positive");
    std::vector<int32_t> start(nblocks + 1, 0), reps(nblocks), rep_block;
    for (int ib = 0; ib < nblocks; ++ib) {
        const int o = oversample ? oversample[ib] : 1;
        if (blocks[ib] < 1 || o < 1) return fail("dl_mh_create: block sizes and oversampling factors must be >= 1");
        start[ib + 1] = start[ib] + blocks[ib];
        reps[ib] = blocks[ib] * o;
        rep_block.insert(rep_block.end(), (size_t)reps[ib], ib);      // mcmc.py:254: every parameter index of the block repeated `o` times
    }
    if (start[nblocks] != P) return fail("dl_mh_create: the blocks must add up to the number of parameters");
    if ((int)rep_block.size() > DL_MH_MAX_REP || nblocks > 64) return fail("dl_mh_create: too many blocks / cycler entries");
    std::vector<int32_t> ord(P), ids(nchains);
    std::vector<char> seen(P, 0);
    for (int i = 0; i < P; ++i) {
        ord[i] = order ? order[i] : i;
        if (ord[i] < 0 || ord[i] >= P || seen[ord[i]]) return fail("dl_mh_create: order must be a permutation of the parameters");
        seen[ord[i]] = 1;
    }
    for (int c = 0; c < nchains; ++c) ids[c] = chain_ids ? chain_ids[c] : c;
    dl_mh* mh = new dl_mh();
    mh->ctx = ctx; mh->device = (int)dl_info(ctx, "device"); mh->C = nchains; mh->V = vectorize; mh->P = P; mh->nblocks = nblocks; mh->n_rep = (int)rep_block.size();
    mh->scale = proposal_scale; mh->seed = seed; mh->offset = offset; mh->max_tries = max_tries;
    mh->deferred = getenv("DL_MH_NO_DEFER") == nullptr;
    auto bail = [&](const std::string& msg) { dl_mh_destroy(mh); return fail(msg); };
    if (hipSetDevice(mh->device) != hipSuccess) return bail("dl_mh_create: hipSetDevice failed");
    const size_t C = nchains, V = vectorize;
    bool ok = true;
    for (int d = 0; d < 2; ++d)
        ok = ok && hipMalloc((void**)&mh->coords[d], C * P * sizeof(double)) == hipSuccess && hipMalloc((void**)&mh->logp[d], C * sizeof(double)) == hipSuccess &&
             hipMalloc((void**)&mh->prop[d], C * V * P * sizeof(double)) == hipSuccess && hipMalloc((void**)&mh->weight[d], C * sizeof(long long)) == hipSuccess &&
             hipMalloc((void**)&mh->naccepted[d], C * sizeof(long long)) == hipSuccess && hipMalloc((void**)&mh->fails[d], C * sizeof(int32_t)) == hipSuccess &&
             hipMemset(mh->prop[d], 0, C * V * P * sizeof(double)) == hipSuccess && hipMemset(mh->naccepted[d], 0, C * sizeof(long long)) == hipSuccess &&
             hipMemset(mh->fails[d], 0, C * sizeof(int32_t)) == hipSuccess;
    ok = ok && hipMalloc((void**)&mh->newlp, C * V * sizeof(double)) == hipSuccess && hipMalloc((void**)&mh->L, (size_t)P * P * sizeof(double)) == hipSuccess &&
              hipMalloc((void**)&mh->chain_ids, C * sizeof(int32_t)) == hipSuccess && hipMalloc((void**)&mh->order, (size_t)P * sizeof(int32_t)) == hipSuccess &&
              hipMalloc((void**)&mh->rep_block, rep_block.size() * sizeof(int32_t)) == hipSuccess &&
              hipMalloc((void**)&mh->block_start, start.size() * sizeof(int32_t)) == hipSuccess && hipMalloc((void**)&mh->block_reps, reps.size() * sizeof(int32_t)) == hipSuccess;
    if (!ok) return bail("dl_mh_create: device allocation failed");
    ok = hipMemset(mh->newlp, 0, C * V * sizeof(double)) == hipSuccess &&
         hipMemset(mh->L, 0, (size_t)P * P * sizeof(double)) == hipSuccess &&
         hipMemcpy(mh->chain_ids, ids.data(), C * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(mh->order, ord.data(), (size_t)P * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(mh->rep_block, rep_block.data(), rep_block.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(mh->block_start, start.data(), start.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(mh->block_reps, reps.data(), reps.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) return bail("dl_mh_create: initialisation of the device arrays failed");
    if (hipDeviceSynchronize() != hipSuccess) return bail("dl_mh_create: hipDeviceSynchronize failed");
    *out = mh;
    return 0;
}

int dl_mh_set_covariance(dl_mh* mh, const double* cholesky, void* hip_stream) {
    if (!mh || !cholesky) return fail("dl_mh_set_covariance: null argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    const int P = mh->P;
    for (int i = 0; i < P; ++i) {
        if (!(cholesky[(size_t)i * P + i] > 0.)) return fail("dl_mh_set_covariance: the Cholesky factor must have a positive diagonal");
        for (int j = 0; j < P; ++j) {
            const double v = cholesky[(size_t)i * P + j];
            if (v != v || (j > i && v != 0.)) return fail("dl_mh_set_covariance: expected a finite lower-triangular factor [P, P] (sorted parameter order)");
        }
    }
    DL_MH_HIP(hipSetDevice(mh->device));
    DL_MH_HIP(hipMemcpyAsync(mh->L, cholesky, (size_t)P * P * sizeof(double), hipMemcpyHostToDevice, stream));
    DL_MH_HIP(hipStreamSynchronize(stream));   // the host buffer may be pageable
    mh->have_cov = true;
    return 0;
}

int dl_mh_set_state(dl_mh* mh, const double* coords, const double* logposterior, const int64_t* weight, const int64_t* naccepted, int64_t tries, void* hip_stream) {
    if (!mh || !coords) return fail("dl_mh_set_state: null argument");
    if (tries < 0) return fail("dl_mh_set_state: negative try counter");
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_MH_HIP(hipSetDevice(mh->device));
    const size_t C = mh->C;
    std::vector<long long> w(C, 1), na(C, 0);
    if (weight) for (size_t c = 0; c < C; ++c) w[c] = weight[c];
    if (naccepted) for (size_t c = 0; c < C; ++c) na[c] = naccepted[c];
    const int d = mh->cur;
    DL_MH_HIP(hipMemcpyAsync(mh->coords[d], coords, C * mh->P * sizeof(double), hipMemcpyHostToDevice, stream));
    if (logposterior) DL_MH_HIP(hipMemcpyAsync(mh->logp[d], logposterior, C * sizeof(double), hipMemcpyHostToDevice, stream));
    DL_MH_HIP(hipMemcpyAsync(mh->weight[d], w.data(), C * sizeof(long long), hipMemcpyHostToDevice, stream));
    DL_MH_HIP(hipMemcpyAsync(mh->naccepted[d], na.data(), C * sizeof(long long), hipMemcpyHostToDevice, stream));
    DL_MH_HIP(hipMemsetAsync(mh->fails[d], 0, C * sizeof(int32_t), stream));
    DL_MH_HIP(hipStreamSynchronize(stream));
    mh->have_logp = logposterior != nullptr;
    mh->tries = tries;
    return 0;
}

int dl_mh_run(dl_mh* mh, int64_t ntries, int32_t thin_by, double* out_coords_dev, double* out_logp_dev, int64_t* out_weight_dev, int32_t* out_count_dev, void* hip_stream) {
    if (!mh) return fail("dl_mh_run: null sampler");
    if (ntries < 0 || thin_by < 1) return fail("dl_mh_run: invalid argument");
    if (ntries > 0 && (!out_coords_dev || !out_logp_dev || !out_weight_dev || !out_count_dev)) return fail("dl_mh_run: the record buffers are required");
    if (!mh->have_cov) return fail("dl_mh_run: no proposal covariance (dl_mh_set_covariance)");
    if (ntries > 0x7fffffff) return fail("dl_mh_run: at most 2^31 - 1 tries per call");
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_MH_HIP(hipSetDevice(mh->device));
    const int C = mh->C, V = mh->V, P = mh->P;
    if (!mh->have_logp) {
        if (dl_eval_logposterior(mh->ctx, mh->coords[mh->cur], C, mh->logp[mh->cur], nullptr, stream)) return 1;
        std::vector<double> tmp(C);
        DL_MH_HIP(hipMemcpyAsync(tmp.data(), mh->logp[mh->cur], (size_t)C * sizeof(double), hipMemcpyDeviceToHost, stream));
        DL_MH_HIP(hipStreamSynchronize(stream));
        for (double& v : tmp) {
            v = (v != v ? -__builtin_huge_val() : v) + mh->offset;
            if (!(v > -__builtin_huge_val())) return fail("dl_mh_run: the log-posterior of a starting position is not finite");
        }
        DL_MH_HIP(hipMemcpyAsync(mh->logp[mh->cur], tmp.data(), (size_t)C * sizeof(double), hipMemcpyHostToDevice, stream));
        DL_MH_HIP(hipStreamSynchronize(stream));
        mh->have_logp = true;
    }
    if (ntries == 0) return 0;
    DL_MH_HIP(hipMemsetAsync(out_count_dev, 0, (size_t)C * sizeof(int32_t), stream));
    DlMhArgs s;
    std::memset(&s, 0, sizeof(s));
    s.newlp = mh->newlp; s.L = mh->L;
    s.chain_ids = mh->chain_ids; s.order = mh->order; s.rep_block = mh->rep_block; s.block_start = mh->block_start; s.block_reps = mh->block_reps;
    s.out_coords = out_coords_dev; s.out_logp = out_logp_dev; s.out_weight = reinterpret_cast<long long*>(out_weight_dev); s.out_count = out_count_dev;
    s.C = C; s.V = V; s.P = P; s.n_rep = mh->n_rep; s.cap = (int32_t)ntries; s.thin_by = thin_by;
    s.scale = mh->scale; s.offset = mh->offset; s.k0 = (uint32_t)mh->seed; s.k1 = (uint32_t)(mh->seed >> 32);
    s.max_tries = mh->max_tries;
    const unsigned grid = (unsigned)(((long long)C * V + DL_MH_WAVES - 1) / DL_MH_WAVES);
    auto launch = [&]() {   // reads the current copy of the state and the pending proposals, writes the other copies
        const int d = mh->cur, e = mh->cur_prop;
        s.coords = mh->coords[d]; s.logp = mh->logp[d]; s.weight = mh->weight[d]; s.naccepted = mh->naccepted[d]; s.fails = mh->fails[d]; s.prop = mh->prop[e];
        s.coords_out = mh->coords[1 - d]; s.logp_out = mh->logp[1 - d]; s.weight_out = mh->weight[1 - d]; s.naccepted_out = mh->naccepted[1 - d]; s.fails_out = mh->fails[1 - d];
        s.prop_out = mh->prop[1 - e];
        hipLaunchKernelGGL(dl_mh_step_kernel, dim3(grid), dim3(64 * DL_MH_WAVES), 0, stream, s);
        mh->cur = 1 - d; mh->cur_prop = 1 - e;
    };
    s.try_acc = -1;
    for (int64_t t = mh->tries; t < mh->tries + ntries; ++t) {
        s.try_prop = t;
        launch();
        s.part = nullptr;
        if (mh->deferred) {
            const int rc = dl_internal_eval_partials(mh->ctx, mh->prop[mh->cur_prop], (int64_t)C * V, &s.part, &s.n_tiles, &s.priors, stream);
            if (rc == 1) return 1;
            if (rc == 2) { mh->deferred = false; s.part = nullptr; }
        }
        if (s.part == nullptr && dl_eval_logposterior(mh->ctx, mh->prop[mh->cur_prop], (int64_t)C * V, mh->newlp, nullptr, stream)) return 1;
        s.try_acc = t;
    }
    s.try_prop = -1;
    launch();
    DL_MH_HIP(hipGetLastError());
    mh->tries += ntries;
    return 0;
}

int dl_mh_run_host(dl_mh* mh, int64_t ntries, int32_t thin_by, double* out_coords, double* out_logp, int64_t* out_weight, int32_t* out_count, void* hip_stream) {
    if (!mh) return fail("dl_mh_run_host: null sampler");
    if (ntries < 0 || (ntries > 0 && (!out_coords || !out_logp || !out_weight || !out_count))) return fail("dl_mh_run_host: invalid argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_MH_HIP(hipSetDevice(mh->device));
    const size_t C = mh->C, P = mh->P;
    if (ntries > mh->rec_cap) {
        DL_MH_HIP(hipStreamSynchronize(stream));
        for (void* p : {(void*)mh->rec_coords, (void*)mh->rec_logp, (void*)mh->rec_weight, (void*)mh->rec_count}) if (p) (void)hipFree(p);
        mh->rec_coords = mh->rec_logp = nullptr; mh->rec_weight = nullptr; mh->rec_count = nullptr; mh->rec_cap = 0;
        if (hipMalloc((void**)&mh->rec_coords, C * ntries * P * sizeof(double)) != hipSuccess || hipMalloc((void**)&mh->rec_logp, C * ntries * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&mh->rec_weight, C * ntries * sizeof(long long)) != hipSuccess || hipMalloc((void**)&mh->rec_count, C * sizeof(int32_t)) != hipSuccess)
            return fail("dl_mh_run_host: device allocation failed");
        mh->rec_cap = ntries;
    }
    if (ntries == 0) return dl_mh_run(mh, 0, thin_by, nullptr, nullptr, nullptr, nullptr, hip_stream);
    if (dl_mh_run(mh, ntries, thin_by, mh->rec_coords, mh->rec_logp, reinterpret_cast<int64_t*>(mh->rec_weight), mh->rec_count, hip_stream)) return 1;
    DL_MH_HIP(hipMemcpyAsync(out_coords, mh->rec_coords, C * ntries * P * sizeof(double), hipMemcpyDeviceToHost, stream));
    DL_MH_HIP(hipMemcpyAsync(out_logp, mh->rec_logp, C * ntries * sizeof(double), hipMemcpyDeviceToHost, stream));
    DL_MH_HIP(hipMemcpyAsync(out_weight, mh->rec_weight, C * ntries * sizeof(long long), hipMemcpyDeviceToHost, stream));
    DL_MH_HIP(hipMemcpyAsync(out_count, mh->rec_count, C * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    DL_MH_HIP(hipStreamSynchronize(stream));
    return 0;
}

int dl_mh_get_state(dl_mh* mh, double* coords, double* logposterior, int64_t* weight, int64_t* naccepted, int32_t* fails, void* hip_stream) {
    if (!mh) return fail("dl_mh_get_state: null sampler");
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_MH_HIP(hipSetDevice(mh->device));
    const size_t C = mh->C;
    const int d = mh->cur;
    if (coords) DL_MH_HIP(hipMemcpyAsync(coords, mh->coords[d], C * mh->P * sizeof(double), hipMemcpyDeviceToHost, stream));
    if (logposterior) DL_MH_HIP(hipMemcpyAsync(logposterior, mh->logp[d], C * sizeof(double), hipMemcpyDeviceToHost, stream));
    if (weight) DL_MH_HIP(hipMemcpyAsync(weight, mh->weight[d], C * sizeof(long long), hipMemcpyDeviceToHost, stream));
    if (naccepted) DL_MH_HIP(hipMemcpyAsync(naccepted, mh->naccepted[d], C * sizeof(long long), hipMemcpyDeviceToHost, stream));
    if (fails) DL_MH_HIP(hipMemcpyAsync(fails, mh->fails[d], C * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    DL_MH_HIP(hipStreamSynchronize(stream));
    return 0;
}

int64_t dl_mh_info(const dl_mh* mh, const char* key) {
    if (!mh || !key) return -1;
    const std::string k(key);
    if (k == "nchains") return mh->C;
    if (k == "vectorize") return mh->V;
    if (k == "n_params") return mh->P;
    if (k == "tries") return mh->tries;
    if (k == "cycle") return mh->n_rep;
    if (k == "max_tries") return mh->max_tries;
    return -1;
}

}  // extern "C"
