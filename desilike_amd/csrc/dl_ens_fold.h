// dl_ens_fold.h -- the ensemble sampler's random draws and accept decisions as PURE functions of (seed, iteration, half-step, slot), shared by the step kernel
// (dl_ensemble.hip), the theory kernel's proposal prologue (dl_kernels.hip: dl_fullshape_ens_kernel) and the chi2 GEMM's accept prologue (dl_chi2_gemm.h).
//
// Folded update (single rank, plain likelihood on the chi2-GEMM path): a half-step is TWO launches instead of three -- the theory launch and the chi2 GEMM;
// the step launch is gone from the critical path:
//
//   * every workgroup of the theory kernel first DERIVES its own proposal: the stretch move of slot j needs the current positions of two walkers (its own and the
//     partner from the complementary half); each is either final in the state buffer the launch reads or waits for the accept decision of half-step t - 1, which the
//     workgroup re-evaluates itself from immutable inputs (proposals, partial chi2, stretch factors of t - 1, the walker's old log-posterior, the counter-based
//     uniform): no cross-workgroup dependency;
//   * a few EXTRA workgroups of the same launch write the state after the accepts of t - 1 -- positions, log-posteriors, accepted counts, the chain record -- into the
//     OTHER state buffer (ping-pong: nothing in the launch reads what they write; the next launch reads it).  They need ~3 us and run beside theory workgroups that
//     live ~9 us: off the critical path.  (In the chi2 GEMM's column-block-0 workgroups they cost the launch 3.5 us: every CU holds one GEMM workgroup, the slowest
//     sets the kernel's end.)
//
// The same decision code runs in the step kernel's successor paths; arithmetic is the NumPy driver's (samplers.py EnsembleStretchMove with CounterRNG), operation
// by operation (fp contraction off): the chain is bit-identical whichever path produced it (tests/test_gpu_sampler.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dl_finalize_part.h"

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

enum { DL_ENS_STREAM_PERM = 0, DL_ENS_STREAM_MOVE = 1, DL_ENS_STREAM_ACCEPT = 3 };   // + half-step for the last two

// Random split of the ensemble into two halves: position r holds walker F(r), F a keyed bijection of [0, nw) -- four rounds of (odd multiplier, offset) mod 2^m
// and a right xor-shift, m = bit length of nw - 1, keyed by eight Philox words of the iteration, cycle-walked back into [0, nw).  Every element is a pure function
// of (seed, iteration, r): nothing is ranked, stored or exchanged; the keys of a launch are made on the host and travel in the kernel arguments.
// samplers.py CounterRNG.permutation is the NumPy statement of the same map.  F is inverted round by round (the xor-shift by >= m / 2 bits is an involution, the
// multipliers are odd: their inverses mod 2^32 are made on the host), cycle-walking the inverse: position of walker w = F^-1(w).
struct DlEnsSplit {
    uint32_t mul[4], add[4], inv[4], mask, shift, nw;
};

inline DlEnsSplit dl_ens_split(long long iteration, int nw, uint32_t k0, uint32_t k1) {
    const DlPhilox ka = dl_philox4x32((uint32_t)iteration, (uint32_t)((unsigned long long)iteration >> 32), 0u, DL_ENS_STREAM_PERM, k0, k1);
    const DlPhilox kb = dl_philox4x32((uint32_t)iteration, (uint32_t)((unsigned long long)iteration >> 32), 1u, DL_ENS_STREAM_PERM, k0, k1);
    DlEnsSplit f;
    int m = 1;
    while (m < 31 && (1u << m) < (uint32_t)nw) ++m;
    for (int round = 0; round < 4; ++round) {
        f.mul[round] = ka.x[round] | 1u; f.add[round] = kb.x[round];
        uint32_t inv = f.mul[round];                      // Newton: x <- x (2 - a x) doubles the number of correct low bits (3 to start with: a a = 1 mod 8)
        for (int it = 0; it < 5; ++it) inv *= 2u - f.mul[round] * inv;
        f.inv[round] = inv;
    }
    f.mask = (1u << m) - 1u; f.shift = (uint32_t)(m + 1) / 2; f.nw = (uint32_t)nw;
    return f;
}

__device__ __forceinline__ int dl_ens_split_at(const DlEnsSplit& f, int r) {
    uint32_t x = (uint32_t)r;
    do {
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            x = (x * f.mul[round] + f.add[round]) & f.mask;
            x ^= x >> f.shift;
        }
    } while (x >= f.nw);
    return (int)x;
}

// position r with dl_ens_split_at(f, r) == w
__device__ __forceinline__ int dl_ens_split_inv(const DlEnsSplit& f, int w) {
    uint32_t x = (uint32_t)w;
    do {
#pragma unroll
        for (int round = 3; round >= 0; --round) {
            x ^= x >> f.shift;
            x = ((x - f.add[round]) * f.inv[round]) & f.mask;
        }
    } while (x >= f.nw);
    return (int)x;
}

// What decides the accepts of the pending half-step (immutable while the launches that read it run)
struct DlEnsPending {
    const double* prop;      // [half, P] its proposals
    const double* factors;   // [half] (P - 1) log z
    const double* part;      // [half, n_tiles] partial chi2 of the proposals (chi2 GEMM)
    long long it;            // its iteration
    int32_t half;            // its half-step (0 / 1); -1: nothing pending
    int32_t pad_;
    DlEnsSplit split;        // the split of its iteration
};

struct DlEnsFold {
    const double* coords;    // [nw, P]  state read by this launch: final for every walker that is not in the pending half
    const double* logp;      // [nw]
    double* coords_out;      // [nw, P]  state after the pending accepts, written by the extra workgroups (the next launch reads it)
    double* logp_out;        // [nw]
    long long* nacc;         // [nw]
    const double* priors;    // [P, 5]
    double a, offset;
    int32_t nw, P, n_tiles, pad_;
    uint32_t k0, k1;
    DlEnsPending pend;
    // theory kernel: the half-step to propose
    long long it_prop;
    int32_t half_prop, pad2_;
    DlEnsSplit split_prop;
    double* prop_out;        // [half, P]
    double* factors_out;     // [half]
    // record target of the accepts this launch writes (null: no record)
    double* chain;           // [nw, P]
    double* chain_logp;      // [nw]
    unsigned long long* stamps;   // DL_ENS_FOLD_STAMPS diagnostics (null in production): 8 x s_memtime per theory workgroup of observable 0
};

__device__ __forceinline__ double dl_ens_readlane(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// TWO accept decisions per wavefront, lane-parallel: decision A in lanes 0-31, decision B in lanes 32-63 (s < 0: nothing to decide).  Lane h of a half loads the
// partial chi2 of column block h, the proposal's parameter h and its prior row (every load of the wave is requested before the first wait: one round trip) and
// evaluates the prior term.  The sums must be those of the scalar code (dl_chi2_of_parts, dl_finalize_from_chi2: term after term, in order) bit for bit: the values go
// through 1 KB of LDS (`scratch`, private to the wave) and FOUR lanes sum one sequence each -- chi2 and log-prior of A and of B -- side by side, 8 values per batch of
// reads; the four results are then made uniform by readlane.  (Summing by readlane, one term at a time with a run-time lane index, took 100 cycles per term:
// 2 us of the theory kernel's prologue.)  P, n_tiles <= 32.
// (emcee moves/red_blue.py: lnpdiff = factors + new_log_prob - log_prob; accepted = log(u) < lnpdiff; samplers/base.py:185-191 for the status rules)
struct DlEnsDecision2 {
    bool acc[2];
    double lp[2];
};

#define DL_ENS_SCRATCH 128   // doubles of LDS per deciding wavefront

__device__ __forceinline__ DlEnsDecision2 dl_ens_decide2(const DlEnsPending& pd, const double* __restrict__ priors, int P, int n_tiles, double offset, uint32_t k0, uint32_t k1,
                                                         int sA, double logpA, int sB, double logpB, double* __restrict__ scratch) {
#pragma clang fp contract(off)
    const double inf = __builtin_huge_val();
    const int lane = threadIdx.x & 63, hl = lane & 31, hb = lane >> 5;
    const int s = hb ? sB : sA;
    const int ss = s >= 0 ? s : 0;
    // one round of loads
    const double part = pd.part[(size_t)ss * n_tiles + (hl < n_tiles ? hl : n_tiles - 1)];
    const int p = hl < P ? hl : P - 1;
    const double x = pd.prop[(size_t)ss * P + p];
    const double* pr = priors + 5 * p;
    const double pr0 = pr[0], pr1 = pr[1], pr2 = pr[2], pr3 = pr[3], pr4 = pr[4];
    const double fj = pd.factors[ss];
    // in the shadow of the round trip: the accept draw
    const DlPhilox r = dl_philox4x32((uint32_t)pd.it, (uint32_t)((unsigned long long)pd.it >> 32), (uint32_t)ss, DL_ENS_STREAM_ACCEPT + pd.half, k0, k1);
    const double logu = log(dl_uniform53(r.x[0], r.x[1]));
    const double row[5] = {pr0, pr1, pr2, pr3, pr4};
    const double term = dl_prior_logpdf(row, x);
    const unsigned long long nan_mask = __ballot(x != x && hl < P);
    // sequences in LDS: [0, 32) partial chi2 of A, [32, 64) prior terms of A, [64, 96) partial chi2 of B, [96, 128) prior terms of B
    scratch[64 * hb + hl] = part;
    scratch[64 * hb + 32 + hl] = term;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double sum = 0.;
    {
        const int q = lane & 3;                                   // lanes 0-3 (every group of four does the same: no divergence)
        const int len = (q & 1) ? P : n_tiles;
        const double* seq = scratch + 32 * q;
        for (int t0 = 0; t0 < 32; t0 += 8) {
            if (t0 >= n_tiles && t0 >= P) break;
            double v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = seq[t0 + t];
#pragma unroll
            for (int t = 0; t < 8; ++t) sum = t0 + t < len ? sum + v[t] : sum;
        }
    }
    DlEnsDecision2 d;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double chi2 = dl_ens_readlane(sum, 2 * h), lpr = dl_ens_readlane(sum, 2 * h + 1);
        const bool nan_in = ((nan_mask >> (32 * h)) & 0xffffffffull) != 0ull;
        double ll;
        int st;
        dl_finalize_status(chi2, lpr, nan_in, ll, st);
        double lp = st == 0 ? ll + lpr : -inf;
        if (lp != lp) lp = -inf;
        lp = lp + offset;
        const double lnpdiff = (dl_ens_readlane(fj, 32 * h) + lp) - (h ? logpB : logpA);
        d.lp[h] = lp;
        d.acc[h] = (h ? sB : sA) >= 0 && dl_ens_readlane(logu, 32 * h) < lnpdiff;
    }
    return d;
}

// slot of walker w in the pending half-step, or -1
__device__ __forceinline__ int dl_ens_pending_slot(const DlEnsPending& pd, int w, int half) {
    if (pd.half < 0) return -1;
    const int pos = dl_ens_split_inv(pd.split, w);
    const int s = pos - pd.half * half;
    return (s >= 0 && s < half) ? s : -1;
}
