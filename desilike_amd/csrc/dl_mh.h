// dl_mh.h -- the random draws of the blocked Metropolis-Hastings sampler (dl_mh.hip; reference: desilike/samplers/mcmc.py) as PURE FUNCTIONS of
// (seed, chain, counter): Philox4x32-10 words, nothing stored between launches.  The reference keeps a numpy RandomState and three pieces of sequential state
// (the cycler's permutation, every block's rotation matrix and loop index: mcmc.py:138-183); here the same quantities are functions of the index n of the proposer
// call (n = try * vectorize + slot), so a chain is reproduced from (seed, chain id, position, try counter) alone, on any number of ranks:
//
//   cycler (mcmc.py:150-155)      cycle q = n / n_rep, position p = n % n_rep; the cycle's permutation is a keyed bijection of [0, n_rep) (identity for n_rep <= 2)
//   block loop index (170-172)    calls made to block ib before n = q * b * o + #{p' < p in the same cycle that fall in ib}; loop index = calls % b, rotation m = calls / b
//   rotation (172: special_ortho_group.rvs)   Haar rotation as a product of b - 1 Householder reflections of fresh Gaussian vectors (Stewart 1980; Mezzadri 2007 --
//                                 the construction scipy's special_ortho_group uses), every Gaussian a function of (chain, ib, m, reflection, element)
//   radial draw (176-183)         0.33 : exponential, else sqrt(chi2(min(b, 2)))
//   Metropolis test (107-112)     standard exponential > current - proposal
//
// oracle/np_oracle.py (mh_* with the Philox draw source) is the NumPy statement of the same functions.
#pragma once
#include <math.h>
#include <stdint.h>

#include "dl_ens_fold.h"   // dl_philox4x32, dl_uniform53

enum { DL_MH_STREAM_PERM_A = 16, DL_MH_STREAM_PERM_B = 17, DL_MH_STREAM_RADIAL = 18, DL_MH_STREAM_RADIAL2 = 19, DL_MH_STREAM_ACCEPT = 20, DL_MH_STREAM_ROT = 21 };

#define DL_MH_MAX_P 64        // parameters (one lane each)
#define DL_MH_MAX_V 64        // speculative proposals per try (one lane each in the Metropolis scan)
#define DL_MH_MAX_REP 1024    // entries of the cycler (sum over blocks of size x oversampling)

struct DlMhKeys {   // the eight words that key the permutation of one cycle
    uint32_t a[4], b[4];
};

__host__ __device__ inline DlMhKeys dl_mh_perm_keys(uint64_t cycle, uint32_t chain, uint32_t k0, uint32_t k1) {
    const DlPhilox ka = dl_philox4x32((uint32_t)cycle, (uint32_t)(cycle >> 32), chain, DL_MH_STREAM_PERM_A, k0, k1);
    const DlPhilox kb = dl_philox4x32((uint32_t)cycle, (uint32_t)(cycle >> 32), chain, DL_MH_STREAM_PERM_B, k0, k1);
    DlMhKeys keys;
    for (int r = 0; r < 4; ++r) { keys.a[r] = ka.x[r] | 1u; keys.b[r] = kb.x[r]; }
    return keys;
}

// entry p of the permutation of [0, n): four rounds of (odd multiplier, offset) mod 2^bits and a right xor-shift, cycle-walked back into [0, n)
__host__ __device__ inline uint32_t dl_mh_perm_at(const DlMhKeys& keys, uint32_t p, uint32_t n) {
    if (n <= 2u) return p;                    // mcmc.py:146-147, 153: two or fewer entries alternate in order
    uint32_t bits = 1;
    while ((1u << bits) < n) ++bits;
    const uint32_t mask = (1u << bits) - 1u, shift = (bits + 1u) / 2u;
    uint32_t y = p;
    do {
        for (int r = 0; r < 4; ++r) { y = (y * keys.a[r] + keys.b[r]) & mask; y ^= y >> shift; }
    } while (y >= n);
    return y;
}

// standard exponential of the Metropolis test of proposer call n
__host__ __device__ inline double dl_mh_accept_exp(uint64_t n, uint32_t chain, uint32_t k0, uint32_t k1) {
    const DlPhilox r = dl_philox4x32((uint32_t)n, (uint32_t)(n >> 32), chain, DL_MH_STREAM_ACCEPT, k0, k1);
    return -log1p(-dl_uniform53(r.x[0], r.x[1]));
}

// radial scale of proposer call n for a block of b parameters, and the sign used by one-parameter blocks (mcmc.py:165-166, 176-183)
__device__ inline double dl_mh_radial(uint64_t n, uint32_t chain, int b, uint32_t k0, uint32_t k1, double* sign) {
    const DlPhilox r = dl_philox4x32((uint32_t)n, (uint32_t)(n >> 32), chain, DL_MH_STREAM_RADIAL, k0, k1);
    const double mix = dl_uniform53(r.x[0], r.x[1]);
    const double e = -log1p(-dl_uniform53(r.x[2], r.x[3]));
    double radius;
    if (b >= 2) radius = mix < 0.33 ? e : sqrt(2. * e);                                     // chi2(2) = 2 x exponential
    else {
        const DlPhilox r2 = dl_philox4x32((uint32_t)n, (uint32_t)(n >> 32), chain, DL_MH_STREAM_RADIAL2, k0, k1);
        const double g = sqrt(2. * e) * cospi(2. * dl_uniform53(r2.x[0], r2.x[1]));   // chi2(1) = (standard normal)^2
        radius = mix < 0.33 ? e : fabs(g);
        if (sign) *sign = (r2.x[2] & 1u) ? 1. : -1.;
    }
    return radius;
}

// Gaussian number `element` of reflection `refl` of rotation m of block ib
__device__ inline double dl_mh_rot_gauss(uint64_t m, uint32_t chain, int ib, int refl, int element, uint32_t k0, uint32_t k1) {
    const uint32_t stream = (uint32_t)DL_MH_STREAM_ROT | ((uint32_t)ib << 8) | ((uint32_t)refl << 14) | ((uint32_t)(element >> 1) << 20);
    const DlPhilox r = dl_philox4x32((uint32_t)m, (uint32_t)(m >> 32), chain, stream, k0, k1);
    const double rho = sqrt(-2. * log1p(-dl_uniform53(r.x[0], r.x[1]))), phi2 = 2. * dl_uniform53(r.x[2], r.x[3]);
    return (element & 1) ? rho * sinpi(phi2) : rho * cospi(phi2);
}
