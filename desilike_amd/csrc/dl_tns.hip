// dl_tns.hip -- TNS one-loop tables on gfx950 (see dl_tns.h): geometry at context creation, loop GEMM + assembly per evaluation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dl_tns.h"
#include "dl_kernels.h"
#include "dl_host.hpp"

typedef double dl_tns_double4 __attribute__((ext_vector_type(4)));
typedef double dl_tns_double2 __attribute__((ext_vector_type(2)));

struct DlTnsPlan {
    DlTnsDev dev;
    std::vector<void*> allocs;
    size_t bytes = 0;
    // per-evaluation workspace (grown on demand; a plan serves one context, whose calls are serialised on its stream)
    int64_t cap_pts = 0;
    double* pk = nullptr;       // [nqp][cap_pts]
    double* qq = nullptr;       // [cap_pts]
    double* tables = nullptr;   // [cap_pts][n11][DL_TNS_NREC] sums of the loop kernel
    int64_t ldp = 0;            // leading dimension of pk / qq at the last launch
};

// ---------------------------------------------------------------------------------------------------------------------------------------------
// geometry (once per context)
// ---------------------------------------------------------------------------------------------------------------------------------------------
// bilinear coefficients + interpolation records: one thread per (k, pair (mu, q)); pairs beyond K are padding (zero coefficients)
__global__ __launch_bounds__(256) void dl_tns_geometry_kernel(DlTnsDev t, int32_t* geomj, double* geomw, double* coef) {
    const int ik = blockIdx.x;
    const double k = t.k11[ik];
    for (int kap = blockIdx.y * blockDim.x + threadIdx.x; kap < t.Kp; kap += gridDim.y * blockDim.x) {
        const size_t kappa = (size_t)ik * t.Kp + kap;
        double c[DL_TNS_NCOL];
        for (int i = 0; i < DL_TNS_NCOL; ++i) c[i] = 0.;
        int j = 0, iq = 0;
        double w0 = 0., w1 = 0.;
        if (kap < t.K) {
            const int im = kap / t.n_q;
            iq = kap % t.n_q;
            DlTnsGeom g;
            dl_tns_geometry(k, t.q[iq], t.jq[iq], t.mus[im], t.mus[t.n_mu + im], g);
            for (int i = 0; i < 27; ++i) c[i] = g.c[i];
            dl_tns_interp_weights(t.q, t.n_q, g.r, j, w0, w1);
        }
        // per lane of a 16-point tile: the LDS byte offsets (within the template tile [nqp][32]) of P(j0) and of P(q) -- the loop kernel uses them as they are
        for (int p = 0; p < 16; ++p) { geomj[(kappa * 16 + p) * 2] = (j * DL_TNS_PTS + p) * 8; geomj[(kappa * 16 + p) * 2 + 1] = (iq * DL_TNS_PTS + p) * 8; }
        geomw[2 * kappa] = w0; geomw[2 * kappa + 1] = w1;
        for (int i = 0; i < 16; ++i) { coef[(kappa * 16 + i) * 2] = c[i]; coef[(kappa * 16 + i) * 2 + 1] = c[16 + i]; }
    }
}

// linear coefficients that are sums over the cosines or closed forms: one thread per (k, q)
__global__ __launch_bounds__(256) void dl_tns_geometry_lin_kernel(DlTnsDev t, double* lin) {
    const int ik = blockIdx.x;
    const double k = t.k11[ik];
    int jk; double wk0, wk1;
    dl_tns_interp_weights(t.q, t.n_q, k, jk, wk0, wk1);   // pk_k = interp(k11, q, pk_q): full_shape.py:763 (k11 lies inside the template's range)
    for (int iq = threadIdx.x; iq < t.nqp; iq += blockDim.x) {
        double* row = lin + ((size_t)ik * t.nqp + iq) * DL_TNS_NLIN;
        for (int i = 0; i < DL_TNS_NLIN; ++i) row[i] = 0.;
        if (iq >= t.n_q) continue;
        const double q = t.q[iq], jq = t.jq[iq];
        row[DL_TL_PK] = (iq == jk ? wk0 : 0.) + (iq == jk + 1 ? wk1 : 0.);
        double s3 = 0.;
        for (int im = 0; im < t.n_mu; ++im) {
            DlTnsGeom g;
            dl_tns_geometry(k, q, jq, t.mus[im], t.mus[t.n_mu + im], g);
            s3 += g.sig3;
        }
        row[DL_TL_SIG3] = s3;
        double ff, gg, ka[4];
        dl_tns_kernels13(q / k, ff, gg);
        dl_tns_kernels_a(q / k, ka);
        row[DL_TL_13D] = 2. * jq * ff;
        row[DL_TL_13T] = 2. * jq * gg;
        for (int i = 0; i < 4; ++i) row[DL_TL_KA0 + i] = jq * ka[i];
    }
}

// A-term kernels multiplying P(k) P(|k - q|), folded over the interpolation onto the template's own wavenumbers: ONE thread per (k, kernel) walks the (mu, q) in
// order (a fixed summation order: the result does not depend on the launch)
__global__ __launch_bounds__(64) void dl_tns_geometry_fold_kernel(DlTnsDev t, const int32_t* geomj, const double* geomw, double* lin) {
    const int ik = blockIdx.x, u = threadIdx.x;
    if (u >= 4) return;
    const int i = (u == 0) ? 0 : (u == 1) ? 1 : (u == 2) ? 2 : 4;
    const double k = t.k11[ik];
    for (int im = 0; im < t.n_mu; ++im)
        for (int iq = 0; iq < t.n_q; ++iq) {
            const size_t kappa = (size_t)ik * t.Kp + (size_t)im * t.n_q + iq;
            const double w0 = geomw[2 * kappa], w1 = geomw[2 * kappa + 1];
            if (w0 == 0. && w1 == 0.) continue;
            DlTnsGeom g;
            dl_tns_geometry(k, t.q[iq], t.jq[iq], t.mus[im], t.mus[t.n_mu + im], g);
            const int j = geomj[kappa * 32] / (8 * DL_TNS_PTS);
            lin[((size_t)ik * t.nqp + j) * DL_TNS_NLIN + DL_TL_EA0 + u] += g.ca[i] * w0;
            lin[((size_t)ik * t.nqp + j + 1) * DL_TNS_NLIN + DL_TL_EA0 + u] += g.ca[i] * w1;
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// per evaluation
// ---------------------------------------------------------------------------------------------------------------------------------------------
// template of every point at the template's wavenumbers, wavenumber-major: pk [nqp][ldp]; qq [part][b] = partial sums of sum_q jq P(q)^2 (added in a fixed order by the
// loop kernel).  Workgroup = 64 points x one of DL_TNS_QPARTS ranges of wavenumbers.
#define DL_TNS_QPARTS 8
__global__ __launch_bounds__(256) void dl_tns_pk_kernel(DlObsDev o, DlTnsDev t, const double* __restrict__ theta, int n_params, int64_t B, int64_t ldp,
                                                        double* __restrict__ pk, double* __restrict__ qq) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * 64 + lane;
    const bool live = b < B;
    const double* th = theta + (size_t)(live ? b : 0) * n_params;
    const double dm_a = dl_get(o.dm, th) / o.a, dn = dl_get(o.dn, th);
    const int per = (t.nqp + DL_TNS_QPARTS - 1) / DL_TNS_QPARTS;
    const int j0 = blockIdx.y * per, j1 = min(j0 + per, t.nqp);
    double acc = 0.;
    for (int j = j0 + grp; j < j1; j += 4) {
        double v = 0.;
        if (live && j < t.n_q) v = (o.templ == 1) ? o.pk_fid[j] * exp(dm_a * o.sf_th[j] + dn * o.sf_lg[j]) : o.pk_fid[j];   // power_template.py:749
        if (b < ldp) pk[(size_t)j * ldp + b] = v;
        acc = fma(t.jq[j] * v, v, acc);
    }
    part[grp][lane] = acc;
    __syncthreads();
    if (grp == 0 && b < ldp) qq[(size_t)blockIdx.y * ldp + b] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// The loop GEMM.  LDS: the templates of 32 points [nqp][32]; a wave forms the left operand of its two 16-point tiles in registers and multiplies it with the [4 x 32]
// coefficient block it loaded from L2 (4 MFMA per step of 4 pairs (mu, q)); then the linear tables (left operand = the templates themselves).  Two ways of dealing
// the work to the 8 waves of a workgroup:
//   W = 1  every wave owns one table wavenumber (8 per workgroup) and runs the whole pair list: no reduction, no barrier after the templates have landed, the template
//          tile is loaded once per 8 wavenumbers -- large batches;
//   W = 8  one wavenumber per workgroup, the waves take every 8th round of the pair list and their partial accumulators are summed in wave order through LDS -- small
//          batches;  W = 2, 4: in between (2 or 4 waves per wavenumber, 4 or 2 wavenumbers per workgroup).  The launcher picks the variant that needs the least time
//          in whole rounds of workgroups.
// Output: the raw sums [point][k][48]; the assembly kernel turns them into the 29 tables.
#define DL_TNS_WAVES 8
#define DL_TNS_UNROLL 4

template <int W>   // waves per table wavenumber: 1 (WAVEK below), 2, 4 or 8 -- a workgroup works on 8 / W wavenumbers
__global__ __launch_bounds__(64 * DL_TNS_WAVES) void dl_tns_loop_kernel(DlTnsDev t, const double* __restrict__ pk, int64_t ldp, int n_tiles, double* __restrict__ sums_out) {
    constexpr bool WAVEK = (W == 1);
    constexpr int KPW = DL_TNS_WAVES / W;                    // wavenumbers per workgroup
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // workgroups of one (group of) k share an XCD (consecutive workgroup ids go round the 8 XCDs): the coefficients are fetched from HBM once
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int kgroup = (slot / n_tiles) * 8 + xcd, tile = slot % n_tiles;
    const int kslot = wave_s / W, part = wave_s % W;         // which of the workgroup's wavenumbers, which share of its pair list
    const int ik_raw = kgroup * KPW + kslot;
    if (kgroup * KPW >= t.n11) return;
    const bool live = ik_raw < t.n11;                        // (waves beyond the last wavenumber repeat it and do not store)
    const int ik = live ? ik_raw : t.n11 - 1;
    const int p16 = lane & 15, kk = lane >> 4;
    double* spk = lds;                                       // [nqp][32]
    // (the records address this tile by absolute LDS byte offsets: it must start at 0 -- true while the kernel has no static LDS)
    if ((uint32_t)(size_t)(const __attribute__((address_space(3))) double*)spk != 0u) __builtin_trap();
    for (int idx = tid; idx < t.nqp * DL_TNS_PTS; idx += 64 * DL_TNS_WAVES)
        spk[idx] = pk[(size_t)(idx >> 5) * ldp + (size_t)tile * DL_TNS_PTS + (idx & 31)];
    __syncthreads();
    dl_tns_double4 acc[2][2], accl[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { acc[m][0] = (dl_tns_double4){0., 0., 0., 0.}; acc[m][1] = acc[m][0]; accl[m] = acc[m][0]; }
    // Every wave takes every 8th group of DL_TNS_UNROLL steps (a step = 4 pairs (mu, q)); Kp holds a whole number of rounds.  Three-stage software pipeline: while
    // group g is multiplied, the LDS operands of group g + 1 are being read (their interpolation records arrived one iteration ago) and the records / coefficients
    // of group g + 2 are in flight from L2.
    typedef int32_t dl_tns_int2 __attribute__((ext_vector_type(2)));
    struct Rec { dl_tns_int2 j[DL_TNS_UNROLL]; dl_tns_double2 w[DL_TNS_UNROLL], c[DL_TNS_UNROLL]; };
    struct Raw { double pq[DL_TNS_UNROLL][2], pa[DL_TNS_UNROLL][2], pb[DL_TNS_UNROLL][2]; };
    struct Lhs { double g[DL_TNS_UNROLL][2]; };
    const int ngroups = t.Kp / (4 * DL_TNS_UNROLL);                                      // rounds of this wave: all of them, or every 8th starting at its index
    const int rounds = ngroups / W + (part < ngroups % W ? 1 : 0);
    // wave-uniform base pointers (scalar registers) + a 32-bit lane offset: the address arithmetic of the requests stays on the scalar unit
    const dl_tns_int2* gj = reinterpret_cast<const dl_tns_int2*>(t.geomj) + (size_t)ik * t.Kp * 16;
    const dl_tns_double2* gw = reinterpret_cast<const dl_tns_double2*>(t.geomw) + (size_t)ik * t.Kp;
    const dl_tns_double2* gc = reinterpret_cast<const dl_tns_double2*>(t.coef) + (size_t)ik * t.Kp * 16;
    const int lane_rec = kk, lane_c = kk * 16 + p16;
    const double* spt = spk + p16;
    // element offset (in pairs) of the first step of a round (past the end: the last round again, requested and not used)
    auto round_offset = [&](int round) {
        const int rr = round < rounds ? round : rounds - 1;
        const int g = rr * W + part;
        return (size_t)g * DL_TNS_UNROLL * 4;
    };
    auto load_step = [&](size_t e, int u, Rec& r) { r.j[u] = (gj + (e + 4 * u) * 16)[lane_c]; r.w[u] = (gw + e + 4 * u)[lane_rec]; r.c[u] = (gc + (e + 4 * u) * 16)[lane_c]; };
    auto read_step = [&](const Rec& r, int u, Raw& o) {
        // The record holds this lane's LDS byte addresses (the template tile is the first thing in the kernel's LDS, which starts at 0: dynamic LDS only -- asserted by
        // the parity tests); built from the integer, the address goes into the ds_read as it is (through `lds + offset` the compiler adds the segment's base, 0, with
        // a vector instruction per read).
        typedef const __attribute__((address_space(3))) double* lds_cptr;
        lds_cptr ra = (lds_cptr)(uintptr_t)(uint32_t)r.j[u].x;     // (through uintptr_t: no -Wint-to-pointer-cast; LDS pointers are 32 bits wide, the upper half is dropped again)
        lds_cptr rq = (lds_cptr)(uintptr_t)(uint32_t)r.j[u].y;
#pragma unroll
        for (int m = 0; m < 2; ++m) { o.pq[u][m] = rq[16 * m]; o.pa[u][m] = ra[16 * m]; o.pb[u][m] = ra[DL_TNS_PTS + 16 * m]; }
    };
    auto form_step = [&](const Rec& r, const Raw& o, int u, Lhs& l) {
#pragma unroll
        for (int m = 0; m < 2; ++m) l.g[u][m] = o.pq[u][m] * fma(r.w[u].x, o.pa[u][m], r.w[u].y * o.pb[u][m]);
    };
    auto mfma_pair = [&](const Rec& r, const Lhs& l, int u, int m) {
        acc[m][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(l.g[u][m], r.c[u].x, acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l.g[u][m], r.c[u].y, acc[m][1], 0, 0, 0);
    };
    auto load = [&](int round, Rec& r) { const size_t e = round_offset(round); for (int u = 0; u < DL_TNS_UNROLL; ++u) load_step(e, u, r); };
    auto read = [&](const Rec& r, Raw& o) { for (int u = 0; u < DL_TNS_UNROLL; ++u) read_step(r, u, o); };
    auto form = [&](const Rec& r, const Raw& o, Lhs& l) { for (int u = 0; u < DL_TNS_UNROLL; ++u) form_step(r, o, u, l); };
    // One round: RA is multiplied (its left operand LA was formed one round ago); the records and coefficients of RC are requested from L2 during the first MFMAs,
    // the LDS operands of RB are read during the next ones and turned into LB during the last ones.  The roles rotate through three register sets (no copies: a copy
    // would wait for the load it moves).  The order is pinned chunk by chunk (two MFMAs, then a few other instructions): left to itself the scheduler sinks the
    // requests to their uses.
static_assert(DL_TNS_UNROLL == 4, "the round below names its four steps");
#define DL_TNS_FENCE __builtin_amdgcn_sched_barrier(0);
#define DL_TNS_ROUND(RA, RB, RC, LA, LB, rnd)                                                          \
    { const size_t e_ = round_offset((rnd) + 2);                                                        \
    mfma_pair(RA, LA, 0, 0); DL_TNS_FENCE load_step(e_, 0, RC); load_step(e_, 1, RC); DL_TNS_FENCE      \
    mfma_pair(RA, LA, 0, 1); DL_TNS_FENCE load_step(e_, 2, RC); load_step(e_, 3, RC); DL_TNS_FENCE      \
    mfma_pair(RA, LA, 1, 0); DL_TNS_FENCE read_step(RB, 0, raw); read_step(RB, 1, raw); DL_TNS_FENCE    \
    mfma_pair(RA, LA, 1, 1); DL_TNS_FENCE read_step(RB, 2, raw); read_step(RB, 3, raw); DL_TNS_FENCE    \
    mfma_pair(RA, LA, 2, 0); DL_TNS_FENCE form_step(RB, raw, 0, LB); DL_TNS_FENCE                       \
    mfma_pair(RA, LA, 2, 1); DL_TNS_FENCE form_step(RB, raw, 1, LB); DL_TNS_FENCE                       \
    mfma_pair(RA, LA, 3, 0); DL_TNS_FENCE form_step(RB, raw, 2, LB); DL_TNS_FENCE                       \
    mfma_pair(RA, LA, 3, 1); DL_TNS_FENCE form_step(RB, raw, 3, LB); DL_TNS_FENCE }
    Rec r0, r1, r2;
    Raw raw;
    Lhs l0, l1;
    load(0, r0);
    load(1, r1);
    read(r0, raw);
    form(r0, raw, l0);
    int round = 0;
    for (; round + 6 <= rounds; round += 6) {
        DL_TNS_ROUND(r0, r1, r2, l0, l1, round)
        DL_TNS_ROUND(r1, r2, r0, l1, l0, round + 1)
        DL_TNS_ROUND(r2, r0, r1, l0, l1, round + 2)
        DL_TNS_ROUND(r0, r1, r2, l1, l0, round + 3)
        DL_TNS_ROUND(r1, r2, r0, l0, l1, round + 4)
        DL_TNS_ROUND(r2, r0, r1, l1, l0, round + 5)
    }
    // tail: the same sequence, round by round
    if (round < rounds) { DL_TNS_ROUND(r0, r1, r2, l0, l1, round) ++round; }
    if (round < rounds) { DL_TNS_ROUND(r1, r2, r0, l1, l0, round) ++round; }
    if (round < rounds) { DL_TNS_ROUND(r2, r0, r1, l0, l1, round) ++round; }
    if (round < rounds) { DL_TNS_ROUND(r0, r1, r2, l1, l0, round) ++round; }
    if (round < rounds) { DL_TNS_ROUND(r1, r2, r0, l0, l1, round) ++round; }
#undef DL_TNS_ROUND
    // linear tables: left operand = the templates
    const double* gl = t.lin + ((size_t)ik * t.nqp + kk) * DL_TNS_NLIN + p16;
    for (int s = part; s < t.nqp / 4; s += W) {
        const double cl = gl[(size_t)4 * s * DL_TNS_NLIN];
#pragma unroll
        for (int m = 0; m < 2; ++m) accl[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(spt[(size_t)(4 * s + kk) * DL_TNS_PTS + 16 * m], cl, accl[m], 0, 0, 0);
    }
    if (WAVEK) {   // accumulator element r of lane (kk, p16): point 16 m + kk + 4 r, column p16
        if (live) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double* rec = sums_out + (((size_t)tile * DL_TNS_PTS + 16 * m + kk + 4 * r) * t.n11 + ik) * DL_TNS_NREC + p16;
                    rec[0] = acc[m][0][r]; rec[16] = acc[m][1][r]; rec[32] = accl[m][r];
                }
        }
        return;
    }
    __syncthreads();                                         // every wave is done with the templates: their space takes the partial accumulators
    double* red = lds;                                       // [wave][m][tile3][r][lane]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[(((size_t)(wave * 2 + m) * 3 + 0) * 4 + r) * 64 + lane] = acc[m][0][r];
            red[(((size_t)(wave * 2 + m) * 3 + 1) * 4 + r) * 64 + lane] = acc[m][1][r];
            red[(((size_t)(wave * 2 + m) * 3 + 2) * 4 + r) * 64 + lane] = accl[m][r];
        }
    __syncthreads();
    for (int idx = tid; idx < KPW * DL_TNS_PTS * DL_TNS_NREC; idx += 64 * DL_TNS_WAVES) {
        const int ks = idx / (DL_TNS_PTS * DL_TNS_NREC), rem = idx % (DL_TNS_PTS * DL_TNS_NREC);
        const int pt = rem / DL_TNS_NREC, col = rem % DL_TNS_NREC;
        const int m = pt >> 4, prow = pt & 15, tl = col >> 4, c16 = col & 15;
        const int r = prow >> 2, g = prow & 3;                // accumulator row = g + 4 r  (g = lane >> 4)
        const int ikk = kgroup * KPW + ks;
        if (ikk >= t.n11) continue;
        double sum = 0.;
        for (int wv = 0; wv < W; ++wv) sum += red[(((size_t)((ks * W + wv) * 2 + m) * 3 + tl) * 4 + r) * 64 + g * 16 + c16];   // in wave order: deterministic
        sums_out[(((size_t)tile * DL_TNS_PTS + pt) * t.n11 + ikk) * DL_TNS_NREC + col] = sum;
    }
}

// Assembly: PPW points per workgroup, PPW x (5 or 6) polynomials = at most 16 rows of one MFMA tile.
// LDS: Q [16][ldq] | M [16][ldq] (ldq = n11 rounded up to odd: rows of a tile land in different banks) | per point: cvec [6][32] | mu records [DL_MAX_MU][8] | scalars [8] | out [n_in + n_kin]

template <int PPW>
__global__ __launch_bounds__(256) void dl_tns_assemble_kernel(DlObsDev o, DlTnsDev t, const double* __restrict__ theta, int n_params, int64_t B, const double* __restrict__ raw,
                                                              const double* __restrict__ qq, int64_t ldp, double* __restrict__ power, int64_t ld_power) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n11 = t.n11, nq = (o.n_ct > 0) ? 6 : 5, ldq = dl_tns_ldq(n11);
    double* Q = lds;                                            // row p nq + n
    double* M = lds + (size_t)16 * ldq;
    double* extra = lds + (size_t)32 * ldq;
    const size_t per = dl_tns_assemble_point_doubles(o.n_in, o.n_kin);
    const int64_t b0 = (int64_t)blockIdx.x * PPW;
    // ---- per-point scalars, combination coefficients, mu records ----
    for (int p = 0; p < PPW; ++p) {
        const int64_t b = b0 + p < B ? b0 + p : B - 1;           // (a ragged last workgroup repeats the last point and does not store it)
        const double* th = theta + (size_t)b * n_params;
        double* cvec = extra + p * per;
        double* murec = cvec + 6 * 32;
        double* sc = murec + (size_t)8 * DL_MAX_MU;
        double qpar, qper;
        dl_ap_qparqper(o, th, qpar, qper);
        const double f = o.f_fid * dl_get(o.df, th);
        if (tid < 6 * 32) cvec[tid] = dl_tns_combine_coef(tid >> 5, tid & 31, f, dl_get(o.b1X, th), dl_get(o.b2, th), dl_get(o.bs, th), dl_get(o.b3, th));
        if (tid >= 192 && tid < 192 + o.n_mu) dl_tns_mu_record(o, qpar, qper, tid - 192, murec);
        if (tid == 255) {
            double qqv = 0.;
            for (int pp = 0; pp < DL_TNS_QPARTS; ++pp) qqv += qq[(size_t)pp * ldp + b];
            sc[0] = qper; sc[1] = dl_get(o.sigmav, th); sc[2] = dl_get(o.sn0, th) / o.nd; sc[3] = qqv;
        }
    }
    for (int idx = tid + PPW * nq * ldq; idx < 16 * ldq; idx += nthr) Q[idx] = 0.;   // unused rows of the tile
    __syncthreads();
    // ---- the 29 tables from the sums, combined into the polynomials Q_n ----
    for (int idx = tid; idx < n11 * PPW; idx += nthr) {
        const int p = idx / n11, i = idx - p * n11;
        const int64_t b = b0 + p < B ? b0 + p : B - 1;
        const double* cvec = extra + p * per;
        const double qqv = cvec[6 * 32 + 8 * DL_MAX_MU + 3];
        double rec[DL_TNS_NREC], v[DL_TNS_NTAB];
        const dl_tns_double4* src = reinterpret_cast<const dl_tns_double4*>(raw + ((size_t)b * n11 + i) * DL_TNS_NREC);
#pragma unroll
        for (int r4 = 0; r4 < DL_TNS_NREC / 4; ++r4) { const dl_tns_double4 x4 = src[r4]; rec[4 * r4] = x4.x; rec[4 * r4 + 1] = x4.y; rec[4 * r4 + 2] = x4.z; rec[4 * r4 + 3] = x4.w; }
#pragma unroll
        for (int r = 0; r < DL_TNS_NTAB; ++r) v[r] = dl_tns_table_entry(r, rec, rec + 32, qqv, t.sumw);
        for (int n = 0; n < nq; ++n) {
            double sum = 0.;
#pragma unroll
            for (int r = 0; r < DL_TNS_NTAB; ++r) sum = fma(cvec[n * 32 + r], v[r], sum);
            Q[(size_t)(p * nq + n) * ldq + i] = sum;
        }
    }
    __syncthreads();
    // ---- second derivatives of the not-a-knot splines, M = Q S^T: one 16-row MFMA tile (the polynomials of the workgroup's points), 16 knots per column tile, the
    //      operator streams from L2 (336 us of a 450 us launch as a scalar loop: 4.5 TFLOP/s) ----
    {
        const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kk = lane >> 4;
        const int ntiles = (n11 + 15) / 16, nsteps = (n11 + 3) / 4;
        const double* qrow = Q + (size_t)r16 * ldq;
        for (int t0 = wave; t0 < ntiles; t0 += 12) {              // three column tiles per pass and wave: t0, t0 + 4, t0 + 8
            dl_tns_double4 acc[3];
            int col[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) { acc[u] = (dl_tns_double4){0., 0., 0., 0.}; const int tt = t0 + 4 * u; col[u] = tt < ntiles ? tt * 16 + r16 : -1; if (col[u] >= n11) col[u] = -1; }
            for (int st = 0; st < nsteps; ++st) {
                const int j = 4 * st + kk;
                const bool jin = j < n11;
                const double a = jin ? qrow[j] : 0.;
                double bv[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) bv[u] = (jin && col[u] >= 0) ? t.spT[(size_t)j * n11 + col[u]] : 0.;
#pragma unroll
                for (int u = 0; u < 3; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[u], acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (col[u] >= 0)
#pragma unroll
                    for (int r = 0; r < 4; ++r) M[(size_t)(kk + 4 * r) * ldq + col[u]] = acc[u][r];   // accumulator element r of lane (kk, r16): row kk + 4 r, column r16
        }
    }
    __syncthreads();
    // ---- evaluation at the AP-distorted (k, mu), damping, projection, bias-independent additions ----
    for (int idx = tid; idx < o.n_kin * PPW; idx += nthr) {
        const int p = idx / o.n_kin, ik = idx - p * o.n_kin;
        const int64_t b = b0 + p < B ? b0 + p : B - 1;
        const double* th = theta + (size_t)b * n_params;
        const double* Qp = Q + (size_t)p * nq * ldq;
        const double* Mp = M + (size_t)p * nq * ldq;
        const double* murec = extra + p * per + 6 * 32;
        const double* sc = murec + (size_t)8 * DL_MAX_MU;
        double* out = extra + p * per + 6 * 32 + (size_t)8 * DL_MAX_MU + 8;
        dl_tns_eval_k(o, t.fog, t.k11_0, t.inv_dk11, t.x11, n11, ldq, nq, Qp, Mp, murec, sc, th, ik, out);
    }
    __syncthreads();
    for (int p = 0; p < PPW; ++p) {
        if (b0 + p >= B) break;
        const double* out = extra + p * per + 6 * 32 + (size_t)8 * DL_MAX_MU + 8;
        double* power_row = power + (size_t)(b0 + p) * (1 + o.n_var) * ld_power + o.col_offset;
        for (int idx = tid; idx < o.n_in; idx += nthr) power_row[idx] = out[idx];
        if (o.n_var > 0 && o.n_ct > 0) {                        // derivative rows of analytically solved counter terms (as dl_fs_phase4)
            for (int c = 0; c < o.n_ct; ++c)
                for (int tt = 0; tt < 2; ++tt) {
                    const int slot = o.marg_ct_slot[c][tt];
                    if (slot < 0) continue;
                    if (tt == 1 && o.marg_ct_slot[c][0] == slot) continue;
                    const double wgt = (o.marg_ct_slot[c][0] == o.marg_ct_slot[c][1]) ? 1. : 0.5;
                    double* drow = power_row + (size_t)(1 + slot) * ld_power;
                    for (int idx = tid; idx < o.n_in; idx += nthr) drow[idx] = wgt * o.ct_matrix[(size_t)idx * o.n_ct + c] * out[o.n_in + idx % o.n_kin];
                }
        }
    }
}

// the 29 tables [B][29][n11] from the loop kernel's sums (diagnostics / parity)
__global__ void dl_tns_tables_kernel(DlTnsDev t, const double* __restrict__ raw, const double* __restrict__ qq, int64_t ldp, int64_t B, double* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * DL_TNS_NTAB * t.n11) return;
    const int i = (int)(idx % t.n11), r = (int)((idx / t.n11) % DL_TNS_NTAB);
    const int64_t b = idx / ((int64_t)DL_TNS_NTAB * t.n11);
    double qqv = 0.;
    for (int p = 0; p < DL_TNS_QPARTS; ++p) qqv += qq[(size_t)p * ldp + b];
    const double* rec = raw + ((size_t)b * t.n11 + i) * DL_TNS_NREC;
    out[idx] = dl_tns_table_entry(r, rec, rec + 32, qqv, t.sumw);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------------------------------
template <typename T>
static T* tns_alloc(DlTnsPlan* plan, size_t n, const T* host = nullptr) {
    T* dev = nullptr;
    if (hipMalloc((void**)&dev, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
    plan->allocs.push_back(dev);
    plan->bytes += n * sizeof(T);
    if (host && hipMemcpy(dev, host, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    if (!host && hipMemset(dev, 0, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
    return dev;
}

DlTnsPlan* dl_tns_create(const double* k11, int n11, const double* q, int n_q, const double* mus, const double* wmus, int n_mu, int fog, const char** err) {
    static thread_local std::string msg;
    auto fail = [&](DlTnsPlan* plan, const std::string& m) { msg = m; if (err) *err = msg.c_str(); dl_tns_destroy(plan); return (DlTnsPlan*)nullptr; };
    if (n11 < 5 || n_q < 4 || n_mu < 1 || n_mu > DL_TNS_MAX_MU) return fail(nullptr, "tns: table / template / cosine grid sizes out of range");
    for (int i = 0; i + 1 < n_q; ++i) if (!(q[i + 1] > q[i])) return fail(nullptr, "tns: template wavenumbers must increase");
    if ((size_t)((n_q + 3) & ~3) * DL_TNS_PTS * sizeof(double) > 150 * 1024) return fail(nullptr, "tns: template grid too large for the loop kernel's LDS tile (32 points x n_t wavenumbers <= 150 KB: n_t <= 600; the reference uses 500, full_shape.py:855)");
    if (!(k11[0] > q[0]) || !(k11[n11 - 1] < q[n_q - 1])) return fail(nullptr, "tns: table wavenumbers must lie inside the template's range (full_shape.py:29, 875)");
    DlTnsPlan* plan = new DlTnsPlan();
    DlTnsDev& t = plan->dev;
    std::memset(&t, 0, sizeof(t));
    t.n11 = n11; t.n_q = n_q; t.nqp = (n_q + 3) & ~3; t.n_mu = n_mu; t.K = n_mu * n_q; t.fog = fog;
    { const int round = 4 * DL_TNS_UNROLL; t.Kp = (t.K + round - 1) / round * round; if (t.Kp < round * DL_TNS_WAVES) t.Kp = round * DL_TNS_WAVES; }   // whole rounds; every wave of the split-K variant at least one
    t.k11_0 = k11[0]; t.inv_dk11 = (n11 - 1) / (k11[n11 - 1] - k11[0]);
    for (int i = 0; i + 1 < n11; ++i)
        if (std::fabs((k11[i + 1] - k11[i]) * t.inv_dk11 - 1.) > 1e-9) return fail(plan, "tns: table wavenumbers must be uniformly spaced (full_shape.py:875)");
    std::vector<double> x11(n11), qp(t.nqp), jq(t.nqp, 0.), mw(2 * n_mu);
    for (int i = 0; i < n11; ++i) x11[i] = std::log10(k11[i]);
    for (int j = 0; j < t.nqp; ++j) qp[j] = q[j < n_q ? j : n_q - 1];
    const double pi = 3.14159265358979323846;
    for (int j = 0; j < n_q; ++j) {   // trapezoidal weights (utils.py:620-622) times q^2 / (4 pi^2) (full_shape.py:753)
        const double wq = (j == 0 ? q[1] - q[0] : j == n_q - 1 ? q[n_q - 1] - q[n_q - 2] : q[j + 1] - q[j - 1]) / 2.;
        jq[j] = q[j] * q[j] * wq / (4. * pi * pi);
    }
    t.sumw = 0.;
    for (int m = 0; m < n_mu; ++m) { mw[m] = mus[m]; mw[n_mu + m] = wmus[m]; t.sumw += wmus[m]; }
    // operator y -> second derivatives of the not-a-knot spline on x11 (transposed): columns from unit vectors through the same sweeps the templates use
    DlSplineSetup sp;
    std::string serr;
    if (!dl_spline_setup(x11, sp, serr)) return fail(plan, "tns: " + serr);
    std::vector<double> spT((size_t)n11 * n11), unit(n11, 0.), Mv;
    for (int j = 0; j < n11; ++j) {
        unit[j] = 1.;
        dl_spline_moments_serial(unit, sp, Mv);
        Mv[0] = sp.end0a * Mv[1] + sp.end0b * Mv[2];
        Mv[n11 - 1] = sp.end1a * Mv[n11 - 2] + sp.end1b * Mv[n11 - 3];
        for (int i = 0; i < n11; ++i) spT[(size_t)j * n11 + i] = Mv[i];
        unit[j] = 0.;
    }
    t.k11 = tns_alloc(plan, n11, k11); t.x11 = tns_alloc(plan, n11, x11.data()); t.q = tns_alloc(plan, t.nqp, qp.data()); t.jq = tns_alloc(plan, t.nqp, jq.data());
    t.mus = tns_alloc(plan, 2 * n_mu, mw.data()); t.spT = tns_alloc(plan, spT.size(), spT.data());
    int32_t* geomj = tns_alloc<int32_t>(plan, (size_t)n11 * t.Kp * 32);
    double* geomw = tns_alloc<double>(plan, (size_t)n11 * t.Kp * 2);
    double* coef = tns_alloc<double>(plan, (size_t)n11 * t.Kp * DL_TNS_NCOL);
    double* lin = tns_alloc<double>(plan, (size_t)n11 * t.nqp * DL_TNS_NLIN);
    if (!t.k11 || !t.x11 || !t.q || !t.jq || !t.mus || !t.spT || !geomj || !geomw || !coef || !lin) return fail(plan, "tns: device allocation failed");
    t.geomj = geomj; t.geomw = geomw; t.coef = coef; t.lin = lin;
    if (hipDeviceSynchronize() != hipSuccess) return fail(plan, "tns: hipDeviceSynchronize failed");
    hipLaunchKernelGGL(dl_tns_geometry_kernel, dim3(n11, 8), dim3(256), 0, 0, t, geomj, geomw, coef);
    hipLaunchKernelGGL(dl_tns_geometry_lin_kernel, dim3(n11), dim3(256), 0, 0, t, lin);
    hipLaunchKernelGGL(dl_tns_geometry_fold_kernel, dim3(n11), dim3(64), 0, 0, t, geomj, geomw, lin);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return fail(plan, "tns: geometry kernels failed");
    return plan;
}

void dl_tns_destroy(DlTnsPlan* plan) {
    if (!plan) return;
    for (void* p : plan->allocs) (void)hipFree(p);
    if (plan->pk) (void)hipFree(plan->pk);
    if (plan->qq) (void)hipFree(plan->qq);
    if (plan->tables) (void)hipFree(plan->tables);
    delete plan;
}

size_t dl_tns_bytes(const DlTnsPlan* plan) { return plan ? plan->bytes : 0; }

static bool tns_reserve(DlTnsPlan* plan, int64_t pts) {
    if (pts <= plan->cap_pts) return true;
    (void)hipDeviceSynchronize();   // (a smaller workspace may still be in use by launches in flight)
    if (plan->pk) (void)hipFree(plan->pk);
    if (plan->qq) (void)hipFree(plan->qq);
    if (plan->tables) (void)hipFree(plan->tables);
    plan->pk = plan->qq = plan->tables = nullptr;
    plan->cap_pts = 0;
    const DlTnsDev& t = plan->dev;
    if (hipMalloc((void**)&plan->pk, (size_t)t.nqp * pts * sizeof(double)) != hipSuccess || hipMalloc((void**)&plan->qq, (size_t)DL_TNS_QPARTS * pts * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&plan->tables, (size_t)pts * t.n11 * DL_TNS_NREC * sizeof(double)) != hipSuccess) return false;
    plan->cap_pts = pts;
    return true;
}

// pass size: the table records of a pass (n11 x 256 B per point) stay below 1 GiB
static int64_t tns_pass_points(const DlTnsDev& t) {
    int64_t pts = ((int64_t)1 << 30) / ((int64_t)t.n11 * DL_TNS_NREC * 8);
    pts = pts / DL_TNS_PTS * DL_TNS_PTS;
    return pts < DL_TNS_PTS ? DL_TNS_PTS : (pts > 8192 ? 8192 : pts);
}

static bool tns_run_loop(DlTnsPlan* plan, const DlObsDev& obs, const double* theta, int n_params, int64_t nb, hipStream_t stream) {
    const DlTnsDev& t = plan->dev;
    const int64_t ldp = (nb + DL_TNS_PTS - 1) / DL_TNS_PTS * DL_TNS_PTS;
    if (!tns_reserve(plan, ldp)) { dl_set_last_error("tns: workspace allocation failed"); return false; }
    hipLaunchKernelGGL(dl_tns_pk_kernel, dim3((unsigned)((ldp + 63) / 64), DL_TNS_QPARTS), dim3(256), 0, stream, obs, t, theta, n_params, nb, ldp, plan->pk, plan->qq);   // (the profiling events go to the loop kernel)
    plan->ldp = ldp;
    const int n_tiles = (int)(ldp / DL_TNS_PTS);
    const size_t tmpl = (size_t)t.nqp * DL_TNS_PTS * sizeof(double), red = (size_t)DL_TNS_WAVES * 2 * 3 * 4 * 64 * sizeof(double);
    // Which variant: all run in rounds of one workgroup per CU; a workgroup with W waves per wavenumber lasts about 20 + 55 (8 / W) microseconds at 500 x 10 pairs
    // (measured: 75 at W = 8, 460 at W = 1) and there are n11 W / 8 x tiles of them -- whichever needs the least time in whole rounds.  DL_TNS_WAVEK=0 / 1 forces
    // W = 8 / 1, DL_TNS_W=<1|2|4|8> any variant (diagnostics, tests).
    static const char* force = getenv("DL_TNS_WAVEK");
    static const char* force_w = getenv("DL_TNS_W");
    static int n_cu = 0;
    if (n_cu == 0) { int dev = 0; hipDeviceProp_t prop; n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
    int w_best = 8;
    double t_best = 1e300;
    for (int w = 8; w >= 1; w /= 2) {
        const int kpw = DL_TNS_WAVES / w;
        const int64_t wgs = (int64_t)((t.n11 + kpw - 1) / kpw) * n_tiles, rounds = (wgs + n_cu - 1) / n_cu;
        const double cost = (double)rounds * (20. + 55. * (8. / w) * ((double)t.Kp / 5008.));
        if (cost < t_best) { t_best = cost; w_best = w; }
    }
    if (force) w_best = atoi(force) != 0 ? 1 : 8;
    if (force_w) { const int w = atoi(force_w); if (w == 1 || w == 2 || w == 4 || w == 8) w_best = w; }
    const int kpw = DL_TNS_WAVES / w_best;
    const int kgroups = (t.n11 + kpw - 1) / kpw;
    const unsigned grid = (unsigned)(((kgroups + 7) / 8) * n_tiles * 8);
    const size_t lds_bytes = w_best == 1 ? tmpl : std::max(tmpl, red);
    (void)hipFuncSetAttribute((const void*)dl_tns_loop_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)dl_tns_loop_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)dl_tns_loop_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)dl_tns_loop_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (w_best == 1) { DL_LAUNCH(dl_tns_loop_kernel<1>, dim3(grid), dim3(64 * DL_TNS_WAVES), lds_bytes, stream, t, plan->pk, ldp, n_tiles, plan->tables); }
    else if (w_best == 2) { DL_LAUNCH(dl_tns_loop_kernel<2>, dim3(grid), dim3(64 * DL_TNS_WAVES), lds_bytes, stream, t, plan->pk, ldp, n_tiles, plan->tables); }
    else if (w_best == 4) { DL_LAUNCH(dl_tns_loop_kernel<4>, dim3(grid), dim3(64 * DL_TNS_WAVES), lds_bytes, stream, t, plan->pk, ldp, n_tiles, plan->tables); }
    else { DL_LAUNCH(dl_tns_loop_kernel<8>, dim3(grid), dim3(64 * DL_TNS_WAVES), lds_bytes, stream, t, plan->pk, ldp, n_tiles, plan->tables); }
    return true;
}

void dl_launch_tns(const DlObsDev& obs, const double* theta, int n_params, int64_t B, double* power, int64_t ld_power, hipStream_t stream) {
    DlTnsPlan* plan = (DlTnsPlan*)obs.tns_plan;
    if (!plan) { dl_set_last_error("tns: observable without a plan"); return; }
    const DlTnsDev& t = plan->dev;
    const int64_t pass = tns_pass_points(t);
    (void)hipFuncSetAttribute((const void*)dl_tns_assemble_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)dl_tns_assemble_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)dl_tns_assemble_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nq = obs.n_ct > 0 ? 6 : 5;
    for (int64_t b0 = 0; b0 < B; b0 += pass) {
        const int64_t nb = std::min(pass, B - b0);
        const double* th = theta + (size_t)b0 * n_params;
        if (!tns_run_loop(plan, obs, th, n_params, nb, stream)) return;
        double* prow = power + (size_t)b0 * (1 + obs.n_var) * ld_power;
        // points per workgroup: as many polynomials as the 16-row tile holds when the batch still gives every CU two workgroups, one point otherwise
        int ppw = nb >= 1536 ? 16 / nq : nb >= 1024 ? 2 : 1;
        while (ppw > 1 && dl_tns_assemble_doubles(t.n11, obs.n_in, obs.n_kin, ppw) * sizeof(double) > 80 * 1024) --ppw;
        const size_t shm = dl_tns_assemble_doubles(t.n11, obs.n_in, obs.n_kin, ppw) * sizeof(double);
        const dim3 grid((unsigned)((nb + ppw - 1) / ppw));
        if (ppw == 3) hipLaunchKernelGGL(dl_tns_assemble_kernel<3>, grid, dim3(256), shm, stream, obs, t, th, n_params, nb, plan->tables, plan->qq, plan->ldp, prow, ld_power);
        else if (ppw == 2) hipLaunchKernelGGL(dl_tns_assemble_kernel<2>, grid, dim3(256), shm, stream, obs, t, th, n_params, nb, plan->tables, plan->qq, plan->ldp, prow, ld_power);
        else hipLaunchKernelGGL(dl_tns_assemble_kernel<1>, grid, dim3(256), shm, stream, obs, t, th, n_params, nb, plan->tables, plan->qq, plan->ldp, prow, ld_power);
    }
}

int dl_tns_tables(const DlObsDev& obs, const double* theta, int n_params, int64_t B, double* tables_dev, hipStream_t stream) {
    DlTnsPlan* plan = (DlTnsPlan*)obs.tns_plan;
    if (!plan) { dl_set_last_error("tns: observable without a plan"); return 1; }
    const DlTnsDev& t = plan->dev;
    const int64_t pass = tns_pass_points(t);
    for (int64_t b0 = 0; b0 < B; b0 += pass) {
        const int64_t nb = std::min(pass, B - b0);
        if (!tns_run_loop(plan, obs, theta + (size_t)b0 * n_params, n_params, nb, stream)) return 1;
        const int64_t total = nb * DL_TNS_NTAB * t.n11;
        hipLaunchKernelGGL(dl_tns_tables_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, t, plan->tables, plan->qq, plan->ldp, nb, tables_dev + (size_t)b0 * DL_TNS_NTAB * t.n11);
    }
    return 0;
}
