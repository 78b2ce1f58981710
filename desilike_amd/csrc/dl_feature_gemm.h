// Feature GEMM of the emulated (velocileptors-table) theories: whitened residual rows without ever forming the feature vectors.
//
//   The theory vector of these observables is separable, phi[(h, m)] = basis_h(theta) * mono_m(theta) (h: emulator basis = last hidden layer of the
//   MLP / Taylor monomials, m: 19 bias monomials, full_shape.py:1182-1186), and so is every derivative row of an analytically solved parameter
//   (d phi / d x_s = basis_h * dmono_s,m).  With G[(m, j)][h] = W~[j][(h, m)] (the folded last layer x k-interpolation x window x L^T operator, regrouped):
//       U[m][j] = sum_h G[(m, j)][h] basis_h            -- ONE GEMM per point, K = n_basis (~65), shared by all rows
//       row_r[j] = sum_m mono_r[m] U[m][j]              -- 19 FMAs per output in the epilogue, r = 0 (residual), 1 .. n_var (derivative rows)
//   instead of (1 + n_var) dense rows of K = 19 n_basis (~1235) through the generic GEMM: 6x fewer flops at 5 solved parameters, and the theory kernel
//   writes 80 + 20 (1 + n_var) doubles per point instead of 1280 (1 + n_var).
//
//   Workgroup = 16 points x 128 output columns, 8 waves = 8 column blocks of 16; a wave holds U[16 points][m][16 columns] for 10 (then 9) monomials in MFMA
//   accumulators (v_mfma_f64_16x16x4_f64), K advances 8 at a time: the A operand (basis, 16 x n_basis) comes from LDS, the B operand streams from L2 in
//   fragment order (the host lays G out as [column block][k / 8][m][lane][2]: one load instruction = 1 KB contiguous), double-buffered in registers,
//   loads never under control flow.  k -> (MFMA step, lane group) is permuted identically for A and B: lane group g owns k = 8 q + 2 g + {0, 1}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double dl_fg_double2 __attribute__((ext_vector_type(2)));
typedef double dl_fg_double4 __attribute__((ext_vector_type(4)));

#define DL_FG_PTS 16          // points per workgroup (one MFMA row tile)
#ifndef DL_FG_NM
#define DL_FG_NM 19           // bias monomials
#define DL_FG_MONO_LD 20      // monomial row padded to 20 doubles
#endif
#define DL_FG_MG 10           // monomials per accumulator group (two groups: 10 + 9)

// LDS row stride (doubles) of a point record: >= rec_len, = 2 mod 32 (the 16 points of an operand read then hit distinct banks, rows stay 16-byte aligned)
static inline __host__ __device__ int dl_fg_lds_stride(int rec_len) { return (rec_len + 31) / 32 * 32 + 2; }

// Gram-matrix epilogue (analytic marginalisation, one observable, N_pad = 128): the rows X = [residual + bias; derivative rows + tconst_s] of the 16 points go to
// LDS instead of memory and each wave forms G = X X^T of two points with v_mfma_f64_16x16x4_f64 (A and B operands are one register); only G [16, 16] per point is
// written -- 2 KB instead of (1 + n_var) x 1 KB of rows that the marginalised finalize would read back (25 MB per 4096 points at 5 solved parameters).
#define DL_FG_XLD 132   // LDS row stride of X (doubles): 128 columns + 4 (the 16 rows of an operand read fall on distinct bank groups)
struct DlFgGram {
    double* x;                 // LDS [16 points][xr][DL_FG_XLD], rows beyond the used ones are not read
    int xr;                    // rows of X = 1 + n_s
    int row_of[6];             // X row of device row r (0: residual; r >= 1: 1 + solved index of the parameter whose derivative row r is)
    const double* cst[6];      // constant part added to device row r: bias, or tconst of that solved parameter ([128] each)
    double* gram;              // [B, 256]; null: the 8 x 8 block of point pt goes to LDS instead, x + pt xr DL_FG_XLD + 8 i + j (the solve follows in the same kernel)
    unsigned long long* stamps;   // DL_EF_STAMPS diagnostics (null in production): 16 x s_memtime per workgroup
    int nz[6][2];              // monomials the derivative row r >= 1 touches (dl_velocileptors_row_support), -1: none
    int scaled;                // every derivative row lives on monomials 12-18: monomials 0-11 feed row 0 only, through registers (dl_fg_gram_epilogue_row0)
};
#define DL_FG_STAMP(slot) if (GRAM && gr->stamps != nullptr && threadIdx.x == 0) gr->stamps[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime();

// G = X X^T of the workgroup's 16 points from their rows in LDS (after a barrier): wave w takes points 2 w, 2 w + 1
__device__ __forceinline__ void dl_fg_gram_phase(const DlFgGram* gr, int wave, int lane, int g, int64_t B, int64_t p0) {
    if (gr->xr <= 8) {
        // TWO points per 16-row tile (rows 0-7: point 2 wave, rows 8-15: point 2 wave + 1; the off-diagonal 8 x 8 blocks are cross products nobody needs): half the
        // MFMAs of a tile per point, and the 32 operand values of a lane are requested from LDS together, ahead of the chain (the loop read two values, waited, issued
        // two MFMAs: 5 us for 64 MFMAs).  One chain in k order per point pair: G equals the one-point-per-tile product bit for bit.
        const int j = lane & 15, pp = j >> 3, row = j & 7;
        const bool live = row < gr->xr;
        const double* xp = gr->x + ((size_t)(2 * wave + pp) * gr->xr + (live ? row : 0)) * DL_FG_XLD + g;
        double xv[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) xv[k] = xp[4 * k];
        dl_fg_double4 acc0 = {0., 0., 0., 0.};
#pragma unroll
        for (int k = 0; k < 32; ++k) { const double x0 = live ? xv[k] : 0.; acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc0, 0, 0, 0); }
        // only the 8 x 8 block of a point's 16 x 16 slot is written: the finalize kernels read entries [i][j], i, j <= n_s, and nothing else (zero-filling the rest
        // of the shared workspace was 6 MB of stores per 4096 points)
        const int64_t pa = p0 + 2 * wave;
        if (gr->gram == nullptr) {
            // over the X rows of the wave's own two points: every operand read of this wave precedes these writes (LDS operations of a wave execute in order), no
            // other wave reads these rows
            double* gl = gr->x + (size_t)(2 * wave + pp) * gr->xr * DL_FG_XLD;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = g + 4 * r;
                if ((i >> 3) == pp) gl[(i & 7) * 8 + row] = acc0[r];
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // C layout: G[(l >> 4) + 4 r][l & 15]
            const int i = g + 4 * r;
            if ((i >> 3) == pp && pa + pp < B) gr->gram[(size_t)(pa + pp) * 256 + (i & 7) * 16 + row] = acc0[r];
        }
    } else {
    const int xrow = lane & 15;
    const bool live = xrow < gr->xr;
    // the two points of this wave side by side: two independent MFMA chains (one chain alone is 32 dependent MFMAs of 64 cycles)
    const double* xp0 = gr->x + ((size_t)(2 * wave) * gr->xr + (live ? xrow : 0)) * DL_FG_XLD + g;
    const double* xp1 = xp0 + (size_t)gr->xr * DL_FG_XLD;
    dl_fg_double4 acc0 = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        const double x0 = live ? xp0[4 * k] : 0., x1 = live ? xp1[4 * k] : 0.;
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, acc1, 0, 0, 0);
    }
    // C layout: G[(l >> 4) + 4 r][l & 15]
    if (p0 + 2 * wave < B) {
#pragma unroll
        for (int r = 0; r < 4; ++r) gr->gram[(size_t)(p0 + 2 * wave) * 256 + (g + 4 * r) * 16 + xrow] = acc0[r];
    }
    if (p0 + 2 * wave + 1 < B) {
#pragma unroll
        for (int r = 0; r < 4; ++r) gr->gram[(size_t)(p0 + 2 * wave + 1) * 256 + (g + 4 * r) * 16 + xrow] = acc1[r];
    }
    }
}

// the product and the epilogue, from 16 point records already in LDS (row stride `stride` doubles: basis [nb_pad] then mono [R][DL_FG_MONO_LD]);
// gfrag: [N_pad / 16][nb_pad / 8][19][64][2]; out: [B * R, ldo] (+= if accumulate); 512 threads, blockIdx.y = group of 8 column blocks
template <bool GRAM = false>
__device__ __forceinline__ void dl_fg_compute(const double* lds, int stride, int nb_pad, int R, const double* __restrict__ gfrag, double* __restrict__ out, int64_t ldo,
                                              int64_t B, int64_t p0, int accumulate, const DlFgGram* gr = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int jb = blockIdx.y * 8 + wave;               // 16-column block of this wave
    const int nq = nb_pad / 8;
    const dl_fg_double2* gw = reinterpret_cast<const dl_fg_double2*>(gfrag) + (size_t)jb * nq * DL_FG_NM * 64 + lane;
    const double* arow = lds + col * stride + 2 * g;                 // A operand of lane (point = col index of the lane, k group g)
    double outv[4][6];   // rows 0-5 of the four points of this lane, carried from the first monomial group to the second
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int u = 0; u < 6; ++u) outv[rr][u] = 0.;
    for (int mg = 0; mg < 2; ++mg) {
        if (mg == 1) { DL_FG_STAMP(3) }
        const int m0 = mg * DL_FG_MG;
        dl_fg_double4 acc[DL_FG_MG];
#pragma unroll
        for (int i = 0; i < DL_FG_MG; ++i) acc[i] = (dl_fg_double4){0., 0., 0., 0.};
        dl_fg_double2 bcur[DL_FG_MG], bnxt[DL_FG_MG];
        // monomial m0 + i, clamped to the last one in the short group (its accumulator is ignored): the load count is the same on every path
#pragma unroll
        for (int i = 0; i < DL_FG_MG; ++i) { int m = m0 + i < DL_FG_NM ? m0 + i : DL_FG_NM - 1; bcur[i] = gw[(size_t)(0 * DL_FG_NM + m) * 64]; }
        if (GRAM) {
            // TWO steps of the operand in flight (the carried rows live in LDS in this variant, which frees the registers for it): a 16-point workgroup has its CU
            // to itself, so the bytes in flight per wave are what the L2 round trip is divided by -- with one step (10 KB) the waves were parked half of their life
            dl_fg_double2 bnx2[DL_FG_MG];
            const int q1 = 1 < nq ? 1 : nq - 1;
#pragma unroll
            for (int i = 0; i < DL_FG_MG; ++i) { int m = m0 + i < DL_FG_NM ? m0 + i : DL_FG_NM - 1; bnxt[i] = gw[(size_t)(q1 * DL_FG_NM + m) * 64]; }
            for (int q = 0; q < nq; ++q) {
                const int qn = q + 2 < nq ? q + 2 : nq - 1;
#pragma unroll
                for (int i = 0; i < DL_FG_MG; ++i) { int m = m0 + i < DL_FG_NM ? m0 + i : DL_FG_NM - 1; bnx2[i] = gw[(size_t)(qn * DL_FG_NM + m) * 64]; }
                const dl_fg_double2 a = *reinterpret_cast<const dl_fg_double2*>(arow + 8 * q);
#pragma unroll
                for (int i = 0; i < DL_FG_MG; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bcur[i].x, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bcur[i].y, acc[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < DL_FG_MG; ++i) { bcur[i] = bnxt[i]; bnxt[i] = bnx2[i]; }
            }
        } else {
        for (int q = 0; q < nq; ++q) {
            const int qn = q + 1 < nq ? q + 1 : q;
#pragma unroll
            for (int i = 0; i < DL_FG_MG; ++i) { int m = m0 + i < DL_FG_NM ? m0 + i : DL_FG_NM - 1; bnxt[i] = gw[(size_t)(qn * DL_FG_NM + m) * 64]; }
            const dl_fg_double2 a = *reinterpret_cast<const dl_fg_double2*>(arow + 8 * q);
#pragma unroll
            for (int i = 0; i < DL_FG_MG; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bcur[i].x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bcur[i].y, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < DL_FG_MG; ++i) bcur[i] = bnxt[i];
        }
        }
        DL_FG_STAMP(2 + 2 * mg)
        // epilogue: accumulator register rr of lane (col, g) = U[point g + 4 rr][m][column jb * 16 + col]; contract with the monomial rows of that point.
        // The first six rows of a point are carried in registers across the two monomial groups and stored once (writing partial rows and adding to them
        // in the second group tripled the output traffic: 14 of 48 us at 6 rows per point); rows beyond six take the read-modify-write route, six at a time
        // with their old values requested together.
        if (GRAM) {
            // Gram variant (R <= 6, rows live in X): per point of this lane the ten monomials of three rows come from LDS in ONE batch of 16-byte reads, then three
            // independent chains of ten FMAs -- written with a load under `if (r < R && ...)` per row the compiler waited for every LDS round trip in turn
            // (epilogues 6.1 + 8.6 us of a 57 us workgroup life; the pad monomial of the short group is an exact zero in the record: no predicate on m).
            const int cbase = jb * 16 + col;
            double cst[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) cst[u] = mg == 1 ? gr->cst[u < R ? u : 0][cbase] : 0.;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int pt = g + 4 * rr;
                const double* mono = lds + pt * stride + nb_pad + m0;
                double* xb = gr->x + (size_t)pt * gr->xr * DL_FG_XLD + cbase;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    dl_fg_double2 mm[3][DL_FG_MG / 2];
                    double v[3];
#pragma unroll
                    for (int u3 = 0; u3 < 3; ++u3) {
                        const int u = 3 * h + u3, uc = u < R ? u : 0;
#pragma unroll
                        for (int j = 0; j < DL_FG_MG / 2; ++j) mm[u3][j] = *reinterpret_cast<const dl_fg_double2*>(mono + uc * DL_FG_MONO_LD + 2 * j);
                        v[u3] = xb[(size_t)gr->row_of[uc] * DL_FG_XLD];      // (mg == 0: whatever is there, not used)
                    }
#pragma unroll
                    for (int u3 = 0; u3 < 3; ++u3) {
                        double w = mg == 1 ? v[u3] : 0.;
#pragma unroll
                        for (int j = 0; j < DL_FG_MG / 2; ++j) {
                            w = fma(mm[u3][j].x, acc[2 * j][rr], w);
                            w = fma(mm[u3][j].y, acc[2 * j + 1][rr], w);
                        }
                        v[u3] = mg == 1 ? w + cst[3 * h + u3] : w;
                    }
#pragma unroll
                    for (int u3 = 0; u3 < 3; ++u3) {
                        const int u = 3 * h + u3;
                        if (u < R) xb[(size_t)gr->row_of[u] * DL_FG_XLD] = v[u3];
                    }
                }
            }
        } else
        for (int r0 = 0; r0 < R; r0 += 6) {
            const bool in_regs = (r0 == 0);
            double old[4][6];
            const bool rmw = in_regs ? (mg == 1 && accumulate) : (mg > 0 || accumulate);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int pt = g + 4 * rr, r = r0 + u;
                    old[rr][u] = (rmw && r < R && p0 + pt < B) ? out[((size_t)(p0 + pt) * R + r) * ldo + jb * 16 + col] : 0.;
                }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int pt = g + 4 * rr;
                const double* mono = lds + pt * stride + nb_pad;
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int r = r0 + u;
                    if (r < R && p0 + pt < B) {
                        double v;
                        v = (in_regs && mg == 1) ? outv[rr][u] + old[rr][u] : old[rr][u];
#pragma unroll
                        for (int i = 0; i < DL_FG_MG; ++i)
                            if (m0 + i < DL_FG_NM) v = fma(mono[r * DL_FG_MONO_LD + m0 + i], acc[i][rr], v);
                        if (in_regs && mg == 0) outv[rr][u] = v;
                        else out[((size_t)(p0 + pt) * R + r) * ldo + jb * 16 + col] = v;
                    }
                }
            }
        }
    }
    if (GRAM) {
        DL_FG_STAMP(5)
        __syncthreads();   // all rows of the 16 points are in LDS
        DL_FG_STAMP(6)
        dl_fg_gram_phase(gr, wave, lane, g, B, p0);
        DL_FG_STAMP(7)
    }
}

// ---- Gram variant, round 4: monomial groups of compile-time size, the two waves of a SIMD on DIFFERENT schedules -----------------------------------------------
// In-kernel stamps (config 3): the main loops run at the matrix pipe's rate (two waves per SIMD), but both waves of a SIMD reached their epilogues -- LDS reads and
// fp64 FMAs, no MFMA -- at the same time, twice: 7 us of a 48 us workgroup life with the matrix pipe idle, and the barrier before the Gram phase waited for the
// slower partner.  Waves 0-3 now take the monomials in groups of (8, 6, 5), waves 4-7 (their SIMD partners) in groups of (4, 8, 7): an epilogue of one meets a main
// loop of the other.  Every output still sums its monomials in the order 0..18 from zero (partial sums wait in X between groups): results are bit-identical.
// The operand registers rotate by unrolling the k loop three times (two steps in flight) instead of being copied (80 moves per step).
// the first two k-steps of a group's operand, requested AHEAD of its main loop (during the previous group's epilogue; for the first group before the barrier that
// follows the forward pass): a group is only nq (9) steps long, and started cold every group paid the L2 round trip with the matrix pipe idle whenever the SIMD
// partner was not streaming (the timelines of both waves of a SIMD: a wave alone reached 57 % of the pipe's rate)
// c0: starting value of the accumulators.  When the last basis function is the constant 1 (MLP engines: the bias row of the folded final layer) and it is the only
// live entry of the last k-step pair (n_basis = 8 j + 1), that pair is not multiplied at all: its product is the row G[n_basis - 1][m][column] itself, the same for
// every point -- the accumulators START from it (gbias: that pair of the operand at the lane's column, null otherwise) and the main loop is one pair (of nine: 11 % of
// the MFMAs and of the operand stream of config 3) shorter.
template <int CNT> struct DlFgAhead { dl_fg_double2 b0[CNT], b1[CNT]; double c0[CNT]; };
// gw: WAVE-UNIFORM base of the wave's operand (scalar registers), lane: the lane's element offset -- kept apart so that every request is a load with a scalar base,
// a constant vector offset and an immediate: addresses formed per lane were 1 - 2 vector instructions per load, and on this chip the fp64 MFMA runs on the vector
// unit's own fp64 lanes -- every vector instruction of the loop is added to its time, not hidden under it (docs/EXPERIMENTS.md).
template <int CNT>
__device__ __forceinline__ void dl_fg_gram_request(DlFgAhead<CNT>& ah, const dl_fg_double2* __restrict__ gw, unsigned lane, int nq, int m0, const dl_fg_double2* __restrict__ gbias, unsigned col) {
    const int q1 = 1 < nq ? 1 : nq - 1;
#pragma unroll
    for (int i = 0; i < CNT; ++i) ah.b0[i] = gw[(size_t)(m0 + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < CNT; ++i) ah.b1[i] = gw[(size_t)(q1 * DL_FG_NM + m0 + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < CNT; ++i) ah.c0[i] = gbias ? gbias[(size_t)(m0 + i) * 64 + col].x : 0.;
}
// D operand buffers, D - 1 steps in flight (the first two come from the request made ahead).  Two steps are enough while both waves of a SIMD stream MFMAs (each
// advances at half speed); the LAST group of a wave often runs with its partner already finished, and alone, two steps ahead, a wave reached 60 % of the pipe's rate
// (timeline of both waves: 104 cycles per MFMA instead of 64): the last groups run deeper (they hold no request for a next group: the registers are there).
template <int CNT, int D = 3>
__device__ __forceinline__ void dl_fg_gram_mainloop(const double* arow, const dl_fg_double2* __restrict__ gw, unsigned lane, int nq, int m0, dl_fg_double4 (&acc)[CNT], const DlFgAhead<CNT>& ah) {
    dl_fg_double2 b[D][CNT];
#define DL_FG_LOAD(bb, qq) { const int q_ = (qq) < nq ? (qq) : nq - 1; _Pragma("unroll") for (int i = 0; i < CNT; ++i) bb[i] = gw[(size_t)(q_ * DL_FG_NM + m0 + i) * 64 + lane]; }
#define DL_FG_MUL(bb, qq) { const dl_fg_double2 a_ = *reinterpret_cast<const dl_fg_double2*>(arow + 8 * (qq)); \
        _Pragma("unroll") for (int i = 0; i < CNT; ++i) { acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_.x, bb[i].x, acc[i], 0, 0, 0); \
                                                          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_.y, bb[i].y, acc[i], 0, 0, 0); } }
#pragma unroll
    for (int i = 0; i < CNT; ++i) { acc[i] = (dl_fg_double4){ah.c0[i], ah.c0[i], ah.c0[i], ah.c0[i]}; b[0][i] = ah.b0[i]; b[1][i] = ah.b1[i]; }
#pragma unroll
    for (int d = 2; d < D - 1; ++d) DL_FG_LOAD(b[d], d)
    int q = 0;
    for (; q + D <= nq; q += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) { DL_FG_LOAD(b[(d + D - 1) % D], q + d + D - 1) DL_FG_MUL(b[d], q + d) }
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (q + d < nq) { DL_FG_MUL(b[d], q + d) }
#undef DL_FG_LOAD
#undef DL_FG_MUL
}

// Row 0 of the monomials NO derivative row touches (velocileptors order: 0-11 = 1, b1 .. b1 b3; the solved alpha* / sn* live on 12-18): their epilogue is four FMAs per
// monomial and point into REGISTERS (w0: the lane's four points) -- no X row is read or written, no derivative row is looked at.  The X rows are written once, by the
// epilogue of monomials 12-18 (first and last at once); w0 is added to row 0 at the very end.
template <int CNT>
__device__ __forceinline__ void dl_fg_gram_epilogue_row0(const dl_fg_double4 (&acc)[CNT], const double* lds, int stride, int nb_pad, int m0, int g, double (&w0)[4]) {
    constexpr int NP = (CNT + 1) / 2;
    dl_fg_double2 mm[4][NP];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const double* mono = lds + (g + 4 * rr) * stride + nb_pad + m0;
#pragma unroll
        for (int j = 0; j < NP; ++j) mm[rr][j] = *reinterpret_cast<const dl_fg_double2*>(mono + 2 * j);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            w0[rr] = fma(mm[rr][j].x, acc[2 * j][rr], w0[rr]);
            if (2 * j + 1 < CNT) w0[rr] = fma(mm[rr][j].y, acc[2 * j + 1][rr], w0[rr]);
        }
    }
}

// The same main loop for nq = 8 (config 3 after the constant basis function left the loop), completely unrolled, the A operand of the eight steps in registers for the
// whole phase (it is the same for every group: one batch of LDS reads per wave instead of one read, and one wait, per step).  Unrolled, every use of an operand
// register waits for exactly the loads it needs; around the back-edge of the rolled loop the compiler waited for ALL outstanding loads (`s_waitcnt vmcnt(0)`) at
// the top of every trip -- the wave streamed with one step in flight instead of D - 1, which its SIMD partner hides while it streams too, and nobody hides while the
// partner sits in an epilogue (a wave alone: 60 % of the pipe's rate).
template <int CNT, int D = 3>
__device__ __forceinline__ void dl_fg_gram_mainloop_u8(const dl_fg_double2 (&areg)[8], const dl_fg_double2* __restrict__ gw, unsigned lane, int m0, dl_fg_double4 (&acc)[CNT],
                                                       const DlFgAhead<CNT>& ah) {
    dl_fg_double2 b[D][CNT];
#define DL_FG_LOAD(bb, qq) { _Pragma("unroll") for (int i = 0; i < CNT; ++i) bb[i] = gw[(size_t)((qq) * DL_FG_NM + m0 + i) * 64 + lane]; }
#pragma unroll
    for (int i = 0; i < CNT; ++i) { acc[i] = (dl_fg_double4){ah.c0[i], ah.c0[i], ah.c0[i], ah.c0[i]}; b[0][i] = ah.b0[i]; b[1][i] = ah.b1[i]; }
#pragma unroll
    for (int d = 2; d < D - 1; ++d) DL_FG_LOAD(b[d], d)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (q + D - 1 < 8) DL_FG_LOAD(b[(q + D - 1) % D], q + D - 1)
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[q].x, b[q % D][i].x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[q].y, b[q % D][i].y, acc[i], 0, 0, 0);
        }
    }
#undef DL_FG_LOAD
}

// rows of the lane's four points += sum over the group's monomials (m0 even); first: the partial rows start from zero; last: the constant part of every row is added.
// Row 0 (the residual) is dense in the monomials.  The derivative rows are not: the row of a solved alpha* / sn* touches one or two monomials (gr->nz), the other
// seventeen entries of its monomial row are exact zeros -- multiplying them was 4/5 of the epilogue's LDS reads and FMAs (7 us of the kernel; the LDS pipe of the CU
// was the bound: 240 16-byte broadcast reads per wave).  The sums run over the same monomials in the same order: bit-identical results.
template <int CNT>
__device__ __forceinline__ void dl_fg_gram_epilogue(const dl_fg_double4 (&acc)[CNT], const double* lds, int stride, int nb_pad, int R, int m0, bool first, bool last,
                                                    const DlFgGram* gr, int cbase, int g, const double (&cst)[6]) {
    constexpr int NP = (CNT + 1) / 2;
    double* xb0 = gr->x + (size_t)g * gr->xr * DL_FG_XLD + cbase;                    // X rows of point g (+ 4 rr: 4 xr DL_FG_XLD doubles further)
    const size_t xpt = (size_t)4 * gr->xr * DL_FG_XLD;
    {   // row 0
        dl_fg_double2 mm[4][NP];
        double v[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const double* mono = lds + (g + 4 * rr) * stride + nb_pad + m0;
#pragma unroll
            for (int j = 0; j < NP; ++j) mm[rr][j] = *reinterpret_cast<const dl_fg_double2*>(mono + 2 * j);
            v[rr] = xb0[rr * xpt + (size_t)gr->row_of[0] * DL_FG_XLD];               // (first group: whatever is there, not used)
        }
        const double c0 = cst[0];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            double w = first ? 0. : v[rr];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                w = fma(mm[rr][j].x, acc[2 * j][rr], w);
                if (2 * j + 1 < CNT) w = fma(mm[rr][j].y, acc[2 * j + 1][rr], w);
            }
            xb0[rr * xpt + (size_t)gr->row_of[0] * DL_FG_XLD] = last ? w + c0 : w;
        }
    }
    // derivative rows, in two batches of rows (1-3, 4-5; register budget: the next group's operand is already in flight), three passes per batch -- every LDS read
    // first (the rows a monomial of this group feeds, or all of them in the last group), then the FMAs and the constant parts, then the writes: row by row each read
    // was waited for in turn (4.4 us for the last group)
#pragma unroll
    for (int u0 = 1; u0 < 6; u0 += 3) {
        double xr_[3][4], dr_[3][2][4];
        bool hit[3][2], touch[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = u0 + k;
            if (u >= 6) { touch[k] = hit[k][0] = hit[k][1] = false; continue; }
#pragma unroll
            for (int z = 0; z < 2; ++z) { const int m = gr->nz[u][z]; hit[k][z] = u < R && m >= m0 && m < m0 + CNT; }
            touch[k] = u < R && (first || last || hit[k][0] || hit[k][1]);
            if (touch[k]) {
                const double* xu = xb0 + (size_t)gr->row_of[u] * DL_FG_XLD;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) xr_[k][rr] = xu[rr * xpt];            // (first group: whatever is there, not used)
#pragma unroll
                for (int z = 0; z < 2; ++z)
                    if (hit[k][z]) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) dr_[k][z][rr] = lds[(g + 4 * rr) * stride + nb_pad + u * DL_FG_MONO_LD + gr->nz[u][z]];
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = u0 + k;
            if (u >= 6 || !touch[k]) continue;
            if (first) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) xr_[k][rr] = 0.;
            }
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                if (!hit[k][z]) continue;
                const int m = gr->nz[u][z];
#pragma unroll
                for (int i = 0; i < CNT; ++i)
                    if (m == m0 + i) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) xr_[k][rr] = fma(dr_[k][z][rr], acc[i][rr], xr_[k][rr]);
                    }
            }
            if (last) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) xr_[k][rr] += cst[u];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = u0 + k;
            if (u >= 6 || !touch[k]) continue;
            double* xu = xb0 + (size_t)gr->row_of[u] * DL_FG_XLD;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) xu[rr * xpt] = xr_[k][rr];
        }
    }
}

// 16 point records in LDS -> G = X X^T of the 16 points (R <= 6 device rows, N_pad = 128, 512 threads).  Called BEFORE the barrier that completes the records: the
// barrier is inside, after the first operand request
template <class AfterRequest>
__device__ __forceinline__ void dl_fg_compute_gram(const double* lds, int stride, int nb_pad, int R, const double* __restrict__ gfrag, int64_t B, int64_t p0, const DlFgGram* gr,
                                                   AfterRequest&& after_request, bool bias_pair = false) {
    constexpr bool GRAM = true;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the operand base of the wave and the team branch)
    const int col = lane & 15, g = lane >> 4;
    const int jb = wave;                                // 16-column block of this wave
    const int nq_all = nb_pad / 8;
    const dl_fg_double2* gw = reinterpret_cast<const dl_fg_double2*>(gfrag) + (size_t)jb * nq_all * DL_FG_NM * 64;   // wave-uniform (jb is a scalar)
    const unsigned ulane = (unsigned)lane, ucol = (unsigned)col;
    // (bias_pair: see DlFgAhead -- the last pair of the operand holds the constant basis function alone; lanes 0-15 of that pair carry k = n_basis - 1)
    const int nq = bias_pair ? nq_all - 1 : nq_all;
    const dl_fg_double2* gbias = bias_pair ? gw + (size_t)(nq_all - 1) * DL_FG_NM * 64 : nullptr;
    const double* arow = lds + col * stride + 2 * g;
    const int cbase = jb * 16 + col;
    double cst[6] = {0., 0., 0., 0., 0., 0.};   // constant parts of the rows (used by the last epilogue): requested before the last main loop, which hides the round trip
#define DL_FG_CST _Pragma("unroll") for (int u = 0; u < 6; ++u) cst[u] = gr->cst[u < R ? u : 0][cbase];
    // monomial groups (8, 6, 5) on waves 0-3, (4, 8, 7) on their SIMD partners -- group starts are even (16-byte reads of the monomial rows); the first two steps of
    // every group's operand are requested before the epilogue of the group before it
    if (gr->scaled) {
        // groups L1 = monomials 0-5, L2 = 6-11 (row 0 only, into registers), T = 12-18 (the X rows, first and last at once).  Waves 0-3: T, L1, L2; their SIMD partners:
        // L1, T, L2 -- the one long epilogue of a wave runs beside a main loop of its partner, and the last thing either does is a short one
        double w0[4] = {0., 0., 0., 0.};
        dl_fg_double2 areg[8];   // A operand of the eight steps: read before every main loop (not kept across the long epilogue: registers)
#define DL_FG_AREG { _Pragma("unroll") for (int q = 0; q < 8; ++q) areg[q] = *reinterpret_cast<const dl_fg_double2*>(arow + 8 * q); }
#define DL_FG_MAIN(CNT_, D_, M0_, ACC_, AH_) { DL_FG_AREG dl_fg_gram_mainloop_u8<CNT_, D_>(areg, gw, ulane, M0_, ACC_, AH_); }   // (nq = 8: the launcher's condition for this path)
        DlFgAhead<6> l1, l2;
        DlFgAhead<7> t;
        if (wave < 4) {
            dl_fg_gram_request<7>(t, gw, ulane, nq, 12, gbias, ucol);
            after_request();   // (barrier: the records of the forward pass are complete)
            { dl_fg_double4 acc[7]; DL_FG_MAIN(7, 3, 12, acc, t) DL_FG_STAMP(2) DL_FG_CST
              dl_fg_gram_epilogue<7>(acc, lds, stride, nb_pad, R, 12, true, true, gr, cbase, g, cst); }
            dl_fg_gram_request<6>(l1, gw, ulane, nq, 0, gbias, ucol);   // (after the long epilogue: the registers it needs are those of a request)
            DL_FG_STAMP(3)
            { dl_fg_double4 acc[6]; DL_FG_MAIN(6, 3, 0, acc, l1) dl_fg_gram_request<6>(l2, gw, ulane, nq, 6, gbias, ucol);
              dl_fg_gram_epilogue_row0<6>(acc, lds, stride, nb_pad, 0, g, w0); }
        } else {
            dl_fg_gram_request<6>(l1, gw, ulane, nq, 0, gbias, ucol);
            after_request();
            { dl_fg_double4 acc[6]; DL_FG_MAIN(6, 3, 0, acc, l1) dl_fg_gram_request<7>(t, gw, ulane, nq, 12, gbias, ucol);
              dl_fg_gram_epilogue_row0<6>(acc, lds, stride, nb_pad, 0, g, w0); }
            { dl_fg_double4 acc[7]; DL_FG_MAIN(7, 3, 12, acc, t) DL_FG_CST
              dl_fg_gram_epilogue<7>(acc, lds, stride, nb_pad, R, 12, true, true, gr, cbase, g, cst); }
            dl_fg_gram_request<6>(l2, gw, ulane, nq, 6, gbias, ucol);
        }
        { dl_fg_double4 acc[6]; DL_FG_MAIN(6, 4, 6, acc, l2) DL_FG_STAMP(4) dl_fg_gram_epilogue_row0<6>(acc, lds, stride, nb_pad, 6, g, w0); }
        {   // row 0 of the lane's four points: what the T epilogue wrote + the register part
            double* xr0 = gr->x + (size_t)g * gr->xr * DL_FG_XLD + cbase + (size_t)gr->row_of[0] * DL_FG_XLD;
            const size_t xpt = (size_t)4 * gr->xr * DL_FG_XLD;
            double v[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) v[rr] = xr0[rr * xpt];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) xr0[rr * xpt] = v[rr] + w0[rr];
        }
#undef DL_FG_AREG
#undef DL_FG_MAIN
    } else if (wave < 4) {
        DlFgAhead<8> a0; DlFgAhead<6> a1; DlFgAhead<5> a2;
        dl_fg_gram_request<8>(a0, gw, ulane, nq, 0, gbias, ucol);
        after_request();   // (barrier: the records of the forward pass are complete)
        { dl_fg_double4 acc[8]; dl_fg_gram_mainloop<8>(arow, gw, ulane, nq, 0, acc, a0); DL_FG_STAMP(2) dl_fg_gram_request<6>(a1, gw, ulane, nq, 8, gbias, ucol); dl_fg_gram_epilogue<8>(acc, lds, stride, nb_pad, R, 0, true, false, gr, cbase, g, cst); }
        DL_FG_STAMP(3)
        { dl_fg_double4 acc[6]; dl_fg_gram_mainloop<6>(arow, gw, ulane, nq, 8, acc, a1); dl_fg_gram_request<5>(a2, gw, ulane, nq, 14, gbias, ucol); DL_FG_CST dl_fg_gram_epilogue<6>(acc, lds, stride, nb_pad, R, 8, false, false, gr, cbase, g, cst); }
        { dl_fg_double4 acc[5]; dl_fg_gram_mainloop<5, 5>(arow, gw, ulane, nq, 14, acc, a2); DL_FG_STAMP(4) dl_fg_gram_epilogue<5>(acc, lds, stride, nb_pad, R, 14, false, true, gr, cbase, g, cst); }
    } else {
        DlFgAhead<4> a0; DlFgAhead<8> a1; DlFgAhead<7> a2;
        dl_fg_gram_request<4>(a0, gw, ulane, nq, 0, gbias, ucol);
        after_request();
        { dl_fg_double4 acc[4]; dl_fg_gram_mainloop<4>(arow, gw, ulane, nq, 0, acc, a0); dl_fg_gram_request<8>(a1, gw, ulane, nq, 4, gbias, ucol); dl_fg_gram_epilogue<4>(acc, lds, stride, nb_pad, R, 0, true, false, gr, cbase, g, cst); }
        { dl_fg_double4 acc[8]; dl_fg_gram_mainloop<8>(arow, gw, ulane, nq, 4, acc, a1); dl_fg_gram_request<7>(a2, gw, ulane, nq, 12, gbias, ucol); DL_FG_CST dl_fg_gram_epilogue<8>(acc, lds, stride, nb_pad, R, 4, false, false, gr, cbase, g, cst); }
        { dl_fg_double4 acc[7]; dl_fg_gram_mainloop<7, 4>(arow, gw, ulane, nq, 12, acc, a2); dl_fg_gram_epilogue<7>(acc, lds, stride, nb_pad, R, 12, false, true, gr, cbase, g, cst); }
    }
#undef DL_FG_CST
    DL_FG_STAMP(5)
    __syncthreads();   // all rows of the 16 points are in LDS
    DL_FG_STAMP(6)
    dl_fg_gram_phase(gr, wave, lane, g, B, p0);
    DL_FG_STAMP(7)
}

// feat: [B, feat_ld] point records written by the theory kernel, this observable's record at column feat_off
__global__ __launch_bounds__(512) void dl_feature_gemm_kernel(const double* __restrict__ feat, int64_t feat_ld, int64_t feat_off, int nb_pad, int R,
                                                              const double* __restrict__ gfrag, double* __restrict__ out, int64_t ldo, int64_t B, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * DL_FG_PTS;
    const int rec_len = nb_pad + R * DL_FG_MONO_LD;
    const int stride = dl_fg_lds_stride(rec_len);
    // stage the 16 point records (rows beyond B repeat the last point; their outputs are not stored)
    for (int idx = tid; idx < DL_FG_PTS * rec_len; idx += 512) {
        int pt = idx / rec_len, c = idx - pt * rec_len;
        int64_t b = p0 + pt < B ? p0 + pt : B - 1;
        lds[pt * stride + c] = feat[(size_t)b * feat_ld + feat_off + c];
    }
    __syncthreads();
    dl_fg_compute(lds, stride, nb_pad, R, gfrag, out, ldo, B, p0, accumulate);
}
