// Eigen models with more than 32 states (CUSTOM_RATES of any size up to 256): P(t) of every branch on the FP64 matrix cores.
//
//   Pt_b[j][i] = sum_m (Ainv[m][j] exp(d_m t_b)) A[i][m]         (pastml/models/generator.py:54-65, CustomRatesModel.py:70-79)
//
// Beyond 64 states every sweep of an eigen model reads P(t) of every branch (between 33 and 64 only pml_pij_batch and
// pml_marginal_counts ask for it), and until round 6 a one-thread-per-entry kernel built it: 2 k^3 flops per branch through
// k^2 chains of k dependent FMAs whose operands came from the caches -- 700 ms for 16 384 tips x 4 characters at k = 128, 1 % of
// the FP64 peak, and all of a likelihood evaluation (profiles/r06u_pij_wide.txt).  The batch is a tall GEMM, as for 16 <= k <= 32
// (pij_eigen_mfma_kernel): the rows (b, j) of all branches times A^T, whose tiles are the B operands of v_mfma_f64_16x16x4_f64
// and the same for every branch; the A operand is formed on the fly, one multiplication per MFMA step.  What changes with k is
// where the operands live: A^T no longer fits in registers (k^2 doubles), so a workgroup keeps a slice of it -- all k rows,
// 16 NTC columns, at most 128 KB -- in LDS and walks its branches once per slice; Ainv comes through the caches, four K steps
// ahead of the MFMAs that use it.  A wave owns a branch at a time: exp(d_m t_b) once per branch and slice, then one 16-row
// tile of the result after the other, NTC accumulator tiles wide.  2 k^3 / (8 k^2) = k / 4 flop per byte written: above 64
// states the batch is bound by the matrix cores, not by its stores, which therefore go out straight from the accumulators
// (128-byte row pieces).  Same operations in the same order as pij_eigen_mfma_kernel: (Ainv e) A, K ascending.
#pragma once
#include "pml_kernels_pij.h"

#define PML_PIJW_BLOCK 512                     // 8 waves: two per SIMD at one workgroup per CU
#define PML_PIJW_WAVES (PML_PIJW_BLOCK / 64)
#define PML_PIJW_AHEAD 4                       // K steps whose Ainv operands are in flight

// column tiles per slice for k states: the slice [KP][16 NTC] must fit 128 KB; the NT tiles in as few slices as that allows, of
// equal width (every slice runs all NTC accumulator tiles -- a tile beyond the matrix multiplies zeros: the last slice of 13
// tiles in widths of 5 wastes two of fifteen --: a matrix instruction under a condition costs the compiler a copy of every
// accumulator, ~100 moves per instruction when it was tried)
static inline int pijw_tiles(int k, int ks) {
    const int KP = (k + 3) & ~3, NT = (ks + 15) / 16;
    int most = 1024 / KP;
    if (most > 8) most = 8;
    if (most < 1) most = 1;
    const int slices = (NT + most - 1) / most;
    return (NT + slices - 1) / slices;
}
static inline size_t pijw_lds_bytes(int k, int ntc) {
    const int KP = (k + 3) & ~3;
    return ((size_t)KP * 16 * ntc + (size_t)PML_PIJW_WAVES * KP) * sizeof(double);
}

template <int NTC>
__global__ void __launch_bounds__(PML_PIJW_BLOCK)
pij_eigen_wide_kernel(PmlTree t, PmlCols c, PmlModel m, double* __restrict__ P, int branches_per_block) {
    extern __shared__ double pijw_smem[];
    const int k = c.k, ks = c.ks;
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    const int KP = (k + 3) & ~3, KS = KP >> 2;
    constexpr int LDR = 16 * NTC;
    double* sR = pijw_smem;                                 // [KP][LDR]  sR[m][ii] = A[i_base + ii][m], zero outside
    double* sE = sR + (size_t)KP * LDR + (size_t)wave * KP; // per wave: exp(d_m t_b), zero for m >= k
    const double* gA = m.A + (size_t)col * k * k;
    const double* gB = m.Ainv + (size_t)col * k * k;
    const double* gd = m.d + (size_t)col * k;
    const double sfc = m.sf[col], tau = m.tau[col], tf = m.tauf[col];
    const int NT = (ks + 15) >> 4, JT = (k + 15) >> 4;
    const int b_lo = blockIdx.x * branches_per_block;
    const int b_hi = min(t.N, b_lo + branches_per_block);
    for (int nt0 = 0; nt0 < NT; nt0 += NTC) {
        const int i_base = 16 * nt0;
        __syncthreads();   // the previous slice has been consumed
        // (read along m: consecutive addresses; the transposing LDS writes are paid once per slice and workgroup)
        for (int e = threadIdx.x; e < LDR * KP; e += PML_PIJW_BLOCK) {
            const int ii = e / KP, mm = e - ii * KP;
            const int i = i_base + ii;
            sR[(size_t)mm * LDR + ii] = (i < k && mm < k) ? gA[(size_t)i * k + mm] : 0.0;
        }
        __syncthreads();
        for (int b = b_lo + wave; b < b_hi; b += PML_PIJW_WAVES) {
            const double tq = (t.dist[b] + tau) * tf * sfc;
            for (int mm = lane; mm < KP; mm += 64) sE[mm] = mm < k ? exp(gd[mm] * tq) : 0.0;
            wave_lds_sync();
            double* const out_b = P + (colN + b) * (size_t)k * ks;
            for (int jt = 0; jt < JT; ++jt) {
                // rows j of this tile; rows beyond k (the last tile) are computed from row k - 1 and never written
                const int jc = min(16 * jt + lo, k - 1);
                const double* pa = gB + jc;   // Ainv[m][jc] = pa[m k]
                pml_v4f64 acc[NTC];
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt) acc[nt] = (pml_v4f64){0.0, 0.0, 0.0, 0.0};
                // K steps in groups of PML_PIJW_AHEAD, the operands of the next group in flight; no matrix instruction under a condition
                // (see pijw_tiles); the last KS % PML_PIJW_AHEAD steps in a loop of their own
                double abuf[PML_PIJW_AHEAD], anext[PML_PIJW_AHEAD];
#pragma unroll
                for (int u = 0; u < PML_PIJW_AHEAD; ++u) abuf[u] = pa[(size_t)min(4 * u + hi, k - 1) * k];
                int s0 = 0;
                for (; s0 + PML_PIJW_AHEAD <= KS; s0 += PML_PIJW_AHEAD) {
                    // (rows beyond the matrix: row k - 1, multiplied by a zero exponential or never used)
#pragma unroll
                    for (int u = 0; u < PML_PIJW_AHEAD; ++u)
                        anext[u] = pa[(size_t)min(4 * (s0 + PML_PIJW_AHEAD + u) + hi, k - 1) * k];
#pragma unroll
                    for (int u = 0; u < PML_PIJW_AHEAD; ++u) {
                        const int mm = 4 * (s0 + u) + hi;
                        const double a = abuf[u] * sE[mm];
                        const double* pr = sR + (size_t)mm * LDR + lo;
#pragma unroll
                        for (int nt = 0; nt < NTC; ++nt)
                            acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pr[16 * nt], acc[nt], 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < PML_PIJW_AHEAD; ++u) abuf[u] = anext[u];
                }
#pragma unroll 1
                for (int s = s0; s < KS; ++s) {
                    const int mm = 4 * s + hi;
                    const double a = pa[(size_t)min(mm, k - 1) * k] * sE[mm];
                    const double* pr = sR + (size_t)mm * LDR + lo;
#pragma unroll
                    for (int nt = 0; nt < NTC; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pr[16 * nt], acc[nt], 0, 0, 0);
                }
                // D: row = hi + 4 reg, column = lo
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int j = 16 * jt + hi + 4 * reg;
                    if (j < k) {
                        double* orow = out_b + (size_t)j * ks + i_base + lo;
#pragma unroll
                        for (int nt = 0; nt < NTC; ++nt)
                            if (i_base + 16 * nt + lo < ks) orow[16 * nt] = acc[nt][reg];   // (columns k .. ks - 1: exact zeros)
                    }
                }
            }
            wave_lds_sync();   // the exponentials have been read before the next branch overwrites them
        }
    }
}
