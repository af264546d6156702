// Per-branch transition matrices P(t) = exp(Q t').
#pragma once
#include "pml_kernels_matrix.h"

// HKY: one thread per (branch, column); stores the transposed 4x4 (row stride ks = 4)
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
pij_hky_kernel(PmlTree t, PmlCols c, PmlModel m, double* __restrict__ P) {
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const double* pi = c.pi + (size_t)col * c.ks;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < t.N; n += gridDim.x * blockDim.x) {
        const double tt = (t.dist[n] + m.tau[col]) * m.tauf[col] * m.sf[col];
        double p[4][4];
        hky_matrix(pi, m.kappa[col], tt, p);
        double* out = P + (colN + n) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[j * 4 + i] = p[i][j];
    }
}
#endif

// Eigen models: P = A diag(exp(d t')) Ainv (pastml/models/generator.py:54-65).
// One workgroup walks a chunk of branches of one column; A and Ainv^T are staged once in LDS (rows padded by one
// double against bank conflicts), exp(d t') per branch in LDS; thread e computes output element (i = e % k, j = e / k)
// so that the transposed store Pt[j][i] is coalesced.
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
pij_eigen_kernel(PmlTree t, PmlCols c, PmlModel m, double* __restrict__ P, int branches_per_block, int use_lds) {
    extern __shared__ double smem[];
    const int k = c.k, ks = c.ks;
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const int ldk = k + 1;
    double* sA = smem;                                // [k][k+1]   A[i][m]
    double* sBt = sA + (use_lds ? k * ldk : 0);       // [k][k+1]   Ainv^T[j][m] = Ainv[m][j]
    double* sE = sBt + (use_lds ? k * ldk : 0);       // [k]
    const double* gA = m.A + (size_t)col * k * k;
    const double* gB = m.Ainv + (size_t)col * k * k;
    const double* gd = m.d + (size_t)col * k;
    if (use_lds) {
        for (int e = threadIdx.x; e < k * k; e += blockDim.x) {
            const int r = e / k, q = e % k;
            sA[r * ldk + q] = gA[e];
            sBt[q * ldk + r] = gB[e];  // gB[r][q] = Ainv[m=r][j=q]
        }
    }
    const double sfc = m.sf[col], tau = m.tau[col], tf = m.tauf[col];
    const int b0 = blockIdx.x * branches_per_block;
    const int b1 = min(t.N, b0 + branches_per_block);
    for (int n = b0; n < b1; ++n) {
        const double tt = (t.dist[n] + tau) * tf * sfc;
        __syncthreads();
        for (int q = threadIdx.x; q < k; q += blockDim.x) sE[q] = exp(gd[q] * tt);
        __syncthreads();
        double* out = P + (colN + n) * (size_t)k * ks;
        for (int e = threadIdx.x; e < k * k; e += blockDim.x) {
            const int i = e % k, j = e / k;
            double acc = 0.0;
            if (use_lds) {
                const double* a = sA + i * ldk;
                const double* b = sBt + j * ldk;
                for (int q = 0; q < k; ++q) acc += (a[q] * sE[q]) * b[q];
            } else {
                for (int q = 0; q < k; ++q) acc += (gA[i * k + q] * sE[q]) * gB[q * k + j];
            }
            out[(size_t)j * ks + i] = acc;
        }
        if (ks > k) {
            const int pad = ks - k;
            for (int e = threadIdx.x; e < k * pad; e += blockDim.x) out[(size_t)(e / pad) * ks + k + e % pad] = 0.0;
        }
    }
}
#endif

// Explicit row-major P for a list of branch lengths (API get_Pij_t, tests): one thread per output element.
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
pij_explicit_kernel(PmlCols c, PmlModel m, int col, int n_t, const double* __restrict__ ts, double* __restrict__ out) {
    const int k = c.k;
    const size_t total = (size_t)n_t * k * k;
    const double* pi = c.pi + (size_t)col * c.ks;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % k);
        const int i = (int)((e / k) % k);
        const size_t b = e / ((size_t)k * k);
        const double tt = (ts[b] + m.tau[col]) * m.tauf[col] * m.sf[col];
        double v;
        if (m.kind == 0) {
            const double mu = m.mu[col];
            const double ex = isinf(mu) ? 0.0 : exp(-mu * tt);
            v = (1.0 - ex) * pi[j] + (i == j ? ex : 0.0);
        } else if (m.kind == 1) {
            double p[4][4];
            hky_matrix(pi, m.kappa[col], tt, p);
            v = p[i][j];
        } else {
            const double* A = m.A + (size_t)col * k * k;
            const double* B = m.Ainv + (size_t)col * k * k;
            const double* d = m.d + (size_t)col * k;
            v = 0.0;
            for (int q = 0; q < k; ++q) v += (A[i * k + q] * exp(d[q] * tt)) * B[q * k + j];
        }
        out[e] = v;
    }
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Eigen models, 16 <= k <= 32: the P(t) batch as a tall GEMM on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).
//
//   Pt_b[j][i] = sum_m (Ainv[m][j] * exp(d_m t_b)) * A[i][m]
//
// Stacking the branches of a chunk along the rows gives  [rows (b, j)] x [k]  times  A^T [k] x [i]: the right factor is
// shared by all branches (held in registers as B fragments), the left factor is formed on the fly from Ainv (LDS) and
// the chunk's exp(d_m t_b) (LDS, computed once per branch and eigenvalue).  K = k rounded up to 4 (k = 20: exact),
// N = k rounded up to 16; rows are processed 16 at a time.  The 16x16 result tile has the output state i along the
// lanes, so the transposed store Pt[b][j][16 consecutive i] is one 128-byte segment per row.
// Operand layouts (cdna_hip_programming.md section 3): A[l & 15][l >> 4], B[l >> 4][l & 15],
// D: col = l & 15, row = (l >> 4) + 4 * reg.
// ---------------------------------------------------------------------------------------------------------------------
#define PML_MFMA_CHUNK 16  // branches per wave pass

typedef double pml_v4f64 __attribute__((ext_vector_type(4)));

// Rows of the result a wave collects in LDS before it writes them out.  In the 16x16 result tile a store instruction
// would cover four row pieces of 128 bytes (and, for the second column tile of k = 20, of 32 bytes): the L2 has to merge
// partial lines and the batch ran at 3.3 TB/s (k = 20).  The rows (b, j) of a chunk are contiguous in memory (row stride
// ks doubles), so the wave stages SROWS of them in LDS in memory order and writes them with 16 bytes per lane
// in address order -- every store instruction covers 1 KB of consecutive bytes.
// SROWS: 16 (one result tile), 32 or 64 -- LDS per wave against bytes per flush (PASTML_HIP_PIJ_STAGE_ROWS).
template <int KS, int SROWS>
struct PijStage {
    static constexpr int KP = 4 * KS;
    static constexpr int ROWS = SROWS;
    static constexpr int DOUBLES = ROWS * KP;
    static constexpr int WAVE_DOUBLES = DOUBLES + 64;   // + a slot per lane for the columns beyond ks
};

template <int NT, int KS, int SROWS>
__global__ void __launch_bounds__(PML_BLOCK)
pij_eigen_mfma_kernel(PmlTree t, PmlCols c, PmlModel m, double* __restrict__ P) {
    extern __shared__ double smem[];
    const int k = c.k, ks = c.ks;
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    constexpr int KP = KS * 4;
    typedef PijStage<KS, SROWS> ST;
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    double* sB = smem;                                  // Ainv padded: [KP][k]   (rows >= k are zero)
    double* sE = sB + KP * k + wave * (PML_MFMA_CHUNK * KP + ST::WAVE_DOUBLES);  // per wave: exp(d_m t_b)  [CHUNK][KP]
    double* sO = sE + PML_MFMA_CHUNK * KP;              // per wave: staged rows of the result [ROWS][ks]
    const double* gA = m.A + (size_t)col * k * k;
    const double* gB = m.Ainv + (size_t)col * k * k;
    const double* gd = m.d + (size_t)col * k;
    for (int e = threadIdx.x; e < KP * k; e += blockDim.x) sB[e] = (e / k < k) ? gB[e] : 0.0;
    // B fragments of A^T: bfrag[nt][s] = A^T[4s + hi][16nt + lo] = A[16nt + lo][4s + hi]
    double bfrag[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int i = 16 * nt + lo, mm = 4 * s + hi;
            bfrag[nt][s] = (i < k && mm < k) ? gA[i * k + mm] : 0.0;
        }
    __syncthreads();
    const double sfc = m.sf[col], tau = m.tau[col], tf = m.tauf[col];
    const int waves_total = gridDim.x * PML_WAVES_PER_BLOCK;
    for (int b0 = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * PML_MFMA_CHUNK; b0 < t.N;
         b0 += waves_total * PML_MFMA_CHUNK) {
        const int nb = min(PML_MFMA_CHUNK, t.N - b0);
        // exp(d_m t_b) for the chunk: CHUNK * KP values over 64 lanes
        for (int e = lane; e < PML_MFMA_CHUNK * KP; e += 64) {
            const int q = e / KP, mm = e % KP;
            double v = 0.0;
            if (q < nb && mm < k) v = exp(gd[mm] * ((t.dist[b0 + q] + tau) * tf * sfc));
            sE[e] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int rows = nb * k;
        double* const chunk_out = P + (colN + b0) * (size_t)k * ks;   // rows (b, j) of the chunk, ks doubles each
        // A operand row of this lane in the current tile: row r0 + lo = (branch qa, state ja), advanced by 16 rows per tile
        // without a division (16 <= k: at most one wrap per step).  Rows beyond the chunk's last (a short last chunk) are
        // computed from the zero rows of the exp table and never written.
        int qa = 0, ja = lo;
        if (ja >= k) {
            ja -= k;
            qa = 1;
        }
        for (int g0 = 0; g0 < rows; g0 += ST::ROWS) {
            const int grows = min(ST::ROWS, rows - g0);
#pragma unroll
            for (int tt = 0; tt < ST::ROWS / 16; ++tt) {
                if (g0 + 16 * tt >= rows) break;
                // the left factor of the tile's rows: Ainv[m][ja] exp(d_m t_qa), all K steps read before the first product
                double a[KS];
                {
                    const double* pb = sB + ja;
                    const double* pe = sE + min(qa, PML_MFMA_CHUNK - 1) * KP;
#pragma unroll
                    for (int s = 0; s < KS; ++s) a[s] = pb[(4 * s + hi) * k] * pe[4 * s + hi];
                }
                ja += 16;
                if (ja >= k) {
                    ja -= k;
                    ++qa;
                }
                pml_v4f64 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = (pml_v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], bfrag[nt][s], acc[nt], 0, 0, 0);
                    }
                }
                // D: row = hi + 4 * reg, col = lo  ->  staged row 16 tt + hi + 4 reg, columns 16 nt + lo
                // (a lane whose column lies beyond ks writes to a slot of its own behind the staged rows instead of branching)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    double* srow = sO + (16 * tt + hi + 4 * reg) * ks;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int i = 16 * nt + lo;
                        double* dst = i < ks ? srow + i : sO + ST::DOUBLES + lane;
                        *dst = acc[nt][reg];  // columns k..ks-1 are exact zeros (zero B fragments)
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the staged rows in address order, 16 bytes per lane (ks is even: whole pairs)
            double* const gout = chunk_out + (size_t)g0 * ks;
            if ((ks & 1) == 0) {
                const int pairs = grows * ks / 2;
                for (int e = lane; e < pairs; e += 64) {
                    const dbl2 v = *reinterpret_cast<const dbl2*>(sO + 2 * e);
                    __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(gout + 2 * e));
                }
            } else {  // (odd row stride: no 16-byte alignment)
                for (int e = lane; e < grows * ks; e += 64) gout[e] = sO[e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}
