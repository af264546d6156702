// Model parameters on the device and the closed-form HKY85 matrix: shared by the P(t) batch (pml_kernels_pij.h) and the
// sweeps that build P(t) in registers (pml_kernels_matrix.h).
#pragma once
#include "pml_kernels_misc.h"

struct PmlModel {
    int kind;
    const double* mu;     // [C]        F81
    const double* kappa;  // [C]        HKY
    const double* d;      // [C][k]     eigen
    const double* A;      // [C][k][k]
    const double* Ainv;   // [C][k][k]
    const double* AinvT;  // [C][ldT][ldT]: Ainv transposed, zero-padded (k <= 64; null otherwise): AinvT[j][m] = Ainv[m][j]
    int ldT;              //   32 for k <= 32, 64 for k <= 64
    const double* Asym;   // [C][k][k], 65 <= k <= 128: Pi^-1/2 U with U = Pi^1/2 A orthonormalised (eig_sym_kernel); else null
    const double* sf;     // [C]
    const double* tau;    // [C]
    const double* tauf;   // [C]
};

// HKY85 closed form (pastml/models/HKYModel.py:55-82); states A, C, G, T = 0..3.  p[i][j].
// No FMA contraction here: entries such as pc sct - pc ect cancel exactly at t = 0 only when both products are rounded
// as numpy rounds them, and whether P(0) holds an exact zero or 1e-17 decides if a zero-length branch between
// conflicting states is a zero likelihood (ml.py:139-145) -- the reference's answer must be ours.
__device__ __forceinline__ void hky_matrix(const double* __restrict__ pi, double kappa, double tt, double (&p)[4][4]) {
#pragma clang fp contract(off)
    const double pa = pi[0], pc = pi[1], pg = pi[2], pt = pi[3];
    const double pag = pa + pg, pct = pc + pt;
    const double beta = .5 / (pag * pct + kappa * (pa * pg + pc * pt));
    const double eb = exp(-beta * tt);
    const double ect = exp(-beta * tt * (1. + pct * (kappa - 1.))) / pct;
    const double eag = exp(-beta * tt * (1. + pag * (kappa - 1.))) / pag;
    const double sct = (pct + pag * eb) / pct;
    const double sag = (pag + pct * eb) / pag;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) p[i][j] = (1. - eb) * pi[j];
    p[3][3] = pt * sct + pc * ect;
    p[3][1] = pc * sct - pc * ect;
    p[1][3] = pt * sct - pt * ect;
    p[1][1] = pc * sct + pt * ect;
    p[0][0] = pa * sag + pg * eag;
    p[0][2] = pg * sag - pg * eag;
    p[2][0] = pa * sag - pa * eag;
    p[2][2] = pg * sag + pa * eag;
}

