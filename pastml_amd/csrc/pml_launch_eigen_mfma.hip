// Launchers of the fused FP64 matrix-core kernels of the eigen models (pml_kernels_eigen_mfma.h) and of the P(t) batch on
// the matrix cores.
#include "pml_launch.h"
#include "pml_kernels_eigen_mfma.h"
#include "pml_kernels_pij_wide.h"

// fused eigen sweeps: one launch over a list (nodes) or a contiguous id range (first) of n nodes
int launch_eigen_fused(pml_ctx* ctx, int mode, const int* nodes, int first, int n, int tips) {
    if (n <= 0) return PML_OK;
    const int k = ctx->k;
    const int KS = (k + 3) / 4, NT = (k + 15) / 16;
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const PmlModel m = model_of(ctx);
#define PML_EIG_CASE(NT_, KS_, MODE_)                                                                              \
    if (NT == NT_ && KS == KS_ && mode == MODE_) {                                                                 \
        typedef EigShape<KS_> S;                                                                                   \
        const size_t lds = ((size_t)S::KP * k + (size_t)PML_WAVES_PER_BLOCK * S::WAVE_LDS) * sizeof(double);       \
        int blocks = (n + PML_WAVES_PER_BLOCK * S::NB - 1) / (PML_WAVES_PER_BLOCK * S::NB);                        \
        const int cap_all = (int)ctx->tune.get(T_EIG_BLOCKS, 8192);                                                  \
        const int cap = std::max(8, cap_all / std::max(1, ctx->C));                                                \
        if (blocks > cap) blocks = cap;                                                                            \
        hipLaunchKernelGGL((eigen_fused_kernel<NT_, KS_, MODE_>), dim3(blocks, ctx->C), dim3(PML_BLOCK), lds,      \
                           ctx->stream, t, c, m, st, nodes, first, n, tips);                                       \
        HIP_TRY(hipGetLastError());                                                                                \
        return PML_OK;                                                                                             \
    }
#define PML_EIG_MODES(NT_, KS_)               \
    PML_EIG_CASE(NT_, KS_, PML_EIG_BU_MARG)   \
    PML_EIG_CASE(NT_, KS_, PML_EIG_BU_JOINT)  \
    PML_EIG_CASE(NT_, KS_, PML_EIG_TD)
    PML_EIG_MODES(1, 4)
    PML_EIG_MODES(2, 5)
    PML_EIG_MODES(2, 6)
    PML_EIG_MODES(2, 7)
    PML_EIG_MODES(2, 8)
#undef PML_EIG_MODES
#undef PML_EIG_CASE
    return fail(PML_ERR_UNSUPPORTED, "no fused eigen kernel for k = %d", k);
}


// fused eigen sweeps: levels [first_level, first_level + n_levels) of a level table in one launch
int launch_eigen_narrow(pml_ctx* ctx, int mode, const int* nodes, const int* d_offsets, int first_level,
                               int n_levels) {
    if (n_levels <= 0) return PML_OK;
    const int k = ctx->k;
    const int KS = (k + 3) / 4, NT = (k + 15) / 16;
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const PmlModel m = model_of(ctx);
#define PML_EIG_CASE(NT_, KS_, MODE_)                                                                              \
    if (NT == NT_ && KS == KS_ && mode == MODE_) {                                                                 \
        typedef EigShape<KS_> S;                                                                                   \
        const size_t lds = ((size_t)S::KP * k + (size_t)PML_WAVES_PER_BLOCK * S::WAVE_LDS) * sizeof(double);       \
        hipLaunchKernelGGL((eigen_narrow_kernel<NT_, KS_, MODE_>), dim3(1, ctx->C), dim3(PML_BLOCK), lds,          \
                           ctx->stream, t, c, m, st, nodes, d_offsets + first_level, n_levels);                    \
        HIP_TRY(hipGetLastError());                                                                                \
        return PML_OK;                                                                                             \
    }
#define PML_EIG_MODES(NT_, KS_)               \
    PML_EIG_CASE(NT_, KS_, PML_EIG_BU_MARG)   \
    PML_EIG_CASE(NT_, KS_, PML_EIG_BU_JOINT)  \
    PML_EIG_CASE(NT_, KS_, PML_EIG_TD)
    PML_EIG_MODES(1, 4)
    PML_EIG_MODES(2, 5)
    PML_EIG_MODES(2, 6)
    PML_EIG_MODES(2, 7)
    PML_EIG_MODES(2, 8)
#undef PML_EIG_MODES
#undef PML_EIG_CASE
    return fail(PML_ERR_UNSUPPORTED, "no fused eigen kernel for k = %d", k);
}


// bottom-up messages of all tips (observed tips 16 to a tile, see eigen_tips_kernel)
int launch_eigen_tips(pml_ctx* ctx, int joint) {
    if (ctx->n_tips <= 0) return PML_OK;
    const int k = ctx->k;
    const int KS = (k + 3) / 4, NT = (k + 15) / 16;
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const PmlModel m = model_of(ctx);
#define PML_EIG_TIPS(NT_, KS_, J_)                                                                                   \
    if (NT == NT_ && KS == KS_ && joint == J_) {                                                                     \
        typedef EigShape<KS_> S;                                                                                     \
        const size_t lds = ((size_t)S::KP * k + (size_t)PML_WAVES_PER_BLOCK * S::WAVE_LDS_TIPS) * sizeof(double);    \
        int blocks = (ctx->n_tips + PML_WAVES_PER_BLOCK * 16 - 1) / (PML_WAVES_PER_BLOCK * 16);                      \
        const int cap = std::max(8, 8192 / std::max(1, ctx->C));                                                     \
        if (blocks > cap) blocks = cap;                                                                              \
        hipLaunchKernelGGL((eigen_tips_kernel<NT_, KS_, J_>), dim3(blocks, ctx->C), dim3(PML_BLOCK), lds,            \
                           ctx->stream, t, c, m, st, ctx->d_tips, ctx->n_tips);                                      \
        HIP_TRY(hipGetLastError());                                                                                  \
        return PML_OK;                                                                                               \
    }
#define PML_EIG_TIPS2(NT_, KS_) PML_EIG_TIPS(NT_, KS_, 0) PML_EIG_TIPS(NT_, KS_, 1)
    PML_EIG_TIPS2(1, 4)
    PML_EIG_TIPS2(2, 5)
    PML_EIG_TIPS2(2, 6)
    PML_EIG_TIPS2(2, 7)
    PML_EIG_TIPS2(2, 8)
#undef PML_EIG_TIPS2
#undef PML_EIG_TIPS
    return fail(PML_ERR_UNSUPPORTED, "no fused eigen kernel for k = %d", k);
}



// P(t) of every branch on the FP64 matrix cores beyond 32 states: A^T in LDS slices (pml_kernels_pij_wide.h)
int launch_pij_wide(pml_ctx* ctx) {
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlModel m = model_of(ctx);
    const int k = ctx->k;
    const int ntc = pijw_tiles(k, ctx->ks);
    const size_t lds = pijw_lds_bytes(k, ntc);
    // a workgroup per CU (LDS); several rounds of workgroups so that the tail is short, every wave with a branch of its own
    int blocks = std::max(1, (int)ctx->tune.get(T_PIJ_BLOCKS, 1024) / std::max(1, ctx->C));
    blocks = std::min(blocks, (ctx->N + PML_PIJW_WAVES - 1) / PML_PIJW_WAVES);
    const int bpb = (ctx->N + blocks - 1) / blocks;
    dim3 grid((ctx->N + bpb - 1) / bpb, ctx->C);
#define PML_PIJW_CASE(NTC_)                                                                                          \
    if (ntc == NTC_) {                                                                                               \
        PML_TRY(with_lds(ctx, pij_eigen_wide_kernel<NTC_>, lds));                                                    \
        hipLaunchKernelGGL((pij_eigen_wide_kernel<NTC_>), grid, dim3(PML_PIJW_BLOCK), lds, ctx->stream, t, c, m, ctx->d_P, bpb); \
    }
    PML_PIJW_CASE(1) PML_PIJW_CASE(2) PML_PIJW_CASE(3) PML_PIJW_CASE(4)
    PML_PIJW_CASE(5) PML_PIJW_CASE(6) PML_PIJW_CASE(7) PML_PIJW_CASE(8)
#undef PML_PIJW_CASE
    HIP_TRY(hipGetLastError());
    return PML_OK;
}


// P(t) of every branch on the FP64 matrix cores, 16 <= k <= 32 (run_prep)
int launch_pij_mfma(pml_ctx* ctx) {
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlModel m = model_of(ctx);
    // FP64 matrix-core path (BASELINE config 3: JTT, k = 20)
    const int k = ctx->k;
    const int KS = (k + 3) / 4, NT = (k + 15) / 16;
    // rows of the result a wave stages in LDS per flush (pml_kernels_pij.h)
    int srows = (int)ctx->tune.get(T_PIJ_STAGE_ROWS, 32);
    if (srows != 16 && srows != 64) srows = 32;
    auto lds_of = [&](int sr) {
        return ((size_t)KS * 4 * k + (size_t)PML_WAVES_PER_BLOCK * ((size_t)(PML_MFMA_CHUNK + sr) * KS * 4 + 64)) * sizeof(double);
    };
    if (srows == 64 && lds_of(64) > 64 * 1024) srows = 32;  // (the default limit of dynamic LDS)
    const size_t lds = lds_of(srows);
    int blocks = (ctx->N + PML_WAVES_PER_BLOCK * PML_MFMA_CHUNK - 1) / (PML_WAVES_PER_BLOCK * PML_MFMA_CHUNK);
    // every wave walks several chunks: the block's set-up (Ainv to LDS, the fragments of A) is paid once
    const int cap = std::max(64, (int)ctx->tune.get(T_PIJ_BLOCKS, 2048) / std::max(1, ctx->C));
    if (blocks > cap) blocks = cap;
    dim3 grid(blocks, ctx->C);
#define PML_MFMA_CASE_R(NT_, KS_, SR_)                                                                             \
    if (NT == NT_ && KS == KS_ && srows == SR_) {                                                                  \
        hipLaunchKernelGGL((pij_eigen_mfma_kernel<NT_, KS_, SR_>), grid, dim3(PML_BLOCK), lds, ctx->stream, t, c, m, ctx->d_P);    \
    }
#define PML_MFMA_CASE(NT_, KS_) PML_MFMA_CASE_R(NT_, KS_, 16) PML_MFMA_CASE_R(NT_, KS_, 32) PML_MFMA_CASE_R(NT_, KS_, 64)
    PML_MFMA_CASE(1, 4)
    PML_MFMA_CASE(2, 5)
    PML_MFMA_CASE(2, 6)
    PML_MFMA_CASE(2, 7)
    PML_MFMA_CASE(2, 8)
#undef PML_MFMA_CASE_R
#undef PML_MFMA_CASE
    HIP_TRY(hipGetLastError());
    return PML_OK;
}
