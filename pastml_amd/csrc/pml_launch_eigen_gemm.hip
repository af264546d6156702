// Launchers of the sum sweeps of the eigen models as two small FP64 matrix-core GEMMs per 16 nodes (pml_kernels_eigen_gemm.h).
#include "pml_launch.h"
#include "pml_kernels_eigen_gemm.h"

int launch_eigen_gemm(pml_ctx* ctx, int mode, const int* nodes, int first, int n) {
    if (n <= 0) return PML_OK;
    const int KS = (ctx->k + 3) / 4;
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const PmlModel m = model_of(ctx);
    int blocks = (n + PML_WAVES_PER_BLOCK * 16 - 1) / (PML_WAVES_PER_BLOCK * 16);
    const int cap = std::max(8, 16384 / std::max(1, ctx->C));
    if (blocks > cap) blocks = cap;
#define PML_EIGG_CASE(KS_, MODE_)                                                                                   \
    if (KS == KS_ && mode == MODE_) {                                                                               \
        hipLaunchKernelGGL((eigen_gemm_kernel<KS_, MODE_>), dim3(blocks, ctx->C), dim3(PML_BLOCK),                  \
                           EigGemm<KS_>::LDS_DOUBLES * sizeof(double), ctx->stream, t, c, m, st, nodes, first, n);  \
        HIP_TRY(hipGetLastError());                                                                                 \
        return PML_OK;                                                                                              \
    }
#define PML_EIGG_MODES(KS_) PML_EIGG_CASE(KS_, PML_EIGG_BU) PML_EIGG_CASE(KS_, PML_EIGG_TIPS) PML_EIGG_CASE(KS_, PML_EIGG_TD)
    PML_EIGG_MODES(1)
    PML_EIGG_MODES(2)
    PML_EIGG_MODES(3)
    PML_EIGG_MODES(4)
    PML_EIGG_MODES(5)
    PML_EIGG_MODES(6)
    PML_EIGG_MODES(7)
    PML_EIGG_MODES(8)
    PML_EIGG_MODES(9)
    PML_EIGG_MODES(10)
    PML_EIGG_MODES(11)
    PML_EIGG_MODES(12)
    PML_EIGG_MODES(13)
    PML_EIGG_MODES(14)
    PML_EIGG_MODES(15)
    PML_EIGG_MODES(16)
#undef PML_EIGG_MODES
#undef PML_EIGG_CASE
    return fail(PML_ERR_UNSUPPORTED, "no eigen kernel for k = %d", ctx->k);
}

int launch_eigen_gemm_narrow(pml_ctx* ctx, int mode, const int* nodes, const int* d_offsets, int first_level,
                                    int n_levels, const int* d_blk_start, int n_blocks) {
    if (n_levels <= 0) return PML_OK;
    const int KS = (ctx->k + 3) / 4;
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const PmlModel m = model_of(ctx);
#define PML_EIGG_CASE(KS_, MODE_)                                                                                      \
    if (KS == KS_ && mode == MODE_) {                                                                                  \
        hipLaunchKernelGGL((eigen_gemm_narrow_kernel<KS_, MODE_>), dim3(n_blocks, ctx->C), dim3(PML_BLOCK),            \
                           EigGemm<KS_>::LDS_DOUBLES * sizeof(double), ctx->stream, t, c, m, st, nodes,                \
                           d_offsets + first_level, n_levels, d_blk_start);                                            \
        HIP_TRY(hipGetLastError());                                                                                    \
        return PML_OK;                                                                                                 \
    }
#define PML_EIGG_MODES(KS_) PML_EIGG_CASE(KS_, PML_EIGG_BU) PML_EIGG_CASE(KS_, PML_EIGG_TD)
    PML_EIGG_MODES(1)
    PML_EIGG_MODES(2)
    PML_EIGG_MODES(3)
    PML_EIGG_MODES(4)
    PML_EIGG_MODES(5)
    PML_EIGG_MODES(6)
    PML_EIGG_MODES(7)
    PML_EIGG_MODES(8)
    PML_EIGG_MODES(9)
    PML_EIGG_MODES(10)
    PML_EIGG_MODES(11)
    PML_EIGG_MODES(12)
    PML_EIGG_MODES(13)
    PML_EIGG_MODES(14)
    PML_EIGG_MODES(15)
    PML_EIGG_MODES(16)
#undef PML_EIGG_MODES
#undef PML_EIGG_CASE
    return fail(PML_ERR_UNSUPPORTED, "no eigen kernel for k = %d", ctx->k);
}

