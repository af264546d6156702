// Launchers of the sum sweeps of the eigen models as two small FP64 matrix-core GEMMs per 16 nodes (pml_kernels_eigen_gemm.h).
#include "pml_launch.h"
#include "pml_kernels_eigen_gemm.h"

// states per lane (k-steps of the products): k / 4 rounded up -- and then up to an even number, so that a lane owns pairs of
// neighbouring states and moves its part of a vector in 16-byte accesses (EigGemm<KS>::PAIRS).  The extra k-step of an odd
// k / 4 costs a fifth more matrix instructions at k = 20 and buys half the memory instructions: 65 536 tips x 32 columns,
// bottom-up sweep 0.975 -> see profiles/r06g_eigen_sweeps.txt.
// The top-down mode keeps k / 4 rounded up: it is bound by its registers and dependent loads, not by memory instructions, and
// the extra k-step only costs (k = 20: 0.88 -> 1.11 ms for 65 536 tips x 32 columns; k = 36: 0.86 -> 0.93).
static int eig_gemm_ks(int k, int mode) {
    const int ks4 = (k + 3) / 4;
    if (mode == PML_EIGG_TD) return ks4;
    return ks4 <= 1 ? 1 : (ks4 + 1) / 2 * 2;
}

int launch_eigen_gemm(pml_ctx* ctx, int mode, const int* nodes, int first, int n) {
    if (n <= 0) return PML_OK;
    if (ctx->k > 64) return launch_eigen_gemm_wide(ctx, mode, nodes, first, n);   // pml_launch_eigen_gemm_wide.hip
    const int KS = eig_gemm_ks(ctx->k, mode);
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const PmlModel m = model_of(ctx);
    int blocks = (n + PML_WAVES_PER_BLOCK * 16 - 1) / (PML_WAVES_PER_BLOCK * 16);
    // Blocks over all columns: every wave loads the constant operands once, so it should walk several passes -- above all
    // where the operands go through LDS (k > 32: 64 KB per block).  Measured on a 65 536-tip tree (profiles/r06g_eigen_sweeps.txt):
    // k = 61 x 32 columns, marginal pass: 16 384 blocks 9.50 ms, 4 096 7.88, 1 024 7.15, 512 6.89; k = 20 x 32: 2.09, 1.96, 2.13,
    // 2.05; k = 8 x 32: 0.87, 0.87, 0.92, 1.08.  PASTML_HIP_EIG_BLOCKS overrides.
    const int cap_all = (int)ctx->tune.get(T_EIG_BLOCKS, KS > 8 ? 1024 : 4096);
    const int cap = std::max(8, cap_all / std::max(1, ctx->C));
    if (blocks > cap) blocks = cap;
#define PML_EIGG_CASE(KS_, MODE_)                                                                                   \
    if (KS == KS_ && mode == MODE_) {                                                                               \
        hipLaunchKernelGGL((eigen_gemm_kernel<KS_, MODE_>), dim3(blocks, ctx->C), dim3(PML_BLOCK),                  \
                           EigGemm<KS_>::LDS_DOUBLES * sizeof(double), ctx->stream, t, c, m, st, nodes, first, n);  \
        HIP_TRY(hipGetLastError());                                                                                 \
        return PML_OK;                                                                                              \
    }
#define PML_EIGG_MODES(KS_) PML_EIGG_CASE(KS_, PML_EIGG_BU) PML_EIGG_CASE(KS_, PML_EIGG_TIPS) PML_EIGG_CASE(KS_, PML_EIGG_TD)
#define PML_EIGG_ODD(KS_) PML_EIGG_CASE(KS_, PML_EIGG_TD)
    PML_EIGG_MODES(1)
    PML_EIGG_MODES(2)
    PML_EIGG_ODD(3)
    PML_EIGG_MODES(4)
    PML_EIGG_ODD(5)
    PML_EIGG_MODES(6)
    PML_EIGG_ODD(7)
    PML_EIGG_MODES(8)
    PML_EIGG_ODD(9)
    PML_EIGG_MODES(10)
    PML_EIGG_ODD(11)
    PML_EIGG_MODES(12)
    PML_EIGG_ODD(13)
    PML_EIGG_MODES(14)
    PML_EIGG_ODD(15)
    PML_EIGG_MODES(16)
#undef PML_EIGG_ODD
#undef PML_EIGG_MODES
#undef PML_EIGG_CASE
    return fail(PML_ERR_UNSUPPORTED, "no eigen kernel for k = %d", ctx->k);
}

int launch_eigen_gemm_narrow(pml_ctx* ctx, int mode, const int* nodes, const int* d_offsets, int first_level,
                                    int n_levels, const int* d_blk_start, int n_blocks) {
    if (n_levels <= 0) return PML_OK;
    if (ctx->k > 64) return launch_eigen_gemm_narrow_wide(ctx, mode, nodes, d_offsets, first_level, n_levels, d_blk_start, n_blocks);
    const int KS = eig_gemm_ks(ctx->k, mode);
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
#define PML_EIGG_ODD(KS_) PML_EIGG_CASE(KS_, PML_EIGG_TD)
    PML_EIGG_MODES(1)
    PML_EIGG_MODES(2)
    PML_EIGG_ODD(3)
    PML_EIGG_MODES(4)
    PML_EIGG_ODD(5)
    PML_EIGG_MODES(6)
    PML_EIGG_ODD(7)
    PML_EIGG_MODES(8)
    PML_EIGG_ODD(9)
    PML_EIGG_MODES(10)
    PML_EIGG_ODD(11)
    PML_EIGG_MODES(12)
    PML_EIGG_ODD(13)
    PML_EIGG_MODES(14)
    PML_EIGG_ODD(15)
    PML_EIGG_MODES(16)
#undef PML_EIGG_ODD
#undef PML_EIGG_MODES
#undef PML_EIGG_CASE
    return fail(PML_ERR_UNSUPPORTED, "no eigen kernel for k = %d", ctx->k);
}

