// Launchers of the sum sweeps of the eigen models with 65 - 128 states: the two-GEMM kernels of pml_kernels_eigen_gemm.h in their
// one-matrix form (EigGemm<KS>::SYM: A alone in LDS, read transposed by the first product), a translation unit of its own.
#include "pml_launch.h"
#include "pml_kernels_eigen_gemm.h"

// states per lane: k / 4 rounded up to a multiple of four (20, 24, 28, 32 -- four shapes instead of sixteen)
static int eig_gemm_wide_ks(int k) { return ((k + 15) / 16) * 4; }

int launch_eigen_gemm_wide(pml_ctx* ctx, int mode, const int* nodes, int first, int n) {
    if (n <= 0) return PML_OK;
    const int KS = eig_gemm_wide_ks(ctx->k);
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const PmlModel m = model_of(ctx);
    int blocks = (n + PML_WAVES_PER_BLOCK * 16 - 1) / (PML_WAVES_PER_BLOCK * 16);
    // one workgroup per CU (A is 50 - 132 KB of LDS and is loaded once per workgroup): a few rounds of them
    const int cap_all = (int)ctx->tune.get(T_EIG_BLOCKS, 1024);
    const int cap = std::max(8, cap_all / std::max(1, ctx->C));
    if (blocks > cap) blocks = cap;
#define PML_EIGG_CASE(KS_, MODE_)                                                                                   \
    if (KS == KS_ && mode == MODE_) {                                                                               \
        const size_t lds = EigGemm<KS_>::LDS_DOUBLES * sizeof(double);                                              \
        PML_TRY(with_lds(ctx, eigen_gemm_kernel<KS_, MODE_>, lds));                                                      \
        hipLaunchKernelGGL((eigen_gemm_kernel<KS_, MODE_>), dim3(blocks, ctx->C), dim3(PML_BLOCK), lds, ctx->stream, t, c, m, st, \
                           nodes, first, n);                                                                        \
        HIP_TRY(hipGetLastError());                                                                                 \
        return PML_OK;                                                                                              \
    }
#define PML_EIGG_MODES(KS_) PML_EIGG_CASE(KS_, PML_EIGG_BU) PML_EIGG_CASE(KS_, PML_EIGG_TIPS) PML_EIGG_CASE(KS_, PML_EIGG_TD)
    PML_EIGG_MODES(20)
    PML_EIGG_MODES(24)
    PML_EIGG_MODES(28)
    PML_EIGG_MODES(32)
#undef PML_EIGG_MODES
#undef PML_EIGG_CASE
    return fail(PML_ERR_UNSUPPORTED, "no eigen kernel for k = %d", ctx->k);
}

int launch_eigen_gemm_narrow_wide(pml_ctx* ctx, int mode, const int* nodes, const int* d_offsets, int first_level, int n_levels,
                                  const int* d_blk_start, int n_blocks) {
    if (n_levels <= 0) return PML_OK;
    const int KS = eig_gemm_wide_ks(ctx->k);
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const PmlModel m = model_of(ctx);
#define PML_EIGG_CASE(KS_, MODE_)                                                                                      \
    if (KS == KS_ && mode == MODE_) {                                                                                  \
        const size_t lds = EigGemm<KS_>::LDS_DOUBLES * sizeof(double);                                                 \
        PML_TRY(with_lds(ctx, eigen_gemm_narrow_kernel<KS_, MODE_>, lds));                                                  \
        hipLaunchKernelGGL((eigen_gemm_narrow_kernel<KS_, MODE_>), dim3(n_blocks, ctx->C), dim3(PML_BLOCK), lds, ctx->stream, t, c, \
                           m, st, nodes, d_offsets + first_level, n_levels, d_blk_start);                              \
        HIP_TRY(hipGetLastError());                                                                                    \
        return PML_OK;                                                                                                 \
    }
#define PML_EIGG_MODES(KS_) PML_EIGG_CASE(KS_, PML_EIGG_BU) PML_EIGG_CASE(KS_, PML_EIGG_TD)
    PML_EIGG_MODES(20)
    PML_EIGG_MODES(24)
    PML_EIGG_MODES(28)
    PML_EIGG_MODES(32)
#undef PML_EIGG_MODES
#undef PML_EIGG_CASE
    return fail(PML_ERR_UNSUPPORTED, "no eigen kernel for k = %d", ctx->k);
}
