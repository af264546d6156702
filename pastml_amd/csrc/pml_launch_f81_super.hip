// Launchers of the two-level and stacked units of the F81-family sweeps (pml_kernels_f81.h).
#include "pml_launch.h"

template <int G, int R>
static void launch_super_f81(pml_ctx* ctx, bool bottom_up) {
    const PmlTree t = tree_of(ctx, true);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const int upb = PML_WAVES_PER_BLOCK * (64 / G);
    // (top-down: one unit per child of a two-level node)
    dim3 grid(grid_for(ctx, bottom_up ? ctx->sup.n : 2 * ctx->sup.n, upb, ctx->C, bottom_up), ctx->C), block(PML_BLOCK);
    if (bottom_up)
        hipLaunchKernelGGL((bu_f81_super_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st, ctx->sup.d_units, ctx->sup.n);
    else
        hipLaunchKernelGGL((td_f81_super_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st, ctx->sup.d_units, ctx->sup.n);
}

// stacked units of bottom-up level / depth `level` (pml_kernels_f81.h)
template <int G, int R>
static void launch_stack_f81(pml_ctx* ctx, bool bottom_up, int a, int n) {
    const PmlTree t = tree_of(ctx, true);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const int upb = PML_WAVES_PER_BLOCK * (64 / G);
    dim3 grid(grid_for(ctx, bottom_up ? n : 2 * n, upb, ctx->C), ctx->C), block(PML_BLOCK);
    if (bottom_up)
        hipLaunchKernelGGL((bu_f81_stack_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st, ctx->sup.d_stack_bu + a, n);
    else
        hipLaunchKernelGGL((td_f81_stack_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st, ctx->sup.d_stack_td + a, n);
}

int dispatch_super_f81(pml_ctx* ctx, bool bottom_up) {
    if (ctx->sup.n <= 0) return PML_OK;  // (a schedule of stacked units only)
    int g, r;
    super_shape(ctx, bottom_up, g, r);
#define X(G_, R_)                                   \
    if (g == G_ && r == R_) {                       \
        launch_super_f81<G_, R_>(ctx, bottom_up);   \
        HIP_TRY(hipGetLastError());                 \
        return PML_OK;                              \
    }
    PML_SUPER_CASES(X)
#undef X
    return fail(PML_ERR_UNSUPPORTED, "no two-level F81 kernel for G=%d R=%d", g, r);
}

int dispatch_stack_f81(pml_ctx* ctx, bool bottom_up, int level) {
    const std::vector<int>& off = bottom_up ? ctx->sup.stack_bu_offsets : ctx->sup.stack_td_offsets;
    if (ctx->sup.n_stack == 0 || level + 1 >= (int)off.size()) return PML_OK;
    const int a = off[level], n = off[level + 1] - a;
    if (n <= 0) return PML_OK;
    int g, r;
    super_shape(ctx, bottom_up, g, r);
#define X(G_, R_)                                         \
    if (g == G_ && r == R_) {                             \
        launch_stack_f81<G_, R_>(ctx, bottom_up, a, n);   \
        HIP_TRY(hipGetLastError());                       \
        return PML_OK;                                    \
    }
    PML_SUPER_CASES(X)
#undef X
    return fail(PML_ERR_UNSUPPORTED, "no stacked F81 kernel for G=%d R=%d", g, r);
}

