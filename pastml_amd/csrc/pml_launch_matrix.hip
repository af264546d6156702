// Launchers of the sweeps over materialised (or closed-form 4 x 4) transition matrices and of the state selection.
#include "pml_launch.h"
#include "pml_kernels_pij.h"

// matrix-model sweeps: contiguous state ownership (state = g * R + r)
template <int G, int R>
static void launch_sweep(pml_ctx* ctx, SweepKind what, const int* level, int n_level) {
    const PmlTree t = tree_of(ctx, false);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const int upb = PML_WAVES_PER_BLOCK * (64 / G);
    dim3 grid(grid_for(ctx, n_level, upb, ctx->C), ctx->C), block(PML_BLOCK);
    const PmlModel m = model_of(ctx);
    if (G == 4 && R == 1 && hky_fused(ctx)) {  // HKY: P(t) from the closed form, in registers (no batch in HBM)
        constexpr int GG = G == 4 ? 4 : 4, RR = R == 1 ? 1 : 1;  // (keeps the other shapes from instantiating it)
        switch (what) {
            case SW_BU_MARG:
                hipLaunchKernelGGL((bu_matrix_kernel<GG, RR, false, PML_P_HKY>), grid, block, 0, ctx->stream, t, c, st,
                                   nullptr, m, level, n_level);
                return;
            case SW_BU_JOINT:
                hipLaunchKernelGGL((bu_matrix_kernel<GG, RR, true, PML_P_HKY>), grid, block, 0, ctx->stream, t, c, st,
                                   nullptr, m, level, n_level);
                return;
            case SW_TD:
                hipLaunchKernelGGL((td_matrix_kernel<GG, RR, PML_P_HKY>), grid, block, 0, ctx->stream, t, c, st, nullptr,
                                   m, level, n_level);
                return;
            default:
                break;
        }
    }
    switch (what) {
        case SW_BU_MARG:
            hipLaunchKernelGGL((bu_matrix_kernel<G, R, false>), grid, block, 0, ctx->stream, t, c, st, ctx->d_P, m, level,
                               n_level);
            break;
        case SW_BU_JOINT:
            hipLaunchKernelGGL((bu_matrix_kernel<G, R, true>), grid, block, 0, ctx->stream, t, c, st, ctx->d_P, m, level,
                               n_level);
            break;
        case SW_TD:
            hipLaunchKernelGGL((td_matrix_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st, ctx->d_P, m, level,
                               n_level);
            break;
        case SW_ROOTS:
            hipLaunchKernelGGL((td_roots_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st);
            break;
        default:
            break;
    }
}


template <int G, int R>
static void launch_select(pml_ctx* ctx, int method, int force_joint, const u64* d_lh_mask) {
    const int upb = PML_WAVES_PER_BLOCK * (64 / G);
    dim3 grid(grid_for(ctx, ctx->N, upb, ctx->C), ctx->C), block(PML_BLOCK);
    hipLaunchKernelGGL((select_states_kernel<G, R>), grid, block, 0, ctx->stream, ctx->N, ctx->k, ctx->ks, ctx->W,
                       ctx->d_post, d_lh_mask, ctx->d_js, method, force_joint, ctx->d_masks, ctx->d_nsel);
}


int dispatch_sweep_matrix(pml_ctx* ctx, SweepKind what, const int* level, int n_level) {
    if (n_level <= 0) return PML_OK;
#define X(G_, R_)                                             \
    if (ctx->G == G_ && ctx->R == R_) {                       \
        launch_sweep<G_, R_>(ctx, what, level, n_level);      \
        HIP_TRY(hipGetLastError());                           \
        return PML_OK;                                        \
    }
    PML_GR_CASES(X)
#undef X
    return fail(PML_ERR_UNSUPPORTED, "no kernel for G=%d R=%d", ctx->G, ctx->R);
}

// lane shape of the selection kernel: 8 states per lane up to k = 64 (8 units per wavefront share the scalar work
// and the arg-max butterflies stay inside a 16-lane row: 3.2 -> 2.6 ms per pass of 4 columns at cfg4 size),
// else the matrix shapes
int dispatch_select(pml_ctx* ctx, int method, int force_joint, const u64* d_lh_mask) {
    int sg = ctx->G, sr = ctx->R;
    if (ctx->k <= 64) {
        sr = 8;
        sg = 1;
        while (sg * sr < ctx->k) sg <<= 1;
    }
#define X(G_, R_)                                                      \
    if (sg == G_ && sr == R_) {                                        \
        launch_select<G_, R_>(ctx, method, force_joint, d_lh_mask);    \
        return PML_OK;                                                 \
    }
    PML_GR_CASES(X)
    X(1, 8)
    X(2, 8)
    X(4, 8)
    X(8, 8)
    X(64, 8)
#undef X
    return fail(PML_ERR_UNSUPPORTED, "no selection kernel for G=%d R=%d", ctx->G, ctx->R);
}
