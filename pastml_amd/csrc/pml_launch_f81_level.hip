// Launchers of the F81-family level kernels (one launch per fused height / depth).
#include "pml_launch_f81_level.h"

int dispatch_sweep_f81(pml_ctx* ctx, SweepKind what, const int* level, int n_level) {
    if (n_level <= 0) return PML_OK;
    if (what == SW_TD) return fail(PML_ERR_INVALID, "the F81 kernels walk descriptor lists: SW_TD has none");
    const bool td = what == SW_TD_FUSED || what == SW_ROOTS;
    int g = td ? ctx->Gt : ctx->Gf, r = td ? ctx->Rt : ctx->Rf;
    // Fused bottom-up levels, 32 < k <= 64 (measured on cfg4): 8 states per lane (8 units per wavefront share the
    // per-unit scalar work) wins on the level that rebuilds cherries (1.80 -> 1.58 ms) and on small levels;
    // 4 states per lane (twice the loads in flight per unit) wins on big levels that stream stored vectors.
    // (SW_BU_CHERRIES rewrites pi . v of the cherries, which the level that rebuilds them has stored: same shape, or
    // a download of the bottom-up vectors would change the last bit of what a later top-down sweep reads)
    if (ctx->bu_wide_lanes && (what == SW_BU_MARG_FUSED_NOVEC || what == SW_BU_CHERRIES ||
                               (what == SW_BU_MARG_FUSED && n_level <= 65536))) {
        g = 8;
        r = 8;
    }
#define X(G_, R_)                                                \
if (g == G_ && r == R_) {                                    \
    launch_sweep_f81<G_, R_>(ctx, what, level, n_level);     \
    HIP_TRY(hipGetLastError());                              \
    return PML_OK;                                           \
}
    PML_F81_CASES(X)
#undef X
    if (g == 64 && r == 8) return dispatch_sweep_f81_wide(ctx, what, level, n_level);   // (more than 256 states)
    return fail(PML_ERR_UNSUPPORTED, "no F81 kernel for G=%d R=%d", g, r);
}
