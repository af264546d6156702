// Launchers of the F81-family kernels that walk a level table in one launch with a workgroup per column: whole sweeps of
// small forests, the narrow end next to the roots of large ones.
#include "pml_launch.h"

// Single-workgroup-per-column launch over a range of levels: the whole sweep of a small forest, or the narrow end of
// a large one (bottom-up: levels first_level .. end, then ln L; top-down: roots, then levels 0 .. n_levels - 1).
template <int G, int R>
static void launch_small_f81(pml_ctx* ctx, bool bottom_up, int do_prep, const PmlUnit* units, const int* d_offsets,
                             int n_levels, int reset_err, int skip_roots) {
    const PmlTree t = tree_of(ctx, true);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    dim3 grid(1, ctx->C), block(PML_SMALL_BLOCK);
    if (bottom_up) {
        // The completion word (bu_f81_small_kernel, wait_bottom_up) for sweeps of few columns, where the host's wait is
        // a tenth of the sweep (HIV1C tree, k = 12: 14 columns 0.1265 -> 0.1127 ms per sweep; at 128 columns the
        // system-scope fences in 128 workgroups cost what the spin saves: 0.203 against 0.207 ms)
        const bool signal = ctx->sched_cols <= 64 && !ctx->tune.on(T_NO_SPIN_WAIT);
        hipLaunchKernelGGL((bu_f81_small_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st, ctx->d_mu, ctx->d_sf,
                           ctx->d_tau, ctx->d_tauf, do_prep, units, d_offsets, n_levels, ctx->h_loglik, ctx->h_err,
                           reset_err, signal ? ctx->d_done : nullptr, signal ? ctx->h_done : nullptr);
        ctx->enqueue_signals = signal;  // (the last launch of a bottom-up sweep whenever it is part of one)
        if (signal) ++ctx->signals_enqueued;
    } else {
        const bool signal = ctx->signal_next_td && ctx->C <= 64 && !ctx->tune.on(T_NO_SPIN_WAIT);
        ctx->signal_next_td = false;
        hipLaunchKernelGGL((td_f81_small_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st, units, d_offsets,
                           n_levels, signal ? ctx->d_done : nullptr, signal ? ctx->h_done : nullptr, skip_roots);
        ctx->td_final_signals = signal;
        if (signal) ++ctx->signals_enqueued;
    }
}

// units / d_offsets: the level table to walk (default: the fused lists of the whole forest from first_level on)
int dispatch_small_f81(pml_ctx* ctx, bool bottom_up, int do_prep, int first_level, int n_levels,
                              const PmlUnit* units, const int* d_offsets, int skip_roots) {
    int g, r;
    multi_level_shape(ctx, bottom_up, g, r);
    if (n_levels < 0) n_levels = bottom_up ? (int)ctx->bu_offsets_f.size() - 1 - first_level : ctx->n_td_levels;
    const int reset_err = (units == nullptr && first_level == 0) ? 1 : 0;
    if (units == nullptr) {
        // (the lists sorted by shape inside every level where the forest has them: a wave of one shape runs that shape's
        // code -- walk_levels; a forest this small sits in the L2, where its rows lie does not matter)
        const bool sorted = ctx->d_bu_units_fs != nullptr && g < 8 && !ctx->tune.on(T_NO_SHAPE_SORT);
        units = bottom_up ? (sorted ? ctx->d_bu_units_fs : ctx->d_bu_units_f) : (sorted ? ctx->d_td_units_fs : ctx->d_td_units_f);
        d_offsets = bottom_up ? ctx->d_bu_offsets_f + first_level : ctx->d_td_parent_offsets_f + first_level;
    }
#define X(G_, R_)                                                                                   \
    if (g == G_ && r == R_) {                                                                       \
        launch_small_f81<G_, R_>(ctx, bottom_up, do_prep, units, d_offsets, n_levels, reset_err, skip_roots);   \
        HIP_TRY(hipGetLastError());                                                                 \
        return PML_OK;                                                                              \
    }
    PML_F81_CASES(X)
#undef X
    return fail(PML_ERR_UNSUPPORTED, "no F81 kernel for G=%d R=%d", g, r);
}

