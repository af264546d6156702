// Launchers of the F81-family subtree-block kernels: mid-size forests, the thin ends of large ragged forests.
#include "pml_launch.h"

// one workgroup per (subtree block, column) walks the block's levels (pml_kernels_f81.h, bottom)
// (the tables of one launch: `blocks`, or the thin ends of a large forest -- `thin` bottom-up, `deep` top-down)
struct BlockTables {
    const PmlUnit* units;
    const int *start, *levels, *lv;
    int n_blocks;
};
static BlockTables block_tables(const pml_ctx* ctx, bool bottom_up, int which) {
    if (which >= 1) {   // (bottom-up: tier which - 1 of the thin levels)
        if (bottom_up) {
            const pml_ctx::ThinSchedule::Tier& T = ctx->thin.tiers[which - 1];
            return {ctx->thin.d_units, ctx->thin.d_start + T.first_block, ctx->thin.d_levels + T.first_block, ctx->thin.d_lv, T.n_blocks};
        }
        return {ctx->deep.d_units, ctx->deep.d_start, ctx->deep.d_levels, ctx->deep.d_lv, ctx->deep.n_blocks};
    }
    const pml_ctx::BlockSchedule& B = ctx->blocks;
    if (bottom_up) return {B.d_bu_units, B.d_bu_start, B.d_bu_levels, B.d_bu_lv, B.n_blocks};
    return {B.d_td_units, B.d_td_start, B.d_td_levels, B.d_td_lv, B.n_blocks};
}

template <int G, int R>
static void launch_blocks_f81(pml_ctx* ctx, bool bottom_up, int which) {
    const PmlTree t = tree_of(ctx, true);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const BlockTables B = block_tables(ctx, bottom_up, which);
    // Workgroup size: 512 threads while every (block, column) workgroup is resident at once; with more workgroups than
    // the chip holds the launch runs in rounds of long-lived workgroups (HIV1C x 14 columns: 980 workgroups of 8 waves,
    // one per CU at 3 waves per SIMD -> four rounds, 97 us for blocks of <= 24 level steps), so the workgroups shrink
    // until they all fit: a thin level needs one or two waves, wider ones take more passes (walk_levels).
    static int waves_per_cu[2] = {0, 0}, n_cus = 0;
    if (n_cus == 0) {
        hipDeviceProp_t prop;
        n_cus = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    int& wpc = waves_per_cu[bottom_up ? 1 : 0];
    if (wpc == 0) {
        int nb = 0;
        const hipError_t e = bottom_up
            ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bu_f81_blocks_kernel<G, R>, 64, 0)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, td_f81_blocks_kernel<G, R>, 64, 0);
        wpc = (e == hipSuccess && nb > 0) ? nb : 8;
    }
    const int forced = (int)ctx->tune.get(T_BLOCK_THREADS, 0);
    int threads = PML_SMALL_BLOCK;
    const long long n_wg = (long long)B.n_blocks * (bottom_up ? ctx->sched_cols : ctx->C);  // (the workgroups that work)
    while (threads > 64 && n_wg * (threads / 64) > (long long)n_cus * wpc) threads /= 2;
    if (forced >= 64 && forced <= PML_SMALL_BLOCK) threads = forced;
    dim3 grid(B.n_blocks, ctx->C), block(threads);
    if (bottom_up)
        hipLaunchKernelGGL((bu_f81_blocks_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st, B.units, B.start, B.levels, B.lv);
    else {
        const bool signal = ctx->signal_next_td && ctx->C <= 64 && !ctx->tune.on(T_NO_SPIN_WAIT);
        ctx->signal_next_td = false;
        hipLaunchKernelGGL((td_f81_blocks_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st, B.units, B.start, B.levels,
                           B.lv, signal ? ctx->d_done : nullptr, signal ? ctx->h_done : nullptr);
        ctx->td_final_signals = signal;
        if (signal) ++ctx->signals_enqueued;
    }
}


int dispatch_blocks_f81(pml_ctx* ctx, bool bottom_up, int which) {
    int g, r;
    multi_level_shape(ctx, bottom_up, g, r);
#define X(G_, R_)                                           \
    if (g == G_ && r == R_) {                               \
        launch_blocks_f81<G_, R_>(ctx, bottom_up, which);   \
        HIP_TRY(hipGetLastError());                         \
        return PML_OK;                                      \
    }
    PML_F81_CASES(X)
#undef X
    return fail(PML_ERR_UNSUPPORTED, "no F81 kernel for G=%d R=%d", g, r);
}

