// The launcher template of the F81-family level kernels, shared by the translation units that instantiate it:
// pml_launch_f81_level.hip (the lane shapes up to 256 states) and pml_launch_f81_wide.hip (64 lanes x 8 states, 257 - 512 states).
#pragma once
#include "pml_launch.h"

// F81-family sweeps: chunked state ownership (pml_kernels_f81.h), their own (G, R)
template <int G, int R>
static void launch_sweep_f81(pml_ctx* ctx, SweepKind what, const int* level, int n_level) {
    const bool fused_lists = what == SW_BU_MARG_FUSED || what == SW_BU_MARG_FUSED_NOVEC ||
                             what == SW_BU_JOINT_FUSED || what == SW_BU_JOINT_FUSED_NOVEC;
    const PmlTree t = tree_of(ctx, fused_lists || what == SW_TD_FUSED);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const int upb = PML_WAVES_PER_BLOCK * (64 / G);
    const bool pipelined = fused_lists || what == SW_BU_MARG || what == SW_BU_CHERRIES || what == SW_BU_JOINT ||
                           what == SW_BU_JOINT_NOVEC || what == SW_BU_CHERRIES_JOINT;
    dim3 grid(grid_for(ctx, n_level, upb, ctx->C, pipelined), ctx->C), block(PML_BLOCK);
    // the level is given as a position in one of the node lists; the kernels read the descriptor list parallel to it
    const PmlUnit* units = nullptr;
    // (level launches of wide units walk the lists sorted by shape, pml_tree_upload)
    const bool sorted = ctx->level_lists_sorted && ctx->d_bu_units_fs != nullptr;
    if (fused_lists) units = ctx->d_bu_units_f + (level - ctx->d_bu_order_f);
    if (sorted && (what == SW_BU_MARG_FUSED || what == SW_BU_MARG_FUSED_NOVEC))
        units = ctx->d_bu_units_fs + (level - ctx->d_bu_order_f);
    if (what == SW_BU_MARG || what == SW_BU_JOINT || what == SW_BU_JOINT_NOVEC)
        units = ctx->d_bu_units + (level - ctx->d_bu_order);
    if (what == SW_TD_FUSED) units = (sorted ? ctx->d_td_units_fs : ctx->d_td_units_f) + (level - ctx->d_td_parents_f);
    if (what == SW_BU_CHERRIES || what == SW_BU_CHERRIES_JOINT) units = ctx->d_cherry_units + (level - ctx->d_cherries);
    if (ctx->units_override != nullptr) units = ctx->units_override;  // a level of the block schedule's top part
    switch (what) {
        case SW_BU_MARG_FUSED:
        case SW_BU_MARG:
            hipLaunchKernelGGL((bu_f81_kernel<G, R, false, true>), grid, block, 0, ctx->stream, t, c, st, units,
                               n_level);
            break;
        case SW_BU_MARG_FUSED_NOVEC:
        case SW_BU_CHERRIES:
            hipLaunchKernelGGL((bu_f81_kernel<G, R, false, false>), grid, block, 0, ctx->stream, t, c, st, units,
                               n_level);
            break;
        case SW_BU_JOINT:
        case SW_BU_JOINT_FUSED:
            hipLaunchKernelGGL((bu_f81_kernel<G, R, true, true>), grid, block, 0, ctx->stream, t, c, st, units,
                               n_level);
            break;
        case SW_BU_JOINT_NOVEC:
        case SW_BU_JOINT_FUSED_NOVEC:
        case SW_BU_CHERRIES_JOINT:
            hipLaunchKernelGGL((bu_f81_kernel<G, R, true, false>), grid, block, 0, ctx->stream, t, c, st, units,
                               n_level);
            break;
        case SW_TD_FUSED: {
            // units narrower than 8 lanes stage their posterior rows in LDS (pml_kernels_f81.h, post_row / post_onehot)
            int stage = 0;
            size_t lds = 0;
            // (measured, 262 144 tips x 32 columns: k = 2 0.59 -> 0.39 ms, k = 4 0.68 -> 0.45, k = 8 0.83 -> 0.71; with four
            // lanes per unit, k = 12 / 16, a loss of 5 - 10 %)
            if (G <= PML_TD_STAGE_MAX_G && (c.ks & 1) == 0) {
                const bool scalars = true;   // (the rows' sums and exponents are staged with them)
                stage = 3;
                // bit 2: some unit of the level has a cherry among its first two children (tip slots in use)
                bool cherries = true;
                if (ctx->units_override == nullptr && !ctx->td_cherry_prefix.empty()) {
                    const size_t a = (size_t)(level - ctx->d_td_parents_f), b = a + (size_t)n_level;
                    if (b < ctx->td_cherry_prefix.size()) cherries = ctx->td_cherry_prefix[b] != ctx->td_cherry_prefix[a];
                }
                if (cherries) stage |= 4;
                lds = (size_t)PML_WAVES_PER_BLOCK * td_stage_doubles(64 / G, c.ks, scalars) * sizeof(double);
            }
            hipLaunchKernelGGL((td_f81_kernel<G, R>), grid, block, lds, ctx->stream, t, c, st, units, n_level, stage);
            break;
        }
        case SW_ROOTS:
            hipLaunchKernelGGL((td_f81_roots_kernel<G, R>), grid, block, 0, ctx->stream, t, c, st);
            break;
        default:
            break;
    }
}


