// Launchers of the joint sweep of the eigen models on the FP64 vector units and of the P(t) batch for fewer than 16 states
// (pml_kernels_eigen_joint.h).
#include "pml_launch.h"
#include "pml_kernels_eigen_joint.h"

// d_offsets == nullptr: one launch over the n nodes of a level (their unit descriptors); otherwise the levels
// [first, first + n) of the level table in one launch (one workgroup per column)
int launch_eigen_joint(pml_ctx* ctx, const PmlUnit* units, const int* d_offsets, int first, int n,
                              const int* d_blk_start, int n_blocks) {
    if (n <= 0) return PML_OK;
    const int KU = 4 * ((ctx->k + 3) / 4);
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const PmlModel m = model_of(ctx);
    const int per_block = PML_WAVES_PER_BLOCK * (64 / ctx->k);
    const int cap_all = (int)ctx->tune.get(T_EIGJ_BLOCKS, 1024);
#define PML_EIGJ_CASE(KU_)                                                                                          \
    if (KU == KU_) {                                                                                                \
        if (d_offsets) {                                                                                            \
            hipLaunchKernelGGL((eigen_joint_narrow_kernel<KU_>), dim3(n_blocks, ctx->C), dim3(PML_BLOCK), 0,        \
                               ctx->stream, t, c, m, st, ctx->d_AinvT, units, d_offsets + first, n, d_blk_start);   \
        } else {                                                                                                    \
            int blocks = (n + per_block - 1) / per_block;                                                           \
            const int cap = std::max(8, cap_all / std::max(1, ctx->C));                                             \
            if (blocks > cap) blocks = cap;                                                                         \
            if (ctx->tune.on(T_NO_EIGJ_PIPE) || KU_ > 32)   /* (beyond 32 states the pipelined form needs a SIMD to itself) */ \
                hipLaunchKernelGGL((eigen_joint_kernel<KU_, false>), dim3(blocks, ctx->C), dim3(PML_BLOCK), 0,      \
                                   ctx->stream, t, c, m, st, ctx->d_AinvT, units, n);                               \
            else                                                                                                    \
                hipLaunchKernelGGL((eigen_joint_kernel<KU_, true>), dim3(blocks, ctx->C), dim3(PML_BLOCK), 0,       \
                                   ctx->stream, t, c, m, st, ctx->d_AinvT, units, n);                               \
        }                                                                                                           \
        HIP_TRY(hipGetLastError());                                                                                 \
        return PML_OK;                                                                                              \
    }
    PML_EIGJ_CASE(4)
    PML_EIGJ_CASE(8)
    PML_EIGJ_CASE(12)
    PML_EIGJ_CASE(16)
    PML_EIGJ_CASE(20)
    PML_EIGJ_CASE(24)
    PML_EIGJ_CASE(28)
    PML_EIGJ_CASE(32)
    PML_EIGJ_CASE(36)
    PML_EIGJ_CASE(40)
    PML_EIGJ_CASE(44)
    PML_EIGJ_CASE(48)
    PML_EIGJ_CASE(52)
    PML_EIGJ_CASE(56)
    PML_EIGJ_CASE(60)
    PML_EIGJ_CASE(64)
#undef PML_EIGJ_CASE
    return fail(PML_ERR_UNSUPPORTED, "no joint eigen kernel for k = %d", ctx->k);
}

int launch_eigen_joint_tips(pml_ctx* ctx) {
    if (ctx->n_tips <= 0) return PML_OK;
    const int KU = 4 * ((ctx->k + 3) / 4);
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlState st = state_of(ctx);
    const PmlModel m = model_of(ctx);
    const int per_block = PML_WAVES_PER_BLOCK * (64 / ctx->k);
    int blocks = (ctx->n_tips + per_block - 1) / per_block;
    const int cap_all = (int)ctx->tune.get(T_EIGJ_TIP_BLOCKS, 2048);
    const int cap = std::max(8, cap_all / std::max(1, ctx->C));
    if (blocks > cap) blocks = cap;
    // observed tips in the lean kernel; what it leaves on the columns' lists (tips with several or all states allowed)
    // in one launch of the general kernel -- PASTML_HIP_EIGJ_ONE_TIPS_KERNEL: everything in the general kernel (round 2)
    const bool one_kernel = ctx->tune.on(T_EIGJ_ONE_TIPS_KERNEL);
    const int rest_blocks = std::min(blocks, std::max(8, 1024 / std::max(1, ctx->C)));
    // (every tip of every column known to be observed -- the masks came from pml_masks_from_tip_states: nothing can be
    // on the lists, their launch is left out)
    bool all_observed = !ctx->tips_observed.empty();
    for (char f : ctx->tips_observed) all_observed = all_observed && f != 0;
#define PML_EIGJ_TIPS(KU_)                                                                                          \
    if (KU == KU_) {                                                                                                \
        if (one_kernel) {                                                                                           \
            hipLaunchKernelGGL((eigen_joint_tips_kernel<KU_>), dim3(blocks, ctx->C), dim3(PML_BLOCK), 0,            \
                               ctx->stream, t, c, m, st, ctx->d_AinvT, ctx->d_tips, ctx->n_tips, nullptr);          \
        } else {                                                                                                    \
            hipLaunchKernelGGL((eigen_joint_obs_tips_kernel<KU_>), dim3(blocks, ctx->C), dim3(PML_BLOCK), 0,        \
                               ctx->stream, t, c, m, st, ctx->d_AinvT, ctx->d_tips, ctx->n_tips, ctx->d_tip_rest,   \
                               ctx->d_tip_rest_count);                                                              \
            if (!all_observed)                                                                                      \
                hipLaunchKernelGGL((eigen_joint_tips_kernel<KU_>), dim3(rest_blocks, ctx->C), dim3(PML_BLOCK), 0,   \
                                   ctx->stream, t, c, m, st, ctx->d_AinvT, ctx->d_tip_rest, ctx->n_tips,            \
                                   ctx->d_tip_rest_count);                                                          \
        }                                                                                                           \
        HIP_TRY(hipGetLastError());                                                                                 \
        return PML_OK;                                                                                              \
    }
    PML_EIGJ_TIPS(4)
    PML_EIGJ_TIPS(8)
    PML_EIGJ_TIPS(12)
    PML_EIGJ_TIPS(16)
    PML_EIGJ_TIPS(20)
    PML_EIGJ_TIPS(24)
    PML_EIGJ_TIPS(28)
    PML_EIGJ_TIPS(32)
    PML_EIGJ_TIPS(36)
    PML_EIGJ_TIPS(40)
    PML_EIGJ_TIPS(44)
    PML_EIGJ_TIPS(48)
    PML_EIGJ_TIPS(52)
    PML_EIGJ_TIPS(56)
    PML_EIGJ_TIPS(60)
    PML_EIGJ_TIPS(64)
#undef PML_EIGJ_TIPS
    return fail(PML_ERR_UNSUPPORTED, "no joint eigen kernel for k = %d", ctx->k);
}


// P(t) of every branch for fewer than 16 states (run_prep)
int launch_pij_valu(pml_ctx* ctx) {
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlModel m = model_of(ctx);
    // vector-unit path: a lane per output row, exact flops, rows written in address order (pml_kernels_eigen_joint.h).
    // Below 16 states, where the matrix-core kernel does not reach (524 287 branches: k = 8 0.289 -> 0.056 ms, k = 5
    // 0.292 -> 0.032 against pij_eigen_kernel); from 16 on the matrix-core kernel is the faster one (k = 20 0.380 against
    // 0.392 ms, k = 32 0.96 against 1.38: the rows of A^T come through the scalar cache a dozen FMAs ahead at best, and
    // with 1 024 FMAs per lane the wave count halves) -- PASTML_HIP_PIJ_VALU forces this path (profiles/r05o_pij_valu.txt)
    const int KU = 4 * ((ctx->k + 3) / 4);
    const long long passes = ((long long)ctx->N * ctx->k + 63) / 64;
    long long blocks = (passes + PML_WAVES_PER_BLOCK - 1) / PML_WAVES_PER_BLOCK;
    const long long cap = std::max(64, (int)ctx->tune.get(T_PIJ_BLOCKS, 4096) / std::max(1, ctx->C));
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks, ctx->C);
#define PML_PIJV_CASE(KU_)                                                                                          \
    if (KU == KU_)                                                                                                  \
hipLaunchKernelGGL((pij_eigen_valu_kernel<KU_>), grid, dim3(PML_BLOCK), 0, ctx->stream, t, c, m, ctx->d_AinvT, \
                   ctx->d_AT, ctx->d_P);
    PML_PIJV_CASE(4)
    PML_PIJV_CASE(8)
    PML_PIJV_CASE(12)
    PML_PIJV_CASE(16)
    PML_PIJV_CASE(20)
    PML_PIJV_CASE(24)
    PML_PIJV_CASE(28)
    PML_PIJV_CASE(32)
#undef PML_PIJV_CASE
    HIP_TRY(hipGetLastError());
    return PML_OK;
}
