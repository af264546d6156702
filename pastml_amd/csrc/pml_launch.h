// What pml_api.hip (schedules, C-ABI) calls of the kernel launchers, which live in translation units of their own
// (pml_launch_*.hip, one per kernel family, compiled in parallel): the dispatch functions, the lane-shape rules they share
// and the predicates that say which family a context's sweeps take.
#pragma once
#include "pml_host.h"

// ---------------------------------------------------------------------------------------------------------------------
// (G, R) dispatch
// ---------------------------------------------------------------------------------------------------------------------
#define PML_GR_CASES(X)  \
    X(8, 4)              \
    X(1, 1)              \
    X(2, 1)              \
    X(4, 1)              \
    X(8, 1)              \
    X(16, 1)             \
    X(64, 1)             \
    X(32, 2)             \
    X(64, 2)             \
    X(64, 4)

enum SweepKind {
    SW_BU_MARG, SW_BU_JOINT, SW_TD, SW_ROOTS, SW_BU_MARG_FUSED, SW_TD_FUSED, SW_BU_CHERRIES,
    SW_BU_MARG_FUSED_NOVEC,  // a fused level none of whose units has a stored node among its first two children
    SW_BU_JOINT_NOVEC,       // the same for a level of the joint sweep (the level whose children are all tips)
    SW_BU_JOINT_FUSED, SW_BU_JOINT_FUSED_NOVEC,  // joint sweep over the cherry-fused level lists
    SW_BU_CHERRIES_JOINT     // materialises the cherries' vectors after a fused joint sweep
};


#define PML_F81_CASES(X) \
    X(2, 2)              \
    X(4, 2)              \
    X(1, 1)              \
    X(1, 2)              \
    X(1, 4)              \
    X(2, 4)              \
    X(4, 4)              \
    X(8, 4)              \
    X(16, 4)             \
    X(32, 4)             \
    X(64, 4)             \
    X(32, 2)             \
    X(16, 2)             \
    X(8, 8)


#define PML_SUPER_CASES(X) \
    X(8, 4)                \
    X(16, 4)               \
    X(8, 8)

// lane shape of the kernels that walk several levels in one launch.  Bottom-up: the shape the level kernels use for
// levels of this size (dispatch_sweep: 8 states per lane up to 65 536 units when 32 < k <= 64).  The reductions over a
// unit's lanes associate differently in different shapes, so a level must get the same shape whether it runs here or in
// a level launch: where the narrow end begins depends on the number of columns, and a column's bits must not.
static void multi_level_shape(const pml_ctx* ctx, bool bottom_up, int& g, int& r) {
    g = bottom_up ? (ctx->bu_wide_lanes ? 8 : ctx->Gf) : ctx->Gt;
    r = bottom_up ? (ctx->bu_wide_lanes ? 8 : ctx->Rf) : ctx->Rt;
}


// Two-level units run in the lane shape the level kernels give the levels they replace where those do not stream stored
// vectors (bottom-up: 8 states per lane for 32 < k <= 64) -- units of 8 lanes and more, single-word masks.
static void super_shape(const pml_ctx* ctx, bool bottom_up, int& g, int& r) {
    g = bottom_up ? (ctx->bu_wide_lanes ? 8 : ctx->Gf) : ctx->Gt;
    r = bottom_up ? (ctx->bu_wide_lanes ? 8 : ctx->Rf) : ctx->Rt;
}

static bool super_units(const pml_ctx* ctx) {
    if (!ctx->sup.ok || ctx->kind != PML_MODEL_F81 || ctx->W != 1) return false;
    int g, r;
    super_shape(ctx, true, g, r);
    if (g < 8 || (g == 16 && r == 2)) return false;   // (16 x 2: the shape of forests with polytomies, no two-level kernels)
    super_shape(ctx, false, g, r);
    return g >= 8 && !(g == 16 && r == 2);
}

// the sweeps of this context run the level schedule with two-level units (not one launch per sweep, not subtree blocks)
static bool super_sweeps(const pml_ctx* ctx) {
    return super_units(ctx) && !single_launch_sweeps(ctx) && !block_schedule(ctx);
}

// the thin ends of a large forest as subtree blocks (pml_tree_upload).  Measured, marginal pass, default against NO_THIN
// (profiles/r05u_thin_ends.txt): 100 000 tips with polytomies x 16 characters k = 4 1.43 -> 1.16 ms, k = 20 1.79 -> 1.55;
// random binary 40 000 tips x 8 k = 4 0.368 -> 0.313, k = 64 0.587 -> 0.504; 262 144 tips x 32 k = 4 1.60 -> 1.54,
// k = 12 2.54 -> 2.49, k = 20 3.50 -> 3.43 (there the wide levels dominate).  NO_THIN_WIDE: units of fewer than 8 lanes only.
static bool thin_bottom_up(const pml_ctx* ctx) {
    return ctx->thin.ok && ctx->kind == PML_MODEL_F81 && !ctx->tune.on(T_NO_THIN) &&
           ((!ctx->bu_wide_lanes && ctx->Gf < 8) || !ctx->tune.on(T_NO_THIN_WIDE));
}
static bool deep_top_down(const pml_ctx* ctx) {
    return ctx->deep.ok && ctx->kind == PML_MODEL_F81 && !ctx->tune.on(T_NO_THIN) && (ctx->Gt < 8 || !ctx->tune.on(T_NO_THIN_WIDE));
}


// sum sweeps of the eigen models without forming P(t) (pml_kernels_eigen_gemm.h): one launch over a list (nodes) or a
// contiguous id range (first) of n nodes
// (any eigen model with up to 64 states; 65 - 128 states: the reversible ones -- pml_model_set_eigen checks what it is given --,
// whose A^-1 is A transposed and rescaled, so that one matrix in LDS serves both products)
static bool eigen_gemm(const pml_ctx* c) {
    const bool off = c->tune.on(T_NO_EIGEN_GEMM) || c->tune.on(T_NO_MFMA) || c->tune.on(T_NO_EIGEN_FUSED);
    if (off || !c->eig_fused_opt || c->kind != PML_MODEL_EIGEN || c->k < 2) return false;
    if (c->k <= 64) return c->W == 1;
    return c->k <= 128 && c->W == 2 && c->eig_sym_all && c->d_Asym != nullptr;
}


// The joint sweep of the eigen models on the vector units (pml_kernels_eigen_joint.h) for 2 <= k <= 64;
// PASTML_HIP_NO_EIGEN_JOINT_VALU keeps the matrix-core kernels (pml_kernels_eigen_mfma.h).
static bool eigen_joint_valu(const pml_ctx* c) {
    const bool off = c->tune.on(T_NO_EIGEN_JOINT_VALU);
    return !off && c->eigj_valu_opt && c->kind == PML_MODEL_EIGEN && c->k >= 2 && c->k <= 64 && c->W == 1 &&
           c->d_AinvT != nullptr;
}


// ---- pml_launch_matrix.hip: sweeps of the models with a materialised (or closed-form 4 x 4) P(t), state selection
PML_INTERNAL int dispatch_sweep_matrix(pml_ctx* ctx, SweepKind what, const int* level, int n_level);
PML_INTERNAL int dispatch_select(pml_ctx* ctx, int method, int force_joint, const u64* d_lh_mask);
// ---- pml_launch_f81_level.hip: F81-family level launches
PML_INTERNAL int dispatch_sweep_f81(pml_ctx* ctx, SweepKind what, const int* level, int n_level);
// ---- pml_launch_f81_wide.hip: the same for more than 256 states (64 lanes x 8 states)
PML_INTERNAL int dispatch_sweep_f81_wide(pml_ctx* ctx, SweepKind what, const int* level, int n_level);
// ---- pml_launch_f81_small.hip / pml_launch_f81_blocks.hip: several levels in one launch (whole sweeps of small forests and
//      the narrow ends; subtree blocks and the thin ends)
PML_INTERNAL int dispatch_small_f81(pml_ctx* ctx, bool bottom_up, int do_prep, int first_level = 0, int n_levels = -1,
                                    const PmlUnit* units = nullptr, const int* d_offsets = nullptr, int skip_roots = 0);
PML_INTERNAL int dispatch_blocks_f81(pml_ctx* ctx, bool bottom_up, int which = 0);
// ---- pml_launch_f81_super.hip: two-level and stacked units
PML_INTERNAL int dispatch_super_f81(pml_ctx* ctx, bool bottom_up);
PML_INTERNAL int dispatch_stack_f81(pml_ctx* ctx, bool bottom_up, int level);
// ---- pml_launch_eigen_mfma.hip: fused FP64 matrix-core sweeps of the eigen models, P(t) batch on the matrix cores
PML_INTERNAL int launch_eigen_fused(pml_ctx* ctx, int mode, const int* nodes, int first, int n, int tips);
PML_INTERNAL int launch_eigen_narrow(pml_ctx* ctx, int mode, const int* nodes, const int* d_offsets, int first_level, int n_levels);
PML_INTERNAL int launch_eigen_tips(pml_ctx* ctx, int joint);
PML_INTERNAL int launch_pij_mfma(pml_ctx* ctx);
PML_INTERNAL int launch_pij_wide(pml_ctx* ctx);
// ---- pml_launch_eigen_gemm.hip: sum sweeps as two small GEMMs per 16 nodes
PML_INTERNAL int launch_eigen_gemm(pml_ctx* ctx, int mode, const int* nodes, int first, int n);
PML_INTERNAL int launch_eigen_gemm_narrow(pml_ctx* ctx, int mode, const int* nodes, const int* d_offsets, int first_level,
                                          int n_levels, const int* d_blk_start = nullptr, int n_blocks = 1);
// ---- pml_launch_eigen_gemm_wide.hip: the same for 65 - 128 states (one matrix in LDS)
PML_INTERNAL int launch_eigen_gemm_wide(pml_ctx* ctx, int mode, const int* nodes, int first, int n);
PML_INTERNAL int launch_eigen_gemm_narrow_wide(pml_ctx* ctx, int mode, const int* nodes, const int* d_offsets, int first_level,
                                               int n_levels, const int* d_blk_start, int n_blocks);
// ---- pml_launch_eigen_joint.hip: joint sweep of the eigen models on the vector units, P(t) batch for k < 16
PML_INTERNAL int launch_eigen_joint(pml_ctx* ctx, const PmlUnit* units, const int* d_offsets, int first, int n,
                                    const int* d_blk_start = nullptr, int n_blocks = 1);
PML_INTERNAL int launch_eigen_joint_tips(pml_ctx* ctx);
PML_INTERNAL int launch_pij_valu(pml_ctx* ctx);

// More than 64 KB of dynamic LDS must be asked for: once per kernel, device and size (the largest asked for so far is what is
// set) -- not per launch: the call is not free and should not sit inside a stream capture.  (A template: one table per kernel
// type and translation unit; the kernels that share a type are told apart by their address.)
template <typename K>
static inline int with_lds(const pml_ctx* ctx, K kernel, size_t bytes) {
    struct Entry {
        const void* fn;
        int device;
        size_t bytes;
    };
    static std::mutex mu;
    static std::vector<Entry> done;
    std::lock_guard<std::mutex> lock(mu);
    Entry* e = nullptr;
    for (auto& d : done)
        if (d.fn == (const void*)kernel && d.device == ctx->device) e = &d;
    if (e != nullptr && e->bytes >= bytes) return PML_OK;
    HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (e != nullptr) e->bytes = bytes;
    else done.push_back(Entry{(const void*)kernel, ctx->device, bytes});
    return PML_OK;
}

static inline int dispatch_sweep(pml_ctx* ctx, SweepKind what, const int* level, int n_level) {
    if (n_level <= 0) return PML_OK;
    return ctx->kind == PML_MODEL_F81 ? dispatch_sweep_f81(ctx, what, level, n_level) : dispatch_sweep_matrix(ctx, what, level, n_level);
}
