// Host side shared by the translation units of libpastml_hip.so: the context behind the opaque pml_ctx, its switches, the
// small helpers every launcher needs.  gfx950 only.  (Round 6: the library was one 5 000-line translation unit that took
// 3.5 minutes to compile; the kernel families now sit in translation units of their own, pml_launch_*.hip, built in
// parallel by pastml_amd/build.py -- this header is what they share with pml_api.hip.)
#pragma once
#include "../../include/pastml_hip.h"

#include <algorithm>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <chrono>
#include <vector>

#include "pml_model.h"   // PmlTree, PmlCols, PmlState, PmlUnit, PmlModel (and the F81 / miscellaneous kernels' headers below it)

struct PmlComm;  // pml_comm.h (pml_api.hip only)

#define PML_VERSION 102

// internal linkage across the library's translation units (not part of the C-ABI)
#define PML_INTERNAL __attribute__((visibility("hidden")))

// records the message pml_last_error() returns on this thread and hands `code` back (defined in pml_api.hip)
PML_INTERNAL int pml_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail pml_fail

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            return fail(PML_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define PML_TRY(expr)            \
    do {                         \
        int _s = (expr);         \
        if (_s != PML_OK) return _s; \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// Every switch of the schedules in one table per context.  The defaults come from the environment (PASTML_HIP_<NAME>) when
// the ctx is created, pml_ctx_set_tunable overrides them for that ctx -- there are no function-local statics: two contexts
// of one process can run different schedules, and a test that sets a switch gets it (round 3 latched several of them at
// their first use in the process).  FLAG: on when present (environment: whatever the value; set_tunable: value != 0).
// TREE: read by pml_tree_upload / pml_chars_alloc, so it must be set before the tree is uploaded.
// ---------------------------------------------------------------------------------------------------------------------
#define PML_TUNABLES(X)                                                                                              \
    X(GRID_CAP, 0, 0) X(SMALL_MANY_NODES, 0, 0) X(BLOCK_MAX_WORK, 0, 0)    \
    X(NO_MFMA, 1, 0) X(NO_EIGEN_FUSED, 1, 0) X(NO_HKY_FUSED, 1, 0)      \
    X(BLOCK_THREADS, 0, 0) X(EIG_BLOCKS, 0, 0) X(NO_EIGEN_GEMM, 1, 0) X(NO_EIGEN_JOINT_VALU, 1, 0) X(EIGJ_BLOCKS, 0, 0) \
    X(EIGJ_TIP_BLOCKS, 0, 0) X(EIGJ_ONE_TIPS_KERNEL, 1, 0) X(EIGJ_TIER_THIN, 0, 1) X(EIGJ_TIER_DEPTH, 0, 1)            \
    X(NO_EIGJ_TIERS, 1, 1) X(NO_BT_TIERS, 1, 1) X(NO_SHAPE_SORT, 1, 1) X(NO_SUPER, 1, 1) X(SUPER_MIN, 0, 1)            \
    X(STACK_MIN, 0, 1) X(NO_STACK, 1, 1) X(DEBUG, 1, 0) X(BLOCK_NODES, 0, 1) X(BLOCK_MAX_STORED, 0, 1)                 \
    X(BLOCK_HEIGHT_CAP, 0, 1) X(SMALL_MAX_NODES, 0, 1) X(F81_R, 0, 1) X(F81_TD_R, 0, 1) X(NO_GRAPH, 1, 1)              \
    X(NARROW_UNITS, 0, 0) X(NO_EIGG_TIERS, 1, 0) X(NO_SPIN_WAIT, 1, 0)   \
    X(PIJ_STAGE_ROWS, 0, 0) X(PIJ_BLOCKS, 0, 0) X(NO_HEIGHT_ORDER, 1, 1) X(NO_TD_TAIL, 1, 0) X(NO_PIJ_VALU, 1, 0) X(PIJ_VALU, 1, 0) X(NO_PIJ_WIDE, 1, 0) X(NO_EIGJ_PIPE, 1, 0) \
    X(THIN_UNITS, 0, 1) X(THIN_BYTES, 0, 1) X(THIN_BLOCK_NODES, 0, 1) X(NO_THIN, 1, 0) X(NO_THIN_WIDE, 1, 0) X(BU_WIDE, 0, 1) X(SORT_LEVELS, 0, 1) X(NO_WIDE_LEAN, 1, 0) X(SHAPE_ORDER, 0, 1)
enum PmlTunable {
#define X(name, flag, tree) T_##name,
    PML_TUNABLES(X)
#undef X
    T_COUNT
};
static const char* const kTunableName[T_COUNT] = {
#define X(name, flag, tree) #name,
    PML_TUNABLES(X)
#undef X
};
static const bool kTunableFlag[T_COUNT] = {
#define X(name, flag, tree) flag != 0,
    PML_TUNABLES(X)
#undef X
};
static const bool kTunableTree[T_COUNT] = {
#define X(name, flag, tree) tree != 0,
    PML_TUNABLES(X)
#undef X
};
struct PmlTune {
    long long val[T_COUNT];
    bool has[T_COUNT];
    PmlTune() {
        for (int i = 0; i < T_COUNT; ++i) {
            const std::string var = std::string("PASTML_HIP_") + kTunableName[i];
            const char* e = getenv(var.c_str());
            has[i] = e != nullptr;
            val[i] = e ? atoll(e) : 0;
        }
    }
    bool on(int i) const { return has[i]; }
    long long get(int i, long long dflt) const { return has[i] ? val[i] : dflt; }
};

struct pml_ctx {
    int device = 0;
    PmlTune tune;  // the schedules' switches (environment at creation, pml_ctx_set_tunable)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // kernel timing (pml_profile_*): event pairs around the level launches, read back when the profile is read -- a
    // bracket never makes the host wait inside a sweep
    struct ProfBracket {
        hipEvent_t a, b;
        int which;
        long long launches;
    };
    std::vector<ProfBracket> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    hipEvent_t prof_open = nullptr;
    bool profile = false;
    double prof_ms[5] = {0, 0, 0, 0, 0};  // bottom-up levels, top-down levels, per-branch pass, two-level launch TD / BU
    long long prof_launches[5] = {0, 0, 0, 0, 0};
    std::vector<void*> allocs;
    size_t held = 0;

    // tree
    int N = 0, n_roots = 0, n_bu_levels = 0, n_td_levels = 0;
    int *d_parent = nullptr, *d_first_child = nullptr, *d_n_children = nullptr, *d_post_rank = nullptr;
    int *d_bu_order = nullptr, *d_td_parents = nullptr;
    int* d_tips = nullptr;  // ids of the tips (the fused eigen sweeps give them a launch of their own)
    int *d_bu_offsets = nullptr, *d_td_offsets = nullptr;  // level tables on the device (narrow end in one launch)
    int n_tips = 0;
    double* d_msg = nullptr;  // fused eigen sweeps: messages of the bottom-up sweep
    int *d_tip_rest = nullptr, *d_tip_rest_count = nullptr;  // eigen joint sweep: [C][n_tips] tips that are not observed, [C]
    double* d_dist = nullptr;
    std::vector<int> bu_offsets, td_offsets, td_parent_offsets, h_parent, h_n_children;
    std::vector<int> h_first_child, h_fh, h_order_f, h_tdp;  // host copies for build_thin_ends (fused heights, fused lists)
    // Internal node numbering (height_order below): the library numbers the nodes of a ragged forest so that the sibling
    // groups a level's units gather lie next to each other; every per-node array that crosses the C-ABI is in the CALLER's
    // numbering and is permuted on the way in / out.  Both empty when the caller's numbering is kept (balanced trees, ...).
    std::vector<int> new_of_old, old_of_new;
    int *d_new_of_old = nullptr, *d_old_of_new = nullptr;   // the same on the device (outputs are permuted there: gather_rows_kernel)
    void* d_stage = nullptr;   // one column's rows in the caller's numbering, on their way out
    size_t stage_bytes = 0;
    // cherry fusion (F81 marginal sweeps): node kinds and level lists over the stored internal nodes only
    bool fuse = true;
    unsigned char* d_kind = nullptr;
    std::vector<unsigned char> h_kind;
    int *d_bu_order_f = nullptr, *d_td_parents_f = nullptr, *d_cherries = nullptr;
    // unit descriptors of the F81 kernels, parallel to d_bu_order_f / d_td_parents_f / d_bu_order
    PmlUnit *d_bu_units_f = nullptr, *d_td_units_f = nullptr, *d_bu_units = nullptr, *d_cherry_units = nullptr;
    // the same fused lists with every level's units sorted by shape (level launches of wide units, see pml_tree_upload)
    PmlUnit *d_bu_units_fs = nullptr, *d_td_units_fs = nullptr;
    int *d_bu_offsets_f = nullptr, *d_td_parent_offsets_f = nullptr;  // level tables for the single-launch kernels
    // subtree blocks (pml_kernels_f81.h, bottom): the stored nodes cut into subtrees of at most PML_BLOCK_NODES stored
    // nodes, walked by one workgroup each, and the "top" above the cuts with level tables of its own
    struct BlockSchedule {
        bool ok = false;
        int n_blocks = 0;
        long long steps = 0;  // sum over the blocks of their levels: workgroup steps of one column's sweep
        PmlUnit *d_bu_units = nullptr, *d_td_units = nullptr;          // units of the blocks, block by block
        int *d_bu_start = nullptr, *d_bu_levels = nullptr, *d_bu_lv = nullptr;
        int *d_td_start = nullptr, *d_td_levels = nullptr, *d_td_lv = nullptr;
        PmlUnit *d_top_bu_units = nullptr, *d_top_td_units = nullptr;  // units of the top part, level by level
        int *d_top_bu_offsets = nullptr, *d_top_td_offsets = nullptr;
        std::vector<int> top_bu_offsets, top_td_offsets;               // host copies (launch geometry)
        std::vector<char> top_bu_vec;                                   // per top level: stored node among children 0, 1
    } blocks;
    // Thin ends of a large ragged forest, units of fewer than 8 lanes (round 5).  Bottom-up: the fused levels from
    // floor_level on (each of at most PASTML_HIP_THIN_UNITS units) in tiers of subtree blocks, like `blocks` but of that
    // part of the forest only and with several small subtrees per workgroup; the wide levels below stay level launches.
    struct ThinSchedule {
        bool ok = false;
        int floor_level = 0;  // the fused levels below stay level launches
        int top_level = 0;    // ... and from this one on they are the narrow end's (level launches where still wide)
        struct Tier { int first_block, n_blocks; };
        std::vector<Tier> tiers;   // runs of levels, each cut into subtrees of at most THIN_BLOCK_NODES units: a launch per tier
        PmlUnit* d_units = nullptr;
        int *d_start = nullptr, *d_levels = nullptr, *d_lv = nullptr;
    } thin;
    // Top-down: the depths from first_depth on (each of at most THIN_UNITS parents): the subtrees hanging at first_depth,
    // packed into bins of about THIN_BLOCK_NODES units, ONE launch walks them all, a workgroup per (bin, column).
    struct DeepSchedule {
        bool ok = false;
        int first_depth = 0, n_blocks = 0;
        PmlUnit* d_units = nullptr;
        int *d_start = nullptr, *d_levels = nullptr, *d_lv = nullptr;
    } deep;
    // two-level units (pml_kernels_f81.h): nodes with two stored children that each carry two cherries of two tips run
    // both levels in one unit; they and their children leave the level lists ("rest" lists, same level structure)
    struct SuperSchedule {
        bool ok = false;
        int n = 0;
        PmlUnit* d_units = nullptr;
        PmlUnit* d_child_units = nullptr;  // the 2 n children of the two-level units, as units of their own (downloads)
        PmlUnit *d_bu_units_r = nullptr, *d_td_units_r = nullptr;
        PmlUnit *d_bu_units_rs = nullptr, *d_td_units_rs = nullptr;  // ... sorted by shape inside every level
        // stacked units (pml_kernels_f81.h): nodes with two plain stored children of two stored children each, by
        // bottom-up level and by depth; their children as units of their own for downloads
        int n_child_units = 0;  // entries of d_child_units: the children of the two-level units
        int n_stack = 0;
        PmlUnit *d_stack_bu = nullptr, *d_stack_td = nullptr, *d_stack_children = nullptr;
        std::vector<int> stack_bu_offsets, stack_td_offsets;
        int *d_bu_offsets_r = nullptr, *d_td_offsets_r = nullptr;
        std::vector<int> bu_offsets_r, td_offsets_r;
        std::vector<char> bu_level_vec_r;
    } sup;
    // Joint sweep of the eigen models: the thin levels of a large forest (runs of levels of at most 4 096 nodes) in tiers
    // of four levels; a tier is cut into subtree blocks and ONE launch walks them, a workgroup per (block, column) with a
    // workgroup barrier between its levels -- a level costs a ~3.5 us pass instead of a ~7.5 us dependent launch.
    struct EigenTiers {
        bool ok = false;
        int first_level = 0;   // plain bottom-up level the first tier starts at
        int top_level = 0;     // ... and the level from which the single-workgroup launch takes over
        struct Tier { int first_block, n_blocks, depth; };
        std::vector<Tier> tiers;
        PmlUnit* d_units = nullptr;
        int *d_lv = nullptr, *d_start = nullptr;
        int* d_nodes = nullptr;  // the node ids parallel to d_units (the sum sweeps walk node lists)
        int widest = 0;        // nodes of the widest level inside the tiers
    } eig_tiers;
    // joint back-trace: the depths beyond its single-workgroup launch in tiers of subtrees (joint_backtrace_blocks_kernel)
    struct BacktraceTiers {
        bool ok = false;
        int first_depth = 0;  // depths 1 .. first_depth - 1 stay with the single-workgroup launch
        struct Tier { int first_block, n_blocks, depth; };
        std::vector<Tier> tiers;
        int *d_nodes = nullptr, *d_lv = nullptr, *d_start = nullptr;
    } bt_tiers;
    bool small = false;  // forest small enough for the one-launch-per-sweep kernels
    bool levels_fit_workgroup = false;  // (nearly) every fused level is one pass of a 512-thread workgroup
    std::vector<int> bu_offsets_f, td_parent_offsets_f;
    std::vector<char> bu_level_vec_f;  // per fused bottom-up level: some unit has a stored node as child 0 or 1
    std::vector<int> td_cherry_prefix; // over the fused top-down units: how many before it have a cherry as child 0 or 1
    std::vector<char> bu_level_vec;    // the same for the plain levels (joint sweep: every internal node is stored)
    int n_cherries = 0;
    bool bu_fused = false;  // the last bottom-up sweep left the cherries unmaterialised
    bool bu_absorbed = false;  // ... and the children of the two-level units
    bool bu_fused_joint = false;  // ... and it was a joint sweep

    // columns
    int C = 0, k = 0, ks = 0, W = 0, G = 0, R = 0;
    int Gf = 0, Rf = 0;  // lane-group shape of the F81-family bottom-up kernels (chunked state ownership)
    bool bu_wide_lanes = false;  // 32 < k <= 64: most bottom-up levels run with 8 states per lane (see dispatch_sweep)
    bool shape_ordered = false;   // the forest's numbering orders a depth's sibling groups by (shape, class) of the gathering unit
    bool level_lists_sorted = false;  // 32 < k <= 64: the level launches walk the lists sorted by shape (pml_tree_upload)
    int Gt = 0, Rt = 0;  // ... and of the F81-family top-down kernels
    u64 *d_masks = nullptr, *d_masks_init = nullptr;
    bool has_init = false;
    int kind = -1;
    double *d_pi = nullptr, *d_mu = nullptr, *d_kappa = nullptr, *d_d = nullptr, *d_A = nullptr, *d_Ainv = nullptr;
    double* d_active = nullptr;   // last array of the parameter block: 0.0 = the column sits the next bottom-up sweep out
    bool active_partial = false;  // ... some column does (pml_bottom_up_submit_columns)
    bool in_bu_enqueue = false;   // the launches being enqueued are a bottom-up sweep's: they look at the flags
    int n_active = 0;             // columns that take part in the next sweep
    int sched_cols = 0;           // the number of columns the schedule of a sweep is chosen for (C; 32 for a few active ones)
    bool bu_signals_few = false;
    double* d_Asym = nullptr;   // [C][k][k], 65 <= k <= 128: the one matrix of the sum sweeps (eig_sym_kernel)
    double* d_eigT = nullptr;   // [C][k][k]: scratch of that kernel
    std::vector<double> h_symA; // [C][k k + k]: the A and pi d_Asym was made from (an unchanged model is not orthonormalised again)
    std::vector<char> eig_sym;  // per column: the identity holds for the matrices pml_model_set_eigen was given
    bool eig_sym_all = false;   // ... for every column: the sum sweeps of 65 - 128 states keep one matrix in LDS
    double* d_AinvT = nullptr;  // [C][ld][ld], ld = 32 (k <= 32) or 64 (k <= 64): Ainv transposed and zero-padded -- eigen_joint_kernel (k <= 32), eigen_gemm_kernel's observed tips
    double* d_AT = nullptr;     // [C][32][32]: A transposed and zero-padded (k <= 32), for pij_eigen_valu_kernel
    double *d_sf = nullptr, *d_tau = nullptr, *d_tauf = nullptr;
    std::vector<char> model_set;  // per column
    std::vector<char> tips_observed;  // per column: every tip has exactly one allowed state (known from pml_masks_from_tip_states)
    bool prep_dirty = true;

    // state
    double *d_E = nullptr, *d_P = nullptr, *d_bu = nullptr, *d_S = nullptr, *d_td = nullptr, *d_post = nullptr,
           *d_lhsum = nullptr;
    i64 *d_be = nullptr, *d_te = nullptr, *d_lhe = nullptr;
    pml_jt* d_J = nullptr;  // arg-max tables, one byte per entry (two beyond 256 states)
    int* d_js = nullptr;
    u64* d_err = nullptr;
    int bu_mode = -1;  // -1 invalid, 1 marginal, 0 joint
    bool js_ever = false;  // joint states of some earlier joint sweep are still in d_js
    bool post_ever = false;  // posteriors of some earlier top-down sweep are still in d_post
    int* d_nsel = nullptr;
    // hipGraph replay of the launch sequence of a sweep (level kernels are launch-bound on mid-size trees)
    struct GraphSlot {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        bool has_init = false;
        bool has_params = false;  // the captured sequence starts with the copy of the parameter block (params_push)
    };
    bool capture_saw_params = false;
    GraphSlot bu_graph[2], td_graph, bt_graph;
    GraphSlot bu_graph_few;        // the marginal sweep as scheduled for a few active columns (submit_bottom_up)
    GraphSlot mp_graph;            // bottom-up + top-down of pml_marginal_pass as ONE graph
    bool in_outer_capture = false; // the sweeps are being captured into mp_graph: no graphs of their own
    bool graphs = true;
    double* h_loglik = nullptr;  // pinned staging of the per-column results
    // pi, sf, tau, tau factor, mu, kappa of all columns live in ONE device block with a pinned host mirror of the same
    // layout: a parameter update (every optimiser step) is one asynchronous copy and no synchronisation
    double *d_params = nullptr, *h_params = nullptr;
    bool params_dirty = false;  // the pinned mirror holds values the device block has not seen (params_push sends them)
    bool capturing = false;     // a sweep's launch sequence is being captured into a graph
    size_t n_params = 0;
    u64* h_err = nullptr;
    // completion of a bottom-up sweep whose last launch is the single-workgroup-per-column kernel: that kernel raises a
    // word in pinned memory when its last column is done (bu_f81_small_kernel), and the collect spins on it
    u64* h_done = nullptr;      // pinned: generation of the last finished launch
    u64* d_done = nullptr;      // device: [0] columns done in the running launch, [1] generation
    u64 done_expect = 0;        // what *h_done shows when the sweep submitted last has finished
    bool enqueue_signals = false;          // set by the launcher while a sweep is enqueued
    bool bu_signals[2] = {false, false};   // per captured sweep (joint / marginal): its last launch signals
    bool wait_signal = false;              // the sweep submitted last signals
    // the same for a whole marginal pass: its last top-down launch signals where the schedule ends in a multi-level
    // kernel (single-launch sweeps, subtree blocks); signals_enqueued counts the signalling launches of what is being
    // enqueued (the bottom-up sweep's and the top-down sweep's), mp_signals / mp_final keep them for the captured pass
    bool signal_next_td = false, td_final_signals = false, mp_final = false, mp_wants_signal = false;
    int signals_enqueued = 0, mp_signals = 0;
    bool td_valid = false, js_valid = false;
    bool keep_td = false;      // PML_OPT_KEEP_TD (or a pml_download of the TD vectors asked for them)
    bool td_vec_valid = false; // the TD vectors of the last top-down sweep are in d_td
    bool td_filled = false;    // ... including those of the nodes the sweeps do not store (td_fill_kernel)
    bool eig_fused_opt = true; // PML_OPT_EIGEN_FUSED
    bool eigj_valu_opt = true; // PML_OPT_EIGEN_JOINT_VALU
    bool implicit_tips = false;    // PML_OPT_IMPLICIT_TIP_POSTERIORS
    bool tip_post_missing = false; // the last top-down sweep left the observed tips' posteriors implicit
    const PmlUnit* units_override = nullptr;  // set around a dispatch_sweep on the block schedule's top lists

    PmlComm* comm = nullptr;   // RCCL communicator attached by pml_comm_init (survives tree uploads)
};

// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
static int dev_alloc(pml_ctx* ctx, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T));
    if (e != hipSuccess)
        return fail(PML_ERR_HIP, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
    ctx->allocs.push_back(q);
    ctx->held += count * sizeof(T);
    *p = (T*)q;
    return PML_OK;
}

static void drop_graph(pml_ctx::GraphSlot& g) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
    g.exec = nullptr;
    g.graph = nullptr;
}

// every captured launch sequence of the two sweeps (the back-trace's graph depends on the tree and the tunables only)
static void drop_sweep_graphs(pml_ctx* ctx) {
    drop_graph(ctx->bu_graph[0]);
    drop_graph(ctx->bu_graph[1]);
    drop_graph(ctx->bu_graph_few);
    drop_graph(ctx->td_graph);
    drop_graph(ctx->mp_graph);
}

static void free_all(pml_ctx* ctx) {
    drop_sweep_graphs(ctx);
    drop_graph(ctx->bt_graph);
    if (ctx->h_loglik) (void)hipHostFree(ctx->h_loglik);
    if (ctx->h_err) (void)hipHostFree(ctx->h_err);
    if (ctx->h_done) (void)hipHostFree(ctx->h_done);
    ctx->h_done = nullptr;
    if (ctx->h_params) (void)hipHostFree(ctx->h_params);
    ctx->h_params = nullptr;
    ctx->h_loglik = nullptr;
    ctx->h_err = nullptr;
    for (void* p : ctx->allocs) (void)hipFree(p);
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    ctx->d_stage = nullptr;
    ctx->stage_bytes = 0;
    ctx->allocs.clear();
    ctx->held = 0;
}

template <typename T>
static int upload(pml_ctx* ctx, T* dst, const T* src, size_t count) {
    HIP_TRY(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return PML_OK;
}

static void pick_group(const pml_ctx* ctx, int k, int& G, int& R) {
    R = k <= 32 ? 1 : (k <= 128 ? 2 : (k <= 256 ? 4 : 8));   // (beyond 256 states: the F81 family only, a wavefront per unit)
    if (k > 16 && k <= 32) R = 4;  // 8 lanes per unit: 8 units per wavefront
    const int need = (k + R - 1) / R;
    G = 1;
    while (G < need) G <<= 1;
}

// Blocks along x for a level of n_units units per column (grid-stride loops take the rest).  The cap on the total
// number of blocks was measured on cfg4 (MI355X): the pipelined bottom-up kernels like ~8192 (a wave then walks several
// units and its prefetch stage pays off), everything else 32768; persistent-size grids (768-2048) were 5-15 % slower.
static int grid_for(const pml_ctx* ctx, int n_units, int units_per_block, int C, bool pipelined = false) {
    int blocks = (n_units + units_per_block - 1) / units_per_block;
    const int cap_env = (int)ctx->tune.get(T_GRID_CAP, 0);
    const int total_cap = cap_env > 0 ? cap_env : (pipelined ? 8192 : 32768);
    int cap = total_cap / (C < 1 ? 1 : C);
    if (cap < 8) cap = 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return blocks;
}

// Whole F81 sweeps in ONE launch (one workgroup per column walks every level): forests of up to
// PASTML_HIP_SMALL_MAX_NODES (2048) nodes, where a sweep is otherwise a chain of latency-bound launches -- and, when
// there are many columns (the optimiser's batches: one workgroup per column already fills the chip), forests of up to
// PASTML_HIP_SMALL_MANY_NODES (16384) nodes whose levels each fit one pass of a workgroup (deep, ragged trees:
// HIV1C-sized sweeps -- 7 237 nodes, 57 height levels -- of 246 binary columns 0.34 -> 0.27 ms; a balanced 4 096-tip
// tree with 20 states has levels of 16 passes and stays with the level kernels: 0.12 against 0.27 ms).  Same unit
// functions and lane shapes as the level kernels: identical bits.
// (more than 256 states -- round 6, the F81 family only, 64 lanes x 8 states -- run the plain level schedule: their one lane
// shape is instantiated for the level kernels alone)
static inline bool wide_states(const pml_ctx* c) { return c->k > 256; }

static bool single_launch_sweeps(const pml_ctx* c) {
    if (wide_states(c)) return false;
    const int many = (int)c->tune.get(T_SMALL_MANY_NODES, 16384);
    return c->small || (c->sched_cols >= 64 && c->N <= many && c->levels_fit_workgroup);
}

// The subtree-block schedule pays where a sweep is a chain of latency-bound launches; once the levels carry enough work
// to fill the chip (stored nodes x columns beyond ~1.6e5: measured on 16 384 - 131 072-tip trees with 1 - 32 columns)
// the level kernels, which spread every level over all compute units, win again.
static bool block_schedule(const pml_ctx* c) {
    const long long limit = c->tune.get(T_BLOCK_MAX_WORK, 160000);
    // (Round 2 also capped the number of (block, level, column) workgroup steps: with 512-thread workgroups a ragged tree
    // times many columns ran in rounds of long-lived workgroups and lost to the level kernels.  The workgroups now
    // shrink until all are resident (launch_blocks_f81) and the blocks end below the top's lowest level
    // (pml_tree_upload): over scripts/schedule_sweep.py's grid the blocks never lose -- profiles/r03c_schedule_sweep.txt.)
    return c->blocks.ok && !wide_states(c) && !c->bu_offsets_f.empty() && (long long)c->bu_offsets_f.back() * c->sched_cols <= limit;
}

static PmlTree tree_of(const pml_ctx* c, bool fused = false) {
    PmlTree t;
    t.kind = fused ? c->d_kind : nullptr;
    t.N = c->N;
    t.n_roots = c->n_roots;
    t.parent = c->d_parent;
    t.first_child = c->d_first_child;
    t.n_children = c->d_n_children;
    t.dist = c->d_dist;
    t.post_rank = c->d_post_rank;
    return t;
}

static PmlCols cols_of(const pml_ctx* c) {
    PmlCols s;
    s.k = c->k;
    s.ks = c->ks;
    s.W = c->W;
    s.no_wide_lean = c->tune.on(T_NO_WIDE_LEAN) ? 1 : 0;
    s.masks = c->d_masks;
    s.masks_init = c->has_init ? c->d_masks_init : nullptr;
    s.pi = c->d_pi;
    s.active = c->in_bu_enqueue ? c->d_active : nullptr;  // (only the sweep itself: downloads rebuild what they need for all)
    return s;
}

static PmlState state_of(const pml_ctx* c) {
    PmlState s;
    s.E = c->d_E;
    s.bu = c->d_bu;
    s.S = c->d_S;
    s.be = c->d_be;
    // F81 family: the top-down sweep runs on the stored posteriors; TD vectors are written only on request
    const bool td_stored = c->kind != PML_MODEL_F81 || c->keep_td;
    s.td = td_stored ? c->d_td : nullptr;
    s.te = td_stored ? c->d_te : nullptr;
    s.post = c->d_post;
    s.implicit_tips = c->implicit_tips && c->kind == PML_MODEL_F81 && c->W == 1;
    s.lhsum = c->d_lhsum;
    s.lhe = c->d_lhe;
    s.J = c->d_J;
    s.jt16 = c->k > 256;
    s.js = c->d_js;
    s.err = c->d_err;
    s.msg = c->d_msg;
    return s;
}

// Eigen models with 16 <= k <= 32 run the fused matrix-core sweeps (pml_kernels_eigen_mfma.h): P(t) is built and
// consumed in registers.  PASTML_HIP_NO_MFMA / PASTML_HIP_NO_EIGEN_FUSED fall back to the materialised-P kernels.
static bool eigen_fused(const pml_ctx* c) {
    const bool off = c->tune.on(T_NO_MFMA) || c->tune.on(T_NO_EIGEN_FUSED);
    return !off && c->eig_fused_opt && c->kind == PML_MODEL_EIGEN && c->k >= 16 && c->k <= 32 && c->W == 1 &&
           c->ks == 4 * ((c->k + 3) / 4);
}

// HKY sweeps build P(t) in registers (pml_kernels_matrix.h, PML_P_HKY); PASTML_HIP_NO_HKY_FUSED reads the batch.
static bool hky_fused(const pml_ctx* c) {
    const bool off = c->tune.on(T_NO_HKY_FUSED);
    return !off && c->kind == PML_MODEL_HKY && c->k == 4 && c->ks == 4 && c->G == 4 && c->R == 1 && c->W == 1;
}

static PmlModel model_of(const pml_ctx* c) {
    PmlModel m;
    m.kind = c->kind;
    m.mu = c->d_mu;
    m.kappa = c->d_kappa;
    m.d = c->d_d;
    m.A = c->d_A;
    m.Ainv = c->d_Ainv;
    m.AinvT = c->d_AinvT;
    m.Asym = c->d_Asym;
    m.ldT = c->k <= PML_EIGJ_STRIDE ? PML_EIGJ_STRIDE : 64;
    m.sf = c->d_sf;
    m.tau = c->d_tau;
    m.tauf = c->d_tauf;
    return m;
}

static int prof_event(pml_ctx* ctx, hipEvent_t* out) {
    if (!ctx->prof_pool.empty()) {
        *out = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
        return PML_OK;
    }
    HIP_TRY(hipEventCreate(out));
    return PML_OK;
}

// adds up the brackets recorded so far (waits for the stream) and returns their events to the pool
static int prof_drain(pml_ctx* ctx) {
    if (ctx->prof_pending.empty()) return PML_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    for (const pml_ctx::ProfBracket& br : ctx->prof_pending) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(br.b));
        HIP_TRY(hipEventElapsedTime(&ms, br.a, br.b));
        ctx->prof_ms[br.which] += ms;
        ctx->prof_launches[br.which] += br.launches;
        ctx->prof_pool.push_back(br.a);
        ctx->prof_pool.push_back(br.b);
    }
    ctx->prof_pending.clear();
    return PML_OK;
}

static void prof_release(pml_ctx* ctx) {
    for (const pml_ctx::ProfBracket& br : ctx->prof_pending) {
        (void)hipEventDestroy(br.a);
        (void)hipEventDestroy(br.b);
    }
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    if (ctx->prof_open) (void)hipEventDestroy(ctx->prof_open);
    ctx->prof_pending.clear();
    ctx->prof_pool.clear();
    ctx->prof_open = nullptr;
}

static int prof_begin(pml_ctx* ctx) {
    if (!ctx->profile) return PML_OK;
    if (!ctx->prof_open) PML_TRY(prof_event(ctx, &ctx->prof_open));
    HIP_TRY(hipEventRecord(ctx->prof_open, ctx->stream));
    return PML_OK;
}

static int prof_end(pml_ctx* ctx, int which, long long launches) {
    if (!ctx->profile || !ctx->prof_open) return PML_OK;
    hipEvent_t b = nullptr;
    PML_TRY(prof_event(ctx, &b));
    HIP_TRY(hipEventRecord(b, ctx->stream));
    ctx->prof_pending.push_back({ctx->prof_open, b, which, launches});
    ctx->prof_open = nullptr;
    if (ctx->prof_pending.size() >= 4096) PML_TRY(prof_drain(ctx));  // (bounds the number of live events)
    return PML_OK;
}
