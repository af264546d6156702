// libpastml_hip.so -- C-ABI (include/pastml_hip.h) over the HIP kernels: contexts, tree upload and schedules, the sweeps'
// launch sequences, downloads, the communicator.  The kernel families are launched through pml_launch.h.  gfx950 only.
#define PML_PLAIN_KERNELS   // the kernels that are not templates are this unit's (pml_device.h, PML_GLOBAL)
#include "pml_launch.h"
#include "pml_kernels_counts.h"
#include "pml_kernels_eigen_gemm.h"   // (eig_sym_kernel)
#include "pml_comm.h"

// sha256 (first 16 hex digits) over the sources this library was compiled from, handed in by pastml_amd/build.py; the
// marker in front lets build.py read it out of the file without loading it
#ifndef PML_BUILD_DIGEST
#define PML_BUILD_DIGEST "unknown"
#endif
static const char kBuildDigest[] = "PML_BUILD_DIGEST=" PML_BUILD_DIGEST;

static thread_local std::string g_last_error;

int pml_fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// ---------------------------------------------------------------------------------------------------------------------
// Height-ordered numbering (round 5).  The C-ABI asks for breadth-first ids -- roots first, the children of a node
// contiguous, every depth a contiguous id range -- which leaves the ORDER OF THE SIBLING GROUPS INSIDE A DEPTH free.  The
// sweeps walk the nodes by fused height (bottom-up) and gather, per unit, 8-byte scalars of the unit's children (E, S, mask,
// exponent) and of the tips under its cherry children; in plain breadth-first order the children of the units of ONE height
// are scattered over their depth, so every gathered scalar costs a 128-byte line of its own (a random 262 144-tip tree moved
// 28.8 GB per marginal pass where the schedule needs 22.2, profiles/r04b_*).  Here the sibling groups of a depth are ordered
// by the height class of the unit that gathers them -- a stored node's children by its fused height, the tips of a cherry by
// the fused height of the cherry's parent -- and, inside a class, in the order of their parents: the children of consecutive
// units of a level are then consecutive in memory.  Same tree, same arithmetic per node, same bits (the order of a node's own
// children is kept); k = 64: marginal pass 5.8 -> 5.2 ms, k = 12: 3.25 -> 2.48, k = 4: 2.39 -> 1.68 (262 144 random tips x 32
// characters, profiles/r05e_height_order.txt).  A balanced tree is in this order already.
// Returns false (and leaves the vectors empty) when the caller's numbering is the height order.
// ---------------------------------------------------------------------------------------------------------------------
// by_shape (round 6): inside a depth the sibling groups are ordered by (shape, height class) of the unit that gathers them instead of
// by the class alone -- the level launches of wide units walk their lists sorted by shape (units_by_shape), so only then are the
// children of CONSECUTIVE units of a launch consecutive in memory (the top-down lists are sorted by (shape, class) to match).
static bool height_order(int N, int R, const int* parent, const int* first_child, const int* n_children,
                         const int* td_offsets, int n_td_levels, bool fuse, bool by_shape, std::vector<int>& old_of_new,
                         std::vector<int>& new_of_old) {
    old_of_new.clear();
    new_of_old.clear();
    std::vector<int> cls(N, 0), fh(N, 0);
    std::vector<char> stored(N, 0);
    for (int i = 0; i < N; ++i) {
        if (n_children[i] == 0) continue;
        bool all_tips = true;
        for (int j = 0; j < n_children[i]; ++j) all_tips &= n_children[first_child[i] + j] == 0;
        stored[i] = !(fuse && all_tips && parent[i] >= 0);
    }
    for (int i = N - 1; i >= 0; --i) {  // children have larger ids than their parent
        if (!stored[i]) continue;
        int h = 0;
        for (int j = 0; j < n_children[i]; ++j) {
            const int ch = first_child[i] + j;
            if (stored[ch] && fh[ch] > h) h = fh[ch];
        }
        fh[i] = h + 1;
    }
    for (int i = 0; i < N; ++i)
        if (n_children[i] > 0) cls[i] = stored[i] ? fh[i] : fh[parent[i]];   // (a cherry is never a root)
    // the shape of the gathering unit: describe_units' packed word of the stored node (of a cherry's parent for the tips of a cherry)
    std::vector<long long> key(N, 0);
    if (by_shape) {
        auto packed_of = [&](int n) {
            const int nc = n_children[n];
            int packed = nc < 15 ? nc : 15;
            bool cherries_ok = true, first_two = true;
            for (int j = 0; j < 4 && j < nc; ++j) {
                const int ch = first_child[n] + j;
                int code = n_children[ch] == 0 ? 0 : 1;
                if (n_children[ch] > 0 && !stored[ch]) {
                    if (n_children[ch] > 4) {
                        cherries_ok = false;
                        code = 2;
                    } else {
                        code = 1 + n_children[ch];
                    }
                }
                packed |= code << (8 + 3 * j);
                if (j >= 2 && code == 1) first_two = false;
            }
            if (cherries_ok) packed |= 1 << 4;
            if (first_two) packed |= 1 << 5;
            return packed;
        };
        for (int i = 0; i < N; ++i)
            if (n_children[i] > 0) key[i] = ((long long)packed_of(stored[i] ? i : parent[i]) << 32) | (unsigned)cls[i];
    } else {
        for (int i = 0; i < N; ++i) key[i] = cls[i];
    }
    std::vector<int> order;
    order.reserve(N);
    for (int i = 0; i < R; ++i) order.push_back(i);
    size_t lo = 0;
    bool identity = true;
    std::vector<int> par;
    for (int d = 0; d + 1 < n_td_levels; ++d) {
        const size_t hi = order.size();
        par.clear();
        for (size_t q = lo; q < hi; ++q)
            if (n_children[order[q]] > 0) par.push_back(order[q]);
        std::stable_sort(par.begin(), par.end(), [&](int x, int y) { return key[x] < key[y]; });
        for (int p : par)
            for (int j = 0; j < n_children[p]; ++j) {
                identity = identity && first_child[p] + j == (int)order.size();
                order.push_back(first_child[p] + j);
            }
        lo = hi;
    }
    (void)td_offsets;
    if (identity || (int)order.size() != N) return false;
    old_of_new.swap(order);
    new_of_old.assign(N, 0);
    for (int q = 0; q < N; ++q) new_of_old[old_of_new[q]] = q;
    return true;
}

// rows of `width` elements between the caller's numbering and the library's, for n_cols columns of N rows each
template <typename T>
static void rows_to_internal(const pml_ctx* ctx, const T* api, T* internal, size_t width, size_t n_cols) {
    const size_t N = (size_t)ctx->N;
    for (size_t c = 0; c < n_cols; ++c)
        for (size_t q = 0; q < N; ++q)
            memcpy(internal + (c * N + q) * width, api + (c * N + (size_t)ctx->old_of_new[q]) * width, width * sizeof(T));
}

template <typename T>
static void rows_to_api(const pml_ctx* ctx, const T* internal, T* api, size_t width, size_t n_cols) {
    const size_t N = (size_t)ctx->N;
    for (size_t c = 0; c < n_cols; ++c)
        for (size_t q = 0; q < N; ++q)
            memcpy(api + (c * N + (size_t)ctx->old_of_new[q]) * width, internal + (c * N + q) * width, width * sizeof(T));
}

static inline bool permuted(const pml_ctx* ctx) { return !ctx->old_of_new.empty(); }

// an output that was fetched in the library's numbering, put into the caller's in place (after the copy has been waited for)
template <typename T>
static void rows_to_api_inplace(const pml_ctx* ctx, T* buf, size_t width, size_t n_cols) {
    if (!permuted(ctx) || buf == nullptr) return;
    const size_t col = (size_t)ctx->N * width;
    std::vector<T> tmp(col);   // (a column at a time: the table is never held twice)
    for (size_t c = 0; c < n_cols; ++c) {
        std::copy(buf + c * col, buf + (c + 1) * col, tmp.begin());
        rows_to_api(ctx, tmp.data(), buf + c * col, width, 1);
    }
}
// Per-node rows of n_cols columns from the device to the caller, in the caller's numbering: `width` elements of every row of
// `src_width` (the device pads rows to ks).  A renumbered forest's rows are gathered on the device, a column at a time, into a
// staging buffer of one column and copied from there -- no second copy of the table on the host, no serial host pass
// (round 5 permuted on the host after the copy: +8.6 GB of host memory for 32 columns of 262 144 tips at k = 64).
// Asynchronous on the ctx's stream; the caller synchronises.
template <typename T>
static int fetch_rows(pml_ctx* ctx, const T* d_src, size_t src_width, size_t width, size_t n_cols, T* out) {
    static_assert(sizeof(T) % 4 == 0, "rows are moved in 4-byte words");
    if (!out || n_cols == 0) return PML_OK;
    const size_t N = (size_t)ctx->N;
    if (!permuted(ctx)) {
        if (src_width == width)
            HIP_TRY(hipMemcpyAsync(out, d_src, n_cols * N * width * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        else
            HIP_TRY(hipMemcpy2DAsync(out, width * sizeof(T), d_src, src_width * sizeof(T), width * sizeof(T), n_cols * N,
                                     hipMemcpyDeviceToHost, ctx->stream));
        return PML_OK;
    }
    const size_t col_bytes = N * width * sizeof(T);
    if (ctx->stage_bytes < col_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_stage) (void)hipFree(ctx->d_stage);
        ctx->d_stage = nullptr;
        ctx->stage_bytes = 0;
        HIP_TRY(hipMalloc(&ctx->d_stage, col_bytes));
        ctx->stage_bytes = col_bytes;
    }
    const int wd = (int)(width * sizeof(T) / 4), ws = (int)(src_width * sizeof(T) / 4);
    const long long total = (long long)N * wd;
    const int blocks = (int)std::min<long long>((total + PML_BLOCK - 1) / PML_BLOCK, 65536);
    for (size_t c = 0; c < n_cols; ++c) {
        hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(PML_BLOCK), 0, ctx->stream,
                           (const unsigned*)(d_src + c * N * src_width), (unsigned*)ctx->d_stage, ctx->d_new_of_old, (long long)N, wd,
                           ws, 0, 1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out + c * N * width, ctx->d_stage, col_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    return PML_OK;
}

static inline int api_id(const pml_ctx* ctx, int internal) { return permuted(ctx) && internal >= 0 ? ctx->old_of_new[internal] : internal; }
static inline int internal_id(const pml_ctx* ctx, int api) { return permuted(ctx) && api >= 0 ? ctx->new_of_old[api] : api; }

// ---------------------------------------------------------------------------------------------------------------------
extern "C" {

const char* pml_last_error(void) { return g_last_error.c_str(); }

int pml_version(void) { return PML_VERSION; }

const char* pml_build_digest(void) { return kBuildDigest + sizeof("PML_BUILD_DIGEST=") - 1; }

int pml_device_count(int* count) {
    if (!count) return fail(PML_ERR_INVALID, "count is NULL");
    HIP_TRY(hipGetDeviceCount(count));
    return PML_OK;
}

int pml_ctx_create(int device, pml_ctx** out) {
    if (!out) return fail(PML_ERR_INVALID, "out is NULL");
    *out = nullptr;
    // (the device list and the architecture names are asked for once per process: hipGetDeviceProperties costs ~10 ms,
    // and an analysis opens a context per group of characters)
    static std::mutex arch_lock;
    static std::vector<std::string> arch;
    {
        std::lock_guard<std::mutex> guard(arch_lock);
        if (arch.empty()) {
            int n = 0;
            HIP_TRY(hipGetDeviceCount(&n));
            std::vector<std::string> names;
            for (int d = 0; d < n; ++d) {
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, d));
                names.push_back(prop.gcnArchName);
            }
            arch.swap(names);
        }
    }
    const int n = (int)arch.size();
    if (device < 0 || device >= n) return fail(PML_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    if (strncmp(arch[device].c_str(), "gfx950", 6) != 0)
        return fail(PML_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                    arch[device].c_str());
    pml_ctx* ctx = new pml_ctx();
    ctx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e != hipSuccess) {
        delete ctx;
        return fail(PML_ERR_HIP, "stream/event creation failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return PML_OK;
}

int pml_ctx_destroy(pml_ctx* ctx) {
    if (!ctx) return PML_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)pml_comm_destroy(ctx);
    free_all(ctx);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    prof_release(ctx);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PML_OK;
}

int pml_ctx_set_option(pml_ctx* ctx, int option, int value) {
    if (!ctx) return fail(PML_ERR_INVALID, "ctx is NULL");
    if (option == PML_OPT_CHERRY_FUSION) {
        if (ctx->N != 0) return fail(PML_ERR_INVALID, "PML_OPT_CHERRY_FUSION must be set before the tree is uploaded");
        ctx->fuse = value != 0;
        return PML_OK;
    }
    if (option == PML_OPT_KEEP_TD) {
        if ((value != 0) != ctx->keep_td) {
            drop_graph(ctx->td_graph);
            drop_graph(ctx->mp_graph);
        }
        ctx->keep_td = value != 0;
        return PML_OK;
    }
    if (option == PML_OPT_EIGEN_FUSED) {
        if ((value != 0) != ctx->eig_fused_opt) {
            drop_sweep_graphs(ctx);
            ctx->prep_dirty = true;
            ctx->bu_mode = -1;
            ctx->td_valid = ctx->js_valid = false;
        }
        ctx->eig_fused_opt = value != 0;
        return PML_OK;
    }
    if (option == PML_OPT_EIGEN_JOINT_VALU) {
        if ((value != 0) != ctx->eigj_valu_opt) {
            drop_graph(ctx->bu_graph[0]);
            ctx->prep_dirty = true;
            ctx->bu_mode = -1;
            ctx->td_valid = ctx->js_valid = false;
        }
        ctx->eigj_valu_opt = value != 0;
        return PML_OK;
    }
    if (option == PML_OPT_IMPLICIT_TIP_POSTERIORS) {
        if ((value != 0) != ctx->implicit_tips) {
            drop_graph(ctx->td_graph);
            drop_graph(ctx->mp_graph);
        }
        ctx->implicit_tips = value != 0;
        return PML_OK;
    }
    return fail(PML_ERR_INVALID, "unknown option %d", option);
}

int pml_ctx_set_tunable(pml_ctx* ctx, const char* name, int64_t value, int is_set) {
    if (!ctx || !name) return fail(PML_ERR_INVALID, "ctx / name is NULL");
    if (strncmp(name, "PASTML_HIP_", 11) == 0) name += 11;
    for (int i = 0; i < T_COUNT; ++i) {
        if (strcmp(name, kTunableName[i]) != 0) continue;
        if (kTunableTree[i] && ctx->N != 0)
            return fail(PML_ERR_INVALID, "%s is read when the tree is uploaded: set it before pml_tree_upload", name);
        const bool on = is_set != 0 && (!kTunableFlag[i] || value != 0);
        ctx->tune.has[i] = on;
        ctx->tune.val[i] = on ? (long long)value : 0;
        // a captured launch sequence was made under the old setting
        drop_sweep_graphs(ctx);
        drop_graph(ctx->bt_graph);
        ctx->prep_dirty = true;
        ctx->bu_mode = -1;
        ctx->td_valid = ctx->js_valid = false;
        return PML_OK;
    }
    return fail(PML_ERR_INVALID, "unknown tunable %s", name);
}

int pml_ctx_sync(pml_ctx* ctx) {
    if (!ctx) return fail(PML_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PML_OK;
}

int pml_ctx_memory(pml_ctx* ctx, uint64_t* held, uint64_t* device_free) {
    if (!ctx) return fail(PML_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    size_t f = 0, tot = 0;
    HIP_TRY(hipMemGetInfo(&f, &tot));
    if (held) *held = ctx->held;
    if (device_free) *device_free = f;
    return PML_OK;
}

static bool super_sweeps(const pml_ctx* ctx);

int pml_schedule_info(pml_ctx* ctx, int32_t* level_schedule, int32_t* n_two_level, int32_t* n_stacked) {
    if (!ctx || ctx->C == 0) return fail(PML_ERR_INVALID, "allocate the columns first");
    // (the model kind is set with the first model; until then the F81 family is assumed)
    const int kind = ctx->kind;
    if (kind < 0) ctx->kind = PML_MODEL_F81;
    const bool on = super_sweeps(ctx);
    ctx->kind = kind;
    if (level_schedule) *level_schedule = on ? 1 : 0;
    if (n_two_level) *n_two_level = on ? ctx->sup.n : 0;
    if (n_stacked) *n_stacked = on ? ctx->sup.n_stack : 0;
    return PML_OK;
}

int pml_tree_order(pml_ctx* ctx, int32_t* new_of_old) {
    if (!ctx || ctx->N == 0) return fail(PML_ERR_INVALID, "upload the tree first");
    if (!new_of_old) return fail(PML_ERR_INVALID, "new_of_old is NULL");
    for (int i = 0; i < ctx->N; ++i) new_of_old[i] = permuted(ctx) ? ctx->new_of_old[i] : i;
    return PML_OK;
}

int pml_sweep_schedule(pml_ctx* ctx, int32_t* kind, int32_t* n_blocks, int32_t* n_absorbed) {
    if (!ctx || ctx->C == 0) return fail(PML_ERR_INVALID, "allocate the columns first");
    const int model = ctx->kind;
    if (model < 0) ctx->kind = PML_MODEL_F81;  // (until the first model is set the F81 family is assumed)
    int k = PML_SCHEDULE_LEVELS;
    if (ctx->kind != PML_MODEL_F81) k = PML_SCHEDULE_OTHER_MODEL;
    else if (single_launch_sweeps(ctx)) k = PML_SCHEDULE_SINGLE_LAUNCH;
    else if (block_schedule(ctx)) k = PML_SCHEDULE_BLOCKS;
    else if (super_sweeps(ctx)) k = PML_SCHEDULE_TWO_LEVEL;
    ctx->kind = model;
    if (kind) *kind = k;
    if (n_blocks) *n_blocks = k == PML_SCHEDULE_BLOCKS ? ctx->blocks.n_blocks : 0;
    if (n_absorbed) *n_absorbed = 0;  // (the general two-level units of round 4 are gone: include/pastml_hip.h)
    return PML_OK;
}

// unit descriptors (PmlUnit, pml_kernels_f81.h) of a node list
static int describe_units(const int* first_child, const int* n_children, const unsigned char* kind, const int* list, int count,
                          bool use_kind, std::vector<PmlUnit>& out) {
    out.resize(count > 0 ? count : 1);
    for (int q = 0; q < count; ++q) {
        const int n = list[q];
        PmlUnit u;
        u.n = n;
        u.fc = first_child[n];
        u.pad = 0;
        const int nc = n_children[n];
        int packed = nc < 15 ? nc : 15;
        bool cherries_ok = true, stored_first_two_only = true;
        for (int j = 0; j < 4; ++j) {
            u.cfc[j] = 0;
            if (j >= nc) continue;
            const int ch = u.fc + j;
            u.cfc[j] = first_child[ch];
            const int kd = use_kind ? (int)kind[ch] : (n_children[ch] == 0 ? PML_KIND_TIP : PML_KIND_STORED);
            int code = kd == PML_KIND_TIP ? 0 : 1;
            if (kd == PML_KIND_CHERRY) {
                if (n_children[ch] > 4) {
                    cherries_ok = false;
                    code = 2;
                } else {
                    code = 1 + n_children[ch];
                }
            }
            packed |= code << (8 + 3 * j);
            if (j >= 2 && code == 1) stored_first_two_only = false;
        }
        if (cherries_ok) packed |= 1 << 4;
        if (stored_first_two_only) packed |= 1 << 5;
        u.packed = packed;
        out[q] = u;
    }
    return 0;
}

// the units of every level (offs) sorted by shape, stable (see pml_tree_upload)
static std::vector<PmlUnit> units_by_shape(const std::vector<PmlUnit>& in, const std::vector<int>& offs, size_t count) {
    auto shape_less = [](const PmlUnit& x, const PmlUnit& y) { return x.packed < y.packed; };
    std::vector<PmlUnit> out(in);
    for (size_t l = 0; l + 1 < offs.size(); ++l) {
        const size_t a = (size_t)offs[l], b = std::min((size_t)offs[l + 1], count);
        if (b > a + 1) std::stable_sort(out.begin() + a, out.begin() + b, shape_less);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
int pml_tree_upload(pml_ctx* ctx, int32_t n_nodes, int32_t n_roots, const int32_t* parent, const int32_t* first_child,
                    const int32_t* n_children, const double* dist, int32_t n_bu_levels, const int32_t* bu_offsets,
                    const int32_t* bu_order, int32_t n_td_levels, const int32_t* td_offsets,
                    const int32_t* td_parent_offsets, const int32_t* td_parents, const int32_t* post_rank) {
    if (!ctx) return fail(PML_ERR_INVALID, "ctx is NULL");
    if (n_nodes <= 0 || n_roots <= 0 || n_roots > n_nodes) return fail(PML_ERR_INVALID, "bad node/root counts");
    if (!parent || !first_child || !n_children || !dist || !bu_offsets || !td_offsets || !td_parent_offsets ||
        !post_rank)
        return fail(PML_ERR_INVALID, "NULL tree array");
    if (n_bu_levels < 0 || n_td_levels < 1) return fail(PML_ERR_INVALID, "bad level counts");
    // host-side validation of everything the kernels index with (a bad index would fault the GPU)
    int n_internal = 0;
    for (int i = 0; i < n_nodes; ++i) {
        const int nc = n_children[i];
        if (nc < 0) return fail(PML_ERR_INVALID, "n_children[%d] < 0", i);
        if (nc > 0) {
            ++n_internal;
            const long long fc = first_child[i];
            if (fc <= i || fc + nc > n_nodes) return fail(PML_ERR_INVALID, "children of node %d out of range", i);
            for (int j = 0; j < nc; ++j)
                if (parent[fc + j] != i) return fail(PML_ERR_INVALID, "parent/child arrays disagree at node %d", i);
        }
        if (i < n_roots ? parent[i] != -1 : (parent[i] < 0 || parent[i] >= i))
            return fail(PML_ERR_INVALID, "parent[%d] = %d is not valid for level-ordered ids", i, parent[i]);
        if (!(dist[i] >= 0.0)) return fail(PML_ERR_INVALID, "dist[%d] is negative or NaN", i);
        if (post_rank[i] < 0 || post_rank[i] >= n_nodes) return fail(PML_ERR_INVALID, "post_rank[%d] out of range", i);
    }
    if (bu_offsets[0] != 0 || bu_offsets[n_bu_levels] != n_internal)
        return fail(PML_ERR_INVALID, "bu_offsets must cover the %d internal nodes", n_internal);
    if (td_parent_offsets[0] != 0 || td_parent_offsets[n_td_levels] != n_internal)
        return fail(PML_ERR_INVALID, "td_parent_offsets must cover the %d internal nodes", n_internal);
    if (td_offsets[0] != 0 || td_offsets[1] != n_roots || td_offsets[n_td_levels] != n_nodes)
        return fail(PML_ERR_INVALID, "td_offsets must start with the roots and cover all nodes");
    if (n_internal > 0 && (!bu_order || !td_parents)) return fail(PML_ERR_INVALID, "NULL level array");
    {
        std::vector<char> seen(n_nodes, 0);
        std::vector<int> height(n_nodes, 0);
        for (int l = 0; l < n_bu_levels; ++l) {
            if (bu_offsets[l + 1] < bu_offsets[l]) return fail(PML_ERR_INVALID, "bu_offsets not monotone");
            for (int q = bu_offsets[l]; q < bu_offsets[l + 1]; ++q) {
                const int n = bu_order[q];
                if (n < 0 || n >= n_nodes || n_children[n] == 0 || seen[n])
                    return fail(PML_ERR_INVALID, "bu_order[%d] = %d is not a distinct internal node", q, n);
                seen[n] = 1;
                // every internal child must sit in an earlier level
                for (int j = 0; j < n_children[n]; ++j) {
                    const int ch = first_child[n] + j;
                    if (n_children[ch] > 0 && (!seen[ch] || height[ch] >= l + 1))
                        return fail(PML_ERR_INVALID, "bu level %d: node %d precedes its child %d", l, n, ch);
                }
                height[n] = l + 1;
            }
        }
        std::fill(seen.begin(), seen.end(), 0);
        for (int l = 0; l < n_td_levels; ++l) {
            if (td_parent_offsets[l + 1] < td_parent_offsets[l] || td_offsets[l + 1] < td_offsets[l])
                return fail(PML_ERR_INVALID, "td offsets not monotone");
            for (int q = td_parent_offsets[l]; q < td_parent_offsets[l + 1]; ++q) {
                const int n = td_parents[q];
                if (n < td_offsets[l] || n >= td_offsets[l + 1] || n_children[n] == 0 || seen[n])
                    return fail(PML_ERR_INVALID, "td_parents[%d] = %d is not a distinct internal node of depth %d", q,
                                n, l);
                seen[n] = 1;
            }
        }
    }

    // ---- the library's own numbering (height_order): from here on every array is in it
    std::vector<int> perm_old_of_new, perm_new_of_old;
    std::vector<int32_t> p_parent, p_first_child, p_n_children, p_bu_order, p_td_parents, p_post_rank;
    std::vector<double> p_dist;
    // Shape-aware order (round 6) for large forests without many polytomies: there the level launches walk shape-sorted lists
    // and a depth's sibling groups follow them.  Measured, marginal pass, class-only -> shape-aware numbering, bits unchanged
    // (profiles/r06q_shape_order.txt): random binary 262 144 tips x 32, k = 64 5.07 -> 4.94 ms, k = 12 2.29 -> 2.07, k = 8 1.91 ->
    // 1.68, k = 4 1.46 -> 1.35 (the last two with their lists sorted by shape as well, which the old numbering punished);
    // forests with polytomies lose 2 - 3 % (many shapes: short runs) and 40 000-tip trees 3 %: they keep the class-only order.
    bool shape_order = false;
    {
        long long n_inner = 0, n34 = 0;   // (the polytomy rule of pml_chars_alloc)
        for (int i = 0; i < n_nodes; ++i) {
            const int nc = n_children[i];
            bool inner = false;
            for (int j = 0; j < nc && !inner; ++j) inner = n_children[first_child[i] + j] > 0;
            if (!inner) continue;
            ++n_inner;
            n34 += nc == 3 || nc == 4;
        }
        const bool polytomies = n_inner > 0 && n34 * 100 >= 15 * n_inner;
        shape_order = n_nodes >= 150000 && !polytomies;
        if (ctx->tune.on(T_SHAPE_ORDER)) shape_order = ctx->tune.get(T_SHAPE_ORDER, 1) != 0;
    }
    if (!ctx->tune.on(T_NO_HEIGHT_ORDER) &&
        height_order(n_nodes, n_roots, parent, first_child, n_children, td_offsets, n_td_levels, ctx->fuse, shape_order,
                     perm_old_of_new, perm_new_of_old)) {
        const std::vector<int>& o = perm_old_of_new;
        const std::vector<int>& nw = perm_new_of_old;
        p_parent.resize(n_nodes);
        p_first_child.resize(n_nodes);
        p_n_children.resize(n_nodes);
        p_post_rank.resize(n_nodes);
        p_dist.resize(n_nodes);
        for (int q = 0; q < n_nodes; ++q) {
            const int old = o[q];
            p_parent[q] = parent[old] >= 0 ? nw[parent[old]] : -1;
            p_n_children[q] = n_children[old];
            // (a tip's entry is never read; it only has to pass for an id)
            p_first_child[q] = n_children[old] > 0 ? nw[first_child[old]] : 0;
            p_post_rank[q] = post_rank[old];
            p_dist[q] = dist[old];
        }
        p_bu_order.assign(n_internal > 0 ? n_internal : 1, 0);
        p_td_parents.assign(n_internal > 0 ? n_internal : 1, 0);
        for (int l = 0; l < n_bu_levels; ++l) {
            for (int q = bu_offsets[l]; q < bu_offsets[l + 1]; ++q) p_bu_order[q] = nw[bu_order[q]];
            std::sort(p_bu_order.begin() + bu_offsets[l], p_bu_order.begin() + bu_offsets[l + 1]);
        }
        for (int l = 0; l < n_td_levels; ++l) {
            for (int q = td_parent_offsets[l]; q < td_parent_offsets[l + 1]; ++q) p_td_parents[q] = nw[td_parents[q]];
            std::sort(p_td_parents.begin() + td_parent_offsets[l], p_td_parents.begin() + td_parent_offsets[l + 1]);
        }
        parent = p_parent.data();
        first_child = p_first_child.data();
        n_children = p_n_children.data();
        post_rank = p_post_rank.data();
        dist = p_dist.data();
        bu_order = p_bu_order.data();
        td_parents = p_td_parents.data();
    }

    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // a new tree resets everything the ctx holds
    free_all(ctx);
    hipStream_t stream = ctx->stream;
    hipEvent_t e0 = ctx->ev0, e1 = ctx->ev1;
    (void)prof_drain(ctx);
    std::vector<hipEvent_t> prof_pool;
    prof_pool.swap(ctx->prof_pool);
    if (ctx->prof_open) prof_pool.push_back(ctx->prof_open);

    int device = ctx->device;
    const bool profile = ctx->profile;
    const bool fuse = ctx->fuse, keep_td = ctx->keep_td, eig_fused_opt = ctx->eig_fused_opt, eigj_valu_opt = ctx->eigj_valu_opt,
               implicit_tips = ctx->implicit_tips;
    PmlComm* comm = ctx->comm;
    const PmlTune tune = ctx->tune;
    *ctx = pml_ctx();
    ctx->tune = tune;
    ctx->fuse = fuse;
    ctx->keep_td = keep_td;
    ctx->eig_fused_opt = eig_fused_opt;
    ctx->eigj_valu_opt = eigj_valu_opt;
    ctx->implicit_tips = implicit_tips;
    ctx->comm = comm;
    ctx->stream = stream;
    ctx->ev0 = e0;
    ctx->ev1 = e1;
    ctx->prof_pool.swap(prof_pool);
    ctx->profile = profile;
    ctx->device = device;
    ctx->old_of_new.swap(perm_old_of_new);
    ctx->new_of_old.swap(perm_new_of_old);
    ctx->shape_ordered = shape_order && !ctx->old_of_new.empty();

    ctx->N = n_nodes;
    ctx->n_roots = n_roots;
    ctx->n_bu_levels = n_bu_levels;
    ctx->n_td_levels = n_td_levels;
    ctx->bu_offsets.assign(bu_offsets, bu_offsets + n_bu_levels + 1);
    ctx->td_offsets.assign(td_offsets, td_offsets + n_td_levels + 1);
    ctx->td_parent_offsets.assign(td_parent_offsets, td_parent_offsets + n_td_levels + 1);
    ctx->h_parent.assign(parent, parent + n_nodes);
    ctx->h_n_children.assign(n_children, n_children + n_nodes);
    PML_TRY(dev_alloc(ctx, &ctx->d_parent, n_nodes));
    PML_TRY(dev_alloc(ctx, &ctx->d_first_child, n_nodes));
    PML_TRY(dev_alloc(ctx, &ctx->d_n_children, n_nodes));
    PML_TRY(dev_alloc(ctx, &ctx->d_post_rank, n_nodes));
    PML_TRY(dev_alloc(ctx, &ctx->d_dist, n_nodes));
    PML_TRY(dev_alloc(ctx, &ctx->d_bu_order, n_internal));
    PML_TRY(dev_alloc(ctx, &ctx->d_td_parents, n_internal));
    PML_TRY(upload(ctx, ctx->d_parent, parent, n_nodes));
    if (permuted(ctx)) {
        PML_TRY(dev_alloc(ctx, &ctx->d_new_of_old, n_nodes));
        PML_TRY(dev_alloc(ctx, &ctx->d_old_of_new, n_nodes));
        PML_TRY(upload(ctx, ctx->d_new_of_old, ctx->new_of_old.data(), n_nodes));
        PML_TRY(upload(ctx, ctx->d_old_of_new, ctx->old_of_new.data(), n_nodes));
    }
    PML_TRY(upload(ctx, ctx->d_first_child, first_child, n_nodes));
    PML_TRY(upload(ctx, ctx->d_n_children, n_children, n_nodes));
    PML_TRY(upload(ctx, ctx->d_post_rank, post_rank, n_nodes));
    PML_TRY(upload(ctx, ctx->d_dist, dist, n_nodes));
    if (n_internal) {
        PML_TRY(upload(ctx, ctx->d_bu_order, bu_order, n_internal));
        PML_TRY(upload(ctx, ctx->d_td_parents, td_parents, n_internal));
    }
    PML_TRY(dev_alloc(ctx, &ctx->d_bu_offsets, n_bu_levels + 1));
    PML_TRY(dev_alloc(ctx, &ctx->d_td_offsets, n_td_levels + 1));
    PML_TRY(upload(ctx, ctx->d_bu_offsets, bu_offsets, n_bu_levels + 1));
    PML_TRY(upload(ctx, ctx->d_td_offsets, td_offsets, n_td_levels + 1));
    {
        std::vector<int> tips;
        tips.reserve(n_nodes - n_internal);
        for (int i = 0; i < n_nodes; ++i)
            if (n_children[i] == 0) tips.push_back(i);
        ctx->n_tips = (int)tips.size();
        PML_TRY(dev_alloc(ctx, &ctx->d_tips, tips.size()));
        if (!tips.empty()) PML_TRY(upload(ctx, ctx->d_tips, tips.data(), tips.size()));
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // the vector goes out of scope
    }
    // ---- cherry fusion tables: kind per node, level lists over the stored internal nodes
    {
        std::vector<unsigned char>& kind = ctx->h_kind;
        kind.assign(n_nodes, PML_KIND_TIP);
        std::vector<int> cherries, fh(n_nodes, 0);
        for (int i = 0; i < n_nodes; ++i) {
            if (n_children[i] == 0) continue;
            bool all_tips = true;
            for (int j = 0; j < n_children[i]; ++j) all_tips &= n_children[first_child[i] + j] == 0;
            if (ctx->fuse && all_tips && parent[i] >= 0) {
                kind[i] = PML_KIND_CHERRY;
                cherries.push_back(i);
            } else {
                kind[i] = PML_KIND_STORED;
            }
        }
        int max_h = 0;
        for (int i = n_nodes - 1; i >= 0; --i) {  // children have larger ids than their parent
            if (kind[i] != PML_KIND_STORED) continue;
            int h = 0;
            for (int j = 0; j < n_children[i]; ++j) {
                const int ch = first_child[i] + j;
                if (kind[ch] == PML_KIND_STORED && fh[ch] > h) h = fh[ch];
            }
            fh[i] = h + 1;
            if (fh[i] > max_h) max_h = fh[i];
        }
        std::vector<int>& off = ctx->bu_offsets_f;
        off.assign(max_h + 1, 0);
        for (int i = 0; i < n_nodes; ++i)
            if (kind[i] == PML_KIND_STORED) ++off[fh[i]];
        // off[h] = count of height h (h >= 1) -> exclusive prefix: level l (0-based) = height l + 1
        {
            int run = 0;
            for (int h = 1; h <= max_h; ++h) {
                const int cnt = off[h];
                off[h - 1] = run;
                run += cnt;
            }
            off[max_h] = run;
        }
        const int n_stored = off[max_h];
        std::vector<int> order(n_stored > 0 ? n_stored : 1), cursor(off.begin(), off.end());
        for (int i = 0; i < n_nodes; ++i)
            if (kind[i] == PML_KIND_STORED) order[cursor[fh[i] - 1]++] = i;
        std::vector<int> tdp;
        ctx->td_parent_offsets_f.assign(n_td_levels + 1, 0);
        for (int l = 0; l < n_td_levels; ++l) {
            for (int q = td_parent_offsets[l]; q < td_parent_offsets[l + 1]; ++q)
                if (kind[td_parents[q]] == PML_KIND_STORED) tdp.push_back(td_parents[q]);
            ctx->td_parent_offsets_f[l + 1] = (int)tdp.size();
        }
        ctx->n_cherries = (int)cherries.size();
        {
            // unit descriptors (PmlUnit, pml_kernels_f81.h) for the three node lists the F81 kernels walk
            auto describe = [&](const int* list, int count, bool use_kind, std::vector<PmlUnit>& out) {
                return describe_units(first_child, n_children, kind.data(), list, count, use_kind, out);
            };
            std::vector<PmlUnit> ub_f, ut_f, ub, uc;
            describe(cherries.data(), (int)cherries.size(), false, uc);
            PML_TRY(dev_alloc(ctx, &ctx->d_cherry_units, uc.size()));
            PML_TRY(upload(ctx, ctx->d_cherry_units, uc.data(), uc.size()));
            describe(order.data(), n_stored, true, ub_f);
            ctx->bu_level_vec_f.assign(max_h > 0 ? max_h : 1, 0);
            for (int l = 0; l < max_h; ++l)
                for (int q = off[l]; q < off[l + 1] && !ctx->bu_level_vec_f[l]; ++q) {
                    const int pk = ub_f[q].packed;
                    if (((pk >> 8) & 7) == 1 || ((pk >> 11) & 7) == 1) ctx->bu_level_vec_f[l] = 1;
                }
            describe(tdp.data(), n_stored, true, ut_f);
            ctx->td_cherry_prefix.assign(n_stored + 1, 0);
            for (int q = 0; q < n_stored; ++q) {
                const int pk = ut_f[q].packed;
                ctx->td_cherry_prefix[q + 1] = ctx->td_cherry_prefix[q] + ((((pk >> 8) & 7) >= 2 || ((pk >> 11) & 7) >= 2) ? 1 : 0);
            }
            describe(bu_order, n_internal, false, ub);
            {
                pml_ctx::EigenTiers& E = ctx->eig_tiers;
                E = pml_ctx::EigenTiers();
                const int thin = (int)ctx->tune.get(T_EIGJ_TIER_THIN, 4096);
                const int depth = std::max(2, (int)ctx->tune.get(T_EIGJ_TIER_DEPTH, 4));
                const int top_nodes = 48;
                int L0 = n_bu_levels;
                while (L0 > 0 && bu_offsets[L0] - bu_offsets[L0 - 1] <= thin) --L0;
                if (!ctx->tune.on(T_NO_EIGJ_TIERS) && n_bu_levels - L0 >= 6) {
                    std::vector<int> level_of(n_nodes, -1);
                    for (int l = 0; l < n_bu_levels; ++l)
                        for (int q = bu_offsets[l]; q < bu_offsets[l + 1]; ++q) level_of[bu_order[q]] = l;
                    std::vector<PmlUnit> tu;
                    std::vector<int> lv, start, tnodes;
                    int a = L0;
                    std::vector<int> block_of(n_nodes, -1);
                    // The depth of a tier is the largest (up to 12 levels) whose blocks still have at most 12 nodes per
                    // level -- one pass of a workgroup per level step at k = 20 (three nodes per wavefront).  A
                    // balanced binary tree gets tiers of four levels, ragged trees deeper ones (measured: HIV1C-shaped
                    // and random 40 000-tip trees are 10 - 20 % faster with 6 - 8 levels than with 4, cfg3 slower).
                    // PASTML_HIP_EIGJ_TIER_DEPTH fixes the depth.
                    const bool fixed_depth = ctx->tune.on(T_EIGJ_TIER_DEPTH);
                    while (a + 2 <= n_bu_levels && bu_offsets[a + 1] - bu_offsets[a] > top_nodes) {
                        int use = 0, nb = 0;
                        std::vector<std::vector<int>> cell;
                        for (int dep = fixed_depth ? depth : 12; dep >= 2; --dep) {
                            if (a + dep > n_bu_levels) {
                                if (fixed_depth) break;
                                continue;
                            }
                            const int b = a + dep;
                            // block of a node: its highest ancestor below level b (higher levels come last in the list,
                            // so walking it backwards meets parents before children)
                            nb = 0;
                            for (int q = bu_offsets[b]; q-- > bu_offsets[a];) {
                                const int n = bu_order[q];
                                const int p = parent[n];
                                block_of[n] = (p >= 0 && level_of[p] >= 0 && level_of[p] < b) ? block_of[p] : nb++;
                            }
                            cell.assign((size_t)nb * dep, std::vector<int>());
                            size_t widest_cell = 0;
                            for (int q = bu_offsets[a]; q < bu_offsets[b]; ++q) {
                                const int n = bu_order[q];
                                std::vector<int>& cl = cell[(size_t)block_of[n] * dep + (level_of[n] - a)];
                                cl.push_back(n);
                                widest_cell = std::max(widest_cell, cl.size());
                            }
                            if (widest_cell <= 12 || dep == 2 || fixed_depth) {
                                use = dep;
                                break;
                            }
                        }
                        if (use == 0) break;
                        const int b = a + use;
                        pml_ctx::EigenTiers::Tier T;
                        T.first_block = (int)start.size();
                        T.n_blocks = nb;
                        T.depth = use;
                        std::vector<int> flat_nodes;
                        const int base = (int)tu.size();
                        for (int bl = 0; bl < nb; ++bl) {
                            start.push_back((int)lv.size());
                            for (int d = 0; d < use; ++d) {
                                lv.push_back(base + (int)flat_nodes.size());
                                for (int n : cell[(size_t)bl * use + d]) flat_nodes.push_back(n);
                            }
                            lv.push_back(base + (int)flat_nodes.size());
                        }
                        std::vector<PmlUnit> part;
                        describe(flat_nodes.data(), (int)flat_nodes.size(), false, part);
                        part.resize(flat_nodes.size());
                        tu.insert(tu.end(), part.begin(), part.end());
                        tnodes.insert(tnodes.end(), flat_nodes.begin(), flat_nodes.end());
                        E.tiers.push_back(T);
                        for (int l = a; l < b; ++l) E.widest = std::max(E.widest, bu_offsets[l + 1] - bu_offsets[l]);
                        a = b;
                    }
                    if (!E.tiers.empty()) {
                        tu.push_back(ub[0]);  // (slack: an empty level at the end of the table is still addressed)
                        tnodes.push_back(ub[0].n);
                        PML_TRY(dev_alloc(ctx, &E.d_nodes, tnodes.size()));
                        PML_TRY(upload(ctx, E.d_nodes, tnodes.data(), tnodes.size()));
                        PML_TRY(dev_alloc(ctx, &E.d_units, tu.size()));
                        PML_TRY(dev_alloc(ctx, &E.d_lv, lv.size()));
                        PML_TRY(dev_alloc(ctx, &E.d_start, start.size()));
                        PML_TRY(upload(ctx, E.d_units, tu.data(), tu.size()));
                        PML_TRY(upload(ctx, E.d_lv, lv.data(), lv.size()));
                        PML_TRY(upload(ctx, E.d_start, start.data(), start.size()));
                        HIP_TRY(hipStreamSynchronize(ctx->stream));
                        E.first_level = L0;
                        E.top_level = a;
                        E.ok = true;
                    }
                }
            }
            ctx->bu_level_vec.assign(n_bu_levels > 0 ? n_bu_levels : 1, 0);
            for (int l = 0; l < n_bu_levels; ++l)
                for (int q = bu_offsets[l]; q < bu_offsets[l + 1] && !ctx->bu_level_vec[l]; ++q) {
                    const int pk = ub[q].packed;
                    if (((pk >> 8) & 7) == 1 || ((pk >> 11) & 7) == 1) ctx->bu_level_vec[l] = 1;
                }
            {
                // joint back-trace tiers: from the first depth of more than 1 024 nodes on, tiers of up to 10 depths whose
                // subtrees keep at most 256 nodes per depth (one pass of a workgroup per step)
                pml_ctx::BacktraceTiers& B = ctx->bt_tiers;
                B = pml_ctx::BacktraceTiers();
                int d1 = 1;
                while (d1 < n_td_levels && td_offsets[d1 + 1] - td_offsets[d1] <= 1024) ++d1;
                if (!ctx->tune.on(T_NO_BT_TIERS) && n_td_levels - d1 >= 2) {
                    std::vector<int> depth_of(n_nodes, 0), anc(n_nodes, 0), tn, lv, start;
                    for (int l = 0; l < n_td_levels; ++l)
                        for (int i = td_offsets[l]; i < td_offsets[l + 1]; ++i) depth_of[i] = l;
                    int da = d1;
                    while (da < n_td_levels) {
                        // one counting pass over up to 10 depths: nodes per (subtree, depth), the widest cell of every
                        // depth; the tier takes the depths before the first one that is too wide
                        const int dmax = std::min(10, n_td_levels - da);
                        const int nb = td_offsets[da + 1] - td_offsets[da];
                        std::vector<int> cnt10((size_t)nb * dmax, 0), widest(dmax, 0);
                        for (int i = td_offsets[da]; i < td_offsets[da + dmax]; ++i) {
                            const int dd = depth_of[i] - da;
                            anc[i] = dd == 0 ? i - td_offsets[da] : anc[parent[i]];
                            widest[dd] = std::max(widest[dd], ++cnt10[(size_t)anc[i] * dmax + dd]);
                        }
                        int use = 1;
                        while (use < dmax && widest[use] <= 256) ++use;
                        std::vector<int> cnt((size_t)nb * use);
                        for (int b = 0; b < nb; ++b)
                            for (int d = 0; d < use; ++d) cnt[(size_t)b * use + d] = cnt10[(size_t)b * dmax + d];
                        // tables of the tier: per subtree its depth offsets into the node list
                        pml_ctx::BacktraceTiers::Tier T;
                        T.first_block = (int)start.size();
                        T.n_blocks = nb;
                        T.depth = use;
                        const int base = (int)tn.size();
                        std::vector<int> cell_start((size_t)nb * use + 1, 0);
                        for (size_t q = 0; q < (size_t)nb * use; ++q) cell_start[q + 1] = cell_start[q] + cnt[q];
                        tn.resize(base + cell_start.back());
                        std::vector<int> cursor(cell_start.begin(), cell_start.end() - 1);
                        for (int i = td_offsets[da]; i < td_offsets[da + use]; ++i)
                            tn[base + cursor[(size_t)anc[i] * use + (depth_of[i] - da)]++] = i;
                        for (int b = 0; b < nb; ++b) {
                            start.push_back((int)lv.size());
                            for (int d = 0; d <= use; ++d) lv.push_back(base + cell_start[(size_t)b * use + d]);
                        }
                        B.tiers.push_back(T);
                        da += use;
                    }
                    if (!B.tiers.empty()) {
                        tn.push_back(0);
                        PML_TRY(dev_alloc(ctx, &B.d_nodes, tn.size()));
                        PML_TRY(dev_alloc(ctx, &B.d_lv, lv.size()));
                        PML_TRY(dev_alloc(ctx, &B.d_start, start.size()));
                        PML_TRY(upload(ctx, B.d_nodes, tn.data(), tn.size()));
                        PML_TRY(upload(ctx, B.d_lv, lv.data(), lv.size()));
                        PML_TRY(upload(ctx, B.d_start, start.data(), start.size()));
                        HIP_TRY(hipStreamSynchronize(ctx->stream));
                        B.first_depth = d1;
                        B.ok = true;
                    }
                }
            }
            // Units of one shape next to each other.  Within a level the order of the units is free, and a wavefront runs
            // the union of its units' control flow: on a balanced tree every unit of a level has the same kinds of children
            // (tip / stored node / cherry of m tips), on a ragged one a wave of 8 units met most combinations and ran them
            // one after the other.  For the level launches of wide units (8 states per lane: 32 < k <= 64) every level's
            // units are sorted by the descriptor's shape word -- stable, ids ascend inside a shape, neighbours still read
            // neighbouring memory.  262 144-tip random binary tree x 32 characters, k = 64: marginal pass 6.5 -> 5.7 ms;
            // 100 000 tips with polytomies: 2.86 -> 2.13 ms; narrow units (k = 4: 64 units per wave, every lane its own
            // rows) lose 20 % to the scattered rows and keep id order.  PASTML_HIP_NO_SHAPE_SORT: id order everywhere.
            auto shape_less = [](const PmlUnit& x, const PmlUnit& y) { return x.packed < y.packed; };
            auto in_shape_order = [&](const std::vector<PmlUnit>& in, const std::vector<int>& offs, size_t count) {
                for (size_t l = 0; l + 1 < offs.size(); ++l) {
                    const size_t a = (size_t)offs[l], b = std::min((size_t)offs[l + 1], count);
                    if (b > a + 1 && !std::is_sorted(in.begin() + a, in.begin() + b, shape_less)) return false;
                }
                return true;
            };
            auto by_shape = [&](const std::vector<PmlUnit>& in, const std::vector<int>& offs, size_t count) {
                std::vector<PmlUnit> out(in);
                for (size_t l = 0; l + 1 < offs.size(); ++l) {
                    const size_t a = (size_t)offs[l], b = std::min((size_t)offs[l + 1], count);
                    if (b > a + 1) std::stable_sort(out.begin() + a, out.begin() + b, shape_less);
                }
                return out;
            };
            // top-down lists: by (shape, height class) -- the order in which height_order lays the children of a depth's units out
            // (bottom-up lists are per class already)
            auto by_shape_class = [&](const std::vector<PmlUnit>& in, const std::vector<int>& offs, size_t count) {
                if (!ctx->shape_ordered) return by_shape(in, offs, count);
                std::vector<PmlUnit> out(in);
                auto less = [&](const PmlUnit& x, const PmlUnit& y) {
                    return x.packed != y.packed ? x.packed < y.packed : fh[x.n] < fh[y.n];
                };
                for (size_t l = 0; l + 1 < offs.size(); ++l) {
                    const size_t a = (size_t)offs[l], b = std::min((size_t)offs[l + 1], count);
                    if (b > a + 1) std::stable_sort(out.begin() + a, out.begin() + b, less);
                }
                return out;
            };
            // (a balanced tree is in shape order as it is: no second copy, the launches walk the id-ordered lists)
            const bool shape_sort = !ctx->tune.on(T_NO_SHAPE_SORT) && n_stored > 0 &&
                                    !(in_shape_order(ub_f, off, (size_t)n_stored) &&
                                      in_shape_order(ut_f, ctx->td_parent_offsets_f, (size_t)n_stored));
            if (shape_sort) {
                const std::vector<PmlUnit> sb = by_shape(ub_f, off, (size_t)n_stored);
                const std::vector<PmlUnit> st_ = by_shape_class(ut_f, ctx->td_parent_offsets_f, (size_t)n_stored);
                PML_TRY(dev_alloc(ctx, &ctx->d_bu_units_fs, sb.size()));
                PML_TRY(dev_alloc(ctx, &ctx->d_td_units_fs, st_.size()));
                PML_TRY(upload(ctx, ctx->d_bu_units_fs, sb.data(), sb.size()));
                PML_TRY(upload(ctx, ctx->d_td_units_fs, st_.data(), st_.size()));
                HIP_TRY(hipStreamSynchronize(ctx->stream));  // the vectors go out of scope
            }
            PML_TRY(dev_alloc(ctx, &ctx->d_bu_units_f, ub_f.size()));
            PML_TRY(dev_alloc(ctx, &ctx->d_td_units_f, ut_f.size()));
            PML_TRY(dev_alloc(ctx, &ctx->d_bu_units, ub.size()));
            PML_TRY(upload(ctx, ctx->d_bu_units_f, ub_f.data(), ub_f.size()));
            PML_TRY(upload(ctx, ctx->d_td_units_f, ut_f.data(), ut_f.size()));
            PML_TRY(upload(ctx, ctx->d_bu_units, ub.data(), ub.size()));
            HIP_TRY(hipStreamSynchronize(ctx->stream));  // the vectors go out of scope
        // ---- two-level units: stored nodes with two stored children that are each the parent of two cherries of two
        // tips (ids of the four cherries and of the eight tips consecutive, as breadth-first numbering makes them)
        {
            pml_ctx::SuperSchedule& U = ctx->sup;
            U = pml_ctx::SuperSchedule();
            std::vector<char> pair(n_nodes, 0), sup(n_nodes, 0), gone(n_nodes, 0);
            std::vector<int> sup_list;
            if (ctx->fuse && !ctx->tune.on(T_NO_SUPER)) {
                auto two = [&](int i) { return kind[i] == PML_KIND_STORED && n_children[i] == 2; };
                for (int i = 0; i < n_nodes; ++i) {
                    if (!two(i)) continue;
                    const int a = first_child[i], b = a + 1;
                    pair[i] = kind[a] == PML_KIND_CHERRY && kind[b] == PML_KIND_CHERRY && n_children[a] == 2 &&
                              n_children[b] == 2 && first_child[b] == first_child[a] + 2;
                }
                for (int i = 0; i < n_nodes; ++i) {
                    if (!two(i)) continue;
                    const int a = first_child[i], b = a + 1;
                    if (pair[a] && pair[b] && first_child[b] == first_child[a] + 2 &&
                        first_child[first_child[b]] == first_child[first_child[a]] + 4) {
                        sup[i] = 1;
                        gone[i] = gone[a] = gone[b] = 1;
                        sup_list.push_back(i);
                    }
                }
            }
            const bool env_min = ctx->tune.on(T_SUPER_MIN);
            const int min_units = (int)ctx->tune.get(T_SUPER_MIN, 64);
            // (a launch of its own per sweep: only where it carries a share of the work)
            // (PASTML_HIP_SUPER_MIN given: whatever their share, for tests on ragged forests)
            const bool use_sup = !sup_list.empty() && (int)sup_list.size() >= min_units &&
                                 (env_min || (long long)sup_list.size() * 16 >= n_stored);
            if (!use_sup) {  // (too few: no launch of their own; the stacked units below may still pay)
                for (int n : sup_list) gone[n] = gone[first_child[n]] = gone[first_child[n] + 1] = 0;
                sup_list.clear();
            }
            // (the rest-list schedule needs wide units and a forest beyond the subtree blocks' reach to be used at all:
            // super_sweeps; here only the tree is known)
            if (ctx->fuse && !ctx->tune.on(T_NO_SUPER) && n_stored > 0) {
                std::vector<PmlUnit> us(std::max<size_t>(1, sup_list.size()));
                {
                    PmlUnit u;   // (padding element of the lists below when there are no two-level units)
                    u.n = order[0];
                    u.fc = first_child[order[0]];
                    u.packed = 0;
                    u.pad = 0;
                    u.cfc[0] = u.cfc[1] = u.cfc[2] = u.cfc[3] = 0;
                    us[0] = u;
                }
                for (size_t q = 0; q < sup_list.size(); ++q) {
                    const int n = sup_list[q];
                    PmlUnit u;
                    u.n = n;
                    u.fc = first_child[n];
                    u.packed = PML_PACKED_TWO_STORED;
                    u.cfc[0] = first_child[u.fc];
                    u.cfc[1] = first_child[u.fc + 1];
                    u.cfc[2] = u.cfc[3] = 0;
                    u.pad = first_child[u.cfc[0]];
                    us[q] = u;
                }
                // stacked units: ascending height, a node takes its two children over when both are plain units (not
                // two-level nodes, not taken over, not stacked themselves) with two stored children whose vectors are in
                // memory; only on levels of 1 024 .. 65 536 nodes (below: the narrow end's single launch; above: the
                // streaming levels' other lane shape)
                std::vector<char> stacked(n_nodes, 0), taken(n_nodes, 0), novec(n_nodes, 0);
                std::vector<int> stack_list;
                // (PASTML_HIP_STACK_MIN: smallest level that gets stacked units -- tests on small forests)
                const int stack_min = (int)ctx->tune.get(T_STACK_MIN, 1024);
                if (!ctx->tune.on(T_NO_STACK)) {
                    for (int n : sup_list) novec[first_child[n]] = novec[first_child[n] + 1] = 1;
                    auto level_size = [&](int node) { return off[fh[node]] - off[fh[node] - 1]; };
                    auto has_vec = [&](int g) { return kind[g] == PML_KIND_STORED && !novec[g]; };
                    auto plain2 = [&](int ch) {
                        return kind[ch] == PML_KIND_STORED && !gone[ch] && !stacked[ch] && !taken[ch] && n_children[ch] == 2 &&
                               has_vec(first_child[ch]) && has_vec(first_child[ch] + 1) && level_size(ch) <= 65536;
                    };
                    for (int q = 0; q < n_stored; ++q) {
                        const int n = order[q];
                        if (gone[n] || taken[n] || n_children[n] != 2 || level_size(n) < stack_min || level_size(n) > 65536) continue;
                        const int a = first_child[n], b = a + 1;
                        if (!plain2(a) || !plain2(b)) continue;
                        stacked[n] = 1;
                        taken[a] = taken[b] = novec[a] = novec[b] = 1;
                        stack_list.push_back(n);
                    }
                }
                // (A level with stacked units costs a launch more per sweep: they pay where they take most of what the
                // two-level units leave -- the balanced part of a tree -- and not at a tenth of the nodes: a random binary
                // tree of 262 144 tips had 6 132 of them, 10 % of its stored nodes, and was 3 % slower with them.
                // PASTML_HIP_STACK_MIN given: whatever their share.)
                if (!ctx->tune.on(T_STACK_MIN) &&
                    (long long)stack_list.size() * 3 * 2 < (long long)n_stored - 3 * (long long)sup_list.size())
                    stack_list.clear();
                for (int n : stack_list) gone[n] = gone[first_child[n]] = gone[first_child[n] + 1] = 1;
                if (!stack_list.empty()) {
                    std::vector<int> depth_of(n_nodes, 0);
                    for (int l = 0; l < n_td_levels; ++l)
                        for (int i = td_offsets[l]; i < td_offsets[l + 1]; ++i) depth_of[i] = l;
                    // by bottom-up level (stack_list is in that order already) and by depth
                    U.stack_bu_offsets.assign(max_h + 1, 0);
                    for (int n : stack_list) ++U.stack_bu_offsets[fh[n]];
                    for (int l = 0; l < max_h; ++l) U.stack_bu_offsets[l + 1] += U.stack_bu_offsets[l];
                    std::vector<int> by_depth(stack_list);
                    std::stable_sort(by_depth.begin(), by_depth.end(), [&](int x, int y) { return depth_of[x] < depth_of[y]; });
                    U.stack_td_offsets.assign(n_td_levels + 1, 0);
                    for (int n : by_depth) ++U.stack_td_offsets[depth_of[n] + 1];
                    for (int l = 0; l < n_td_levels; ++l) U.stack_td_offsets[l + 1] += U.stack_td_offsets[l];
                    std::vector<PmlUnit> sb, sd, sc;
                    describe(stack_list.data(), (int)stack_list.size(), true, sb);
                    describe(by_depth.data(), (int)by_depth.size(), true, sd);
                    std::vector<int> ch_list;
                    for (int n : stack_list) {
                        ch_list.push_back(first_child[n]);
                        ch_list.push_back(first_child[n] + 1);
                    }
                    describe(ch_list.data(), (int)ch_list.size(), true, sc);
                    PML_TRY(dev_alloc(ctx, &U.d_stack_bu, sb.size()));
                    PML_TRY(dev_alloc(ctx, &U.d_stack_td, sd.size()));
                    PML_TRY(dev_alloc(ctx, &U.d_stack_children, sc.size()));
                    PML_TRY(upload(ctx, U.d_stack_bu, sb.data(), sb.size()));
                    PML_TRY(upload(ctx, U.d_stack_td, sd.data(), sd.size()));
                    PML_TRY(upload(ctx, U.d_stack_children, sc.data(), sc.size()));
                    HIP_TRY(hipStreamSynchronize(ctx->stream));
                    U.n_stack = (int)stack_list.size();
                    if (ctx->tune.on(T_DEBUG)) fprintf(stderr, "pastml_hip: %d stacked units\n", U.n_stack);
                }
                if (!sup_list.empty() || !stack_list.empty()) {  // (else: the plain level lists, nothing to build)
                // rest lists: the level structure of the fused lists, without the nodes the two-level units take over
                std::vector<int> bu_r, td_r;
                U.bu_offsets_r.assign(1, 0);
                for (int l = 0; l < max_h; ++l) {
                    for (int q = off[l]; q < off[l + 1]; ++q)
                        if (!gone[order[q]]) bu_r.push_back(order[q]);
                    U.bu_offsets_r.push_back((int)bu_r.size());
                }
                U.td_offsets_r.assign(1, 0);
                for (int l = 0; l < n_td_levels; ++l) {
                    for (int q = ctx->td_parent_offsets_f[l]; q < ctx->td_parent_offsets_f[l + 1]; ++q)
                        if (!gone[tdp[q]]) td_r.push_back(tdp[q]);
                    U.td_offsets_r.push_back((int)td_r.size());
                }
                std::vector<PmlUnit> ubr, utr, uch;
                {
                    std::vector<int> ch_list;
                    for (int n : sup_list) {
                        ch_list.push_back(first_child[n]);
                        ch_list.push_back(first_child[n] + 1);
                    }
                    U.n_child_units = (int)ch_list.size();
                    describe(ch_list.data(), (int)ch_list.size(), true, uch);  // (at least one element)
                    PML_TRY(dev_alloc(ctx, &U.d_child_units, uch.size()));
                    PML_TRY(upload(ctx, U.d_child_units, uch.data(), uch.size()));
                }
                describe(bu_r.data(), (int)bu_r.size(), true, ubr);
                describe(td_r.data(), (int)td_r.size(), true, utr);
                U.bu_level_vec_r.assign(max_h > 0 ? max_h : 1, 0);
                for (int l = 0; l < max_h; ++l)
                    for (int q = U.bu_offsets_r[l]; q < U.bu_offsets_r[l + 1] && !U.bu_level_vec_r[l]; ++q) {
                        const int pk = ubr[q].packed;
                        if (((pk >> 8) & 7) == 1 || ((pk >> 11) & 7) == 1) U.bu_level_vec_r[l] = 1;
                    }
                // (one element of slack: the walk over a level table reads the descriptor at a level's start even when
                // the level is empty)
                ubr.resize(bu_r.size() + 1, us[0]);
                utr.resize(td_r.size() + 1, us[0]);
                PML_TRY(dev_alloc(ctx, &U.d_units, us.size()));
                PML_TRY(dev_alloc(ctx, &U.d_bu_units_r, ubr.size()));
                PML_TRY(dev_alloc(ctx, &U.d_td_units_r, utr.size()));
                PML_TRY(dev_alloc(ctx, &U.d_bu_offsets_r, U.bu_offsets_r.size()));
                PML_TRY(dev_alloc(ctx, &U.d_td_offsets_r, U.td_offsets_r.size()));
                PML_TRY(upload(ctx, U.d_units, us.data(), us.size()));
                PML_TRY(upload(ctx, U.d_bu_units_r, ubr.data(), ubr.size()));
                PML_TRY(upload(ctx, U.d_td_units_r, utr.data(), utr.size()));
                if (shape_sort) {
                    const std::vector<PmlUnit> sb = by_shape(ubr, U.bu_offsets_r, bu_r.size());
                    const std::vector<PmlUnit> st_ = by_shape_class(utr, U.td_offsets_r, td_r.size());
                    PML_TRY(dev_alloc(ctx, &U.d_bu_units_rs, sb.size()));
                    PML_TRY(dev_alloc(ctx, &U.d_td_units_rs, st_.size()));
                    PML_TRY(upload(ctx, U.d_bu_units_rs, sb.data(), sb.size()));
                    PML_TRY(upload(ctx, U.d_td_units_rs, st_.data(), st_.size()));
                    HIP_TRY(hipStreamSynchronize(ctx->stream));
                }
                PML_TRY(upload(ctx, U.d_bu_offsets_r, U.bu_offsets_r.data(), U.bu_offsets_r.size()));
                PML_TRY(upload(ctx, U.d_td_offsets_r, U.td_offsets_r.data(), U.td_offsets_r.size()));
                HIP_TRY(hipStreamSynchronize(ctx->stream));  // the vectors go out of scope
                U.n = (int)sup_list.size();
                // worth its lists: two-level units, or stacked units that take a sixteenth of the stored nodes over
                U.ok = U.n > 0 || U.n_stack >= 64 || (U.n_stack > 0 && ctx->tune.on(T_STACK_MIN));
                if (ctx->tune.on(T_DEBUG))
                    fprintf(stderr, "pastml_hip: %d two-level units (%d of %d stored nodes)%s\n", U.n, 3 * U.n, n_stored,
                            U.ok ? "" : " -- plain level lists");
                }
            }
        }
        // ---- subtree blocks: stored nodes -> blocks (maximal subtrees of <= S stored nodes) + top
        {
            const int S = (int)ctx->tune.get(T_BLOCK_NODES, 256);  // measured: 128-512 are within a few per cent, 1024+ loses at k >= 16
            pml_ctx::BlockSchedule& B = ctx->blocks;
            B = pml_ctx::BlockSchedule();
            const int cap_stored = (int)ctx->tune.get(T_BLOCK_MAX_STORED, 1 << 17);  // beyond: the streaming level kernels
            if (S > 0 && n_stored > S && n_stored <= cap_stored) {
                std::vector<int> ssz(n_nodes, 0), blk(n_nodes, -1), depth(n_nodes, 0);
                for (int l = 0; l < n_td_levels; ++l)
                    for (int i = td_offsets[l]; i < td_offsets[l + 1]; ++i) depth[i] = l;
                for (int i = n_nodes - 1; i >= 0; --i) {
                    if (kind[i] != PML_KIND_STORED) continue;
                    ssz[i] += 1;
                    if (parent[i] >= 0) ssz[parent[i]] += ssz[i];
                }
                // Height cap.  All blocks run in one launch and the top starts after it: a sweep costs (levels of the
                // tallest block) + (levels of the top).  Ragged trees have thin subtrees of few nodes and many levels;
                // uncapped, such a block outlasts all others while the top's lowest levels wait for it (HIV1C: 47 + 27
                // level steps for a tree of 57 levels).  Blocks therefore end below the lowest level of the top: what
                // sticks out joins levels the top walks anyway, and blocks + top together are as many level steps as
                // the forest has levels.  (PASTML_HIP_BLOCK_HEIGHT_CAP: 0 = no cap, n = cap at fused height n.)
                int h_cap = max_h;
                for (int q = 0; q < n_stored; ++q)
                    if (ssz[order[q]] > S) h_cap = std::min(h_cap, fh[order[q]]);
                if (ctx->tune.on(T_BLOCK_HEIGHT_CAP)) {
                    const int v = (int)ctx->tune.get(T_BLOCK_HEIGHT_CAP, 0);
                    h_cap = v > 0 ? v : max_h + 1;
                }
                int nb = 0;
                for (int i = 0; i < n_nodes; ++i) {  // parents have smaller ids
                    if (kind[i] != PML_KIND_STORED || ssz[i] > S || fh[i] >= h_cap) continue;
                    const int p = parent[i];
                    blk[i] = (p >= 0 && blk[p] >= 0) ? blk[p] : nb++;
                }
                // per block: its nodes by fused height (bottom-up) and by depth (top-down), each as consecutive levels
                std::vector<std::vector<int>> members(nb);
                for (int i = 0; i < n_nodes; ++i)
                    if (blk[i] >= 0) members[blk[i]].push_back(i);
                std::vector<int> bu_list, td_list, bu_start(nb), bu_levels(nb), bu_lv, td_start(nb), td_levels(nb), td_lv;
                for (int b = 0; b < nb; ++b) {
                    std::vector<int>& mem = members[b];  // ascending ids = non-decreasing depth
                    td_start[b] = (int)td_lv.size();
                    int nl = 0;
                    for (size_t q = 0; q < mem.size(); ++q) {
                        if (q == 0 || depth[mem[q]] != depth[mem[q - 1]]) {
                            td_lv.push_back((int)td_list.size());
                            ++nl;
                        }
                        td_list.push_back(mem[q]);
                    }
                    td_lv.push_back((int)td_list.size());
                    td_levels[b] = nl;
                    std::stable_sort(mem.begin(), mem.end(), [&](int x, int y) { return fh[x] < fh[y]; });
                    bu_start[b] = (int)bu_lv.size();
                    nl = 0;
                    for (size_t q = 0; q < mem.size(); ++q) {
                        if (q == 0 || fh[mem[q]] != fh[mem[q - 1]]) {
                            bu_lv.push_back((int)bu_list.size());
                            ++nl;
                        }
                        bu_list.push_back(mem[q]);
                    }
                    bu_lv.push_back((int)bu_list.size());
                    bu_levels[b] = nl;
                }
                // the top: stored nodes outside the blocks, by fused height / by depth
                std::vector<int> top_bu, top_td;
                B.top_bu_offsets.assign(1, 0);
                for (int l = 0; l < max_h; ++l) {
                    for (int q = off[l]; q < off[l + 1]; ++q)
                        if (blk[order[q]] < 0) top_bu.push_back(order[q]);
                    if ((int)top_bu.size() > B.top_bu_offsets.back()) B.top_bu_offsets.push_back((int)top_bu.size());
                }
                B.top_td_offsets.assign(n_td_levels + 1, 0);
                for (int l = 0; l < n_td_levels; ++l) {
                    for (int q = ctx->td_parent_offsets_f[l]; q < ctx->td_parent_offsets_f[l + 1]; ++q)
                        if (blk[tdp[q]] < 0) top_td.push_back(tdp[q]);
                    B.top_td_offsets[l + 1] = (int)top_td.size();
                }
                const int n_top_levels = (int)B.top_bu_offsets.size() - 1;
                if (nb > 0 && n_top_levels + 1 < max_h) {  // fewer dependent launches than the level schedule
                    std::vector<PmlUnit> u1, u2, u3, u4;
                    describe(bu_list.data(), (int)bu_list.size(), true, u1);
                    describe(td_list.data(), (int)td_list.size(), true, u2);
                    describe(top_bu.data(), (int)top_bu.size(), true, u3);
                    describe(top_td.data(), (int)top_td.size(), true, u4);
                    if (!ctx->tune.on(T_NO_SHAPE_SORT)) {
                        // inside every level by shape (a wave of one shape runs that shape's code only: walk_levels); the
                        // blocks' level tables lie one behind the other, so the whole array delimits the segments
                        u1 = by_shape(u1, bu_lv, bu_list.size());
                        u2 = by_shape(u2, td_lv, td_list.size());
                        u3 = by_shape(u3, B.top_bu_offsets, top_bu.size());
                        u4 = by_shape(u4, B.top_td_offsets, top_td.size());
                    }
                    B.top_bu_vec.assign(n_top_levels > 0 ? n_top_levels : 1, 0);
                    for (int l = 0; l < n_top_levels; ++l)
                        for (int q = B.top_bu_offsets[l]; q < B.top_bu_offsets[l + 1] && !B.top_bu_vec[l]; ++q) {
                            const int pk = u3[q].packed;
                            if (((pk >> 8) & 7) == 1 || ((pk >> 11) & 7) == 1) B.top_bu_vec[l] = 1;
                        }
                    auto put = [&](auto** dst, const auto& v) -> int {
                        PML_TRY(dev_alloc(ctx, dst, v.size()));
                        if (!v.empty()) PML_TRY(upload(ctx, *dst, v.data(), v.size()));
                        return PML_OK;
                    };
                    PML_TRY(put(&B.d_bu_units, u1));
                    PML_TRY(put(&B.d_td_units, u2));
                    PML_TRY(put(&B.d_top_bu_units, u3));
                    PML_TRY(put(&B.d_top_td_units, u4));
                    PML_TRY(put(&B.d_bu_start, bu_start));
                    PML_TRY(put(&B.d_bu_levels, bu_levels));
                    PML_TRY(put(&B.d_bu_lv, bu_lv));
                    PML_TRY(put(&B.d_td_start, td_start));
                    PML_TRY(put(&B.d_td_levels, td_levels));
                    PML_TRY(put(&B.d_td_lv, td_lv));
                    PML_TRY(put(&B.d_top_bu_offsets, B.top_bu_offsets));
                    PML_TRY(put(&B.d_top_td_offsets, B.top_td_offsets));
                    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the vectors go out of scope
                    B.n_blocks = nb;
                    for (int b = 0; b < nb; ++b) B.steps += bu_levels[b];
                    if (ctx->tune.on(T_DEBUG))
                        fprintf(stderr, "pastml_hip: %d stored nodes, %d subtree blocks, %lld block levels, %d top levels of %d\n",
                                n_stored, nb, B.steps, n_top_levels, max_h);
                    B.ok = true;
                }
            }
        }
        // (the thin ends of a large forest are cut into subtree blocks when the columns are known: build_thin_ends)
        ctx->h_first_child.assign(first_child, first_child + n_nodes);
        ctx->h_fh = fh;
        ctx->h_order_f.assign(order.begin(), order.begin() + n_stored);
        ctx->h_tdp = tdp;
        ctx->thin = pml_ctx::ThinSchedule();
        ctx->deep = pml_ctx::DeepSchedule();
        }
        PML_TRY(dev_alloc(ctx, &ctx->d_kind, n_nodes));
        PML_TRY(dev_alloc(ctx, &ctx->d_bu_order_f, n_stored));
        PML_TRY(dev_alloc(ctx, &ctx->d_td_parents_f, n_stored));
        PML_TRY(dev_alloc(ctx, &ctx->d_cherries, cherries.size()));
        PML_TRY(upload(ctx, ctx->d_kind, kind.data(), n_nodes));
        if (n_stored) {
            PML_TRY(upload(ctx, ctx->d_bu_order_f, order.data(), n_stored));
            PML_TRY(upload(ctx, ctx->d_td_parents_f, tdp.data(), n_stored));
        }
        if (!cherries.empty()) PML_TRY(upload(ctx, ctx->d_cherries, cherries.data(), cherries.size()));
        PML_TRY(dev_alloc(ctx, &ctx->d_bu_offsets_f, ctx->bu_offsets_f.size()));
        PML_TRY(dev_alloc(ctx, &ctx->d_td_parent_offsets_f, ctx->td_parent_offsets_f.size()));
        PML_TRY(upload(ctx, ctx->d_bu_offsets_f, ctx->bu_offsets_f.data(), ctx->bu_offsets_f.size()));
        PML_TRY(upload(ctx, ctx->d_td_parent_offsets_f, ctx->td_parent_offsets_f.data(), ctx->td_parent_offsets_f.size()));
        {
            const int limit = (int)ctx->tune.get(T_SMALL_MAX_NODES, 2048);
            ctx->small = n_nodes <= limit;
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PML_OK;
}

// ---- thin ends of a large forest (thin_bottom_up / deep_top_down).
// A ragged forest has many levels that hold a few hundred to a few thousand units: a launch of its own costs
// 5 - 15 us each, a level step inside a workgroup's walk 2 us.  Bottom-up, the thin levels are the high ones
// (every fused level from floor_level on holds at most THIN_UNITS units): subtree blocks + top over that part
// of the forest, the blocks in ONE launch behind the wide levels' launches, the top in the narrow end's launch.
// Top-down, they are the deep ones: the subtrees hanging at first_depth in ONE launch behind the wide depths.
// Small subtrees share a workgroup (bins of up to THIN_BLOCK_NODES units; units of one level of different
// subtrees do not depend on each other), so a workgroup's waves have work.
// thin: the most units a thin level holds (pml_chars_alloc: by the bytes a level of that many units moves).
static int build_thin_ends(pml_ctx* ctx, int thin) {
    const int n_nodes = ctx->N, n_td_levels = ctx->n_td_levels;
    const std::vector<int>& off = ctx->bu_offsets_f;
    const int max_h = (int)off.size() - 1;
    if (max_h <= 0) return PML_OK;
    const int n_stored = off[max_h];
    const std::vector<unsigned char>& kind = ctx->h_kind;
    const std::vector<int>&parent = ctx->h_parent, &fh = ctx->h_fh, &order = ctx->h_order_f, &tdp = ctx->h_tdp;
    auto describe = [&](const int* list, int count, bool use_kind, std::vector<PmlUnit>& out) {
        describe_units(ctx->h_first_child.data(), ctx->h_n_children.data(), kind.data(), list, count, use_kind, out);
    };
    auto by_shape = [&](const std::vector<PmlUnit>& in, const std::vector<int>& offs, size_t count) { return units_by_shape(in, offs, count); };
    pml_ctx::ThinSchedule& H = ctx->thin;
    H = pml_ctx::ThinSchedule();
    pml_ctx::DeepSchedule& D = ctx->deep;
    D = pml_ctx::DeepSchedule();
    const int S = std::max(8, (int)ctx->tune.get(T_THIN_BLOCK_NODES, 256));
    const int narrow = std::min(128, std::max(1, thin / 32));  // (levels of about this many units are the single-workgroup launch's anyway)
    auto put = [&](auto** dst, const auto& v) -> int {
        PML_TRY(dev_alloc(ctx, dst, v.size()));
        if (!v.empty()) PML_TRY(upload(ctx, *dst, v.data(), v.size()));
        return PML_OK;
    };
    // bottom-up: levels [L0, Ltop) are thin and not yet narrow
    int L0 = max_h, Ltop = max_h;
    while (L0 > 0 && off[L0] - off[L0 - 1] <= thin) --L0;
    while (Ltop > L0 && off[Ltop] - off[Ltop - 1] <= narrow) --Ltop;
    if (thin > 0 && L0 > 0 && Ltop - L0 >= 3) {
        std::vector<int> ssz(n_nodes), blk(n_nodes), bu_list, bu_start, bu_levels, bu_lv;
        int a = L0;
        while (a < Ltop) {
            // the tier's nodes: fused height in (a, hc), hc = the lowest height at which a subtree of them exceeds S
            std::fill(ssz.begin(), ssz.end(), 0);
            for (int i = n_nodes - 1; i >= 0; --i) {
                if (kind[i] != PML_KIND_STORED || fh[i] <= a) continue;
                ssz[i] += 1;
                if (parent[i] >= 0) ssz[parent[i]] += ssz[i];
            }
            int hc = max_h + 1;
            for (int q = off[a]; q < n_stored; ++q)
                if (ssz[order[q]] > S) hc = std::min(hc, fh[order[q]]);
            // subtrees into bins: the open one while it fits (parents have smaller ids)
            std::fill(blk.begin(), blk.end(), -1);
            int nb = 0, fill = 0;
            for (int i = 0; i < n_nodes; ++i) {
                if (kind[i] != PML_KIND_STORED || fh[i] <= a || fh[i] >= hc) continue;
                const int p = parent[i];
                if (p >= 0 && blk[p] >= 0) {
                    blk[i] = blk[p];
                } else {
                    if (nb == 0 || fill + ssz[i] > S) {
                        ++nb;
                        fill = 0;
                    }
                    fill += ssz[i];
                    blk[i] = nb - 1;
                }
            }
            std::vector<std::vector<int>> members(nb);
            for (int q = off[a]; q < off[hc - 1]; ++q) members[blk[order[q]]].push_back(order[q]);   // (ascending height)
            pml_ctx::ThinSchedule::Tier T;
            T.first_block = (int)bu_start.size();
            T.n_blocks = nb;
            for (int b = 0; b < nb; ++b) {
                const std::vector<int>& mem = members[b];
                bu_start.push_back((int)bu_lv.size());
                int nl = 0;
                for (size_t q = 0; q < mem.size(); ++q) {
                    if (q == 0 || fh[mem[q]] != fh[mem[q - 1]]) {
                        bu_lv.push_back((int)bu_list.size());
                        ++nl;
                    }
                    bu_list.push_back(mem[q]);
                }
                bu_lv.push_back((int)bu_list.size());
                bu_levels.push_back(nl);
            }
            H.tiers.push_back(T);
            if (ctx->tune.on(T_DEBUG))
                fprintf(stderr, "pastml_hip: thin bottom-up tier: levels %d .. %d of %d, %d units in %d bins\n", a, hc - 2,
                        max_h, off[hc - 1] - off[a], nb);
            a = hc - 1;
        }
        if ((int)H.tiers.size() + 2 <= a - L0) {   // (launches saved)
            std::vector<PmlUnit> u1;
            describe(bu_list.data(), (int)bu_list.size(), true, u1);
            if (!ctx->tune.on(T_NO_SHAPE_SORT)) u1 = by_shape(u1, bu_lv, bu_list.size());
            u1.push_back(u1[0]);  // (slack: walk_levels fetches a level's first unit before it looks at its size)
            PML_TRY(put(&H.d_units, u1));
            PML_TRY(put(&H.d_start, bu_start));
            PML_TRY(put(&H.d_levels, bu_levels));
            PML_TRY(put(&H.d_lv, bu_lv));
            HIP_TRY(hipStreamSynchronize(ctx->stream));  // the vectors go out of scope
            H.floor_level = L0;
            H.top_level = a;
            H.ok = true;
        }
    }
    // top-down: the depths behind the widest one
    const std::vector<int>& toff = ctx->td_parent_offsets_f;
    int widest = 0;
    for (int l = 1; l < n_td_levels; ++l)
        if (toff[l + 1] - toff[l] > toff[widest + 1] - toff[widest]) widest = l;
    int D0 = n_td_levels;
    while (D0 > widest + 1 && toff[D0] - toff[D0 - 1] <= thin) --D0;
    int n_mid = 0;
    for (int l = D0; l < n_td_levels; ++l) n_mid += toff[l + 1] - toff[l] > narrow;
    if (thin > 0 && D0 > 0 && D0 < n_td_levels && n_mid >= 3) {
        // a unit's bin: that of its parent's unit; the units of depth D0 open the subtrees
        std::vector<int> bin(n_nodes, -1), size(n_nodes, 0);
        for (int q = n_stored - 1; q >= toff[D0]; --q) {   // (the lists ascend in depth: children come later)
            const int n = tdp[q];
            size[n] += 1;
            if (q >= toff[D0 + 1]) size[parent[n]] += size[n];
        }
        int nb = 0, fill = 0;
        for (int q = toff[D0]; q < toff[D0 + 1]; ++q) {
            const int n = tdp[q];
            if (nb == 0 || fill + size[n] > S) {
                ++nb;
                fill = 0;
            }
            fill += size[n];
            bin[n] = nb - 1;
        }
        std::vector<std::vector<int>> members(nb);
        for (int q = toff[D0]; q < n_stored; ++q) {
            const int n = tdp[q];
            if (q >= toff[D0 + 1]) bin[n] = bin[parent[n]];
            members[bin[n]].push_back(q);   // (positions: the depth of a unit is that of its list segment)
        }
        std::vector<int> depth_of_pos(n_stored - toff[D0]);
        for (int l = D0; l < n_td_levels; ++l)
            for (int q = toff[l]; q < toff[l + 1]; ++q) depth_of_pos[q - toff[D0]] = l;
        std::vector<int> td_list, td_start(nb), td_levels(nb), td_lv;
        for (int b = 0; b < nb; ++b) {
            const std::vector<int>& mem = members[b];   // ascending positions = non-decreasing depth
            td_start[b] = (int)td_lv.size();
            int nl = 0;
            for (size_t q = 0; q < mem.size(); ++q) {
                if (q == 0 || depth_of_pos[mem[q] - toff[D0]] != depth_of_pos[mem[q - 1] - toff[D0]]) {
                    td_lv.push_back((int)td_list.size());
                    ++nl;
                }
                td_list.push_back(tdp[mem[q]]);
            }
            td_lv.push_back((int)td_list.size());
            td_levels[b] = nl;
        }
        std::vector<PmlUnit> u2;
        describe(td_list.data(), (int)td_list.size(), true, u2);
        if (!ctx->tune.on(T_NO_SHAPE_SORT)) u2 = by_shape(u2, td_lv, td_list.size());
        u2.push_back(u2[0]);
        PML_TRY(put(&D.d_units, u2));
        PML_TRY(put(&D.d_start, td_start));
        PML_TRY(put(&D.d_levels, td_levels));
        PML_TRY(put(&D.d_lv, td_lv));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        D.first_depth = D0;
        D.n_blocks = nb;
        D.ok = true;
        if (ctx->tune.on(T_DEBUG))
            fprintf(stderr, "pastml_hip: thin top-down depths from %d of %d: %d units in %d bins\n", D0, n_td_levels,
                    (int)td_list.size(), nb);
    }
    return PML_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
int pml_chars_alloc(pml_ctx* ctx, int32_t n_cols, int32_t k) {
    if (!ctx || ctx->N == 0) return fail(PML_ERR_INVALID, "upload the tree first");
    if (n_cols <= 0 || n_cols > 65535) return fail(PML_ERR_INVALID, "n_cols must be in 1..65535");
    if (k <= 0) return fail(PML_ERR_INVALID, "k must be positive");
    if (k > PML_MAX_STATES) return fail(PML_ERR_UNSUPPORTED, "k = %d states; at most %d are supported", k, PML_MAX_STATES);
    if (ctx->C != 0) return fail(PML_ERR_INVALID, "columns already allocated for this tree (upload the tree again to reset)");
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->C = n_cols;
    ctx->k = k;
    pick_group(ctx, k, ctx->G, ctx->R);
    ctx->ks = (k + ctx->R - 1) / ctx->R * ctx->R;
    {
        // F81 family: 4 states per lane (two 16-byte pairs).  Most of a unit's work is scalar (per child, per tip), so
        // the top-down kernels, which have the most of it, take 8 states per lane for 32 < k <= 64: 8 units per
        // wavefront share each scalar instruction.  PASTML_HIP_F81_R / PASTML_HIP_F81_TD_R = 2 / 4 / 8: tuning variants
        // Forests with many nodes of three or four children (15 % of the nodes with grandchildren) take 16 lanes per unit where
        // k allows it: the lane-parallel gather of a unit's children (Gather<G>) takes G / 4 of them -- two with 8 lanes, four
        // with 16 --, a unit with more walks them one after the other and holds up the other units of its wavefront.
        bool polytomies = false;
        {
            long long n_inner = 0, n34 = 0;   // (nodes with a child that has children: what is a stored node under cherry fusion)
            for (int i = 0; i < ctx->N; ++i) {
                const int nc = ctx->h_n_children[i];
                bool inner = false;
                for (int j = 0; j < nc && !inner; ++j) inner = ctx->h_n_children[ctx->h_first_child[i] + j] > 0;
                if (!inner) continue;
                ++n_inner;
                n34 += nc == 3 || nc == 4;
            }
            polytomies = n_inner > 0 && n34 * 100 >= 15 * n_inner;
        }
        auto shape = [&](int var, int dflt, int& G, int& R) {
            int rf = k >= 3 ? dflt : k;
            // Up to 8 states: two per lane (2 or 4 lanes per unit).  In the latency-bound schedules -- small and mid-size
            // forests -- a level is one wavefront's instruction stream, and half the states per lane are a shorter one
            // (HIV1C tree, 14 columns: bottom-up sweep k = 4 0.111 -> 0.094 ms, k = 8 0.107 -> 0.095; marginal pass
            // 0.290 -> 0.262, 0.319 -> 0.262; cfg2 0.0765 -> 0.0713 ms); the level kernels of large forests are
            // indifferent (262 144 tips x 32 characters, k = 4: balanced 0.73 -> 0.77 ms, ragged 2.46 -> 2.34).  One state
            // per lane is another 3 - 5 % on the small forests and costs the large balanced one a third: not taken.
            if (k >= 3 && k <= 8 && dflt == 4) rf = 2;
            // 16 < k <= 32 on forests with polytomies: 16 lanes x 2 states instead of 8 x 4 (100 000 tips, at most 3 children, x 16
            // columns, k = 20 / 32: marginal pass 1.33 -> 0.95 / 1.34 -> 0.96 ms; at most 5 children: no change; binary trees lose 6 %)
            if (k > 16 && k <= 32 && dflt == 4 && polytomies) rf = 2;
            if (ctx->tune.on(var)) {
                const int v = (int)ctx->tune.get(var, 0);
                if ((v == 2 || v == 4 || v == 8) && k > 32 && k <= 64) rf = v;
                if ((v == 2 || v == 4) && k >= 3 && k <= 8) rf = v;
                if ((v == 2 || v == 4) && k > 16 && k <= 32) rf = v;   // (16 lanes x 2 states or 8 x 4)
            }
            if (k > 256) rf = 8;   // (a wavefront per unit: 64 lanes x 8 states, k <= 512)
            R = rf;
            const int need = (k + rf - 1) / rf;
            G = 1;
            while (G < need) G <<= 1;
        };
        shape(T_F81_R, 4, ctx->Gf, ctx->Rf);
        // Balanced parts: nodes whose two children each carry two cherries of two tips, ids consecutive (what pml_tree_upload
        // makes two-level units of) -- counted on the topology alone, whatever the switches, so that the lane shape, and with
        // it a column's bits, is a function of k and the forest.
        bool balanced_parts = false;
        {
            const int* fc = ctx->h_first_child.data();
            const int* nch = ctx->h_n_children.data();
            auto tips2 = [&](int x) { return nch[x] == 2 && nch[fc[x]] == 0 && nch[fc[x] + 1] == 0; };
            auto pair = [&](int x) { return nch[x] == 2 && tips2(fc[x]) && tips2(fc[x] + 1) && fc[fc[x] + 1] == fc[fc[x]] + 2; };
            long long n_internal = 0, n_two = 0;
            for (int i = 0; i < ctx->N; ++i) {
                if (nch[i] == 0) continue;
                ++n_internal;
                if (nch[i] != 2) continue;
                const int a = fc[i], b = a + 1;
                if (pair(a) && pair(b) && fc[b] == fc[a] + 2 && fc[fc[b]] == fc[fc[a]] + 4) ++n_two;
            }
            balanced_parts = n_two > 0 && n_two * 32 >= n_internal;
        }
        // 8 states per lane bottom-up (32 < k <= 64) where the forest has such parts (cfg4: the level that rebuilds cherries
        // 1.80 -> 1.58 ms; 262 144-tip balanced tree x 32: bottom-up 0.77 -> 0.63 ms) and few nodes of three or four children
        // (8 lanes gather two children in parallel, see below); elsewhere 4: polytomies x 16 columns bottom-up 0.64 -> 0.50 ms
        // (at most 3 children), 0.61 -> 0.52 (at most 5); random binary 40 000 tips x 8 0.184 -> 0.161; 262 144 x 16 1.17 -> 1.13
        // (profiles/r05y_lane_shapes_and_sorted_levels.txt).
        ctx->bu_wide_lanes = k > 32 && k <= 64 && ctx->Rf == 4 && !ctx->tune.on(T_F81_R) && balanced_parts && !polytomies;
        if (ctx->tune.on(T_BU_WIDE)) ctx->bu_wide_lanes = k > 32 && k <= 64 && ctx->Rf == 4 && ctx->tune.get(T_BU_WIDE, 1) != 0;
        // Level launches walk the lists sorted by shape inside every level (pml_tree_upload) from 4 lanes per unit on: 262 144
        // tips x 32, marginal pass: k = 8 1.82 -> 1.78 ms, k = 12 2.35 -> 2.26, k = 16 2.38 -> 2.26, k = 20 3.49 -> 3.06, k = 32
        // 3.53 -> 3.11; polytomies k = 12 1.36 -> 1.28, k = 20 1.55 -> 1.36; two lanes per unit (k <= 4) lose 11 % and keep id order.
        ctx->level_lists_sorted = ctx->Gf >= 4 || ctx->shape_ordered;   // (narrow units too where the numbering follows the shapes)
        if (ctx->tune.on(T_SORT_LEVELS)) ctx->level_lists_sorted = ctx->tune.get(T_SORT_LEVELS, 1) != 0;
        // Top-down: 8 states per lane for 32 < k <= 64 (above) unless the forest has many nodes of three or four children
        // (15 % of those with grandchildren): the lane-parallel gather of a unit's children takes
        // G / 4 of them (Gather<G>::CH: two with 8 lanes, four with 16), a unit with more walks them one after the other and
        // holds up the other units of its wavefront.  Measured, marginal pass, k = 64 (profiles/r05y_lane_shapes_and_sorted_levels.txt):
        // 100 000 tips, at most 3 children per node, x 8 / 16 / 32 columns 1.37 -> 0.93 / 1.79 -> 1.38 / 2.84 -> 2.26 ms; at
        // most 5 children x 16 2.12 -> 1.90; at most 8 2.52 -> 2.28; binary trees lose 5 % with 4 states per lane.  The shape
        // follows k and the forest, never the columns.
        shape(T_F81_TD_R, (k > 32 && k <= 64 && !polytomies) ? 8 : 4, ctx->Gt, ctx->Rt);
        if (k >= 2 && (ctx->ks & 1)) ctx->ks += 1;  // 16-byte lane accesses
        {
            const int g = ctx->bu_wide_lanes ? 8 : ctx->Gf;
            const int nl = (int)ctx->bu_offsets_f.size() - 1;
            long long passes = 0;
            for (int l = 0; l < nl; ++l)
                passes += ((long long)(ctx->bu_offsets_f[l + 1] - ctx->bu_offsets_f[l]) * g + PML_SMALL_BLOCK - 1) / PML_SMALL_BLOCK;
            // (up to 4 lanes per unit, k <= 16: with wider units one workgroup per column is too little parallelism --
            // HIV1C tree, 64 columns: k = 12 0.28 against 0.32 ms with level launches, k = 20 0.27 against 0.22)
            ctx->levels_fit_workgroup = nl > 0 && g <= 4 && passes * 4 <= (long long)nl * 5;
        }
    }
    ctx->W = (k + 63) / 64;
    if ((size_t)ctx->N * ctx->ks >= (1ull << 31)) {
        // the kernels address one column's slab with 32-bit element offsets
        ctx->C = 0;
        return fail(PML_ERR_UNSUPPORTED, "n_nodes * k = %zu exceeds 2^31 elements per column", (size_t)ctx->N * ctx->ks);
    }
    const size_t CN = (size_t)n_cols * ctx->N;
    PML_TRY(dev_alloc(ctx, &ctx->d_masks, CN * ctx->W));
    ctx->n_params = (size_t)n_cols * (ctx->ks + 6);  // pi, sf, tau, tau factor, mu, kappa, active
    PML_TRY(dev_alloc(ctx, &ctx->d_params, ctx->n_params));
    HIP_TRY(hipHostMalloc((void**)&ctx->h_params, sizeof(double) * ctx->n_params));
    memset(ctx->h_params, 0, sizeof(double) * ctx->n_params);
    for (int i = 0; i < n_cols; ++i) ctx->h_params[(size_t)n_cols * (ctx->ks + 5) + i] = 1.0;  // every column active
    ctx->d_pi = ctx->d_params;
    ctx->d_sf = ctx->d_pi + (size_t)n_cols * ctx->ks;
    ctx->d_tau = ctx->d_sf + n_cols;
    ctx->d_tauf = ctx->d_tau + n_cols;
    ctx->d_mu = ctx->d_tauf + n_cols;
    ctx->d_kappa = ctx->d_mu + n_cols;
    ctx->d_active = ctx->d_kappa + n_cols;
    ctx->active_partial = false;
    ctx->n_active = ctx->sched_cols = n_cols;
    PML_TRY(dev_alloc(ctx, &ctx->d_err, n_cols));
    HIP_TRY(hipHostMalloc((void**)&ctx->h_loglik, sizeof(double) * n_cols));
    HIP_TRY(hipHostMalloc((void**)&ctx->h_err, sizeof(u64) * n_cols));
    HIP_TRY(hipHostMalloc((void**)&ctx->h_done, 64));
    *ctx->h_done = 0;
    PML_TRY(dev_alloc(ctx, &ctx->d_done, 2));
    HIP_TRY(hipMemsetAsync(ctx->d_done, 0, 2 * sizeof(u64), ctx->stream));
    ctx->done_expect = 0;
    ctx->wait_signal = false;
    ctx->graphs = !ctx->tune.on(T_NO_GRAPH);
    PML_TRY(dev_alloc(ctx, &ctx->d_bu, CN * ctx->ks));
    PML_TRY(dev_alloc(ctx, &ctx->d_S, CN));
    PML_TRY(dev_alloc(ctx, &ctx->d_be, CN));
    PML_TRY(dev_alloc(ctx, &ctx->d_E, CN));
    HIP_TRY(hipMemcpyAsync(ctx->d_params, ctx->h_params, ctx->n_params * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_be, 0, CN * sizeof(i64), ctx->stream));
    // default masks: everything allowed
    {
        dim3 grid(grid_for(ctx, (int)std::min<size_t>((size_t)ctx->N * ctx->W, 1u << 30), PML_BLOCK, n_cols), n_cols);
        hipLaunchKernelGGL(masks_fill_kernel, grid, dim3(PML_BLOCK), 0, ctx->stream, ctx->N, ctx->W, ctx->k,
                           ctx->d_masks, 0);
        HIP_TRY(hipGetLastError());
    }
    ctx->model_set.assign(n_cols, 0);
    ctx->eig_sym.assign(n_cols, 0);
    ctx->eig_sym_all = false;
    ctx->tips_observed.assign(n_cols, 0);
    ctx->prep_dirty = true;
    ctx->bu_mode = -1;
    ctx->td_valid = ctx->js_valid = false;
    // Thin ends of a large forest.  A level is thin while a launch of its own is mostly latency: up to 4 096 units, and up to
    // THIN_BYTES (20 MB) of state vectors over all columns -- beyond, the level kernels stream it faster than workgroups
    // that walk subtrees.  Measured (profiles/r05u_thin_ends.txt, THIN_UNITS sweeps): k = 4 x 32 columns 4 096 (16 384 loses
    // 20 %); k = 64: x 32 columns 1 024 - 2 048, x 16 2 048, x 8 4 096; k = 20 x 32 4 096.
    if (!wide_states(ctx) && !ctx->bu_offsets_f.empty() && (ctx->bu_offsets_f.back() > 2048 || ctx->tune.on(T_THIN_UNITS)) && !ctx->tune.on(T_NO_THIN)) {
        long long thin = ctx->tune.get(T_THIN_UNITS, 0);
        if (!ctx->tune.on(T_THIN_UNITS)) {
            const long long bytes = ctx->tune.get(T_THIN_BYTES, 20ll << 20);
            thin = std::min(4096ll, std::max(256ll, bytes / ((long long)n_cols * ctx->ks * 8)));
        }
        PML_TRY(build_thin_ends(ctx, (int)thin));
    }
    return PML_OK;
}

static int check_cols(pml_ctx* ctx, int cb, int ce) {
    if (!ctx || ctx->C == 0) return fail(PML_ERR_INVALID, "allocate the columns first");
    if (cb < 0 || ce > ctx->C || cb >= ce) return fail(PML_ERR_INVALID, "column range [%d, %d) out of 0..%d", cb, ce, ctx->C);
    HIP_TRY(hipSetDevice(ctx->device));
    return PML_OK;
}

static int materialize_tip_posteriors(pml_ctx* ctx);

static void invalidate(pml_ctx* ctx) {
    ctx->prep_dirty = true;
    ctx->bu_mode = -1;
    ctx->td_valid = ctx->js_valid = false;
}

// what is known about the tips of columns [col_begin, col_end); the captured launch sequence of the joint sweep depends
// on whether ALL columns' tips are observed (launch_eigen_joint_tips), so a change of that drops the graph
static void note_tips_observed(pml_ctx* ctx, int col_begin, int col_end, bool observed) {
    auto all = [&]() {
        bool a = !ctx->tips_observed.empty();
        for (char f : ctx->tips_observed) a = a && f != 0;
        return a;
    };
    const bool before = all();
    for (int col = col_begin; col < col_end; ++col) ctx->tips_observed[col] = observed ? 1 : 0;
    if (all() != before) drop_graph(ctx->bu_graph[0]);
}

int pml_masks_upload(pml_ctx* ctx, int32_t col_begin, int32_t col_end, const uint64_t* masks) {
    PML_TRY(check_cols(ctx, col_begin, col_end));
    PML_TRY(materialize_tip_posteriors(ctx));  // (rows left implicit are defined by the masks about to change)
    if (!masks) return fail(PML_ERR_INVALID, "masks is NULL");
    const size_t per_col = (size_t)ctx->N * ctx->W;
    // bits beyond k must be clear: the kernels trust the words
    if (ctx->k % 64) {
        const u64 valid = (1ull << (ctx->k % 64)) - 1ull;
        const size_t n = per_col * (col_end - col_begin);
        for (size_t i = ctx->W - 1; i < n; i += ctx->W)
            if (masks[i] & ~valid) return fail(PML_ERR_INVALID, "mask word %zu has bits beyond k = %d", i, ctx->k);
    }
    std::vector<u64> reordered;   // (the caller's rows in the library's numbering; alive until the copy below has been waited for)
    const u64* src = (const u64*)masks;
    if (permuted(ctx)) {
        reordered.resize(per_col * (col_end - col_begin));
        rows_to_internal(ctx, src, reordered.data(), (size_t)ctx->W, (size_t)(col_end - col_begin));
        src = reordered.data();
    }
    PML_TRY(upload(ctx, ctx->d_masks + col_begin * per_col, src, per_col * (col_end - col_begin)));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    note_tips_observed(ctx, col_begin, col_end, false);
    invalidate(ctx);
    return PML_OK;
}

int pml_masks_from_tip_states(pml_ctx* ctx, int32_t col_begin, int32_t col_end, int32_t n_tips,
                              const int32_t* tip_ids, const int32_t* states) {
    PML_TRY(check_cols(ctx, col_begin, col_end));
    PML_TRY(materialize_tip_posteriors(ctx));
    if (n_tips < 0 || (n_tips > 0 && (!tip_ids || !states))) return fail(PML_ERR_INVALID, "bad tip arrays");
    const int nc = col_end - col_begin;
    std::vector<int32_t> own_ids;   // the caller's tip ids in the library's numbering
    for (int j = 0; j < n_tips; ++j)
        if (tip_ids[j] < 0 || tip_ids[j] >= ctx->N || ctx->h_n_children[internal_id(ctx, tip_ids[j])] != 0)
            return fail(PML_ERR_INVALID, "tip_ids[%d] = %d is not a tip", j, tip_ids[j]);
    if (permuted(ctx) && n_tips > 0) {
        own_ids.resize(n_tips);
        for (int j = 0; j < n_tips; ++j) own_ids[j] = ctx->new_of_old[tip_ids[j]];
        tip_ids = own_ids.data();
    }
    for (size_t i = 0; i < (size_t)nc * n_tips; ++i)
        if (states[i] >= ctx->k) return fail(PML_ERR_INVALID, "state %d out of range (k = %d)", states[i], ctx->k);
    dim3 grid(grid_for(ctx, (int)std::min<size_t>((size_t)ctx->N * ctx->W, 1u << 30), PML_BLOCK, nc), nc);
    hipLaunchKernelGGL(masks_fill_kernel, grid, dim3(PML_BLOCK), 0, ctx->stream, ctx->N, ctx->W, ctx->k, ctx->d_masks,
                       col_begin);
    HIP_TRY(hipGetLastError());
    if (n_tips > 0) {
        int *d_ids = nullptr, *d_states = nullptr;
        HIP_TRY(hipMalloc((void**)&d_ids, sizeof(int) * n_tips));
        hipError_t e = hipMalloc((void**)&d_states, sizeof(int) * (size_t)nc * n_tips);
        if (e != hipSuccess) {
            (void)hipFree(d_ids);
            return fail(PML_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
        }
        e = hipMemcpyAsync(d_ids, tip_ids, sizeof(int) * n_tips, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(d_states, states, sizeof(int) * (size_t)nc * n_tips, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            dim3 g2(grid_for(ctx, n_tips, PML_BLOCK, nc), nc);
            hipLaunchKernelGGL(masks_tips_kernel, g2, dim3(PML_BLOCK), 0, ctx->stream, ctx->N, ctx->W, ctx->k,
                               ctx->d_masks, col_begin, n_tips, d_ids, d_states);
            e = hipGetLastError();
        }
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_ids);
        (void)hipFree(d_states);
        if (e != hipSuccess) return fail(PML_ERR_HIP, "tip mask upload failed: %s", hipGetErrorString(e));
        if (e2 != hipSuccess) return fail(PML_ERR_HIP, "tip mask upload failed: %s", hipGetErrorString(e2));
    }
    for (int col = col_begin; col < col_end; ++col) {
        bool all = n_tips == ctx->n_tips;   // (the ids are distinct tips or the masks would not be what the caller meant)
        for (int j = 0; all && j < n_tips; ++j) all = states[(size_t)(col - col_begin) * n_tips + j] >= 0;
        note_tips_observed(ctx, col, col + 1, all);
    }
    invalidate(ctx);
    return PML_OK;
}

int pml_masks_initial_upload(pml_ctx* ctx, int32_t col_begin, int32_t col_end, const uint64_t* masks) {
    PML_TRY(check_cols(ctx, col_begin, col_end));
    if (!masks) {
        ctx->has_init = false;
        return PML_OK;
    }
    const size_t per_col = (size_t)ctx->N * ctx->W;
    if (!ctx->d_masks_init) {
        PML_TRY(dev_alloc(ctx, &ctx->d_masks_init, per_col * ctx->C));
        // columns never given initial masks compare equal to nothing altered: start from the current masks
        HIP_TRY(hipMemcpyAsync(ctx->d_masks_init, ctx->d_masks, per_col * ctx->C * sizeof(u64), hipMemcpyDeviceToDevice,
                               ctx->stream));
    }
    if (ctx->k % 64) {
        const u64 valid = (1ull << (ctx->k % 64)) - 1ull;
        const size_t n = per_col * (col_end - col_begin);
        for (size_t i = ctx->W - 1; i < n; i += ctx->W)
            if (masks[i] & ~valid) return fail(PML_ERR_INVALID, "mask word %zu has bits beyond k = %d", i, ctx->k);
    }
    std::vector<u64> reordered;
    const u64* src = (const u64*)masks;
    if (permuted(ctx)) {
        reordered.resize(per_col * (col_end - col_begin));
        rows_to_internal(ctx, src, reordered.data(), (size_t)ctx->W, (size_t)(col_end - col_begin));
        src = reordered.data();
    }
    PML_TRY(upload(ctx, ctx->d_masks_init + col_begin * per_col, src, per_col * (col_end - col_begin)));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->has_init = true;
    ctx->bu_mode = -1;
    return PML_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
static int set_common(pml_ctx* ctx, int kind, int cb, int ce, const double* pi, const double* sf, const double* tau,
                      const double* tauf) {
    PML_TRY(check_cols(ctx, cb, ce));
    if (!pi || !sf || !tau || !tauf) return fail(PML_ERR_INVALID, "NULL parameter array");
    if (ctx->kind != -1 && ctx->kind != kind) {
        // one model kind per ctx; setting ALL columns at once may change it (a pooled ctx serving the next analysis)
        if (cb != 0 || ce != ctx->C) return fail(PML_ERR_INVALID, "all columns of a ctx must use one model kind");
        drop_sweep_graphs(ctx);
    }
    if (kind == PML_MODEL_HKY && ctx->k != 4) return fail(PML_ERR_INVALID, "HKY needs k = 4");
    const int nc = ce - cb;
    for (int i = 0; i < nc; ++i)
        if (!(sf[i] > 0.0) || !(tau[i] >= 0.0) || !(tauf[i] > 0.0) || !std::isfinite(sf[i]) || !std::isfinite(tau[i]))
            return fail(PML_ERR_INVALID, "bad sf/tau/tau_factor for column %d", cb + i);
    ctx->kind = kind;
    // into the pinned mirror (the caller's arrays need not outlive the call); params_flush sends it
    double* h = ctx->h_params;
    const size_t C = ctx->C, ks = ctx->ks, k = ctx->k;
    for (int i = 0; i < nc; ++i) {
        double* row = h + (size_t)(cb + i) * ks;
        memcpy(row, pi + (size_t)i * k, k * sizeof(double));
        for (size_t q = k; q < ks; ++q) row[q] = 0.0;
    }
    memcpy(h + C * ks + cb, sf, nc * sizeof(double));
    memcpy(h + C * ks + C + cb, tau, nc * sizeof(double));
    memcpy(h + C * ks + 2 * C + cb, tauf, nc * sizeof(double));
    for (int i = cb; i < ce; ++i) ctx->model_set[i] = 1;
    invalidate(ctx);
    return PML_OK;
}

// One asynchronous copy of the parameter block (small: (k + 5) doubles per column).  A copy still in flight when the
// mirror is written again is harmless: copies are stream-ordered and the later one carries the final contents.
// A parameter update only marks the pinned mirror; the copy goes out with the next sweep (params_push) -- inside its
// graph when the sweep is replayed as one, so that an optimiser step is ONE call into the runtime (a graph launch) instead
// of two (≈4 us of a 90 us pass on the latency-bound configurations).  The mirror is not touched while a sweep is in
// flight: every sweep is waited for before its results are used, and parameters change between sweeps.
static int params_flush(pml_ctx* ctx) {
    ctx->params_dirty = true;
    return PML_OK;
}

static int params_push(pml_ctx* ctx) {
    if (!ctx->capturing && !ctx->params_dirty) return PML_OK;
    HIP_TRY(hipMemcpyAsync(ctx->d_params, ctx->h_params, ctx->n_params * sizeof(double), hipMemcpyHostToDevice,
                           ctx->stream));
    if (ctx->capturing) ctx->capture_saw_params = true;
    else ctx->params_dirty = false;
    return PML_OK;
}

int pml_model_set_f81(pml_ctx* ctx, int32_t col_begin, int32_t col_end, const double* pi, const double* sf,
                      const double* tau, const double* tau_factor) {
    PML_TRY(set_common(ctx, PML_MODEL_F81, col_begin, col_end, pi, sf, tau, tau_factor));
    const int nc = col_end - col_begin;
    double* mu = ctx->h_params + (size_t)ctx->C * (ctx->ks + 3) + col_begin;
    for (int c = 0; c < nc; ++c) {
        // mu = 1 / (1 - sum pi^2), F81Model.py:18-26 (numpy dot)
        double dot = 0.0;
        for (int s = 0; s < ctx->k; ++s) dot += pi[(size_t)c * ctx->k + s] * pi[(size_t)c * ctx->k + s];
        mu[c] = 1.0 / (1.0 - dot);
    }
    return params_flush(ctx);
}

int pml_model_set_hky(pml_ctx* ctx, int32_t col_begin, int32_t col_end, const double* pi, const double* kappa,
                      const double* sf, const double* tau, const double* tau_factor) {
    if (!kappa) return fail(PML_ERR_INVALID, "kappa is NULL");
    PML_TRY(set_common(ctx, PML_MODEL_HKY, col_begin, col_end, pi, sf, tau, tau_factor));
    memcpy(ctx->h_params + (size_t)ctx->C * (ctx->ks + 4) + col_begin, kappa, (col_end - col_begin) * sizeof(double));
    return params_flush(ctx);
}

int pml_model_set_eigen(pml_ctx* ctx, int32_t col_begin, int32_t col_end, const double* pi, const double* d,
                        const double* A, const double* Ainv, const double* sf, const double* tau,
                        const double* tau_factor) {
    if (!d || !A || !Ainv) return fail(PML_ERR_INVALID, "NULL eigen array");
    if (ctx && ctx->k > PML_MAX_STATES_MATRIX)
        return fail(PML_ERR_UNSUPPORTED, "k = %d states: the eigen-decomposed models take at most %d (the F81 family %d)", ctx->k,
                    PML_MAX_STATES_MATRIX, PML_MAX_STATES);
    PML_TRY(set_common(ctx, PML_MODEL_EIGEN, col_begin, col_end, pi, sf, tau, tau_factor));
    const size_t k = ctx->k;
    if (!ctx->d_d) {
        PML_TRY(dev_alloc(ctx, &ctx->d_d, (size_t)ctx->C * k));
        PML_TRY(dev_alloc(ctx, &ctx->d_A, (size_t)ctx->C * k * k));
        PML_TRY(dev_alloc(ctx, &ctx->d_Ainv, (size_t)ctx->C * k * k));
    }
    const int nc = col_end - col_begin;
    PML_TRY(params_flush(ctx));
    PML_TRY(upload(ctx, ctx->d_d + col_begin * k, d, nc * k));
    PML_TRY(upload(ctx, ctx->d_A + col_begin * k * k, A, nc * k * k));
    PML_TRY(upload(ctx, ctx->d_Ainv + col_begin * k * k, Ainv, nc * k * k));
    if (k > 64 && k <= 128) {
        // More than 64 states: the sum sweeps keep ONE matrix in LDS (pml_kernels_eigen_gemm.h, EigGemm::SYM), which needs
        //   Ainv[m][j] = c_m A[j][m] pi[j],  c_m = 1 / sum_j pi[j] A[j][m]^2
        // -- true of a reversible model's eigenvectors (generator.py:33-51 builds nothing else) unless an eigenvalue repeats and
        // numpy left its eigenvectors far from orthogonal.  Checked entry by entry on what was handed in (to 1e-8: what is left
        // below that is numpy's rounding, which eig_sym_kernel's Newton-Schulz step removes); the sweeps of a ctx with a column
        // that fails read materialised P(t).
        if (!ctx->d_Asym) {
            PML_TRY(dev_alloc(ctx, &ctx->d_Asym, (size_t)ctx->C * k * k));
            PML_TRY(dev_alloc(ctx, &ctx->d_eigT, (size_t)ctx->C * k * k));
        }
        for (int c = 0; c < nc; ++c) {
            const double *Ac = A + (size_t)c * k * k, *Bc = Ainv + (size_t)c * k * k, *pc = pi + (size_t)c * k;
            bool ok = true;
            for (size_t j = 0; j < k; ++j) ok = ok && pc[j] > 0.0;
            for (size_t mm = 0; mm < k && ok; ++mm) {
                double g = 0.0, big = 0.0;
                for (size_t j = 0; j < k; ++j) {
                    g += pc[j] * Ac[j * k + mm] * Ac[j * k + mm];
                    big = std::max(big, std::fabs(Bc[mm * k + j]));
                }
                const double cm = 1.0 / g;
                if (!(g > 0.0) || !std::isfinite(cm)) ok = false;
                for (size_t j = 0; j < k && ok; ++j)
                    if (!(std::fabs(Bc[mm * k + j] - cm * Ac[j * k + mm] * pc[j]) <= 1e-8 * big)) ok = false;
            }
            ctx->eig_sym[col_begin + c] = ok ? 1 : 0;
        }
        // (an optimiser that moves the scaling factor alone hands in the same matrices evaluation after evaluation)
        const size_t kk = k * k;
        if (ctx->h_symA.size() != (size_t)ctx->C * (kk + k)) ctx->h_symA.assign((size_t)ctx->C * (kk + k), 0.0);
        bool same = true;
        for (int c = 0; c < nc; ++c) {
            double* last = ctx->h_symA.data() + (size_t)(col_begin + c) * (kk + k);
            if (memcmp(last, A + (size_t)c * kk, kk * sizeof(double)) != 0 || memcmp(last + kk, pi + (size_t)c * k, k * sizeof(double)) != 0) {
                same = false;
                memcpy(last, A + (size_t)c * kk, kk * sizeof(double));
                memcpy(last + kk, pi + (size_t)c * k, k * sizeof(double));
            }
        }
        if (!same) {
            PML_TRY(params_push(ctx));   // (the kernel reads the frequencies from the parameter block)
            const size_t lds = (k * (k + 1) + k) * sizeof(double);
            PML_TRY(with_lds(ctx, eig_sym_kernel, lds));
            for (int phase = 0; phase < 2; ++phase)
                hipLaunchKernelGGL(eig_sym_kernel, dim3(PML_ESYM_PARTS, nc), dim3(PML_BLOCK), lds, ctx->stream, (int)k, ctx->ks,
                                   (int)col_begin, phase, ctx->d_A, cols_of(ctx).pi, ctx->d_eigT, ctx->d_Asym);
            HIP_TRY(hipGetLastError());
        }
        bool all = true;
        for (int i = 0; i < ctx->C; ++i) all = all && (ctx->eig_sym[i] != 0 || !ctx->model_set[i]);
        if (all != ctx->eig_sym_all) drop_sweep_graphs(ctx);   // the sweeps change kernels
        ctx->eig_sym_all = all;
    }
    std::vector<double> at, a_t;
    if (k <= 64) {
        // transposed, zero-padded copies: A^-1 for the joint sweep on the vector units (k <= 32: its rows through the scalar cache) and
        // for the observed tips of the sum sweeps (a column of A^-1 is a contiguous row here); A for the P(t) batch below 16 states
        const size_t ld = k <= PML_EIGJ_STRIDE ? PML_EIGJ_STRIDE : 64, sq = ld * ld;
        if (!ctx->d_AinvT) PML_TRY(dev_alloc(ctx, &ctx->d_AinvT, (size_t)ctx->C * sq));
        at.assign((size_t)nc * sq, 0.0);
        for (int c = 0; c < nc; ++c)
            for (size_t mm = 0; mm < k; ++mm)
                for (size_t j = 0; j < k; ++j)
                    at[c * sq + j * ld + mm] = Ainv[(size_t)c * k * k + mm * k + j];
        PML_TRY(upload(ctx, ctx->d_AinvT + (size_t)col_begin * sq, at.data(), at.size()));
        if (k <= PML_EIGJ_STRIDE) {
            if (!ctx->d_AT) PML_TRY(dev_alloc(ctx, &ctx->d_AT, (size_t)ctx->C * sq));
            a_t.assign((size_t)nc * sq, 0.0);
            for (int c = 0; c < nc; ++c)
                for (size_t i = 0; i < k; ++i)
                    for (size_t mm = 0; mm < k; ++mm)
                        a_t[c * sq + mm * ld + i] = A[(size_t)c * k * k + i * k + mm];
            PML_TRY(upload(ctx, ctx->d_AT + (size_t)col_begin * sq, a_t.data(), a_t.size()));
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PML_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
static int require_model(pml_ctx* ctx) {
    if (!ctx || ctx->C == 0) return fail(PML_ERR_INVALID, "allocate the columns first");
    for (int i = 0; i < ctx->C; ++i)
        if (!ctx->model_set[i]) return fail(PML_ERR_INVALID, "model parameters of column %d were never set", i);
    HIP_TRY(hipSetDevice(ctx->device));
    return PML_OK;
}

static int ensure_transition_storage(pml_ctx* ctx) {
    if (ctx->kind != PML_MODEL_F81 && !ctx->d_P)
        PML_TRY(dev_alloc(ctx, &ctx->d_P, (size_t)ctx->C * ctx->N * ctx->k * ctx->ks));
    return PML_OK;
}

static int run_prep(pml_ctx* ctx, bool force = false) {
    if (!ctx->capturing) PML_TRY(params_push(ctx));  // (callers outside a sweep: pml_pij_batch, pml_marginal_counts, downloads)
    if (!ctx->prep_dirty && !force) return PML_OK;
    const PmlTree t = tree_of(ctx);
    const PmlCols c = cols_of(ctx);
    const PmlModel m = model_of(ctx);
    PML_TRY(ensure_transition_storage(ctx));
    PML_TRY(prof_begin(ctx));
    if (ctx->kind == PML_MODEL_F81) {
        // columns per thread: as many as still leave a few thousand blocks (the branch length is read once per chunk)
        int cpy = 1;
        const int bx = (ctx->N + PML_BLOCK - 1) / PML_BLOCK;
        while (cpy < 8 && cpy * 2 <= ctx->C && (long long)bx * ((ctx->C + 2 * cpy - 1) / (2 * cpy)) >= 4096) cpy *= 2;
        const int ny = (ctx->C + cpy - 1) / cpy;
        dim3 grid(grid_for(ctx, ctx->N, PML_BLOCK, ny), ny);
        hipLaunchKernelGGL(f81_prep_kernel, grid, dim3(PML_BLOCK), 0, ctx->stream, t, c, ctx->d_mu, ctx->d_sf,
                           ctx->d_tau, ctx->d_tauf, state_of(ctx), ctx->C, cpy);
        HIP_TRY(hipGetLastError());
    } else {
        if (ctx->kind == PML_MODEL_HKY) {
            dim3 grid(grid_for(ctx, ctx->N, PML_BLOCK, ctx->C), ctx->C);
            hipLaunchKernelGGL(pij_hky_kernel, grid, dim3(PML_BLOCK), 0, ctx->stream, t, c, m, ctx->d_P);
        } else if (ctx->k >= 2 && ctx->k <= PML_EIGJ_STRIDE && ctx->d_AT != nullptr && ctx->d_AinvT != nullptr && (ctx->ks & 1) == 0 &&
                   !ctx->tune.on(T_NO_PIJ_VALU) && (ctx->k < 16 || ctx->tune.on(T_NO_MFMA) || ctx->tune.on(T_PIJ_VALU))) {
            PML_TRY(launch_pij_valu(ctx));   // vector-unit path, pml_launch_eigen_joint.hip
        } else if (ctx->k >= 16 && ctx->k <= 32 && !ctx->tune.on(T_NO_MFMA)) {
            PML_TRY(launch_pij_mfma(ctx));   // FP64 matrix-core path, pml_launch_eigen_mfma.hip
        } else if (ctx->k > 32 && !ctx->tune.on(T_NO_MFMA) && !ctx->tune.on(T_NO_PIJ_WIDE)) {
            PML_TRY(launch_pij_wide(ctx));   // the same beyond 32 states: A^T in LDS slices
        } else {
            const int k = ctx->k;
            size_t lds = ((size_t)2 * k * (k + 1) + k) * sizeof(double);
            int use_lds = 1;
            if (lds > 64 * 1024) {  // default dynamic-LDS limit; larger state spaces read A / Ainv through the caches
                use_lds = 0;
                lds = (size_t)k * sizeof(double);
            }
            int bpb = (ctx->N + 2047) / 2048;
            if (bpb < 16) bpb = 16;
            dim3 grid((ctx->N + bpb - 1) / bpb, ctx->C);
            hipLaunchKernelGGL(pij_eigen_kernel, grid, dim3(PML_BLOCK), lds, ctx->stream, t, c, m, ctx->d_P, bpb, use_lds);
        }
        HIP_TRY(hipGetLastError());
    }
    PML_TRY(prof_end(ctx, 2, 1));
    ctx->prep_dirty = false;
    return PML_OK;
}

int pml_pij(pml_ctx* ctx, int32_t col, int32_t n_t, const double* ts, double* P_out) {
    PML_TRY(require_model(ctx));
    if (col < 0 || col >= ctx->C) return fail(PML_ERR_INVALID, "column out of range");
    if (n_t <= 0 || !ts || !P_out) return fail(PML_ERR_INVALID, "bad t / output arrays");
    PML_TRY(params_push(ctx));
    const size_t kk = (size_t)ctx->k * ctx->k;
    double *d_t = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_t, sizeof(double) * n_t));
    hipError_t e = hipMalloc((void**)&d_out, sizeof(double) * kk * n_t);
    if (e != hipSuccess) {
        (void)hipFree(d_t);
        return fail(PML_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    e = hipMemcpyAsync(d_t, ts, sizeof(double) * n_t, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        const size_t total = kk * n_t;
        dim3 grid((unsigned)std::min<size_t>((total + PML_BLOCK - 1) / PML_BLOCK, 65535));
        hipLaunchKernelGGL(pij_explicit_kernel, grid, dim3(PML_BLOCK), 0, ctx->stream, cols_of(ctx), model_of(ctx), col,
                           n_t, d_t, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(P_out, d_out, sizeof(double) * kk * n_t, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_t);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(PML_ERR_HIP, "pml_pij failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(PML_ERR_HIP, "pml_pij failed: %s", hipGetErrorString(e2));
    return PML_OK;
}

int pml_pij_batch(pml_ctx* ctx, double* P_out) {
    PML_TRY(require_model(ctx));
    PML_TRY(run_prep(ctx));
    if (P_out) {
        const size_t kk = (size_t)ctx->k * ctx->k;
        double* d_out = nullptr;
        HIP_TRY(hipMalloc((void**)&d_out, sizeof(double) * kk * ctx->N));
        hipError_t e = hipSuccess;
        for (int col = 0; col < ctx->C && e == hipSuccess; ++col) {
            if (ctx->kind == PML_MODEL_F81) {
                // expand the stored e per branch: same arithmetic as the explicit kernel
                const size_t total = kk * ctx->N;
                dim3 grid((unsigned)std::min<size_t>((total + PML_BLOCK - 1) / PML_BLOCK, 65535));
                hipLaunchKernelGGL(pij_explicit_kernel, grid, dim3(PML_BLOCK), 0, ctx->stream, cols_of(ctx),
                                   model_of(ctx), col, ctx->N, ctx->d_dist, d_out);
                e = hipGetLastError();
                if (e == hipSuccess)
                    e = hipMemcpyAsync(P_out + (size_t)col * ctx->N * kk, d_out, sizeof(double) * kk * ctx->N,
                                       hipMemcpyDeviceToHost, ctx->stream);
            } else {
                // the matrices the sweeps use, transposed back on the host
                std::vector<double> tmp((size_t)ctx->N * ctx->k * ctx->ks);
                e = hipMemcpyAsync(tmp.data(), ctx->d_P + (size_t)col * ctx->N * ctx->k * ctx->ks,
                                   tmp.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e == hipSuccess) {
                    double* out = P_out + (size_t)col * ctx->N * kk;
                    for (int n = 0; n < ctx->N; ++n)
                        for (int i = 0; i < ctx->k; ++i)
                            for (int j = 0; j < ctx->k; ++j)
                                out[(size_t)n * kk + (size_t)i * ctx->k + j] =
                                    tmp[((size_t)n * ctx->k + j) * ctx->ks + i];
                }
            }
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        (void)hipFree(d_out);
        if (e != hipSuccess) return fail(PML_ERR_HIP, "pml_pij_batch failed: %s", hipGetErrorString(e));
        rows_to_api_inplace(ctx, P_out, kk, (size_t)ctx->C);
    }
    return PML_OK;
}

// The narrow end of a large forest (the levels near the roots hold a handful of nodes each) is walked by ONE launch
// with a workgroup barrier between levels instead of one latency-bound launch per level: returns the number of
// consecutive levels, counted from the root end, that hold at most `limit` units each (0 if fewer than two do).
// Measured on MI355X: with one column (cfg2, 65 536 tips) 512 units per level is the best cut (0.205 -> 0.181 ms per
// marginal pass); one workgroup per column walks the levels, so with many columns the level kernels, which spread a
// level over the whole chip, win earlier: the limit shrinks with the number of columns.
// The fused eigen sweeps pass their own limit: a pass of theirs is a ~10 us dependent chain, so only levels that one
// workgroup finishes in a single pass per wave belong to the narrow end.
static int narrow_levels(const pml_ctx* ctx, const std::vector<int>& off, int n_levels, bool from_front, int C, int fixed_limit = 0,
                         int top_down = -1) {
    if (wide_states(ctx)) return 0;   // (no multi-level kernels beyond 256 states)
    const int limit_env = (int)ctx->tune.get(T_NARROW_UNITS, 0);
    int limit = fixed_limit > 0 ? fixed_limit : (limit_env > 0 ? limit_env : std::max(8, 512 / std::max(1, C)));
    if (fixed_limit <= 0 && limit_env <= 0 && ctx->kind == PML_MODEL_F81) {
        // ... but never below half a pass of the walking workgroup (512 threads, g lanes per unit): such a level is one
        // wavefront's work per SIMD either way, and a launch of its own costs 5 - 10 us where a level step inside the
        // walk costs 2.  Random 262 144-tip tree x 32 characters, marginal pass: k = 4 1.66 -> 1.52 ms, k = 12 2.56 -> 2.37,
        // k = 64 unchanged (profiles/r05j_narrow_units.txt, r05l_narrow_ab.txt); same bits (multi_level_shape).
        const bool td = top_down < 0 ? from_front : top_down != 0;   // (which sweep's lane shape walks the levels)
        const int g = td ? ctx->Gt : (ctx->bu_wide_lanes ? 8 : ctx->Gf);
        limit = std::max(limit, 256 / std::max(1, g));
    }
    int n = 0;
    for (int q = 0; q < n_levels; ++q) {
        const int l = from_front ? q : n_levels - 1 - q;
        if (off[l + 1] - off[l] > limit) break;
        ++n;
    }
    return n >= 2 ? n : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Everything a bottom-up sweep puts on the stream, without host synchronisation (so that it can be captured).
static int enqueue_bottom_up(pml_ctx* ctx, int is_marginal, bool small_path, bool force_prep) {
    struct Scope {
        pml_ctx* c;
        explicit Scope(pml_ctx* x) : c(x) { c->in_bu_enqueue = true; }
        ~Scope() { c->in_bu_enqueue = false; }
    } scope(ctx);
    ctx->enqueue_signals = false;
    const bool eig = eigen_fused(ctx);
    const bool gemm = is_marginal && eigen_gemm(ctx);
    const bool eigj = !is_marginal && eigen_joint_valu(ctx);
    PML_TRY(params_push(ctx));  // what the last model update left in the pinned mirror (part of the graph when captured)
    if (!small_path) {  // the single-launch kernel resets the error words itself
        hipLaunchKernelGGL(reset_err_kernel, dim3((ctx->C + 63) / 64), dim3(64), 0, ctx->stream, ctx->d_err, ctx->C,
                           eigj ? ctx->d_tip_rest_count : nullptr);
        HIP_TRY(hipGetLastError());
        // the fused eigen sweeps build P(t) themselves, the two-GEMM sweeps never need it
        if (!eig && !gemm && !eigj && !hky_fused(ctx)) PML_TRY(run_prep(ctx, force_prep));
    }
    PML_TRY(prof_begin(ctx));
    const bool fused = is_marginal && ctx->kind == PML_MODEL_F81;
    bool loglik_done = small_path;
    bool joint_fused = false;
    if (small_path) {
        // prep + every level + ln L in one launch
        PML_TRY(dispatch_small_f81(ctx, true, (ctx->prep_dirty || force_prep) ? 1 : 0));
        PML_TRY(prof_end(ctx, 0, 1));
    } else if (fused && block_schedule(ctx)) {
        // subtree blocks in one launch, then the top part: level launches, its narrow end (and ln L) in one launch
        const pml_ctx::BlockSchedule& B = ctx->blocks;
        PML_TRY(dispatch_blocks_f81(ctx, true));
        const int nl = (int)B.top_bu_offsets.size() - 1;
        const int tail = narrow_levels(ctx, B.top_bu_offsets, nl, false, ctx->sched_cols);
        for (int l = 0; l < nl - tail; ++l) {
            const int a = B.top_bu_offsets[l], b = B.top_bu_offsets[l + 1];
            ctx->units_override = B.d_top_bu_units + a;
            const int status = dispatch_sweep(ctx, B.top_bu_vec[l] ? SW_BU_MARG_FUSED : SW_BU_MARG_FUSED_NOVEC,
                                              ctx->d_bu_order_f, b - a);
            ctx->units_override = nullptr;
            PML_TRY(status);
        }
        PML_TRY(prof_end(ctx, 0, 1 + nl - tail));
        if (tail > 0) {
            PML_TRY(dispatch_small_f81(ctx, true, 0, 0, tail, B.d_top_bu_units, B.d_top_bu_offsets + (nl - tail)));
            loglik_done = true;
        }
    } else if (fused && super_sweeps(ctx)) {
        // the two-level units first (they depend on tips only), then the levels of what is left
        const pml_ctx::SuperSchedule& U = ctx->sup;
        PML_TRY(dispatch_super_f81(ctx, true));
        PML_TRY(prof_end(ctx, 4, U.n > 0 ? 1 : 0));
        PML_TRY(prof_begin(ctx));
        const int nl = (int)U.bu_offsets_r.size() - 1;
        int tail = narrow_levels(ctx, U.bu_offsets_r, nl, false, ctx->sched_cols);
        // (the narrow end's single launch walks the rest lists only: it starts above the last level with stacked units)
        for (int l = nl - 1; l >= 0 && U.n_stack > 0; --l)
            if (U.stack_bu_offsets[l + 1] > U.stack_bu_offsets[l]) {
                tail = std::min(tail, nl - 1 - l);
                break;
            }
        if (tail < 2) tail = 0;
        long long n_launch = 0;
        for (int l = 0; l < nl - tail; ++l) {
            const int a = U.bu_offsets_r[l], b = U.bu_offsets_r[l + 1];
            if (b > a) {
                ctx->units_override = (ctx->level_lists_sorted && U.d_bu_units_rs ? U.d_bu_units_rs : U.d_bu_units_r) + a;
                const int status = dispatch_sweep(ctx, U.bu_level_vec_r[l] ? SW_BU_MARG_FUSED : SW_BU_MARG_FUSED_NOVEC,
                                                  ctx->d_bu_order_f, b - a);
                ctx->units_override = nullptr;
                PML_TRY(status);
                ++n_launch;
            }
            // the level's stacked units (they read vectors of two levels down: independent of the launch above)
            if (U.n_stack > 0 && U.stack_bu_offsets[l + 1] > U.stack_bu_offsets[l]) {
                PML_TRY(dispatch_stack_f81(ctx, true, l));
                ++n_launch;
            }
        }
        PML_TRY(prof_end(ctx, 0, n_launch));
        if (tail > 0) {
            PML_TRY(dispatch_small_f81(ctx, true, 0, 0, tail, U.d_bu_units_r, U.d_bu_offsets_r + (nl - tail)));
            loglik_done = true;
        }
    } else if (fused && thin_bottom_up(ctx)) {
        // the wide levels one launch each, the thin ones in tiers of subtree blocks (a launch per tier), then the narrow
        // end's single launch (in between, level launches where a level is still too wide for it)
        const pml_ctx::ThinSchedule& H = ctx->thin;
        auto level_launch = [&](int l) -> int {
            const int a = ctx->bu_offsets_f[l], b = ctx->bu_offsets_f[l + 1];
            return dispatch_sweep(ctx, ctx->bu_level_vec_f[l] ? SW_BU_MARG_FUSED : SW_BU_MARG_FUSED_NOVEC, ctx->d_bu_order_f + a, b - a);
        };
        for (int l = 0; l < H.floor_level; ++l) PML_TRY(level_launch(l));
        for (size_t q = 0; q < H.tiers.size(); ++q) PML_TRY(dispatch_blocks_f81(ctx, true, 1 + (int)q));
        const int nl = (int)ctx->bu_offsets_f.size() - 1;
        int tail = std::min(nl - H.top_level, narrow_levels(ctx, ctx->bu_offsets_f, nl, false, ctx->sched_cols));
        for (int l = H.top_level; l < nl - tail; ++l) PML_TRY(level_launch(l));
        PML_TRY(prof_end(ctx, 0, H.floor_level + (long long)H.tiers.size() + (nl - tail - H.top_level)));
        if (tail > 0) {
            PML_TRY(dispatch_small_f81(ctx, true, 0, nl - tail, tail));
            loglik_done = true;
        }
    } else if (fused) {
        const int nl = (int)ctx->bu_offsets_f.size() - 1;
        const int tail = narrow_levels(ctx, ctx->bu_offsets_f, nl, false, ctx->sched_cols);
        for (int l = 0; l < nl - tail; ++l) {
            const int a = ctx->bu_offsets_f[l], b = ctx->bu_offsets_f[l + 1];
            PML_TRY(dispatch_sweep(ctx, ctx->bu_level_vec_f[l] ? SW_BU_MARG_FUSED : SW_BU_MARG_FUSED_NOVEC,
                                   ctx->d_bu_order_f + a, b - a));
        }
        PML_TRY(prof_end(ctx, 0, nl - tail));  // the profile brackets the level kernel's launches only
        if (tail > 0) {  // the levels next to the roots and ln L in one launch
            PML_TRY(dispatch_small_f81(ctx, true, 0, nl - tail, tail));
            loglik_done = true;
        }
    } else if (!is_marginal && ctx->kind == PML_MODEL_F81 && ctx->fuse && !ctx->has_init && ctx->n_cherries > 0 &&
               ctx->W == 1) {
        // joint sweep over the cherry-fused lists (no altered nodes whose tables would need rewriting)
        const int nl = (int)ctx->bu_offsets_f.size() - 1;
        for (int l = 0; l < nl; ++l) {
            const int a = ctx->bu_offsets_f[l], b = ctx->bu_offsets_f[l + 1];
            PML_TRY(dispatch_sweep(ctx, ctx->bu_level_vec_f[l] ? SW_BU_JOINT_FUSED : SW_BU_JOINT_FUSED_NOVEC,
                                   ctx->d_bu_order_f + a, b - a));
        }
        PML_TRY(prof_end(ctx, 0, nl));
        joint_fused = true;
    } else if (eigj) {
        // joint sweep of an eigen model on the vector units (pml_kernels_eigen_joint.h): the tips, then the levels
        PML_TRY(launch_eigen_joint_tips(ctx));
        const pml_ctx::EigenTiers& E = ctx->eig_tiers;
        // (tiers only while their levels are thin for the whole batch: with many columns a level fills the chip)
        if (E.ok && (long long)E.widest * ctx->C <= 16384) {
            for (int l = 0; l < E.first_level; ++l) {
                const int a = ctx->bu_offsets[l], b = ctx->bu_offsets[l + 1];
                PML_TRY(launch_eigen_joint(ctx, ctx->d_bu_units + a, nullptr, 0, b - a));
            }
            for (const pml_ctx::EigenTiers::Tier& T : E.tiers)
                PML_TRY(launch_eigen_joint(ctx, E.d_units, E.d_lv, 0, T.depth, E.d_start + T.first_block, T.n_blocks));
            // what is left above the tiers: levels of a launch each while they are wide (a forest of many trees), then
            // the narrow end in one launch
            int l = E.top_level;
            long long extra = 0;
            for (; l < ctx->n_bu_levels && ctx->bu_offsets[l + 1] - ctx->bu_offsets[l] > 48; ++l, ++extra) {
                const int a = ctx->bu_offsets[l], b = ctx->bu_offsets[l + 1];
                PML_TRY(launch_eigen_joint(ctx, ctx->d_bu_units + a, nullptr, 0, b - a));
            }
            PML_TRY(launch_eigen_joint(ctx, ctx->d_bu_units, ctx->d_bu_offsets, l, ctx->n_bu_levels - l));
            PML_TRY(prof_end(ctx, 0, E.first_level + 2 + extra + (long long)E.tiers.size()));
        } else {
        const int tail = narrow_levels(ctx, ctx->bu_offsets, ctx->n_bu_levels, false, ctx->C,
                                       PML_WAVES_PER_BLOCK * (64 / ctx->k));
        for (int l = 0; l < ctx->n_bu_levels - tail; ++l) {
            const int a = ctx->bu_offsets[l], b = ctx->bu_offsets[l + 1];
            PML_TRY(launch_eigen_joint(ctx, ctx->d_bu_units + a, nullptr, 0, b - a));
        }
        PML_TRY(launch_eigen_joint(ctx, ctx->d_bu_units, ctx->d_bu_offsets, ctx->n_bu_levels - tail, tail));
        PML_TRY(prof_end(ctx, 0, ctx->n_bu_levels + 1 - tail + (tail > 0 ? 1 : 0)));
        }
    } else if (gemm) {
        // marginal sweep: P(t) is never formed, msg = A (e o (A^-1 v)) as two small GEMMs per 16 nodes
        PML_TRY(launch_eigen_gemm(ctx, PML_EIGG_TIPS, ctx->d_tips, 0, ctx->n_tips));
        const pml_ctx::EigenTiers& E = ctx->eig_tiers;
        const bool gemm_tiers = !ctx->tune.on(T_NO_EIGG_TIERS);
        if (gemm_tiers && E.ok && (long long)E.widest * ctx->C <= 16384) {
            // thin levels in tiers of subtree blocks, as in the joint sweep (pml_ctx::EigenTiers)
            for (int l = 0; l < E.first_level; ++l) {
                const int a = ctx->bu_offsets[l], b = ctx->bu_offsets[l + 1];
                PML_TRY(launch_eigen_gemm(ctx, PML_EIGG_BU, ctx->d_bu_order + a, 0, b - a));
            }
            for (const pml_ctx::EigenTiers::Tier& T : E.tiers)
                PML_TRY(launch_eigen_gemm_narrow(ctx, PML_EIGG_BU, E.d_nodes, E.d_lv, 0, T.depth, E.d_start + T.first_block,
                                                 T.n_blocks));
            int l = E.top_level;
            long long extra = 0;
            for (; l < ctx->n_bu_levels && ctx->bu_offsets[l + 1] - ctx->bu_offsets[l] > 2 * PML_WAVES_PER_BLOCK * 16; ++l, ++extra) {
                const int a = ctx->bu_offsets[l], b = ctx->bu_offsets[l + 1];
                PML_TRY(launch_eigen_gemm(ctx, PML_EIGG_BU, ctx->d_bu_order + a, 0, b - a));
            }
            PML_TRY(launch_eigen_gemm_narrow(ctx, PML_EIGG_BU, ctx->d_bu_order, ctx->d_bu_offsets, l, ctx->n_bu_levels - l));
            PML_TRY(prof_end(ctx, 0, E.first_level + 2 + extra + (long long)E.tiers.size()));
        } else {
        // levels one workgroup finishes in a pass or two per wave (4 waves x 16 nodes) share one launch
        const int tail = narrow_levels(ctx, ctx->bu_offsets, ctx->n_bu_levels, false, ctx->C, 2 * PML_WAVES_PER_BLOCK * 16);
        for (int l = 0; l < ctx->n_bu_levels - tail; ++l) {
            const int a = ctx->bu_offsets[l], b = ctx->bu_offsets[l + 1];
            PML_TRY(launch_eigen_gemm(ctx, PML_EIGG_BU, ctx->d_bu_order + a, 0, b - a));
        }
        PML_TRY(launch_eigen_gemm_narrow(ctx, PML_EIGG_BU, ctx->d_bu_order, ctx->d_bu_offsets, ctx->n_bu_levels - tail, tail));
        PML_TRY(prof_end(ctx, 0, ctx->n_bu_levels + 1 - tail + (tail > 0 ? 1 : 0)));
        }
    } else if (eig) {
        // every node once, in the launch of its level: the tips first, then the internal nodes by height
        const int mode = is_marginal ? PML_EIG_BU_MARG : PML_EIG_BU_JOINT;
        PML_TRY(launch_eigen_tips(ctx, is_marginal ? 0 : 1));
        {
        const int eig_nb = ((ctx->k + 3) / 4) % 4 == 0 ? 1 : (((ctx->k + 3) / 4) % 2 == 0 ? 2 : 4);  // EigShape::NB
        const int tail = narrow_levels(ctx, ctx->bu_offsets, ctx->n_bu_levels, false, ctx->C, PML_WAVES_PER_BLOCK * eig_nb);
        for (int l = 0; l < ctx->n_bu_levels - tail; ++l) {
            const int a = ctx->bu_offsets[l], b = ctx->bu_offsets[l + 1];
            PML_TRY(launch_eigen_fused(ctx, mode, ctx->d_bu_order + a, 0, b - a, 0));
        }
        PML_TRY(launch_eigen_narrow(ctx, mode, ctx->d_bu_order, ctx->d_bu_offsets, ctx->n_bu_levels - tail, tail));
        PML_TRY(prof_end(ctx, 0, ctx->n_bu_levels + 1 - tail + (tail > 0 ? 1 : 0)));
        }
    } else {
        for (int l = 0; l < ctx->n_bu_levels; ++l) {
            const int a = ctx->bu_offsets[l], b = ctx->bu_offsets[l + 1];
            const SweepKind sk = is_marginal ? SW_BU_MARG : (ctx->kind == PML_MODEL_F81 && !ctx->bu_level_vec[l]
                                                                     ? SW_BU_JOINT_NOVEC : SW_BU_JOINT);
            PML_TRY(dispatch_sweep(ctx, sk, ctx->d_bu_order + a, b - a));
        }
        PML_TRY(prof_end(ctx, 0, ctx->n_bu_levels));
    }
    if (!loglik_done) {
        hipLaunchKernelGGL(loglik_kernel, dim3((ctx->C + PML_BLOCK - 1) / PML_BLOCK), dim3(PML_BLOCK), 0, ctx->stream,
                           tree_of(ctx), cols_of(ctx), state_of(ctx), ctx->C, is_marginal ? 1 : 0, ctx->h_loglik,
                           ctx->h_err);
        HIP_TRY(hipGetLastError());
    }
    ctx->bu_fused_joint = joint_fused;
    return PML_OK;  // ln L and the error words are written straight into pinned host memory by the last kernel
}

// Captures fn's stream work once and replays it afterwards; falls back to direct submission if capture fails.
static int run_captured(pml_ctx* ctx, pml_ctx::GraphSlot& slot, const std::function<int()>& enqueue) {
    if (ctx->in_outer_capture) return enqueue();  // part of a larger capture
    if (slot.exec && slot.has_init != ctx->has_init) drop_graph(slot);
    if (!slot.exec) {
        HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        ctx->capturing = true;
        ctx->capture_saw_params = false;
        const int status = enqueue();
        ctx->capturing = false;
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
        if (status != PML_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return status;
        }
        if (e != hipSuccess || !graph) return fail(PML_ERR_HIP, "stream capture failed: %s", hipGetErrorString(e));
        hipGraphExec_t exec = nullptr;
        const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e2 != hipSuccess) {
            (void)hipGraphDestroy(graph);
            return fail(PML_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e2));
        }
        slot.graph = graph;
        slot.exec = exec;
        slot.has_init = ctx->has_init;
        slot.has_params = ctx->capture_saw_params;
    }
    HIP_TRY(hipGraphLaunch(slot.exec, ctx->stream));
    if (slot.has_params) ctx->params_dirty = false;
    return PML_OK;
}

// puts a bottom-up sweep on the stream (no host synchronisation)
// Which columns the next bottom-up sweep computes (nullptr: all).  The flags travel with the parameter block (params_push:
// inside the captured sweep when it is replayed), so they are written while no sweep is in flight, like the parameters.
static void set_active_columns(pml_ctx* ctx, const uint8_t* active) {
    if (!ctx->h_params) return;
    double* flags = ctx->h_params + (size_t)ctx->C * (ctx->ks + 5);
    if (active == nullptr && !ctx->active_partial) {  // (all ones already)
        ctx->n_active = ctx->C;
        return;
    }
    bool partial = false, changed = false;
    int n = 0;
    for (int i = 0; i < ctx->C; ++i) {
        const double v = (active == nullptr || active[i]) ? 1.0 : 0.0;
        changed = changed || flags[i] != v;
        partial = partial || v == 0.0;
        n += v != 0.0;
        flags[i] = v;
    }
    ctx->active_partial = partial;
    ctx->n_active = n;
    if (changed) ctx->params_dirty = true;
}

static int submit_bottom_up(pml_ctx* ctx, int is_marginal, const uint8_t* active = nullptr) {
    set_active_columns(ctx, active);
    if (ctx->wait_signal) {  // a sweep was submitted and never collected: the generation below must be read on an idle stream
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->wait_signal = false;
    }
    // A sweep of a few columns of a context of many is scheduled as a context of few would be (the workgroups of the
    // other columns return at once): subtree blocks instead of one workgroup per column walking every level, the
    // completion word -- HIV1C tree, 6 of 246 binary columns: 0.15 -> 0.10 ms per sweep.  Two schedules, two captured
    // sequences: all the columns' (sched_cols = C) and a few columns' (sched_cols = 32).
    bool few = ctx->active_partial && ctx->n_active <= 32 && ctx->C > 32 && is_marginal && ctx->kind == PML_MODEL_F81;
    if (few) {  // (only where the few-column schedule is the subtree blocks: never from one launch to one per level)
        ctx->sched_cols = 32;
        few = !single_launch_sweeps(ctx) && block_schedule(ctx);
        ctx->sched_cols = ctx->C;
    }
    struct Sched {
        pml_ctx* c;
        Sched(pml_ctx* x, int cols) : c(x) { c->sched_cols = cols; }
        ~Sched() { c->sched_cols = c->C; }
    } sched(ctx, few ? 32 : ctx->C);
    const bool small_path = single_launch_sweeps(ctx) && is_marginal && ctx->kind == PML_MODEL_F81;
    const size_t CN = (size_t)ctx->C * ctx->N;
    if (!is_marginal && !ctx->d_J) {
        PML_TRY(dev_alloc(ctx, &ctx->d_J, CN * ctx->ks * (ctx->k > 256 ? 2 : 1)));
        PML_TRY(dev_alloc(ctx, &ctx->d_js, CN));
    }
    const bool no_p = eigen_fused(ctx) || (is_marginal && eigen_gemm(ctx)) || (!is_marginal && eigen_joint_valu(ctx)) ||
                      hky_fused(ctx);
    if (eigen_fused(ctx) || eigen_gemm(ctx) || eigen_joint_valu(ctx)) {
        if (!ctx->d_msg) PML_TRY(dev_alloc(ctx, &ctx->d_msg, CN * ctx->ks));
    }
    if (!is_marginal && eigen_joint_valu(ctx) && !ctx->d_tip_rest) {
        PML_TRY(dev_alloc(ctx, &ctx->d_tip_rest, (size_t)ctx->C * std::max(1, ctx->n_tips)));
        PML_TRY(dev_alloc(ctx, &ctx->d_tip_rest_count, (size_t)ctx->C));
    }
    if (!no_p) PML_TRY(ensure_transition_storage(ctx));
    ctx->bu_mode = -1;
    ctx->td_valid = ctx->js_valid = false;
    // mid-size forests: the level launches are latency-bound, replay them as one hipGraph
    const int n_launches = small_path ? 1 : (is_marginal && ctx->kind == PML_MODEL_F81 ? (int)ctx->bu_offsets_f.size() - 1
                                                                                          : ctx->n_bu_levels);
    const u64 generation_before = ctx->h_done ? *reinterpret_cast<volatile u64*>(ctx->h_done) : 0;  // (the stream is idle)
    if (ctx->graphs && !ctx->profile && n_launches >= 4) {  // (the block schedule's few launches replay as a graph too)
        const int slot = is_marginal ? 1 : 0;
        pml_ctx::GraphSlot& graph = few ? ctx->bu_graph_few : ctx->bu_graph[slot];
        bool& signals = few ? ctx->bu_signals_few : ctx->bu_signals[slot];
        const bool replay = graph.exec != nullptr;
        PML_TRY(run_captured(ctx, graph, [&]() { return enqueue_bottom_up(ctx, is_marginal, small_path, true); }));
        if (!replay) signals = ctx->enqueue_signals;  // (a replay runs what was captured)
        ctx->wait_signal = signals;
    } else {
        // (inside the capture of a whole marginal pass the per-branch pass must be part of the graph)
        PML_TRY(enqueue_bottom_up(ctx, is_marginal, small_path, ctx->in_outer_capture));
        ctx->wait_signal = ctx->enqueue_signals;
    }
    ctx->done_expect = generation_before + 1;
    if (ctx->in_outer_capture || ctx->tune.on(T_NO_SPIN_WAIT)) ctx->wait_signal = false;
    // the fused eigen sweeps build P(t) in registers, the two-GEMM sweeps never form it: no batch ran
    if (!no_p) ctx->prep_dirty = false;
    ctx->bu_fused = (is_marginal && ctx->kind == PML_MODEL_F81 && ctx->n_cherries > 0) || ctx->bu_fused_joint;
    // (a sweep of some of the columns says nothing about the others: where an earlier sweep of the level schedule left the
    // children of their two-level units out of memory they still are -- rebuilding rows that are in memory is harmless)
    ctx->bu_absorbed = (is_marginal && ctx->kind == PML_MODEL_F81 && !small_path && super_sweeps(ctx)) ||
                       (ctx->active_partial && ctx->bu_absorbed);
    return PML_OK;
}

// Waits for the bottom-up sweep submitted last.  Where its last launch raises the pinned word (bu_f81_small_kernel) the
// host spins on that word -- the results lie in pinned memory behind it -- instead of asking the runtime, which notices
// the end of a short launch sequence ~4 us later (scripts/ub/syncwait.hip); after 2 ms, or for any other sweep, it is
// hipStreamSynchronize.  Everything queued afterwards is ordered behind the sweep by the stream as before.
static int wait_bottom_up(pml_ctx* ctx) {
    if (ctx->wait_signal && ctx->h_done) {
        ctx->wait_signal = false;
        const u64* flag = ctx->h_done;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            // (acquire: the results the host reads next were written before the word was raised)
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) >= ctx->done_expect) return PML_OK;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if ((spins & 1023u) == 1023u &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
                break;
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PML_OK;
}

// after the stream has been synchronised: log-likelihoods and zero-likelihood reports of the last sweep
static int collect_bottom_up(pml_ctx* ctx, int is_marginal, double* loglik_out, int32_t* err_parent, int32_t* err_child) {
    memcpy(loglik_out, ctx->h_loglik, sizeof(double) * ctx->C);
    const u64* err = ctx->h_err;
    int status = PML_OK;
    const double* flags = ctx->h_params + (size_t)ctx->C * (ctx->ks + 5);
    for (int c = 0; c < ctx->C; ++c) {
        int ep = -1, ec = -1;
        if (err[c] != ~0ull && flags[c] != 0.0) {  // (a column that sat the sweep out reports nothing)
            ec = (int)(err[c] & 0xffffffffull);
            ep = api_id(ctx, ctx->h_parent[ec]);
            ec = api_id(ctx, ec);
            if (status == PML_OK)
                status = fail(PML_ZERO_LIKELIHOOD, "zero likelihood in column %d between parent %d and child %d", c, ep, ec);
        }
        if (err_parent) err_parent[c] = ep;
        if (err_child) err_child[c] = ec;
    }
    if (status == PML_OK) ctx->bu_mode = is_marginal ? 1 : 0;
    return status;
}

int pml_bottom_up(pml_ctx* ctx, int is_marginal, double* loglik_out, int32_t* err_parent, int32_t* err_child) {
    PML_TRY(require_model(ctx));
    if (!loglik_out) return fail(PML_ERR_INVALID, "loglik_out is NULL");
    PML_TRY(submit_bottom_up(ctx, is_marginal));
    PML_TRY(wait_bottom_up(ctx));
    return collect_bottom_up(ctx, is_marginal, loglik_out, err_parent, err_child);
}

int pml_bottom_up_submit(pml_ctx* ctx, int is_marginal) {
    PML_TRY(require_model(ctx));
    return submit_bottom_up(ctx, is_marginal);
}

int pml_bottom_up_submit_columns(pml_ctx* ctx, int is_marginal, const uint8_t* active) {
    PML_TRY(require_model(ctx));
    if (!is_marginal || ctx->kind != PML_MODEL_F81) active = nullptr;  // (only the F81 marginal kernels look at the flags)
    return submit_bottom_up(ctx, is_marginal, active);
}

int pml_bottom_up_collect(pml_ctx* ctx, int is_marginal, double* loglik_out, int32_t* err_parent, int32_t* err_child) {
    if (!ctx || ctx->C == 0) return fail(PML_ERR_INVALID, "allocate the columns first");
    if (!loglik_out) return fail(PML_ERR_INVALID, "loglik_out is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    PML_TRY(wait_bottom_up(ctx));
    return collect_bottom_up(ctx, is_marginal, loglik_out, err_parent, err_child);
}

// the top-down launches (shared by pml_top_down_marginals and the lazy TD materialisation of pml_download)
static int run_top_down(pml_ctx* ctx) {
    const size_t CN = (size_t)ctx->C * ctx->N;
    const bool td_stored = ctx->kind != PML_MODEL_F81 || ctx->keep_td;
    if (td_stored && !ctx->d_td) {
        PML_TRY(dev_alloc(ctx, &ctx->d_td, CN * ctx->ks));
        PML_TRY(dev_alloc(ctx, &ctx->d_te, CN));
    }
    if (!ctx->d_post) {
        PML_TRY(dev_alloc(ctx, &ctx->d_post, CN * ctx->ks));
        PML_TRY(dev_alloc(ctx, &ctx->d_lhsum, CN));
        PML_TRY(dev_alloc(ctx, &ctx->d_lhe, CN));
    }
    const bool td_small = single_launch_sweeps(ctx) && ctx->kind == PML_MODEL_F81;
    const bool td_fused = ctx->kind == PML_MODEL_F81;
    auto enqueue = [&]() -> int {
        if (td_fused && !td_small && block_schedule(ctx)) {
            // block schedule: the top part (roots, its narrow end in one launch, its wide levels one launch each),
            // then all subtree blocks in one launch
            const pml_ctx::BlockSchedule& B = ctx->blocks;
            const int head = ctx->n_roots <= 64 ? narrow_levels(ctx, B.top_td_offsets, ctx->n_td_levels, true, ctx->C) : 0;
            if (head == 0) PML_TRY(dispatch_sweep(ctx, SW_ROOTS, nullptr, ctx->n_roots));
            if (head > 0) PML_TRY(dispatch_small_f81(ctx, false, 0, 0, head, B.d_top_td_units, B.d_top_td_offsets));
            PML_TRY(prof_begin(ctx));
            long long n_launch = 0;
            for (int l = head; l < ctx->n_td_levels; ++l) {
                const int a = B.top_td_offsets[l], b = B.top_td_offsets[l + 1];
                if (b <= a) continue;
                ctx->units_override = B.d_top_td_units + a;
                const int status = dispatch_sweep(ctx, SW_TD_FUSED, ctx->d_td_parents_f, b - a);
                ctx->units_override = nullptr;
                PML_TRY(status);
                ++n_launch;
            }
            ctx->signal_next_td = ctx->mp_wants_signal;  // (the last launch of the pass)
            PML_TRY(dispatch_blocks_f81(ctx, false));
            PML_TRY(prof_end(ctx, 1, n_launch + 1));
            return PML_OK;
        }
        if (td_fused && !td_small && super_sweeps(ctx)) {
            // the levels of the rest lists, then every two-level unit in one launch (it needs its node's row only, and
            // that comes from a unit of the rest lists or from the roots)
            const pml_ctx::SuperSchedule& U = ctx->sup;
            int head = ctx->n_roots <= 64 ? narrow_levels(ctx, U.td_offsets_r, ctx->n_td_levels, true, ctx->C) : 0;
            // (... and the single launch below the roots ends above the first depth with stacked nodes)
            for (int l = 0; l < ctx->n_td_levels && U.n_stack > 0; ++l)
                if (U.stack_td_offsets[l + 1] > U.stack_td_offsets[l]) {
                    head = std::min(head, l);
                    break;
                }
            if (head < 2) head = 0;
            if (head == 0) PML_TRY(dispatch_sweep(ctx, SW_ROOTS, nullptr, ctx->n_roots));
            if (head > 0) PML_TRY(dispatch_small_f81(ctx, false, 0, 0, head, U.d_td_units_r, U.d_td_offsets_r));
            PML_TRY(prof_begin(ctx));
            long long n_launch = 0;
            for (int l = head; l < ctx->n_td_levels; ++l) {
                const int a = U.td_offsets_r[l], b = U.td_offsets_r[l + 1];
                if (b > a) {
                    ctx->units_override = (ctx->level_lists_sorted && U.d_td_units_rs ? U.d_td_units_rs : U.d_td_units_r) + a;
                    const int status = dispatch_sweep(ctx, SW_TD_FUSED, ctx->d_td_parents_f, b - a);
                    ctx->units_override = nullptr;
                    PML_TRY(status);
                    ++n_launch;
                }
                // the children of the stacked nodes of this depth (their rows come from the depth above)
                if (U.n_stack > 0 && U.stack_td_offsets[l + 1] > U.stack_td_offsets[l]) {
                    PML_TRY(dispatch_stack_f81(ctx, false, l));
                    ++n_launch;
                }
            }
            PML_TRY(prof_end(ctx, 1, n_launch));
            PML_TRY(prof_begin(ctx));
            PML_TRY(dispatch_super_f81(ctx, false));
            PML_TRY(prof_end(ctx, 3, U.n > 0 ? 1 : 0));
            return PML_OK;
        }
        // F81 family: the roots and the levels right below them in one launch
        const int head = (td_fused && !td_small && ctx->n_roots <= 64)
                             ? narrow_levels(ctx, ctx->td_parent_offsets_f, ctx->n_td_levels, true, ctx->C) : 0;
        if (!td_small && head == 0) PML_TRY(dispatch_sweep(ctx, SW_ROOTS, nullptr, ctx->n_roots));
        if (head > 0) PML_TRY(dispatch_small_f81(ctx, false, 0, 0, head));
        PML_TRY(prof_begin(ctx));  // the profile brackets the level kernel's launches only
        long long n_launch = 0;
        if (td_small) {
            ctx->signal_next_td = ctx->mp_wants_signal;  // (the last launch of the pass)
            PML_TRY(dispatch_small_f81(ctx, false, 0));
            n_launch = 1;
        }
        if (eigen_gemm(ctx)) {
            int head = 0;
            {
                std::vector<int> off(ctx->td_offsets.begin() + 1, ctx->td_offsets.end());
                head = narrow_levels(ctx, off, ctx->n_td_levels - 1, true, ctx->C, 2 * PML_WAVES_PER_BLOCK * 16);
            }
            if (head > 0) {
                PML_TRY(launch_eigen_gemm_narrow(ctx, PML_EIGG_TD, nullptr, ctx->d_td_offsets, 1, head));
                ++n_launch;
            }
            for (int d = 1 + head; d < ctx->n_td_levels; ++d) {
                const int a = ctx->td_offsets[d], b = ctx->td_offsets[d + 1];
                PML_TRY(launch_eigen_gemm(ctx, PML_EIGG_TD, nullptr, a, b - a));
                if (b > a) ++n_launch;
            }
            PML_TRY(prof_end(ctx, 1, n_launch));
            return PML_OK;
        }
        if (eigen_fused(ctx)) {
            // child-centric: the nodes of a depth are a contiguous id range (roots are depth 0, done above)
            // td_offsets[d] .. td_offsets[d + 1] = the nodes of depth d: the run of narrow depths below the roots
            int head = 0;
            {
                std::vector<int> off(ctx->td_offsets.begin() + 1, ctx->td_offsets.end());
                const int ks4 = (ctx->k + 3) / 4;
                const int eig_nb = ks4 % 4 == 0 ? 1 : (ks4 % 2 == 0 ? 2 : 4);  // EigShape::NB
                head = narrow_levels(ctx, off, ctx->n_td_levels - 1, true, ctx->C, PML_WAVES_PER_BLOCK * eig_nb);
            }
            if (head > 0) {
                PML_TRY(launch_eigen_narrow(ctx, PML_EIG_TD, nullptr, ctx->d_td_offsets, 1, head));
                ++n_launch;
            }
            for (int d = 1 + head; d < ctx->n_td_levels; ++d) {
                const int a = ctx->td_offsets[d], b = ctx->td_offsets[d + 1];
                PML_TRY(launch_eigen_fused(ctx, PML_EIG_TD, nullptr, a, b - a, 0));
                if (b > a) ++n_launch;
            }
            PML_TRY(prof_end(ctx, 1, n_launch));
            return PML_OK;
        }
        // F81 family: the thin depths at the DEEP end of a ragged forest (a handful of parents each) in one launch as well
        // -- a launch of their own costs 9 - 11 us each, a level step of the walk 2 - 3 (round 5)
        int tail = 0;
        // (units of fewer than 8 lanes only: at k = 64 a level step inside the walk costs what the launch does)
        // ... and when many of the deep depths are thin, all of them: the subtrees hanging at the first one, a workgroup per
        // (bin of subtrees, column) walking its depths (pml_tree_upload, "thin ends")
        const bool deep = td_fused && !td_small && deep_top_down(ctx) && ctx->deep.first_depth > head;
        if (deep) {
            for (int l = head; l < ctx->deep.first_depth; ++l) {
                const int a = ctx->td_parent_offsets_f[l], b = ctx->td_parent_offsets_f[l + 1];
                PML_TRY(dispatch_sweep(ctx, SW_TD_FUSED, ctx->d_td_parents_f + a, b - a));
                if (b > a) ++n_launch;
            }
            PML_TRY(dispatch_blocks_f81(ctx, false, 1));
            PML_TRY(prof_end(ctx, 1, n_launch + 1));
            return PML_OK;
        }
        if (td_fused && !td_small && ctx->Gt < 8 && !ctx->tune.on(T_NO_TD_TAIL)) {
            tail = narrow_levels(ctx, ctx->td_parent_offsets_f, ctx->n_td_levels, false, ctx->C, 0, 1);
            if (tail > ctx->n_td_levels - head) tail = ctx->n_td_levels - head;
            if (tail < 2) tail = 0;
        }
        for (int l = head; l < (td_small ? 0 : ctx->n_td_levels - tail); ++l) {
            const std::vector<int>& off = td_fused ? ctx->td_parent_offsets_f : ctx->td_parent_offsets;
            const int a = off[l], b = off[l + 1];
            PML_TRY(dispatch_sweep(ctx, td_fused ? SW_TD_FUSED : SW_TD,
                                   (td_fused ? ctx->d_td_parents_f : ctx->d_td_parents) + a, b - a));
            if (b > a) ++n_launch;
        }
        PML_TRY(prof_end(ctx, 1, n_launch));
        if (tail > 0) PML_TRY(dispatch_small_f81(ctx, false, 0, ctx->n_td_levels - tail, tail, nullptr, nullptr, 1));
        return PML_OK;
    };
    if (ctx->graphs && !ctx->profile && !td_small && ctx->n_td_levels >= 4) {
        PML_TRY(run_captured(ctx, ctx->td_graph, enqueue));
    } else {
        PML_TRY(enqueue());
    }
    ctx->td_valid = true;
    ctx->td_vec_valid = td_stored;
    ctx->td_filled = false;
    ctx->post_ever = true;
    ctx->tip_post_missing = state_of(ctx).implicit_tips;
    return PML_OK;
}

// PML_OPT_IMPLICIT_TIP_POSTERIORS: whoever reads the posterior table gets the rows the sweep left implicit first
static int materialize_tip_posteriors(pml_ctx* ctx) {
    if (!ctx->tip_post_missing || !ctx->d_post || ctx->n_tips == 0) return PML_OK;
    dim3 grid(grid_for(ctx, ctx->n_tips, PML_BLOCK, ctx->C), ctx->C);
    hipLaunchKernelGGL(tip_posteriors_kernel, grid, dim3(PML_BLOCK), 0, ctx->stream, cols_of(ctx), state_of(ctx), ctx->N,
                       ctx->d_tips, ctx->n_tips);
    HIP_TRY(hipGetLastError());
    ctx->tip_post_missing = false;
    return PML_OK;
}

static int materialize_cherries(pml_ctx* ctx);

// For inspection (pml_download of the TD buffers): every non-root node gets its top-down vector.  F81 family: the
// sweep did not write its TD vectors, it is repeated with the stores switched on (same arithmetic); then the vectors of
// the nodes no sweep stores (tips; fused cherries) are filled in level by level (td_fill_kernel).
static int materialize_td(pml_ctx* ctx) {
    if (!ctx->td_vec_valid) {
        // one sweep with the stores on; the option itself stays as the caller set it (a pooled ctx must not keep
        // paying for TD stores because somebody once looked at them)
        const bool was = ctx->keep_td;
        drop_graph(ctx->td_graph);
        drop_graph(ctx->mp_graph);
        ctx->keep_td = true;
        const int status = run_top_down(ctx);
        ctx->keep_td = was;
        if (!was) {
            drop_graph(ctx->td_graph);
            drop_graph(ctx->mp_graph);
        }
        PML_TRY(status);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (!ctx->td_filled) {
        const bool f81 = ctx->kind == PML_MODEL_F81;
        if (f81) {
            PML_TRY(materialize_cherries(ctx));  // the cherries' bottom-up vectors
        } else {
            PML_TRY(run_prep(ctx, ctx->d_P == nullptr || eigen_fused(ctx) || eigen_gemm(ctx)));  // P(t) of every branch in HBM
        }
        PmlState st = state_of(ctx);
        st.td = ctx->d_td;  // state_of hides them when the option is off
        st.te = ctx->d_te;
        for (int d = 1; d < ctx->n_td_levels; ++d) {
            const int a = ctx->td_offsets[d], b = ctx->td_offsets[d + 1];
            if (b <= a) continue;
            dim3 grid(grid_for(ctx, b - a, PML_WAVES_PER_BLOCK, ctx->C), ctx->C);
            hipLaunchKernelGGL(td_fill_kernel, grid, dim3(PML_BLOCK), 0, ctx->stream, tree_of(ctx, f81), cols_of(ctx), st,
                               f81 ? nullptr : ctx->d_P, f81 ? 1 : 0, a, b);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->td_filled = true;
    }
    return PML_OK;
}

// copies of the marginal results the caller asked for (stream-ordered behind the sweep), then one synchronisation
static int fetch_marginals(pml_ctx* ctx, double* posterior_out, double* lh_sum_out, double* lh_sf_out) {
    const size_t CN = (size_t)ctx->C * ctx->N;
    if (posterior_out) {
        PML_TRY(materialize_tip_posteriors(ctx));
        PML_TRY(fetch_rows(ctx, ctx->d_post, (size_t)ctx->ks, (size_t)ctx->k, (size_t)ctx->C, posterior_out));
    }
    PML_TRY(fetch_rows(ctx, ctx->d_lhsum, 1, 1, (size_t)ctx->C, lh_sum_out));
    // (the exponents arrive in the output array itself -- same width -- and are converted in place)
    PML_TRY(fetch_rows(ctx, ctx->d_lhe, 1, 1, (size_t)ctx->C, reinterpret_cast<i64*>(lh_sf_out)));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (lh_sf_out) {
        const double l2 = std::log10(2.0);
        const i64* e = reinterpret_cast<const i64*>(lh_sf_out);
        for (size_t i = 0; i < CN; ++i) lh_sf_out[i] = -(double)e[i] * l2;
    }
    return PML_OK;
}

int pml_top_down_marginals(pml_ctx* ctx, double* posterior_out, double* lh_sum_out, double* lh_sf_out) {
    PML_TRY(require_model(ctx));
    if (ctx->bu_mode != 1) return fail(PML_ERR_INVALID, "pml_top_down_marginals needs a successful marginal pml_bottom_up first");
    PML_TRY(run_top_down(ctx));
    return fetch_marginals(ctx, posterior_out, lh_sum_out, lh_sf_out);
}

int pml_marginal_pass(pml_ctx* ctx, double* loglik_out, int32_t* err_parent, int32_t* err_child, double* posterior_out,
                      double* lh_sum_out, double* lh_sf_out) {
    PML_TRY(require_model(ctx));
    if (!loglik_out) return fail(PML_ERR_INVALID, "loglik_out is NULL");
    // both sweeps go on the stream before the host looks at anything: the top-down sweep does not wait for a round trip.
    // Latency-bound forests (the sweeps replay as hipGraphs): ONE graph holds both sweeps -- one launch call, no gap
    // between the last bottom-up kernel and the first top-down one.  (F81 family, once the buffers of a top-down sweep
    // exist: nothing may be allocated while a stream is captured.)
    const bool td_small = single_launch_sweeps(ctx) && ctx->kind == PML_MODEL_F81;
    const int bu_launches = td_small ? 1 : (int)ctx->bu_offsets_f.size() - 1;
    const bool one_graph = ctx->graphs && !ctx->profile && ctx->kind == PML_MODEL_F81 && ctx->d_post != nullptr &&
                           (!ctx->keep_td || ctx->d_td != nullptr) &&
                           (bu_launches >= 4 || (!td_small && ctx->n_td_levels >= 4));
    // Where the pass ends in a multi-level kernel and has few columns, that kernel says when it is done (pml_signal_done)
    // and the wait at the end of this call is a spin on the word it raises (wait_signals) -- as for a bottom-up sweep.
    // The word counts the signalling launches (the bottom-up sweep's last one may be one too).
    if (ctx->wait_signal) {  // (a sweep submitted and never collected: the count below is read on an idle stream)
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->wait_signal = false;
    }
    set_active_columns(ctx, nullptr);  // (the replayed pass does not go through submit_bottom_up)
    const u64 generation_before = ctx->h_done ? *reinterpret_cast<volatile u64*>(ctx->h_done) : 0;
    int n_signals = 0;
    bool final_signals = false;
    if (one_graph) {
        if (ctx->mp_graph.exec && ctx->mp_graph.has_init == ctx->has_init) {
            HIP_TRY(hipGraphLaunch(ctx->mp_graph.exec, ctx->stream));
            if (ctx->mp_graph.has_params) ctx->params_dirty = false;
        } else {
            ctx->signals_enqueued = 0;
            ctx->td_final_signals = false;
            PML_TRY(run_captured(ctx, ctx->mp_graph, [&]() {
                ctx->in_outer_capture = true;
                ctx->mp_wants_signal = true;
                int status = submit_bottom_up(ctx, 1);
                if (status == PML_OK) {
                    ctx->bu_mode = 1;
                    status = run_top_down(ctx);
                }
                ctx->mp_wants_signal = false;
                ctx->in_outer_capture = false;
                return status;
            }));
            ctx->mp_signals = ctx->signals_enqueued;
            ctx->mp_final = ctx->td_final_signals;
        }
        n_signals = ctx->mp_signals;
        final_signals = ctx->mp_final;
        // the bookkeeping of submit_bottom_up / run_top_down (a replay runs neither)
        ctx->js_valid = false;
        ctx->prep_dirty = false;
        ctx->bu_fused = ctx->n_cherries > 0;
        ctx->bu_fused_joint = false;
        ctx->bu_absorbed = super_sweeps(ctx);
        ctx->td_valid = true;
        ctx->td_vec_valid = ctx->keep_td;
        ctx->td_filled = false;
        ctx->post_ever = true;
        ctx->tip_post_missing = state_of(ctx).implicit_tips;
    } else {
        PML_TRY(submit_bottom_up(ctx, 1));
        n_signals = ctx->wait_signal ? 1 : 0;  // (its last launch signals)
        ctx->wait_signal = false;  // (this call waits for the whole pass: fetch_marginals)
        ctx->bu_mode = 1;  // provisional, for run_top_down's bookkeeping; collect_bottom_up has the last word
        ctx->signals_enqueued = 0;
        ctx->td_final_signals = false;
        ctx->mp_wants_signal = td_small;  // (the single launch is enqueued afresh every time; level sweeps may replay a graph)
        const int td_status = run_top_down(ctx);
        ctx->mp_wants_signal = false;
        PML_TRY(td_status);
        n_signals += ctx->signals_enqueued;
        final_signals = ctx->td_final_signals;
    }
    const bool spin = final_signals && ctx->comm == nullptr && !posterior_out && !lh_sum_out && !lh_sf_out &&
                      !ctx->tune.on(T_NO_SPIN_WAIT);
    ctx->done_expect = generation_before + (u64)n_signals;
    // a communicator is attached: the one collective of the path goes on the stream right here, behind the sweeps --
    // the rank's sum formed on the device from the values the sweep left in pinned memory, all-reduced over RCCL, copied
    // back; the single wait of this call (fetch_marginals) covers it.  pml_loglik_total hands the result out.
    if (ctx->comm != nullptr) {
        PmlComm* cm = ctx->comm;
        hipLaunchKernelGGL(sum_loglik_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->h_loglik, ctx->C, cm->d_total);
        HIP_TRY(hipGetLastError());
        if (cm->comm) {
            PmlRccl* r = pml_rccl();
            const ncclResult_t e = r->AllReduce(cm->d_total, cm->d_total, 1, ncclDouble, ncclSum, cm->comm, ctx->stream);
            if (e != ncclSuccess) return fail(PML_ERR_HIP, "ncclAllReduce failed: %s", r->GetErrorString(e));
        }
        HIP_TRY(hipMemcpyAsync(cm->h_total, cm->d_total, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        cm->total_fresh = true;
    }
    int fetched = PML_OK;
    if (spin) {
        ctx->wait_signal = true;
        fetched = wait_bottom_up(ctx);  // (nothing to copy: the spin on the last launch's word, or the stream)
    } else {
        fetched = fetch_marginals(ctx, posterior_out, lh_sum_out, lh_sf_out);  // synchronises
    }
    ctx->bu_mode = -1;
    const int status = collect_bottom_up(ctx, 1, loglik_out, err_parent, err_child);
    if (status != PML_OK) {
        ctx->td_valid = false;  // a column without likelihood: its top-down results mean nothing
        return status;
    }
    return fetched;
}

// the back-trace launches (no host synchronisation): the narrow depths below the roots in one launch, the wide ones one
// launch each
static int submit_joint_backtrace(pml_ctx* ctx) {
    int head = 0;
    {
        std::vector<int> off(ctx->td_offsets.begin() + 1, ctx->td_offsets.end());
        head = narrow_levels(ctx, off, ctx->n_td_levels - 1, true, ctx->C, 1024);
    }
    const pml_ctx::BacktraceTiers& B = ctx->bt_tiers;
    // (tiers while one column's narrow end is the limit they were cut for: with many columns the narrow end is shorter
    // and the depths in between keep their launches)
    bool tiers = B.ok && 1 + head >= B.first_depth;
    for (const pml_ctx::BacktraceTiers::Tier& T : B.tiers)
        tiers = tiers && (long long)T.n_blocks * ctx->C <= 8192;  // (a workgroup per subtree and column: only while few)
    auto enqueue = [&]() -> int {
        if (head > 0) {
            hipLaunchKernelGGL(joint_backtrace_narrow_kernel, dim3(1, ctx->C), dim3(PML_BLOCK), 0, ctx->stream,
                               tree_of(ctx), cols_of(ctx), state_of(ctx), ctx->d_td_offsets, 1,
                               tiers ? B.first_depth - 1 : head);
            HIP_TRY(hipGetLastError());
        }
        if (tiers) {
            for (const pml_ctx::BacktraceTiers::Tier& T : B.tiers) {
                hipLaunchKernelGGL(joint_backtrace_blocks_kernel, dim3(T.n_blocks, ctx->C), dim3(PML_BLOCK), 0, ctx->stream,
                                   tree_of(ctx), cols_of(ctx), state_of(ctx), B.d_nodes, B.d_lv, B.d_start + T.first_block,
                                   T.depth);
                HIP_TRY(hipGetLastError());
            }
            return PML_OK;
        }
        for (int l = 1 + head; l < ctx->n_td_levels; ++l) {
            const int a = ctx->td_offsets[l], b = ctx->td_offsets[l + 1];
            if (b <= a) continue;
            dim3 grid(grid_for(ctx, b - a, PML_BLOCK, ctx->C), ctx->C);
            hipLaunchKernelGGL(joint_backtrace_kernel, grid, dim3(PML_BLOCK), 0, ctx->stream, tree_of(ctx), cols_of(ctx),
                               state_of(ctx), a, b);
            HIP_TRY(hipGetLastError());
        }
        return PML_OK;
    };
    if (ctx->graphs && !ctx->profile && ctx->n_td_levels - head >= 4) {
        PML_TRY(run_captured(ctx, ctx->bt_graph, enqueue));
    } else {
        PML_TRY(enqueue());
    }
    return PML_OK;
}

static int fetch_joint_states(pml_ctx* ctx, int32_t* joint_state_out) {
    ctx->js_valid = true;
    ctx->js_ever = true;
    PML_TRY(fetch_rows(ctx, ctx->d_js, 1, 1, (size_t)ctx->C, joint_state_out));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PML_OK;
}

int pml_joint_backtrace(pml_ctx* ctx, int32_t* joint_state_out) {
    PML_TRY(require_model(ctx));
    if (ctx->bu_mode != 0) return fail(PML_ERR_INVALID, "pml_joint_backtrace needs a successful joint pml_bottom_up first");
    PML_TRY(submit_joint_backtrace(ctx));
    return fetch_joint_states(ctx, joint_state_out);
}

int pml_joint_pass(pml_ctx* ctx, double* loglik_out, int32_t* err_parent, int32_t* err_child, int32_t* joint_state_out) {
    PML_TRY(require_model(ctx));
    if (!loglik_out) return fail(PML_ERR_INVALID, "loglik_out is NULL");
    // joint sweep and back-trace submitted together: one host round trip
    PML_TRY(submit_bottom_up(ctx, 0));
    ctx->wait_signal = false;  // (fetch_joint_states waits for both)
    PML_TRY(submit_joint_backtrace(ctx));
    const int fetched = fetch_joint_states(ctx, joint_state_out);  // synchronises
    const int status = collect_bottom_up(ctx, 0, loglik_out, err_parent, err_child);
    if (status != PML_OK) {
        ctx->js_valid = false;
        return status;
    }
    return fetched;
}

// altered (caller's ids, or null): see counts_level_kernel.  result_out: the k x k sums of the draws (not divided by
// n_repetitions when altered is given); state_counts_out / same_out: [N][k] in the caller's numbering.
static int marginal_counts_impl(pml_ctx* ctx, int32_t col, int32_t n_repetitions, uint64_t seed, const uint8_t* altered,
                                double* result_out, int32_t* state_counts_out, int32_t* same_out) {
    PML_TRY(require_model(ctx));
    if (col < 0 || col >= ctx->C || !result_out) return fail(PML_ERR_INVALID, "bad column / output");
    if (n_repetitions <= 0) return fail(PML_ERR_INVALID, "n_repetitions must be positive");
    if (ctx->bu_mode != 1 || !ctx->td_valid)
        return fail(PML_ERR_INVALID, "pml_marginal_counts needs a marginal pml_bottom_up and pml_top_down_marginals first");
    if (ctx->k > PML_COUNTS_MAX_K) return fail(PML_ERR_UNSUPPORTED, "k = %d: at most %d states", ctx->k, PML_COUNTS_MAX_K);
    PML_TRY(materialize_cherries(ctx));  // the conditional probabilities need every bottom-up vector
    PML_TRY(materialize_tip_posteriors(ctx));
    PML_TRY(run_prep(ctx));  // P(t) of every branch (the fused eigen sweeps never materialise it) / exp(-mu t')
    const size_t k = ctx->k, N = (size_t)ctx->N;
    int *d_counts = nullptr, *d_same = nullptr;
    long long* d_result = nullptr;
    unsigned char* d_alt = nullptr;
    hipError_t e = hipMalloc((void**)&d_counts, N * k * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&d_result, k * k * sizeof(long long));
    if (e == hipSuccess) e = hipMemsetAsync(d_result, 0, k * k * sizeof(long long), ctx->stream);
    std::vector<unsigned char> alt;
    if (e == hipSuccess && altered != nullptr) {
        alt.assign(N, 0);
        for (size_t i = 0; i < N; ++i) alt[(size_t)internal_id(ctx, (int)i)] = altered[i] ? 1 : 0;
        e = hipMalloc((void**)&d_alt, N);
        if (e == hipSuccess) e = hipMalloc((void**)&d_same, N * k * sizeof(int));
        if (e == hipSuccess) e = hipMemcpyAsync(d_alt, alt.data(), N, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_same, 0, N * k * sizeof(int), ctx->stream);
    }
    if (e == hipSuccess) {
        const PmlTree t = tree_of(ctx);
        const PmlCols c = cols_of(ctx);
        const PmlState st = state_of(ctx);
        const PmlModel m = model_of(ctx);
        const double* P = ctx->kind == PML_MODEL_F81 ? nullptr : ctx->d_P;
        // (the draws are keyed by the CALLER's node ids: the library's internal numbering must not show in the result)
        hipLaunchKernelGGL(counts_roots_kernel, dim3(std::min(ctx->n_roots, 1024)), dim3(64), 0, ctx->stream, t, c, st, col,
                           n_repetitions, seed, d_counts, ctx->d_old_of_new);
        for (int l = 0; l < ctx->n_td_levels; ++l) {
            const int a = ctx->td_parent_offsets[l], b = ctx->td_parent_offsets[l + 1];
            if (b <= a) continue;
            hipLaunchKernelGGL(counts_level_kernel, dim3(std::min(b - a, 65536)), dim3(64), 0, ctx->stream, t, c, st, m, P,
                               col, n_repetitions, seed, ctx->d_td_parents + a, b - a, d_counts, d_result, ctx->d_old_of_new,
                               d_alt, d_same);
        }
        e = hipGetLastError();
    }
    std::vector<long long> h(k * k);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_result, k * k * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream);
    int fetched = PML_OK;
    if (e == hipSuccess && altered != nullptr) {
        // (tips' rows of d_counts are written by their parents' passes: every node has its row)
        if (state_counts_out) fetched = fetch_rows(ctx, d_counts, k, k, 1, state_counts_out);
        if (fetched == PML_OK && same_out) fetched = fetch_rows(ctx, d_same, k, k, 1, same_out);
    }
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_counts);
    if (d_result) (void)hipFree(d_result);
    if (d_alt) (void)hipFree(d_alt);
    if (d_same) (void)hipFree(d_same);
    if (e != hipSuccess) return fail(PML_ERR_HIP, "pml_marginal_counts failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(PML_ERR_HIP, "pml_marginal_counts failed: %s", hipGetErrorString(e2));
    PML_TRY(fetched);
    for (size_t i = 0; i < k * k; ++i) result_out[i] = altered != nullptr ? (double)h[i] : (double)h[i] / (double)n_repetitions;
    return PML_OK;
}

int pml_marginal_counts(pml_ctx* ctx, int32_t col, int32_t n_repetitions, uint64_t seed, double* counts_out) {
    return marginal_counts_impl(ctx, col, n_repetitions, seed, nullptr, counts_out, nullptr, nullptr);
}

int pml_marginal_counts_altered(pml_ctx* ctx, int32_t col, int32_t n_repetitions, uint64_t seed, const uint8_t* altered,
                                double* sums_out, int32_t* state_counts_out, int32_t* same_out) {
    if (!altered || !state_counts_out || !same_out) return fail(PML_ERR_INVALID, "NULL array");
    return marginal_counts_impl(ctx, col, n_repetitions, seed, altered, sums_out, state_counts_out, same_out);
}

int pml_select_states(pml_ctx* ctx, int method, int force_joint, const uint64_t* lh_mask, uint64_t* masks_out,
                      int32_t* n_states_out) {
    PML_TRY(require_model(ctx));
    if (!ctx->post_ever) return fail(PML_ERR_INVALID, "pml_select_states needs pml_top_down_marginals first");
    if (method != 0 && method != 1) return fail(PML_ERR_INVALID, "method must be 0 (MAP) or 1 (MPPA)");
    if (method == 1 && force_joint && !ctx->js_ever)
        return fail(PML_ERR_INVALID, "force_joint needs the joint states of a pml_joint_backtrace");
    PML_TRY(materialize_tip_posteriors(ctx));  // the selection reads every row, and rewrites the masks
    note_tips_observed(ctx, 0, ctx->C, false);  // (masks from now on: whatever was selected)
    const size_t CN = (size_t)ctx->C * ctx->N;
    if (!ctx->d_nsel) PML_TRY(dev_alloc(ctx, &ctx->d_nsel, CN));
    u64* d_lh_mask = nullptr;
    std::vector<u64> lh_mask_own;   // (the caller's rows in the library's numbering, alive until the call has waited for its copies)
    if (lh_mask) {
        if (ctx->k % 64) {
            const u64 valid = (1ull << (ctx->k % 64)) - 1ull;
            for (size_t i = ctx->W - 1; i < CN * ctx->W; i += ctx->W)
                if (lh_mask[i] & ~valid) return fail(PML_ERR_INVALID, "lh_mask word %zu has bits beyond k", i);
        }
        HIP_TRY(hipMalloc((void**)&d_lh_mask, CN * ctx->W * sizeof(u64)));
        if (permuted(ctx)) {
            lh_mask_own.resize(CN * ctx->W);
            rows_to_internal(ctx, (const u64*)lh_mask, lh_mask_own.data(), (size_t)ctx->W, (size_t)ctx->C);
            lh_mask = (const uint64_t*)lh_mask_own.data();
        }
        hipError_t e = hipMemcpyAsync(d_lh_mask, lh_mask, CN * ctx->W * sizeof(u64), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(d_lh_mask);
            return fail(PML_ERR_HIP, "lh_mask upload failed: %s", hipGetErrorString(e));
        }
    }
    const int status = dispatch_select(ctx, method, force_joint, d_lh_mask);   // pml_launch_matrix.hip
    hipError_t e = hipGetLastError();
    int fetched = PML_OK;
    if (e == hipSuccess && status == PML_OK) {
        fetched = fetch_rows(ctx, (const u64*)ctx->d_masks, (size_t)ctx->W, (size_t)ctx->W, (size_t)ctx->C, (u64*)masks_out);
        if (fetched == PML_OK) fetched = fetch_rows(ctx, ctx->d_nsel, 1, 1, (size_t)ctx->C, n_states_out);
    }
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (d_lh_mask) (void)hipFree(d_lh_mask);
    if (status != PML_OK) return status;
    if (e != hipSuccess) return fail(PML_ERR_HIP, "pml_select_states failed: %s", hipGetErrorString(e));
    if (fetched != PML_OK) return fetched;
    if (e2 != hipSuccess) return fail(PML_ERR_HIP, "pml_select_states failed: %s", hipGetErrorString(e2));
    // the columns' masks changed: sweeps must be redone, the posteriors themselves stay valid for inspection
    ctx->prep_dirty = true;
    ctx->bu_mode = -1;
    return PML_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
static int fetch_vectors(pml_ctx* ctx, const double* src, int col, double* out) {
    const size_t N = ctx->N;
    if (ctx->ks == ctx->k) {
        HIP_TRY(hipMemcpyAsync(out, src + (size_t)col * N * ctx->ks, N * ctx->k * sizeof(double), hipMemcpyDeviceToHost,
                               ctx->stream));
    } else {
        HIP_TRY(hipMemcpy2DAsync(out, ctx->k * sizeof(double), src + (size_t)col * N * ctx->ks, ctx->ks * sizeof(double),
                                 ctx->k * sizeof(double), N, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PML_OK;
}

static int fetch_exponents(pml_ctx* ctx, const i64* src, int col, double* out) {
    std::vector<i64> tmp(ctx->N);
    HIP_TRY(hipMemcpyAsync(tmp.data(), src + (size_t)col * ctx->N, ctx->N * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const double l2 = std::log10(2.0);
    for (int i = 0; i < ctx->N; ++i) out[i] = -(double)tmp[i] * l2;
    return PML_OK;
}

// after a fused sweep the cherries' bottom-up vectors only ever existed in registers: compute them for inspection
static int materialize_cherries(pml_ctx* ctx) {
    if (ctx->bu_absorbed) {
        // the children of the two-level units: their own units (two cherries of two tips), from the tips
        ctx->units_override = ctx->sup.d_child_units;
        const int status = dispatch_sweep(ctx, SW_BU_MARG_FUSED_NOVEC, ctx->d_bu_order_f, ctx->sup.n_child_units);  // (nothing if 0)
        ctx->units_override = nullptr;
        PML_TRY(status);
        // ... and the children of the stacked units (two stored children each; in chunks of at most 65 536 units: the
        // lane shape, hence the rounding of pi . v, of the levels they were taken from)
        for (int a = 0; a < 2 * ctx->sup.n_stack; a += 65536) {
            ctx->units_override = ctx->sup.d_stack_children + a;
            const int st2 = dispatch_sweep(ctx, SW_BU_MARG_FUSED, ctx->d_bu_order_f, std::min(65536, 2 * ctx->sup.n_stack - a));
            ctx->units_override = nullptr;
            PML_TRY(st2);
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->bu_absorbed = false;
    }
    if (!ctx->bu_fused) return PML_OK;
    PML_TRY(dispatch_sweep(ctx, ctx->bu_fused_joint ? SW_BU_CHERRIES_JOINT : SW_BU_CHERRIES, ctx->d_cherries,
                           ctx->n_cherries));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->bu_fused = false;
    return PML_OK;
}

static int download_internal(pml_ctx* ctx, int what, int32_t col, void* out);

int pml_download(pml_ctx* ctx, int what, int32_t col, void* out) {
    PML_TRY(download_internal(ctx, what, col, out));   // rows in the library's numbering
    if (permuted(ctx)) {
        switch (what) {
            case PML_BUF_BU:
            case PML_BUF_TD:
            case PML_BUF_POSTERIOR:
                rows_to_api_inplace(ctx, (double*)out, (size_t)ctx->k, 1);
                break;
            case PML_BUF_JOINT_TABLE:
                rows_to_api_inplace(ctx, (int32_t*)out, (size_t)ctx->k, 1);
                break;
            case PML_BUF_JOINT_STATE:
                rows_to_api_inplace(ctx, (int32_t*)out, 1, 1);
                break;
            default:
                rows_to_api_inplace(ctx, (double*)out, 1, 1);
                break;
        }
    }
    return PML_OK;
}

static int download_internal(pml_ctx* ctx, int what, int32_t col, void* out) {
    PML_TRY(require_model(ctx));
    if (col < 0 || col >= ctx->C || !out) return fail(PML_ERR_INVALID, "bad column / output");
    // (a pass that ended in a spin on the completion word may have left the stream busy, and the blocking copies below run
    // on the NULL stream, which a non-blocking stream is not ordered with)
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->wait_signal = false;
    if (what == PML_BUF_BU || what == PML_BUF_BU_SF) PML_TRY(materialize_cherries(ctx));
    const size_t N = ctx->N;
    const double nan = std::numeric_limits<double>::quiet_NaN();
    switch (what) {
        case PML_BUF_BU: {
            if (ctx->bu_mode < 0) return fail(PML_ERR_INVALID, "no valid bottom-up sweep");
            double* o = (double*)out;
            PML_TRY(fetch_vectors(ctx, ctx->d_bu, col, o));
            std::vector<u64> m(N * ctx->W);
            HIP_TRY(hipMemcpy(m.data(), ctx->d_masks + (size_t)col * N * ctx->W, m.size() * sizeof(u64), hipMemcpyDeviceToHost));
            for (size_t n = 0; n < N; ++n)
                if (ctx->h_n_children[n] == 0)
                    for (int s = 0; s < ctx->k; ++s) o[n * ctx->k + s] = (double)((m[n * ctx->W + (s >> 6)] >> (s & 63)) & 1ull);
            return PML_OK;
        }
        case PML_BUF_BU_SF: {
            if (ctx->bu_mode < 0) return fail(PML_ERR_INVALID, "no valid bottom-up sweep");
            PML_TRY(fetch_exponents(ctx, ctx->d_be, col, (double*)out));
            for (size_t n = 0; n < N; ++n)
                if (ctx->h_n_children[n] == 0) ((double*)out)[n] = 0.0;
            return PML_OK;
        }
        case PML_BUF_TD:
            if (!ctx->td_valid) return fail(PML_ERR_INVALID, "no valid top-down sweep");
            PML_TRY(materialize_td(ctx));
            return fetch_vectors(ctx, ctx->d_td, col, (double*)out);
        case PML_BUF_TD_SF:
            if (!ctx->td_valid) return fail(PML_ERR_INVALID, "no valid top-down sweep");
            PML_TRY(materialize_td(ctx));
            return fetch_exponents(ctx, ctx->d_te, col, (double*)out);
        case PML_BUF_POSTERIOR:
            if (!ctx->td_valid) return fail(PML_ERR_INVALID, "no valid top-down sweep");
            PML_TRY(materialize_tip_posteriors(ctx));
            return fetch_vectors(ctx, ctx->d_post, col, (double*)out);
        case PML_BUF_LH_SUM:
            if (!ctx->td_valid) return fail(PML_ERR_INVALID, "no valid top-down sweep");
            HIP_TRY(hipMemcpy(out, ctx->d_lhsum + (size_t)col * N, N * sizeof(double), hipMemcpyDeviceToHost));
            return PML_OK;
        case PML_BUF_LH_SF:
            if (!ctx->td_valid) return fail(PML_ERR_INVALID, "no valid top-down sweep");
            return fetch_exponents(ctx, ctx->d_lhe, col, (double*)out);
        case PML_BUF_JOINT_TABLE: {
            if (ctx->bu_mode != 0) return fail(PML_ERR_INVALID, "no valid joint sweep");
            // the tables hold one byte per entry on the device (two beyond 256 states); the interface hands out int32
            const size_t width = ctx->k > 256 ? 2 : 1;
            std::vector<pml_jt> tmp(N * ctx->ks * width);
            HIP_TRY(hipMemcpy(tmp.data(), ctx->d_J + (size_t)col * N * ctx->ks * width, tmp.size(), hipMemcpyDeviceToHost));
            int32_t* o = (int32_t*)out;
            const unsigned short* wide = reinterpret_cast<const unsigned short*>(tmp.data());
            for (size_t n = 0; n < N; ++n)
                for (int i = 0; i < ctx->k; ++i) o[n * ctx->k + i] = width == 2 ? (int32_t)wide[n * ctx->ks + i] : (int32_t)tmp[n * ctx->ks + i];
            return PML_OK;
        }
        case PML_BUF_JOINT_STATE:
            if (!ctx->js_valid) return fail(PML_ERR_INVALID, "no valid joint back-trace");
            HIP_TRY(hipMemcpy(out, ctx->d_js + (size_t)col * N, N * sizeof(int), hipMemcpyDeviceToHost));
            return PML_OK;
        case PML_BUF_BRANCH_EXP:
            if (ctx->kind != PML_MODEL_F81) return fail(PML_ERR_INVALID, "branch exponentials exist for the F81 family only");
            PML_TRY(run_prep(ctx));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipMemcpy(out, ctx->d_E + (size_t)col * N, N * sizeof(double), hipMemcpyDeviceToHost));
            return PML_OK;
        default:
            return fail(PML_ERR_INVALID, "unknown buffer id %d", what);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// multi-GPU: one collective, the sum of the per-rank log-likelihoods
int pml_comm_unique_id(unsigned char* id_out) {
    if (!id_out) return fail(PML_ERR_INVALID, "id_out is NULL");
    PmlRccl* r = pml_rccl();
    if (!r->handle) return fail(PML_ERR_UNSUPPORTED, "librccl could not be loaded: %s", r->error.c_str());
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return fail(PML_ERR_HIP, "ncclGetUniqueId failed: %s", r->GetErrorString(e));
    static_assert(sizeof(id.internal) == PML_COMM_ID_BYTES, "unique id size");
    memcpy(id_out, id.internal, PML_COMM_ID_BYTES);
    return PML_OK;
}

int pml_comm_init(pml_ctx* ctx, int rank, int world, const unsigned char* id) {
    if (!ctx) return fail(PML_ERR_INVALID, "ctx is NULL");
    if (world < 1 || rank < 0 || rank >= world) return fail(PML_ERR_INVALID, "rank %d out of 0..%d", rank, world - 1);
    if (ctx->comm) return fail(PML_ERR_INVALID, "the ctx already has a communicator");
    HIP_TRY(hipSetDevice(ctx->device));
    PmlComm* c = new PmlComm();
    c->rank = rank;
    c->world = world;
    // PASTML_HIP_COMM_FORCE_RCCL: a world of one still goes through librccl (exercises the whole path on one GPU)
    if (world > 1 || getenv("PASTML_HIP_COMM_FORCE_RCCL")) {
        if (!id) {
            delete c;
            return fail(PML_ERR_INVALID, "id is NULL");
        }
        PmlRccl* r = pml_rccl();
        if (!r->handle) {
            delete c;
            return fail(PML_ERR_UNSUPPORTED, "librccl could not be loaded: %s", r->error.c_str());
        }
        ncclUniqueId uid;
        memcpy(uid.internal, id, PML_COMM_ID_BYTES);
        const ncclResult_t e = r->CommInitRank(&c->comm, world, uid, rank);
        if (e != ncclSuccess) {
            delete c;
            return fail(PML_ERR_HIP, "ncclCommInitRank failed: %s", r->GetErrorString(e));
        }
    }
    if (hipMalloc((void**)&c->d_total, sizeof(double)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_total, sizeof(double)) != hipSuccess) {
        if (c->comm) (void)pml_rccl()->CommDestroy(c->comm);
        if (c->d_total) (void)hipFree(c->d_total);
        delete c;
        return fail(PML_ERR_HIP, "allocation of the communicator's buffers failed");
    }
    ctx->comm = c;
    return PML_OK;
}

int pml_comm_info(pml_ctx* ctx, int32_t* rank, int32_t* world, int32_t* backend, int32_t* rccl_ranks) {
    if (!ctx || !ctx->comm) return fail(PML_ERR_INVALID, "no communicator: call pml_comm_init first");
    const PmlComm* c = ctx->comm;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (backend) *backend = c->comm ? 1 : 0;
    if (rccl_ranks) {
        *rccl_ranks = 0;
        if (c->comm) {
            PmlRccl* r = pml_rccl();
            int n = -1;
            if (r->CommCount) {
                const ncclResult_t e = r->CommCount(c->comm, &n);
                if (e != ncclSuccess) return fail(PML_ERR_HIP, "ncclCommCount failed: %s", r->GetErrorString(e));
            }
            *rccl_ranks = n;
        }
    }
    return PML_OK;
}

int pml_device_uuid(int device, char* uuid_out) {
    if (!uuid_out) return fail(PML_ERR_INVALID, "uuid_out is NULL");
    hipUUID id;
    HIP_TRY(hipDeviceGetUuid(&id, device));
    static const char hex[] = "0123456789abcdef";
    for (int i = 0; i < 16; ++i) {
        const unsigned char b = (unsigned char)id.bytes[i];
        uuid_out[2 * i] = hex[b >> 4];
        uuid_out[2 * i + 1] = hex[b & 15];
    }
    uuid_out[32] = 0;
    return PML_OK;
}

int pml_comm_destroy(pml_ctx* ctx) {
    if (!ctx || !ctx->comm) return PML_OK;
    PmlComm* c = ctx->comm;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (c->comm) (void)pml_rccl()->CommDestroy(c->comm);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->h_buf) (void)hipHostFree(c->h_buf);
    if (c->d_total) (void)hipFree(c->d_total);
    if (c->h_total) (void)hipHostFree(c->h_total);
    delete c;
    ctx->comm = nullptr;
    return PML_OK;
}

int pml_comm_allreduce(pml_ctx* ctx, const double* in, double* out, int32_t count, int op) {
    if (!ctx || !ctx->comm) return fail(PML_ERR_INVALID, "no communicator: call pml_comm_init first");
    if (!in || !out || count <= 0) return fail(PML_ERR_INVALID, "bad in / out / count");
    if (op != PML_COMM_SUM && op != PML_COMM_MAX) return fail(PML_ERR_INVALID, "op must be PML_COMM_SUM or PML_COMM_MAX");
    PmlComm* c = ctx->comm;
    if (!c->comm) {
        if (out != in) memmove(out, in, sizeof(double) * count);
        return PML_OK;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    if (c->cap < (size_t)count) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (c->d_buf) (void)hipFree(c->d_buf);
        if (c->h_buf) (void)hipHostFree(c->h_buf);
        c->d_buf = c->h_buf = nullptr;
        c->cap = 0;
        const size_t cap = std::max<size_t>(64, (size_t)count);
        HIP_TRY(hipMalloc((void**)&c->d_buf, cap * sizeof(double)));
        HIP_TRY(hipHostMalloc((void**)&c->h_buf, cap * sizeof(double)));
        c->cap = cap;
    }
    memcpy(c->h_buf, in, sizeof(double) * count);
    HIP_TRY(hipMemcpyAsync(c->d_buf, c->h_buf, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
    PmlRccl* r = pml_rccl();
    const ncclResult_t e = r->AllReduce(c->d_buf, c->d_buf, (size_t)count, ncclDouble, op == PML_COMM_SUM ? ncclSum : ncclMax,
                                        c->comm, ctx->stream);
    if (e != ncclSuccess) return fail(PML_ERR_HIP, "ncclAllReduce failed: %s", r->GetErrorString(e));
    HIP_TRY(hipMemcpyAsync(c->h_buf, c->d_buf, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(out, c->h_buf, sizeof(double) * count);
    return PML_OK;
}

int pml_allreduce_loglik(pml_ctx* ctx, const double* loglik, int32_t n_cols, double* total_out) {
    if (!loglik || !total_out || n_cols <= 0) return fail(PML_ERR_INVALID, "bad loglik / total_out / n_cols");
    // the local sum in column order (as pastml/acr.py would add the characters' results up), then one 8-byte all-reduce
    double local = 0.0;
    for (int i = 0; i < n_cols; ++i) local += loglik[i];
    return pml_comm_allreduce(ctx, &local, total_out, 1, PML_COMM_SUM);
}

int pml_loglik_total(pml_ctx* ctx, double* total_out) {
    if (!ctx || !ctx->comm) return fail(PML_ERR_INVALID, "no communicator: call pml_comm_init first");
    if (!total_out) return fail(PML_ERR_INVALID, "total_out is NULL");
    if (!ctx->comm->total_fresh) return fail(PML_ERR_INVALID, "no pml_marginal_pass since the last pml_loglik_total");
    *total_out = ctx->comm->h_total[0];  // (pml_marginal_pass has waited for the stream)
    ctx->comm->total_fresh = false;
    return PML_OK;
}

// numpy's pairwise summation of a contiguous run (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum: what
// ndarray.sum() does along a contiguous axis): the frequencies must come out with numpy's bits, because the decoded
// points are what the reference-side arithmetic (models/__init__.py:328-330) would hand to the sweeps
static double np_pairwise_sum(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

int pml_host_f81_fd_points(int32_t n, int32_t k, const double* x, const double* lower, const double* upper, int32_t opt_sf,
                           int32_t opt_tau, int32_t free_pi, double sf_fixed, double tau_fixed, const double* pi_fixed,
                           double forest_length, double num_nodes, double* pi_out, double* sf_out, double* tau_out,
                           double* tf_out, double* steps_out, double step) {
    if (n < 0 || k < 1 || !x || !lower || !upper || !pi_out || !sf_out || !tau_out || !tf_out || (n > 0 && !steps_out))
        return fail(PML_ERR_INVALID, "NULL array / bad sizes");
    if (n != (opt_sf ? 1 : 0) + (opt_tau ? 1 : 0) + (free_pi ? k - 1 : 0))
        return fail(PML_ERR_INVALID, "n = %d does not match the parameter layout", n);
    if (!free_pi && !pi_fixed) return fail(PML_ERR_INVALID, "pi_fixed is NULL");
    if (!(step > 0.0)) return fail(PML_ERR_INVALID, "step must be positive");
    const double h = step;
    for (int i = 0; i < n; ++i) {
        const volatile double moved = x[i] + h;   // (scipy: the step as the floating-point numbers see it)
        const double step = moved - x[i];
        if (step == 0.0 || moved < lower[i] || moved > upper[i] || !(x[i] >= lower[i] && x[i] <= upper[i]))
            return PML_ERR_UNSUPPORTED;
    }
    std::vector<double> ratios((size_t)k);
    for (int row = 0; row <= n; ++row) {
        const int moved = row - 1;   // row 0 is x itself
        auto at = [&](int i) { return i == moved ? x[i] + h : x[i]; };
        int pos = 0;
        const double sf = opt_sf ? at(pos++) : sf_fixed;
        const double tau = opt_tau ? at(pos++) : tau_fixed;
        sf_out[row] = sf;
        tau_out[row] = tau;
        tf_out[row] = tau != 0.0 ? forest_length / (forest_length + tau * (num_nodes - 1.0)) : 1.0;
        double* pi = pi_out + (size_t)row * k;
        if (free_pi) {
            for (int j = 0; j < k - 1; ++j) ratios[j] = at(pos + j);
            ratios[k - 1] = 1.0;
            const double total = np_pairwise_sum(ratios.data(), k);
            for (int j = 0; j < k; ++j) pi[j] = ratios[j] / total;
        } else {
            memcpy(pi, pi_fixed, sizeof(double) * k);
        }
        if (moved >= 0) steps_out[moved] = (x[moved] + h) - x[moved];
    }
    return PML_OK;
}

int pml_device_sync(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return PML_OK;
}

int pml_download_strided(pml_ctx* ctx, int what, int32_t col, int32_t first, int32_t stride, int32_t count, void* out) {
    PML_TRY(require_model(ctx));
    if (col < 0 || col >= ctx->C || !out) return fail(PML_ERR_INVALID, "bad column / output");
    if (first < 0 || stride < 1 || count < 1 || (long long)first + (long long)(count - 1) * stride >= ctx->N)
        return fail(PML_ERR_INVALID, "rows first=%d stride=%d count=%d leave 0..%d", first, stride, count, ctx->N - 1);
    const size_t N = ctx->N;
    const void* src = nullptr;
    size_t row_bytes = 0, src_row_bytes = 0;
    switch (what) {
        case PML_BUF_POSTERIOR:
            if (!ctx->td_valid) return fail(PML_ERR_INVALID, "no valid top-down sweep");
            PML_TRY(materialize_tip_posteriors(ctx));
            src = ctx->d_post + ((size_t)col * N + first) * ctx->ks;
            row_bytes = ctx->k * sizeof(double);
            src_row_bytes = ctx->ks * sizeof(double);
            break;
        case PML_BUF_LH_SUM:
            if (!ctx->td_valid) return fail(PML_ERR_INVALID, "no valid top-down sweep");
            src = ctx->d_lhsum + (size_t)col * N + first;
            row_bytes = src_row_bytes = sizeof(double);
            break;
        case PML_BUF_LH_SF:
            if (!ctx->td_valid) return fail(PML_ERR_INVALID, "no valid top-down sweep");
            src = ctx->d_lhe + (size_t)col * N + first;
            row_bytes = src_row_bytes = sizeof(i64);
            break;
        case PML_BUF_JOINT_STATE:
            if (!ctx->js_valid) return fail(PML_ERR_INVALID, "no valid joint back-trace");
            src = ctx->d_js + (size_t)col * N + first;
            row_bytes = src_row_bytes = sizeof(int);
            break;
        default:
            return fail(PML_ERR_INVALID, "pml_download_strided serves PML_BUF_POSTERIOR, _LH_SUM, _LH_SF, _JOINT_STATE");
    }
    if (permuted(ctx)) {
        // the rows asked for are scattered in the library's numbering: gathered on the device, copied once (round 5 issued one
        // small copy per row, ~10 us each)
        const char* col0 = (const char*)src - (size_t)first * src_row_bytes;   // (src points at row `first`: back to the column's row 0)
        const size_t bytes = (size_t)count * row_bytes;
        if (ctx->stage_bytes < bytes) {
            if (ctx->d_stage) (void)hipFree(ctx->d_stage);
            ctx->d_stage = nullptr;
            ctx->stage_bytes = 0;
            HIP_TRY(hipMalloc(&ctx->d_stage, bytes));
            ctx->stage_bytes = bytes;
        }
        const int wd = (int)(row_bytes / 4), ws = (int)(src_row_bytes / 4);
        const long long total = (long long)count * wd;
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)std::min<long long>((total + PML_BLOCK - 1) / PML_BLOCK, 65536)),
                           dim3(PML_BLOCK), 0, ctx->stream, (const unsigned*)col0, (unsigned*)ctx->d_stage, ctx->d_new_of_old,
                           (long long)count, wd, ws, first, stride);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, ctx->d_stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        HIP_TRY(hipMemcpy2DAsync(out, row_bytes, src, src_row_bytes * stride, row_bytes, count, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (what == PML_BUF_LH_SF) {  // base-2 exponents -> the reference's base-10 scale, in place (same width)
        const double l2 = std::log10(2.0);
        i64* e = (i64*)out;
        double* o = (double*)out;
        for (int i = 0; i < count; ++i) o[i] = -(double)e[i] * l2;
    }
    return PML_OK;
}

int pml_profile_enable(pml_ctx* ctx, int on) {
    if (!ctx) return fail(PML_ERR_INVALID, "ctx is NULL");
    ctx->profile = on != 0;
    return PML_OK;
}

int pml_profile_read(pml_ctx* ctx, int which, double* total_ms, int64_t* launches, int reset) {
    if (!ctx || which < 0 || which > 4) return fail(PML_ERR_INVALID, "bad profile slot");
    PML_TRY(prof_drain(ctx));
    if (total_ms) *total_ms = ctx->prof_ms[which];
    if (launches) *launches = ctx->prof_launches[which];
    if (reset) {
        ctx->prof_ms[which] = 0;
        ctx->prof_launches[which] = 0;
    }
    return PML_OK;
}

int pml_timer_start(pml_ctx* ctx) {
    if (!ctx) return fail(PML_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    return PML_OK;
}

int pml_timer_stop(pml_ctx* ctx, float* milliseconds) {
    if (!ctx || !milliseconds) return fail(PML_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(milliseconds, ctx->ev0, ctx->ev1));
    return PML_OK;
}

}  // extern "C"

