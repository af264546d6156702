// F81-family sweeps (F81 / JC / EFT): P(t) = (1 - e) 1 pi^T + e I with e = exp(-mu t') is never materialised;
// P v = (1 - e)(pi . v) 1 + e v costs O(k) per branch (pastml/models/F81Model.py:28-46 in closed form).
//
// Cherry fusion (marginal sweeps): an internal node whose children are all tips ("cherry", kind 1) is never written
// to HBM.  Its bottom-up vector is a product of closed-form tip messages (masks + two scalars per tip), so whoever
// needs it -- its parent in the bottom-up sweep, its parent again in the top-down sweep -- recomputes it in
// registers; its own top-down vector only lives in registers while its tips are finished.  On a balanced tree half of
// the internal nodes are cherries: the bottom-up traffic halves and the top-down traffic drops by ~40 %.
//
// Observed tips (exactly one allowed state s) are finished in closed form in the top-down sweep: their posterior is
// the unit vector and their marginal likelihood needs only pi . (TD o BU) of the parent (one reduction per parent,
// shared by all its observed tips) and the parent's entry at s -- no per-tip reduction.
#pragma once
#include "pml_device.h"

#define PML_KIND_TIP 0
#define PML_KIND_CHERRY 1
#define PML_KIND_STORED 2

typedef unsigned char pml_jt;  // entry of an arg-max table: a state index (k <= 256; beyond, the F81 family stores 16-bit entries
                               // in the same buffer: PmlState::jt16)

struct PmlTree {
    int N;
    int n_roots;
    const int* parent;
    const int* first_child;
    const int* n_children;
    const double* dist;
    const int* post_rank;
    const unsigned char* kind;  // per node: tip / cherry (recomputed, never stored) / stored internal; may be null
};

struct PmlCols {
    int k, ks, W;
    int no_wide_lean;       // (a test switch: round 5's lean units -- several mask words, polytomies -- off: the sequential path)
    const u64* masks;       // [C][N][W]
    const u64* masks_init;  // [C][N][W] or nullptr
    const double* pi;       // [C][ks]
    // [C]: 0.0 = the bottom-up sweep leaves this column alone -- its parameters and masks are those of the last sweep
    // that computed it, and so are its results (pml_bottom_up_submit_columns: the optimisers of most characters of a
    // group are done long before the last one).  The F81 bottom-up kernels with a workgroup per column check it.
    const double* active;
};
__device__ __forceinline__ bool column_active(const PmlCols& c, int col) { return c.active == nullptr || c.active[col] != 0.0; }

struct PmlState {
    double* E;      // [C][N]      F81: exp(-mu t') per branch
    double* bu;     // [C][N][ks]  bottom-up vectors (stored internal nodes only; tips are their masks)
    double* S;      // [C][N]      F81 marginal: pi . bu (tips and stored internal nodes)
    i64* be;        // [C][N]      base-2 exponent of bu, accumulated over the subtree
    double* td;     // [C][N][ks]  top-down vectors; F81 sweeps write them only when asked to (null otherwise)
    i64* te;        // [C][N]
    double* post;   // [C][N][ks]
    double* lhsum;  // [C][N]
    i64* lhe;       // [C][N]
    bool implicit_tips;  // top-down sweep: the unit-vector posteriors of observed tips are not written (PML_OPT_IMPLICIT_TIP_POSTERIORS)
    pml_jt* J;      // [C][N][ks]  joint argmax tables (one byte per entry: k <= 256; two where jt16)
    bool jt16;      // more than 256 states: the entries of J are unsigned shorts
    int* js;        // [C][N]      joint states
    u64* err;       // [C]         min over failing (post_rank << 32 | child id)
    double* msg;    // [C][N][ks]  fused eigen sweeps: message of a node to its parent, P(t) applied to its BU vector
};

__device__ __forceinline__ int node_kind(const PmlTree& t, int n) {
    if (t.kind != nullptr) return t.kind[n];
    return t.n_children[n] == 0 ? PML_KIND_TIP : PML_KIND_STORED;
}

// ---------------------------------------------------------------------------------------------------------------------
// per-branch e = exp(-mu t') for every (node, column); for tips also S = pi . mask
// replaces: transform_t (models/__init__.py:269) + the exp of F81Model.get_Pij_t (F81Model.py:42-45)
// ---------------------------------------------------------------------------------------------------------------------
// A thread takes node n for `cpy` consecutive columns (blockIdx.y = column chunk): the branch length is read once per
// chunk instead of once per column (a third of this pass's traffic on a wide batch).
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
f81_prep_kernel(PmlTree t, PmlCols c, const double* __restrict__ mu, const double* __restrict__ sf,
                const double* __restrict__ tau, const double* __restrict__ tauf, PmlState st, int n_cols, int cpy) {
    const int col0 = blockIdx.y * cpy;
    const int col1 = min(n_cols, col0 + cpy);
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < t.N; n += gridDim.x * blockDim.x) {
        const double d = t.dist[n];
        const bool tip = t.n_children[n] == 0;
        for (int col = col0; col < col1; ++col) {
            if (!column_active(c, col)) continue;
            const size_t colN = (size_t)col * t.N;
            const double m = mu[col];
            const double tt = (d + tau[col]) * tauf[col] * sf[col];
            // if mu == inf (a single state) it wins over t == 0 (F81Model.py:44-45)
            st.E[colN + n] = isinf(m) ? 0.0 : exp(-m * tt);
            if (tip) {
                double acc = 0.0;
                for (int w = 0; w < c.W; ++w) {
                    u64 word = c.masks[(colN + n) * c.W + w];
                    while (word) {
                        const int b = __builtin_ctzll(word);
                        acc += c.pi[(size_t)col * c.ks + w * 64 + b];
                        word &= word - 1ull;
                    }
                }
                st.S[colN + n] = acc;  // a tip's exponent word stays 0 from the allocation of the column arrays
            }
        }
    }
}
#endif

// Per-lane context of a unit's lane group: lane geometry and the column's slabs (element offsets inside one column
// fit 32 bits: N * ks < 2^31 is checked on the host).
//
// State ownership ("chunked"): a lane owns R states, in R/2 pairs; pair q of lane g is states q*2G + 2g, +1.  So for
// every q the G lanes of a unit read one contiguous run of 2G doubles with 16-byte lane loads (R = 1: state g).
// widest unit (lanes) whose top-down level kernel stages its posterior rows in LDS (compile-time: the kernels of wider
// units carry none of that code; four lanes per unit measured 5 - 10 % slower staged, and slower unstaged with the code in)
#define PML_TD_STAGE_MAX_G 2

// orders a wave's LDS traffic (write by some lanes, read by others) without a workgroup barrier
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int G, int R>
struct LaneCtx {
    int col, g, group_base, k;
    double pi_r[R];
    double ipi_r[R];  // 1 / pi (0 where pi = 0): turns a stored posterior back into TD o BU (top-down sweep)
    const u64* mask;
    const double* pi;  // the column's frequency vector (ks doubles)
    const double* E;
    double* S;
    i64* be;
    double* bu;
    double* td;
    i64* te;
    double* post;
    double* lhsum;
    i64* lhe;
    // top-down level kernel, units narrower than 8 lanes: posterior rows of the unit are staged in LDS (slots of
    // c.ks doubles, node ids beside them) and written out by the whole wave in address order (td_stage_flush)
    // Slots of a unit: 0, 1 = its first two children (rows of c.ks doubles, ids beside them); 2 + 2 j + q = tip q < 2 of
    // child j < 2 when that is a cherry: an observed tip's posterior is a unit vector, so (id, state) stands for the row.
    double* srow;
    int* sid;
    int* stid;
    int* stst;
    double* ssum;  // staged lhsum / lhe of the six slots (null: not staged)
    i64* sexp;
    bool implicit_tips;  // observed tips' posteriors stay implicit (their mask says which unit vector it is)
    __device__ __forceinline__ int st(int r) const { return R == 1 ? g : ((r >> 1) * 2 * G + 2 * g + (r & 1)); }
};

template <int G, int R>
__device__ __forceinline__ void lane_ctx_init(LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                              const PmlState& st) {
    const int lane = threadIdx.x & 63;
    L.col = blockIdx.y;
    L.k = c.k;
    L.g = lane & (G - 1);
    L.group_base = lane & ~(G - 1);
    const size_t colN = (size_t)L.col * t.N;
    L.mask = c.masks + colN * c.W;
    L.pi = c.pi + (size_t)L.col * c.ks;
    L.E = st.E + colN;
    L.S = st.S + colN;
    L.be = st.be + colN;
    L.bu = st.bu + colN * c.ks;
    L.td = st.td + colN * c.ks;
    L.te = st.te + colN;
    L.post = st.post + colN * c.ks;
    L.lhsum = st.lhsum + colN;
    L.lhe = st.lhe + colN;
    L.srow = nullptr;
    L.sid = nullptr;
    L.stid = nullptr;
    L.stst = nullptr;
    L.ssum = nullptr;
    L.sexp = nullptr;
    L.implicit_tips = st.implicit_tips;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        L.pi_r[r] = (L.st(r) < c.k) ? c.pi[(size_t)L.col * c.ks + L.st(r)] : 0.0;
        L.ipi_r[r] = L.pi_r[r] > 0.0 ? 1.0 / L.pi_r[r] : 0.0;
    }
}

__device__ __forceinline__ u64 state_bits(int k) { return k >= 64 ? ~0ull : (1ull << k) - 1ull; }

template <int G, int R>
__device__ __forceinline__ void node_load_vec(const LaneCtx<G, R>& L, const PmlCols& c, const double* base, int n,
                                              double (&v)[R]);

// prod = TD_p o BU_p of a finished node p, recovered from what the sweep already stored for it: the marginal
// likelihoods are lh = TD o BU o pi o mask (ml.py:456-460) and the posterior is lh / sum(lh), so
// TD o BU = posterior * sum(lh) / pi wherever mask and pi are non-zero; elsewhere BU (hence the product) is 0 or the
// state has pi = 0 and cannot reach any output below p.  One 512-byte read instead of the BU and TD vectors, and the
// TD vectors need not be written at all.  Exponent of the result: lhe[p].
template <int G, int R>
__device__ __forceinline__ void f81_parent_prod(const LaneCtx<G, R>& L, const PmlCols& c, int p, double (&prod)[R],
                                                i64& pe) {
    double po[R];
    node_load_vec<G, R>(L, c, L.post, p, po);
    const double ls = L.lhsum[p];
    pe = L.lhe[p];
#pragma unroll
    for (int r = 0; r < R; ++r) prod[r] = po[r] * (ls * L.ipi_r[r]);
}

// 0/1 vector of the lane's states from a single mask word (k <= 64, hence G * R <= 64) whose bits >= k are clear:
// bit b -> (0 or 0x3FF00000) as the high word of the double by one signed bit-field extract and one AND -- no
// compares, no selects.  For G <= 16 a state pair never straddles the two halves of the word (2G divides 32), so the
// half is known at compile time and no 64-bit shift is needed.
template <int G, int R>
__device__ __forceinline__ void clean_word_to_vec(const LaneCtx<G, R>& L, const PmlCols& c, u64 w, double (&v)[R]) {
    if (R == 1) {
        v[0] = (L.g < c.k && ((w >> (L.g & 63)) & 1ull)) ? 1.0 : 0.0;
    } else if (G <= 16) {
        const int lo = (int)(unsigned)w, hi = (int)(unsigned)(w >> 32);
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const int half = ((2 * G * q) & 32) ? hi : lo;
            const unsigned sh = (unsigned)(((2 * G * q) & 31) + 2 * L.g);
            v[2 * q] = __hiloint2double(__builtin_amdgcn_sbfe(half, sh, 1u) & 0x3FF00000, 0);
            v[2 * q + 1] = __hiloint2double(__builtin_amdgcn_sbfe(half, sh + 1u, 1u) & 0x3FF00000, 0);
        }
    } else {
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const unsigned t = (unsigned)(w >> ((2 * G * q + 2 * L.g) & 63));
            v[2 * q] = __hiloint2double(((int)(t << 31) >> 31) & 0x3FF00000, 0);
            v[2 * q + 1] = __hiloint2double(((int)(t << 30) >> 31) & 0x3FF00000, 0);
        }
    }
}

template <int G, int R>
__device__ __forceinline__ void word_to_vec(const LaneCtx<G, R>& L, const PmlCols& c, u64 word, double (&v)[R]) {
    clean_word_to_vec<G, R>(L, c, word & state_bits(c.k), v);
}

template <int G, int R>
__device__ __forceinline__ void node_mask_vec(const LaneCtx<G, R>& L, const PmlCols& c, int n, double (&v)[R]) {
    if (c.W == 1) {
        word_to_vec<G, R>(L, c, L.mask[(unsigned)n], v);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = L.st(r);
            const u64 word = s < c.k ? L.mask[(unsigned)n * (unsigned)c.W + (unsigned)(s >> 6)] : 0ull;
            v[r] = ((word >> (s & 63)) & 1ull) ? 1.0 : 0.0;
        }
    }
}

template <int G, int R>
__device__ __forceinline__ void node_load_vec(const LaneCtx<G, R>& L, const PmlCols& c, const double* base, int n,
                                              double (&v)[R]) {
    const double* p = base + (unsigned)n * (unsigned)c.ks;
    if (R == 1) {
        v[0] = L.st(0) < c.ks ? p[L.st(0)] : 0.0;
    } else {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            if (L.st(r) < c.ks) {
                const double2 t2 = *reinterpret_cast<const double2*>(p + L.st(r));
                v[r] = t2.x;
                v[r + 1] = t2.y;
            } else {
                v[r] = v[r + 1] = 0.0;
            }
        }
    }
}

// streaming (non-temporal) variant for data that is written once and not re-read soon (posteriors).  Only where a
// store instruction of a unit covers whole 128-byte lines (G >= 8 lanes x 16 bytes): with fewer lanes per unit the
// lanes of a wavefront write 16-byte pieces of different rows, and it is the L2 that merges the pieces of a line --
// streamed past it they reach HBM as partial writes (k = 4, 262 144 tips, 32 characters: last top-down level 1.10 ms
// non-temporal).
template <int G, int R>
__device__ __forceinline__ void node_store_vec_nt(const LaneCtx<G, R>& L, const PmlCols& c, double* base, int n,
                                                  const double (&v)[R]) {
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    double* p = base + (unsigned)n * (unsigned)c.ks;
    if (R == 1) {
        if (L.st(0) < c.ks) {
            if (G >= 8) __builtin_nontemporal_store(v[0], p + L.st(0));
            else p[L.st(0)] = v[0];
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            if (L.st(r) < c.ks) {
                dbl2 t2;
                t2.x = v[r];
                t2.y = v[r + 1];
                if (G >= 8) __builtin_nontemporal_store(t2, reinterpret_cast<dbl2*>(p + L.st(r)));
                else *reinterpret_cast<dbl2*>(p + L.st(r)) = t2;
            }
        }
    }
}

template <int G, int R>
__device__ __forceinline__ void node_store_vec(const LaneCtx<G, R>& L, const PmlCols& c, double* base, int n,
                                               const double (&v)[R]) {
    double* p = base + (unsigned)n * (unsigned)c.ks;
    if (R == 1) {
        if (L.st(0) < c.ks) p[L.st(0)] = v[0];
    } else {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            if (L.st(r) < c.ks) {
                double2 t2;
                t2.x = v[r];
                t2.y = v[r + 1];
                *reinterpret_cast<double2*>(p + L.st(r)) = t2;
            }
        }
    }
}

// Posterior row of `node`: into the unit's staging slot when the kernel stages (slot >= 0 and within the unit's
// slots), straight to memory otherwise.  Why staged: one lane per unit writes 16-byte pieces of different rows per
// store instruction (64 - 128 bytes apart) and reaches 3.5 TB/s; the same bytes written by the wave in address order,
// every instruction covering 1 KB, reach 6.7 (scripts/ub/wr2.hip).
template <int G, int R>
__device__ __forceinline__ bool post_row(const LaneCtx<G, R>& L, const PmlCols& c, int slot, int node,
                                         const double (&v)[R]) {
    if (G <= PML_TD_STAGE_MAX_G && L.srow != nullptr && slot >= 0 && slot < 2) {
        double* p = L.srow + slot * c.ks;
        if (R == 1) {
            if (L.st(0) < c.ks) p[L.st(0)] = v[0];
        } else {
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                if (L.st(r) < c.ks) {
                    double2 t2;
                    t2.x = v[r];
                    t2.y = v[r + 1];
                    *reinterpret_cast<double2*>(p + L.st(r)) = t2;
                }
            }
        }
        if (L.g == 0) L.sid[slot] = node;
        return true;
    }
    node_store_vec_nt<G, R>(L, c, L.post, node, v);
    return false;
}

// Posterior of an observed tip with state s (the unit vector oh; ok = its likelihood is a positive finite number,
// otherwise oh holds NaNs and is written as it is): staged as (id, state) in a tip slot.
template <int G, int R>
__device__ __forceinline__ bool post_onehot(const LaneCtx<G, R>& L, const PmlCols& c, int slot, int tip, int s,
                                            const double (&oh)[R], bool ok) {
    if (L.implicit_tips && ok) return false;  // the row is the unit vector of the tip's one allowed state: not written
    if (G <= PML_TD_STAGE_MAX_G && L.srow != nullptr && slot >= 2 && slot < 6 && ok) {
        if (L.g == 0) {
            L.stid[slot - 2] = tip;
            L.stst[slot - 2] = s;
        }
        return true;
    }
    node_store_vec_nt<G, R>(L, c, L.post, tip, oh);
    return false;
}

// the two scalars that go with a posterior row (sum of the marginal likelihoods and its exponent): staged with it
// (slot = the row's slot if the row was staged, -1 otherwise)
template <int G, int R>
__device__ __forceinline__ void post_scalars(const LaneCtx<G, R>& L, int slot, int node, double lsum, i64 le) {
    if (L.g != 0) return;
    if (G <= PML_TD_STAGE_MAX_G && L.ssum != nullptr && slot >= 0 && slot < 6) {
        L.ssum[slot] = lsum;
        L.sexp[slot] = le;
    } else {
        L.lhsum[node] = lsum;
        L.lhe[node] = le;
    }
}

template <int G, int R>
__device__ __forceinline__ double pi_dot(const LaneCtx<G, R>& L, const double (&v)[R]) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += L.pi_r[r] * v[r];
    return group_sum<G>(s);
}

// Multiplies acc by the message of child ch (vector v, S = pi . v, branch factor e) and performs the zero check of
// ml.py:139-145.  Every term of the closed form is >= 0, so the clamp of ml.py:137 is a no-op.
template <int G, int R>
__device__ __forceinline__ void f81_absorb_child(const LaneCtx<G, R>& L, const PmlTree& t, const PmlState& st, int n,
                                                 int ch, double e, double s_child, const double (&v)[R],
                                                 double (&acc)[R], bool report) {
    const double a = (1.0 - e) * s_child;
    bool nz = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r] *= a + e * v[r];
        nz |= acc[r] != 0.0;
    }
    if (report && !group_any<G>(nz)) {
        if (L.g == 0) atomicMin(&st.err[L.col], ((u64)(unsigned)t.post_rank[n] << 32) | (u64)(unsigned)ch);
    }
}

// Bottom-up vector of a cherry (all children are tips) in registers: mask * prod of tip messages (ml.py:124-148).
// Messages of tips are >= (1 - e) pi_min, so the band is checked every fourth child only.
template <int G, int R>
__device__ __forceinline__ void f81_cherry_vector(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                  const PmlState& st, int n, double (&acc)[R], i64& esum,
                                                  bool report, bool tips_known = false, int fc_known = 0,
                                                  int nc_known = 0) {
    node_mask_vec<G, R>(L, c, n, acc);
    esum = 0;
    const int fc = tips_known ? fc_known : t.first_child[n];
    const int nc = tips_known ? nc_known : t.n_children[n];
    for (int j = 0; j < nc; ++j) {
        const int ch = fc + j;
        double v[R];
        node_mask_vec<G, R>(L, c, ch, v);
        f81_absorb_child<G, R>(L, t, st, n, ch, L.E[ch], L.S[ch], v, acc, report);
        if ((j & 3) == 3 || j == nc - 1) esum += lazy_rescale<G, R>(acc);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Lane-parallel metadata gather.  The scalar data of a unit's children (kind, e, S, mask word, exponent, child range)
// and of the tips below its cherry children are loaded by *different lanes* of the unit's lane group in two rounds --
// lane j takes child j, lane j * GC + q takes tip q of child j -- and broadcast with ds_bpermute when needed, instead
// of every lane walking the children one after the other (one dependent memory round trip per child and per tip).
// Needs single-word masks (k <= 64), at most CH children and at most GC tips per cherry; other units take the
// sequential path.  The arithmetic is the same sequence of operations, so both paths give identical bits.
// ---------------------------------------------------------------------------------------------------------------------
template <int G>
struct Gather {
    static constexpr int GC = 4;
    static constexpr int CH = (G / 4 < 4) ? G / 4 : 4;  // the descriptor codes the first four children
    static constexpr bool enabled = G >= 8;
};

// Unit descriptor, built once per tree on the host (pml_tree_upload): everything about the topology around node n that
// a unit needs, so that all data loads of a unit can be issued in one round trip (and the descriptor of the wave's next
// unit is fetched while the current one is computed) instead of chasing n -> first_child -> children -> their children.
//   packed: bits 0-3 number of children (15 = 15 or more), bit 4 = every cherry among the first four children has
//           1..GC tips, bit 5 = no stored internal node among children 2 and 3 (the bottom-up pipeline prefetches the
//           vectors of children 0 and 1), bits 8+3j..10+3j (j < 4) code of child j: 0 tip, 1 stored internal node,
//           2+m cherry with m+1 tips
//   cfc[j]: first child of child j (j < 4), i.e. where the tips of a cherry child start
struct __attribute__((aligned(32))) PmlUnit {
    int n, fc, packed, pad;
    int cfc[4];
};

__device__ __forceinline__ int unit_nc(int packed) { return packed & 15; }
__device__ __forceinline__ int unit_code(int packed, int j) { return (packed >> (8 + 3 * j)) & 7; }

// per lane: the descriptor's header and the cfc entry of the child this lane gathers tips for (units of 8 lanes and
// more); units of fewer lanes walk their children one after the other and keep all four entries (cfc = cfc[0])
struct UnitRegs {
    int n, fc, packed, cfc;
    int cfc1, cfc2, cfc3;
};

template <int G>
__device__ __forceinline__ UnitRegs load_unit(const PmlUnit* __restrict__ units, int idx, int g) {
    const int4 h = *reinterpret_cast<const int4*>(units + idx);
    UnitRegs u;
    u.n = h.x;
    u.fc = h.y;
    u.packed = h.z;
    if (G < 8) {
        const int4 f = *reinterpret_cast<const int4*>(units[idx].cfc);
        u.cfc = f.x;
        u.cfc1 = f.y;
        u.cfc2 = f.z;
        u.cfc3 = f.w;
    } else {
        u.cfc = units[idx].cfc[(g / Gather<G>::GC) & 3];
        u.cfc1 = u.cfc2 = u.cfc3 = 0;
    }
    return u;
}

// What the sequential paths know about child j of a unit without touching the tree arrays: its kind from the
// descriptor's codes (children 0 - 3), and for a cherry of a narrow unit where its tips start and how many they are
// (bit 4 of packed: every cherry among the first four children has 1 - 4 tips, so code - 1 is the number).
struct ChildDesc {
    int kind;
    bool tips_known;
    int fc2, nc2;
};
template <int G>
__device__ __forceinline__ ChildDesc unit_child(const PmlTree& t, const UnitRegs& u, int j, int ch) {
    ChildDesc d;
    d.tips_known = false;
    d.fc2 = d.nc2 = 0;
    if (j < 4) {
        const int code = unit_code(u.packed, j);
        d.kind = code == 0 ? PML_KIND_TIP : (code == 1 ? PML_KIND_STORED : PML_KIND_CHERRY);
        if (G < 8 && code >= 2 && ((u.packed >> 4) & 1)) {
            d.tips_known = true;
            d.fc2 = j == 0 ? u.cfc : (j == 1 ? u.cfc1 : (j == 2 ? u.cfc2 : u.cfc3));
            d.nc2 = code - 1;
        }
    } else {
        d.kind = node_kind(t, ch);
    }
    return d;
}

template <int G>
__device__ __forceinline__ bool unit_is_fast(int packed) {
    return unit_nc(packed) <= Gather<G>::CH && ((packed >> 4) & 1);
}
template <int G, bool VEC>
__device__ __forceinline__ bool unit_is_fast_bu(int packed) {
    if (!VEC && (unit_code(packed, 0) == 1 || unit_code(packed, 1) == 1)) return false;  // vectors are not prefetched
    return unit_nc(packed) <= Gather<G>::CH && ((packed >> 4) & 3) == 3;
}

// The mask words are stripped of their bits >= k when gathered, and the constant part a = (1 - e) S of a message
// (1 - e) S + e v is formed by the lane that owns the data, before the broadcast, instead of by every lane after it.
struct ChildLane {  // what lane j holds about child j
    double e, s;
    u64 mask;
    i64 be;
};

struct TipLane {  // what lane j * GC + q holds about tip q of cherry child j
    double e, s, a;
    u64 mask;
};

template <int G, int R, bool TIPS_OF_CHERRIES = true>
__device__ __forceinline__ void f81_gather_issue(const LaneCtx<G, R>& L, const UnitRegs& u, ChildLane& cl,
                                                 TipLane& tl) {
    constexpr int GC = Gather<G>::GC;
    const int nc = unit_nc(u.packed);
    const int j = L.g;
    const int ch = u.fc + (j < nc ? j : 0);
    cl.e = L.E[ch];
    cl.s = L.S[ch];
    cl.mask = L.mask[(unsigned)ch];
    cl.be = L.be[ch];
    if (!TIPS_OF_CHERRIES) {
        tl.e = tl.s = 0.0;
        tl.mask = 0ull;
        return;
    }
    const int jj = L.g / GC, q = L.g % GC;
    const int code = unit_code(u.packed, jj & 3);
    const bool has_t = jj < nc && q < code - 1;  // code - 1 = number of tips of a cherry child (<= 0 otherwise)
    const int tip = has_t ? u.cfc + q : u.fc;
    tl.e = L.E[tip];
    tl.s = L.S[tip];
    tl.mask = L.mask[(unsigned)tip];
}

// the part of the gather that needs the loaded values (kept apart so that the loads can be issued one unit ahead)
template <int G, int R>
__device__ __forceinline__ void f81_gather_finish(const LaneCtx<G, R>& L, ChildLane& cl, TipLane& tl) {
    const u64 kbits = state_bits(L.k);
    cl.mask &= kbits;
    tl.a = (1.0 - tl.e) * tl.s;
    tl.mask &= kbits;
}

template <int G, int R>
__device__ __forceinline__ void f81_gather(const LaneCtx<G, R>& L, const UnitRegs& u, ChildLane& cl, TipLane& tl) {
    f81_gather_issue<G, R>(L, u, cl, tl);
    f81_gather_finish<G, R>(L, cl, tl);
}

// out[r] = a + e * [state st(r) allowed by word] without forming the 0/1 vector: the two possible values are a (the fused
// multiply-add with 0.0 returns its addend) and c1 = a + e (with 1.0: the rounded sum), so the entry is picked by the
// mask bit -- one signed bit-field extract and two v_bfi_b32 per entry instead of extract, and, mov, fma.  Same bits.
template <int G, int R>
__device__ __forceinline__ void word_select_vec(const LaneCtx<G, R>& L, const PmlCols& c, u64 w, double a, double e,
                                                double (&out)[R]) {
    if (R >= 2 && G <= 16) {
        const double c1 = a + e;
        const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
        const unsigned clo = (unsigned)__double2loint(c1), chi = (unsigned)__double2hiint(c1);
        const int lo = (int)(unsigned)w, hi = (int)(unsigned)(w >> 32);
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const int half = ((2 * G * q) & 32) ? hi : lo;
            const unsigned sh = (unsigned)(((2 * G * q) & 31) + 2 * L.g);
            const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe(half, sh, 1u);
            const unsigned m1 = (unsigned)__builtin_amdgcn_sbfe(half, sh + 1u, 1u);
            out[2 * q] = __hiloint2double((int)((chi & m0) | (ahi & ~m0)), (int)((clo & m0) | (alo & ~m0)));
            out[2 * q + 1] = __hiloint2double((int)((chi & m1) | (ahi & ~m1)), (int)((clo & m1) | (alo & ~m1)));
        }
    } else {
        double tv[R];
        clean_word_to_vec<G, R>(L, c, w, tv);
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = a + e * tv[r];
    }
}

// cherry child jx of the unit rebuilt from the gathered tip data; same operations in the same order as
// f81_cherry_vector.  Band check: every factor of a non-zero entry is a_q or a_q + e_q <= 1 with a_q = (1 - e_q) S_q,
// so the non-zero entries lie in [prod a_q, 1]: when that product is >= 2^-200 (the usual case; one multiplication per
// tip) nothing can have left the band and the per-state test is skipped -- lazy_rescale would have returned 0.
// full: the cherry's own mask allows every state (wave-uniform): its 0/1 vector is all ones and the product starts with
// the first tip's message itself (1.0 * x = x).
template <int G, int R>
__device__ __forceinline__ void f81_cherry_from_lanes(const LaneCtx<G, R>& L, const PmlCols& c, const ChildLane& cl,
                                                      const TipLane& tl, int jx, int cnc, double (&v)[R], i64& esum,
                                                      bool full = false) {
    constexpr int GC = Gather<G>::GC;
    esum = 0;
    double amin = 1.0;
    int q0 = 0;
    if (full && cnc > 0) {
        const int ts = L.group_base + jx * GC;
        const double a = __shfl(tl.a, ts, 64);
        word_select_vec<G, R>(L, c, __shfl(tl.mask, ts, 64), a, __shfl(tl.e, ts, 64), v);
        amin = a;
        q0 = 1;
    } else {
        clean_word_to_vec<G, R>(L, c, __shfl(cl.mask, L.group_base + jx, 64), v);
    }
    for (int q = q0; q < cnc; ++q) {  // cnc <= GC = 4: one band check, after the last tip
        const int ts = L.group_base + jx * GC + q;
        double msg[R];
        const double a = __shfl(tl.a, ts, 64);
        word_select_vec<G, R>(L, c, __shfl(tl.mask, ts, 64), a, __shfl(tl.e, ts, 64), msg);
        amin *= a;
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] *= msg[r];
    }
    if (!(amin >= 0x1p-200)) esum = lazy_rescale<G, R>(v);
}

// Message of a child with bottom-up vector v in Pupko's max variant (ml.py:134-136) and the arg-max row that goes
// with it.  Row i of P * diag(v): off-diagonal entries w_j = ((1-e) pi_j) v_j, diagonal ((1-e) pi_i + e) v_i (same
// rounding sequence as the reference's P * v broadcast, ml.py:130 with F81Model.py:46), so the row maximum is the larger
// of the diagonal entry and the best off-diagonal one: the top two of w over the unit's lanes, two butterflies.
template <int G, int R>
__device__ __forceinline__ void f81_joint_message(const LaneCtx<G, R>& L, const PmlCols& c, double e,
                                                  const double (&v)[R], double (&msg)[R], int (&jj)[R]) {
    const double ome = 1.0 - e;
    double w[R], dg[R];
    double m1 = -INFINITY;
    int j1 = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double a = ome * L.pi_r[r];
        const bool ok = L.st(r) < c.k;
        w[r] = ok ? a * v[r] : -INFINITY;
        dg[r] = (a + e) * v[r];
        if (ok && w[r] > m1) {
            m1 = w[r];
            j1 = L.st(r);
        }
    }
    group_argmax_first<G>(m1, j1);
    double m2 = -INFINITY;
    int j2 = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (L.st(r) < c.k && L.st(r) != j1 && w[r] > m2) {
            m2 = w[r];
            j2 = L.st(r);
        }
    }
    group_argmax_first<G>(m2, j2);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = L.st(r);
        const double mo = (i == j1) ? m2 : m1;
        const int jo = (i == j1) ? j2 : j1;
        int arg;
        if (jo >= c.k || dg[r] > mo) {  // no off-diagonal candidate (k == 1) or the diagonal wins
            msg[r] = dg[r];
            arg = i;
        } else if (dg[r] < mo) {
            msg[r] = mo;
            arg = jo;
        } else {  // tie: numpy's argmax returns the first index
            msg[r] = mo;
            arg = min(i, jo);
        }
        jj[r] = (i < c.k) ? arg : 0;
    }
}

// One row of an arg-max table: a byte per state; the two states of a lane's pair go out as one 16-bit store, so the G
// lanes of a unit write 2G consecutive bytes per pair index.
template <int G, int R>
__device__ __forceinline__ void f81_store_table(const LaneCtx<G, R>& L, const PmlCols& c, pml_jt* __restrict__ jp,
                                                const int (&jj)[R]) {
    if (G * R > 256) {
        // more than 256 states (64 lanes x 8): 16-bit entries, jp counts entries -- the caller's pointer arithmetic is in entries
        // of the byte table, so the row starts at twice the offset: rebuilt here from the row's entry index
        unsigned short* wp = reinterpret_cast<unsigned short*>(jp);
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            if (L.st(r) < c.ks)
                *reinterpret_cast<unsigned*>(wp + L.st(r)) = (unsigned)(jj[r] & 0xffff) | ((unsigned)(jj[r + 1] & 0xffff) << 16);
        }
    } else if (R == 1) {
        if (L.st(0) < c.ks) jp[L.st(0)] = (pml_jt)jj[0];
    } else {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            if (L.st(r) < c.ks)
                *reinterpret_cast<unsigned short*>(jp + L.st(r)) = (unsigned short)((jj[r] & 0xff) | ((jj[r + 1] & 0xff) << 8));
        }
    }
}

// the row of node `node` of column L.col in the arg-max tables (entries of one or two bytes)
template <int G, int R>
__device__ __forceinline__ pml_jt* f81_table_row(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c, const PmlState& st,
                                                 int node) {
    const size_t entry = ((size_t)L.col * t.N + node) * c.ks;
    return st.J + (G * R > 256 ? 2 * entry : entry);
}

// Everything a fast bottom-up unit reads, as issued loads: the gathered scalars, the unit's own mask word and the
// vectors of its first two children where those are stored nodes.  Filled one unit ahead by the level kernel (software
// pipeline: the loads of unit i + 1 are in flight while unit i is computed), on the spot elsewhere.
template <int R>
struct BuLoads {
    ChildLane cl;
    TipLane tl;
    u64 own;
    double v0[R], v1[R];
};

// Valid for every descriptor (fast or not): all addresses are those of existing nodes.  The number of loads is the
// same for every unit (VEC: a child that is not a stored node reads the column's frequency vector instead, a cache
// hit), so that the compiler can wait for exactly the loads of one unit while those of the next stay in flight.
// VEC = false is for levels without stored children (the level that rebuilds cherries): no vector loads at all.
template <int G, int R, bool VEC, bool JOINT = false>
__device__ __forceinline__ void bu_f81_issue(const LaneCtx<G, R>& L, const PmlCols& c, const UnitRegs& u,
                                             BuLoads<R>& ld) {
    f81_gather_issue<G, R>(L, u, ld.cl, ld.tl);
    ld.own = L.mask[(unsigned)u.n];
    if (VEC) {
        const bool s0 = unit_code(u.packed, 0) == 1, s1 = unit_nc(u.packed) > 1 && unit_code(u.packed, 1) == 1;
        node_load_vec<G, R>(L, c, s0 ? L.bu : L.pi, s0 ? u.fc : 0, ld.v0);
        node_load_vec<G, R>(L, c, s1 ? L.bu : L.pi, s1 ? u.fc + 1 : 0, ld.v1);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) ld.v0[r] = ld.v1[r] = 0.0;
    }
}

// Makes every value of ld available here: the compiler places ONE s_waitcnt with the right count at this point (the
// loads issued after ld's -- the next unit's -- stay in flight) instead of conservative vmcnt(0) waits wherever the
// unit's control flow first touches a value.
template <int R, bool VEC, bool JOINT = false>
__device__ __forceinline__ void bu_loads_arrived(BuLoads<R>& ld) {
    asm volatile("" : "+v"(ld.cl.e), "+v"(ld.cl.s), "+v"(ld.cl.mask), "+v"(ld.cl.be), "+v"(ld.tl.e), "+v"(ld.tl.s),
                      "+v"(ld.tl.mask), "+v"(ld.own));
    if (VEC) {
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" : "+v"(ld.v0[r]), "+v"(ld.v1[r]));
    }
}

// The marginal fast unit.  FULL (wave-uniform, decided by the caller): the unit's own mask and the masks of its cherry
// children allow every state, the usual case for internal nodes -- their 0/1 vectors are all ones and drop out of the
// products (1.0 * x = x: same bits, no mask expansion and no multiplication).  Band checks: the factor a child
// contributes to a non-zero entry is fma(e, v, a) >= a = (1 - e) S, and <= 1 for tips and cherries (v <= 1), so after
// children that are not stored nodes the non-zero entries lie in [prod a, 1]: while that product stays above 2^-190 no
// entry can have left the band [2^-200, 2^200] and lazy_rescale -- 24 compares per call -- would have returned 0.
// PAIR (wave-uniform, decided by the caller): every unit of the wave has two children, both cherries of two tips -- the
// shape of a whole level of a balanced binary tree.  Child and tip counts are then compile-time constants: the loops
// unroll into straight-line code (no per-child exec masks, the lane exchanges of all four tips issued together).
// What a unit's body leaves in registers for a caller that goes on with the node's parent (two-level units below).
template <int R>
struct BuResult {
    double v[R];
    double s;
    i64 e;
};

// the vectors (and rescaling exponents) of a PAIR unit's two cherries, for a caller that goes on with them
template <int R>
struct CherryKeep {
    double v[2][R];
    i64 e[2];
};

template <int G, int R, bool VEC, bool FULL, bool PAIR = false>
__device__ __forceinline__ bool bu_f81_marg_body(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                 const PmlState& st, const UnitRegs& u, BuLoads<R>& ld,
                                                 BuResult<R>* keep = nullptr, bool store_vec = true,
                                                 bool vec_only = false, CherryKeep<R>* cherries = nullptr) {
    const int n = u.n, fc = u.fc;
    const int nc = PAIR ? 2 : unit_nc(u.packed);
    ChildLane& cl = ld.cl;
    TipLane& tl = ld.tl;
    double acc[R];
    if (!FULL) word_to_vec<G, R>(L, c, ld.own, acc);
    i64 esum = 0;
    double lob = 1.0;     // lower bound of the non-zero entries of acc
    bool bounded = true;  // ... valid (no stored child so far, no rescaling so far)
    auto child = [&](const int jx) {
        const int src = L.group_base + jx;
        const int ch = fc + jx;
        const int code = PAIR ? 3 : unit_code(u.packed, jx);
        const double e = __shfl(cl.e, src, 64);
        double msg[R];
        double a;
        if (code == 0) {
            a = (1.0 - e) * __shfl(cl.s, src, 64);
            word_select_vec<G, R>(L, c, __shfl(cl.mask, src, 64), a, e, msg);
        } else {
            double v[R];
            double s_child;
            if (code == 1) {
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = jx == 0 ? ld.v0[r] : ld.v1[r];  // unit_is_fast_bu: jx < 2 here
                esum += __shfl(cl.be, src, 64);
                s_child = __shfl(cl.s, src, 64);
                bounded = false;
            } else {
                i64 ce;
                f81_cherry_from_lanes<G, R>(L, c, cl, tl, jx, code - 1, v, ce, FULL);
                esum += ce;
                if (PAIR && cherries != nullptr) {  // (jx is a constant in the straight-line body)
#pragma unroll
                    for (int r = 0; r < R; ++r) cherries->v[jx & 1][r] = v[r];
                    cherries->e[jx & 1] = ce;
                }
                if (vec_only) {
                    s_child = __shfl(cl.s, src, 64);  // as the bottom-up sweep stored it
                } else {
                    s_child = pi_dot<G, R>(L, v);
                    if (L.g == 0) L.S[ch] = s_child;  // 8 bytes kept for the top-down sweep (saves its reduction there)
                }
            }
            a = (1.0 - e) * s_child;
#pragma unroll
            for (int r = 0; r < R; ++r) msg[r] = a + e * v[r];
        }
        lob *= a;
        if (FULL && jx == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = msg[r];
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] *= msg[r];
        }
        if ((jx & 1) == 1 || jx == nc - 1) {
            if (!bounded || !(lob >= 0x1p-190)) {
                const int ex = lazy_rescale<G, R>(acc);
                esum += ex;
                if (ex != 0) bounded = false;
            }
        }
    };
    if (PAIR) {
        child(0);
        child(1);
    } else {
        for (int jx = 0; jx < nc; ++jx) child(jx);
    }
    if (FULL && nc == 0) {  // (no descriptor of a level has no children; keeps acc defined)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = L.st(r) < c.k ? 1.0 : 0.0;
    }
    if (vec_only) {  // (the top-down two-level unit: the vector again, nothing stored)
#pragma unroll
        for (int r = 0; r < R; ++r) keep->v[r] = acc[r];
        keep->s = 0.0;
        keep->e = esum;
        return true;
    }
    const double s = pi_dot<G, R>(L, acc);
    if (!(s > 0.0)) {  // pi . acc > 0 already says that acc is not all zero
        bool nz = false;
#pragma unroll
        for (int r = 0; r < R; ++r) nz |= acc[r] != 0.0 && L.st(r) < c.k;
        if (!group_any<G>(nz)) return false;
    }
    if (L.g == 0) {
        L.S[n] = s;
        L.be[n] = esum;
    }
    if (store_vec) node_store_vec<G, R>(L, c, L.bu, n, acc);
    if (keep != nullptr) {
#pragma unroll
        for (int r = 0; r < R; ++r) keep->v[r] = acc[r];
        keep->s = s;
        keep->e = esum;
    }
    return true;
}

// Returns false when the unit's vector came out all zero: a product only ever gains zeros, so the zero check of
// ml.py:139-145 is made once at the end, and the caller repeats the unit on the sequential path, which checks after
// every child and reports the pair the reference would name.
template <int G, int R, bool VEC, bool JOINT = false>
__device__ __forceinline__ bool bu_f81_unit_fast(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                 const PmlState& st, const UnitRegs& u, BuLoads<R>& ld) {
    const int n = u.n, fc = u.fc;
    const int nc = unit_nc(u.packed);
    bu_loads_arrived<R, VEC, JOINT>(ld);
    double acc[R];
    word_to_vec<G, R>(L, c, ld.own, acc);
    ChildLane& cl = ld.cl;
    TipLane& tl = ld.tl;
    f81_gather_finish<G, R>(L, cl, tl);
    if (!JOINT) {
        // every mask that would only contribute a vector of ones?  (own mask; masks of the cherry children.  Lanes
        // beyond the children hold child 0's data again.)  Wave-uniform over the units that take this path.
        const u64 kbits = state_bits(c.k);
        const int cj = L.g < nc ? L.g : 0;
        // (only without padding states, k = G * R: a padding entry must stay 0, and only a mask makes it so)
        const bool ones = c.k == G * R && (ld.own & kbits) == kbits && (unit_code(u.packed, cj & 3) < 2 || cl.mask == kbits);
        // (specialised only where it pays: the level that rebuilds cherries is bound by its instruction stream, the
        // levels that stream stored vectors by memory -- there the second body only costs registers: spills at R = 4)
        if (!VEC && __all(ones)) {
            // nc = 2, both children cherries of two tips (codes 3, 3; bits 4, 5 of packed are set for fast units)
            const bool pair = (u.packed & 0x3f0f) == ((3 << 11) | (3 << 8) | 2);
            if (__all(pair)) return bu_f81_marg_body<G, R, VEC, true, true>(L, t, c, st, u, ld);
            return bu_f81_marg_body<G, R, VEC, true>(L, t, c, st, u, ld);
        }
        return bu_f81_marg_body<G, R, VEC, false>(L, t, c, st, u, ld);
    }
    i64 esum = 0;
    for (int jx = 0; jx < nc; ++jx) {
        const int src = L.group_base + jx;
        const int ch = fc + jx;
        const int code = unit_code(u.packed, jx);
        const double e = __shfl(cl.e, src, 64);
        double v[R];
        if (JOINT) {
            // Pupko's max variant (every internal node is a stored node here: codes 0 and 1 only)
            double msg[R];
            int jj[R];
            bool closed = false;
            if (code == 0) {
                // An observed tip has one non-zero product per row, so max = sum: its message is the marginal one,
                // a + e [i == s] with a = (1 - e) pi_s, and every row's arg-max is s (a > 0; else the general scan).
                const u64 word = __shfl(cl.mask, src, 64);
                const double a = (1.0 - e) * __shfl(cl.s, src, 64);
                clean_word_to_vec<G, R>(L, c, word, v);
                if (__popcll(word) == 1 && a > 0.0) {
                    const int s = __builtin_ctzll(word);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        msg[r] = a + e * v[r];
                        jj[r] = L.st(r) < c.k ? s : 0;
                    }
                    closed = true;
                }
            } else if (code == 1) {
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = jx == 0 ? ld.v0[r] : ld.v1[r];
                esum += __shfl(cl.be, src, 64);
            } else {
                // Cherry child (fused joint sweep): rebuilt from its tips as in the marginal sweep when every tip is
                // observed with a > 0 -- their max-messages are then the marginal ones and their arg-max rows are
                // constant; any other tip sends the unit to the sequential path.
                constexpr int GC = Gather<G>::GC;
                const int cnc = code - 1;
                const int cfc = __shfl(u.cfc, L.group_base + jx * GC, 64);
                bool ok = true;
                for (int q = 0; q < cnc; ++q) {
                    const int ts = L.group_base + jx * GC + q;
                    ok &= __popcll(__shfl(tl.mask, ts, 64)) == 1 && __shfl(tl.a, ts, 64) > 0.0;
                }
                if (!ok) return false;
                i64 ce;
                f81_cherry_from_lanes<G, R>(L, c, cl, tl, jx, cnc, v, ce);
                esum += ce;
                for (int q = 0; q < cnc; ++q) {
                    const int s = __builtin_ctzll(__shfl(tl.mask, L.group_base + jx * GC + q, 64));
                    int tj[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) tj[r] = L.st(r) < c.k ? s : 0;
                    f81_store_table<G, R>(L, c, f81_table_row<G, R>(L, t, c, st, cfc + q), tj);
                }
            }
            if (!closed) f81_joint_message<G, R>(L, c, e, v, msg, jj);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] *= fmax(msg[r], 0.0);
            f81_store_table<G, R>(L, c, f81_table_row<G, R>(L, t, c, st, ch), jj);
            esum += lazy_rescale<G, R>(acc);
            continue;
        }
        double s_child;
        if (code == 0) {
            clean_word_to_vec<G, R>(L, c, __shfl(cl.mask, src, 64), v);
            s_child = __shfl(cl.s, src, 64);
        } else if (code == 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = jx == 0 ? ld.v0[r] : ld.v1[r];  // unit_is_fast_bu: jx < 2 here
            esum += __shfl(cl.be, src, 64);
            s_child = __shfl(cl.s, src, 64);
        } else {
            i64 ce;
            f81_cherry_from_lanes<G, R>(L, c, cl, tl, jx, code - 1, v, ce);
            esum += ce;
            s_child = pi_dot<G, R>(L, v);
            if (L.g == 0) L.S[ch] = s_child;  // 8 bytes kept for the top-down sweep (saves its reduction there)
        }
        f81_absorb_child<G, R>(L, t, st, n, ch, e, s_child, v, acc, false);
        if ((jx & 1) == 1 || jx == nc - 1) esum += lazy_rescale<G, R>(acc);
    }
    if (JOINT) {
        bool nz = false;
#pragma unroll
        for (int r = 0; r < R; ++r) nz |= acc[r] != 0.0 && L.st(r) < c.k;
        if (!group_any<G>(nz)) return false;
        if (L.g == 0) L.be[n] = esum;
        node_store_vec<G, R>(L, c, L.bu, n, acc);
        return true;
    }
    const double s = pi_dot<G, R>(L, acc);
    if (!(s > 0.0)) {  // pi . acc > 0 already says that acc is not all zero
        bool nz = false;
#pragma unroll
        for (int r = 0; r < R; ++r) nz |= acc[r] != 0.0 && L.st(r) < c.k;
        if (!group_any<G>(nz)) return false;
    }
    if (L.g == 0) {
        L.S[n] = s;
        L.be[n] = esum;
    }
    node_store_vec<G, R>(L, c, L.bu, n, acc);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Lean unit for units of fewer than 8 lanes.  In the kernels that walk several levels in one launch (walk_levels: small
// forests, subtree blocks, the narrow end of large forests) a level holds a handful of units and a level step is the
// latency of ONE unit, which on the sequential path is a chain of dependent round trips to L2 (child -> its
// kind -> its scalars -> the cherry's tips -> their scalars; ~45 s_waitcnt in the unit's code).  The descriptor says
// where everything lives, so this unit issues every load it can need -- own mask, both children's scalars and vectors,
// the scalars of up to two tips under each -- before it touches any value: one round trip.  Arithmetic: the operations
// of the sequential path in its order (tip messages picked by the mask bit, word_select_vec; band checks skipped where
// the product of the message floors proves them void, as in bu_f81_marg_body): the same bits.
// Units it takes: single-word masks, at most two children, cherries of at most two tips.  Returns false for an all-zero
// result: the caller repeats the unit on the sequential path, which reports the pair the reference would name.
// The level kernels of large forests take it too (units of fewer than 8 lanes, k <= 28): 262 144 tips x 32 characters,
// marginal pass k = 2 0.66 -> 0.56 ms, k = 4 0.76 -> 0.68, k = 8 1.11 -> 1.00, k = 12 1.76 -> 1.39 (round 2's
// descriptor-driven path for these units, which lost 2x, had every lane load every child / tip slot of up to four
// children; this one keeps to two children and two tips and to the loads the codes ask for).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool unit_is_lean(int packed) {
    if (unit_nc(packed) > 2 || !((packed >> 4) & 1)) return false;
    return unit_code(packed, 0) <= 3 && unit_code(packed, 1) <= 3;
}

// SC0, SC1: the codes of the two children when the caller knows them at compile time (SC1 = -1: one child; walk_levels
// dispatches the common shapes: the unit's decoding, the loads and the branches of the other shapes fall away -- where a
// level is a unit or two, the wave's instruction count is the level's time); PML_SHAPE_ANY: read from the descriptor.
#define PML_SHAPE_ANY (-9)
template <int G, int R, int SC0 = PML_SHAPE_ANY, int SC1 = PML_SHAPE_ANY>
__device__ __forceinline__ bool bu_f81_unit_lean(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                 const PmlState& st, const UnitRegs& u) {
    constexpr bool ANY = SC0 == PML_SHAPE_ANY;
    const int n = u.n, fc = u.fc;
    const int nc = ANY ? unit_nc(u.packed) : (SC1 < 0 ? 1 : 2);
    const u64 kbits = state_bits(c.k);
    // ---- loads
    const u64 own = L.mask[(unsigned)n];
    double ce[2], cs[2], vv[2][R];
    u64 cm[2];
    i64 cbe[2];
    double te[2][2], ts[2][2];
    u64 tm[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = fc + (j < nc ? j : 0);
        const int code = j < nc ? (ANY ? unit_code(u.packed, j) : (j == 0 ? SC0 : SC1)) : 0;
        ce[j] = L.E[ch];
        cm[j] = code != 1 ? L.mask[(unsigned)ch] : 0ull;
        cs[j] = code <= 1 ? L.S[ch] : 0.0;
        cbe[j] = code == 1 ? L.be[ch] : 0;
        if (code == 1) {
            node_load_vec<G, R>(L, c, L.bu, ch, vv[j]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) vv[j][r] = 0.0;
        }
        const int cfc = j == 0 ? u.cfc : u.cfc1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // (only what the codes ask for: in the thin levels a wave holds a unit or two and the loads it issues,
            // address arithmetic included, are a third of the level step)
            te[j][q] = 0.0;
            ts[j][q] = 0.0;
            tm[j][q] = 0ull;
            if (code >= 2 && q < code - 1) {
                te[j][q] = L.E[cfc + q];
                ts[j][q] = L.S[cfc + q];
                tm[j][q] = L.mask[(unsigned)(cfc + q)];
            }
        }
    }
    // ---- arithmetic (bu_f81_unit_seq's operations)
    double acc[R];
    clean_word_to_vec<G, R>(L, c, own & kbits, acc);
    i64 esum = 0;
    double lob = 1.0;
    bool bounded = true;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j < nc) {
            const int code = ANY ? unit_code(u.packed, j) : (j == 0 ? SC0 : SC1);
            const double e = ce[j];
            double msg[R];
            double a;
            if (code == 0) {
                a = (1.0 - e) * cs[j];
                word_select_vec<G, R>(L, c, cm[j] & kbits, a, e, msg);
            } else {
                double v[R];
                double s_child;
                if (code == 1) {
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = vv[j][r];
                    esum += cbe[j];
                    s_child = cs[j];
                    bounded = false;
                } else {
                    // cherry: mask, then per tip its message and the product; band check after the last tip
                    clean_word_to_vec<G, R>(L, c, cm[j] & kbits, v);
                    double amin = 1.0;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q < code - 1) {
                            const double ta = (1.0 - te[j][q]) * ts[j][q];
                            double tmsg[R];
                            word_select_vec<G, R>(L, c, tm[j][q] & kbits, ta, te[j][q], tmsg);
                            amin *= ta;
#pragma unroll
                            for (int r = 0; r < R; ++r) v[r] *= tmsg[r];
                        }
                    }
                    if (!(amin >= 0x1p-190)) esum += lazy_rescale<G, R>(v);
                    s_child = pi_dot<G, R>(L, v);
                    if (L.g == 0) L.S[fc + j] = s_child;
                }
                a = (1.0 - e) * s_child;
#pragma unroll
                for (int r = 0; r < R; ++r) msg[r] = a + e * v[r];
            }
            lob *= a;
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] *= msg[r];
            if (j == 1 || j == nc - 1) {
                if (!bounded || !(lob >= 0x1p-190)) {
                    const int ex = lazy_rescale<G, R>(acc);
                    esum += ex;
                    if (ex != 0) bounded = false;
                }
            }
        }
    }
    const double s = pi_dot<G, R>(L, acc);
    if (!(s > 0.0)) {
        bool nz = false;
#pragma unroll
        for (int r = 0; r < R; ++r) nz |= acc[r] != 0.0 && L.st(r) < c.k;
        if (!group_any<G>(nz)) return false;
    }
    if (L.g == 0) {
        L.S[n] = s;
        L.be[n] = esum;
    }
    node_store_vec<G, R>(L, c, L.bu, n, acc);
    return true;
}

// Lean unit for polytomies in the kernels that walk several levels in one launch (units of fewer than 8 lanes): up to four
// children, cherries of up to four tips -- what the descriptor codes.  The children are taken two at a time: every load of
// a pair (scalars, stored vectors, the scalars of up to four tips under each) is issued before any of its values is used,
// so a unit of three or four children is two round trips instead of one per child and per tip (the sequential path) --
// in those kernels a level step is the latency of one unit.  The arithmetic is the sequential path's, operation by
// operation (bu_f81_unit_lean's body with the band checks after every second child): the same bits.  Not in the level
// kernels of large forests: there the registers of the 2 x 4 tips cost binary forests 7 - 10 % and a wavefront that mixes
// these units with sequential ones loses more than it gains (profiles/r05z_lean_polytomies.txt).
__device__ __forceinline__ bool unit_is_lean_poly(int packed) {
    return unit_nc(packed) <= 4 && ((packed >> 4) & 1) && !unit_is_lean(packed);
}

template <int G, int R>
__device__ __forceinline__ bool bu_f81_unit_lean_poly(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                      const PmlState& st, const UnitRegs& u) {
    const int n = u.n, fc = u.fc;
    const int nc = unit_nc(u.packed);
    const u64 kbits = state_bits(c.k);
    const u64 own = L.mask[(unsigned)n];
    double acc[R];
    i64 esum = 0;
    double lob = 1.0;
    bool bounded = true;
    for (int j0 = 0; j0 < nc; j0 += 2) {
        // ---- loads of the pair
        double ce[2], cs[2], vv[2][R];
        u64 cm[2];
        i64 cbe[2];
        double te[2][4], ts[2][4];
        u64 tm[2][4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = j0 + jj;
            const bool valid = j < nc;
            const int ch = fc + (valid ? j : j0);
            const int code = valid ? unit_code(u.packed, j) : 0;
            ce[jj] = L.E[ch];
            cm[jj] = code != 1 ? L.mask[(unsigned)ch] : 0ull;
            cs[jj] = code <= 1 ? L.S[ch] : 0.0;
            cbe[jj] = code == 1 ? L.be[ch] : 0;
            if (code == 1) {
                node_load_vec<G, R>(L, c, L.bu, ch, vv[jj]);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) vv[jj][r] = 0.0;
            }
            const int cfc = j == 0 ? u.cfc : (j == 1 ? u.cfc1 : (j == 2 ? u.cfc2 : u.cfc3));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                te[jj][q] = 0.0;
                ts[jj][q] = 0.0;
                tm[jj][q] = 0ull;
                if (code >= 2 && q < code - 1) {
                    te[jj][q] = L.E[cfc + q];
                    ts[jj][q] = L.S[cfc + q];
                    tm[jj][q] = L.mask[(unsigned)(cfc + q)];
                }
            }
        }
        if (j0 == 0) clean_word_to_vec<G, R>(L, c, own & kbits, acc);
        // ---- arithmetic (bu_f81_unit_seq's operations)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = j0 + jj;
            if (j < nc) {
                const int code = unit_code(u.packed, j);
                const double e = ce[jj];
                double msg[R];
                double a;
                if (code == 0) {
                    a = (1.0 - e) * cs[jj];
                    word_select_vec<G, R>(L, c, cm[jj] & kbits, a, e, msg);
                } else {
                    double v[R];
                    double s_child;
                    if (code == 1) {
#pragma unroll
                        for (int r = 0; r < R; ++r) v[r] = vv[jj][r];
                        esum += cbe[jj];
                        s_child = cs[jj];
                        bounded = false;
                    } else {
                        // cherry: mask, then per tip its message and the product; band check after the last tip
                        clean_word_to_vec<G, R>(L, c, cm[jj] & kbits, v);
                        double amin = 1.0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (q < code - 1) {
                                const double ta = (1.0 - te[jj][q]) * ts[jj][q];
                                double tmsg[R];
                                word_select_vec<G, R>(L, c, tm[jj][q] & kbits, ta, te[jj][q], tmsg);
                                amin *= ta;
#pragma unroll
                                for (int r = 0; r < R; ++r) v[r] *= tmsg[r];
                            }
                        }
                        if (!(amin >= 0x1p-190)) esum += lazy_rescale<G, R>(v);
                        s_child = pi_dot<G, R>(L, v);
                        if (L.g == 0) L.S[fc + j] = s_child;
                    }
                    a = (1.0 - e) * s_child;
#pragma unroll
                    for (int r = 0; r < R; ++r) msg[r] = a + e * v[r];
                }
                lob *= a;
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] *= msg[r];
                if (jj == 1 || j == nc - 1) {
                    if (!bounded || !(lob >= 0x1p-190)) {
                        const int ex = lazy_rescale<G, R>(acc);
                        esum += ex;
                        if (ex != 0) bounded = false;
                    }
                }
            }
        }
    }
    const double s = pi_dot<G, R>(L, acc);
    if (!(s > 0.0)) {
        bool nz = false;
#pragma unroll
        for (int r = 0; r < R; ++r) nz |= acc[r] != 0.0 && L.st(r) < c.k;
        if (!group_any<G>(nz)) return false;
    }
    if (L.g == 0) {
        L.S[n] = s;
        L.be[n] = esum;
    }
    node_store_vec<G, R>(L, c, L.bu, n, acc);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// More than 64 states (masks of several words; units of 32 and 64 lanes).  The lane-parallel and lean paths above take
// single-word masks, so such units ran the sequential path: a dependent round trip to L2 per child and per tip, twice
// the level step of k = 64 (HIV1C tree x 68 columns, bottom-up sweep: k = 67 0.36 ms against 0.17 at k = 64).  The lean
// unit again, with the mask bits of the lane's own state pairs instead of a word: every load up front (own mask, both
// children's scalars / vectors / mask words, up to two tips under each), then the sequential path's operations in its order
// (a message entry is a + e or a by the mask bit: fma(e, 1.0, a) and fma(e, 0.0, a) are exactly that).
// ---------------------------------------------------------------------------------------------------------------------
// the words that hold the lane's state pairs: pair q of lane g is states 2 G q + 2 g and + 1, word (s >> 6)
template <int G, int R>
__device__ __forceinline__ void lane_words_issue(const LaneCtx<G, R>& L, const PmlCols& c, int n, u64 (&w)[(R + 1) / 2]) {
#pragma unroll
    for (int q = 0; q < (R + 1) / 2; ++q) {
        const int s = L.st(2 * q);
        w[q] = s < c.k ? L.mask[(unsigned)n * (unsigned)c.W + (unsigned)(s >> 6)] : 0ull;
    }
}
// ... and the pair's two bits of it (bit 0: the pair's first state; a state >= k is not allowed)
template <int G, int R>
__device__ __forceinline__ void lane_words_bits(const LaneCtx<G, R>& L, const PmlCols& c, const u64 (&w)[(R + 1) / 2],
                                                unsigned (&b)[(R + 1) / 2]) {
#pragma unroll
    for (int q = 0; q < (R + 1) / 2; ++q) {
        const int s = L.st(2 * q);
        unsigned x = (unsigned)(w[q] >> (s & 63)) & 3u;
        if (s + 1 >= c.k) x &= 1u;
        b[q] = x;
    }
}
template <int R>
__device__ __forceinline__ void bits_to_vec(const unsigned (&b)[(R + 1) / 2], double (&v)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = ((b[r >> 1] >> (r & 1)) & 1u) ? 1.0 : 0.0;
}
template <int R>
__device__ __forceinline__ void bits_select_vec(const unsigned (&b)[(R + 1) / 2], double a, double e, double (&out)[R]) {
    const double c1 = a + e;
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = ((b[r >> 1] >> (r & 1)) & 1u) ? c1 : a;
}

template <int G, int R>
__device__ __forceinline__ bool bu_f81_unit_lean_w(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                   const PmlState& st, const UnitRegs& u, int cfc0, int cfc1) {
    constexpr int Q = (R + 1) / 2;
    const int n = u.n, fc = u.fc;
    const int nc = unit_nc(u.packed);
    // ---- loads
    u64 ow[Q];
    lane_words_issue<G, R>(L, c, n, ow);
    double ce[2], cs[2], vv[2][R];
    u64 cw[2][Q];
    i64 cbe[2];
    double te[2][2], ts[2][2];
    u64 tw[2][2][Q];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = fc + (j < nc ? j : 0);
        const int code = j < nc ? unit_code(u.packed, j) : 0;
        ce[j] = L.E[ch];
        if (code != 1) {
            lane_words_issue<G, R>(L, c, ch, cw[j]);
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) cw[j][q] = 0ull;
        }
        cs[j] = code <= 1 ? L.S[ch] : 0.0;
        cbe[j] = code == 1 ? L.be[ch] : 0;
        if (code == 1) {
            node_load_vec<G, R>(L, c, L.bu, ch, vv[j]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) vv[j][r] = 0.0;
        }
        const int cfc = j == 0 ? cfc0 : cfc1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            te[j][q] = 0.0;
            ts[j][q] = 0.0;
#pragma unroll
            for (int x = 0; x < Q; ++x) tw[j][q][x] = 0ull;
            if (code >= 2 && q < code - 1) {
                te[j][q] = L.E[cfc + q];
                ts[j][q] = L.S[cfc + q];
                lane_words_issue<G, R>(L, c, cfc + q, tw[j][q]);
            }
        }
    }
    // ---- arithmetic (bu_f81_unit_seq's operations)
    double acc[R];
    {
        unsigned ob[Q];
        lane_words_bits<G, R>(L, c, ow, ob);
        bits_to_vec<R>(ob, acc);
    }
    i64 esum = 0;
    double lob = 1.0;
    bool bounded = true;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j < nc) {
            const int code = unit_code(u.packed, j);
            const double e = ce[j];
            double msg[R];
            double a;
            unsigned cb[Q];
            lane_words_bits<G, R>(L, c, cw[j], cb);
            if (code == 0) {
                a = (1.0 - e) * cs[j];
                bits_select_vec<R>(cb, a, e, msg);
            } else {
                double v[R];
                double s_child;
                if (code == 1) {
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = vv[j][r];
                    esum += cbe[j];
                    s_child = cs[j];
                    bounded = false;
                } else {
                    bits_to_vec<R>(cb, v);
                    double amin = 1.0;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q < code - 1) {
                            const double ta = (1.0 - te[j][q]) * ts[j][q];
                            double tmsg[R];
                            unsigned tb[Q];
                            lane_words_bits<G, R>(L, c, tw[j][q], tb);
                            bits_select_vec<R>(tb, ta, te[j][q], tmsg);
                            amin *= ta;
#pragma unroll
                            for (int r = 0; r < R; ++r) v[r] *= tmsg[r];
                        }
                    }
                    if (!(amin >= 0x1p-190)) esum += lazy_rescale<G, R>(v);
                    s_child = pi_dot<G, R>(L, v);
                    if (L.g == 0) L.S[fc + j] = s_child;
                }
                a = (1.0 - e) * s_child;
#pragma unroll
                for (int r = 0; r < R; ++r) msg[r] = a + e * v[r];
            }
            lob *= a;
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] *= msg[r];
            if (j == 1 || j == nc - 1) {
                if (!bounded || !(lob >= 0x1p-190)) {
                    const int ex = lazy_rescale<G, R>(acc);
                    esum += ex;
                    if (ex != 0) bounded = false;
                }
            }
        }
    }
    const double s = pi_dot<G, R>(L, acc);
    if (!(s > 0.0)) {
        bool nz = false;
#pragma unroll
        for (int r = 0; r < R; ++r) nz |= acc[r] != 0.0 && L.st(r) < c.k;
        if (!group_any<G>(nz)) return false;
    }
    if (L.g == 0) {
        L.S[n] = s;
        L.be[n] = esum;
    }
    node_store_vec<G, R>(L, c, L.bu, n, acc);
    return true;
}

template <int G, int R, bool JOINT>
__device__ __forceinline__ bool bu_f81_unit_is_fast(const PmlCols& c, const UnitRegs& u) {
    return !JOINT && Gather<G>::enabled && c.W == 1 && unit_is_fast_bu<G, true>(u.packed);
}

// ---------------------------------------------------------------------------------------------------------------------
// bottom-up level kernel. One unit = (stored internal node of the level, column).
// replaces calc_node_bu_likelihood (pastml/ml.py:124-148) for the F81 family.
// JOINT: Pupko's max / arg-max variant (never fused: t.kind is null there).
// ---------------------------------------------------------------------------------------------------------------------
// One bottom-up unit: node n of the current level, this lane group's column.
template <int G, int R, bool JOINT>
__device__ __forceinline__ void bu_f81_unit_seq(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                const PmlState& st, const UnitRegs& u);

#define PML_SHAPE_KEY(nc, c0, c1) ((nc) | ((c0) << 8) | ((c1) << 11))
template <int G, int R, bool JOINT, bool LEAN = false, bool SHAPES = false>
__device__ __forceinline__ void bu_f81_unit(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                            const PmlState& st, const UnitRegs& u) {
    if (LEAN && !JOINT && G < 8 && c.W == 1 && unit_is_lean(u.packed)) {
        bool ok;
        const int key = u.packed & (15 | (63 << 8));
        const int key0 = __builtin_amdgcn_readfirstlane(key);
        // (only where every unit of the wave has the shape: a wave of mixed shapes would run one variant after the other)
        if (SHAPES && __ballot(key != key0) == 0ull) {
            switch (key0) {
                case PML_SHAPE_KEY(2, 1, 1): ok = bu_f81_unit_lean<G, R, 1, 1>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 3, 3): ok = bu_f81_unit_lean<G, R, 3, 3>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 1, 0): ok = bu_f81_unit_lean<G, R, 1, 0>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 0, 1): ok = bu_f81_unit_lean<G, R, 0, 1>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 1, 3): ok = bu_f81_unit_lean<G, R, 1, 3>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 3, 1): ok = bu_f81_unit_lean<G, R, 3, 1>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 0, 3): ok = bu_f81_unit_lean<G, R, 0, 3>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 3, 0): ok = bu_f81_unit_lean<G, R, 3, 0>(L, t, c, st, u); break;
                default: ok = bu_f81_unit_lean<G, R>(L, t, c, st, u); break;
            }
        } else {
            ok = bu_f81_unit_lean<G, R>(L, t, c, st, u);
        }
        if (ok) return;
    }
    if (SHAPES && LEAN && !JOINT && G >= 2 && G < 8 && c.W == 1 && !c.no_wide_lean) {
        // (only where none of the wave's remaining units needs the sequential path: a wave that ran both would lose;
        // not for one-lane units, k <= 2: there the extra code costs binary trees 4 %)
        const bool poly = unit_is_lean_poly(u.packed);
        if (__ballot(!poly) == 0ull) {
            if (bu_f81_unit_lean_poly<G, R>(L, t, c, st, u)) return;
        }
    }
    if constexpr (G >= 32) {
        // (masks of several words; where the tips of cherry children 0 and 1 start: units of 8 lanes and more keep the
        // descriptor's entry of the child their lane gathers for -- lanes 0 and GC hold the two)
        if (LEAN && !JOINT && c.W > 1 && !c.no_wide_lean && unit_is_lean(u.packed)) {
            const int cfc0 = __shfl(u.cfc, L.group_base, 64), cfc1 = __shfl(u.cfc, L.group_base + Gather<G>::GC, 64);
            if (bu_f81_unit_lean_w<G, R>(L, t, c, st, u, cfc0, cfc1)) return;
        }
    }
    if (bu_f81_unit_is_fast<G, R, JOINT>(c, u)) {
        BuLoads<R> ld;
        bu_f81_issue<G, R, true>(L, c, u, ld);
        if (bu_f81_unit_fast<G, R, true>(L, t, c, st, u, ld)) return;
    }
    bu_f81_unit_seq<G, R, JOINT>(L, t, c, st, u);
}

// the sequential path: any number of children, multi-word masks, the joint variant, per-child zero check
template <int G, int R, bool JOINT>
__device__ __forceinline__ void bu_f81_unit_seq(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                const PmlState& st, const UnitRegs& u) {
    const int n = u.n;

    double acc[R];
    node_mask_vec<G, R>(L, c, n, acc);
    i64 esum = 0;
    // topology from the descriptor (no chain node -> first_child -> kind -> the cherry's children through memory)
    const int fc = u.fc;
    int nc = unit_nc(u.packed);
    if (nc == 15) nc = t.n_children[n];  // the descriptor counts up to 14
    for (int j = 0; j < nc; ++j) {
        const int ch = fc + j;
        const double e = L.E[ch];
        const ChildDesc cd = unit_child<G>(t, u, j, ch);
        const int kd = cd.kind;
        double v[R];
        double s_child = 0.0;
        if (kd == PML_KIND_TIP) {
            node_mask_vec<G, R>(L, c, ch, v);
            if (!JOINT) s_child = L.S[ch];
        } else if (kd == PML_KIND_STORED) {
            node_load_vec<G, R>(L, c, L.bu, ch, v);
            esum += L.be[ch];
            if (!JOINT) s_child = L.S[ch];
        } else if (JOINT) {
            // cherry child of the fused joint sweep: the operations of this function for a node whose children are
            // tips (mask, then per tip: message, product, zero check, rescaling; the tip's arg-max row), in registers
            node_mask_vec<G, R>(L, c, ch, v);
            const int fc2 = cd.tips_known ? cd.fc2 : t.first_child[ch];
            const int nc2 = cd.tips_known ? cd.nc2 : t.n_children[ch];
            for (int q = 0; q < nc2; ++q) {
                const int tip = fc2 + q;
                double tv[R], tmsg[R];
                int tj[R];
                node_mask_vec<G, R>(L, c, tip, tv);
                f81_joint_message<G, R>(L, c, L.E[tip], tv, tmsg, tj);
                bool tnz = false;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    v[r] *= fmax(tmsg[r], 0.0);
                    tnz |= v[r] != 0.0;
                }
                f81_store_table<G, R>(L, c, f81_table_row<G, R>(L, t, c, st, tip), tj);
                if (!group_any<G>(tnz)) {
                    if (L.g == 0)
                        atomicMin(&st.err[L.col], ((u64)(unsigned)t.post_rank[ch] << 32) | (u64)(unsigned)tip);
                }
                esum += lazy_rescale<G, R>(v);
            }
        } else {
            i64 ce;
            f81_cherry_vector<G, R>(L, t, c, st, ch, v, ce, true, cd.tips_known, cd.fc2, cd.nc2);
            esum += ce;
            s_child = pi_dot<G, R>(L, v);
            if (L.g == 0) L.S[ch] = s_child;
        }
        if (!JOINT) {
            f81_absorb_child<G, R>(L, t, st, n, ch, e, s_child, v, acc, true);
            // stored vectors and cherry vectors are in band, tips are 0/1: two factors cannot leave the
            // double range, so the band is checked every second child and at the end
            if ((j & 1) == 1 || j == nc - 1) esum += lazy_rescale<G, R>(acc);
        } else {
            double msg[R];
            int jj[R];
            f81_joint_message<G, R>(L, c, e, v, msg, jj);
            bool nz = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r] *= fmax(msg[r], 0.0);
                nz |= acc[r] != 0.0;
            }
            // altered nodes get their tables rewritten w.r.t. their initial masks (ml.py:408-428)
            if (c.masks_init != nullptr) {
                const size_t mo_ = ((size_t)L.col * t.N + ch) * c.W;
                const u64* mi = c.masks_init + mo_;
                const u64* mc = c.masks + mo_;
                bool altered = false;
                for (int w_ = 0; w_ < c.W; ++w_) altered |= (mi[w_] != mc[w_]);
                if (altered) {
                    const int fa = first_allowed(mi, c.W);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int a = jj[r];
                        if (!((mi[a >> 6] >> (a & 63)) & 1ull)) jj[r] = fa;
                    }
                }
            }
            f81_store_table<G, R>(L, c, f81_table_row<G, R>(L, t, c, st, ch), jj);
            if (!group_any<G>(nz)) {
                if (L.g == 0)
                    atomicMin(&st.err[L.col], ((u64)(unsigned)t.post_rank[n] << 32) | (u64)(unsigned)ch);
            }
            esum += lazy_rescale<G, R>(acc);
        }
    }
    if (!JOINT) {
        const double s = pi_dot<G, R>(L, acc);
        if (L.g == 0) L.S[n] = s;
    }
    node_store_vec<G, R>(L, c, L.bu, n, acc);
    if (L.g == 0) L.be[n] = esum;
}

// Level kernel.  Marginal sweeps with the lane-parallel gather run a two-stage software pipeline per wave: while unit i
// is computed, every load of unit i + 1 (scalars, own mask, stored children's vectors) and the descriptor of unit
// i + 2 are in flight, so a wave waits for memory once per sweep of its units instead of once per unit (the level that
// rebuilds cherries spent 65 % of its wave cycles in s_waitcnt before).  The two halves of the loop body ping-pong
// between two register sets instead of copying one into the other.
// (Occupancy: three waves per SIMD at four states per lane.  The joint variants that stream vectors then spill 12 bytes
// per lane -- the only scratch in a level kernel; without the spill, at two waves, the cfg4-size joint sweep takes
// 8.9 instead of 7.85 ms: measured in round 3, kept.)
template <int G, int R, bool JOINT, bool VEC>
__global__ void __launch_bounds__(PML_BLOCK) __attribute__((amdgpu_waves_per_eu(R >= 8 ? 2 : 3, R >= 8 ? 2 : 4)))
bu_f81_kernel(PmlTree t, PmlCols c, PmlState st, const PmlUnit* __restrict__ units, int n_level) {
    if (!JOINT && !column_active(c, blockIdx.y)) return;
    constexpr int UW = 64 / G;  // units per wave
    const int wave = threadIdx.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    int idx = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW + sub;
    int base = idx - sub;  // wave-uniform trip count; whole groups drop out together
    // the joint variant takes the pipeline unless altered nodes need their tables rewritten (masks_init, rare)
    if (Gather<G>::enabled && c.W == 1 && (!JOINT || c.masks_init == nullptr)) {
        UnitRegs ua = load_unit<G>(units, idx < n_level ? idx : 0, L.g);
        UnitRegs ub = load_unit<G>(units, idx + stride < n_level ? idx + stride : 0, L.g);
        BuLoads<R> la, lb;
        bu_f81_issue<G, R, VEC, JOINT>(L, c, ua, la);
        while (base < n_level) {
            {  // compute a; loads of b and the descriptor after b in flight
                const int i2 = idx + 2 * stride;
                const UnitRegs un = load_unit<G>(units, i2 < n_level ? i2 : 0, L.g);
                bu_f81_issue<G, R, VEC, JOINT>(L, c, ub, lb);
                if (idx < n_level) {
                    if (!unit_is_fast_bu<G, VEC>(ua.packed) || !bu_f81_unit_fast<G, R, VEC, JOINT>(L, t, c, st, ua, la))
                        bu_f81_unit_seq<G, R, JOINT>(L, t, c, st, ua);
                }
                ua = un;
                idx += stride;
                base += stride;
            }
            if (base >= n_level) break;
            {  // compute b; loads of a (the unit after b) in flight
                const int i2 = idx + 2 * stride;
                const UnitRegs un = load_unit<G>(units, i2 < n_level ? i2 : 0, L.g);
                bu_f81_issue<G, R, VEC, JOINT>(L, c, ua, la);
                if (idx < n_level) {
                    if (!unit_is_fast_bu<G, VEC>(ub.packed) || !bu_f81_unit_fast<G, R, VEC, JOINT>(L, t, c, st, ub, lb))
                        bu_f81_unit_seq<G, R, JOINT>(L, t, c, st, ub);
                }
                ub = un;
                idx += stride;
                base += stride;
            }
        }
        return;
    }
    UnitRegs cur = load_unit<G>(units, idx < n_level ? idx : 0, L.g);
    for (; base < n_level; base += stride) {  // the next descriptor is in flight during the unit
        const int nxt_idx = idx + stride;
        const UnitRegs nxt = load_unit<G>(units, nxt_idx < n_level ? nxt_idx : 0, L.g);
        if (idx < n_level) bu_f81_unit<G, R, JOINT, true>(L, t, c, st, cur);
        cur = nxt;
        idx = nxt_idx;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-level units (round 3).  Where a tree is locally balanced -- a stored node n with two stored children, each of
// them the parent of two cherries of two tips -- the level schedule writes the children's vectors in one launch and
// reads them straight back in the next (bottom-up), and writes the children's posterior rows in one launch and reads
// them back as "parent" rows in the next (top-down): on a balanced binary tree that is a fifth of the bottom-up bytes
// and a twelfth of the top-down bytes.  A two-level unit is (n, column): it runs the units of both children and then
// n's own unit on what they left in registers.  The three units are the level kernels' bodies, called with the same
// arguments in the same lane shape: the same bits.  The children of such nodes ("absorbed") and the nodes themselves
// leave the level lists (pml_tree_upload: rest lists); everything still lands in memory where the other sweeps,
// the downloads and the other schedules expect it.
// Descriptor of a two-level unit: n, fc (children fc, fc + 1), cfc[0] = g0 (the four cherries g0 .. g0 + 3),
// pad = t0 (the eight tips t0 .. t0 + 7).
// ---------------------------------------------------------------------------------------------------------------------
#define PML_PACKED_PAIR ((3 << 11) | (3 << 8) | (3 << 4) | 2)        // two children, both cherries of two tips
#define PML_PACKED_TWO_STORED ((1 << 11) | (1 << 8) | (3 << 4) | 2)  // two children, both stored nodes

struct SuperRegs {
    int n, fc, g0, t0;
};

__device__ __forceinline__ SuperRegs load_super(const PmlUnit* __restrict__ units, int idx) {
    const int4 h = *reinterpret_cast<const int4*>(units + idx);
    SuperRegs s;
    s.n = h.x;
    s.fc = h.y;
    s.t0 = h.w;
    s.g0 = units[idx].cfc[0];
    return s;
}

// the descriptor registers child j of a two-level unit would get from load_unit (G >= 8)
template <int G>
__device__ __forceinline__ UnitRegs super_child(const SuperRegs& s, int j, int g) {
    UnitRegs u;
    u.n = s.fc + j;
    u.fc = s.g0 + 2 * j;
    u.packed = PML_PACKED_PAIR;
    u.cfc = s.t0 + 4 * j + 2 * ((g / Gather<G>::GC) & 1);
    u.cfc1 = u.cfc2 = u.cfc3 = 0;
    return u;
}

__device__ __forceinline__ UnitRegs super_own(const SuperRegs& s) {
    UnitRegs u;
    u.n = s.n;
    u.fc = s.fc;
    u.packed = PML_PACKED_TWO_STORED;
    u.cfc = u.cfc1 = u.cfc2 = u.cfc3 = 0;
    return u;
}

template <int R>
struct SuperLoads {
    BuLoads<R> c[2];  // what the two children's own units read
    double e;         // lane j < 2: E of child j
    u64 own;          // the node's own mask word
};

template <int G, int R>
__device__ __forceinline__ void bu_f81_super_issue(const LaneCtx<G, R>& L, const SuperRegs& s, SuperLoads<R>& ld) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const UnitRegs u = super_child<G>(s, j, L.g);
        f81_gather_issue<G, R>(L, u, ld.c[j].cl, ld.c[j].tl);
        ld.c[j].own = L.mask[(unsigned)u.n];
    }
    ld.e = L.E[s.fc + (L.g & 1)];
    ld.own = L.mask[(unsigned)s.n];
}

// one wait for exactly this unit's loads (see bu_loads_arrived); S and the exponent word of a cherry are not read
template <int R>
__device__ __forceinline__ void super_loads_arrived(SuperLoads<R>& ld) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
        asm volatile("" : "+v"(ld.c[j].cl.e), "+v"(ld.c[j].cl.mask), "+v"(ld.c[j].tl.e), "+v"(ld.c[j].tl.s),
                          "+v"(ld.c[j].tl.mask), "+v"(ld.c[j].own));
    asm volatile("" : "+v"(ld.e), "+v"(ld.own));
}

// false: some vector came out all zero (nothing of the failing unit was stored); the caller repeats the three units
// on the sequential path, which names the pair the reference would.
template <int G, int R>
__device__ __forceinline__ bool bu_f81_super_unit(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                  const PmlState& st, const SuperRegs& s, SuperLoads<R>& ld) {
    super_loads_arrived<R>(ld);
    const u64 kbits = state_bits(c.k);
    bool ones = c.k == G * R;  // (a padding entry must stay 0, and only a mask makes it so)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f81_gather_finish<G, R>(L, ld.c[j].cl, ld.c[j].tl);
        ones = ones && (ld.c[j].own & kbits) == kbits && ld.c[j].cl.mask == kbits;
    }
    const bool full = __all(ones);  // wave-uniform choice of the body, as in bu_f81_unit_fast
    BuResult<R> res[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const UnitRegs u = super_child<G>(s, j, L.g);
        // (the children's vectors are not written: the top-down two-level unit rebuilds them from the tips, as both
        // sweeps rebuild cherries; pi . v and the exponent are -- 16 bytes instead of 8 ks + 16)
        const bool ok = full ? bu_f81_marg_body<G, R, false, true, true>(L, t, c, st, u, ld.c[j], &res[j], false)
                             : bu_f81_marg_body<G, R, false, false, true>(L, t, c, st, u, ld.c[j], &res[j], false);
        if (!ok) return false;
    }
    // the node's own unit, as the level above would run it (bu_f81_unit_fast, streaming level): the children's vectors,
    // pi . v and exponents from registers instead of from memory
    BuLoads<R> top;
    top.cl.e = ld.e;
    top.cl.s = (L.g & 1) ? res[1].s : res[0].s;
    top.cl.be = (L.g & 1) ? res[1].e : res[0].e;
    top.cl.mask = kbits;
    top.tl.e = top.tl.s = top.tl.a = 0.0;
    top.tl.mask = 0ull;
    top.own = ld.own;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        top.v0[r] = res[0].v[r];
        top.v1[r] = res[1].v[r];
    }
    return bu_f81_marg_body<G, R, true, false>(L, t, c, st, super_own(s), top);
}

template <int G, int R>
__device__ __forceinline__ void bu_f81_super_seq(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                 const PmlState& st, const SuperRegs& s) {
    bu_f81_unit_seq<G, R, false>(L, t, c, st, super_child<G>(s, 0, L.g));
    bu_f81_unit_seq<G, R, false>(L, t, c, st, super_child<G>(s, 1, L.g));
    __threadfence();  // the node's unit reads what the wave has just stored
    bu_f81_unit_seq<G, R, false>(L, t, c, st, super_own(s));
}

// The same two-stage software pipeline as bu_f81_kernel: the loads of unit i + 1 and the descriptor of unit i + 2 are
// in flight while unit i is computed.
// (two waves per SIMD in every shape: at three, four states per lane spill 64 - 72 bytes)
template <int G, int R>
__global__ void __launch_bounds__(PML_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
bu_f81_super_kernel(PmlTree t, PmlCols c, PmlState st, const PmlUnit* __restrict__ units, int n_level) {
    constexpr int UW = 64 / G;
    const int wave = threadIdx.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    int idx = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW + sub;
    int base = idx - sub;
    SuperRegs ua = load_super(units, idx < n_level ? idx : 0);
    SuperRegs ub = load_super(units, idx + stride < n_level ? idx + stride : 0);
    SuperLoads<R> la, lb;
    bu_f81_super_issue<G, R>(L, ua, la);
    while (base < n_level) {
        {
            const int i2 = idx + 2 * stride;
            const SuperRegs un = load_super(units, i2 < n_level ? i2 : 0);
            bu_f81_super_issue<G, R>(L, ub, lb);
            if (idx < n_level) {
                if (!bu_f81_super_unit<G, R>(L, t, c, st, ua, la)) bu_f81_super_seq<G, R>(L, t, c, st, ua);
            }
            ua = un;
            idx += stride;
            base += stride;
        }
        if (base >= n_level) break;
        {
            const int i2 = idx + 2 * stride;
            const SuperRegs un = load_super(units, i2 < n_level ? i2 : 0);
            bu_f81_super_issue<G, R>(L, ua, la);
            if (idx < n_level) {
                if (!bu_f81_super_unit<G, R>(L, t, c, st, ub, lb)) bu_f81_super_seq<G, R>(L, t, c, st, ub);
            }
            ub = un;
            idx += stride;
            base += stride;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// top-down + marginal likelihoods + posteriors.
//
// f81_finish_child: given prod = TD_parent o BU_parent (exponent pe) and the child's own data, divides the child's
// message out of the parent (ml.py:279-283), pushes the result through the child's branch (ml.py:287-289), forms the
// marginal likelihoods pi o mask o BU o TD (ml.py:456-460) and stores the posteriors (ml.py:498-500).
// The parent's prod arrives normalised (f81_parent_prod), so the child's TD vector stays within the bottom-up band of
// the child and needs no rescaling of its own.
// ---------------------------------------------------------------------------------------------------------------------
template <int G, int R>
__device__ __forceinline__ void f81_finish_child(const LaneCtx<G, R>& L, const PmlCols& c, const double (&prod)[R],
                                                 i64 pe, int ch, double e, double s_child, i64 bec,
                                                 const double (&v)[R], bool full_mask, const double (&mb)[R],
                                                 double (&tdc)[R], i64& xe, double (&lh)[R], double& lsum, i64& le,
                                                 int slot = -1) {
    const double a = (1.0 - e) * s_child;
    double x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        // a zero message entry comes with a zero in prod (it is one of prod's factors): 0 * 2^1022 = 0
        const double cn = fmax(a + e * v[r], 0x1p-1022);
        x[r] = prod[r] * rcp1(cn);
    }
    xe = pe - bec;
    const double b = (1.0 - e) * pi_dot<G, R>(L, x);
#pragma unroll
    for (int r = 0; r < R; ++r) tdc[r] = b + e * x[r];  // >= 0 by construction (ml.py:289's clamp is a no-op)
    double lhs = 0.0;
    if (full_mask) {  // every state allowed (the usual case for an internal node): pi * 1.0 = pi, same bits
#pragma unroll
        for (int r = 0; r < R; ++r) {
            lh[r] = v[r] * tdc[r] * L.pi_r[r];
            lhs += lh[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            lh[r] = v[r] * tdc[r] * (L.pi_r[r] * mb[r]);
            lhs += lh[r];
        }
    }
    lhs = group_sum<G>(lhs);
    const int lex = (lhs > 0.0 && !isinf(lhs)) ? exponent_of(lhs) : 0;
    const double inv = rcp1(lhs);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        // correctly rounded lh / lhs (one residual step)
        const double q = lh[r] * inv;
        lh[r] = fma(fma(-lhs, q, lh[r]), inv, q);
    }
    const bool staged = post_row<G, R>(L, c, slot, ch, lh);
    lsum = __builtin_ldexp(lhs, -lex);
    le = xe + bec + lex;
    post_scalars<G, R>(L, staged ? slot : -1, ch, lsum, le);
}

// A tip below a parent with prod = TD_parent o BU_parent.  Observed tips (one allowed state s) in closed form:
// with c0 = (1 - e) pi_s, c1 = c0 + e the divided vector is X_i = prod_i / (i == s ? c1 : c0), hence
// pi . X = (P - pi_s prod_s) / c0 + pi_s prod_s / c1 with P = pi . prod (computed once per parent), the tip's
// TD_s = (1 - e) pi . X + e prod_s / c1, its marginal likelihood vector is TD_s pi_s at s and 0 elsewhere, and its
// posterior is exactly the unit vector (what lh / lh.sum() gives in the reference, ml.py:500).
// Observed-tip closed form / general tip, given the tip's data (single mask word without bits >= k: k <= 64).
template <int G, int R>
__device__ __forceinline__ void f81_finish_tip_word(const LaneCtx<G, R>& L, const PmlCols& c, const double (&prod)[R],
                                                    i64 pe, double& P, bool& have_P, int tip, u64 word, double e,
                                                    double pis, int slot = -1) {
    if (__popcll(word) == 1) {
        if (!have_P) {
            P = pi_dot<G, R>(L, prod);
            have_P = true;
        }
        // unit vector of the observed state: the posterior, and the tool that picks prod_s (sum of zeros and one term)
        double oh[R];
        clean_word_to_vec<G, R>(L, c, word, oh);
        double pick = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) pick = fma(oh[r], prod[r], pick);
        const double ps = group_sum<G>(pick);
        const double q = pis * ps;
        double c0 = (1.0 - e) * pis;
        const double c1 = c0 + e;
        if (!(c0 > 0.0)) c0 = 1.0;
        const double r1 = rcp1(c1);
        const double sx = (P - q) * rcp1(c0) + q * r1;
        const double tds = (1.0 - e) * sx + e * (ps * r1);
        const double lhs = tds * pis;
        const bool ok = lhs > 0.0 && !isinf(lhs);
        const int lex = ok ? exponent_of(lhs) : 0;
        if (!ok) {
#pragma unroll
            for (int r = 0; r < R; ++r) oh[r] = __builtin_nan("");
        }
        const bool staged = post_onehot<G, R>(L, c, slot, tip, (int)__builtin_ctzll(word), oh, ok);
        post_scalars<G, R>(L, staged ? slot : -1, tip, __builtin_ldexp(lhs, -lex), pe + lex);
    } else {
        double mt[R], tdt[R], pt[R], lt;
        clean_word_to_vec<G, R>(L, c, word, mt);
        i64 xt, et;
        f81_finish_child<G, R>(L, c, prod, pe, tip, e, pis, 0, mt, false, mt, tdt, xt, pt, lt, et, slot);
    }
}

// A tip below a parent with prod = TD_parent o BU_parent.  Observed tips (one allowed state s) in closed form:
// with c0 = (1 - e) pi_s, c1 = c0 + e the divided vector is X_i = prod_i / (i == s ? c1 : c0), hence
// pi . X = (P - pi_s prod_s) / c0 + pi_s prod_s / c1 with P = pi . prod (computed once per parent), the tip's
// TD_s = (1 - e) pi . X + e prod_s / c1, its marginal likelihood vector is TD_s pi_s at s and 0 elsewhere, and its
// posterior is exactly the unit vector (what lh / lh.sum() gives in the reference, ml.py:500).
template <int G, int R>
__device__ __forceinline__ void f81_finish_tip(const LaneCtx<G, R>& L, const PmlCols& c, const double (&prod)[R], i64 pe,
                                               double& P, bool& have_P, int tip, int slot = -1) {
    const double e = L.E[tip];
    const double pis = L.S[tip];
    if (c.W == 1) {
        f81_finish_tip_word<G, R>(L, c, prod, pe, P, have_P, tip, L.mask[(unsigned)tip] & state_bits(c.k), e, pis,
                                  slot);
    } else {
        // k > 64: general path (the closed form above would need the state's word; not worth a special case)
        double mt[R], tdt[R], pt[R], lt;
        node_mask_vec<G, R>(L, c, tip, mt);
        i64 xt, et;
        f81_finish_child<G, R>(L, c, prod, pe, tip, e, pis, 0, mt, false, mt, tdt, xt, pt, lt, et, slot);
    }
}

template <int G, int R>
__device__ __forceinline__ void td_f81_fast_children(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                     const PmlState& st, const UnitRegs& u, const double (&prod)[R],
                                                     i64 pe, const ChildLane& cl, const TipLane& tl, double (&vn)[R]);

template <int G, int R>
__device__ __forceinline__ void td_f81_unit_fast(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                 const PmlState& st, const UnitRegs& u) {
    const int p = u.n, fc = u.fc;
    double prod[R];
    i64 pe;
    f81_parent_prod<G, R>(L, c, p, prod, pe);
    ChildLane cl;
    TipLane tl;
    f81_gather_issue<G, R>(L, u, cl, tl);
    double vn[R];
    if (unit_code(u.packed, 0) == 1) node_load_vec<G, R>(L, c, L.bu, fc, vn);
    f81_gather_finish<G, R>(L, cl, tl);
    td_f81_fast_children<G, R>(L, t, c, st, u, prod, pe, cl, tl, vn);
}

// The children of a fast unit's node, given prod = TD o BU of the node (exponent pe), the gathered scalars and -- if
// child 0 is a stored node -- its bottom-up vector in vn.
template <int G, int R>
__device__ __forceinline__ void td_f81_fast_children(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                     const PmlState& st, const UnitRegs& u, const double (&prod)[R],
                                                     i64 pe, const ChildLane& cl, const TipLane& tl, double (&vn)[R]) {
    constexpr int GC = Gather<G>::GC;
    const int fc = u.fc;
    const int nc = unit_nc(u.packed);
    double P = 0.0;
    bool have_P = false;
    for (int jx = 0; jx < nc; ++jx) {
        const int src = L.group_base + jx;
        const int ch = fc + jx;
        const int code = unit_code(u.packed, jx);
        const double e = __shfl(cl.e, src, 64);
        const u64 word = __shfl(cl.mask, src, 64);
        double v[R];
        if (code == 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = vn[r];
        }
        if (jx + 1 < nc && unit_code(u.packed, jx + 1) == 1) node_load_vec<G, R>(L, c, L.bu, ch + 1, vn);
        if (code == 0) {
            f81_finish_tip_word<G, R>(L, c, prod, pe, P, have_P, ch, word, e, __shfl(cl.s, src, 64));
            continue;
        }
        double mb[R], tdc[R], po[R], ls;
        const bool full = word == state_bits(c.k);
        if (!full) clean_word_to_vec<G, R>(L, c, word, mb);
        i64 xe, le;
        if (code == 1) {
            f81_finish_child<G, R>(L, c, prod, pe, ch, e, __shfl(cl.s, src, 64), __shfl(cl.be, src, 64), v, full, mb,
                                   tdc, xe, po, ls, le);
            if (st.td != nullptr) {
                node_store_vec<G, R>(L, c, L.td, ch, tdc);
                if (L.g == 0) L.te[ch] = xe;
            }
        } else {
            i64 bec;
            const int cfc = __shfl(u.cfc, L.group_base + jx * GC, 64);
            const int cnc = code - 1;
            f81_cherry_from_lanes<G, R>(L, c, cl, tl, jx, cnc, v, bec);
            const double s_child = __shfl(cl.s, src, 64);  // pi . v, stored by the bottom-up sweep
            f81_finish_child<G, R>(L, c, prod, pe, ch, e, s_child, bec, v, full, mb, tdc, xe, po, ls, le);
            // same rounding as f81_parent_prod, so that the schedule without cherry fusion gives the same bits
            double prod2[R];
#pragma unroll
            for (int r = 0; r < R; ++r) prod2[r] = po[r] * (ls * L.ipi_r[r]);
            const i64 pe2 = le;
            double P2 = 0.0;
            bool have_P2 = false;
            for (int q = 0; q < cnc; ++q) {
                const int ts = L.group_base + jx * GC + q;
                f81_finish_tip_word<G, R>(L, c, prod2, pe2, P2, have_P2, cfc + q, __shfl(tl.mask, ts, 64),
                                          __shfl(tl.e, ts, 64), __shfl(tl.s, ts, 64));
            }
        }
    }
}

// Lean top-down unit for the kernels that walk several levels in one launch (see bu_f81_unit_lean): every load the unit
// can need -- the parent's posterior row and scalars, both children's scalars and vectors, the scalars of up to two tips
// under each -- is issued before any value is used; then the operations of td_f81_unit's sequential path in its order.
template <int G, int R, int SC0 = PML_SHAPE_ANY, int SC1 = PML_SHAPE_ANY>
__device__ __forceinline__ void td_f81_unit_lean(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                 const PmlState& st, const UnitRegs& u) {
    constexpr bool ANY = SC0 == PML_SHAPE_ANY;  // (compile-time shapes: see bu_f81_unit_lean)
    const int p = u.n, fc = u.fc;
    const int nc = ANY ? unit_nc(u.packed) : (SC1 < 0 ? 1 : 2);
    const u64 kbits = state_bits(c.k);
    // ---- loads
    double po[R];
    node_load_vec<G, R>(L, c, L.post, p, po);
    const double ls = L.lhsum[p];
    const i64 pe = L.lhe[p];
    double ce[2], cs[2], vv[2][R];
    u64 cm[2];
    i64 cbe[2];
    double te[2][2], ts[2][2];
    u64 tm[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = fc + (j < nc ? j : 0);
        const int code = j < nc ? (ANY ? unit_code(u.packed, j) : (j == 0 ? SC0 : SC1)) : 0;
        ce[j] = L.E[ch];
        cm[j] = L.mask[(unsigned)ch];
        cs[j] = L.S[ch];
        cbe[j] = code == 1 ? L.be[ch] : 0;
        if (code == 1) {
            node_load_vec<G, R>(L, c, L.bu, ch, vv[j]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) vv[j][r] = 0.0;
        }
        const int cfc = j == 0 ? u.cfc : u.cfc1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool has = code >= 2 && q < code - 1;
            const int tip = has ? cfc + q : ch;
            te[j][q] = L.E[tip];
            ts[j][q] = has ? L.S[tip] : 0.0;
            tm[j][q] = L.mask[(unsigned)tip];
        }
    }
    // ---- arithmetic
    double prod[R];
#pragma unroll
    for (int r = 0; r < R; ++r) prod[r] = po[r] * (ls * L.ipi_r[r]);  // f81_parent_prod
    double P = 0.0;
    bool have_P = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j < nc) {
            const int ch = fc + j;
            const int code = ANY ? unit_code(u.packed, j) : (j == 0 ? SC0 : SC1);
            const double e = ce[j];
            if (code == 0) {
                f81_finish_tip_word<G, R>(L, c, prod, pe, P, have_P, ch, cm[j] & kbits, e, cs[j], j);
            } else {
                double mb[R], v[R], tdc[R], pc[R], lsc;
                clean_word_to_vec<G, R>(L, c, cm[j] & kbits, mb);
                i64 xe, le;
                if (code == 1) {
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = vv[j][r];
                    f81_finish_child<G, R>(L, c, prod, pe, ch, e, cs[j], cbe[j], v, false, mb, tdc, xe, pc, lsc, le, j);
                    if (st.td != nullptr) {
                        node_store_vec<G, R>(L, c, L.td, ch, tdc);
                        if (L.g == 0) L.te[ch] = xe;
                    }
                } else {
                    // cherry: its bottom-up vector again (f81_cherry_vector's operations), finished, then its tips
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = mb[r];
                    double amin = 1.0;
                    i64 bec = 0;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q < code - 1) {
                            const double ta = (1.0 - te[j][q]) * ts[j][q];
                            double tmsg[R];
                            word_select_vec<G, R>(L, c, tm[j][q] & kbits, ta, te[j][q], tmsg);
                            amin *= ta;
#pragma unroll
                            for (int r = 0; r < R; ++r) v[r] *= tmsg[r];
                        }
                    }
                    if (!(amin >= 0x1p-190)) bec = lazy_rescale<G, R>(v);
                    f81_finish_child<G, R>(L, c, prod, pe, ch, e, cs[j], bec, v, false, mb, tdc, xe, pc, lsc, le, j);
                    double prod2[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) prod2[r] = pc[r] * (lsc * L.ipi_r[r]);
                    double P2 = 0.0;
                    bool have_P2 = false;
                    const int cfc = j == 0 ? u.cfc : u.cfc1;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q < code - 1)
                            f81_finish_tip_word<G, R>(L, c, prod2, le, P2, have_P2, cfc + q, tm[j][q] & kbits, te[j][q],
                                                      ts[j][q], 2 + 2 * j + q);
                    }
                }
            }
        }
    }
}

// Lean top-down unit for polytomies (see bu_f81_unit_lean_poly): up to four children, cherries of up to four tips, the
// children two at a time -- every load of a pair before any of its values; the operations of td_f81_unit's sequential path
// in its order (every child is finished from the parent's product on its own).
template <int G, int R>
__device__ __forceinline__ void td_f81_unit_lean_poly(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                      const PmlState& st, const UnitRegs& u) {
    const int p = u.n, fc = u.fc;
    const int nc = unit_nc(u.packed);
    const u64 kbits = state_bits(c.k);
    double po[R];
    node_load_vec<G, R>(L, c, L.post, p, po);
    const double ls = L.lhsum[p];
    const i64 pe = L.lhe[p];
    double prod[R];
    double P = 0.0;
    bool have_P = false;
    for (int j0 = 0; j0 < nc; j0 += 2) {
        // ---- loads of the pair
        double ce[2], cs[2], vv[2][R];
        u64 cm[2];
        i64 cbe[2];
        double te[2][4], ts[2][4];
        u64 tm[2][4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = j0 + jj;
            const bool valid = j < nc;
            const int ch = fc + (valid ? j : j0);
            const int code = valid ? unit_code(u.packed, j) : 0;
            ce[jj] = L.E[ch];
            cm[jj] = L.mask[(unsigned)ch];
            cs[jj] = L.S[ch];
            cbe[jj] = code == 1 ? L.be[ch] : 0;
            if (code == 1) {
                node_load_vec<G, R>(L, c, L.bu, ch, vv[jj]);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) vv[jj][r] = 0.0;
            }
            const int cfc = j == 0 ? u.cfc : (j == 1 ? u.cfc1 : (j == 2 ? u.cfc2 : u.cfc3));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                te[jj][q] = 0.0;
                ts[jj][q] = 0.0;
                tm[jj][q] = 0ull;
                if (code >= 2 && q < code - 1) {
                    te[jj][q] = L.E[cfc + q];
                    ts[jj][q] = L.S[cfc + q];
                    tm[jj][q] = L.mask[(unsigned)(cfc + q)];
                }
            }
        }
        if (j0 == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) prod[r] = po[r] * (ls * L.ipi_r[r]);  // f81_parent_prod
        }
        // ---- arithmetic
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = j0 + jj;
            if (j < nc) {
                const int ch = fc + j;
                const int code = unit_code(u.packed, j);
                const double e = ce[jj];
                const int slot = j < 2 ? j : -1;  // (staging slots: children 0 and 1 and two tips of each, as on the sequential path)
                if (code == 0) {
                    f81_finish_tip_word<G, R>(L, c, prod, pe, P, have_P, ch, cm[jj] & kbits, e, cs[jj], slot);
                } else {
                    double mb[R], v[R], tdc[R], pc[R], lsc;
                    clean_word_to_vec<G, R>(L, c, cm[jj] & kbits, mb);
                    i64 xe, le;
                    if (code == 1) {
#pragma unroll
                        for (int r = 0; r < R; ++r) v[r] = vv[jj][r];
                        f81_finish_child<G, R>(L, c, prod, pe, ch, e, cs[jj], cbe[jj], v, false, mb, tdc, xe, pc, lsc, le, slot);
                        if (st.td != nullptr) {
                            node_store_vec<G, R>(L, c, L.td, ch, tdc);
                            if (L.g == 0) L.te[ch] = xe;
                        }
                    } else {
                        // cherry: its bottom-up vector again (f81_cherry_vector's operations), finished, then its tips
#pragma unroll
                        for (int r = 0; r < R; ++r) v[r] = mb[r];
                        double amin = 1.0;
                        i64 bec = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (q < code - 1) {
                                const double ta = (1.0 - te[jj][q]) * ts[jj][q];
                                double tmsg[R];
                                word_select_vec<G, R>(L, c, tm[jj][q] & kbits, ta, te[jj][q], tmsg);
                                amin *= ta;
#pragma unroll
                                for (int r = 0; r < R; ++r) v[r] *= tmsg[r];
                            }
                        }
                        if (!(amin >= 0x1p-190)) bec = lazy_rescale<G, R>(v);
                        f81_finish_child<G, R>(L, c, prod, pe, ch, e, cs[jj], bec, v, false, mb, tdc, xe, pc, lsc, le, slot);
                        double prod2[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) prod2[r] = pc[r] * (lsc * L.ipi_r[r]);
                        double P2 = 0.0;
                        bool have_P2 = false;
                        const int cfc = j == 0 ? u.cfc : (j == 1 ? u.cfc1 : (j == 2 ? u.cfc2 : u.cfc3));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (q < code - 1)
                                f81_finish_tip_word<G, R>(L, c, prod2, le, P2, have_P2, cfc + q, tm[jj][q] & kbits, te[jj][q],
                                                          ts[jj][q], (j < 2 && q < 2) ? 2 + 2 * j + q : -1);
                        }
                    }
                }
            }
        }
    }
}

// Lean top-down unit for masks of several words (see bu_f81_unit_lean_w): every load up front, then the operations of
// td_f81_unit's sequential path in its order -- tips on its general path (the closed form of observed tips needs the
// single word).
template <int G, int R>
__device__ __forceinline__ void td_f81_unit_lean_w(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                   const PmlState& st, const UnitRegs& u, int cfc0, int cfc1) {
    constexpr int Q = (R + 1) / 2;
    const int p = u.n, fc = u.fc;
    const int nc = unit_nc(u.packed);
    // ---- loads
    double po[R];
    node_load_vec<G, R>(L, c, L.post, p, po);
    const double ls = L.lhsum[p];
    const i64 pe = L.lhe[p];
    double ce[2], cs[2], vv[2][R];
    u64 cw[2][Q];
    i64 cbe[2];
    double te[2][2], ts[2][2];
    u64 tw[2][2][Q];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = fc + (j < nc ? j : 0);
        const int code = j < nc ? unit_code(u.packed, j) : 0;
        ce[j] = L.E[ch];
        lane_words_issue<G, R>(L, c, ch, cw[j]);
        cs[j] = L.S[ch];
        cbe[j] = code == 1 ? L.be[ch] : 0;
        if (code == 1) {
            node_load_vec<G, R>(L, c, L.bu, ch, vv[j]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) vv[j][r] = 0.0;
        }
        const int cfc = j == 0 ? cfc0 : cfc1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            te[j][q] = 0.0;
            ts[j][q] = 0.0;
#pragma unroll
            for (int x = 0; x < Q; ++x) tw[j][q][x] = 0ull;
            if (code >= 2 && q < code - 1) {
                te[j][q] = L.E[cfc + q];
                ts[j][q] = L.S[cfc + q];
                lane_words_issue<G, R>(L, c, cfc + q, tw[j][q]);
            }
        }
    }
    // ---- arithmetic
    double prod[R];
#pragma unroll
    for (int r = 0; r < R; ++r) prod[r] = po[r] * (ls * L.ipi_r[r]);  // f81_parent_prod
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j < nc) {
            const int ch = fc + j;
            const int code = unit_code(u.packed, j);
            const double e = ce[j];
            unsigned cb[Q];
            lane_words_bits<G, R>(L, c, cw[j], cb);
            double mb[R], tdc[R], pc[R], lsc;
            bits_to_vec<R>(cb, mb);
            i64 xe, le;
            if (code == 0) {
                // f81_finish_tip's general path
                f81_finish_child<G, R>(L, c, prod, pe, ch, e, cs[j], 0, mb, false, mb, tdc, xe, pc, lsc, le, -1);
            } else if (code == 1) {
                double v[R];
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = vv[j][r];
                f81_finish_child<G, R>(L, c, prod, pe, ch, e, cs[j], cbe[j], v, false, mb, tdc, xe, pc, lsc, le, -1);
                if (st.td != nullptr) {
                    node_store_vec<G, R>(L, c, L.td, ch, tdc);
                    if (L.g == 0) L.te[ch] = xe;
                }
            } else {
                // cherry: its bottom-up vector again (f81_cherry_vector's operations), finished, then its tips
                double v[R];
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = mb[r];
                double amin = 1.0;
                i64 bec = 0;
                unsigned tb[2][Q];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    lane_words_bits<G, R>(L, c, tw[j][q], tb[q]);
                    if (q < code - 1) {
                        const double ta = (1.0 - te[j][q]) * ts[j][q];
                        double tmsg[R];
                        bits_select_vec<R>(tb[q], ta, te[j][q], tmsg);
                        amin *= ta;
#pragma unroll
                        for (int r = 0; r < R; ++r) v[r] *= tmsg[r];
                    }
                }
                if (!(amin >= 0x1p-190)) bec = lazy_rescale<G, R>(v);
                f81_finish_child<G, R>(L, c, prod, pe, ch, e, cs[j], bec, v, false, mb, tdc, xe, pc, lsc, le, -1);
                double prod2[R];
#pragma unroll
                for (int r = 0; r < R; ++r) prod2[r] = pc[r] * (lsc * L.ipi_r[r]);
                const int cfc = j == 0 ? cfc0 : cfc1;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (q < code - 1) {
                        double mt[R], tdt[R], pt[R], lt;
                        bits_to_vec<R>(tb[q], mt);
                        i64 xt, et;
                        f81_finish_child<G, R>(L, c, prod2, le, cfc + q, te[j][q], ts[j][q], 0, mt, false, mt, tdt, xt, pt, lt, et, -1);
                    }
                }
            }
        }
    }
}

// One unit = (stored internal node of the depth level, column): the parent's BU and TD vectors are loaded once and
// every child is finished from them; cherry children are recomputed and their tips finished in the same unit.
// replaces calc_node_td_likelihood (ml.py:273-290), calc_node_marginal_likelihood (:454-460) and the normalisation
// of convert_likelihoods_to_probabilities (:498-500) for the F81 family.
// One top-down unit: stored internal node p of the current depth level.
template <int G, int R, bool LEAN = false, bool SHAPES = false>
__device__ __forceinline__ void td_f81_unit(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                            const PmlState& st, const UnitRegs& u) {
    if (LEAN && G < 8 && c.W == 1 && unit_is_lean(u.packed)) {
        const int key = u.packed & (15 | (63 << 8));
        const int key0 = __builtin_amdgcn_readfirstlane(key);
        if (SHAPES && __ballot(key != key0) == 0ull) {  // (a wave of one shape: bu_f81_unit)
            switch (key0) {
                case PML_SHAPE_KEY(2, 1, 1): td_f81_unit_lean<G, R, 1, 1>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 3, 3): td_f81_unit_lean<G, R, 3, 3>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 1, 0): td_f81_unit_lean<G, R, 1, 0>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 0, 1): td_f81_unit_lean<G, R, 0, 1>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 1, 3): td_f81_unit_lean<G, R, 1, 3>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 3, 1): td_f81_unit_lean<G, R, 3, 1>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 0, 3): td_f81_unit_lean<G, R, 0, 3>(L, t, c, st, u); break;
                case PML_SHAPE_KEY(2, 3, 0): td_f81_unit_lean<G, R, 3, 0>(L, t, c, st, u); break;
                default: td_f81_unit_lean<G, R>(L, t, c, st, u); break;
            }
        } else {
            td_f81_unit_lean<G, R>(L, t, c, st, u);
        }
        return;
    }
    if (SHAPES && LEAN && G >= 2 && G < 8 && c.W == 1 && !c.no_wide_lean) {
        const bool poly = unit_is_lean_poly(u.packed);   // (a wave of such units only: bu_f81_unit)
        if (__ballot(!poly) == 0ull) {
            td_f81_unit_lean_poly<G, R>(L, t, c, st, u);
            return;
        }
    }
    if constexpr (G >= 32) {
        if (LEAN && c.W > 1 && !c.no_wide_lean && unit_is_lean(u.packed)) {  // (masks of several words: bu_f81_unit)
            const int cfc0 = __shfl(u.cfc, L.group_base, 64), cfc1 = __shfl(u.cfc, L.group_base + Gather<G>::GC, 64);
            td_f81_unit_lean_w<G, R>(L, t, c, st, u, cfc0, cfc1);
            return;
        }
    }
    if (Gather<G>::enabled && c.W == 1 && unit_is_fast<G>(u.packed)) {
        td_f81_unit_fast<G, R>(L, t, c, st, u);
        return;
    }
    const int p = u.n;

    double prod[R];
    i64 pe;
    f81_parent_prod<G, R>(L, c, p, prod, pe);
    double P = 0.0;
    bool have_P = false;
    const int fc = u.fc;
    int nc = unit_nc(u.packed);
    if (nc == 15) nc = t.n_children[p];  // the descriptor counts up to 14
    for (int j = 0; j < nc; ++j) {
        const int ch = fc + j;
        const ChildDesc cd = unit_child<G>(t, u, j, ch);
        const int kd = cd.kind;
        const int slot = j < 2 ? j : -1;  // staging slots: children 0 and 1, then two tips of each if it is a cherry
        if (kd == PML_KIND_TIP) {
            f81_finish_tip<G, R>(L, c, prod, pe, P, have_P, ch, slot);
            continue;
        }
        const double e = L.E[ch];
        double mb[R], v[R], tdc[R], po[R], ls;
        node_mask_vec<G, R>(L, c, ch, mb);
        i64 xe, le;
        if (kd == PML_KIND_STORED) {
            node_load_vec<G, R>(L, c, L.bu, ch, v);
            f81_finish_child<G, R>(L, c, prod, pe, ch, e, L.S[ch], L.be[ch], v, false, mb, tdc, xe, po, ls, le, slot);
            if (st.td != nullptr) {
                node_store_vec<G, R>(L, c, L.td, ch, tdc);
                if (L.g == 0) L.te[ch] = xe;
            }
        } else {
            // cherry: rebuild its bottom-up vector, finish it, then finish its tips from registers
            i64 bec;
            f81_cherry_vector<G, R>(L, t, c, st, ch, v, bec, false, cd.tips_known, cd.fc2, cd.nc2);
            const double s_child = L.S[ch];  // pi . v, stored by the bottom-up sweep
            f81_finish_child<G, R>(L, c, prod, pe, ch, e, s_child, bec, v, false, mb, tdc, xe, po, ls, le, slot);
            double prod2[R];
#pragma unroll
            for (int r = 0; r < R; ++r) prod2[r] = po[r] * (ls * L.ipi_r[r]);
            const i64 pe2 = le;
            double P2 = 0.0;
            bool have_P2 = false;
            const int fc2 = cd.tips_known ? cd.fc2 : t.first_child[ch];
            const int nc2 = cd.tips_known ? cd.nc2 : t.n_children[ch];
            for (int q = 0; q < nc2; ++q)
                f81_finish_tip<G, R>(L, c, prod2, pe2, P2, have_P2, fc2 + q, (j < 2 && q < 2) ? 2 + 2 * j + q : -1);
        }
    }
}

// Staging area of a wave (UW units): rows [UW][2][ks], then (optional) lhsum [UW][6], lhe [UW][6], then the ints: row
// ids [UW][2], tip ids [UW][4], tip states [UW][4].
struct TdStage {
    double* rows;
    double* sum;
    i64* exp;
    int* id;
    int* tid;
    int* tst;
};
__host__ __device__ __forceinline__ int td_stage_doubles(int uw, int ks, bool scalars) {
    return uw * 2 * ks + (scalars ? uw * 12 : 0) + uw * 5;  // 10 ints per unit
}

// The wave writes out what its units staged: piece p (16 bytes) of the wave's slots by lane p mod 64, so that
// consecutive lanes write consecutive memory wherever consecutive units have consecutive children / tips.
template <int G, int R>
__device__ __forceinline__ void td_stage_flush(const LaneCtx<G, R>& L, const PmlCols& c, const TdStage& S, bool tips) {
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    constexpr int UW = 64 / G;
    wave_sync_lds();
    const int lane = threadIdx.x & 63;
    const int ppr = c.ks >> 1;  // 16-byte pieces per row (ks is even for k >= 2)
    for (int p = lane; p < UW * 2 * ppr; p += 64) {
        const int slot = p / ppr, piece = p - slot * ppr;
        const int node = S.id[slot];
        if (node >= 0) {
            const dbl2 v = *reinterpret_cast<const dbl2*>(S.rows + slot * c.ks + 2 * piece);
            __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(L.post + (size_t)(unsigned)node * c.ks + 2 * piece));
        }
    }
    for (int p = lane; tips && p < UW * 4 * ppr; p += 64) {
        const int slot = p / ppr, piece = p - slot * ppr;
        const int node = S.tid[slot];
        if (node >= 0) {
            const int s = S.tst[slot];
            dbl2 v;
            v.x = s == 2 * piece ? 1.0 : 0.0;
            v.y = s == 2 * piece + 1 ? 1.0 : 0.0;
            __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(L.post + (size_t)(unsigned)node * c.ks + 2 * piece));
        }
    }
    if (S.sum != nullptr) {
        for (int e = lane; e < UW * 6; e += 64) {
            const int u = e / 6, j = e - u * 6;
            if (j >= 2 && !tips) continue;
            const int node = j < 2 ? S.id[u * 2 + j] : S.tid[u * 4 + j - 2];
            if (node >= 0) {
                L.lhsum[node] = S.sum[e];
                L.lhe[node] = S.exp[e];
            }
        }
    }
    wave_sync_lds();
    for (int e = lane; e < UW * 2; e += 64) S.id[e] = -1;
    for (int e = lane; tips && e < UW * 4; e += 64) S.tid[e] = -1;
    wave_sync_lds();
}

// stage: 0 = posteriors straight to memory; bit 0 = rows through the LDS slots, bit 1 = also the two scalars per row,
// bit 2 = the level has cherries among the first two children of its units (tip slots in use)
template <int G, int R>
__global__ void __launch_bounds__(PML_BLOCK)
td_f81_kernel(PmlTree t, PmlCols c, PmlState st, const PmlUnit* __restrict__ units, int n_level, int stage) {
    extern __shared__ double td_stage[];
    constexpr int UW = 64 / G;
    const int wave = threadIdx.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    TdStage S = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const bool staged = G <= PML_TD_STAGE_MAX_G && (stage & 1) && (c.ks & 1) == 0;
    if (staged) {
        const bool scalars = (stage & 2) != 0;
        double* base = td_stage + wave * td_stage_doubles(UW, c.ks, scalars);
        S.rows = base;
        base += UW * 2 * c.ks;
        if (scalars) {
            S.sum = base;
            S.exp = reinterpret_cast<i64*>(base + UW * 6);
            base += UW * 12;
        }
        S.id = reinterpret_cast<int*>(base);
        S.tid = S.id + UW * 2;
        S.tst = S.tid + UW * 4;
        L.srow = S.rows + sub * 2 * c.ks;
        L.sid = S.id + sub * 2;
        L.stid = S.tid + sub * 4;
        L.stst = S.tst + sub * 4;
        if (scalars) {
            L.ssum = S.sum + sub * 6;
            L.sexp = S.exp + sub * 6;
        }
        for (int e = threadIdx.x & 63; e < UW * 2; e += 64) S.id[e] = -1;
        for (int e = threadIdx.x & 63; e < UW * 4; e += 64) S.tid[e] = -1;
        wave_sync_lds();
    }
    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    int idx = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW + sub;
    UnitRegs cur = load_unit<G>(units, idx < n_level ? idx : 0, L.g);
    for (int base = idx - sub; base < n_level; base += stride) {
        const int nxt_idx = idx + stride;
        const UnitRegs nxt = load_unit<G>(units, nxt_idx < n_level ? nxt_idx : 0, L.g);
        if (idx < n_level) td_f81_unit<G, R, true>(L, t, c, st, cur);
        if (staged) td_stage_flush<G, R>(L, c, S, (stage & 4) != 0);
        cur = nxt;
        idx = nxt_idx;
    }
}

// Top-down side of the two-level units (see bu_f81_super_unit): one unit per CHILD j of a two-level node p.  The unit
// rebuilds the child's bottom-up vector from the tips (the bottom-up body again, with pi . v of the cherries as that
// sweep stored it: the vector was never written), finishes the child from p's posterior row -- the step of p's own unit
// that concerns this child -- and goes on with the child's own unit (two cherries of two tips) on the row, sum and
// exponent it has just produced: what the next level launch would have read back.  The two children of p are adjacent
// units of one wavefront, so p's row comes from memory once.  Same functions, same arguments, same lane shape as the
// level kernels: the same bits.
template <int G, int R>
__device__ __forceinline__ void td_f81_super_unit(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                  const PmlState& st, const SuperRegs& s, int j) {
    const u64 kbits = state_bits(c.k);
    const int ch = s.fc + j;
    const UnitRegs u = super_child<G>(s, j, L.g);
    double prod[R];
    i64 pe;
    f81_parent_prod<G, R>(L, c, s.n, prod, pe);
    const double e = L.E[ch];
    const double s_child = L.S[ch];
    const i64 bec = L.be[ch];
    BuLoads<R> ld;
    ld.own = L.mask[(unsigned)ch];
    f81_gather_issue<G, R>(L, u, ld.cl, ld.tl);
    f81_gather_finish<G, R>(L, ld.cl, ld.tl);
    ld.own &= kbits;
    const bool all_ones = __all(c.k == G * R && ld.own == kbits && ld.cl.mask == kbits);
    BuResult<R> v;
    CherryKeep<R> ck;  // the two cherries' vectors: the child's own unit below needs them again
    if (all_ones) bu_f81_marg_body<G, R, false, true, true>(L, t, c, st, u, ld, &v, false, true, &ck);
    else bu_f81_marg_body<G, R, false, false, true>(L, t, c, st, u, ld, &v, false, true, &ck);
    double prod2[R];
    i64 pe2;
    {
        double mb[R], tdc[R], po[R], ls;
        const bool full = ld.own == kbits;
        if (!full) clean_word_to_vec<G, R>(L, c, ld.own, mb);
        i64 xe;
        f81_finish_child<G, R>(L, c, prod, pe, ch, e, s_child, bec, v.v, full, mb, tdc, xe, po, ls, pe2);
        if (st.td != nullptr) {
            node_store_vec<G, R>(L, c, L.td, ch, tdc);
            if (L.g == 0) L.te[ch] = xe;
        }
        // f81_parent_prod on the row, sum and exponent just stored
#pragma unroll
        for (int r = 0; r < R; ++r) prod2[r] = po[r] * (ls * L.ipi_r[r]);
    }
    // the child's own unit: td_f81_fast_children for two cherries of two tips, with the cherries' vectors at hand
    constexpr int GC = Gather<G>::GC;
#pragma unroll
    for (int jx = 0; jx < 2; ++jx) {
        const int src = L.group_base + jx;
        const int cch = u.fc + jx;
        const double ce = __shfl(ld.cl.e, src, 64);
        const u64 word = __shfl(ld.cl.mask, src, 64);
        double mb[R], tdc[R], po[R], ls;
        const bool full = word == kbits;
        if (!full) clean_word_to_vec<G, R>(L, c, word, mb);
        i64 xe, le;
        const int cfc = __shfl(u.cfc, L.group_base + jx * GC, 64);
        f81_finish_child<G, R>(L, c, prod2, pe2, cch, ce, __shfl(ld.cl.s, src, 64), ck.e[jx], ck.v[jx], full, mb, tdc, xe, po,
                               ls, le);
        double prod3[R];
#pragma unroll
        for (int r = 0; r < R; ++r) prod3[r] = po[r] * (ls * L.ipi_r[r]);
        double P3 = 0.0;
        bool have_P3 = false;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ts = L.group_base + jx * GC + q;
            f81_finish_tip_word<G, R>(L, c, prod3, le, P3, have_P3, cfc + q, __shfl(ld.tl.mask, ts, 64),
                                      __shfl(ld.tl.e, ts, 64), __shfl(ld.tl.s, ts, 64));
        }
    }
}

// n_level = number of two-level nodes; unit i is child i & 1 of node i >> 1
template <int G, int R>
__global__ void __launch_bounds__(PML_BLOCK)
td_f81_super_kernel(PmlTree t, PmlCols c, PmlState st, const PmlUnit* __restrict__ units, int n_level) {
    constexpr int UW = 64 / G;
    const int wave = threadIdx.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    const int n_units = 2 * n_level;
    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    int idx = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW + sub;
    SuperRegs cur = load_super(units, idx < n_units ? idx >> 1 : 0);
    for (int base = idx - sub; base < n_units; base += stride) {
        const int nxt_idx = idx + stride;
        const SuperRegs nxt = load_super(units, nxt_idx < n_units ? nxt_idx >> 1 : 0);
        if (idx < n_units) td_f81_super_unit<G, R>(L, t, c, st, cur, idx & 1);
        cur = nxt;
        idx = nxt_idx;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stacked units (round 3): the two-level idea above the tips.  A stored node n with two plain stored children that each
// have two stored children (whose vectors are in memory) runs, bottom-up, its children's units and its own in one unit:
// four vectors in, one out -- the children's vectors are not written (pi . v and the exponent are), nor read back.
// Top-down there is one unit per child c: it rebuilds c's vector from its two children's vectors (which it needs anyway),
// finishes c from n's row, then c's two children from the row it has just formed.  The level kernels' bodies with their
// arguments, in their lane shape: the same bits.  Descriptor: the standard one of n (cfc[j] = first child of child j).
// ---------------------------------------------------------------------------------------------------------------------
struct StackRegs {
    int n, fc, g0, g1;
};

__device__ __forceinline__ StackRegs load_stack(const PmlUnit* __restrict__ units, int idx) {
    const int4 h = *reinterpret_cast<const int4*>(units + idx);
    const int4 f = *reinterpret_cast<const int4*>(units[idx].cfc);
    StackRegs s;
    s.n = h.x;
    s.fc = h.y;
    s.g0 = f.x;
    s.g1 = f.y;
    return s;
}

__device__ __forceinline__ UnitRegs stack_child(const StackRegs& s, int j) {
    UnitRegs u;
    u.n = s.fc + j;
    u.fc = j ? s.g1 : s.g0;
    u.packed = PML_PACKED_TWO_STORED;
    u.cfc = u.cfc1 = u.cfc2 = u.cfc3 = 0;
    return u;
}

// what a unit with two stored children reads about them: lane j < 2 holds child j's scalars, the vectors in v0 / v1
template <int G, int R>
__device__ __forceinline__ void stack_child_loads(const LaneCtx<G, R>& L, const PmlCols& c, int node, int first,
                                                  BuLoads<R>& ld) {
    const int ch = first + (L.g & 1);
    ld.cl.e = L.E[ch];
    ld.cl.s = L.S[ch];
    ld.cl.be = L.be[ch];
    ld.cl.mask = 0ull;
    ld.tl.e = ld.tl.s = ld.tl.a = 0.0;
    ld.tl.mask = 0ull;
    ld.own = L.mask[(unsigned)node];
    node_load_vec<G, R>(L, c, L.bu, first, ld.v0);
    node_load_vec<G, R>(L, c, L.bu, first + 1, ld.v1);
}

template <int G, int R>
__device__ __forceinline__ bool bu_f81_stack_unit(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                  const PmlState& st, const StackRegs& s) {
    BuLoads<R> ld[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) stack_child_loads<G, R>(L, c, s.fc + j, j ? s.g1 : s.g0, ld[j]);
    const double e_top = L.E[s.fc + (L.g & 1)];
    const u64 own = L.mask[(unsigned)s.n];
    BuResult<R> res[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (!bu_f81_marg_body<G, R, true, false>(L, t, c, st, stack_child(s, j), ld[j], &res[j], false)) return false;
    BuLoads<R> top;
    top.cl.e = e_top;
    top.cl.s = (L.g & 1) ? res[1].s : res[0].s;
    top.cl.be = (L.g & 1) ? res[1].e : res[0].e;
    top.cl.mask = 0ull;
    top.tl.e = top.tl.s = top.tl.a = 0.0;
    top.tl.mask = 0ull;
    top.own = own;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        top.v0[r] = res[0].v[r];
        top.v1[r] = res[1].v[r];
    }
    UnitRegs un;
    un.n = s.n;
    un.fc = s.fc;
    un.packed = PML_PACKED_TWO_STORED;
    un.cfc = un.cfc1 = un.cfc2 = un.cfc3 = 0;
    return bu_f81_marg_body<G, R, true, false>(L, t, c, st, un, top);
}

template <int G, int R>
__global__ void __launch_bounds__(PML_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
bu_f81_stack_kernel(PmlTree t, PmlCols c, PmlState st, const PmlUnit* __restrict__ units, int n_level) {
    constexpr int UW = 64 / G;
    const int wave = threadIdx.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    int idx = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW + sub;
    StackRegs cur = load_stack(units, idx < n_level ? idx : 0);
    for (int base = idx - sub; base < n_level; base += stride) {
        const int nxt_idx = idx + stride;
        const StackRegs nxt = load_stack(units, nxt_idx < n_level ? nxt_idx : 0);
        if (idx < n_level) {
            if (!bu_f81_stack_unit<G, R>(L, t, c, st, cur)) {
                // an all-zero vector: the three units on the sequential path, which names the pair the reference would
                bu_f81_unit_seq<G, R, false>(L, t, c, st, stack_child(cur, 0));
                bu_f81_unit_seq<G, R, false>(L, t, c, st, stack_child(cur, 1));
                __threadfence();
                UnitRegs un = stack_child(cur, 0);
                un.n = cur.n;
                un.fc = cur.fc;
                bu_f81_unit_seq<G, R, false>(L, t, c, st, un);
            }
        }
        cur = nxt;
        idx = nxt_idx;
    }
}

// top-down: unit i is child i & 1 of stacked node i >> 1
template <int G, int R>
__device__ __forceinline__ void td_f81_stack_unit(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                                  const PmlState& st, const StackRegs& s, int j) {
    const u64 kbits = state_bits(c.k);
    const int ch = s.fc + j;
    const int first = j ? s.g1 : s.g0;
    double prod[R];
    i64 pe;
    f81_parent_prod<G, R>(L, c, s.n, prod, pe);
    const double e = L.E[ch];
    const double s_child = L.S[ch];
    const i64 bec = L.be[ch];
    BuLoads<R> ld;
    stack_child_loads<G, R>(L, c, ch, first, ld);
    // the grandchildren's masks (for their own rows): lane j < 2
    const u64 gmask = L.mask[(unsigned)(first + (L.g & 1))] & kbits;
    ld.own &= kbits;
    BuResult<R> v;
    double g0v[R], g1v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        g0v[r] = ld.v0[r];
        g1v[r] = ld.v1[r];
    }
    bu_f81_marg_body<G, R, true, false>(L, t, c, st, stack_child(s, j), ld, &v, false, true);
    double prod2[R];
    i64 pe2;
    {
        double mb[R], tdc[R], po[R], ls;
        const bool full = ld.own == kbits;
        if (!full) clean_word_to_vec<G, R>(L, c, ld.own, mb);
        i64 xe;
        f81_finish_child<G, R>(L, c, prod, pe, ch, e, s_child, bec, v.v, full, mb, tdc, xe, po, ls, pe2);
        if (st.td != nullptr) {
            node_store_vec<G, R>(L, c, L.td, ch, tdc);
            if (L.g == 0) L.te[ch] = xe;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) prod2[r] = po[r] * (ls * L.ipi_r[r]);  // f81_parent_prod on the row just stored
    }
#pragma unroll
    for (int jx = 0; jx < 2; ++jx) {
        const int src = L.group_base + jx;
        const u64 word = __shfl(gmask, src, 64);
        double mb[R], tdc[R], po[R], ls;
        const bool full = word == kbits;
        if (!full) clean_word_to_vec<G, R>(L, c, word, mb);
        i64 xe, le;
        f81_finish_child<G, R>(L, c, prod2, pe2, first + jx, __shfl(ld.cl.e, src, 64), __shfl(ld.cl.s, src, 64),
                               __shfl(ld.cl.be, src, 64), jx ? g1v : g0v, full, mb, tdc, xe, po, ls, le);
        if (st.td != nullptr) {
            node_store_vec<G, R>(L, c, L.td, first + jx, tdc);
            if (L.g == 0) L.te[first + jx] = xe;
        }
    }
}

template <int G, int R>
__global__ void __launch_bounds__(PML_BLOCK)
td_f81_stack_kernel(PmlTree t, PmlCols c, PmlState st, const PmlUnit* __restrict__ units, int n_level) {
    constexpr int UW = 64 / G;
    const int wave = threadIdx.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    const int n_units = 2 * n_level;
    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    int idx = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW + sub;
    StackRegs cur = load_stack(units, idx < n_units ? idx >> 1 : 0);
    for (int base = idx - sub; base < n_units; base += stride) {
        const int nxt_idx = idx + stride;
        const StackRegs nxt = load_stack(units, nxt_idx < n_units ? nxt_idx >> 1 : 0);
        if (idx < n_units) td_f81_stack_unit<G, R>(L, t, c, st, cur, idx & 1);
        cur = nxt;
        idx = nxt_idx;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small forests: the whole sweep in ONE launch.  One workgroup per column walks the levels with a workgroup barrier
// between them, so a sweep costs one kernel launch instead of one per level (a 150-tip tree has ~16 levels of a few
// nodes each: the level-per-launch schedule is pure launch latency there, and the optimiser repeats the bottom-up
// sweep hundreds of times, pastml/ml.py:174-237).  Same unit functions, hence the same bits as the level kernels.
// ---------------------------------------------------------------------------------------------------------------------
#define PML_SMALL_BLOCK 512

// ln L of one column (ml.py:112-121), shared by loglik_kernel and the single-launch kernels
__device__ __forceinline__ double column_loglik(const PmlTree& t, const PmlCols& c, const PmlState& st, int col,
                                                int is_marginal) {
    const size_t colN = (size_t)col * t.N;
    double total = 0.0;
    for (int r = 0; r < t.n_roots; ++r) {
        const bool tip = t.n_children[r] == 0;
        const u64* m = c.masks + (colN + r) * c.W;
        double term;
        if (is_marginal) {
            term = 0.0;
            for (int s = 0; s < c.k; ++s) {
                const double b = tip ? (double)((m[s >> 6] >> (s & 63)) & 1ull) : st.bu[(colN + r) * c.ks + s];
                term += b * c.pi[(size_t)col * c.ks + s];
            }
        } else {
            term = -INFINITY;
            int arg = 0;
            for (int s = 0; s < c.k; ++s) {
                const double b = tip ? (double)((m[s >> 6] >> (s & 63)) & 1ull) : st.bu[(colN + r) * c.ks + s];
                const double v = b * c.pi[(size_t)col * c.ks + s];
                if (v > term) {
                    term = v;
                    arg = s;
                }
            }
            st.js[colN + r] = arg;
        }
        const double e2 = tip ? 0.0 : (double)st.be[colN + r];
        total += log(term) + e2 * 0.693147180559945309417232121458;
    }
    return total;
}

// Walks the levels [0, nl) of a level table inside one workgroup (barrier between levels).  A level's units depend on
// the previous level's results, its DESCRIPTORS do not: the descriptor a lane needs first in the next level (and that
// level's bounds) are fetched before the current level is computed, which takes one round trip out of the dependent
// chain descriptor -> gathered scalars -> vectors of every level.  The units run with SHAPES: a wave whose units all have
// one of the common shapes takes that shape's variant of the lean unit (a level here is one wavefront's instruction
// stream: profiles/r04h_thin_level_step_breakdown.txt).
template <int G, int R, bool BU>
__device__ __forceinline__ void walk_levels(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c, const PmlState& st,
                                            const PmlUnit* __restrict__ units, const int* __restrict__ lv, int nl) {
    constexpr int UW = 64 / G;
    const int wave = threadIdx.x >> 6;
    const int n_waves = blockDim.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    const int first = wave * UW + sub;  // the unit of a level this lane's group takes first
    if (nl <= 0) return;
    int a = lv[0], b = lv[1];
    UnitRegs nxt = load_unit<G>(units, first < b - a ? a + first : a, L.g);
    for (int l = 0; l < nl; ++l) {
        const int n_level = b - a;
        const UnitRegs cur = nxt;
        int a2 = a, b2 = a;
        if (l + 1 < nl) {
            a2 = lv[l + 1];
            b2 = lv[l + 2];
            nxt = load_unit<G>(units, first < b2 - a2 ? a2 + first : a2, L.g);
        }
        if (first < n_level) {
            if (BU) bu_f81_unit<G, R, false, true, true>(L, t, c, st, cur);
            else td_f81_unit<G, R, true, true>(L, t, c, st, cur);
        }
        for (int base = wave * UW + n_waves * UW; base < n_level; base += n_waves * UW) {
            const int idx = base + sub;
            if (idx < n_level) {
                if (BU) bu_f81_unit<G, R, false, true, true>(L, t, c, st, load_unit<G>(units, a + idx, L.g));
                else td_f81_unit<G, R, true, true>(L, t, c, st, load_unit<G>(units, a + idx, L.g));
            }
        }
        __syncthreads();
        a = a2;
        b = b2;
    }
}


template <int G, int R>
__global__ void __launch_bounds__(PML_SMALL_BLOCK)
bu_f81_small_kernel(PmlTree t, PmlCols c, PmlState st, const double* __restrict__ mu, const double* __restrict__ sf,
                    const double* __restrict__ tau, const double* __restrict__ tauf, int do_prep,
                    const PmlUnit* __restrict__ units, const int* __restrict__ level_offsets, int n_levels,
                    double* __restrict__ loglik, u64* __restrict__ err_out, int reset_err,
                    u64* __restrict__ done_state, u64* done_flag) {
    const int col = blockIdx.y;
    if (!column_active(c, col)) {  // (its words in pinned memory stay what they were; the launch still counts it)
        if (threadIdx.x == 0 && done_flag != nullptr) {
            if (atomicAdd(&done_state[0], 1ull) == (u64)gridDim.y - 1ull) {
                done_state[0] = 0ull;
                const u64 generation = done_state[1] + 1ull;
                done_state[1] = generation;
                __threadfence_system();
                *reinterpret_cast<volatile u64*>(done_flag) = generation;
            }
        }
        return;
    }
    if (reset_err) {  // whole sweep in this launch: the column's error word is reset here, not by a launch of its own
        if (threadIdx.x == 0) st.err[col] = ~0ull;
        __syncthreads();
    }
    if (do_prep) {
        const size_t colN = (size_t)col * t.N;
        const double m = mu[col], s = sf[col], ta = tau[col], tf = tauf[col];
        for (int n = threadIdx.x; n < t.N; n += blockDim.x) {
            const double tt = (t.dist[n] + ta) * tf * s;
            st.E[colN + n] = isinf(m) ? 0.0 : exp(-m * tt);
            if (t.n_children[n] == 0) {
                double acc = 0.0;
                for (int w = 0; w < c.W; ++w) {
                    u64 word = c.masks[(colN + n) * c.W + w];
                    while (word) {
                        const int b = __builtin_ctzll(word);
                        acc += c.pi[(size_t)col * c.ks + w * 64 + b];
                        word &= word - 1ull;
                    }
                }
                st.S[colN + n] = acc;
            }
        }
        __syncthreads();
    }
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    walk_levels<G, R, true>(L, t, c, st, units, level_offsets, n_levels);
    if (threadIdx.x == 0) {
        // pinned host memory: the results land where the caller reads them
        loglik[col] = column_loglik(t, c, st, col, 1);
        err_out[col] = atomicMin(&st.err[col], ~0ull);  // the value in L2, whatever this CU's L1 holds
        // The host need not wait for the stream to drain: the workgroup that finishes last says so in pinned memory
        // (done_state[0] counts the columns of this launch, done_state[1] the launches; pml_bottom_up_collect spins on
        // *done_flag, which saves the ~4 us a stream synchronisation takes to notice -- scripts/ub/syncwait.hip).
        if (done_flag != nullptr) {
            __threadfence_system();
            if (atomicAdd(&done_state[0], 1ull) == (u64)gridDim.y - 1ull) {
                done_state[0] = 0ull;
                const u64 generation = done_state[1] + 1ull;
                done_state[1] = generation;
                __threadfence_system();
                *reinterpret_cast<volatile u64*>(done_flag) = generation;
            }
        }
    }
}

// root of a tree: TD = 1 with exponent 0 (ml.py:274-277), marginal likelihoods BU * pi * mask
template <int G, int R>
__device__ __forceinline__ void f81_root_unit(const LaneCtx<G, R>& L, const PmlTree& t, const PmlCols& c,
                                              const PmlState& st, int n) {
    const bool tip = t.n_children[n] == 0;
    double mb[R], v[R], one[R], lh[R];
    node_mask_vec<G, R>(L, c, n, mb);
    if (tip) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = mb[r];
    } else {
        node_load_vec<G, R>(L, c, L.bu, n, v);
    }
    double lhs = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        one[r] = (L.st(r) < c.k) ? 1.0 : 0.0;
        lh[r] = v[r] * (L.pi_r[r] * mb[r]);
        lhs += lh[r];
    }
    lhs = group_sum<G>(lhs);
    const int lex = (lhs > 0.0 && !isinf(lhs)) ? exponent_of(lhs) : 0;
#pragma unroll
    for (int r = 0; r < R; ++r) lh[r] = lh[r] / lhs;
    if (st.td != nullptr) node_store_vec<G, R>(L, c, L.td, n, one);
    node_store_vec_nt<G, R>(L, c, L.post, n, lh);
    if (L.g == 0) {
        if (st.td != nullptr) L.te[n] = 0;
        L.lhsum[n] = __builtin_ldexp(lhs, -lex);
        L.lhe[n] = (tip ? 0 : L.be[n]) + lex;
    }
}

// roots of the level-per-launch schedule (same unit function as the single-launch kernel: same bits)
template <int G, int R>
__global__ void __launch_bounds__(PML_BLOCK)
td_f81_roots_kernel(PmlTree t, PmlCols c, PmlState st) {
    constexpr int UW = 64 / G;
    const int wave = threadIdx.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    for (int base = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW; base < t.n_roots; base += stride) {
        const int idx = base + sub;
        if (idx < t.n_roots) f81_root_unit<G, R>(L, t, c, st, idx);
    }
}

// The last launch of a short marginal pass tells the host that it is done (see bu_f81_small_kernel; what the pass leaves in
// device memory is read by later work on the stream, what it left in pinned memory was written by earlier launches):
// called by thread 0 of every workgroup behind the barrier that ends its walk.
__device__ __forceinline__ void pml_signal_done(u64* __restrict__ done_state, u64* done_flag) {
    if (done_flag == nullptr) return;
    __threadfence();
    if (atomicAdd(&done_state[0], 1ull) == (u64)gridDim.x * gridDim.y - 1ull) {
        done_state[0] = 0ull;
        const u64 generation = done_state[1] + 1ull;
        done_state[1] = generation;
        __threadfence_system();
        *reinterpret_cast<volatile u64*>(done_flag) = generation;
    }
}

template <int G, int R>
__global__ void __launch_bounds__(PML_SMALL_BLOCK)
td_f81_small_kernel(PmlTree t, PmlCols c, PmlState st, const PmlUnit* __restrict__ units,
                    const int* __restrict__ level_offsets, int n_levels, u64* __restrict__ done_state, u64* done_flag,
                    int skip_roots) {
    constexpr int UW = 64 / G;
    const int wave = threadIdx.x >> 6;
    const int n_waves = blockDim.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    // (skip_roots: the launch walks the thin depths at the deep end of a large forest -- the roots were done long before)
    for (int base = wave * UW; base < t.n_roots && !skip_roots; base += n_waves * UW) {
        const int idx = base + sub;
        if (idx < t.n_roots) f81_root_unit<G, R>(L, t, c, st, idx);
    }
    __syncthreads();
    walk_levels<G, R, false>(L, t, c, st, units, level_offsets, n_levels);
    if (threadIdx.x == 0) pml_signal_done(done_state, done_flag);
}


// ---------------------------------------------------------------------------------------------------------------------
// Subtree blocks: mid-size forests in a handful of launches.  The stored nodes are cut into blocks -- maximal subtrees
// of at most PASTML_HIP_BLOCK_NODES (256) stored nodes (pml_tree_upload) -- and the nodes above the cuts (the "top").  A block
// depends on nothing outside itself in the bottom-up sweep, and only on its root's parent (a top node) in the top-down
// sweep, so ONE launch walks all blocks, one workgroup per (block, column) stepping through the block's levels with a
// workgroup barrier (~2 us per level) instead of one launch per level of the forest (~4.6 us each, and the levels of a
// 65 536-tip tree are 16): the bottom-up sweep is blocks + top, the top-down sweep top + blocks.  Same unit functions
// and lane shapes as the level kernels, hence the same bits.
// units: the blocks' units, block by block, each block level by level; level j of block b is
// units[lv[start[b] + j] .. lv[start[b] + j + 1]).
// ---------------------------------------------------------------------------------------------------------------------
template <int G, int R>
__global__ void __launch_bounds__(PML_SMALL_BLOCK)
bu_f81_blocks_kernel(PmlTree t, PmlCols c, PmlState st, const PmlUnit* __restrict__ units,
                     const int* __restrict__ blk_start, const int* __restrict__ blk_levels, const int* __restrict__ lv) {
    if (!column_active(c, blockIdx.y)) return;
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    walk_levels<G, R, true>(L, t, c, st, units, lv + blk_start[blockIdx.x], blk_levels[blockIdx.x]);
}

template <int G, int R>
__global__ void __launch_bounds__(PML_SMALL_BLOCK)
td_f81_blocks_kernel(PmlTree t, PmlCols c, PmlState st, const PmlUnit* __restrict__ units,
                     const int* __restrict__ blk_start, const int* __restrict__ blk_levels, const int* __restrict__ lv,
                     u64* __restrict__ done_state, u64* done_flag) {
    LaneCtx<G, R> L;
    lane_ctx_init<G, R>(L, t, c, st);
    walk_levels<G, R, false>(L, t, c, st, units, lv + blk_start[blockIdx.x], blk_levels[blockIdx.x]);
    if (threadIdx.x == 0) pml_signal_done(done_state, done_flag);
}
