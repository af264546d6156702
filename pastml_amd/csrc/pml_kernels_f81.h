// F81-family sweeps (F81 / JC / EFT): P(t) = (1 - e) 1 pi^T + e I with e = exp(-mu t') is never materialised;
// P v = (1 - e)(pi . v) 1 + e v costs O(k) per branch (pastml/models/F81Model.py:28-46 in closed form).
//
// Cherry fusion (marginal sweeps): an internal node whose children are all tips ("cherry", kind 1) is never written
// to HBM.  Its bottom-up vector is a product of closed-form tip messages (masks + two scalars per tip), so whoever
// needs it -- its parent in the bottom-up sweep, its parent again in the top-down sweep -- recomputes it in
// registers; its own top-down vector only lives in registers while its tips are finished.  On a balanced tree half of
// the internal nodes are cherries: the bottom-up traffic halves and the top-down traffic drops by ~40 %.
#pragma once
#include "pml_device.h"

#define PML_KIND_TIP 0
#define PML_KIND_CHERRY 1
#define PML_KIND_STORED 2

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
    const u64* masks;       // [C][N][W]
    const u64* masks_init;  // [C][N][W] or nullptr
    const double* pi;       // [C][ks]
};

struct PmlState {
    double* E;      // [C][N]      F81: exp(-mu t') per branch
    double* bu;     // [C][N][ks]  bottom-up vectors (stored internal nodes only; tips are their masks)
    double* S;      // [C][N]      F81 marginal: pi . bu (tips and stored internal nodes)
    i64* be;        // [C][N]      base-2 exponent of bu, accumulated over the subtree
    double* td;     // [C][N][ks]
    i64* te;        // [C][N]
    double* post;   // [C][N][ks]
    double* lhsum;  // [C][N]
    i64* lhe;       // [C][N]
    int* J;         // [C][N][ks]  joint argmax tables
    int* js;        // [C][N]      joint states
    u64* err;       // [C]         min over failing (post_rank << 32 | child id)
};

__device__ __forceinline__ int node_kind(const PmlTree& t, int n) {
    if (t.kind != nullptr) return t.kind[n];
    return t.n_children[n] == 0 ? PML_KIND_TIP : PML_KIND_STORED;
}

// ---------------------------------------------------------------------------------------------------------------------
// per-branch e = exp(-mu t') for every (node, column); for tips also S = pi . mask and be = 0
// replaces: transform_t (models/__init__.py:269) + the exp of F81Model.get_Pij_t (F81Model.py:42-45)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PML_BLOCK)
f81_prep_kernel(PmlTree t, PmlCols c, const double* __restrict__ mu, const double* __restrict__ sf,
                const double* __restrict__ tau, const double* __restrict__ tauf, PmlState st) {
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const double m = mu[col], s = sf[col], ta = tau[col], tf = tauf[col];
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < t.N; n += gridDim.x * blockDim.x) {
        const double tt = (t.dist[n] + ta) * tf * s;
        // if mu == inf (a single state) it wins over t == 0 (F81Model.py:44-45)
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
            st.be[colN + n] = 0;
        }
    }
}

// Per-lane context of a unit's lane group.
template <int R>
struct LaneCtx {
    int col, g, s0, w0;
    bool lane_valid;
    size_t colN;
    double pi_r[R];
};

template <int G, int R>
__device__ __forceinline__ void lane_ctx_init(LaneCtx<R>& L, const PmlTree& t, const PmlCols& c) {
    const int lane = threadIdx.x & 63;
    L.col = blockIdx.y;
    L.g = lane & (G - 1);
    L.s0 = L.g * R;
    L.w0 = L.s0 >> 6;
    L.lane_valid = L.s0 < c.ks;
    L.colN = (size_t)L.col * t.N;
#pragma unroll
    for (int r = 0; r < R; ++r) L.pi_r[r] = (L.s0 + r < c.k) ? c.pi[(size_t)L.col * c.ks + L.s0 + r] : 0.0;
}

template <int R>
__device__ __forceinline__ void node_mask_vec(const LaneCtx<R>& L, const PmlCols& c, int n, double (&v)[R]) {
    const u64 word = L.lane_valid ? c.masks[(L.colN + n) * c.W + L.w0] : 0ull;
    mask_to_vec<R>(word, L.s0, c.k, v);
}

template <int R>
__device__ __forceinline__ void node_load_vec(const LaneCtx<R>& L, const PmlCols& c, const double* base, int n,
                                              double (&v)[R]) {
    if (L.lane_valid) {
        load_vec<R>(base + (L.colN + n) * c.ks + L.s0, v);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = 0.0;
    }
}

// Multiplies acc by the message of child ch (vector v, S = pi . v, branch factor e), then the zero check of
// ml.py:139-145 and the lazy rescale.  Returns the exponent taken out.
template <int G, int R>
__device__ __forceinline__ int f81_absorb_child(const LaneCtx<R>& L, const PmlTree& t, const PmlState& st, int n,
                                                int ch, double e, double s_child, const double (&v)[R],
                                                double (&acc)[R], bool report) {
    const double a = (1.0 - e) * s_child;
    bool nz = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r] *= a + e * v[r];  // every term of the closed form is >= 0: the clamp of ml.py:137 is a no-op
        nz |= acc[r] != 0.0;
    }
    if (report && !group_any<G>(nz)) {
        if (L.g == 0) atomicMin(&st.err[L.col], ((u64)(unsigned)t.post_rank[n] << 32) | (u64)(unsigned)ch);
    }
    return lazy_rescale<G, R>(acc);
}

// Bottom-up vector of a cherry (all children are tips) in registers: mask * prod of tip messages (ml.py:124-148).
template <int G, int R>
__device__ __forceinline__ void f81_cherry_vector(const LaneCtx<R>& L, const PmlTree& t, const PmlCols& c,
                                                  const PmlState& st, int n, double (&acc)[R], i64& esum,
                                                  bool report) {
    node_mask_vec<R>(L, c, n, acc);
    esum = 0;
    const int fc = t.first_child[n];
    const int nc = t.n_children[n];
    for (int j = 0; j < nc; ++j) {
        const int ch = fc + j;
        double v[R];
        node_mask_vec<R>(L, c, ch, v);
        esum += f81_absorb_child<G, R>(L, t, st, n, ch, st.E[L.colN + ch], st.S[L.colN + ch], v, acc, report);
    }
}

template <int G, int R>
__device__ __forceinline__ double pi_dot(const LaneCtx<R>& L, const double (&v)[R]) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += L.pi_r[r] * v[r];
    return group_sum<G>(s);
}

// ---------------------------------------------------------------------------------------------------------------------
// bottom-up level kernel. One unit = (stored internal node of the level, column).
// replaces calc_node_bu_likelihood (pastml/ml.py:124-148) for the F81 family.
// JOINT: Pupko's max / arg-max variant (never fused: t.kind is null there).
// ---------------------------------------------------------------------------------------------------------------------
template <int G, int R, bool JOINT>
__global__ void __launch_bounds__(PML_BLOCK)
bu_f81_kernel(PmlTree t, PmlCols c, PmlState st, const int* __restrict__ level_nodes, int n_level) {
    constexpr int UW = 64 / G;  // units per wave
    const int wave = threadIdx.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    LaneCtx<R> L;
    lane_ctx_init<G, R>(L, t, c);

    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    for (int base = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW; base < n_level; base += stride) {
        const int idx = base + sub;
        if (idx >= n_level) continue;  // whole groups drop out together
        const int n = level_nodes[idx];

        double acc[R];
        node_mask_vec<R>(L, c, n, acc);
        i64 esum = 0;
        const int fc = t.first_child[n];
        const int nc = t.n_children[n];
        for (int j = 0; j < nc; ++j) {
            const int ch = fc + j;
            const double e = st.E[L.colN + ch];
            const int kd = node_kind(t, ch);
            double v[R];
            double s_child = 0.0;
            if (kd == PML_KIND_TIP) {
                node_mask_vec<R>(L, c, ch, v);
                if (!JOINT) s_child = st.S[L.colN + ch];
            } else if (kd == PML_KIND_STORED) {
                node_load_vec<R>(L, c, st.bu, ch, v);
                esum += st.be[L.colN + ch];
                if (!JOINT) s_child = st.S[L.colN + ch];
            } else {
                i64 ce;
                f81_cherry_vector<G, R>(L, t, c, st, ch, v, ce, true);
                esum += ce;
                s_child = pi_dot<G, R>(L, v);
            }
            if (!JOINT) {
                esum += f81_absorb_child<G, R>(L, t, st, n, ch, e, s_child, v, acc, true);
            } else {
                // row i of P * diag(v): off-diagonal entries w_j = ((1-e) pi_j) v_j, diagonal ((1-e) pi_i + e) v_i
                // (same rounding sequence as the reference's P * v broadcast, ml.py:130 with F81Model.py:46)
                const double ome = 1.0 - e;
                double w[R], dg[R];
                double m1 = -INFINITY;
                int j1 = 0x7fffffff;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double a = ome * L.pi_r[r];
                    const bool ok = L.s0 + r < c.k;
                    w[r] = ok ? a * v[r] : -INFINITY;
                    dg[r] = (a + e) * v[r];
                    if (ok && w[r] > m1) {
                        m1 = w[r];
                        j1 = L.s0 + r;
                    }
                }
                group_argmax_first<G>(m1, j1);
                double m2 = -INFINITY;
                int j2 = 0x7fffffff;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (L.s0 + r < c.k && L.s0 + r != j1 && w[r] > m2) {
                        m2 = w[r];
                        j2 = L.s0 + r;
                    }
                }
                group_argmax_first<G>(m2, j2);
                int jj[R];
                bool nz = false;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = L.s0 + r;
                    const double mo = (i == j1) ? m2 : m1;
                    const int jo = (i == j1) ? j2 : j1;
                    double msg;
                    int arg;
                    if (jo >= c.k || dg[r] > mo) {  // no off-diagonal candidate (k == 1) or the diagonal wins
                        msg = dg[r];
                        arg = i;
                    } else if (dg[r] < mo) {
                        msg = mo;
                        arg = jo;
                    } else {  // tie: numpy's argmax returns the first index
                        msg = mo;
                        arg = min(i, jo);
                    }
                    acc[r] *= fmax(msg, 0.0);
                    nz |= acc[r] != 0.0;
                    jj[r] = (i < c.k) ? arg : 0;
                }
                // altered nodes get their tables rewritten w.r.t. their initial masks (ml.py:408-428)
                if (c.masks_init != nullptr) {
                    const u64* mi = c.masks_init + (L.colN + ch) * c.W;
                    const u64* mc = c.masks + (L.colN + ch) * c.W;
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
                if (L.lane_valid) store_vec_i32<R>(st.J + (L.colN + ch) * c.ks + L.s0, jj);
                if (!group_any<G>(nz)) {
                    if (L.g == 0)
                        atomicMin(&st.err[L.col], ((u64)(unsigned)t.post_rank[n] << 32) | (u64)(unsigned)ch);
                }
                esum += lazy_rescale<G, R>(acc);
            }
        }
        if (!JOINT) {
            const double s = pi_dot<G, R>(L, acc);
            if (L.g == 0) st.S[L.colN + n] = s;
        }
        if (L.lane_valid) store_vec<R>(st.bu + (L.colN + n) * c.ks + L.s0, acc);
        if (L.g == 0) st.be[L.colN + n] = esum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// top-down + marginal likelihoods + posteriors.
//
// f81_finish_child: given prod = TD_parent o BU_parent (exponent pe) and the child's own data, divides the child's
// message out of the parent (ml.py:279-283), pushes the result through the child's branch (ml.py:287-289), forms the
// marginal likelihoods pi o mask o BU o TD (ml.py:456-460) and stores the posteriors (ml.py:498-500).
// ---------------------------------------------------------------------------------------------------------------------
template <int G, int R>
__device__ __forceinline__ void f81_finish_child(const LaneCtx<R>& L, const PmlCols& c, const PmlState& st,
                                                 const double (&prod)[R], i64 pe, int ch, double e, double s_child,
                                                 i64 bec, const double (&v)[R], const double (&mb)[R],
                                                 double (&tdc)[R], i64& xe) {
    const double a = (1.0 - e) * s_child;
    double x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double cn = a + e * v[r];
        if (!(cn > 0.0)) cn = 1.0;
        x[r] = prod[r] * fast_rcp(cn);
    }
    xe = pe - bec;
    xe += lazy_rescale<G, R>(x);
    const double b = (1.0 - e) * pi_dot<G, R>(L, x);
    double lh[R];
    double lhs = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        tdc[r] = b + e * x[r];  // >= 0 by construction (ml.py:289's clamp is a no-op for the closed form)
        lh[r] = v[r] * tdc[r] * (L.pi_r[r] * mb[r]);
        lhs += lh[r];
    }
    lhs = group_sum<G>(lhs);
    const int lex = (lhs > 0.0 && !isinf(lhs)) ? exponent_of(lhs) : 0;
    const double inv = fast_rcp(lhs);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        // correctly rounded lh / lhs (one residual step), so that an observed tip gets exactly 1.0
        const double q = lh[r] * inv;
        lh[r] = fma(fma(-lhs, q, lh[r]), inv, q);
    }
    if (L.lane_valid) store_vec<R>(st.post + (L.colN + ch) * c.ks + L.s0, lh);
    if (L.g == 0) {
        st.lhsum[L.colN + ch] = __builtin_ldexp(lhs, -lex);
        st.lhe[L.colN + ch] = xe + bec + lex;
    }
}

// One unit = (stored internal node of the depth level, column): the parent's BU and TD vectors are loaded once and
// every child is finished from them; cherry children are recomputed and their tips finished in the same unit.
// replaces calc_node_td_likelihood (ml.py:273-290), calc_node_marginal_likelihood (:454-460) and the normalisation
// of convert_likelihoods_to_probabilities (:498-500) for the F81 family.
template <int G, int R>
__global__ void __launch_bounds__(PML_BLOCK)
td_f81_kernel(PmlTree t, PmlCols c, PmlState st, const int* __restrict__ level_parents, int n_level) {
    constexpr int UW = 64 / G;
    const int wave = threadIdx.x >> 6;
    const int sub = (threadIdx.x & 63) / G;
    LaneCtx<R> L;
    lane_ctx_init<G, R>(L, t, c);

    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    for (int base = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW; base < n_level; base += stride) {
        const int idx = base + sub;
        if (idx >= n_level) continue;
        const int p = level_parents[idx];

        double prod[R];
        {
            double bp[R], tp[R];
            node_load_vec<R>(L, c, st.bu, p, bp);
            node_load_vec<R>(L, c, st.td, p, tp);
#pragma unroll
            for (int r = 0; r < R; ++r) prod[r] = tp[r] * bp[r];
        }
        const i64 pe = st.te[L.colN + p] + st.be[L.colN + p];
        const int fc = t.first_child[p];
        const int nc = t.n_children[p];
        for (int j = 0; j < nc; ++j) {
            const int ch = fc + j;
            const double e = st.E[L.colN + ch];
            const int kd = node_kind(t, ch);
            double mb[R], v[R], tdc[R];
            node_mask_vec<R>(L, c, ch, mb);
            i64 xe;
            if (kd == PML_KIND_TIP) {
                f81_finish_child<G, R>(L, c, st, prod, pe, ch, e, st.S[L.colN + ch], 0, mb, mb, tdc, xe);
            } else if (kd == PML_KIND_STORED) {
                node_load_vec<R>(L, c, st.bu, ch, v);
                f81_finish_child<G, R>(L, c, st, prod, pe, ch, e, st.S[L.colN + ch], st.be[L.colN + ch], v, mb, tdc,
                                       xe);
                if (L.lane_valid) store_vec<R>(st.td + (L.colN + ch) * c.ks + L.s0, tdc);
                if (L.g == 0) st.te[L.colN + ch] = xe;
            } else {
                // cherry: rebuild its bottom-up vector, finish it, then finish its tips from registers
                i64 bec;
                f81_cherry_vector<G, R>(L, t, c, st, ch, v, bec, false);
                const double s_child = pi_dot<G, R>(L, v);
                f81_finish_child<G, R>(L, c, st, prod, pe, ch, e, s_child, bec, v, mb, tdc, xe);
                double prod2[R];
#pragma unroll
                for (int r = 0; r < R; ++r) prod2[r] = tdc[r] * v[r];
                const i64 pe2 = xe + bec;
                const int fc2 = t.first_child[ch];
                const int nc2 = t.n_children[ch];
                for (int q = 0; q < nc2; ++q) {
                    const int tip = fc2 + q;
                    double mt[R], tdt[R];
                    node_mask_vec<R>(L, c, tip, mt);
                    i64 xt;
                    f81_finish_child<G, R>(L, c, st, prod2, pe2, tip, st.E[L.colN + tip], st.S[L.colN + tip], 0, mt,
                                           mt, tdt, xt);
                }
            }
        }
    }
}
