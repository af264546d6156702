// F81-family sweeps (F81 / JC / EFT): P(t) = (1 - e) 1 pi^T + e I with e = exp(-mu t') is never materialised;
// P v = (1 - e)(pi . v) 1 + e v costs O(k) per branch (pastml/models/F81Model.py:28-46 in closed form).
#pragma once
#include "pml_device.h"

struct PmlTree {
    int N;
    int n_roots;
    const int* parent;
    const int* first_child;
    const int* n_children;
    const double* dist;
    const int* post_rank;
};

struct PmlCols {
    int k, ks, W;
    const u64* masks;       // [C][N][W]
    const u64* masks_init;  // [C][N][W] or nullptr
    const double* pi;       // [C][ks]
};

struct PmlState {
    double* E;      // [C][N]      F81: exp(-mu t') per branch
    double* bu;     // [C][N][ks]  bottom-up vectors (internal nodes only; tips are their masks)
    double* S;      // [C][N]      F81 marginal: pi . bu
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

// ---------------------------------------------------------------------------------------------------------------------
// bottom-up level kernel. One unit = (internal node of the level, column).
// replaces calc_node_bu_likelihood (pastml/ml.py:124-148) for the F81 family.
// ---------------------------------------------------------------------------------------------------------------------
template <int G, int R, bool JOINT>
__global__ void __launch_bounds__(PML_BLOCK)
bu_f81_kernel(PmlTree t, PmlCols c, PmlState st, const int* __restrict__ level_nodes, int n_level) {
    constexpr int UW = 64 / G;  // units per wave
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane & (G - 1);
    const int sub = lane / G;
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const int s0 = g * R;
    const int w0 = s0 >> 6;
    const bool lane_valid = s0 < c.ks;  // lanes beyond the (padded) vector neither load nor store

    double pi_r[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pi_r[r] = (s0 + r < c.k) ? c.pi[(size_t)col * c.ks + s0 + r] : 0.0;

    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    for (int base = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW; base < n_level; base += stride) {
        const int idx = base + sub;
        if (idx >= n_level) continue;  // whole groups drop out together
        const int n = level_nodes[idx];

        double acc[R];
        {
            const u64 word = lane_valid ? c.masks[(colN + n) * c.W + w0] : 0ull;
            mask_to_vec<R>(word, s0, c.k, acc);
        }
        i64 esum = 0;
        const int fc = t.first_child[n];
        const int nc = t.n_children[n];
        for (int j = 0; j < nc; ++j) {
            const int ch = fc + j;
            const double e = st.E[colN + ch];
            const bool tip = t.n_children[ch] == 0;
            double v[R];
            if (tip) {
                const u64 word = lane_valid ? c.masks[(colN + ch) * c.W + w0] : 0ull;
                mask_to_vec<R>(word, s0, c.k, v);
            } else {
                if (lane_valid) {
                    load_vec<R>(st.bu + (colN + ch) * c.ks + s0, v);
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = 0.0;
                }
                esum += st.be[colN + ch];
            }
            if (!JOINT) {
                const double a = (1.0 - e) * st.S[colN + ch];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] *= fmax(a + e * v[r], 0.0);
            } else {
                // row i of P * diag(v): off-diagonal entries w_j = ((1-e) pi_j) v_j, diagonal ((1-e) pi_i + e) v_i
                // (same rounding sequence as the reference's P * v broadcast, ml.py:130 with F81Model.py:46)
                const double ome = 1.0 - e;
                double w[R], dg[R];
                double m1 = -INFINITY;
                int j1 = 0x7fffffff;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double a = ome * pi_r[r];
                    const bool ok = s0 + r < c.k;
                    w[r] = ok ? a * v[r] : -INFINITY;
                    dg[r] = (a + e) * v[r];
                    if (ok && w[r] > m1) {
                        m1 = w[r];
                        j1 = s0 + r;
                    }
                }
                group_argmax_first<G>(m1, j1);
                double m2 = -INFINITY;
                int j2 = 0x7fffffff;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (s0 + r < c.k && s0 + r != j1 && w[r] > m2) {
                        m2 = w[r];
                        j2 = s0 + r;
                    }
                }
                group_argmax_first<G>(m2, j2);
                int jj[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = s0 + r;
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
                    jj[r] = (i < c.k) ? arg : 0;
                }
                // altered nodes get their tables rewritten w.r.t. their initial masks (ml.py:408-428)
                if (c.masks_init != nullptr) {
                    const u64* mi = c.masks_init + (colN + ch) * c.W;
                    const u64* mc = c.masks + (colN + ch) * c.W;
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
                if (lane_valid) store_vec_i32<R>(st.J + (colN + ch) * c.ks + s0, jj);
            }
            bool nz = false;
#pragma unroll
            for (int r = 0; r < R; ++r) nz |= acc[r] != 0.0;
            if (!group_any<G>(nz)) {
                // np.all(log_likelihood_array == -inf) after this child (ml.py:139)
                if (g == 0) atomicMin(&st.err[col], ((u64)(unsigned)t.post_rank[n] << 32) | (u64)(unsigned)ch);
            }
            esum += lazy_rescale<G, R>(acc);
        }
        if (!JOINT) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) s += pi_r[r] * acc[r];
            s = group_sum<G>(s);
            if (g == 0) st.S[colN + n] = s;
        }
        if (lane_valid) store_vec<R>(st.bu + (colN + n) * c.ks + s0, acc);
        if (g == 0) st.be[colN + n] = esum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// top-down + marginal likelihoods + posteriors, one unit = (parent node of the depth level, column): the parent's
// BU and TD vectors are loaded once and every child is finished from them.
// replaces calc_node_td_likelihood (ml.py:273-290), calc_node_marginal_likelihood (:454-460) and the normalisation
// of convert_likelihoods_to_probabilities (:498-500) for the F81 family.
// ---------------------------------------------------------------------------------------------------------------------
template <int G, int R>
__global__ void __launch_bounds__(PML_BLOCK)
td_f81_kernel(PmlTree t, PmlCols c, PmlState st, const int* __restrict__ level_parents, int n_level) {
    constexpr int UW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane & (G - 1);
    const int sub = lane / G;
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const int s0 = g * R;
    const int w0 = s0 >> 6;
    const bool lane_valid = s0 < c.ks;

    double pi_r[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pi_r[r] = (s0 + r < c.k) ? c.pi[(size_t)col * c.ks + s0 + r] : 0.0;

    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    for (int base = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW; base < n_level; base += stride) {
        const int idx = base + sub;
        if (idx >= n_level) continue;
        const int p = level_parents[idx];

        double prod[R];
        {
            double bp[R], tp[R];
            if (lane_valid) {
                load_vec<R>(st.bu + (colN + p) * c.ks + s0, bp);
                load_vec<R>(st.td + (colN + p) * c.ks + s0, tp);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) bp[r] = tp[r] = 0.0;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) prod[r] = tp[r] * bp[r];
        }
        const i64 pe = st.te[colN + p] + st.be[colN + p];
        const int fc = t.first_child[p];
        const int nc = t.n_children[p];
        for (int j = 0; j < nc; ++j) {
            const int ch = fc + j;
            const double e = st.E[colN + ch];
            const bool tip = t.n_children[ch] == 0;
            const u64 word = lane_valid ? c.masks[(colN + ch) * c.W + w0] : 0ull;
            double mb[R], v[R];
            mask_to_vec<R>(word, s0, c.k, mb);
            i64 bec = 0;
            if (tip) {
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = mb[r];
            } else {
                if (lane_valid) {
                    load_vec<R>(st.bu + (colN + ch) * c.ks + s0, v);
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = 0.0;
                }
                bec = st.be[colN + ch];
            }
            // the child's own message to the parent is divided out of the parent's vector (ml.py:279-283)
            const double a = (1.0 - e) * st.S[colN + ch];
            double x[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double cn = a + e * v[r];
                if (!(cn > 0.0)) cn = 1.0;
                x[r] = prod[r] / cn;
            }
            i64 xe = pe - bec;
            xe += lazy_rescale<G, R>(x);
            double sx = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) sx += pi_r[r] * x[r];
            sx = group_sum<G>(sx);
            const double b = (1.0 - e) * sx;
            double tdc[R], lh[R];
            double lhs = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                tdc[r] = fmax(b + e * x[r], 0.0);
                lh[r] = v[r] * tdc[r] * (pi_r[r] * mb[r]);
                lhs += lh[r];
            }
            lhs = group_sum<G>(lhs);
            if (!tip) {
                if (lane_valid) store_vec<R>(st.td + (colN + ch) * c.ks + s0, tdc);
                if (g == 0) st.te[colN + ch] = xe;
            }
            const int lex = (lhs > 0.0 && !isinf(lhs)) ? ilogb(lhs) : 0;
#pragma unroll
            for (int r = 0; r < R; ++r) lh[r] = lh[r] / lhs;
            if (lane_valid) store_vec<R>(st.post + (colN + ch) * c.ks + s0, lh);
            if (g == 0) {
                st.lhsum[colN + ch] = scalbn(lhs, -lex);
                st.lhe[colN + ch] = xe + bec + lex;
            }
        }
    }
}
