// Sweeps for models whose P(t) is materialised per branch (HKY, eigen-decomposed CUSTOM_RATES / JTT).
// P is stored transposed per branch: Pt[branch][j][i] = P[i][j] (row stride ks), so that for a fixed source state j
// the lanes of a unit (which own the target rows i) read consecutive doubles.
#pragma once
#include "pml_model.h"

#define PML_MAX_STATES 512       // F81 family; the matrix / eigen models stop at PML_MAX_STATES_MATRIX
#define PML_MAX_STATES_MATRIX 256

// Stages the R values of every lane of a unit into the unit's LDS slot; the unit's lanes belong to one wavefront and
// LDS operations of a wavefront complete in order, so the following reads by other lanes of the unit see them.
template <int R>
__device__ __forceinline__ void stage_vec(double* __restrict__ slot, int s0, const double (&v)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) slot[s0 + r] = v[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Rows of Pt are fetched PML_ROWS_AT_ONCE at a time (independent loads issued back to back, then consumed in order):
// a unit waits for memory k / PML_ROWS_AT_ONCE times per child instead of k times.  The order of the arithmetic is
// unchanged (j ascending), so the results are the bits of the one-row-at-a-time loop.
#define PML_ROWS_AT_ONCE 8

// HKY (k = 4, one state per lane): the sweeps build P(t) in registers instead of reading the materialised batch -- a
// branch's 4x4 matrix is 128 bytes against the 32 bytes of a state vector, four fifths of what these sweeps read.
// A lane keeps row i of P (its own state): PRow4.  Same closed form (hky_matrix), hence the same bits as the batch.
struct PRow4 {
    double e[4];  // P[i][j], j = 0 .. 3
};
__device__ __forceinline__ PRow4 hky_row(const double* __restrict__ pi, double kappa, double tt, int i) {
    double p[4][4];
    hky_matrix(pi, kappa, tt, p);
    PRow4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r.e[j] = i == 0 ? p[0][j] : (i == 1 ? p[1][j] : (i == 2 ? p[2][j] : p[3][j]));
    return r;
}
__device__ __forceinline__ double prow_pick(const PRow4& r, int j) {
    return j == 0 ? r.e[0] : (j == 1 ? r.e[1] : (j == 2 ? r.e[2] : r.e[3]));
}
// out = sum_j P[i][j] v[j], j ascending as matvec_rows (the same additions in the same order)
__device__ __forceinline__ void matvec_prow(const PRow4& r, const double* __restrict__ slot, double (&out)[1]) {
    out[0] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[0] += r.e[j] * slot[j];
}

#define PML_P_MATERIALISED 0
#define PML_P_HKY 1

// out[i] = sum_j P[i][j] v[j]  (rows i = s0 .. s0+R-1 of this lane), v staged in LDS
template <int R>
__device__ __forceinline__ void matvec_rows(const double* __restrict__ Pt, int k, int ks, int s0, bool lane_valid,
                                            const double* __restrict__ slot, double (&out)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = 0.0;
    if (!lane_valid) return;
    for (int j0 = 0; j0 < k; j0 += PML_ROWS_AT_ONCE) {
        double p[PML_ROWS_AT_ONCE][R];
#pragma unroll
        for (int u = 0; u < PML_ROWS_AT_ONCE; ++u) {
            const int j = j0 + u < k ? j0 + u : k - 1;  // rows past the end re-read the last one and are not used
            load_vec<R>(Pt + (size_t)j * ks + s0, p[u]);
        }
#pragma unroll
        for (int u = 0; u < PML_ROWS_AT_ONCE; ++u) {
            if (j0 + u < k) {
                const double vj = slot[j0 + u];
#pragma unroll
                for (int r = 0; r < R; ++r) out[r] += p[u][r] * vj;
            }
        }
    }
}

// replaces calc_node_bu_likelihood (pastml/ml.py:124-148) for materialised P
// SRC = PML_P_HKY (only G = 4, R = 1, k = 4): P(t) of a child's branch from the closed form, in registers.
template <int G, int R, bool JOINT, int SRC = PML_P_MATERIALISED>
__global__ void __launch_bounds__(PML_BLOCK)
bu_matrix_kernel(PmlTree t, PmlCols c, PmlState st, const double* __restrict__ P, PmlModel m,
                 const int* __restrict__ level_nodes, int n_level) {
    static_assert(SRC == PML_P_MATERIALISED || (G == 4 && R == 1), "the HKY source is for one state per lane, k = 4");
    constexpr int UW = 64 / G;
    __shared__ double lds[PML_WAVES_PER_BLOCK * UW][G * R];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane & (G - 1);
    const int sub = lane / G;
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const int s0 = g * R;
    const int w0 = s0 >> 6;
    const bool lane_valid = s0 < c.ks;
    const size_t pstride = (size_t)c.k * c.ks;
    double* slot = lds[wave * UW + sub];

    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    for (int base = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW; base < n_level; base += stride) {
        const int idx = base + sub;
        if (idx >= n_level) continue;
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
            const bool tip = t.n_children[ch] == 0;
            PRow4 prow;
            if (SRC == PML_P_HKY)
                prow = hky_row(c.pi + (size_t)col * c.ks, m.kappa[col],
                               (t.dist[ch] + m.tau[col]) * m.tauf[col] * m.sf[col], s0);
            double v[R];
            int observed = -1;  // state of an observed tip (exactly one allowed state)
            if (tip) {
                int cnt = 0, first = 0;
                for (int w_ = 0; w_ < c.W; ++w_) {
                    const u64 mw = c.masks[(colN + ch) * c.W + w_];
                    if (cnt == 0 && mw) first = w_ * 64 + __builtin_ctzll(mw);
                    cnt += __popcll(mw);
                }
                if (cnt == 1 && first < c.k) observed = first;
            }
            if (observed >= 0) {
                // v is a unit vector: sum_j P[i][j] v[j] is P[i][s] exactly (the other terms are zeros), and the
                // arg-max scan has a closed form, so one 8k-byte row of Pt is read instead of the whole matrix
                const double* Pt = P + (colN + ch) * pstride;
                double msg[R];
                if (SRC == PML_P_HKY) {
                    msg[0] = prow_pick(prow, observed);
                } else if (lane_valid) {
                    load_vec<R>(Pt + (size_t)observed * c.ks + s0, msg);
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) msg[r] = 0.0;
                }
                if (JOINT) {
                    int jj[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        // numpy's first maximum of (.., 0, P[i][s], 0, ..): s if P[i][s] > 0, else the first zero
                        const double pv = msg[r];
                        if (pv > 0.0) {
                            jj[r] = observed;
                        } else if (observed != 0) {
                            msg[r] = 0.0;
                            jj[r] = 0;
                        } else if (pv == 0.0 || c.k == 1) {
                            jj[r] = 0;
                        } else {
                            msg[r] = 0.0;
                            jj[r] = 1;
                        }
                        if (s0 + r >= c.k) {
                            msg[r] = 0.0;
                            jj[r] = 0;
                        }
                    }
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
                    if (lane_valid) store_vec_u8<R>(st.J + (colN + ch) * c.ks + s0, jj);
                }
                bool nz = false;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r] *= fmax(msg[r], 0.0);
                    nz |= acc[r] != 0.0;
                }
                if (!group_any<G>(nz)) {
                    if (g == 0) atomicMin(&st.err[col], ((u64)(unsigned)t.post_rank[n] << 32) | (u64)(unsigned)ch);
                }
                esum += lazy_rescale<G, R>(acc);
                continue;
            }
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
            stage_vec<R>(slot, s0, v);
            const double* Pt = P + (colN + ch) * pstride;
            double msg[R];
            if (!JOINT) {
                if (SRC == PML_P_HKY) {
                    double m1[1];
                    matvec_prow(prow, slot, m1);
                    msg[0] = m1[0];
                } else {
                    matvec_rows<R>(Pt, c.k, c.ks, s0, lane_valid, slot, msg);
                }
            } else {
                int jj[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    msg[r] = -INFINITY;
                    jj[r] = 0;
                }
                if (SRC == PML_P_HKY) {
#pragma unroll
                    for (int jc = 0; jc < 4; ++jc) {
                        const double pr = prow.e[jc] * slot[jc];
                        if (pr > msg[0]) {
                            msg[0] = pr;
                            jj[0] = jc;
                        }
                    }
                } else if (lane_valid) {
                    // sequential scan over j keeps numpy's first-maximum semantics (ml.py:134)
                    for (int j0 = 0; j0 < c.k; j0 += PML_ROWS_AT_ONCE) {
                        double p[PML_ROWS_AT_ONCE][R];
#pragma unroll
                        for (int u = 0; u < PML_ROWS_AT_ONCE; ++u) {
                            const int jc = j0 + u < c.k ? j0 + u : c.k - 1;
                            load_vec<R>(Pt + (size_t)jc * c.ks + s0, p[u]);
                        }
#pragma unroll
                        for (int u = 0; u < PML_ROWS_AT_ONCE; ++u) {
                            const int jc = j0 + u;
                            if (jc < c.k) {
                                const double vj = slot[jc];
#pragma unroll
                                for (int r = 0; r < R; ++r) {
                                    const double pr = p[u][r] * vj;
                                    if (pr > msg[r]) {
                                        msg[r] = pr;
                                        jj[r] = jc;
                                    }
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (s0 + r >= c.k) {
                        msg[r] = 0.0;
                        jj[r] = 0;
                    }
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
                if (lane_valid) store_vec_u8<R>(st.J + (colN + ch) * c.ks + s0, jj);
            }
            bool nz = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r] *= fmax(msg[r], 0.0);
                nz |= acc[r] != 0.0;
            }
            if (!group_any<G>(nz)) {
                if (g == 0) atomicMin(&st.err[col], ((u64)(unsigned)t.post_rank[n] << 32) | (u64)(unsigned)ch);
            }
            esum += lazy_rescale<G, R>(acc);
            __builtin_amdgcn_wave_barrier();  // all reads of the slot are done before the next child overwrites it
        }
        if (lane_valid) store_vec<R>(st.bu + (colN + n) * c.ks + s0, acc);
        if (g == 0) st.be[colN + n] = esum;
    }
}

// replaces calc_node_td_likelihood (ml.py:273-290) + marginals (:454-460, :498-500) for materialised P
template <int G, int R, int SRC = PML_P_MATERIALISED>
__global__ void __launch_bounds__(PML_BLOCK)
td_matrix_kernel(PmlTree t, PmlCols c, PmlState st, const double* __restrict__ P, PmlModel m,
                 const int* __restrict__ level_parents, int n_level) {
    static_assert(SRC == PML_P_MATERIALISED || (G == 4 && R == 1), "the HKY source is for one state per lane, k = 4");
    constexpr int UW = 64 / G;
    __shared__ double lds[PML_WAVES_PER_BLOCK * UW][G * R];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane & (G - 1);
    const int sub = lane / G;
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const int s0 = g * R;
    const int w0 = s0 >> 6;
    const bool lane_valid = s0 < c.ks;
    const size_t pstride = (size_t)c.k * c.ks;
    double* slot = lds[wave * UW + sub];

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
            const double* Pt = P + (colN + ch) * pstride;
            PRow4 prow;
            if (SRC == PML_P_HKY)
                prow = hky_row(c.pi + (size_t)col * c.ks, m.kappa[col],
                               (t.dist[ch] + m.tau[col]) * m.tauf[col] * m.sf[col], s0);
            stage_vec<R>(slot, s0, v);
            double cn[R], x[R];
            if (SRC == PML_P_HKY) {
                double m1[1];
                matvec_prow(prow, slot, m1);
                cn[0] = m1[0];
            } else {
                matvec_rows<R>(Pt, c.k, c.ks, s0, lane_valid, slot, cn);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double d = (cn[r] > 0.0) ? cn[r] : 1.0;
                x[r] = prod[r] / d;
            }
            i64 xe = pe - bec;
            xe += lazy_rescale<G, R>(x);
            __builtin_amdgcn_wave_barrier();
            stage_vec<R>(slot, s0, x);
            double tdc[R], lh[R];
            if (SRC == PML_P_HKY) {
                double m1[1];
                matvec_prow(prow, slot, m1);
                tdc[0] = m1[0];
            } else {
                matvec_rows<R>(Pt, c.k, c.ks, s0, lane_valid, slot, tdc);
            }
            double lhs = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                tdc[r] = fmax(tdc[r], 0.0);
                lh[r] = v[r] * tdc[r] * (pi_r[r] * mb[r]);
                lhs += lh[r];
            }
            lhs = group_sum<G>(lhs);
            if (!tip) {
                if (lane_valid) store_vec<R>(st.td + (colN + ch) * c.ks + s0, tdc);
                if (g == 0) st.te[colN + ch] = xe;
            }
            const int lex = (lhs > 0.0 && !isinf(lhs)) ? exponent_of(lhs) : 0;
#pragma unroll
            for (int r = 0; r < R; ++r) lh[r] = lh[r] / lhs;
            if (lane_valid) store_vec<R>(st.post + (colN + ch) * c.ks + s0, lh);
            if (g == 0) {
                st.lhsum[colN + ch] = __builtin_ldexp(lhs, -lex);
                st.lhe[colN + ch] = xe + bec + lex;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}
