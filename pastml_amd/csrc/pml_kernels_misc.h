// Model-independent kernels: masks from tip states, root handling, log-likelihood, joint back-trace.
#pragma once
#include "pml_kernels_f81.h"

#define PML_MAX_STATES_SEL 512

// masks: internal nodes all ones; tip j one-hot at states[col][j] (all ones if negative = missing data)
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
masks_fill_kernel(int N, int W, int k, u64* __restrict__ masks, int col_begin) {
    const int col = col_begin + blockIdx.y;
    const size_t total = (size_t)N * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int bits = min(64, k - w * 64);
        masks[(size_t)col * total + i] = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    }
}
#endif

#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
masks_tips_kernel(int N, int W, int k, u64* __restrict__ masks, int col_begin, int n_tips,
                  const int* __restrict__ tip_ids, const int* __restrict__ states) {
    const int col = col_begin + blockIdx.y;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_tips; j += gridDim.x * blockDim.x) {
        const int s = states[(size_t)blockIdx.y * n_tips + j];
        if (s < 0) continue;
        u64* m = masks + ((size_t)col * N + tip_ids[j]) * W;
        for (int w = 0; w < W; ++w) m[w] = (s >> 6) == w ? (1ull << (s & 63)) : 0ull;
    }
}
#endif

// Roots: TD = 1 with exponent 0 (ml.py:274-277) and their marginal likelihoods / posteriors (BU * pi * mask).
template <int G, int R>
__global__ void __launch_bounds__(PML_BLOCK)
td_roots_kernel(PmlTree t, PmlCols c, PmlState st) {
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
    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    for (int base = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW; base < t.n_roots; base += stride) {
        const int n = base + sub;
        if (n >= t.n_roots) continue;
        const bool tip = t.n_children[n] == 0;
        const u64 word = lane_valid ? c.masks[(colN + n) * c.W + w0] : 0ull;
        double mb[R], v[R], one[R], lh[R];
        mask_to_vec<R>(word, s0, c.k, mb);
        if (tip || !lane_valid) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = mb[r];
        } else {
            load_vec<R>(st.bu + (colN + n) * c.ks + s0, v);
        }
        double lhs = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            one[r] = (s0 + r < c.k) ? 1.0 : 0.0;
            const double p = (s0 + r < c.k) ? c.pi[(size_t)col * c.ks + s0 + r] : 0.0;
            lh[r] = v[r] * (p * mb[r]);
            lhs += lh[r];
        }
        lhs = group_sum<G>(lhs);
        const int lex = (lhs > 0.0 && !isinf(lhs)) ? exponent_of(lhs) : 0;
#pragma unroll
        for (int r = 0; r < R; ++r) lh[r] = lh[r] / lhs;
        if (lane_valid) {
            if (st.td != nullptr) store_vec<R>(st.td + (colN + n) * c.ks + s0, one);
            store_vec<R>(st.post + (colN + n) * c.ks + s0, lh);
        }
        if (g == 0) {
            if (st.td != nullptr) st.te[colN + n] = 0;
            st.lhsum[colN + n] = __builtin_ldexp(lhs, -lex);
            st.lhe[colN + n] = (tip ? 0 : st.be[colN + n]) + lex;
        }
    }
}

// ln L per column = sum over trees of ln(root term) + E_root ln 2 (ml.py:112-121).
// marginal: root term = sum_i pi_i BU_i; joint: max_i pi_i BU_i, whose first arg-max is the root's joint state
// (ml.py:622).  One thread per column; forests have few roots.
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
loglik_kernel(PmlTree t, PmlCols c, PmlState st, int n_cols, int is_marginal, double* __restrict__ loglik,
              u64* __restrict__ err_out) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_cols) return;
    // loglik / err_out are pinned host memory: the results land where the caller reads them, no copy is queued
    loglik[col] = column_loglik(t, c, st, col, is_marginal);
    err_out[col] = st.err[col];
    // (a later launch of the pass may raise the completion word the host spins on: these writes must be visible to the
    // host before anything that launch publishes)
    __threadfence_system();
}
#endif

// an entry of the arg-max tables (one byte; two beyond 256 states)
__device__ __forceinline__ int table_entry(const PmlState& st, size_t entry) {
    return st.jt16 ? (int)reinterpret_cast<const unsigned short*>(st.J)[entry] : (int)st.J[entry];
}

// joint back-trace, one depth level per launch: state[n] = table[n][state[parent]] (ml.py:615-620)
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
joint_backtrace_kernel(PmlTree t, PmlCols c, PmlState st, int begin, int end) {
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    for (int n = begin + blockIdx.x * blockDim.x + threadIdx.x; n < end; n += gridDim.x * blockDim.x) {
        const int ps = st.js[colN + t.parent[n]];
        st.js[colN + n] = table_entry(st, (colN + n) * c.ks + ps);
    }
}
#endif

// the depth levels right below the roots (a handful of nodes each) in one launch: one workgroup per column, a
// workgroup barrier between levels; depth_offsets[d] .. depth_offsets[d + 1] are the node ids of depth d
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
joint_backtrace_narrow_kernel(PmlTree t, PmlCols c, PmlState st, const int* __restrict__ depth_offsets, int first_depth,
                              int n_depths) {
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    for (int d = first_depth; d < first_depth + n_depths; ++d) {
        const int begin = depth_offsets[d], end = depth_offsets[d + 1];
        for (int n = begin + threadIdx.x; n < end; n += blockDim.x) {
            const int ps = st.js[colN + t.parent[n]];
            st.js[colN + n] = table_entry(st, (colN + n) * c.ks + ps);
        }
        __syncthreads();
    }
}
#endif

// The depths below the narrow end in tiers: a tier of depths is cut into the subtrees hanging off its first depth, one
// workgroup per (subtree, column) walks its depths with a workgroup barrier between them -- one launch per tier instead
// of one per depth (the work is one table look-up per node: a depth is pure launch latency).  nodes: the tier's nodes,
// subtree by subtree, depth by depth; subtree b's depth table starts at lv[blk_start[b]], n_depths + 1 entries.
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
joint_backtrace_blocks_kernel(PmlTree t, PmlCols c, PmlState st, const int* __restrict__ nodes,
                              const int* __restrict__ lv, const int* __restrict__ blk_start, int n_depths) {
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const int* off = lv + blk_start[blockIdx.x];
    for (int d = 0; d < n_depths; ++d) {
        const int begin = off[d], end = off[d + 1];
        for (int q = begin + threadIdx.x; q < end; q += blockDim.x) {
            const int n = nodes[q];
            const int ps = st.js[colN + t.parent[n]];
            st.js[colN + n] = table_entry(st, (colN + n) * c.ks + ps);
        }
        __syncthreads();
    }
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// State selection from the marginal posteriors: MAP (pastml/ml.py:577-595) and MPPA (pastml/ml.py:505-574).
// One unit = (node, column), G lanes with R contiguous states each; the unit's vectors live in LDS.
//   lh_i   = posterior_i * [i allowed by lh_mask]          (the reference multiplies LH by the '.initial' masks)
//   MAP    : first arg-max of lh
//   MPPA   : p = lh / sum(lh); q = p sorted ascending, with the joint state's probability moved last if force_joint;
//            best m = first minimiser of sum_i (u_m[i] - q[i])^2, u_m = (0,..,0, 1/m x m); the m states with the
//            largest lh are kept (ties: lower index first).
// The selected masks replace the columns' allowed-state masks (ready for the restricted sweep).
// ---------------------------------------------------------------------------------------------------------------------
template <int G, int R>
__global__ void __launch_bounds__(PML_BLOCK)
select_states_kernel(int N, int k, int ks, int W, const double* __restrict__ post, const u64* __restrict__ lh_mask,
                     const int* __restrict__ js, int method, int force_joint, u64* __restrict__ masks,
                     int* __restrict__ n_states) {
    constexpr int UW = 64 / G;
    constexpr int KP = G * R;
    __shared__ double s_lh[PML_WAVES_PER_BLOCK * UW][KP];
    __shared__ double s_q[PML_WAVES_PER_BLOCK * UW][KP];
    __shared__ double s_p[PML_WAVES_PER_BLOCK * UW][KP];
    __shared__ u64 s_words[PML_WAVES_PER_BLOCK * UW][(PML_MAX_STATES_SEL + 63) / 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane & (G - 1);
    const int sub = lane / G;
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * N;
    const int s0 = g * R;
    const int unit = wave * UW + sub;
    double* lhv = s_lh[unit];
    double* qv = s_q[unit];
    double* pv = s_p[unit];
    u64* words = s_words[unit];
    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * UW;
    for (int base = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * UW; base < N; base += stride) {
        const int n = base + sub;
        if (n >= N) continue;
        double lh[R];
        double sum = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = s0 + r;
            double v = 0.0;
            if (s < k) {
                v = post[(colN + n) * ks + s];
                if (lh_mask != nullptr && !((lh_mask[(colN + n) * W + (s >> 6)] >> (s & 63)) & 1ull)) v = 0.0;
            }
            lh[r] = v;
            sum += v;
        }
        if (g < W) words[g] = 0ull;
        for (int w = G; w < W; w += G)
            if (g + w < W) words[g + w] = 0ull;
        sum = group_sum<G>(sum);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int best_k = 1;
        const int ji = (method != 0 && force_joint) ? js[colN + n] : -1;
        bool general = method != 0;
        bool lds_filled = false;  // lhv / pv are only needed by the reference-shaped scans below
        // The probabilities are lh / sum; with sum > 0 every comparison below is made on the likelihoods themselves.
        // With q the ascending probabilities (joint state last) and T_m the sum of the last m of them,
        //     sum_i (u_m[i] - q[i])^2 = sum q^2 + (1 - 2 T_m) / m,
        // so the best m is the first minimiser of f(m) = (1 - 2 T_m) / m.
        //  * A dominant state settles it without sorting: T_m <= 1 gives f(m) - f(1) >= (2 q_last - 1) - 1 / 2 for
        //    every m >= 2, which is >= 0.02 -- far from any rounding -- once q_last >= 0.76 (observed tips, and most
        //    nodes of most trees).
        //  * Otherwise the candidates are walked from the top of the sorted order -- the joint state, then the largest
        //    remaining likelihood (of equal ones the higher index: it has the higher ascending rank), one arg-max
        //    butterfly each.  T_m <= 1 bounds every later f from below by -1 / m, so the walk stops as soon as the
        //    best value so far is below that: after two or three candidates for a typical node instead of k^2
        //    comparisons plus k sums of k terms.
        // A node whose likelihoods do not sum to a positive finite number takes the reference-shaped scan.
        if (general && sum > 0.0 && !isinf(sum)) {
            general = false;
            double l_last;
            if (ji >= 0) {
                l_last = -INFINITY;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (s0 + r == ji) l_last = lh[r];
                l_last = group_max<G>(l_last);
            } else {
                l_last = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) l_last = fmax(l_last, (s0 + r < k) ? lh[r] : 0.0);
                l_last = group_max<G>(l_last);
            }
            if (!(l_last >= 0.76 * sum)) {
                double cand[R];
#pragma unroll
                for (int r = 0; r < R; ++r) cand[r] = (s0 + r < k) ? lh[r] : -INFINITY;
                double L2 = 0.0, best_f = INFINITY;  // f scaled by sum: (sum - 2 L_m) / m
                int best_m = k;
                for (int m = 1; m <= k; ++m) {
                    double bv = -INFINITY;
                    int bi = -1;
                    if (m == 1 && ji >= 0 && ji < k) {
                        bv = l_last;
                        bi = ji;
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            if (cand[r] >= bv && cand[r] > -INFINITY) {  // r ascends: of equal ones the higher index
                                bv = cand[r];
                                bi = s0 + r;
                            }
                        }
                        int nbi = -bi;  // the butterfly keeps the LOWER index among equal values: negate the indices
                        group_argmax_first<G>(bv, nbi);
                        bi = -nbi;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (s0 + r == bi) cand[r] = -INFINITY;
                    L2 += bv;
                    const double f = (sum - 2.0 * L2) / (double)m;
                    if (f < best_f) {  // m ascends: strict < keeps the first minimum
                        best_f = f;
                        best_m = m;
                    }
                    if (best_f < -(1.0 + 1e-9) * sum / (double)(m + 1)) break;
                }
                best_k = best_m;
            }
        }
        if (general) {
            // reference-shaped scan: the likelihoods and the probabilities of the unit in LDS
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (s0 + r < KP) lhv[s0 + r] = lh[r];
                if (s0 + r < k) pv[s0 + r] = lh[r] / sum;
            }
            lds_filled = true;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (general) {
            // ascending ranks (joint state last)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = s0 + r;
                if (i >= k) continue;
                const double pi_ = pv[i];
                int rank = 0;
                if (i == ji) {
                    rank = k - 1;
                } else {
                    for (int j = 0; j < k; ++j) {
                        if (j == ji || j == i) continue;
                        const double pj = pv[j];
                        rank += (pj < pi_ || (pj == pi_ && j < i)) ? 1 : 0;
                    }
                }
                qv[rank] = pi_;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double best_c = INFINITY;
            int best_m = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int m = s0 + r + 1;
                if (m > k) continue;
                const double inv = 1.0 / (double)m;
                double corr = 0.0;
                for (int i = 0; i < k; ++i) {
                    const double d = (i < k - m ? 0.0 : inv) - qv[i];
                    corr += d * d;
                }
                if (corr < best_c) {  // ascending m inside a lane: strict < keeps the first minimum
                    best_c = corr;
                    best_m = m;
                }
            }
            // first minimum over the lanes: smaller corr, ties -> smaller m (NaN never wins: m stays k as in numpy)
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {
                const double oc = __shfl_xor(best_c, o, 64);
                const int om = __shfl_xor(best_m, o, 64);
                if (oc < best_c || (oc == best_c && om < best_m)) {
                    best_c = oc;
                    best_m = om;
                }
            }
            best_k = (best_m > k) ? k : best_m;
        }
        // keep the best_k states with the largest lh (stable: ties go to the lower index)
        if (best_k == 1) {
            // the first maximum of lh: one arg-max butterfly instead of ranking every state
            double bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (s0 + r < k && lh[r] > bv) {  // a NaN never wins: with all-NaN likelihoods nothing is kept,
                    bv = lh[r];                  // as ranks computed by comparisons would have it
                    bi = s0 + r;
                }
            }
            group_argmax_first<G>(bv, bi);
            if (g == 0 && bi < k) words[bi >> 6] = 1ull << (bi & 63);
        } else if (best_k <= 8) {
            // a few arg-max butterflies: the largest remaining lh, of equal ones the lower index
            double cand[R];
#pragma unroll
            for (int r = 0; r < R; ++r) cand[r] = (s0 + r < k) ? lh[r] : -INFINITY;
            for (int m = 0; m < best_k; ++m) {
                double bv = -INFINITY;
                int bi = 0x7fffffff;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (cand[r] > bv) {
                        bv = cand[r];
                        bi = s0 + r;
                    }
                }
                group_argmax_first<G>(bv, bi);
                if (bi >= k) break;  // nothing comparable is left (NaN likelihoods)
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (s0 + r == bi) {
                        cand[r] = -INFINITY;
                        atomicOr(&words[bi >> 6], 1ull << (bi & 63));
                    }
            }
        } else {
            if (!lds_filled) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (s0 + r < KP) lhv[s0 + r] = lh[r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = s0 + r;
                if (i >= k) continue;
                int rank = 0;
                for (int j = 0; j < k; ++j) {
                    const double lj = lhv[j];
                    rank += (lj > lh[r] || (lj == lh[r] && j < i)) ? 1 : 0;
                }
                if (rank < best_k) atomicOr(&words[i >> 6], 1ull << (i & 63));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int w = g; w < W; w += G) masks[(colN + n) * W + w] = words[w];
        if (g == 0) n_states[colN + n] = best_k;
        __builtin_amdgcn_wave_barrier();
    }
}

// err[col] = all ones (a kernel rather than a memset node so that the sweep's launch sequence captures cleanly)
// ---------------------------------------------------------------------------------------------------------------------
// Top-down vectors of the nodes the sweeps never store -- tips, and in the F81 family the cherries that only live in
// registers -- computed on request (pml_download(PML_BUF_TD / PML_BUF_TD_SF)) so that every non-root node has what
// calc_node_td_likelihood (pastml/ml.py:273-290) leaves on it:
//     msg[i] = sum_j P[i][j] BU_n[j]   (<= 0 -> 1),   X = TD_p o BU_p / msg,   TD_n[i] = max(sum_j P[i][j] X[j], 0)
// One wavefront per (node, column), the levels of the tree top-down (a cherry's tips need the cherry's vector).
// Inspection only, not on the measured path: plain loops, k <= 512 states in up to eight registers per lane.
// P: transposed matrices of the matrix models ([C][N][k][ks], Pt[j][i] = P[i][j]) or nullptr for the F81 family, whose
// P = (1 - e) 1 pi^T + e I is applied in closed form.
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
td_fill_kernel(PmlTree t, PmlCols c, PmlState st, const double* __restrict__ P, int f81, int begin, int end) {
    __shared__ double sv[PML_WAVES_PER_BLOCK][PML_MAX_STATES_SEL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y;
    const int k = c.k, ks = c.ks;
    const size_t colN = (size_t)col * t.N;
    for (int n = begin + blockIdx.x * PML_WAVES_PER_BLOCK + wave; n < end; n += gridDim.x * PML_WAVES_PER_BLOCK) {
        const int kind = node_kind(t, n);
        const bool tip = t.n_children[n] == 0;
        const bool stored = f81 ? kind == PML_KIND_STORED : !tip;
        if (stored) continue;  // wave-uniform
        const int p = t.parent[n];
        const size_t row = (colN + n) * ks, prow = (colN + p) * ks;
        const double* Pt = P != nullptr ? P + (colN + n) * (size_t)k * ks : nullptr;
        const i64 be_n = tip ? 0 : st.be[colN + n];
        double v[PML_MAX_STATES_SEL / 64], x[PML_MAX_STATES_SEL / 64];
        for (int q = 0; q < PML_MAX_STATES_SEL / 64; ++q) {
            const int i = lane + 64 * q;
            v[q] = 0.0;
            if (i < k) v[q] = tip ? (double)((c.masks[(colN + n) * c.W + (i >> 6)] >> (i & 63)) & 1ull) : st.bu[row + i];
            if (i < k) sv[wave][i] = v[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double e = f81 ? st.E[colN + n] : 0.0;
        const double s_n = f81 ? st.S[colN + n] : 0.0;
        double big = 0.0;
        bool out_of_band = false;
        for (int q = 0; q < PML_MAX_STATES_SEL / 64; ++q) {
            const int i = lane + 64 * q;
            x[q] = 0.0;
            if (i >= k) continue;
            double msg;
            if (f81) {
                msg = (1.0 - e) * s_n + e * v[q];
            } else {
                msg = 0.0;
                for (int j = 0; j < k; ++j) msg += Pt[(size_t)j * ks + i] * sv[wave][j];
            }
            if (!(msg > 0.0)) msg = 1.0;
            x[q] = st.td[prow + i] * st.bu[prow + i] / msg;
            big = fmax(big, x[q]);
            out_of_band |= x[q] != 0.0 && (x[q] < 0x1p-200 || x[q] > 0x1p+200);
        }
        i64 xe = st.te[colN + p] + st.be[colN + p] - be_n;
        if (__any(out_of_band)) {
            for (int off = 32; off > 0; off >>= 1) big = fmax(big, __shfl_xor(big, off, 64));
            if (big > 0.0 && !isinf(big)) {
                const int ex = exponent_of(big);
                for (int q = 0; q < PML_MAX_STATES_SEL / 64; ++q) x[q] = __builtin_ldexp(x[q], -ex);
                xe += ex;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int q = 0; q < PML_MAX_STATES_SEL / 64; ++q) {
            const int i = lane + 64 * q;
            if (i < k) sv[wave][i] = x[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double dot = 0.0;
        if (f81) {
            for (int q = 0; q < PML_MAX_STATES_SEL / 64; ++q) {
                const int i = lane + 64 * q;
                if (i < k) dot += c.pi[(size_t)col * ks + i] * x[q];
            }
            for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
        }
        for (int q = 0; q < PML_MAX_STATES_SEL / 64; ++q) {
            const int i = lane + 64 * q;
            if (i >= ks) continue;
            double td = 0.0;
            if (i < k) {
                if (f81) {
                    td = (1.0 - e) * dot + e * x[q];
                } else {
                    for (int j = 0; j < k; ++j) td += Pt[(size_t)j * ks + i] * sv[wave][j];
                }
            }
            st.td[row + i] = fmax(td, 0.0);
        }
        if (lane == 0) st.te[colN + n] = xe;
        __builtin_amdgcn_wave_barrier();
    }
}
#endif

// sum of the columns' log-likelihoods in column order (as a host loop over the characters would add them), on the device:
// the values are where the last kernel of the sweep put them (pinned host memory, visible to the device)
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void sum_loglik_kernel(const double* __restrict__ loglik, int n, double* __restrict__ total) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += loglik[i];
        total[0] = acc;
    }
}
#endif

// PML_OPT_IMPLICIT_TIP_POSTERIORS: the rows the top-down sweep left implicit -- an observed tip (one allowed state) with a
// positive finite likelihood has the unit vector of its state as its posterior (pastml/ml.py:498-500 gives exactly that)
// -- written out when somebody reads the table.  One thread per (tip, column).
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
tip_posteriors_kernel(PmlCols c, PmlState st, int N, const int* __restrict__ tips, int n_tips) {
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * N;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n_tips; q += gridDim.x * blockDim.x) {
        const int tip = tips[q];
        const u64 word = c.masks[colN + tip] & (c.k >= 64 ? ~0ull : (1ull << c.k) - 1ull);
        const double ls = st.lhsum[colN + tip];
        if (__popcll(word) != 1 || !(ls > 0.0) || isinf(ls)) continue;  // (such rows were written by the sweep)
        const int s = __builtin_ctzll(word);
        double* row = st.post + (colN + tip) * c.ks;
        for (int i = 0; i < c.ks; ++i) row[i] = i == s ? 1.0 : 0.0;
    }
}
#endif

// start of a bottom-up sweep: the columns' error words, and (if given) the counters of the tips the lean tips kernel of
// the eigen joint sweep hands on
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void reset_err_kernel(u64* __restrict__ err, int n, int* __restrict__ counters = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        err[i] = ~0ull;
        if (counters != nullptr) counters[i] = 0;
    }
}
#endif

// Rows of a per-node table from the library's numbering into the caller's, on the device (outputs of a renumbered forest:
// the copy to the host is then one contiguous transfer and the host never holds a second copy of the table).  Row i of dst
// (words_dst 4-byte words) = the first words_dst words of row map[first + i * stride] of src (rows of words_src words).
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
gather_rows_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, const int* __restrict__ map, long long n_rows,
                   int words_dst, int words_src, int first, int stride) {
    const long long total = n_rows * words_dst;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / words_dst;
        const int w = (int)(idx - row * words_dst);
        dst[idx] = src[(size_t)map[first + row * stride] * words_src + w];
    }
}
#endif

