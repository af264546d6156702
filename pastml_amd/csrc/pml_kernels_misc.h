// Model-independent kernels: masks from tip states, root handling, log-likelihood, joint back-trace.
#pragma once
#include "pml_kernels_f81.h"

// masks: internal nodes all ones; tip j one-hot at states[col][j] (all ones if negative = missing data)
__global__ void __launch_bounds__(PML_BLOCK)
masks_fill_kernel(int N, int W, int k, u64* __restrict__ masks, int col_begin) {
    const int col = col_begin + blockIdx.y;
    const size_t total = (size_t)N * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int bits = min(64, k - w * 64);
        masks[(size_t)col * total + i] = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    }
}

__global__ void __launch_bounds__(PML_BLOCK)
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
            store_vec<R>(st.td + (colN + n) * c.ks + s0, one);
            store_vec<R>(st.post + (colN + n) * c.ks + s0, lh);
        }
        if (g == 0) {
            st.te[colN + n] = 0;
            st.lhsum[colN + n] = __builtin_ldexp(lhs, -lex);
            st.lhe[colN + n] = (tip ? 0 : st.be[colN + n]) + lex;
        }
    }
}

// ln L per column = sum over trees of ln(root term) + E_root ln 2 (ml.py:112-121).
// marginal: root term = sum_i pi_i BU_i; joint: max_i pi_i BU_i, whose first arg-max is the root's joint state
// (ml.py:622).  One thread per column; forests have few roots.
__global__ void __launch_bounds__(PML_BLOCK)
loglik_kernel(PmlTree t, PmlCols c, PmlState st, int n_cols, int is_marginal, double* __restrict__ loglik) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_cols) return;
    loglik[col] = column_loglik(t, c, st, col, is_marginal);
}

// joint back-trace, one depth level per launch: state[n] = table[n][state[parent]] (ml.py:615-620)
__global__ void __launch_bounds__(PML_BLOCK)
joint_backtrace_kernel(PmlTree t, PmlCols c, PmlState st, int begin, int end) {
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    for (int n = begin + blockIdx.x * blockDim.x + threadIdx.x; n < end; n += gridDim.x * blockDim.x) {
        const int ps = st.js[colN + t.parent[n]];
        st.js[colN + n] = st.J[(colN + n) * c.ks + ps];
    }
}
