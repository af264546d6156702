// Scenario sampling of marginal_counts (pastml/ml.py:753-862) on the device.
//
// The reference walks the tree in level order carrying, per node, how many of the n_repetitions scenarios put the node
// in each state.  The root's counts are n_repetitions draws from its marginal posterior; for a child n of parent p and
// every parent state a with count c_a > 0 it draws c_a child states from
//     prob(b | a)  ~  BU_n[b] * pi_b * mask_n[b] * P_n[b][a]                           (ml.py:819-824)
// adds the draws to result[a][.] and to the child's counts, and at the end of a parent subtracts
// min(c_a, number of a -> a draws over its children) from result[a][a] (ml.py:857-858).  Drawing c_a categorical
// values is what np.random.choice(size = c_a, p = ...) does there, so the estimator is the same; only the random
// numbers differ (Philox-4x32-10 keyed by seed, child, parent state and draw index: results do not depend on the
// launch geometry).
//
// One wavefront per parent, one launch per depth level (the parents of a depth are a contiguous piece of the
// td_parents list).  Per (child, a): the k weights are formed by the lanes, an inclusive scan gives the cumulative
// table in LDS, the lanes draw and bisect in parallel; counts are integer LDS atomics, flushed to the global k x k table
// (64-bit integer atomics) per parent, so sums are exact and independent of the order of the additions.
// Nodes altered by the zero-branch handling (ml.py:352-387) follow other rules there: the host keeps that case.
#pragma once
#include "pml_kernels_pij.h"

__device__ __forceinline__ void philox4x32_10(unsigned (&ctr)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * ctr[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * ctr[2];
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
        const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        const unsigned n0 = hi1 ^ ctr[1] ^ k0, n1 = lo1, n2 = hi0 ^ ctr[3] ^ k1, n3 = lo0;
        ctr[0] = n0;
        ctr[1] = n1;
        ctr[2] = n2;
        ctr[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// uniform in [0, 1) with 53 random bits, a pure function of (seed, node, parent state, draw index)
__device__ __forceinline__ double counts_uniform(u64 seed, unsigned node, unsigned state, unsigned draw) {
    unsigned ctr[4] = {draw, state, node, 0x51ed270bu};
    philox4x32_10(ctr, (unsigned)seed, (unsigned)(seed >> 32));
    const u64 bits = ((u64)ctr[0] << 32) | ctr[1];
    return (double)(bits >> 11) * 0x1p-53;
}

#define PML_COUNTS_MAX_K 256

// inclusive scan over the 64 lanes of a wavefront (Hillis-Steele on shuffles: 6 steps)
__device__ __forceinline__ double wave_inclusive_scan(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(v, o, 64);
        if (lane >= o) v += up;
    }
    return v;
}

// counts of the roots: n_rep draws from the posterior of each root (ml.py:786-793)
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(64)
counts_roots_kernel(PmlTree t, PmlCols c, PmlState st, int col, int n_rep, u64 seed, int* __restrict__ counts,
                    const int* __restrict__ api_id) {
    __shared__ double cdf[PML_COUNTS_MAX_K];
    __shared__ int hist[PML_COUNTS_MAX_K];
    const int lane = threadIdx.x;
    const int k = c.k, ks = c.ks;
    const size_t colN = (size_t)col * t.N;
    for (int r = blockIdx.x; r < t.n_roots; r += gridDim.x) {
        double run = 0.0;
        for (int b0 = 0; b0 < k; b0 += 64) {
            const int b = b0 + lane;
            const double w = b < k ? st.post[(colN + r) * ks + b] : 0.0;
            const double inc = wave_inclusive_scan(w, lane) + run;
            if (b < k) {
                cdf[b] = inc;
                hist[b] = 0;
            }
            run = __shfl(inc, 63, 64);
        }
        __syncthreads();
        const double W = run;
        for (int i = lane; i < n_rep; i += 64) {
            const double u = counts_uniform(seed, (unsigned)(api_id ? api_id[r] : r), 0xffffffffu, (unsigned)i) * W;
            int lo = 0, hi = k - 1;  // first b with cdf[b] > u (the last state if rounding put u at W)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] > u) hi = mid; else lo = mid + 1;
            }
            atomicAdd(&hist[lo], 1);
        }
        __syncthreads();
        for (int b = lane; b < k; b += 64) counts[(size_t)r * k + b] = hist[b];
        __syncthreads();
    }
}
#endif

// one depth level: parents[0 .. n_parents), one wavefront (= one 64-thread block) per parent
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only: the other translation units leave it out)
PML_GLOBAL void __launch_bounds__(64)
counts_level_kernel(PmlTree t, PmlCols c, PmlState st, PmlModel m, const double* __restrict__ P, int col, int n_rep,
                    u64 seed, const int* __restrict__ parents, int n_parents, int* __restrict__ counts,
                    long long* __restrict__ result, const int* __restrict__ api_id,
                    const unsigned char* __restrict__ altered, int* __restrict__ same_out) {
    // altered (or null): the nodes whose masks the zero-branch handling changed (ml.py:352-387).  A (parent, child) pair with an
    // altered end adds nothing to the result here and a parent with such a pair keeps its diagonal correction: the reference gives
    // those pairs fractional counts (ml.py:806-812, 840-853), which the caller forms from the state counts of the nodes (`counts`)
    // and this parent's same-state draws over its other children (same_out) -- pml_marginal_counts_altered.
    __shared__ double cdf[PML_COUNTS_MAX_K];
    __shared__ double base[PML_COUNTS_MAX_K];  // BU_n[b] pi_b mask_n[b]
    __shared__ int pc[PML_COUNTS_MAX_K];       // counts of the parent
    __shared__ int cc[PML_COUNTS_MAX_K];       // counts of the child
    __shared__ int same[PML_COUNTS_MAX_K];     // a -> a draws over the children
    __shared__ int hist[PML_COUNTS_MAX_K];     // draws of one (child, a)
    const int lane = threadIdx.x;
    const int k = c.k, ks = c.ks;
    const size_t colN = (size_t)col * t.N;
    const double* pi = c.pi + (size_t)col * ks;
    for (int q = blockIdx.x; q < n_parents; q += gridDim.x) {
        const int p = parents[q];
        for (int a = lane; a < k; a += 64) {
            pc[a] = counts[(size_t)p * k + a];
            same[a] = 0;
        }
        __syncthreads();
        const int fc = t.first_child[p], nc = t.n_children[p];
        bool dirty = altered != nullptr && altered[p] != 0;
        for (int j = 0; j < nc; ++j) {
            const int n = fc + j;
            const bool upd = altered == nullptr || !(altered[p] | altered[n]);   // (block-uniform)
            dirty |= !upd;
            const unsigned key = (unsigned)(api_id ? api_id[n] : n);   // the node as the caller numbers it
            const bool tip = t.n_children[n] == 0;
            for (int b = lane; b < k; b += 64) {
                const bool allowed = (c.masks[(colN + n) * c.W + (b >> 6)] >> (b & 63)) & 1ull;
                const double bu = tip ? 1.0 : st.bu[(colN + n) * ks + b];
                base[b] = allowed ? bu * pi[b] : 0.0;
                cc[b] = 0;
            }
            __syncthreads();
            double e = 0.0;
            if (m.kind == PML_MODEL_F81) e = st.E[colN + n];
            const double* Pt = P != nullptr ? P + (colN + n) * (size_t)k * ks : nullptr;
            for (int a = 0; a < k; ++a) {
                const int ca = pc[a];
                if (ca == 0) continue;  // block-uniform
                double run = 0.0;
                for (int b0 = 0; b0 < k; b0 += 64) {
                    const int b = b0 + lane;
                    double w = 0.0;
                    if (b < k) {
                        // P_n[b][a]: F81 closed form (F81Model.py:42-46), else row a of the stored P^T
                        const double pba = Pt != nullptr ? Pt[(size_t)a * ks + b] : ((1.0 - e) * pi[a] + (a == b ? e : 0.0));
                        w = base[b] * fmax(pba, 0.0);
                    }
                    const double inc = wave_inclusive_scan(w, lane) + run;
                    if (b < k) {
                        cdf[b] = inc;
                        hist[b] = 0;
                    }
                    run = __shfl(inc, 63, 64);
                }
                __syncthreads();
                const double W = run;
                if (W > 0.0) {
                    for (int i = lane; i < ca; i += 64) {
                        const double u = counts_uniform(seed, key, (unsigned)a, (unsigned)i) * W;
                        int lo = 0, hi = k - 1;
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (cdf[mid] > u) hi = mid; else lo = mid + 1;
                        }
                        atomicAdd(&hist[lo], 1);
                    }
                }
                __syncthreads();
                for (int b = lane; b < k; b += 64) {
                    const int h = hist[b];
                    if (h) {
                        cc[b] += h;
                        if (upd) {
                            atomicAdd((unsigned long long*)&result[(size_t)a * k + b], (unsigned long long)h);
                            if (b == a) same[a] += h;
                        }
                    }
                }
                __syncthreads();
            }
            for (int b = lane; b < k; b += 64) counts[(size_t)n * k + b] = cc[b];
            __syncthreads();
        }
        // result[a][a] -= min(c_a, same-state draws)   (ml.py:857-858); a parent with an altered end of a pair: the caller's
        for (int a = lane; a < k; a += 64) {
            if (dirty) {
                same_out[(size_t)p * k + a] = same[a];
            } else {
                const int d = min(pc[a], same[a]);
                if (d) atomicAdd((unsigned long long*)&result[(size_t)a * k + a], (unsigned long long)(-(long long)d));
            }
        }
        __syncthreads();
    }
}
#endif
