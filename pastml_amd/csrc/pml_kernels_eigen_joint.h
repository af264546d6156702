// Eigen models (CUSTOM_RATES, JTT), 2 <= k <= 64: the joint (max-product) bottom-up sweep, ml.py:124-148 with
// is_marginal=False, on the vector FP64 units.
//
// The joint sweep needs every entry of P(t_n) -- msg_n[i] = max_j P[i][j] v_n[j] and its arg-max -- so P cannot be
// avoided the way the sum sweeps avoid it (pml_kernels_eigen_gemm.h).  What can be chosen is where its 2 k^3 flops per
// node run.  Measured on MI355X (scripts/ub/overlap.hip): v_mfma_f64_16x16x4_f64 and v_fma_f64
// deliver the same 32 flop / clock / SIMD, and they do not overlap -- a wave's FP64 MFMAs and the vector instructions
// of every wave of the SIMD add up.  The matrix cores therefore buy nothing for FP64 but their tile shape, and the
// tile shape costs: k = 20 pads to 32 columns (62 % useful), the products of a tile land spread over lanes and
// registers, and folding them per node took 21 vector instructions per MFMA (pml_kernels_eigen_mfma.h: 0.24 of peak).
//
// Here a lane stands for (node, parent state i) -- floor(64 / k) nodes per wavefront, no padding but the idle tail
// lanes -- and holds row i of A in registers.  Per node it forms u[m] = A[i][m] exp(d_m t) once (k multiplications)
// and then, for j = 0 .. k-1 in order, P[i][j] = sum_m u[m] Ainv[m][j] as a chain of k FMAs whose second operand is
// the same for every lane: it comes from a transposed, zero-padded copy of Ainv through the scalar cache into SGPRs
// (s_load_dwordx16, one SGPR operand per v_fma_f64), so the inner loop is FMAs only.  P[i][j] v[j] is folded into the
// running maximum in the lane itself (j ascends: the first maximum stays, as numpy's argmax) -- no cross-lane
// reduction at all.  v and exp(d t) are shared between the lanes of a node through 1 KB of LDS per wavefront.
#pragma once
#include "pml_kernels_eigen_mfma.h"

typedef const __attribute__((address_space(4))) double* pml_const_f64;
#define PML_EIGJ_WAVE_LDS 128                  // doubles per wave and array: (64 / k) * KU <= 128
// rows of A in LDS: 16-byte aligned, and a stride that is not a multiple of 64 bytes so that the rows of the states
// of a node start in different banks
#define PML_EIGJ_ASTRIDE(KU) ((KU) + 2)
#define PML_EIGJ_CHUNK_LDS 160                 // doubles per wave: 64 x (branch length, mask word, node id)
// row stride (and rows) of the transposed zero-padded copy of Ainv: 32 up to 32 states, 64 beyond (pml_model_set_eigen)
#define PML_EIGJ_LD(KU) ((KU) <= 32 ? 32 : 64)
#define PML_EIGJ_LDS(KU) (PML_WAVES_PER_BLOCK * 2 * PML_EIGJ_WAVE_LDS + PML_EIGJ_LD(KU) * PML_EIGJ_ASTRIDE(KU))

template <int KU>
struct EigJWave {
    int k, ks, col, b, i, npw;
    bool lane_ok;   // the lane stands for a (node slot, state) pair
    u64 gmask;      // the lanes of the lane's node slot
    size_t colN;
    const double* sA;  // row i of A in LDS (shared by the block), zero for m >= k; rows PML_EIGJ_ASTRIDE(KU) apart
    double d_i;
    double sfc, tau, tf;
    pml_const_f64 ainvT;
    double* sE;     // per wave: exp(d_m t') of the slot's node, [npw][KU]
    double* sV;     // per wave: the vectors P is applied to, [npw][KU]
};

template <int KU>
__device__ __forceinline__ void eigj_wave_init(EigJWave<KU>& w, const PmlTree& t, const PmlCols& c, const PmlModel& m,
                                               const double* ainvT, double* smem) {
    const int k = c.k;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    w.k = k;
    w.ks = c.ks;
    w.col = blockIdx.y;
    w.colN = (size_t)w.col * t.N;
    w.npw = 64 / k;
    const int b = lane / k;
    w.lane_ok = b < w.npw;
    w.b = w.lane_ok ? b : w.npw - 1;  // idle lanes read the last slot's LDS entries and write nothing
    w.i = w.lane_ok ? lane - b * k : 0;
    w.gmask = (k >= 64 ? ~0ull : ((1ull << k) - 1ull)) << (w.b * k);
    w.sE = smem + (size_t)wave * 2 * PML_EIGJ_WAVE_LDS;
    w.sV = w.sE + PML_EIGJ_WAVE_LDS;
    for (int e = lane; e < 2 * PML_EIGJ_WAVE_LDS; e += 64) w.sE[e] = 0.0;  // the padding entries stay zero
    double* sA = smem + PML_WAVES_PER_BLOCK * 2 * PML_EIGJ_WAVE_LDS;
    const double* gA = m.A + (size_t)w.col * k * k;
    for (int e = threadIdx.x; e < k * PML_EIGJ_ASTRIDE(KU); e += blockDim.x) {
        const int r = e / PML_EIGJ_ASTRIDE(KU), q = e % PML_EIGJ_ASTRIDE(KU);
        sA[e] = q < k ? gA[r * k + q] : 0.0;
    }
    w.sA = sA + w.i * PML_EIGJ_ASTRIDE(KU);
    w.d_i = m.d[(size_t)w.col * k + w.i];
    w.sfc = m.sf[w.col];
    w.tau = m.tau[w.col];
    w.tf = m.tauf[w.col];
    w.ainvT = (pml_const_f64)(ainvT + (size_t)w.col * PML_EIGJ_LD(KU) * PML_EIGJ_LD(KU));
    __syncthreads();
}

// any() over the lanes of the lane's node slot (all of them are in the same control flow: they share the node)
template <int KU>
__device__ __forceinline__ bool eigj_node_any(const EigJWave<KU>& W, bool p) {
    return (__ballot(p) & W.gmask) != 0ull;
}

// One pass of a wave: npw nodes, lane (b, i) owns state i of the node n of slot b (act: the slot has a node).  In three
// parts, so that the level kernel can have the loads of the NEXT pass in flight while it runs the FMAs of this one
// (eigen_joint_kernel); run one after the other they are the pass as it always was -- the same operations in the same order.
struct EigJFront {   // what the front of a pass reads from memory: the node's own data and its first two children's
    u64 word;
    double dist;
    i64 cbe[2];
    double mv[2];
};

template <int KU>
__device__ __forceinline__ void eigj_front_issue(const EigJWave<KU>& W, const PmlTree& t, const PmlCols& c,
                                                 const PmlState& st, bool act, int n, int fc, int nc, EigJFront& f) {
    f.word = 0ull;
    f.dist = 0.0;
    f.cbe[0] = f.cbe[1] = 0;
    f.mv[0] = f.mv[1] = 0.0;
    if (act) {
        const size_t colN = W.colN;
        f.word = c.masks[colN + n];  // k <= 32: one word
        f.dist = t.dist[n];
        if (nc > 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ch = fc + (u < nc ? u : 0);
                f.cbe[u] = st.be[colN + ch];
                f.mv[u] = st.msg[(colN + ch) * W.ks + W.i];
            }
        }
    }
}

// the vector of the node (ml.py:126-148) from what eigj_front_issue asked for; children beyond the first two are read here
template <int KU>
__device__ __forceinline__ void eigj_front_finish(const EigJWave<KU>& W, const PmlTree& t, const PmlCols& c,
                                                  const PmlState& st, bool act, int n, int fc, int nc, const EigJFront& f,
                                                  double& v_out, double& tq_out) {
    const int k = W.k, ks = W.ks, i = W.i;
    const size_t colN = W.colN;
    const size_t row = (colN + n) * ks;
    double v = 0.0;
    double tq = 0.0;
    if (act) {
        tq = (f.dist + W.tau) * W.tf * W.sfc;
        v = ((f.word >> i) & 1ull) ? 1.0 : 0.0;
        i64 esum = 0;
        // the children two at a time: the loads of a pair go out together (a tip's exponent word is zero)
        for (int j0 = 0; j0 < nc; j0 += 2) {
            double mv[2];
            i64 cbe[2];
            if (j0 == 0) {
                mv[0] = f.mv[0];
                mv[1] = f.mv[1];
                cbe[0] = f.cbe[0];
                cbe[1] = f.cbe[1];
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ch = fc + (j0 + u < nc ? j0 + u : j0);
                    cbe[u] = st.be[colN + ch];
                    mv[u] = st.msg[(colN + ch) * ks + i];
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (j0 + u >= nc) break;
                v *= fmax(mv[u], 0.0);
                if (!eigj_node_any<KU>(W, v != 0.0)) {
                    if (i == 0)
                        atomicMin(&st.err[W.col], ((u64)(unsigned)t.post_rank[n] << 32) | (u64)(unsigned)(fc + j0 + u));
                }
                esum += cbe[u];
                // lazy_rescale of pml_device.h over the k lanes of the node: out of the band [2^-200, 2^200] is rare
                if (eigj_node_any<KU>(W, v != 0.0 && (v < 0x1p-200 || v > 0x1p+200))) {
                    double mx = 0.0;
                    for (int q = 0; q < k; ++q) mx = fmax(mx, __shfl(v, (W.b * k + q) & 63, 64));
                    if (mx > 0.0 && !isinf(mx)) {
                        const int ex = exponent_of(mx);
                        v = __builtin_ldexp(v, -ex);
                        esum += ex;
                    }
                }
            }
        }
        if (nc > 0) {  // a tip's bottom-up vector is its mask and its exponent word stays zero: nothing is stored
            st.bu[row + i] = v;
            if (i == 0) st.be[colN + n] = esum;
            if (i == k - 1)
                for (int q = k; q < ks; ++q) st.bu[row + q] = 0.0;
        }
    }
    v_out = v;
    tq_out = tq;
}

// v and exp(d t') of the node shared between its lanes, and u[m] = A[i][m] exp(d_m t')
template <int KU>
__device__ __forceinline__ void eigj_share(const EigJWave<KU>& W, double v, double tq, double (&u)[KU]) {
    const int slot = W.b * KU;
    if (W.lane_ok) {
        W.sE[slot + W.i] = exp(W.d_i * tq);
        W.sV[slot + W.i] = v;
    }
    wave_lds_sync();
#pragma unroll
    for (int mm = 0; mm < KU; ++mm) u[mm] = W.sA[mm] * W.sE[slot + mm];
}

// P[i][j] v[j], folded as it is produced, and the results of the node
template <int KU>
__device__ __forceinline__ void eigj_back(const EigJWave<KU>& W, const PmlCols& c, const PmlState& st, bool act, int n,
                                          const double (&u)[KU]) {
    const int k = W.k, ks = W.ks, i = W.i;
    const size_t colN = W.colN;
    const size_t row = (colN + n) * ks;
    const int slot = W.b * KU;
    double best = -INFINITY;
    int bj = 0;
    // the matrix is the same in every pass, and the compiler would keep all of it in (spilled) SGPRs across the loop
    // over the passes: hide the pointer from it so that the rows are streamed through the scalar cache pass by pass
    pml_const_f64 at = W.ainvT;
    asm volatile("" : "+s"(at));
    // (fully unrolled up to 32 states; beyond, KU x KU FMAs would be 32 KB of straight-line code: two columns per iteration)
    constexpr int COLUMNS_UNROLLED = KU <= 32 ? KU : 2;
#pragma unroll COLUMNS_UNROLLED
    for (int j = 0; j < KU; ++j) {
        double p = u[0] * at[j * PML_EIGJ_LD(KU)];
#pragma unroll
        for (int mm = 1; mm < KU; ++mm) p = __builtin_fma(u[mm], at[j * PML_EIGJ_LD(KU) + mm], p);
        double w = p * W.sV[slot + j];
        if (j >= KU - 3 && j >= k) w = -INFINITY;  // padding columns (k > KU - 4) must not win
        if (w > best) {  // j ascends: the first maximum stays
            best = w;
            bj = j;
        }
    }
    if (act) {
        if (c.masks_init != nullptr) {
            // altered nodes get their tables rewritten w.r.t. their initial masks (ml.py:408-428)
            const u64 mi = c.masks_init[colN + n], mc = c.masks[colN + n];
            if (mi != mc && !((mi >> bj) & 1ull)) bj = mi ? __builtin_ctzll(mi) : 0;
        }
        st.msg[row + i] = best;
        st.J[row + i] = (pml_jt)bj;
        if (i == k - 1)
            for (int q = k; q < ks; ++q) {
                st.msg[row + q] = 0.0;
                st.J[row + q] = (pml_jt)0;
            }
    }
    wave_lds_sync();  // the pass's LDS reads are done before the next pass overwrites the slots
}

template <int KU>
__device__ __forceinline__ void eigj_pass(const EigJWave<KU>& W, const PmlTree& t, const PmlCols& c,
                                          const PmlState& st, bool act, int n, int fc, int nc) {
    EigJFront f;
    eigj_front_issue<KU>(W, t, c, st, act, n, fc, nc, f);
    double v, tq;
    eigj_front_finish<KU>(W, t, c, st, act, n, fc, nc, f, v, tq);
    double u[KU];
    eigj_share<KU>(W, v, tq, u);
    eigj_back<KU>(W, c, st, act, n, u);
}

#define PML_EIGJ_ATTR __launch_bounds__(PML_BLOCK)

// The node, its first child and its number of children come from the unit descriptors of the bottom-up order
// (PmlUnit, pml_kernels_f81.h): one load instead of the chain node list -> first_child / n_children.
struct EigJUnit {
    int n, fc, nc;
};
__device__ __forceinline__ EigJUnit eigj_load_unit(const PmlTree& t, const PmlUnit* __restrict__ units, int idx) {
    const int4 h = *reinterpret_cast<const int4*>(units + idx);
    EigJUnit u;
    u.n = h.x;
    u.fc = h.y;
    u.nc = unit_nc(h.z);
    if (u.nc == 15) u.nc = t.n_children[u.n];  // the descriptor counts up to 14
    return u;
}

// one launch over the n_nodes internal nodes of one height level (their unit descriptors).  Software pipeline (round 5): half
// of a wide level's time was not arithmetic but a pass's chain of dependent loads -- descriptor -> the node's and its
// children's data -> ... (profiles/r05p_cfg3_joint_sweep.txt).  The descriptor of the pass after next and the data of the
// next pass are now in flight while the FMAs of the current pass run; PIPE = false is the plain loop (same bits).
template <int KU, bool PIPE = true>
__global__ void PML_EIGJ_ATTR eigen_joint_kernel(PmlTree t, PmlCols c, PmlModel m, PmlState st,
                                                 const double* __restrict__ ainvT,
                                                 const PmlUnit* __restrict__ units, int n_nodes) {
    __shared__ double smem[PML_EIGJ_LDS(KU)];
    EigJWave<KU> W;
    eigj_wave_init<KU>(W, t, c, m, ainvT, smem);
    const int wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * PML_WAVES_PER_BLOCK;
    const int stride = waves_total * W.npw;
    int b0 = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * W.npw;
    if (!PIPE) {
        for (; b0 < n_nodes; b0 += stride) {
            const bool act = W.lane_ok && b0 + W.b < n_nodes;
            EigJUnit u = {0, 0, 0};
            if (act) u = eigj_load_unit(t, units, b0 + W.b);
            eigj_pass<KU>(W, t, c, st, act, u.n, u.fc, u.nc);
        }
        return;
    }
    if (b0 >= n_nodes) return;
    // prologue: the first pass's descriptor and data, the second pass's descriptor
    bool act = W.lane_ok && b0 + W.b < n_nodes;
    EigJUnit cur = {0, 0, 0};
    if (act) cur = eigj_load_unit(t, units, b0 + W.b);
    bool act1 = W.lane_ok && b0 + stride + W.b < n_nodes;
    EigJUnit nxt = {0, 0, 0};
    if (act1) nxt = eigj_load_unit(t, units, b0 + stride + W.b);
    EigJFront f;
    eigj_front_issue<KU>(W, t, c, st, act, cur.n, cur.fc, cur.nc, f);
    for (; b0 < n_nodes; b0 += stride) {
        // the descriptor of the pass after next
        const bool act2 = W.lane_ok && b0 + 2 * stride + W.b < n_nodes;
        EigJUnit nn = {0, 0, 0};
        if (act2) nn = eigj_load_unit(t, units, b0 + 2 * stride + W.b);
        double v, tq;
        eigj_front_finish<KU>(W, t, c, st, act, cur.n, cur.fc, cur.nc, f, v, tq);
        double u[KU];
        eigj_share<KU>(W, v, tq, u);
        // the next pass's data go out before this pass's FMAs (its nodes are of this level: nothing this pass writes)
        EigJFront fn;
        eigj_front_issue<KU>(W, t, c, st, act1, nxt.n, nxt.fc, nxt.nc, fn);
        eigj_back<KU>(W, c, st, act, cur.n, u);
        f = fn;
        cur = nxt;
        act = act1;
        nxt = nn;
        act1 = act2;
    }
}

// the narrow end of a forest in one launch: one workgroup per column walks the levels [0, n_levels) of a level table
// (offsets into the unit descriptors)
template <int KU>
__global__ void PML_EIGJ_ATTR eigen_joint_narrow_kernel(PmlTree t, PmlCols c, PmlModel m, PmlState st,
                                                        const double* __restrict__ ainvT,
                                                        const PmlUnit* __restrict__ units,
                                                        const int* __restrict__ level_offsets, int n_levels,
                                                        const int* __restrict__ blk_start) {
    // (blk_start: the launch walks the subtree blocks of a tier of thin levels, one workgroup per (block, column);
    // block b's level table starts at level_offsets[blk_start[b]], n_levels + 1 entries)
    if (blk_start != nullptr) level_offsets += blk_start[blockIdx.x];
    __shared__ double smem[PML_EIGJ_LDS(KU)];
    EigJWave<KU> W;
    eigj_wave_init<KU>(W, t, c, m, ainvT, smem);
    const int wave = threadIdx.x >> 6;
    for (int l = 0; l < n_levels; ++l) {
        const int a = level_offsets[l], n_level = level_offsets[l + 1] - a;
        for (int b0 = wave * W.npw; b0 < n_level; b0 += PML_WAVES_PER_BLOCK * W.npw) {
            const bool act = W.lane_ok && b0 + W.b < n_level;
            EigJUnit u = {0, 0, 0};
            if (act) u = eigj_load_unit(t, units, a + b0 + W.b);
            eigj_pass<KU>(W, t, c, st, act, u.n, u.fc, u.nc);
        }
        __syncthreads();
    }
}

// Messages of the tips.  An observed tip (one allowed state s) has the unit vector as its bottom-up vector, so its
// message is column s of P -- k FMAs per lane instead of k^2, against row s of the transposed Ainv in LDS (the tips of
// a wave differ in s, so this operand cannot come through SGPRs) -- and the arg-max table has a closed form (numpy's
// first maximum of (.., 0, P[i][s], 0, ..)).  A pass that holds a tip with several allowed states runs the general pass.
template <int KU>
__global__ void PML_EIGJ_ATTR eigen_joint_tips_kernel(PmlTree t, PmlCols c, PmlModel m, PmlState st,
                                                      const double* __restrict__ ainvT,
                                                      const int* __restrict__ tip_ids, int n_tips,
                                                      const int* __restrict__ n_listed) {
    // (n_listed: the launch serves the list eigen_joint_obs_tips_kernel left for this column -- tip_ids is that list,
    // n_tips its stride -- and most of the time there is nothing on it)
    if (n_listed != nullptr) {
        tip_ids += (size_t)blockIdx.y * n_tips;
        n_tips = min(n_tips, n_listed[blockIdx.y]);
        if (n_tips <= 0) return;
    }
    constexpr int AS = PML_EIGJ_ASTRIDE(KU);
    // (beyond 32 states the transposed copy does not fit beside the rows of A -- 2 x 34 KB -- and is left out: every pass is then
    // the general one; the observed tips have been served by eigen_joint_obs_tips_kernel)
    constexpr int TS = KU <= 32 ? PML_EIGJ_LD(KU) * AS : 0;
    __shared__ double smem[PML_EIGJ_LDS(KU) + TS + PML_WAVES_PER_BLOCK * PML_EIGJ_CHUNK_LDS];
    EigJWave<KU> W;
    eigj_wave_init<KU>(W, t, c, m, ainvT, smem);
    const int k = W.k, ks = W.ks, i = W.i;
    const size_t colN = W.colN;
    double* sT = smem + PML_EIGJ_LDS(KU);  // sT[j * AS + m] = Ainv[m][j]
    if (KU <= 32) {
        const double* g = ainvT + (size_t)W.col * PML_EIGJ_LD(KU) * PML_EIGJ_LD(KU);
        for (int e = threadIdx.x; e < k * AS; e += blockDim.x) {
            const int r = e / AS, q = e % AS;
            sT[e] = q < KU ? g[r * PML_EIGJ_LD(KU) + q] : 0.0;
        }
    }
    __syncthreads();
    const u64 kbits = k >= 64 ? ~0ull : (1ull << k) - 1ull;
    const int wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * PML_WAVES_PER_BLOCK;
    const int slot = W.b * KU;
    // The tips of a wave come in chunks: lane L fetches the id, mask word and branch length of tip c0 + L -- up to 64
    // tips per round trip to memory instead of the 64 / k of one pass -- and leaves them in LDS, from where the
    // passes over the chunk read them.  A chunk holds the passes a wave has to make anyway when the tips are spread
    // over all waves (at most 64 / npw of them).
    const int lane = threadIdx.x & 63;
    const int npw = W.npw;
    const int passes_total = (n_tips + npw - 1) / npw;
    int cp = (passes_total + waves_total - 1) / waves_total;
    if (cp > 64 / npw) cp = 64 / npw;
    const int chunk = cp * npw;
    double* sTq = smem + PML_EIGJ_LDS(KU) + TS + wave * PML_EIGJ_CHUNK_LDS;
    u64* sWord = reinterpret_cast<u64*>(sTq + 64);
    int* sTip = reinterpret_cast<int*>(sWord + 64);
    for (int c0 = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * chunk; c0 < n_tips; c0 += waves_total * chunk) {
        {
            int tip_l = -1;
            u64 word_l = 0ull;
            double tq_l = 0.0;
            if (lane < chunk && c0 + lane < n_tips) {
                tip_l = tip_ids[c0 + lane];
                word_l = c.masks[colN + tip_l] & kbits;
                tq_l = (t.dist[tip_l] + W.tau) * W.tf * W.sfc;
            }
            sTip[lane] = tip_l;
            sWord[lane] = word_l;
            sTq[lane] = tq_l;
        }
        wave_lds_sync();
        for (int ps = 0; ps < cp && c0 + ps * npw < n_tips; ++ps) {
        const int sq = ps * npw + W.b;
        const int tip_c = W.lane_ok ? sTip[sq] : -1;
        const bool act = tip_c >= 0;
        const int tip = act ? tip_c : 0;
        const u64 word = act ? sWord[sq] : 0ull;
        const double tq = act ? sTq[sq] : 0.0;
        if (KU > 32 || !__all(!act || __popcll(word) == 1)) {
            eigj_pass<KU>(W, t, c, st, act, tip, 0, 0);
            continue;
        }
        const int s_own = act ? __builtin_ctzll(word) : 0;
        if (W.lane_ok) W.sE[slot + i] = exp(W.d_i * tq);
        wave_lds_sync();
        const double* row_s = sT + s_own * AS;
        double p = (W.sA[0] * W.sE[slot]) * row_s[0];
#pragma unroll
        for (int mm = 1; mm < KU; ++mm) p = __builtin_fma(W.sA[mm] * W.sE[slot + mm], row_s[mm], p);
        if (act) {
            const size_t row = (colN + tip) * ks;
            double pv = p;
            int arg;
            if (pv > 0.0) {
                arg = s_own;
            } else if (s_own != 0) {
                pv = 0.0;
                arg = 0;
            } else if (pv == 0.0 || k == 1) {
                arg = 0;
            } else {
                pv = 0.0;
                arg = 1;
            }
            if (c.masks_init != nullptr) {
                const u64 mi = c.masks_init[colN + tip], mc = c.masks[colN + tip];
                if (mi != mc && !((mi >> arg) & 1ull)) arg = mi ? __builtin_ctzll(mi) : 0;
            }
            st.msg[row + i] = pv;
            st.J[row + i] = (pml_jt)arg;
            if (i == k - 1)
                for (int q = k; q < ks; ++q) {
                    st.msg[row + q] = 0.0;
                    st.J[row + q] = (pml_jt)0;
                }
        }
        wave_lds_sync();
        }
        wave_lds_sync();  // the chunk has been consumed before the next one overwrites it
    }
}

// Observed tips only, lean (round 3).  The kernel above carries the general pass inline (its registers: 4 waves per
// SIMD) and reads three LDS operands per FMA -- A[i][m], exp(d_m t), Ainv[m][s] -- which is what bounded it (44 us for
// cfg3's 262 144 tips: the LDS pipe, not the FMAs).  Here row i of A sits in the lane's registers and the other two
// operands are folded before the sum: lane m forms w[m] = exp(d_m t) Ainv[m][s] (one exponential and one
// multiplication per lane, as before) and leaves it in LDS; the lane's P[i][s] = sum_m A[i][m] w[m] is then k FMAs
// against k / 2 16-byte LDS reads that all lanes of a tip share.  (Rounding: A (e Ainv) instead of (A e) Ainv -- the
// reference's own P(t) is numpy's A.dot(diag).dot(Ainv), neither order; ln L of cfg3 agrees to 1e-13.)  Tips that are
// not observed go on a list (per column; order irrelevant, every tip is independent) that one launch of the general
// kernel serves afterwards.
template <int KU>
__global__ void __launch_bounds__(PML_BLOCK) eigen_joint_obs_tips_kernel(PmlTree t, PmlCols c, PmlModel m, PmlState st,
                                                                          const double* __restrict__ ainvT,
                                                                          const int* __restrict__ tip_ids, int n_tips,
                                                                          int* __restrict__ rest_list,
                                                                          int* __restrict__ rest_count) {
    constexpr int AS = PML_EIGJ_ASTRIDE(KU);
    __shared__ double smem[PML_EIGJ_LD(KU) * AS + PML_WAVES_PER_BLOCK * (128 + PML_EIGJ_CHUNK_LDS)];
    const int k = c.k, ks = c.ks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y;
    const size_t colN = (size_t)col * t.N;
    const int npw = 64 / k;
    const int b0 = lane / k;
    const bool lane_ok = b0 < npw;
    const int b = lane_ok ? b0 : npw - 1;
    const int i = lane_ok ? lane - b0 * k : 0;
    double a[KU];
    {
        const double* gA = m.A + (size_t)col * k * k + (size_t)i * k;
#pragma unroll
        for (int mm = 0; mm < KU; ++mm) a[mm] = (lane_ok && mm < k) ? gA[mm] : 0.0;
    }
    const double d_i = m.d[(size_t)col * k + i];
    const double sfc = m.sf[col], tau = m.tau[col], tf = m.tauf[col];
    double* sT = smem;  // sT[j * AS + m] = Ainv[m][j]
    {
        const double* g = ainvT + (size_t)col * PML_EIGJ_LD(KU) * PML_EIGJ_LD(KU);
        for (int e = threadIdx.x; e < k * AS; e += blockDim.x) {
            const int r = e / AS, q = e % AS;
            sT[e] = q < KU ? g[r * PML_EIGJ_LD(KU) + q] : 0.0;
        }
    }
    double* sW = smem + PML_EIGJ_LD(KU) * AS + wave * (128 + PML_EIGJ_CHUNK_LDS);
    for (int e = lane; e < 128; e += 64) sW[e] = 0.0;  // the padding entries (m >= k) stay zero
    double* sTq = sW + 128;
    u64* sWord = reinterpret_cast<u64*>(sTq + 64);
    int* sTip = reinterpret_cast<int*>(sWord + 64);
    __syncthreads();
    const u64 kbits = k >= 64 ? ~0ull : (1ull << k) - 1ull;
    const int waves_total = gridDim.x * PML_WAVES_PER_BLOCK;
    const int slot = b * KU;
    const int passes_total = (n_tips + npw - 1) / npw;
    int cp = (passes_total + waves_total - 1) / waves_total;
    if (cp > 64 / npw) cp = 64 / npw;
    const int chunk = cp * npw;
    for (int c0 = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * chunk; c0 < n_tips; c0 += waves_total * chunk) {
        {
            int tip_l = -1;
            u64 word_l = 0ull;
            double tq_l = 0.0;
            if (lane < chunk && c0 + lane < n_tips) {
                tip_l = tip_ids[c0 + lane];
                word_l = c.masks[colN + tip_l] & kbits;
                tq_l = (t.dist[tip_l] + tau) * tf * sfc;
                if (__popcll(word_l) != 1) {  // not observed: for the general kernel
                    rest_list[(size_t)col * n_tips + atomicAdd(&rest_count[col], 1)] = tip_l;
                    tip_l = -1;
                }
            }
            sTip[lane] = tip_l;
            sWord[lane] = word_l;
            sTq[lane] = tq_l;
        }
        wave_lds_sync();
        for (int ps = 0; ps < cp && c0 + ps * npw < n_tips; ++ps) {
            const int sq = ps * npw + b;
            const int tip_c = lane_ok ? sTip[sq] : -1;
            const bool act = tip_c >= 0;
            const int tip = act ? tip_c : 0;
            const int s_own = act ? __builtin_ctzll(sWord[sq]) : 0;
            const double tq = act ? sTq[sq] : 0.0;
            if (lane_ok) sW[slot + i] = exp(d_i * tq) * sT[s_own * AS + i];
            wave_lds_sync();
            const double* w = sW + slot;
            double p = a[0] * w[0];
#pragma unroll
            for (int mm = 1; mm < KU; ++mm) p = __builtin_fma(a[mm], w[mm], p);
            if (act) {
                // numpy's first maximum of (.., 0, P[i][s], 0, ..) with the reference's clamp of negative entries
                const size_t row = (colN + tip) * ks;
                double pv = p;
                int arg;
                if (pv > 0.0) {
                    arg = s_own;
                } else if (s_own != 0) {
                    pv = 0.0;
                    arg = 0;
                } else if (pv == 0.0 || k == 1) {
                    arg = 0;
                } else {
                    pv = 0.0;
                    arg = 1;
                }
                if (c.masks_init != nullptr) {
                    const u64 mi = c.masks_init[colN + tip], mc = c.masks[colN + tip];
                    if (mi != mc && !((mi >> arg) & 1ull)) arg = mi ? __builtin_ctzll(mi) : 0;
                }
                st.msg[row + i] = pv;
                st.J[row + i] = (pml_jt)arg;
                if (i == k - 1)
                    for (int q = k; q < ks; ++q) {
                        st.msg[row + q] = 0.0;
                        st.J[row + q] = (pml_jt)0;
                    }
            }
            wave_lds_sync();
        }
        wave_lds_sync();  // the chunk has been consumed before the next one overwrites it
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The P(t) batch of the eigen models on the vector units (round 5; 2 <= k <= 32): P(t) = A diag(exp(d t')) Ainv of every
// branch (pastml/models/generator.py:54-65, CustomRatesModel.py:70-79), stored transposed, Pt[b][j][i], ks doubles per row.
//
// The matrix-core kernel (pij_eigen_mfma_kernel) pads k = 20 to 32 columns -- 1.6 x the flops -- and FP64 matrix and vector
// instructions share one issue resource at the same peak, so the tile shape is all the matrix cores give; without its stores
// that kernel takes 0.30 ms for 524 287 branches where the 2 k^3 flops are 0.107 ms of the peak (profiles/r05g_pij_kernel.txt).
// Here a lane stands for one output ROW (b, j): the rows of a column are one contiguous run of memory, a wave takes 64
// consecutive ones.  The lane forms w[m] = Ainv[m][j] exp(d_m t_b) (Ainv's column j from LDS, the exponentials of the pass's
// few branches shared through LDS) and then out[i] = sum_m w[m] A[i][m] as k independent chains of k FMAs whose second operand
// is the same for every lane: row m of a transposed zero-padded copy of A streams through the scalar cache into SGPRs, as
// Ainv does in eigen_joint_kernel.  No padding (all 64 lanes work, exact 2 k^3 flops for k a multiple of 4), no fold across
// lanes.  The wave's 64 rows are 64 ks contiguous doubles: staged in LDS half a wave at a time and written in address order,
// 16 bytes per lane.
// ---------------------------------------------------------------------------------------------------------------------
#define PML_PIJV_EXP_LDS 160   // doubles: exp(d_m t) of the branches one pass touches, (64 / k + 2) * KU <= 136

template <int KU>
__global__ void __launch_bounds__(PML_BLOCK)
pij_eigen_valu_kernel(PmlTree t, PmlCols c, PmlModel m, const double* __restrict__ ainvT, const double* __restrict__ aT,
                      double* __restrict__ P) {
    constexpr int AS = PML_EIGJ_ASTRIDE(KU);
    __shared__ double smem[PML_EIGJ_STRIDE * AS + PML_WAVES_PER_BLOCK * (PML_PIJV_EXP_LDS + 32 * KU)];
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    const int k = c.k, ks = c.ks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y;
    double* sT = smem;   // sT[j * AS + m] = Ainv[m][j]
    {
        const double* g = ainvT + (size_t)col * PML_EIGJ_STRIDE * PML_EIGJ_STRIDE;
        for (int e = threadIdx.x; e < k * AS; e += blockDim.x) {
            const int r = e / AS, q = e % AS;
            sT[e] = q < KU ? g[r * PML_EIGJ_STRIDE + q] : 0.0;
        }
    }
    double* sE = smem + PML_EIGJ_STRIDE * AS + wave * (PML_PIJV_EXP_LDS + 32 * KU);
    double* sO = sE + PML_PIJV_EXP_LDS;   // 32 staged rows of ks doubles
    __syncthreads();
    const double* gd = m.d + (size_t)col * k;
    const double sfc = m.sf[col], tau = m.tau[col], tf = m.tauf[col];
    const pml_const_f64 at_col = (pml_const_f64)(aT + (size_t)col * PML_EIGJ_STRIDE * PML_EIGJ_STRIDE);
    const long long rows_total = (long long)t.N * k;
    double* const Pcol = P + (size_t)col * t.N * (size_t)k * ks;
    const long long waves_total = (long long)gridDim.x * PML_WAVES_PER_BLOCK;
    for (long long r0 = ((long long)blockIdx.x * PML_WAVES_PER_BLOCK + wave) * 64; r0 < rows_total; r0 += waves_total * 64) {
        const int b_lo = (int)(r0 / k);
        const long long r_last = (r0 + 63 < rows_total ? r0 + 63 : rows_total - 1);
        const int nbr = (int)(r_last / k) - b_lo + 1;
        // exp(d_m t') of the branches of this pass
        for (int e = lane; e < nbr * KU; e += 64) {
            const int bl = e / KU, mm = e % KU;
            double v = 0.0;
            if (mm < k) v = exp(gd[mm] * ((t.dist[b_lo + bl] + tau) * tf * sfc));
            sE[e] = v;
        }
        wave_lds_sync();
        const long long r = r0 + lane;
        const bool act = r < rows_total;
        const int b = act ? (int)(r / k) : b_lo;
        const int j = act ? (int)(r - (long long)b * k) : 0;
        double w[KU];
        {
            const double* pt = sT + j * AS;
            const double* pe = sE + (b - b_lo) * KU;
#pragma unroll
            for (int mm = 0; mm < KU; ++mm) w[mm] = pt[mm] * pe[mm];
        }
        // out[i] = sum_m w[m] A[i][m]: rows of A^T through the scalar cache (hidden from the compiler, which would otherwise
        // keep the whole matrix in spilled SGPRs across the passes)
        pml_const_f64 at = at_col;
        asm volatile("" : "+s"(at));
        double out[KU];
#pragma unroll
        for (int i = 0; i < KU; ++i) out[i] = w[0] * at[i];
#pragma unroll
        for (int mm = 1; mm < KU; ++mm)
#pragma unroll
            for (int i = 0; i < KU; ++i) out[i] = __builtin_fma(w[mm], at[mm * PML_EIGJ_STRIDE + i], out[i]);
        // the wave's rows are contiguous in memory: half a wave's rows through LDS at a time, written in address order
        double* const gout = Pcol + (size_t)r0 * ks;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((lane >> 5) == half) {
                double* srow = sO + (lane & 31) * ks;
#pragma unroll
                for (int i = 0; i < KU; ++i)
                    if (i < ks) srow[i] = out[i];   // (columns k .. ks - 1: exact zeros, the padding rows of A^T are zero)
            }
            wave_lds_sync();
            const long long first = r0 + 32 * half;
            const int hrows = first >= rows_total ? 0 : (int)(rows_total - first < 32 ? rows_total - first : 32);
            const int pairs = hrows * ks / 2;   // (32 ks is even; a short last half of odd length: the scalar tail below)
            double* const hout = gout + (size_t)32 * half * ks;
            for (int e = lane; e < pairs; e += 64) {
                const dbl2 v = *reinterpret_cast<const dbl2*>(sO + 2 * e);
                __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(hout + 2 * e));
            }
            if (((hrows * ks) & 1) && lane == 0) hout[hrows * ks - 1] = sO[hrows * ks - 1];
            wave_lds_sync();
        }
    }
}
