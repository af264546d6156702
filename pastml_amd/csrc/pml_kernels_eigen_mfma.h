// Eigen models (CUSTOM_RATES, JTT), 16 <= k <= 32: sweeps that build P(t) on the FP64 matrix cores and consume it in
// registers -- P never reaches HBM (the stand-alone batch of pml_kernels_pij.h writes 8 k^2 bytes per branch and the
// sweeps of pml_kernels_matrix.h read them back: 3.2 KB per branch at k = 20 against 160 B for a state vector).
//
// Every sweep of pastml/ml.py applies P(t_n) of the branch above a node n to one vector of that node:
//   bottom-up  (ml.py:124-148)   msg_n[i] = sum_j / max_j  P[i][j] v_n[j]      v_n = BU vector of n
//   top-down   (ml.py:273-290)   td_n[i]  = sum_j          P[i][j] x_n[j]      x_n = TD_p o BU_p / msg_n
// so a node is visited once per sweep, in the launch of its own level: it first forms its vector from what earlier
// launches left (the messages of its children / the vectors of its parent), then pushes it through its branch.  The
// messages msg_n are kept (8k bytes per node): the parent multiplies them, and the top-down sweep divides by them
// instead of recomputing P v (one application of P per node and sweep instead of the reference's two in top-down).
//
// One wavefront takes NB nodes at a time.  Their P^T matrices are the rows of a tall GEMM, as in pij_eigen_mfma_kernel:
//   Pt_b[j][i] = sum_m (Ainv[m][j] exp(d_m t_b)) A[i][m],   rows (b, j) with j padded to KP = 4 ceil(k / 4),
// NB chosen so that NB * KP is a multiple of the 16 rows of a v_mfma_f64_16x16x4_f64 tile.  In the 16x16 result tile
// lane (lo, hi) holds column i = lo (+16 nt) of rows hi + 4 reg, so a group of four rows (one reg) lies inside one
// node (KP is a multiple of 4): which node a (tile, reg) slot belongs to is known at compile time.  Each lane
// multiplies its entries by v[j], folds them per node (sum, or first maximum with its index as numpy's argmax,
// ml.py:134) and the four row groups of a column are combined across lanes.  The tile entries are produced by the same
// MFMA sequence as the stand-alone batch, so P has the same bits as what pml_pij_batch returns.
//
// Lane (lo, hi) owns node hi of the pass and its states lo, 16 + lo in all vector steps.
#pragma once
#include "pml_kernels_pij.h"

template <int KS>
struct EigShape {
    static constexpr int KP = 4 * KS;                                           // padded states per node
    static constexpr int NB = (KS % 4 == 0) ? 1 : ((KS % 2 == 0) ? 2 : 4);      // nodes per wave pass
    static constexpr int ROWS = NB * KP;
    static constexpr int TILES = ROWS / 16;
    static constexpr int WAVE_LDS = 2 * NB * KP + 4;                            // doubles: exp table, vectors, t'
    // tips kernel: exp table of 16 tips, vectors of a general pass, t', and 16 states + 16 node ids (as ints)
    static constexpr int WAVE_LDS_TIPS = 16 * KP + NB * KP + 16 + 16;
};


// Per-wave state of the fused kernels: LDS slots, the B fragments of A^T, the column's scalars.
template <int NT, int KS>
struct EigWave {
    int k, ks, col, lo, hi;
    size_t colN;
    double* sB;  // Ainv padded: [KP][k] (rows >= k are zero), shared by the block
    double* sE;  // per wave: exp(d_m t'_q)
    double* sV;  // per wave: the vectors P is applied to [NB][KP]
    double* sT;  // per wave: t' of the branches of the pass
    const double* gd;
    double bfrag[NT][KS];  // bfrag[nt][s] = A[16 nt + lo][4 s + hi]
    double pi_r[NT];
    double sfc, tau, tf;
};

// wave_lds: doubles of LDS per wave; sE takes n_exp_rows * KP of them, then sV (NB * KP), then sT
template <int NT, int KS>
__device__ __forceinline__ void eig_wave_init(EigWave<NT, KS>& w, const PmlCols& c, const PmlModel& m, double* smem,
                                              int wave_lds, int n_exp_rows) {
    typedef EigShape<KS> S;
    constexpr int KP = S::KP, NB = S::NB;
    const int k = c.k;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    w.k = k;
    w.ks = c.ks;  // == KP (checked on the host)
    w.col = blockIdx.y;
    w.colN = 0;
    w.lo = lane & 15;
    w.hi = lane >> 4;
    w.sB = smem;
    w.sE = smem + KP * k + wave * wave_lds;
    w.sV = w.sE + n_exp_rows * KP;
    w.sT = w.sV + NB * KP;
    const double* gA = m.A + (size_t)w.col * k * k;
    const double* gB = m.Ainv + (size_t)w.col * k * k;
    w.gd = m.d + (size_t)w.col * k;
    for (int e = threadIdx.x; e < KP * k; e += blockDim.x) w.sB[e] = (e / k < k) ? gB[e] : 0.0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int i = 16 * nt + w.lo, mm = 4 * s + w.hi;
            w.bfrag[nt][s] = (i < k && mm < k) ? gA[i * k + mm] : 0.0;
        }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        w.pi_r[nt] = (16 * nt + w.lo < k) ? c.pi[(size_t)w.col * c.ks + 16 * nt + w.lo] : 0.0;
    w.sfc = m.sf[w.col];
    w.tau = m.tau[w.col];
    w.tf = m.tauf[w.col];
    __syncthreads();
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// combines two partial results of a node: sum, or the larger value with the smaller index among equal ones
template <int MODE>
__device__ __forceinline__ void eig_combine(double& v, int& j, double ov, int oj) {
    if (MODE == PML_EIG_BU_JOINT) {
        if (ov > v || (ov == v && oj < j)) {
            v = ov;
            j = oj;
        }
    } else {
        v += ov;
    }
}

// One pass of a wave: NB nodes, lane (lo, hi) owns node n of row hi (act: the row has a node).
template <int NT, int KS, int MODE>
__device__ __forceinline__ void eig_pass(const EigWave<NT, KS>& W, const PmlTree& t, const PmlCols& c,
                                         const PmlState& st, bool act, int n, int tips) {
    typedef EigShape<KS> S;
    constexpr int KP = S::KP, NB = S::NB, TILES = S::TILES;
    const int k = W.k, ks = W.ks, col = W.col, lo = W.lo, hi = W.hi;
    const size_t colN = W.colN;
    double* sB = W.sB;
    double* sE = W.sE;
    double* sV = W.sV;
    double* sT = W.sT;
    const double* gd = W.gd;
    const double sfc = W.sfc, tau = W.tau, tf = W.tf;
    const int lane = threadIdx.x & 63;
    {
        const size_t row = (colN + n) * ks;
        // ------------------------------------------------------------------ the vector of the node
        double v[NT];   // what P is applied to
        double vc[NT];  // top-down: BU vector of the node; mb: its mask
        double mb[NT];
        i64 esum = 0;   // bottom-up: exponent of v; top-down: exponent of x
        i64 bec = 0;
        bool tipc = false;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) v[nt] = vc[nt] = mb[nt] = 0.0;
        // everything that only needs the node id goes out first, together: one wait instead of one per array
        u64 word = 0ull;
        int fc = 0, nc = 0, p = 0;
        double dist_n = 0.0;
        i64 be_n = 0, te_p = 0, be_p = 0;
        double own_bu[NT], own_msg[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) own_bu[nt] = own_msg[nt] = 0.0;
        if (act) {
            word = c.masks[colN + n];  // k <= 32: one word
            nc = t.n_children[n];
            dist_n = t.dist[n];
            if (MODE != PML_EIG_TD) {
                fc = t.first_child[n];
            } else {
                p = t.parent[n];
                be_n = st.be[colN + n];  // a tip's exponent word is 0 (never written by these sweeps)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int i = 16 * nt + lo;
                    if (i < ks) {
                        own_bu[nt] = st.bu[row + i];  // a tip's row is allocated but never written: not used below
                        own_msg[nt] = st.msg[row + i];
                    }
                }
            }
        }
        if (act) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int i = 16 * nt + lo;
                mb[nt] = (i < k && ((word >> i) & 1ull)) ? 1.0 : 0.0;
            }
            if (MODE != PML_EIG_TD) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) v[nt] = mb[nt];
                if (!tips) {
                    // mask o prod of the children's messages (ml.py:126-148), zero check and rescaling per child
                    // The children are taken two at a time: the loads of a pair (messages, exponents) go out
                    // together, so a binary node waits for memory once, not once per child and array.  A tip's
                    // exponent word is zero (set when the column arrays are allocated, never written by these
                    // sweeps), so it is read without asking whether the child is a tip.
                    for (int j0 = 0; j0 < nc; j0 += 2) {
                        double mv[2][NT];
                        i64 cbe[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int ch = fc + (j0 + u < nc ? j0 + u : j0);
                            cbe[u] = st.be[colN + ch];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                const int i = 16 * nt + lo;
                                mv[u][nt] = i < ks ? st.msg[(colN + ch) * ks + i] : 0.0;
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            if (j0 + u >= nc) break;
                            const int ch = fc + j0 + u;
                            bool nz = false;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                v[nt] *= fmax(mv[u][nt], 0.0);
                                nz |= v[nt] != 0.0;
                            }
                            if (!group_any<16>(nz)) {
                                if (lo == 0)
                                    atomicMin(&st.err[col], ((u64)(unsigned)t.post_rank[n] << 32) | (u64)(unsigned)ch);
                            }
                            esum += cbe[u];
                            esum += lazy_rescale<16, NT>(v);
                        }
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int i = 16 * nt + lo;
                        if (i < ks) st.bu[row + i] = v[nt];
                    }
                    if (lo == 0) st.be[colN + n] = esum;
                }
            } else {
                // x = TD_p o BU_p / msg_n (ml.py:279-283); the message is what the bottom-up sweep left
                tipc = nc == 0;
                bec = tipc ? 0 : be_n;
                te_p = st.te[colN + p];
                be_p = st.be[colN + p];
                double prod[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int i = 16 * nt + lo;
                    prod[nt] = i < ks ? st.td[(colN + p) * ks + i] * st.bu[(colN + p) * ks + i] : 0.0;
                }
                const i64 pe = te_p + be_p;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    vc[nt] = tipc ? mb[nt] : own_bu[nt];
                    const double mc = own_msg[nt];
                    v[nt] = prod[nt] / (mc > 0.0 ? mc : 1.0);
                }
                esum = pe - bec;
                esum += lazy_rescale<16, NT>(v);
            }
        }
        // ------------------------------------------------------------------ stage v, t' and exp(d t') in LDS
        if (hi < NB) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int i = 16 * nt + lo;
                if (i < KP) sV[hi * KP + i] = v[nt];
            }
            if (lo == 0) sT[hi] = act ? (dist_n + tau) * tf * sfc : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e = lane; e < NB * KP; e += 64) {
            const int q = e / KP, mm = e % KP;
            sE[e] = mm < k ? exp(gd[mm] * sT[q]) : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ------------------------------------------------------------------ P^T tiles, folded per node
        double acc_v[NB][NT];
        int acc_j[NB][NT];
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc_v[q][nt] = MODE == PML_EIG_BU_JOINT ? -INFINITY : 0.0;
                acc_j[q][nt] = 0x7fffffff;
            }
#pragma unroll
        for (int tile = 0; tile < TILES; ++tile) {
            const int ra = 16 * tile + lo;  // A operand row of this lane: (node qa, state ja)
            const int qa = ra / KP, ja = ra % KP;
            pml_v4f64 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (pml_v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int mm = 4 * s + hi;
                const double a = ja < k ? sB[mm * k + ja] * sE[qa * KP + mm] : 0.0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, W.bfrag[nt][s], acc[nt], 0, 0, 0);
            }
            // D: row = 16 tile + hi + 4 reg, col = lo (+16 nt); the four rows of a reg belong to one node
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                constexpr int dummy = 0;
                (void)dummy;
                const int g0 = 16 * tile + 4 * reg;
                const int q = g0 / KP;          // compile-time after unrolling
                const int j = g0 % KP + hi;
                if (j < k) {
                    const double vj = sV[q * KP + j];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const double w = acc[nt][reg] * vj;
                        if (MODE == PML_EIG_BU_JOINT) {
                            if (w > acc_v[q][nt]) {  // j ascends within a lane: the first maximum stays
                                acc_v[q][nt] = w;
                                acc_j[q][nt] = j;
                            }
                        } else {
                            acc_v[q][nt] += w;
                        }
                    }
                }
            }
        }
        // The four row groups of a column (lanes lo, lo + 16, lo + 32, lo + 48) hold partial results for all NB
        // nodes; lane (lo, hi) needs the total of node hi only.  Reduce-scatter: across the halves (xor 32) a lane
        // hands over the partials of the nodes the other half owns, across the rows (xor 16) likewise -- 3 NT
        // exchanges for NB = 4 instead of the 8 NT of reducing everything everywhere.
        double r[NT];
        int rj[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double kv[2];
            int kj[2];
            if (NB == 4) {
                const bool upper = (hi & 2) != 0;  // owns nodes 2, 3
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double mine = upper ? acc_v[2 + j][nt] : acc_v[j][nt];
                    const double give = upper ? acc_v[j][nt] : acc_v[2 + j][nt];
                    const int minej = upper ? acc_j[2 + j][nt] : acc_j[j][nt];
                    const int givej = upper ? acc_j[j][nt] : acc_j[2 + j][nt];
                    kv[j] = mine;
                    kj[j] = minej;
                    eig_combine<MODE>(kv[j], kj[j], __shfl_xor(give, 32, 64),
                                      MODE == PML_EIG_BU_JOINT ? __shfl_xor(givej, 32, 64) : 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = j < NB ? j : 0;
                    kv[j] = acc_v[q][nt];
                    kj[j] = acc_j[q][nt];
                    eig_combine<MODE>(kv[j], kj[j], __shfl_xor(acc_v[q][nt], 32, 64),
                                      MODE == PML_EIG_BU_JOINT ? __shfl_xor(acc_j[q][nt], 32, 64) : 0);
                }
            }
            if (NB >= 2) {
                const bool odd = (hi & 1) != 0;  // owns the second of the two nodes left
                r[nt] = odd ? kv[1] : kv[0];
                rj[nt] = odd ? kj[1] : kj[0];
                const double give = odd ? kv[0] : kv[1];
                const int givej = odd ? kj[0] : kj[1];
                eig_combine<MODE>(r[nt], rj[nt], __shfl_xor(give, 16, 64),
                                  MODE == PML_EIG_BU_JOINT ? __shfl_xor(givej, 16, 64) : 0);
            } else {
                r[nt] = kv[0];
                rj[nt] = kj[0];
                eig_combine<MODE>(r[nt], rj[nt], __shfl_xor(kv[0], 16, 64),
                                  MODE == PML_EIG_BU_JOINT ? __shfl_xor(kj[0], 16, 64) : 0);
            }
        }
        // ------------------------------------------------------------------ results of the node
        if (act) {
            if (MODE != PML_EIG_TD) {
                if (MODE == PML_EIG_BU_JOINT && c.masks_init != nullptr) {
                    // altered nodes get their tables rewritten w.r.t. their initial masks (ml.py:408-428)
                    const u64 mi = c.masks_init[colN + n], mc = c.masks[colN + n];
                    if (mi != mc) {
                        const int fa = mi ? __builtin_ctzll(mi) : 0;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            if (16 * nt + lo < k && !((mi >> rj[nt]) & 1ull)) rj[nt] = fa;
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int i = 16 * nt + lo;
                    if (i < ks) {
                        st.msg[row + i] = i < k ? r[nt] : 0.0;
                        if (MODE == PML_EIG_BU_JOINT) st.J[row + i] = (pml_jt)(i < k ? rj[nt] : 0);
                    }
                }
            } else {
                // marginal likelihoods pi o mask o BU o TD (ml.py:456-460) and posteriors (ml.py:498-500)
                double tdc[NT], lh[NT];
                double lhs = 0.0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    tdc[nt] = fmax(r[nt], 0.0);
                    lh[nt] = vc[nt] * tdc[nt] * (W.pi_r[nt] * mb[nt]);
                    lhs += lh[nt];
                }
                lhs = group_sum<16>(lhs);
                const int lex = (lhs > 0.0 && !isinf(lhs)) ? exponent_of(lhs) : 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int i = 16 * nt + lo;
                    if (i < ks) {
                        if (!tipc) st.td[row + i] = i < k ? tdc[nt] : 0.0;
                        st.post[row + i] = lh[nt] / lhs;
                    }
                }
                if (lo == 0) {
                    if (!tipc) st.te[colN + n] = esum;
                    st.lhsum[colN + n] = __builtin_ldexp(lhs, -lex);
                    st.lhe[colN + n] = esum + bec + lex;
                }
            }
        }
        wave_lds_sync();  // the pass's LDS reads are done before the next pass overwrites the slots
    }
}

template <int NT, int KS, int MODE>
__global__ void __launch_bounds__(PML_BLOCK)
eigen_fused_kernel(PmlTree t, PmlCols c, PmlModel m, PmlState st, const int* __restrict__ nodes, int first,
                   int n_nodes, int tips) {
    typedef EigShape<KS> S;
    constexpr int NB = S::NB;
    extern __shared__ double smem[];
    EigWave<NT, KS> W;
    eig_wave_init<NT, KS>(W, c, m, smem, S::WAVE_LDS, NB);
    W.colN = (size_t)W.col * t.N;
    const int wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * PML_WAVES_PER_BLOCK;
    for (int b0 = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * NB; b0 < n_nodes; b0 += waves_total * NB) {
        const bool act = W.hi < NB && b0 + W.hi < n_nodes;
        const int n = act ? (nodes != nullptr ? nodes[b0 + W.hi] : first + b0 + W.hi) : 0;
        eig_pass<NT, KS, MODE>(W, t, c, st, act, n, tips);
    }
}

// The narrow end of a large forest in one launch: one workgroup per column walks the levels [0, n_levels) of a level
// table (offsets into `nodes`, or node id ranges when nodes == nullptr) with a workgroup barrier between levels.  The
// prologue of the fused kernels (A, Ainv into registers / LDS) and the launch are paid once instead of once per level.
template <int NT, int KS, int MODE>
__global__ void __launch_bounds__(PML_BLOCK)
eigen_narrow_kernel(PmlTree t, PmlCols c, PmlModel m, PmlState st, const int* __restrict__ nodes,
                    const int* __restrict__ level_offsets, int n_levels) {
    typedef EigShape<KS> S;
    constexpr int NB = S::NB;
    extern __shared__ double smem[];
    EigWave<NT, KS> W;
    eig_wave_init<NT, KS>(W, c, m, smem, S::WAVE_LDS, NB);
    W.colN = (size_t)W.col * t.N;
    const int wave = threadIdx.x >> 6;
    for (int l = 0; l < n_levels; ++l) {
        const int a = level_offsets[l], n_level = level_offsets[l + 1] - a;
        for (int b0 = wave * NB; b0 < n_level; b0 += PML_WAVES_PER_BLOCK * NB) {
            const bool act = W.hi < NB && b0 + W.hi < n_level;
            const int n = act ? (nodes != nullptr ? nodes[a + b0 + W.hi] : a + b0 + W.hi) : 0;
            eig_pass<NT, KS, MODE>(W, t, c, st, act, n, 0);
        }
        __syncthreads();
    }
}

// Bottom-up messages of the tips, 16 at a time.  An observed tip (one allowed state s) has the unit vector as its
// bottom-up vector, so its message is row s of P^T: the 16 tips of a pass are the 16 rows of ONE tile (NT * KS MFMAs
// for 16 nodes instead of 12.5 per node), and the arg-max table of the joint variant has a closed form (numpy's first
// maximum of (.., 0, P[i][s], 0, ..)).  A pass that holds a tip with several allowed states runs the general passes.
template <int NT, int KS, int JOINT>
__global__ void __launch_bounds__(PML_BLOCK)
eigen_tips_kernel(PmlTree t, PmlCols c, PmlModel m, PmlState st, const int* __restrict__ tip_ids, int n_tips) {
    typedef EigShape<KS> S;
    constexpr int KP = S::KP, NB = S::NB;
    constexpr int MODE = JOINT ? PML_EIG_BU_JOINT : PML_EIG_BU_MARG;
    extern __shared__ double smem[];
    EigWave<NT, KS> W;
    eig_wave_init<NT, KS>(W, c, m, smem, S::WAVE_LDS_TIPS, 16);
    W.colN = (size_t)W.col * t.N;
    const int k = W.k, ks = W.ks, lo = W.lo, hi = W.hi;
    const size_t colN = W.colN;
    int* sS = reinterpret_cast<int*>(W.sT + 16);  // states of the 16 tips
    int* sN = sS + 16;                            // their node ids (-1: no tip in this row)
    const u64 kbits = state_bits(k);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * PML_WAVES_PER_BLOCK;
    for (int b0 = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * 16; b0 < n_tips; b0 += waves_total * 16) {
        const bool have = b0 + lo < n_tips;          // lane (lo, .) looks at tip lo of the pass
        const int tip = tip_ids[have ? b0 + lo : b0];
        const u64 word = c.masks[colN + tip] & kbits;
        const bool observed = __popcll(word) == 1;
        if (!__all(observed || !have)) {
            for (int g = 0; g < 16; g += NB) {
                const bool act = hi < NB && b0 + g + hi < n_tips;
                const int n = act ? tip_ids[b0 + g + hi] : 0;
                eig_pass<NT, KS, MODE>(W, t, c, st, act, n, 1);
            }
            continue;
        }
        const int s_own = observed ? __builtin_ctzll(word) : 0;
        if (hi == 0) {
            W.sT[lo] = have ? (t.dist[tip] + W.tau) * W.tf * W.sfc : 0.0;
            sS[lo] = s_own;
            sN[lo] = have ? tip : -1;
        }
        wave_lds_sync();
        for (int e = lane; e < 16 * KP; e += 64) {
            const int q = e / KP, mm = e % KP;
            W.sE[e] = mm < k ? exp(W.gd[mm] * W.sT[q]) : 0.0;
        }
        wave_lds_sync();
        pml_v4f64 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (pml_v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int mm = 4 * s + hi;
            const double a = have ? W.sB[mm * k + s_own] * W.sE[lo * KP + mm] : 0.0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, W.bfrag[nt][s], acc[nt], 0, 0, 0);
        }
        // D: row hi + 4 reg = tip of the pass, col lo (+16 nt) = state i:  acc = P[i][s_tip]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rr = hi + 4 * reg;
            const int n = sN[rr];
            if (n < 0) continue;
            const int sr = sS[rr];
            const size_t row = (colN + n) * ks;
            u64 mi = 0, mc = 0;
            bool altered = false;
            if (JOINT && c.masks_init != nullptr) {
                mi = c.masks_init[colN + n];
                mc = c.masks[colN + n];
                altered = mi != mc;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int i = 16 * nt + lo;
                if (i >= ks) continue;
                double pv = i < k ? acc[nt][reg] : 0.0;
                if (JOINT) {
                    int arg;
                    if (pv > 0.0) {
                        arg = sr;
                    } else if (sr != 0) {
                        pv = 0.0;
                        arg = 0;
                    } else if (pv == 0.0 || k == 1) {
                        arg = 0;
                    } else {
                        pv = 0.0;
                        arg = 1;
                    }
                    if (altered && !((mi >> arg) & 1ull)) arg = mi ? __builtin_ctzll(mi) : 0;
                    st.J[row + i] = (pml_jt)(i < k ? arg : 0);
                }
                st.msg[row + i] = pv;
            }
        }
        wave_lds_sync();
    }
}
