// Eigen models (CUSTOM_RATES, JTT), 2 <= k <= 128: the SUM sweeps (marginal bottom-up, top-down) without ever forming
// P(t).  Both apply P(t_n) = A diag(exp(d t_n)) A^-1 of the branch above a node n to ONE vector of that node
//   bottom-up  (ml.py:124-148)   msg_n = P v_n        v_n = mask o prod of the children's messages
//   top-down   (ml.py:273-290)   td_n  = P x_n        x_n = TD_p o BU_p / msg_n
// so  P v = A (e o (A^-1 v)),  e_m = exp(d_m t_n):  two k x k matrix-vector products and k exponentials per node -- 4 k^2
// flops instead of the 2 k^3 of building the matrix first (pml_kernels_eigen_mfma.h, which the joint sweep keeps: a
// maximum over j of P[i][j] v[j] needs the entries of P).  Sixteen nodes of a level are the sixteen columns of the
// right-hand operand, so the products are two small GEMMs on the FP64 matrix cores with the constant matrices A^-1, A as
// the left operands held in registers:
//   Y[m][node] = sum_j Ainv[m][j] V[j][node],   Z = Y o E,   MSG[i][node] = sum_m A[i][m] Z[m][node].
// In v_mfma_f64_16x16x4_f64 the right operand of k-step s is held by lane (lo, hi) as element [4 s + hi][column lo] and
// the result tile hands lane (lo, hi) the rows hi + 4 reg of column lo: with the operands' rows and columns enumerated
// accordingly (EigGemm<KS>::st) a lane that owns KS states of node lo gets exactly those states of Y, and then of MSG,
// back -- both GEMMs, the exponentials and everything per node stay in the lane; only per-node reductions (zero check,
// rescaling, likelihood sum) cross the four lanes (lo, lo + 16, lo + 32, lo + 48) of a node.  20 MFMAs per 16 nodes at k = 20 (the matrix-building sweep: 200).
//
// More than 32 states (round 6: codon models, k = 61 / 64; HIV1C's k = 36 / 67 columns under CUSTOM_RATES used to fall back to
// P(t) materialised in HBM, 32 KB per branch and column at k = 64): the same two GEMMs with 3 or 4 row tiles.  The constant
// operands no longer fit the register file (2 x 4 x 16 doubles per lane), so the block keeps them in LDS in operand layout --
// [tile][k-step][lane], one conflict-free ds_read_b64 per MFMA, 64 KB at k = 64, shared by the block's four waves -- and a
// matrix instruction (64 clocks of its SIMD) hides the read.  EigGemm<KS>::LDS says which form a shape takes: the LDS form
// from 17 states on -- the registers it frees are a wave more per SIMD (65 536 tips x 32 columns, marginal pass: k = 32
// 2.75 -> 2.29 ms, k = 24 1.97 -> 1.81, k = 20 1.76 -> 1.67; no difference at 16 states and below).
//
// More than 64 states (round 6, up to 128): both operands in LDS would be 2 x 128 KB.  The models this library is handed are
// reversible (pastml/models/generator.py:33-51: Q[i][j] = r[i][j] pi[j] with symmetric r), so Pi^1/2 Q Pi^-1/2 is symmetric,
// its eigenvectors U = Pi^1/2 A (columns normalised) are orthonormal and
//     P(t) = As diag(exp(d t)) As^T Pi,     As = Pi^-1/2 U:
// ONE matrix serves both products -- the first reads it transposed, pi goes into the vector before it (EigGemm<KS>::SYM,
// KS > 16).  The copy in LDS is a plain [4 KS][4 KS + 1] array (an odd row stride: the 16 rows, or the 16 columns, of an operand
// tile fall into different banks either way).  numpy's eigenvectors (a general, not a symmetric solver: generator.py:16-30) are
// orthonormal to 1e-12 only, and 1e-12 of inconsistency between the two products is 1e-9 on the entries of P(t) v that are
// t q[i][j] small: eig_sym_kernel makes As from A with one Newton-Schulz step U <- U (3 I - U^T U) / 2 (orthonormal to 1e-15;
// the step moves the eigenvectors by what numpy's error already was), after which the sweeps agree with those on the
// reference's P(t) to 7e-11.  pml_model_set_eigen checks on what it is handed that A^-1 IS A transposed and rescaled, entry by
// entry; the sweeps of a model where it is not (a repeated eigenvalue, whose eigenvectors numpy leaves far from orthogonal)
// read materialised P(t) as before.
//
// Rounding: P v is evaluated in a different order than the reference's (P built, then applied).  Both carry an
// absolute error of a few ulps of |A| |A^-1| |v| (cond(A) = 2 for JTT); messages are bounded below by
// min_j P[i][j] max v, so relative errors stay ~1e-11 (tests: 1e-9 on posteriors, 1e-11 on ln L).
#pragma once
#include "pml_kernels_eigen_mfma.h"


template <int KS>
struct EigGemm {
    static constexpr int KP = 4 * KS;             // padded states
    static constexpr int MT = (KP + 15) / 16;     // row tiles of the constant matrices
    static constexpr bool LDS = KS >= 5;          // the constant operands live in LDS, not in registers (k > 16)
    // Which state (and which eigenvalue) slot s of lane (lo, hi) stands for.  The matrix instruction only fixes that k-step s
    // takes its four k-indices from the four hi-lanes and hands rows hi + 4 r back: WHICH index sits there is the operands'
    // business (both constant operands are loaded with their rows and columns in this order).  For an even KS a lane owns
    // PAIRS of neighbouring states -- slots 2 q, 2 q + 1 = states 8 q + 2 hi, + 1 -- so that a vector's part of a lane moves
    // as 16-byte accesses and the four lanes of a node cover 64 contiguous bytes per instruction: half the load / store
    // instructions and half the (instruction x cache line) touches of 4 s + hi, which is what these sweeps are bound by
    // (measured: KS consecutive states per lane -- every lane in a line of its own -- lost 25 %; profiles/r06g_eigen_sweeps.txt).
    static constexpr bool PAIRS = KS % 2 == 0;
    static __device__ __forceinline__ int st(int s, int hi) { return PAIRS ? 8 * (s >> 1) + 2 * hi + (s & 1) : 4 * s + hi; }
    // more than 64 states: one matrix, A, as a plain [KP][LD] array; the first product reads it transposed (see above)
    static constexpr bool SYM = KS > 16;
    static constexpr int LD = KP + 1;
    static constexpr int LDS_DOUBLES = SYM ? KP * LD : (LDS ? 2 * MT * KS * 64 : 0);
    static constexpr int CST_DOUBLES = 8 * KS;   // d, pi of the column, zero beyond k
    // the part of an operand's LDS address that does not depend on the lane (SYM; pairs: even KS only)
    static __device__ __forceinline__ int cs(int s) { return 8 * (s >> 1) + (s & 1); }
};

// per-node reductions over the four lanes (lo, lo + 16 q) that share a node
__device__ __forceinline__ bool node_any(bool p, int lo) {
    const u64 b = __ballot(p);
    return (((b | (b >> 16) | (b >> 32) | (b >> 48)) >> lo) & 1ull) != 0ull;
}

__device__ __forceinline__ double node_max(double v) {
    v = fmax(v, __shfl_xor(v, 16, 64));
    return fmax(v, __shfl_xor(v, 32, 64));
}

__device__ __forceinline__ double node_sum(double v) {  // fixed order: bit-reproducible
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// exact lazy rescaling of a node's vector spread over its four lanes (same rule as lazy_rescale in pml_device.h)
template <int KS>
__device__ __forceinline__ int node_lazy_rescale(double (&v)[KS], int lo) {
    bool out_of_band = false;
    double m = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        m = fmax(m, v[s]);
        out_of_band |= (v[s] != 0.0) && (v[s] < 0x1p-200 || v[s] > 0x1p+200);
    }
    if (!__any(out_of_band)) return 0;           // wave-uniform early out (the common case)
    const bool mine = node_any(out_of_band, lo);
    m = node_max(m);
    if (!mine || !(m > 0.0) || isinf(m)) return 0;
    const int ex = exponent_of(m);
#pragma unroll
    for (int s = 0; s < KS; ++s) v[s] = __builtin_ldexp(v[s], -ex);
    return ex;
}

// per-wave constants of a column: the left operands of the two products in registers
template <int KS>
struct EigGemmWave {
    int k, ks, col, lo, hi;
    size_t colN;
    // (registers for k <= 32; one element each when the operands are in LDS: la1 / la2 point at the lane's entries there)
    double a1[EigGemm<KS>::LDS ? 1 : EigGemm<KS>::MT][EigGemm<KS>::LDS ? 1 : KS];
    double a2[EigGemm<KS>::LDS ? 1 : EigGemm<KS>::MT][EigGemm<KS>::LDS ? 1 : KS];
    const double* lds;   // the block's copy of the operands: [2][MT][KS][64]; SYM: A as [KP][LD]
    const double* ainvT; // the column's transposed A^-1 in memory (observed tips), rows ldT apart; null if there is none
    int ldT;
    const double* cst;   // LDS: the column's eigenvalues d [4 KS] and frequencies pi [4 KS], zero beyond k
                         // (read when needed: as per-lane registers they cost 64 VGPRs through both products)
    double sfc, tau, tf;
    int b1, b2;          // SYM: the lane's part of the operands' addresses -- first product (A read transposed), second product
    // `at`: the lane's offset, made opaque once per pass (eig_gemm_pass) -- the operands are the same in every pass and the
    // compiler would otherwise hoist all 2 MT KS reads out of the node loop, back into 256 registers
    __device__ __forceinline__ double op1(int mt, int s, int at) const {
        if (EigGemm<KS>::SYM) return lds[at + EigGemm<KS>::cs(s) * EigGemm<KS>::LD + 16 * mt];   // As[st(s, hi)][16 mt + r0]
        return EigGemm<KS>::LDS ? lds[at + (mt * KS + s) * 64] : a1[EigGemm<KS>::LDS ? 0 : mt][EigGemm<KS>::LDS ? 0 : s];
    }
    __device__ __forceinline__ double op2(int mt, int s, int at) const {
        if (EigGemm<KS>::SYM) return lds[at + 16 * mt * EigGemm<KS>::LD + EigGemm<KS>::cs(s)];   // As[16 mt + r0][st(s, hi)]
        return EigGemm<KS>::LDS ? lds[at + (EigGemm<KS>::MT * KS + mt * KS + s) * 64]
                                : a2[EigGemm<KS>::LDS ? 0 : mt][EigGemm<KS>::LDS ? 0 : s];
    }
};

// smem: EigGemm<KS>::LDS_DOUBLES doubles of dynamic LDS (nullptr for the register form)
template <int KS>
__device__ __forceinline__ void eig_gemm_init(EigGemmWave<KS>& W, const PmlTree& t, const PmlCols& c, const PmlModel& m,
                                              double* smem, double* cst) {
    constexpr int MT = EigGemm<KS>::MT;
    const int k = c.k, ks = c.ks;  // k <= ks <= 4 KS: states 4 s + hi >= ks do not exist in memory
    const int col = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int lo = lane & 15, hi = lane >> 4;
    W.k = k;
    W.ks = ks;
    W.col = col;
    W.lo = lo;
    W.hi = hi;
    W.colN = (size_t)col * t.N;
    // constant operands: tile mt, lane lo holds the row of index st(4 mt + lo / 4, lo % 4) of A^-1 (first product) and of A
    // (second) -- the result tile then hands lane (lo, hi), in register r, the row st(4 mt + r, hi): its own slot 4 mt + r --;
    // k-step s: column st(s, hi)
    const double* gA = m.A + (size_t)col * k * k;
    const double* gB = m.Ainv + (size_t)col * k * k;
    W.lds = smem;
    W.b1 = W.b2 = 0;
    if (EigGemm<KS>::SYM) {
        static_assert(!EigGemm<KS>::SYM || EigGemm<KS>::PAIRS, "more than 64 states: even KS only");
        constexpr int LD = EigGemm<KS>::LD;
        const double* gS = m.Asym + (size_t)col * k * k;   // As = Pi^-1/2 U (eig_sym_kernel)
        for (int e = threadIdx.x; e < 4 * KS * LD; e += blockDim.x) {
            const int r = e / LD, q = e - r * LD;
            smem[e] = (r < k && q < k) ? gS[r * k + q] : 0.0;
        }
        // row of the lane within an operand tile: st(4 mt + (lo >> 2), lo & 3) = 16 mt + r0
        const int r0 = 8 * (lo >> 3) + 2 * (lo & 3) + ((lo >> 2) & 1);
        W.b1 = 2 * hi * LD + r0;
        W.b2 = r0 * LD + 2 * hi;
        __syncthreads();
    } else if (EigGemm<KS>::LDS) {
        // operand layout [tile][k-step][lane]: entry e of a matrix belongs to lane e % 64 of (mt, s) = (e / 64 / KS, e / 64 % KS)
        for (int e = threadIdx.x; e < MT * KS * 64; e += blockDim.x) {
            const int l = e & 63, ms = e >> 6, mt = ms / KS, s2 = ms - mt * KS;
            const int lo2 = l & 15;
            const int row = EigGemm<KS>::st(4 * mt + (lo2 >> 2), lo2 & 3), cc = EigGemm<KS>::st(s2, l >> 4);
            const bool in = 4 * mt + (lo2 >> 2) < KS && row < k && cc < k;   // (slots beyond KS: rows nobody reads, kept zero)
            smem[e] = in ? gB[row * k + cc] : 0.0;
            smem[MT * KS * 64 + e] = in ? gA[row * k + cc] : 0.0;
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int row = EigGemm<KS>::st(4 * mt + (lo >> 2), lo & 3), cc = EigGemm<KS>::st(s, hi);
                const bool in = 4 * mt + (lo >> 2) < KS && row < k && cc < k;   // (slots beyond KS: rows nobody reads, kept zero)
                W.a1[EigGemm<KS>::LDS ? 0 : mt][EigGemm<KS>::LDS ? 0 : s] = in ? gB[row * k + cc] : 0.0;
                W.a2[EigGemm<KS>::LDS ? 0 : mt][EigGemm<KS>::LDS ? 0 : s] = in ? gA[row * k + cc] : 0.0;
            }
    }
    for (int e = threadIdx.x; e < 4 * KS; e += blockDim.x) {
        cst[e] = e < k ? m.d[(size_t)col * k + e] : 0.0;
        cst[4 * KS + e] = e < k ? c.pi[(size_t)col * ks + e] : 0.0;
    }
    W.cst = cst;
    __syncthreads();
    W.sfc = m.sf[col];
    W.tau = m.tau[col];
    W.tf = m.tauf[col];
    W.ldT = m.ldT;
    W.ainvT = (m.AinvT != nullptr && k <= m.ldT) ? m.AinvT + (size_t)col * m.ldT * m.ldT : nullptr;
}

// A lane's part of a row of ks doubles (states st(s, hi)): 16-byte accesses where the lane owns pairs and the rows are 16-byte
// aligned (ks even), 8-byte ones otherwise.  Entries at or beyond ks: zero on loads, skipped on stores.
template <int KS>
__device__ __forceinline__ void eig_row_load(const double* __restrict__ row, int hi, int ks, double (&v)[KS]) {
    if (EigGemm<KS>::PAIRS && (ks & 1) == 0) {
#pragma unroll
        for (int q = 0; q < KS / 2; ++q) {
            const int j = 8 * q + 2 * hi;
            double2 t2 = {0.0, 0.0};
            if (j < ks) t2 = *reinterpret_cast<const double2*>(row + j);
            v[2 * q] = t2.x;
            v[2 * q + 1] = t2.y;
        }
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) v[s] = EigGemm<KS>::st(s, hi) < ks ? row[EigGemm<KS>::st(s, hi)] : 0.0;
    }
}

template <int KS>
__device__ __forceinline__ void eig_row_store(double* __restrict__ row, int hi, int ks, const double (&v)[KS]) {
    if (EigGemm<KS>::PAIRS && (ks & 1) == 0) {
#pragma unroll
        for (int q = 0; q < KS / 2; ++q) {
            const int j = 8 * q + 2 * hi;
            if (j < ks) {
                double2 t2;
                t2.x = v[2 * q];
                t2.y = v[2 * q + 1];
                *reinterpret_cast<double2*>(row + j) = t2;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (EigGemm<KS>::st(s, hi) < ks) row[EigGemm<KS>::st(s, hi)] = v[s];
    }
}

// What a pass knows about its node before it reads any vector: the kernels load it ONE PASS AHEAD (round 6: a pass was a chain of
// dependent round trips -- node list -> tree arrays -> first child's message -> second child's message -- during which the
// wave did nothing: 73 % of a wave's cycles at k = 61 were waits, profiles/r06g_eigen_sweeps.txt).
struct EigNode {
    int n, fc, nc, p;
    u64 word;     // allowed-state mask (k <= 64: one word)
    u64 word2;    // states 64 .. 127 (more than 64 states: two words per node)
    double dist;
};

template <int MODE, bool TWO = false>
__device__ __forceinline__ EigNode eig_node_load(const PmlTree& t, const PmlCols& c, size_t colN, bool act, int n) {
    EigNode q;
    q.n = act ? n : 0;
    q.word = TWO ? c.masks[(colN + q.n) * 2] : c.masks[colN + q.n];
    q.word2 = TWO ? c.masks[(colN + q.n) * 2 + 1] : 0ull;
    q.nc = t.n_children[q.n];
    q.dist = t.dist[q.n];
    q.fc = MODE == PML_EIGG_BU ? t.first_child[q.n] : 0;
    q.p = MODE == PML_EIGG_TD ? t.parent[q.n] : 0;
    return q;
}

// One pass of a wave: 16 nodes, lane (lo, hi) owns the states st(s, hi) = hi KS + s of node n (act: slot lo holds a node).
template <int KS, int MODE>
__device__ __forceinline__ void eig_gemm_pass(const EigGemmWave<KS>& W, const PmlTree& t, const PmlCols& c,
                                              const PmlState& st, bool act, const EigNode& q) {
    constexpr int MT = EigGemm<KS>::MT;
    const int k = W.k, ks = W.ks, col = W.col, lo = W.lo, hi = W.hi;
    const size_t colN = W.colN;
    const double sfc = W.sfc, tau = W.tau, tf = W.tf;
    int at = EigGemm<KS>::SYM ? W.b1 : (int)(threadIdx.x & 63), at2 = EigGemm<KS>::SYM ? W.b2 : at;
    if (EigGemm<KS>::LDS) asm volatile("" : "+v"(at));
    if (EigGemm<KS>::SYM) asm volatile("" : "+v"(at2));
    else at2 = at;
    const int n = q.n;
    {
        const size_t row = (colN + n) * ks;
        const u64 word = q.word, word2 = q.word2;
        const int nc = q.nc;
        const double tt = (q.dist + tau) * tf * sfc;
        double v[KS];
        // the node's mask as a 0 / 1 vector (formed where it is used: two registers of mask word instead of 2 KS through the products)
        auto mask_vec = [&](double (&mb)[KS]) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int j = EigGemm<KS>::st(s, hi);
                const u64 wj = (EigGemm<KS>::SYM && j >= 64) ? word2 : word;
                mb[s] = (j < k && ((wj >> (j & 63)) & 1ull)) ? 1.0 : 0.0;
            }
        };
        i64 esum = 0;
        // ------------------------------------------------------------------ the vector P is applied to
        i64 bec = 0;
        bool tipc = false;
        if (MODE != PML_EIGG_TD) {
            mask_vec(v);
            if (MODE == PML_EIGG_BU) {
                // mask o prod of the children's messages (ml.py:126-148), zero check and rescaling per child; a tip's
                // exponent word is zero (never written by these sweeps)
                const int fc = q.fc;
                int most = nc;
                most = max(most, __shfl_xor(most, 1, 64));
                most = max(most, __shfl_xor(most, 2, 64));
                most = max(most, __shfl_xor(most, 4, 64));
                most = max(most, __shfl_xor(most, 8, 64));
                // the children two at a time: the loads of a pair go out together (one round trip for a binary node)
                for (int j0 = 0; j0 < most; j0 += 2) {   // wave-uniform trip count (ballots inside)
                    double mv[2][KS];
                    i64 cbe[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int ch = (act && j0 + u < nc) ? fc + j0 + u : n;
                        eig_row_load<KS>(st.msg + (colN + ch) * ks, hi, ks, mv[u]);
                        cbe[u] = st.be[colN + ch];
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (j0 + u >= most) break;
                        const bool has = act && j0 + u < nc;
                        bool nz = false;
                        if (has) {
#pragma unroll
                            for (int s = 0; s < KS; ++s) {
                                v[s] *= fmax(mv[u][s], 0.0);
                                nz |= v[s] != 0.0;
                            }
                            esum += cbe[u];
                        }
                        const bool alive = node_any(nz, lo);
                        if (has && !alive && hi == 0)
                            atomicMin(&st.err[col], ((u64)(unsigned)t.post_rank[n] << 32) | (u64)(unsigned)(fc + j0 + u));
                        const int ex = node_lazy_rescale<KS>(v, lo);
                        if (has) esum += ex;
                    }
                }
                if (act) {
                    eig_row_store<KS>(st.bu + row, hi, ks, v);
                    if (hi == 0) st.be[colN + n] = esum;
                }
            }
        } else {
            // x = TD_p o BU_p / msg_n (ml.py:279-283); the message is what the bottom-up sweep left
            const int p = act ? q.p : 0;
            const size_t prow = (colN + (p < 0 ? 0 : p)) * ks;
            tipc = nc == 0;
            bec = tipc ? 0 : st.be[colN + n];
            const i64 pe = st.te[colN + (p < 0 ? 0 : p)] + st.be[colN + (p < 0 ? 0 : p)];
            {
                // (rows at or beyond ks read as zero: 0 * 0 / 1 = 0; a tip's own row is allocated but never written)
                double tdp[KS], bup[KS], mc[KS];
                eig_row_load<KS>(st.td + prow, hi, ks, tdp);
                eig_row_load<KS>(st.bu + prow, hi, ks, bup);
                eig_row_load<KS>(st.msg + row, hi, ks, mc);
#pragma unroll
                for (int s = 0; s < KS; ++s) v[s] = (tdp[s] * bup[s]) / (mc[s] > 0.0 ? mc[s] : 1.0);
            }
            esum = pe - bec;
            esum += node_lazy_rescale<KS>(v, lo);
        }
        // ------------------------------------------------------------------ Y = A^-1 V, Z = Y o exp(d t), OUT = A Z
        pml_v4f64 acc[MT];
        double z[KS];
        // Observed tips (one allowed state s: the vector is the unit vector e_s): Y = A^-1 e_s is COLUMN s of A^-1 -- a row of the
        // transposed copy, which the four lanes of a node read 32 contiguous bytes at a time -- so the first product is a gather
        // instead of MT x KS matrix instructions (half of a tip's pass; tips are half of the nodes).  Wave-uniform choice; the sums
        // of the product degenerate to their one non-zero term, so the bits are those of the GEMM.
        const u64 kb = k >= 64 ? ~0ull : (1ull << k) - 1ull;
        constexpr bool SYM = EigGemm<KS>::SYM;
        const u64 kb2 = !SYM || k <= 64 ? 0ull : (k >= 128 ? ~0ull : (1ull << (k - 64)) - 1ull);
        if (MODE == PML_EIGG_TIPS && (SYM || W.ainvT != nullptr) &&
            __all(!act || __popcll(word & kb) + (SYM ? __popcll(word2 & kb2) : 0) == 1)) {
            if (SYM) {
                // column s of A^-1 = row s of As, times pi_s: from the block's copy of As
                const int so = !act ? 0 : ((word & kb) ? __builtin_ctzll(word & kb) : 64 + __builtin_ctzll(word2 & kb2));
                const double* rowA = W.lds + (size_t)so * EigGemm<KS>::LD;
                const double pis = W.cst[4 * KS + so];
#pragma unroll
                for (int s = 0; s < KS; ++s) z[s] = act ? rowA[EigGemm<KS>::st(s, hi)] * pis : 0.0;
            } else {
                const double* colT = W.ainvT + (size_t)(act ? __builtin_ctzll(word & kb) : 0) * W.ldT;
                eig_row_load<KS>(colT, hi, act ? min(W.ldT, 4 * KS) : 0, z);   // (padded with zeros beyond k)
            }
            {
                double dl[KS];
                eig_row_load<KS>(W.cst, hi, 4 * KS, dl);
#pragma unroll
                for (int s = 0; s < KS; ++s) z[s] *= exp(dl[s] * tt);
            }
        } else {
            if (SYM) {   // A^-1 v = As^T (pi o v)
                double pil[KS];
                eig_row_load<KS>(W.cst + 4 * KS, hi, 4 * KS, pil);
#pragma unroll
                for (int s = 0; s < KS; ++s) v[s] *= pil[s];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt] = (pml_v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(W.op1(mt, s, at), act ? v[s] : 0.0, acc[mt], 0, 0, 0);
            }
            double dl[KS];
            eig_row_load<KS>(W.cst, hi, 4 * KS, dl);
#pragma unroll
            for (int s = 0; s < KS; ++s) z[s] = acc[s / 4][s % 4] * exp(dl[s] * tt);  // rows st(s, hi) of Y: the lane's own
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt] = (pml_v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(W.op2(mt, s, at2), z[s], acc[mt], 0, 0, 0);
        }
        double out[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) out[s] = (EigGemm<KS>::st(s, hi) < k) ? acc[s / 4][s % 4] : 0.0;
        // ------------------------------------------------------------------ results of the node
        if (MODE != PML_EIGG_TD) {
            if (act) {
                eig_row_store<KS>(st.msg + row, hi, ks, out);
            }
        } else {
            // marginal likelihoods pi o mask o BU o TD (ml.py:456-460) and posteriors (ml.py:498-500)
            // (the node's own bottom-up vector is read here, behind the products: nothing of the node but its mask word and
            // scalars lives through them)
            double tdc[KS], lh[KS], vc[KS], mb[KS], pil[KS];
            mask_vec(mb);
            if (!__all(tipc)) eig_row_load<KS>(st.bu + row, hi, ks, vc);   // (a tip's row is allocated but never written)
            eig_row_load<KS>(W.cst + 4 * KS, hi, 4 * KS, pil);
            double lhs = 0.0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (tipc) vc[s] = mb[s];
                tdc[s] = fmax(out[s], 0.0);
                lh[s] = vc[s] * tdc[s] * (pil[s] * mb[s]);
                lhs += lh[s];
            }
            lhs = node_sum(lhs);
            const int lex = (lhs > 0.0 && !isinf(lhs)) ? exponent_of(lhs) : 0;
            if (act) {
                if (!tipc) eig_row_store<KS>(st.td + row, hi, ks, tdc);
#pragma unroll
                for (int s = 0; s < KS; ++s) lh[s] = lh[s] / lhs;
                eig_row_store<KS>(st.post + row, hi, ks, lh);
                if (hi == 0) {
                    if (!tipc) st.te[colN + n] = esum;
                    st.lhsum[colN + n] = __builtin_ldexp(lhs, -lex);
                    st.lhe[colN + n] = esum + bec + lex;
                }
            }
        }
    }
}

// one level of a sweep
// (two waves per SIMD wherever the top-down mode's registers allow it: the bottom-up modes of the widest shapes would
// otherwise settle a few registers above the limit and run one)
template <int KS, int MODE>
__global__ void __launch_bounds__(PML_BLOCK) __attribute__((amdgpu_waves_per_eu(((MODE == PML_EIGG_TD && KS >= 8) || KS > 16) ? 1 : 2, 8)))
eigen_gemm_kernel(PmlTree t, PmlCols c, PmlModel m, PmlState st, const int* __restrict__ nodes, int first, int n_nodes) {
    extern __shared__ double eigg_smem[];
    __shared__ double eigg_cst[EigGemm<KS>::CST_DOUBLES];
    EigGemmWave<KS> W;
    eig_gemm_init<KS>(W, t, c, m, eigg_smem, eigg_cst);
    const int wave = threadIdx.x >> 6;
    const int stride = gridDim.x * PML_WAVES_PER_BLOCK * 16;
    int b0 = (blockIdx.x * PML_WAVES_PER_BLOCK + wave) * 16;
    if (b0 >= n_nodes) return;
    auto id_of = [&](int b, bool a) { return a ? (nodes != nullptr ? nodes[b + W.lo] : first + b + W.lo) : 0; };
    // the node data of the next pass and the node id of the one after it are in flight while this pass runs
    bool act = b0 + W.lo < n_nodes;
    EigNode cur = eig_node_load<MODE, EigGemm<KS>::SYM>(t, c, W.colN, act, id_of(b0, act));
    bool act1 = b0 + stride + W.lo < n_nodes;
    int n1 = id_of(b0 + stride, act1);
    for (; b0 < n_nodes; b0 += stride) {
        const bool act2 = b0 + 2 * stride + W.lo < n_nodes;
        const int n2 = id_of(b0 + 2 * stride, act2);
        const EigNode nxt = eig_node_load<MODE, EigGemm<KS>::SYM>(t, c, W.colN, act1, n1);
        eig_gemm_pass<KS, MODE>(W, t, c, st, act, cur);
        cur = nxt;
        act = act1;
        n1 = n2;
        act1 = act2;
    }
}

// The narrow end of a large forest in one launch: one workgroup per column walks the levels [0, n_levels) of a level
// table (offsets into `nodes`, or node id ranges when nodes == nullptr) with a workgroup barrier between levels: the
// launch and the loading of the constant operands are paid once instead of once per level.
template <int KS, int MODE>
__global__ void __launch_bounds__(PML_BLOCK)
eigen_gemm_narrow_kernel(PmlTree t, PmlCols c, PmlModel m, PmlState st, const int* __restrict__ nodes,
                         const int* __restrict__ level_offsets, int n_levels, const int* __restrict__ blk_start) {
    // (blk_start: the subtree blocks of a tier of thin levels, one workgroup per (block, column) -- pml_ctx::EigenTiers)
    if (blk_start != nullptr) level_offsets += blk_start[blockIdx.x];
    extern __shared__ double eigg_smem[];
    __shared__ double eigg_cst[EigGemm<KS>::CST_DOUBLES];
    EigGemmWave<KS> W;
    eig_gemm_init<KS>(W, t, c, m, eigg_smem, eigg_cst);
    const int wave = threadIdx.x >> 6;
    for (int l = 0; l < n_levels; ++l) {
        const int a = level_offsets[l], n_level = level_offsets[l + 1] - a;
        for (int b0 = wave * 16; b0 < n_level; b0 += PML_WAVES_PER_BLOCK * 16) {
            const bool act = b0 + W.lo < n_level;
            const int n = act ? (nodes != nullptr ? nodes[a + b0 + W.lo] : a + b0 + W.lo) : 0;
            eig_gemm_pass<KS, MODE>(W, t, c, st, act, eig_node_load<MODE, EigGemm<KS>::SYM>(t, c, W.colN, act, n));
        }
        __threadfence_block();
        __syncthreads();
    }
}


// As = Pi^-1/2 U for the one-matrix form above (65 - 128 states): U = Pi^1/2 A with normalised columns, made orthonormal by one
// Newton-Schulz step.  Two launches of PML_ESYM_PARTS workgroups per column (every workgroup holds U in LDS and makes its share of
// the k^2 entries): T = (3 I - U^T U) / 2 into `scratch` ([C][k][k], L2), then As = Pi^-1/2 U T.
#define PML_ESYM_PARTS 8
#ifdef PML_PLAIN_KERNELS   // (launched by pml_api.hip only)
PML_GLOBAL void __launch_bounds__(PML_BLOCK)
eig_sym_kernel(int k, int ks, int col_begin, int phase, const double* __restrict__ A, const double* __restrict__ pi,
               double* __restrict__ scratch, double* __restrict__ Asym) {
    extern __shared__ double esym_smem[];
    const int col = col_begin + blockIdx.y;
    const int LD = k + 1;
    double* sU = esym_smem;            // [k][LD]
    double* sG = sU + (size_t)k * LD;  // [k]: 1 / norm of the columns
    const double* gA = A + (size_t)col * k * k;
    const double* gp = pi + (size_t)col * ks;
    double* T = scratch + (size_t)col * k * k;
    double* out = Asym + (size_t)col * k * k;
    for (int e = threadIdx.x; e < k * k; e += blockDim.x) {
        const int j = e / k, mm = e - j * k;
        sU[j * LD + mm] = sqrt(gp[j]) * gA[e];
    }
    __syncthreads();
    for (int mm = threadIdx.x; mm < k; mm += blockDim.x) {
        double g = 0.0;
        for (int j = 0; j < k; ++j) g = __builtin_fma(sU[j * LD + mm], sU[j * LD + mm], g);
        sG[mm] = 1.0 / sqrt(g);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < k * k; e += blockDim.x) {
        const int j = e / k, mm = e - j * k;
        sU[j * LD + mm] *= sG[mm];
    }
    __syncthreads();
    const int stride = gridDim.x * blockDim.x;
    if (phase == 0) {
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < k * k; e += stride) {
            const int q = e / k, mm = e - q * k;
            double g = 0.0;
#pragma unroll 4
            for (int j = 0; j < k; ++j) g = __builtin_fma(sU[j * LD + q], sU[j * LD + mm], g);
            T[e] = (q == mm ? 1.5 : 0.0) - 0.5 * g;
        }
    } else {
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < k * k; e += stride) {
            const int i = e / k, mm = e - i * k;
            double g = 0.0;
#pragma unroll 4
            for (int q = 0; q < k; ++q) g = __builtin_fma(sU[i * LD + q], T[q * k + mm], g);
            out[e] = g / sqrt(gp[i]);
        }
    }
}
#endif
