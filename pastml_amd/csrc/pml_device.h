// Device-side helpers shared by the sweep kernels (gfx950, wave64).
//
// Work decomposition used everywhere: a *unit* is one (node, column) pair; it is processed by a group of G
// consecutive lanes of one wavefront (G a power of two, 1..64), each lane owning R states (F81 kernels: two 16-byte
// pairs per lane, see pml_kernels_f81.h; matrix kernels: state = g * R + r).  G * R >= k.  64 / G units share a
// wavefront; reductions over the states of a unit are butterflies over the G lanes, so they are deterministic
// (fixed order) and never leave the wavefront.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PML_BLOCK 256
#define PML_WAVES_PER_BLOCK 4
#define PML_EIGJ_STRIDE 32                     // row stride (and rows) of the transposed padded copy of Ainv (pml_kernels_eigen_joint.h)
// modes of eigen_gemm_kernel (pml_kernels_eigen_gemm.h) and of the fused matrix-core sweeps (pml_kernels_eigen_mfma.h)
#define PML_EIGG_BU 0    // marginal bottom-up, internal nodes
#define PML_EIGG_TIPS 1  // marginal bottom-up, tips (no children)
#define PML_EIGG_TD 2    // top-down + marginal likelihoods + posteriors
#define PML_EIG_BU_MARG 0
#define PML_EIG_BU_JOINT 1
#define PML_EIG_TD 2
// A kernel that is not a template: internal linkage, and compiled only by the translation unit that launches it -- all
// of them are pml_api.hip's, which defines PML_PLAIN_KERNELS (the compiler emits a static kernel whether it is launched or
// not, so the definitions are left out elsewhere).
#define PML_GLOBAL static __global__

typedef unsigned long long u64;
typedef long long i64;

// lazy rescaling band: a vector is renormalised (max -> [1, 2)) when a non-zero entry leaves [2^-200, 2^200]

// lane exchange inside a 16-lane row by a DPP modifier (VALU only, no LDS crossbar).  Every permutation used here
// reads a lane of the same unit, and the lanes of a unit are always active together, so no lane ever needs the "old"
// value: bound_ctrl lets the compiler drop the two copies that would otherwise initialise the destination.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// x[row 2q] + x[row 2q+1] in both rows (v_permlane16_swap: odd rows of one operand <-> even rows of the other)
__device__ __forceinline__ double row_pair_sum(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// x[lanes 0..31] + x[lanes 32..63] in both halves (v_permlane32_swap)
__device__ __forceinline__ double half_pair_sum(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// Sum over the G lanes of a unit, result in every lane, fixed order (bit-reproducible).
// quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror, then the two gfx950 swap instructions.
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    if (G >= 2) v += dpp_f64<0xB1>(v);
    if (G >= 4) v += dpp_f64<0x4E>(v);
    if (G >= 8) v += dpp_f64<0x141>(v);
    if (G >= 16) v += dpp_f64<0x140>(v);
    if (G >= 32) v = row_pair_sum(v);
    if (G >= 64) v = half_pair_sum(v);
    return v;
}

// 1/x for normal positive x: v_rcp_f64 + two Newton steps (the sweeps never divide by denormals or zeros)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

// 1/x with one Newton step: relative error <= 2.2e-15 measured on gfx950 (v_rcp_f64 alone: 4.6e-8), used where the
// result feeds quantities whose parity bar is 1e-6 and which are renormalised at every level (top-down sweep)
__device__ __forceinline__ double rcp1(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

__device__ __forceinline__ double row_pair_max(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
}

__device__ __forceinline__ double half_pair_max(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
}

// Max over the G lanes of a unit, result in every lane: same lane exchanges as group_sum (no index arithmetic and no
// LDS crossbar, unlike __shfl_xor, whose address computation the compiler hoists into the common path).
template <int G>
__device__ __forceinline__ double group_max(double v) {
    if (G >= 2) v = fmax(v, dpp_f64<0xB1>(v));
    if (G >= 4) v = fmax(v, dpp_f64<0x4E>(v));
    if (G >= 8) v = fmax(v, dpp_f64<0x141>(v));
    if (G >= 16) v = fmax(v, dpp_f64<0x140>(v));
    if (G >= 32) v = row_pair_max(v);
    if (G >= 64) v = half_pair_max(v);
    return v;
}

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

__device__ __forceinline__ void argmax_take(double& v, int& idx, double ov, int oi) {
    if (ov > v || (ov == v && oi < idx)) {
        v = ov;
        idx = oi;
    }
}

// (value, index) arg-max over the G lanes of a unit with numpy semantics -- the first (lowest) index among equal
// maxima wins -- result in every lane.  Same lane exchanges as group_sum (DPP inside a row, then the two swaps).
template <int G>
__device__ __forceinline__ void group_argmax_first(double& v, int& idx) {
    if (G >= 2) argmax_take(v, idx, dpp_f64<0xB1>(v), dpp_i32<0xB1>(idx));
    if (G >= 4) argmax_take(v, idx, dpp_f64<0x4E>(v), dpp_i32<0x4E>(idx));
    if (G >= 8) argmax_take(v, idx, dpp_f64<0x141>(v), dpp_i32<0x141>(idx));
    if (G >= 16) argmax_take(v, idx, dpp_f64<0x140>(v), dpp_i32<0x140>(idx));
    if (G >= 32) {
        const unsigned lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const auto q = __builtin_amdgcn_permlane16_swap((unsigned)idx, (unsigned)idx, false, false);
        // the two results are this row's and the partner row's values, in an order that depends on the row: taking
        // the better of the two is symmetric
        double v0 = __hiloint2double(b[0], a[0]), v1 = __hiloint2double(b[1], a[1]);
        int i0 = (int)q[0], i1 = (int)q[1];
        argmax_take(v0, i0, v1, i1);
        v = v0;
        idx = i0;
    }
    if (G >= 64) {
        const unsigned lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        const auto q = __builtin_amdgcn_permlane32_swap((unsigned)idx, (unsigned)idx, false, false);
        double v0 = __hiloint2double(b[0], a[0]), v1 = __hiloint2double(b[1], a[1]);
        int i0 = (int)q[0], i1 = (int)q[1];
        argmax_take(v0, i0, v1, i1);
        v = v0;
        idx = i0;
    }
}

template <int G>
__device__ __forceinline__ bool group_any(bool p) {
    u64 b = __ballot(p);
    if (G == 64) return b != 0ull;
    const int base = (threadIdx.x & 63) & ~(G - 1);
    return ((b >> base) & ((1ull << (G & 63)) - 1ull)) != 0ull;
}

template <int R>
__device__ __forceinline__ void load_vec(const double* __restrict__ p, double (&v)[R]) {
    if (R == 1) {
        v[0] = p[0];
    } else if (R == 2) {
        double2 t = *reinterpret_cast<const double2*>(p);
        v[0] = t.x;
        v[1] = t.y;
    } else {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            double2 t = *reinterpret_cast<const double2*>(p + r);
            v[r] = t.x;
            v[r + 1] = t.y;
        }
    }
}

template <int R>
__device__ __forceinline__ void store_vec(double* __restrict__ p, const double (&v)[R]) {
    if (R == 1) {
        p[0] = v[0];
    } else {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            double2 t;
            t.x = v[r];
            t.y = v[r + 1];
            *reinterpret_cast<double2*>(p + r) = t;
        }
    }
}

// R consecutive entries of an arg-max table (one byte each)
template <int R>
__device__ __forceinline__ void store_vec_u8(unsigned char* __restrict__ p, const int (&v)[R]) {
    if (R == 4) {
        *reinterpret_cast<unsigned*>(p) = (unsigned)((v[0] & 0xff) | ((v[1] & 0xff) << 8) | ((v[2] & 0xff) << 16) |
                                                     ((unsigned)(v[3] & 0xff) << 24));
    } else if (R == 2) {
        *reinterpret_cast<unsigned short*>(p) = (unsigned short)((v[0] & 0xff) | ((v[1] & 0xff) << 8));
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) p[r] = (unsigned char)v[r];
    }
}

template <int R>
__device__ __forceinline__ void store_vec_i32(int* __restrict__ p, const int (&v)[R]) {
    if (R == 1) {
        p[0] = v[0];
    } else {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            int2 t;
            t.x = v[r];
            t.y = v[r + 1];
            *reinterpret_cast<int2*>(p + r) = t;
        }
    }
}

// mask bits of the R states owned by a lane -> 0.0 / 1.0 (R divides 64, so they share a word)
template <int R>
__device__ __forceinline__ void mask_to_vec(u64 word, int s0, int k, double (&v)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = s0 + r;
        v[r] = (s < k && ((word >> (s & 63)) & 1ull)) ? 1.0 : 0.0;
    }
}

// floor(log2(x)) of a finite positive double (v_frexp_exp_i32_f64 returns the exponent of a mantissa in [0.5, 1))
__device__ __forceinline__ int exponent_of(double x) { return __builtin_amdgcn_frexp_exp(x) - 1; }

// Renormalises acc (max -> [1, 2)) if any non-zero entry left the band; returns the exponent taken out
// (true value = acc * 2^returned).
template <int G, int R>
__device__ __forceinline__ int lazy_rescale(double (&acc)[R]) {
    bool out_of_band = false;
    double m = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double a = acc[r];
        m = fmax(m, a);
        out_of_band |= (a != 0.0) && (a < 0x1p-200 || a > 0x1p+200);
    }
    if (!group_any<G>(out_of_band)) return 0;
    m = group_max<G>(m);
    if (!(m > 0.0) || isinf(m)) return 0;
    const int ex = exponent_of(m);
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = __builtin_ldexp(acc[r], -ex);
    return ex;
}

// first allowed state of a multi-word mask (== np.argmax of the 0/1 array, pastml/ml.py:421)
__device__ __forceinline__ int first_allowed(const u64* __restrict__ m, int W) {
    for (int w = 0; w < W; ++w) {
        if (m[w]) return w * 64 + __builtin_ctzll(m[w]);
    }
    return 0;
}
