// What does the memory system give a bottom-up level of a RAGGED tree?  A unit (8 lanes) reads two adjacent 512-byte rows
// (its stored children) and writes one (its own vector), 32 column slabs of 524 287 rows each (8.6 GB), 34 485 units per
// slab -- the second level of the random 262 144-tip tree.  "dense": rows in unit order (a balanced tree); "sparse":
// ascending random subsets of the slab (what height levels over breadth-first ids look like).  Two-stage software pipeline
// as in bu_f81_kernel; occupancy held at 2 or 8 waves per SIMD by an LDS allocation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double dbl2 __attribute__((ext_vector_type(2)));
struct Row { dbl2 q[4]; };
__device__ __forceinline__ Row load_row(const double* p, int g) {
    Row r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r.q[q] = *reinterpret_cast<const dbl2*>(p + (q * 8 + g) * 2);
    return r;
}
template <int MODE>
__global__ void __launch_bounds__(256) k(const double* __restrict__ in, double* __restrict__ out, const int* __restrict__ rd,
                                         const int* __restrict__ wr, int n_units, size_t slab) {
    extern __shared__ double pad[];
    const int lane = threadIdx.x & 63, g = lane & 7, sub = lane >> 3;
    const double* cin = in + blockIdx.y * slab;
    double* cout = out + blockIdx.y * slab;
    const int stride = gridDim.x * 4 * 8;
    int idx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + sub;
    if (idx >= n_units) return;
    int r = rd[idx], w = wr[idx];
    Row a = load_row(cin + (size_t)r * 64, g), b = load_row(cin + (size_t)(r + 1) * 64, g);
    while (true) {
        const int nxt = idx + stride;
        Row a2 = a, b2 = b;
        int r2 = 0, w2 = 0;
        if (nxt < n_units) {
            r2 = rd[nxt];
            w2 = wr[nxt];
            a2 = load_row(cin + (size_t)r2 * 64, g);
            b2 = load_row(cin + (size_t)(r2 + 1) * 64, g);
        }
        double* o = cout + (size_t)w * 64;
        dbl2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = a.q[q] * b.q[q];
        if (MODE == 1) {  // the next unit's rows are waited for BEFORE this unit's stores (the counter runs in issue order)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(a2.q[q]), "+v"(b2.q[q]));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<dbl2*>(o + (q * 8 + g) * 2) = v[q];
        if (nxt >= n_units) break;
        idx = nxt; a = a2; b = b2; w = w2;
    }
    if (pad[0] == 12345.0) out[0] = 1.0;
}
int main() {
    const int N = 524287, C = 32, U = 34485;
    const size_t slab = (size_t)N * 64;
    double *in, *out;
    hipMalloc(&in, slab * C * 8);
    hipMalloc(&out, slab * C * 8);
    hipMemset(in, 0, slab * C * 8);
    srand(7);
    auto sparse = [&](int top) {
        std::vector<int> v(U);
        // ascending subset of [0, top): keep a row with probability U / top
        int n = 0;
        for (int i = 0; i < top && n < U; ++i)
            if ((double)rand() / RAND_MAX < (double)(U - n) / (top - i)) v[n++] = i;
        return v;
    };
    std::vector<int> dense_r(U), dense_w(U);
    for (int i = 0; i < U; ++i) { dense_r[i] = 2 * i; dense_w[i] = 200000 + i; }
    std::vector<int> sp_r = sparse(N - 2), sp_w = sparse(N - 2);
    int *d[4];
    const std::vector<int>* h[4] = {&dense_r, &dense_w, &sp_r, &sp_w};
    for (int i = 0; i < 4; ++i) { hipMalloc(&d[i], U * 4); hipMemcpy(d[i], h[i]->data(), U * 4, hipMemcpyHostToDevice); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; int r, w; } pat[4] = {{"dense reads, dense writes", 0, 1}, {"sparse reads, sparse writes", 2, 3},
                                                    {"sparse reads, dense writes", 2, 1}, {"dense reads, sparse writes", 0, 3}};
    for (int occ : {2, 8}) {
        const size_t lds = occ == 2 ? 72 * 1024 : 16 * 1024;  // 160 KB per CU: 2 or 8+ workgroups of 4 waves
        hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        for (int blocks : {64, 256}) {
            for (auto& p : pat) {
              for (int mode = 0; mode < 2; ++mode) {
                dim3 grid(blocks, C);
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), lds, 0, in, out, d[p.r], d[p.w], U, slab);
                    else hipLaunchKernelGGL(k<1>, grid, dim3(256), lds, 0, in, out, d[p.r], d[p.w], U, slab);
                };
                launch();
                hipEventRecord(e0);
                for (int i = 0; i < 5; ++i) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
                printf("%d waves/SIMD, %4d x %d blocks, %s: %-30s %.3f ms  %.2f TB/s\n", occ, blocks, C,
                       mode ? "next rows waited for before the stores" : "waited for at the head (after the stores)", p.name, ms,
                       (double)U * C * 1536 / (ms * 1e-3) / 1e12);
              }
            }
        }
    }
    return 0;
}
