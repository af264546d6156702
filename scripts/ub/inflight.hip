// How does the throughput of a load -> store loop depend on the bytes a CU keeps in flight?  A unit (8 lanes) reads NR
// 512-byte rows (ascending random subset of a slab: a height level over breadth-first ids) and writes one; a wave has the
// reads of ONE pass (8 units) in flight; the occupancy (waves per SIMD) is set by an LDS allocation.  32 slabs of 524 287
// rows, 34 485 units per slab.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int NR>
__global__ void __launch_bounds__(256) k(const double* __restrict__ in, double* __restrict__ out, const int* __restrict__ rd,
                                         const int* __restrict__ wr, int n_units, size_t slab) {
    extern __shared__ double pad[];
    const int lane = threadIdx.x & 63, g = lane & 7, sub = lane >> 3;
    const double* cin = in + blockIdx.y * slab;
    double* cout = out + blockIdx.y * slab;
    const int stride = gridDim.x * 4 * 8;
    for (int idx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + sub; idx < n_units; idx += stride) {
        const int r = rd[idx], w = wr[idx];
        dbl2 acc[4] = {{1, 1}, {1, 1}, {1, 1}, {1, 1}};
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] *= *reinterpret_cast<const dbl2*>(cin + (size_t)(r + j) * 64 + (q * 8 + g) * 2);
        double* o = cout + (size_t)w * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<dbl2*>(o + (q * 8 + g) * 2) = acc[q];
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
        int n = 0;
        for (int i = 0; i < top && n < U; ++i)
            if ((double)rand() / RAND_MAX < (double)(U - n) / (top - i)) v[n++] = i;
        return v;
    };
    std::vector<int> sp_r = sparse(N - 4), sp_w = sparse(N - 4);
    int *dr, *dw;
    hipMalloc(&dr, U * 4); hipMalloc(&dw, U * 4);
    hipMemcpy(dr, sp_r.data(), U * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, sp_w.data(), U * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int nr : {1, 2, 4})
        for (int occ : {1, 2, 3, 4, 8}) {
            const size_t lds = (size_t)(150 * 1024 / occ) & ~1023;
            dim3 grid(256, C);
            auto launch = [&]() {
                if (nr == 1) hipLaunchKernelGGL(k<1>, grid, dim3(256), lds, 0, in, out, dr, dw, U, slab);
                if (nr == 2) hipLaunchKernelGGL(k<2>, grid, dim3(256), lds, 0, in, out, dr, dw, U, slab);
                if (nr == 4) hipLaunchKernelGGL(k<4>, grid, dim3(256), lds, 0, in, out, dr, dw, U, slab);
            };
            launch();
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            const double bytes = (double)U * C * 512 * (nr + 1);
            const double passes_per_slot = (double)U * C / 8 / (256.0 * 4 * occ);
            printf("%d rows read per unit, %d waves/SIMD: %.3f ms  %.2f TB/s  (%.1f KB of reads in flight per wave, %.2f us per pass)\n",
                   nr, occ, ms, bytes / (ms * 1e-3) / 1e12, nr * 4.0, ms * 1e3 / passes_per_slot);
        }
    return 0;
}
