// Write pattern of the small-k top-down sweep: one lane per unit writes 6 rows of 32 bytes (2 children: 64 B
// contiguous; 4 tips: 128 B contiguous) as 16-byte pieces.  A: pieces issued as they are computed (dummy ALU work of
// DELAY dependent FMAs between rows); B: all pieces of a unit back to back at the end; C: transposed through LDS so
// that every store instruction covers contiguous memory.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int MODE, int DELAY>
__global__ void __launch_bounds__(256) k(double* __restrict__ kids, double* __restrict__ tips, int n_units) {
    __shared__ double lds[4][64 * 24];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int u = blockIdx.x * 256 + threadIdx.x; u < n_units; u += gridDim.x * 256) {
        double x = u * 1e-9 + 1.0;
        double rows[6][4];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int d = 0; d < DELAY; ++d) x = __builtin_fma(x, 1.0000001, 1e-9);
#pragma unroll
            for (int s = 0; s < 4; ++s) rows[r][s] = x + s;
            if (MODE == 0) {
                double* p = r < 2 ? kids + ((size_t)u * 2 + r) * 4 : tips + ((size_t)u * 4 + (r - 2)) * 4;
                *reinterpret_cast<dbl2*>(p) = (dbl2){rows[r][0], rows[r][1]};
                *reinterpret_cast<dbl2*>(p + 2) = (dbl2){rows[r][2], rows[r][3]};
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double* p = r < 2 ? kids + ((size_t)u * 2 + r) * 4 : tips + ((size_t)u * 4 + (r - 2)) * 4;
                *reinterpret_cast<dbl2*>(p) = (dbl2){rows[r][0], rows[r][1]};
                *reinterpret_cast<dbl2*>(p + 2) = (dbl2){rows[r][2], rows[r][3]};
            }
        }
        if (MODE == 2) {
            // wave's units are consecutive: kids region = 64 units x 8 doubles, tips region = 64 x 16 doubles
            double* L = lds[wave];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int s = 0; s < 4; ++s) L[lane * 8 + r * 4 + s] = rows[r][s];
#pragma unroll
            for (int r = 2; r < 6; ++r)
#pragma unroll
                for (int s = 0; s < 4; ++s) L[512 + lane * 16 + (r - 2) * 4 + s] = rows[r][s];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const size_t u0 = (size_t)(u - lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // 512 doubles of kids: 4 x (64 lanes x 2 doubles)
                const int e = q * 128 + lane * 2;
                *reinterpret_cast<dbl2*>(kids + u0 * 8 + e) = (dbl2){L[e], L[e + 1]};
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = q * 128 + lane * 2;
                *reinterpret_cast<dbl2*>(tips + u0 * 16 + e) = (dbl2){L[512 + e], L[512 + e + 1]};
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}
template <int MODE, int DELAY>
void run(const char* name, double* a, double* b, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, DELAY>), dim3(8192), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, DELAY>), dim3(8192), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-40s delay %3d: %.3f ms  %.2f TB/s\n", name, DELAY, ms, (double)n * 192 / (ms * 1e-3) / 1e12);
}
int main() {
    const int n = 8 << 20;  // units: 8M x 192 B = 1.6 GB
    double *a, *b; hipMalloc(&a, (size_t)n * 64); hipMalloc(&b, (size_t)n * 128);
    run<0, 0>("A pieces as computed", a, b, n);
    run<0, 16>("A pieces as computed", a, b, n);
    run<0, 64>("A pieces as computed", a, b, n);
    run<1, 0>("B all pieces at the end", a, b, n);
    run<1, 64>("B all pieces at the end", a, b, n);
    run<2, 0>("C transposed through LDS", a, b, n);
    run<2, 64>("C transposed through LDS", a, b, n);
    return 0;
}
