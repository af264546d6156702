// Read pattern of the small-k top-down sweep: one lane per unit reads rows of 32 bytes as 16-byte pieces (parent
// posterior: 1 row per unit, rows 32 B apart -> pieces at 32-byte stride; children vectors: 2 rows per unit, 64 B
// contiguous per lane -> pieces at 64-byte stride).  A: as the kernel does; C: the wave reads the same bytes with
// every load instruction covering 1 KB and passes them through LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(const double* __restrict__ par, const double* __restrict__ kids, double* __restrict__ out, int n_units) {
    __shared__ double lds[4][64 * 12];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int u = blockIdx.x * 256 + threadIdx.x; u < n_units; u += gridDim.x * 256) {
        double r[12];
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { dbl2 v = *reinterpret_cast<const dbl2*>(par + (size_t)u * 4 + q * 2); r[q * 2] = v.x; r[q * 2 + 1] = v.y; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { dbl2 v = *reinterpret_cast<const dbl2*>(kids + (size_t)u * 8 + q * 2); r[4 + q * 2] = v.x; r[5 + q * 2] = v.y; }
        } else {
            double* L = lds[wave];
            const size_t u0 = (size_t)(u - lane);
#pragma unroll
            for (int q = 0; q < 2; ++q) { const int e = q * 128 + lane * 2; dbl2 v = *reinterpret_cast<const dbl2*>(par + u0 * 4 + e); L[e] = v.x; L[e + 1] = v.y; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int e = q * 128 + lane * 2; dbl2 v = *reinterpret_cast<const dbl2*>(kids + u0 * 8 + e); L[256 + e] = v.x; L[256 + e + 1] = v.y; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int s = 0; s < 4; ++s) r[s] = L[lane * 4 + s];
#pragma unroll
            for (int s = 0; s < 8; ++s) r[4 + s] = L[256 + lane * 8 + s];
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int s = 0; s < 12; ++s) acc += r[s];
    }
    if (acc == 123.456) out[0] = acc;
}
template <int MODE>
void run(const char* name, double* a, double* b, double* o, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(8192), dim3(256), 0, 0, a, b, o, n);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE>), dim3(8192), dim3(256), 0, 0, a, b, o, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, (double)n * 96 / (ms * 1e-3) / 1e12);
}
int main() {
    const int n = 32 << 20;  // units: 32M x 96 B = 3.2 GB
    double *a, *b, *o; hipMalloc(&a, (size_t)n * 32); hipMalloc(&b, (size_t)n * 64); hipMalloc(&o, 64);
    hipMemset(a, 0, (size_t)n * 32); hipMemset(b, 0, (size_t)n * 64);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("A 16-byte pieces per lane (as the kernel)", a, b, o, n);
        run<1>("C contiguous 1 KB per instruction via LDS", a, b, o, n);
    }
    return 0;
}
