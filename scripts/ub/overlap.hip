// Does FP64 MFMA overlap with VALU work of the same / other waves on a SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NM, int NV, int KIND>
__global__ void __launch_bounds__(256) k(double* out, int iters) {
    v4f64 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    int i0 = threadIdx.x, i1 = threadIdx.x * 3, i2 = 5, i3 = 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (KIND == 0) {  // f64 fma
                x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a);
                x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a); x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a);
            } else {  // 32-bit integer
                i0 = i0 * 3 + i1; i1 = i1 ^ (i0 >> 3); i2 = i2 * 5 + i3; i3 = i3 ^ (i2 >> 2);
                i0 = i0 + (i1 & 0xff); i1 = i1 | (i0 << 1); i2 = i2 + (i3 & 0x7f); i3 = i3 | (i2 << 2);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1 + i2 + i3;
}
template <int NM, int NV, int KIND>
void run(const char* name, double* d, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, KIND>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, KIND>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves = blocks*4/1024
    const double waves_per_simd = blocks * 4 / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / iters / waves_per_simd;
    printf("%-28s blocks %5d: %.3f ms, %.0f cycles (at 2.4 GHz) per iteration per wave  [%d mfma, %d valu]\n", name, blocks, ms, cyc, 2 * NM, 8 * NV);
}
int main() {
    double* d; hipMalloc(&d, 8192 * 256 * 8);
    for (int blocks : {256, 512, 1024}) {  // 1, 2, 4 waves per SIMD
        run<4, 0, 0>("mfma only", d, blocks);
        run<0, 8, 0>("f64 fma only", d, blocks);
        run<4, 8, 0>("mfma + f64 fma", d, blocks);
        run<0, 8, 1>("int only", d, blocks);
        run<4, 8, 1>("mfma + int", d, blocks);
    }
    return 0;
}
