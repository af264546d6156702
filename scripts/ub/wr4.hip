// k = 64 top-down write shapes: 8 lanes per unit, 512-byte rows.  P1: every store instruction writes 128 contiguous
// bytes per unit, 8 units of a wave 3 KB apart (what td_f81_kernel<8,8> does for its 2 cherries + 4 tips per unit);
// P2: the same bytes, every instruction 1 KB contiguous.  Non-temporal stores, as in the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int MODE, bool NT>
__global__ void __launch_bounds__(256) k(double* __restrict__ out, int n_units) {
    const int lane = threadIdx.x & 63;
    const int unit_in_wave = lane >> 3, g = lane & 7;
    const int waves = gridDim.x * 4;
    for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w * 8 < n_units; w += waves) {
        double* base = out + (size_t)w * 8 * 6 * 64;  // 8 units x 6 rows x 64 doubles
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dbl2 v = {(double)r, (double)q};
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(base + (unit_in_wave * 6 + r) * 64 + (q * 8 + g) * 2)); else *reinterpret_cast<dbl2*>(base + (unit_in_wave * 6 + r) * 64 + (q * 8 + g) * 2) = v;
                }
        } else {
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                dbl2 v = {(double)j, 1.0};
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(base + (j * 64 + lane) * 2)); else *reinterpret_cast<dbl2*>(base + (j * 64 + lane) * 2) = v;
            }
        }
    }
}
template <int MODE, bool NT>
void run(const char* name, double* a, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(8192), dim3(256), 0, 0, a, n);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, NT>), dim3(8192), dim3(256), 0, 0, a, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, (double)n * 3072 / (ms * 1e-3) / 1e12);
}
int main() {
    const int n = 4 << 20;  // units: 4M x 3 KB = 12.9 GB
    double* a; hipMalloc(&a, (size_t)n * 3072);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, true>("P1 128 B per unit, non-temporal", a, n);
        run<0, false>("P1 128 B per unit, regular stores", a, n);
        run<1, true>("P2 1 KB contiguous, non-temporal", a, n);
        run<1, false>("P2 1 KB contiguous, regular stores", a, n);
    }
    return 0;
}
