// Write streams, 16 bytes per lane: how long a run of consecutive 1 KB store instructions a wave writes (J), how many
// blocks, non-temporal or not -- looking for what moves a pure write stream between 5.4 and 6.2 TB/s (wr5.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) k(char* __restrict__ out, size_t bytes, int J) {
    const size_t per_iter = (size_t)1024 * J;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    for (size_t base = wave * per_iter; base + per_iter <= bytes; base += waves * per_iter) {
        for (int j = 0; j < J; ++j) {
            char* p = out + base + (size_t)j * 1024 + (size_t)lane * 16;
            f4 v = {(float)j, 1.f, 2.f, 3.f};
            if (NT) __builtin_nontemporal_store(v, (f4*)p); else *(f4*)p = v;
        }
    }
}
template <bool NT>
double run(char* a, size_t bytes, int blocks, int J) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NT>), dim3(blocks), dim3(256), 0, 0, a, bytes, J);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<NT>), dim3(blocks), dim3(256), 0, 0, a, bytes, J);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 4;
    return (double)bytes / (ms * 1e-3) / 1e12;
}
int main() {
    const size_t bytes = (size_t)12 << 30;
    char* a; if (hipMalloc(&a, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("TB/s, regular / non-temporal; rows: 1 KB store instructions per wave and iteration; columns: blocks of 256 threads\n");
    const int blocks[] = {1024, 2048, 4096, 8192, 16384, 32768, 65536};
    printf("%6s", "J");
    for (int b : blocks) printf("  %11d", b);
    printf("\n");
    for (int J : {1, 2, 4, 8, 16, 24, 32, 64}) {
        printf("%6d", J);
        for (int b : blocks) printf("  %5.2f/%5.2f", run<false>(a, bytes, b, J), run<true>(a, bytes, b, J));
        printf("\n");
    }
    return 0;
}
