// Pure write streams on one MI355X: bytes per lane (4 / 8 / 16), non-temporal or regular stores, grid sizes.
// Question (round 5): the top-down kernels' write stream runs at 5.4 - 5.6 TB/s with 16-byte stores (wr4.hip); the guide
// quotes 6.0 - 6.2 TB/s for dword-per-lane stores of random 2 304-byte rows -- does the store width move the ceiling?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int W, bool NT>
__global__ void __launch_bounds__(256) k(char* __restrict__ out, size_t bytes) {
    const size_t per_instr = (size_t)64 * W;            // bytes one wave instruction covers
    const size_t per_iter = per_instr * 8;               // eight instructions per iteration
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    for (size_t base = wave * per_iter; base + per_iter <= bytes; base += waves * per_iter) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            char* p = out + base + j * per_instr + (size_t)lane * W;
            if (W == 4) { float v = (float)j; if (NT) __builtin_nontemporal_store(v, (float*)p); else *(float*)p = v; }
            if (W == 8) { f2 v = {(float)j, 1.f}; if (NT) __builtin_nontemporal_store(v, (f2*)p); else *(f2*)p = v; }
            if (W == 16) { f4 v = {(float)j, 1.f, 2.f, 3.f}; if (NT) __builtin_nontemporal_store(v, (f4*)p); else *(f4*)p = v; }
        }
    }
}
template <int W, bool NT>
void run(char* a, size_t bytes, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<W, NT>), dim3(blocks), dim3(256), 0, 0, a, bytes);
    hipEventRecord(e0);
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<W, NT>), dim3(blocks), dim3(256), 0, 0, a, bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 4;
    printf("%2d B per lane, %-12s %6d blocks: %.3f ms  %.2f TB/s\n", W, NT ? "non-temporal" : "regular", blocks, ms, (double)bytes / (ms * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = (size_t)12 << 30;
    char* a; if (hipMalloc(&a, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int blocks : {1024, 2048, 8192, 32768}) {
        run<4, false>(a, bytes, blocks); run<4, true>(a, bytes, blocks);
        run<8, false>(a, bytes, blocks); run<8, true>(a, bytes, blocks);
        run<16, false>(a, bytes, blocks); run<16, true>(a, bytes, blocks);
    }
    // the runtime's own fill for comparison
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemsetAsync(a, 1, bytes, 0); hipEventRecord(e0);
    for (int i = 0; i < 4; ++i) hipMemsetAsync(a, 1, bytes, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 4;
    printf("hipMemsetAsync: %.3f ms  %.2f TB/s\n", ms, (double)bytes / (ms * 1e-3) / 1e12);
    return 0;
}
