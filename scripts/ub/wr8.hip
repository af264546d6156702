// Write streams shaped like td_f81_super_kernel's (k = 64): per batch of 8 units a wave writes 4 KB of child rows, 8 KB of
// cherry rows and 16 KB of tip rows, into three regions of a column's slab (ids are level-ordered: every depth is one
// region).  CYCLIC: batch b goes to wave b mod n_waves (the kernel's grid-stride loop).  BLOCKED(m): a wave takes m
// consecutive batches (so its tip rows form a run of 16 m KB, ...).  32 columns = 32 slabs, blockIdx.y picks one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void burst(char* p, int kb, int lane, bool nt) {
    for (int j = 0; j < kb; ++j) {
        f4 v = {(float)j, 1.f, 2.f, 3.f};
        if (nt) __builtin_nontemporal_store(v, (f4*)(p + (size_t)j * 1024 + (size_t)lane * 16));
        else *(f4*)(p + (size_t)j * 1024 + (size_t)lane * 16) = v;
    }
}
// slab layout per column: [children: n_batches * 4 KB][cherries: n_batches * 8 KB][tips: n_batches * 16 KB]
__global__ void __launch_bounds__(256) k(char* __restrict__ out, int n_batches, int m, int nt, int major, size_t skew, size_t col_skew) {
    const size_t slab = (size_t)n_batches * 28 * 1024 + 3 * skew + col_skew;
    char* col = out + (size_t)blockIdx.y * slab;
    char* r0 = col;
    char* r1 = col + (size_t)n_batches * 4 * 1024 + skew;
    char* r2 = col + (size_t)n_batches * 12 * 1024 + 2 * skew;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), waves = gridDim.x * 4;
    for (int g = wave; (size_t)g * m < (size_t)n_batches; g += waves) {
        if (major) {   // region-major: the m batches' child rows as one burst, then their cherry rows, then their tip rows
            const size_t b = (size_t)g * m;
            burst(r0 + b * 4096, 4 * m, lane, nt);
            burst(r1 + b * 8192, 8 * m, lane, nt);
            burst(r2 + b * 16384, 16 * m, lane, nt);
        } else {
            for (int q = 0; q < m; ++q) {
                const size_t b = (size_t)g * m + q;
                if (b >= (size_t)n_batches) break;
                burst(r0 + b * 4096, 4, lane, nt);
                burst(r1 + b * 8192, 8, lane, nt);
                burst(r2 + b * 16384, 16, lane, nt);
            }
        }
    }
}
int main() {
    const int C = 32;
    const int n_batches = 16384;
    const size_t max_skew = (size_t)8 << 20;
    const size_t bytes = (size_t)C * ((size_t)n_batches * 28 * 1024 + 4 * max_skew);
    const size_t payload = (size_t)C * n_batches * 28 * 1024;
    char* a; if (hipMalloc(&a, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("three-region write stream (4 + 8 + 16 KB bursts, regions 64 MB and 192 MB into a column's slab), %.1f GB, 1024 x 32 blocks, m = 1\n", payload / 1e9);
    printf("skew = bytes added between the regions (r1 += skew, r2 += 2 skew); col = bytes added per column slab; TB/s regular / non-temporal\n");
    const size_t skews[] = {0, 256, 1024, 4096, 4096 + 256, 16384, 65536, 65536 + 4096, 262144, 1 << 20, (1 << 20) + 4096 + 256, 3 << 20};
    const size_t cols[] = {0, 4096 + 256, 65536 + 4096};
    for (size_t cs : cols)
        for (size_t sk : skews) {
            double r[2];
            for (int nt = 0; nt < 2; ++nt) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                hipLaunchKernelGGL(k, dim3(1024, C), dim3(256), 0, 0, a, n_batches, 1, nt, 0, sk, cs);
                (void)hipEventRecord(e0);
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(1024, C), dim3(256), 0, 0, a, n_batches, 1, nt, 0, sk, cs);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
                r[nt] = (double)payload / (ms * 1e-3) / 1e12;
            }
            printf("col %7zu  skew %8zu   %5.2f / %5.2f\n", cs, sk, r[0], r[1]);
        }
    return 0;
}
