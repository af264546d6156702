// Round 6: would a posterior table whose rows are NOT in breadth-first order lift td_f81_super_kernel's write ceiling?
// The kernel's store pattern exactly (k = 64, 8 lanes x 8 states per unit, 8 units per wave): a unit = one child of a two-level
// node writes 7 rows of 512 B -- the child, cherry 0, its two tips, cherry 1, its two tips -- each row as 4 store instructions
// of 16 B per lane (a unit's 8 lanes cover 128 B; the wave's instruction covers 8 pieces of 128 B, one per unit).  Per column
// 262 144 units (1 048 576 tips), 32 columns, grid 1024 x 32 blocks of 256 threads, grid-stride over batches of 8 units: the
// launch geometry of the cfg4 step.  What varies is WHERE the 7 rows of a unit live in the column's slab:
//   layout 0  breadth-first (today): three regions -- child rows, cherry rows, tip rows (ids level-ordered)
//   layout 1  unit-major: the 7 rows of a unit contiguous (3 584 B), units in order (a wave's 8 units = 28 KB contiguous)
//   layout 2  batch-major: per batch of 8 units 28 KB: [8 child rows][16 cherry rows][32 tip rows]
//   layout 3  unit-major padded to 4 KB per unit (8 units = 32 KB, aligned)
//   layout 4  batch-major padded to 32 KB per batch
//   layout 6 / 7  batch-major with the rows of a kind together (8 child rows, 8 first cherries, ...), every store instruction 1 KB
//             contiguous (what an LDS transpose of the wave's rows would give); 7: padded to 32 KB per batch
//   layout 5  family-major: two sibling units = 14 rows = 7 KB: [2 children][4 cherries][8 tips] (breadth-first inside a family)
// `rd`: every unit also reads its parent's 512-B row (shared by the two siblings), as the kernel does (11 % of its bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void row(char* p, int g, d2 v, int nt) {   // one 512-B row by the unit's 8 lanes: 4 instructions
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        d2* a = (d2*)(p + q * 128 + g * 16);
        if (nt) __builtin_nontemporal_store(v, a); else *a = v;
    }
}

__global__ void __launch_bounds__(256) k(char* __restrict__ out, const char* __restrict__ parents, int n_units, int layout, int nt, int rd,
                                         size_t slab) {
    char* col = out + (size_t)blockIdx.y * slab;
    const char* pcol = parents + (size_t)blockIdx.y * ((size_t)n_units / 2 * 512);
    const int lane = threadIdx.x & 63, g = lane & 7, sub = lane >> 3;
    const int stride = gridDim.x * 4 * 8;
    for (int idx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + sub; idx < n_units; idx += stride) {
        d2 v = {(double)idx, 1.0};
        if (rd) {
            const d2* pr = (const d2*)(pcol + (size_t)(idx >> 1) * 512 + g * 16);
            d2 acc = pr[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) { d2 t = pr[q * 8]; acc.x += t.x; acc.y += t.y; }
            v.x += acc.x; v.y += acc.y;
        }
        char *pc, *pg[2], *pt[4];
        const size_t u = (size_t)idx, b = u >> 3, s = u & 7;
        if (layout == 0) {
            char* r0 = col;
            char* r1 = col + (size_t)n_units * 512;
            char* r2 = col + (size_t)n_units * 1536;
            pc = r0 + u * 512;
            for (int j = 0; j < 2; ++j) pg[j] = r1 + (2 * u + j) * 512;
            for (int j = 0; j < 4; ++j) pt[j] = r2 + (4 * u + j) * 512;
        } else if (layout == 1 || layout == 3) {
            char* base = col + u * (layout == 1 ? 3584 : 4096);
            pc = base;
            pg[0] = base + 512; pt[0] = base + 1024; pt[1] = base + 1536;
            pg[1] = base + 2048; pt[2] = base + 2560; pt[3] = base + 3072;
        } else if (layout == 2 || layout == 4) {
            char* base = col + b * (layout == 2 ? 28672 : 32768);
            pc = base + s * 512;
            for (int j = 0; j < 2; ++j) pg[j] = base + 4096 + (2 * s + j) * 512;
            for (int j = 0; j < 4; ++j) pt[j] = base + 12288 + (4 * s + j) * 512;
        } else if (layout >= 6) {
            // kind-major inside the batch, written through an (imagined) LDS transpose: every store instruction covers 1 KB contiguous
            char* base = col + b * (layout == 6 ? 28672 : 32768);
            d2 w = v;
#pragma unroll
            for (int kind = 0; kind < 7; ++kind)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    d2* a = (d2*)(base + kind * 4096 + q * 1024 + lane * 16);
                    if (nt) __builtin_nontemporal_store(w, a); else *a = w;
                }
            continue;
        } else {
            char* base = col + (u >> 1) * 7168;
            const size_t j2 = u & 1;
            pc = base + j2 * 512;
            for (int j = 0; j < 2; ++j) pg[j] = base + 1024 + (2 * j2 + j) * 512;
            for (int j = 0; j < 4; ++j) pt[j] = base + 3072 + (4 * j2 + j) * 512;
        }
        row(pc, g, v, nt);
        row(pg[0], g, v, nt); row(pt[0], g, v, nt); row(pt[1], g, v, nt);
        row(pg[1], g, v, nt); row(pt[2], g, v, nt); row(pt[3], g, v, nt);
    }
}

int main() {
    const int C = 32, n_units = 262144;
    const size_t slab_max = (size_t)n_units * 4096;
    char *a, *p;
    if (hipMalloc(&a, (size_t)C * slab_max) != hipSuccess) { printf("alloc failed\n"); return 1; }
    if (hipMalloc(&p, (size_t)C * (n_units / 2) * 512) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(p, 0, (size_t)C * (n_units / 2) * 512);
    const double wbytes = (double)C * n_units * 3584, rbytes = (double)C * (n_units / 2) * 512;
    printf("td_f81_super_kernel's store pattern, %d units x %d columns: %.2f GB written (+ %.2f GB read with rd)\n", n_units, C, wbytes / 1e9, rbytes / 1e9);
    printf("TB/s of the bytes moved; columns: regular / non-temporal stores; blocks along x\n");
    const char* names[] = {"0 breadth-first, three regions", "1 unit-major 3584 B", "2 batch-major 28 KB", "3 unit-major padded 4 KB",
                           "4 batch-major padded 32 KB", "5 family-major 7 KB",
                           "6 kind-major 28 KB, 1 KB / instr", "7 kind-major padded 32 KB, 1 KB / instr"};
    const int blocks[] = {512, 1024, 2048, 8192};
    for (int rd = 0; rd < 2; ++rd) {
        printf("--- %s\n", rd ? "with the parent-row reads" : "writes only");
        printf("%-34s", "layout \\ blocks");
        for (int bx : blocks) printf("   %5d x %d  ", bx, C);
        printf("\n");
        for (int layout = 0; layout < 8; ++layout) {
            const size_t slab = layout == 3 ? (size_t)n_units * 4096 : (layout == 4 || layout == 7) ? (size_t)(n_units / 8) * 32768 : (size_t)n_units * 3584;
            printf("%-34s", names[layout]);
            for (int bx : blocks) {
                double r[2];
                for (int nt = 0; nt < 2; ++nt) {
                    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                    hipLaunchKernelGGL(k, dim3(bx, C), dim3(256), 0, 0, a, p, n_units, layout, nt, rd, slab);
                    (void)hipEventRecord(e0);
                    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(bx, C), dim3(256), 0, 0, a, p, n_units, layout, nt, rd, slab);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
                    r[nt] = (wbytes + (rd ? rbytes : 0)) / (ms * 1e-3) / 1e12;
                    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
                }
                printf("  %5.2f /%5.2f  ", r[0], r[1]);
            }
            printf("\n");
        }
    }
    return 0;
}
