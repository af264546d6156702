// Load-to-use latencies seen by one wave (pointer chase, 64 lanes in lockstep): data resident in HBM only, in L2,
// in the vector L1 (TCP), in LDS; and a workgroup barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const int* __restrict__ next, int start, int n, int* out, long long* cycles) {
    int p = start + threadIdx.x;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) p = next[p];
    long long t1 = wall_clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}
__global__ void chase_lds(int n, int* out, long long* cycles) {
    __shared__ int nx[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) nx[i] = (i * 67 + 13) & 4095;
    __syncthreads();
    int p = threadIdx.x;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) p = nx[p];
    long long t1 = wall_clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}
__global__ void barriers(int n, long long* cycles) {
    long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) __syncthreads();
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}
int main() {
    // wall_clock64 ticks at 100 MHz
    const size_t big = 1ull << 28;  // 1 GiB of ints: beyond L2 + MALL
    std::vector<int> h(big);
    // random cycle with stride of at least a cache line per hop
    {
        const size_t lines = big / 64;  // 256-byte granules
        std::vector<unsigned> perm(lines);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1);
        std::shuffle(perm.begin(), perm.end(), rng);
        for (size_t i = 0; i < lines; ++i) {
            const size_t a = (size_t)perm[i] * 64, b = (size_t)perm[(i + 1) % lines] * 64;
            for (int l = 0; l < 64; ++l) h[a + l] = (int)(b + l);
        }
    }
    int* d; hipMalloc(&d, big * 4); hipMemcpy(d, h.data(), big * 4, hipMemcpyHostToDevice);
    int* out; hipMalloc(&out, 4096); long long* cyc; hipMallocManaged(&cyc, 8);
    auto report = [&](const char* name, int n) { hipDeviceSynchronize(); printf("%-44s %8.1f ns per hop\n", name, cyc[0] * 10.0 / n); };
    hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0, 2000, out, cyc); report("HBM (1 GiB random lines)", 2000);
    // small working set: 2000 hops inside 64 KB -> after one pass resident in L2 (and L1 if it fits)
    {
        const int lines = 2048;  // 2048 granules x 256 B = 512 KB: L2, not L1 (32 KB)
        std::vector<int> s((size_t)lines * 64);
        std::vector<unsigned> perm(lines); std::iota(perm.begin(), perm.end(), 0u); std::mt19937 rng(2); std::shuffle(perm.begin(), perm.end(), rng);
        for (int i = 0; i < lines; ++i) for (int l = 0; l < 64; ++l) s[(size_t)perm[i] * 64 + l] = (int)((size_t)perm[(i + 1) % lines] * 64 + l);
        hipMemcpy(d, s.data(), s.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0, 4096, out, cyc); hipDeviceSynchronize();
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0, 4096, out, cyc); report("L2 (512 KB working set, second pass)", 4096);
    }
    {
        const int lines = 64;  // 16 KB: fits the vector L1
        std::vector<int> s((size_t)lines * 64);
        std::vector<unsigned> perm(lines); std::iota(perm.begin(), perm.end(), 0u); std::mt19937 rng(3); std::shuffle(perm.begin(), perm.end(), rng);
        for (int i = 0; i < lines; ++i) for (int l = 0; l < 64; ++l) s[(size_t)perm[i] * 64 + l] = (int)((size_t)perm[(i + 1) % lines] * 64 + l);
        hipMemcpy(d, s.data(), s.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0, 4096, out, cyc); report("vector L1 (16 KB working set)", 4096);
    }
    hipLaunchKernelGGL(chase_lds, dim3(1), dim3(64), 0, 0, 4096, out, cyc); report("LDS", 4096);
    hipLaunchKernelGGL(barriers, dim3(1), dim3(512), 0, 0, 4096, cyc); report("workgroup barrier (512 threads)", 4096);
    return 0;
}
