// Cost of a hand-rolled grid barrier (all blocks resident, one atomic counter in L2) against the boundary between two
// dependent kernels replayed from a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) bar_kernel(unsigned* counter, int n, double* data) {
    const unsigned nb = gridDim.x;
    for (int i = 0; i < n; ++i) {
        data[blockIdx.x * 256 + threadIdx.x] += 1.0;  // some global traffic that must be visible across the barrier
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(i + 1) * nb;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) step_kernel(double* data) { data[blockIdx.x * 256 + threadIdx.x] += 1.0; }
int main() {
    unsigned* counter; hipMalloc(&counter, 4);
    double* data; hipMalloc(&data, 1024 * 256 * 8); hipMemset(data, 0, 1024 * 256 * 8);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 200;
    for (int blocks : {16, 64, 256, 512}) {
        hipMemsetAsync(counter, 0, 4, s);
        hipLaunchKernelGGL(bar_kernel, dim3(blocks), dim3(256), 0, s, counter, 5, data);
        hipMemsetAsync(counter, 0, 4, s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(bar_kernel, dim3(blocks), dim3(256), 0, s, counter, n, data);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid barrier, %3d blocks: %.2f us per barrier\n", blocks, ms * 1e3 / n);
        // the same as n dependent kernels in a graph
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(256), 0, s, data);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("graph of kernels, %3d blocks: %.2f us per kernel\n", blocks, ms * 1e3 / n);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
