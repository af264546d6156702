// How long does the host wait after the last kernel of a short chain?  hipStreamSynchronize against a spin on a word in
// pinned host memory that the last kernel writes (system-scope store behind a fence).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void work(double* x, int n) {
    double a = x[threadIdx.x];
    for (int i = 0; i < n; ++i) a = a * 1.0000001 + 1e-9;
    x[threadIdx.x] = a;
}
__global__ void flag(volatile unsigned long long* f, unsigned long long seq) {
    __threadfence_system();
    *f = seq;
}
int main() {
    double* d; hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    unsigned long long* h; hipHostMalloc((void**)&h, 64, hipHostMallocDefault);
    *h = 0;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int n : {100, 20000}) {
        for (int mode = 0; mode < 2; ++mode) {
            double best = 1e9, sum = 0;
            for (int rep = 0; rep < 200; ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                for (int kq = 0; kq < 4; ++kq) hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, d, n);
                if (mode == 0) {
                    hipStreamSynchronize(s);
                } else {
                    const unsigned long long seq = rep + 1 + 1000ull * mode + 100000ull * n;
                    hipLaunchKernelGGL(flag, dim3(1), dim3(1), 0, s, (volatile unsigned long long*)h, seq);
                    while (*(volatile unsigned long long*)h != seq) {}
                }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep >= 20) { best = us < best ? us : best; sum += us; }
            }
            printf("4 kernels of %5d iterations + %s: mean %.1f us, best %.1f us\n", n, mode ? "spin on a pinned flag  " : "hipStreamSynchronize   ", sum / 180, best);
        }
    }
    hipStreamSynchronize(s);
    return 0;
}
