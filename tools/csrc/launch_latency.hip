// Micro-benchmark behind DESIGN 5: what a chain of short dependent launches on one stream costs while other streams keep the device
// full of long-lived one-wavefront workgroups (the shape of the alignment stage's score kernels).
// usage: launch_latency [wgMicroseconds] [ldsBytes] [nLoadStreams] [tinyThreads] [tinyLds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(64) busy_kernel(long long cycles, int *sink) {
    extern __shared__ int lds[];
    const long long t0 = wall_clock64();
    int v = threadIdx.x;
    while (wall_clock64() - t0 < cycles) v = v * 1664525 + 1013904223;
    if (v == 42) lds[threadIdx.x] = v;
    if (v == 43) *sink = lds[0];
}
__global__ void tiny_kernel(int *p) {
    extern __shared__ int lds[];
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
    if (p[1] == 12345) lds[threadIdx.x] = 1;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const double wgUs = argc > 1 ? atof(argv[1]) : 400.0;
    const int ldsBytes = argc > 2 ? atoi(argv[2]) : 9000;
    const int nLoad = argc > 3 ? atoi(argv[3]) : 2;
    const int tinyThreads = argc > 4 ? atoi(argv[4]) : 256;
    const int tinyLds = argc > 5 ? atoi(argv[5]) : 0;
    int *d = nullptr;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    hipStream_t sB;
    hipStreamCreateWithFlags(&sB, hipStreamNonBlocking);
    std::vector<hipStream_t> sA(nLoad);
    for (auto &s : sA) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const long long cycles = (long long) (wgUs * 100.0);   // wall_clock64: 100 MHz
    auto chain = [&](int n) {
        const double t0 = now();
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(tinyThreads), tinyLds, sB, d);
        hipStreamSynchronize(sB);
        return (now() - t0) / n * 1e6;
    };
    chain(50);
    printf("idle device: %.1f us per dependent launch (%d threads, %d B LDS)\n", chain(400), tinyThreads, tinyLds);
    // load: each stream gets kernels of 100 000 one-wavefront workgroups
    for (int rep = 0; rep < 20; rep++)
        for (auto &s : sA) hipLaunchKernelGGL(busy_kernel, dim3(100000), dim3(64), ldsBytes, s, cycles, d + 8);
    const double us = chain(400);
    printf("beside %d stream(s) of %.0f-us workgroups with %d B LDS: %.1f us per dependent launch\n", nLoad, wgUs, ldsBytes, us);
    for (auto &s : sA) hipStreamSynchronize(s);
    return 0;
}
