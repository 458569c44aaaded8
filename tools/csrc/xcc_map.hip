#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(1024) k(int *out, long long cycles) {
    extern __shared__ int lds[];
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    const long long t0 = wall_clock64();
    int v = threadIdx.x;
    while (wall_clock64() - t0 < cycles) v = v * 1664525 + 1013904223;
    if (v == 42) lds[threadIdx.x] = v;
    if (threadIdx.x == 0) out[blockIdx.x] = (int) (x & 0xF);
}
int main() {
    int *d; hipMalloc(&d, 4096 * 4);
    for (int wgs : {256, 512, 1024}) {
        hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), 49216, 0, d, 20000LL);
        std::vector<int> h(wgs);
        hipMemcpy(h.data(), d, wgs * 4, hipMemcpyDeviceToHost);
        int match = 0; int cnt[16] = {0};
        for (int i = 0; i < wgs; i++) { match += (h[i] == (i & 7)); cnt[h[i]]++; }
        printf("%d WGs: xcc == blockIdx & 7 for %d; per xcc:", wgs, match);
        for (int x = 0; x < 8; x++) printf(" %d", cnt[x]);
        printf("  first 24:");
        for (int i = 0; i < 24; i++) printf(" %d", h[i]);
        printf("\n");
    }
}
