// tools/csrc/fetch_calib.hip -- measurement tool (not part of the product): known byte counts in the access patterns of the
// prefilter kernels, so that rocprofv3's FETCH_SIZE / WRITE_SIZE (and the per-size request counters TCC_EA0_RDREQ_32B / _64B /
// _128B, TCC_EA0_WRREQ / _64B) can be calibrated per pattern instead of applying the guide's x2 (measured there for 16-B/lane
// coalesced streams only, MI355X_MICROARCH.md "HBM") to every kernel.  Run under rocprofv3 --pmc by tools/fetch_calib.sh;
// tools/fetch_calib.py reads the counter CSVs.  Buffers are 4 GiB (16x the Infinity Cache) so that nothing is served on-die.
//   cal_read16 / cal_read8 / cal_read4   coalesced streaming reads, 16 / 8 / 4 B per lane           requested = N * width
//   cal_gather8                          one random 8-B read per lane (the join's table / entry lookups, diagOf)
//   cal_gather8_run16                    random runs of 16 consecutive 8-B elements (= one 128-B line; short index lists)
//   cal_write16 / cal_write8             coalesced streaming stores
//   cal_scatter8                         one random 8-B store per lane (join_scatter's worst case)
//   cal_scatter8_run8                    random runs of 8 consecutive 8-B stores (64 B: join_scatter's per-query runs)
//   cal_scatter8_nt                      cal_scatter8 with non-temporal stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__global__ void __launch_bounds__(256) cal_read16(const uint4 *in, uint64_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256) {
        const uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) cal_read8(const uint2 *in, uint64_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256) {
        const uint2 v = in[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) cal_read4(const uint32_t *in, uint64_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256) acc ^= in[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
// nReq random elements (run consecutive ones each) out of nElem
__global__ void __launch_bounds__(256) cal_gather8(const uint2 *in, uint64_t nElem, uint64_t nReq, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < nReq; i += (uint64_t) gridDim.x * 256) {
        const uint2 v = in[mix(i) % nElem];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) cal_gather8_run16(const uint2 *in, uint64_t nElem, uint64_t nReq, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < nReq; i += (uint64_t) gridDim.x * 256) {
        const uint64_t run = mix(i >> 4) % (nElem >> 4);
        const uint2 v = in[(run << 4) + (i & 15)];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) cal_write16(uint4 *out, uint64_t n) {
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256)
        out[i] = make_uint4((uint32_t) i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) cal_write8(uint2 *out, uint64_t n) {
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256)
        out[i] = make_uint2((uint32_t) i, 1u);
}
__global__ void __launch_bounds__(256) cal_scatter8(uint2 *out, uint64_t nElem, uint64_t nReq) {
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < nReq; i += (uint64_t) gridDim.x * 256)
        out[mix(i) % nElem] = make_uint2((uint32_t) i, 1u);
}
__global__ void __launch_bounds__(256) cal_scatter8_nt(unsigned long long *out, uint64_t nElem, uint64_t nReq) {
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < nReq; i += (uint64_t) gridDim.x * 256)
        __builtin_nontemporal_store((unsigned long long) i, out + mix(i) % nElem);
}
__global__ void __launch_bounds__(256) cal_scatter8_run8(uint2 *out, uint64_t nElem, uint64_t nReq) {
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < nReq; i += (uint64_t) gridDim.x * 256) {
        const uint64_t run = mix(i >> 3) % (nElem >> 3);
        out[(run << 3) + (i & 7)] = make_uint2((uint32_t) i, 1u);
    }
}

int main() {
    const uint64_t BYTES = 4ull << 30;
    void *buf;
    uint32_t *sink;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, BYTES));
    CK(hipDeviceSynchronize());
    const uint64_t nReq = 1ull << 27;   // random accesses: 1 GiB requested
    const dim3 g(256 * 16), b(256);
    // every pattern twice (the first launch of a kernel pays its code fetch)
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(cal_read16, g, b, 0, 0, (const uint4 *) buf, BYTES / 16, sink);
        hipLaunchKernelGGL(cal_read8, g, b, 0, 0, (const uint2 *) buf, BYTES / 8, sink);
        hipLaunchKernelGGL(cal_read4, g, b, 0, 0, (const uint32_t *) buf, BYTES / 4, sink);
        hipLaunchKernelGGL(cal_gather8, g, b, 0, 0, (const uint2 *) buf, BYTES / 8, nReq, sink);
        hipLaunchKernelGGL(cal_gather8_run16, g, b, 0, 0, (const uint2 *) buf, BYTES / 8, nReq, sink);
        hipLaunchKernelGGL(cal_write16, g, b, 0, 0, (uint4 *) buf, BYTES / 16);
        hipLaunchKernelGGL(cal_write8, g, b, 0, 0, (uint2 *) buf, BYTES / 8);
        hipLaunchKernelGGL(cal_scatter8, g, b, 0, 0, (uint2 *) buf, BYTES / 8, nReq);
        hipLaunchKernelGGL(cal_scatter8_nt, g, b, 0, 0, (unsigned long long *) buf, BYTES / 8, nReq);
        hipLaunchKernelGGL(cal_scatter8_run8, g, b, 0, 0, (uint2 *) buf, BYTES / 8, nReq);
        CK(hipDeviceSynchronize());
    }
    // requested bytes per launch, for tools/fetch_calib.py
    printf("{\"cal_read16\": %llu, \"cal_read8\": %llu, \"cal_read4\": %llu, \"cal_gather8\": %llu, \"cal_gather8_run16\": %llu, "
           "\"cal_write16\": %llu, \"cal_write8\": %llu, \"cal_scatter8\": %llu, \"cal_scatter8_nt\": %llu, \"cal_scatter8_run8\": %llu}\n",
           (unsigned long long) BYTES, (unsigned long long) BYTES, (unsigned long long) BYTES, (unsigned long long) nReq * 8,
           (unsigned long long) nReq * 8, (unsigned long long) BYTES, (unsigned long long) BYTES, (unsigned long long) nReq * 8,
           (unsigned long long) nReq * 8, (unsigned long long) nReq * 8);
    return 0;
}
