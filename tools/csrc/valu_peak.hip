// VALU issue micro-benchmark for gfx950 (tools/valu_peak.py drives it): how many lane-instructions per second the chip
// retires for the instruction classes the packed Smith-Waterman score kernel is made of.  Settles the denominator of
// bench.py's sw_valu.frac by measurement instead of a datasheet literal: every wavefront runs a long unrolled stream of
// one instruction over eight independent registers (no memory, no LDS), 32 wavefronts per CU.
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {

constexpr int UNROLL = 64;   // instructions per loop body (8 registers x 8)

#define OP8(INS)                                                                                                          \
    asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %8\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %8\n\t" INS " %4, %4, %8\n\t" \
                 INS " %5, %5, %8\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %8"                                          \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                      \
                 : "v"(k))

#define DPP8()                                                                                                            \
    asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                        \
                 "v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                        \
                 "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                        \
                 "v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                        \
                 "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                        \
                 "v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                        \
                 "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                        \
                 "v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf"                                            \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))

template <int KIND>
__global__ void __launch_bounds__(256) issue_kernel(int iters, uint32_t seed, uint32_t *__restrict__ out) {
    uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, r6 = r0 * 17u, r7 = r0 * 19u;
    const uint32_t k = seed ^ 0x00010001u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; u++) {
            if (KIND == 0) OP8("v_pk_max_i16");
            else if (KIND == 1) OP8("v_pk_add_i16");
            else if (KIND == 2) OP8("v_max_i32");
            else if (KIND == 3) OP8("v_add_u32");
            else if (KIND == 4) OP8("v_pk_sub_u16");
            else if (KIND == 5) DPP8();
            else if (KIND == 6) OP8("v_and_b32");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

// packed FP32 FMA (2 lanes' worth per lane) -- the instruction the 157 TF vector figure of the datasheet is made of
__global__ void __launch_bounds__(256) pkfma_kernel(int iters, float seed, float *__restrict__ out) {
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v a0 = {seed, seed + 1.f}, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    const float2v k = {1.0001f, 0.9999f}, c = {1e-7f, -1e-7f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; u++) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\t"
                         "v_pk_fma_f32 %3, %3, %8, %9\n\tv_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\t"
                         "v_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k), "v"(c));
        }
    }
    const float2v s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

}  // namespace

extern "C" {

// kind: 0 v_pk_max_i16, 1 v_pk_add_i16, 2 v_max_i32, 3 v_add_u32, 4 v_pk_sub_u16, 5 v_mov_b32 DPP row_shr:1, 6 v_and_b32,
// 7 v_pk_fma_f32.  Returns 0 and fills lane-instructions per second (64 per wavefront instruction), the kernel time,
// the CU count and the shader clock the runtime reports.
int valu_peak_run(int device, int kind, int iters, int blocksPerCu, double *laneInstrPerSec, double *ms, int *cus, double *clockGHz) {
    if (hipSetDevice(device) != hipSuccess) return 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 1;
    const int nCu = prop.multiProcessorCount;
    const int blocks = nCu * blocksPerCu;
    uint32_t *out = nullptr;
    if (hipMalloc(&out, (size_t) blocks * 256 * 4) != hipSuccess) return 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {   // first repetition warms up
        hipEventRecord(e0, 0);
        switch (kind) {
            case 0: hipLaunchKernelGGL(issue_kernel<0>, dim3(blocks), dim3(256), 0, 0, iters, 12345u, out); break;
            case 1: hipLaunchKernelGGL(issue_kernel<1>, dim3(blocks), dim3(256), 0, 0, iters, 12345u, out); break;
            case 2: hipLaunchKernelGGL(issue_kernel<2>, dim3(blocks), dim3(256), 0, 0, iters, 12345u, out); break;
            case 3: hipLaunchKernelGGL(issue_kernel<3>, dim3(blocks), dim3(256), 0, 0, iters, 12345u, out); break;
            case 4: hipLaunchKernelGGL(issue_kernel<4>, dim3(blocks), dim3(256), 0, 0, iters, 12345u, out); break;
            case 5: hipLaunchKernelGGL(issue_kernel<5>, dim3(blocks), dim3(256), 0, 0, iters, 12345u, out); break;
            case 6: hipLaunchKernelGGL(issue_kernel<6>, dim3(blocks), dim3(256), 0, 0, iters, 12345u, out); break;
            default: hipLaunchKernelGGL(pkfma_kernel, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, (float *) out); break;
        }
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return 3;
        float t = 0;
        hipEventElapsedTime(&t, e0, e1);
        if (rep > 0 && t < best) best = t;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(out);
    const double instr = (double) blocks * 256.0 * (double) iters * UNROLL;   // lane-instructions
    *laneInstrPerSec = instr / (best * 1e-3);
    *ms = best;
    *cus = nCu;
    *clockGHz = prop.clockRate * 1e-6;
    return 0;
}
}
