// Composition bias on the device (optional: sd_host_comp_bias is the host form of the same stage).
// SubstitutionMatrix::calcLocalAaBiasCorrection (M/src/commons/SubstitutionMatrix.cpp:79-109) per residue: an
// integer sum of matrix entries over a +-20 window, then a 21-step float/double accumulation that depends only on
// (residue, window length, sum).  The kernel forms the sums; the float tail is read from the table the host built
// with the reference's expression order (sd::biasTableFull), so every value is bit-identical to the host path by
// construction.  The three integer roundings (SW bias StripedSmithWaterman.cpp:1231-1235, diagonal bias
// UngappedAlignment.cpp:392-396, k-mer threshold bias QueryMatcher.cpp:230-240) follow in a second kernel.
#include "sd_common.h"

#include "../host/sd_host.h"

namespace {

#pragma clang fp contract(off)

__device__ __forceinline__ uint32_t seqOf(uint64_t g, const uint64_t *__restrict__ off, uint32_t n) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256)
bias_cb_kernel(uint64_t total, const uint8_t *__restrict__ res, const uint64_t *__restrict__ off, uint32_t n,
               const int16_t *__restrict__ matSeed, const int16_t *__restrict__ matBlosum, const float *__restrict__ tabSeed, int loSeed,
               int spanSeed, const float *__restrict__ tabBlosum, int loBlosum, int spanBlosum, float *__restrict__ cbSeed,
               float *__restrict__ cbBlosum) {
    __shared__ int16_t sSeed[441], sBlosum[441];
    for (int x = threadIdx.x; x < 441; x += 256) {
        sSeed[x] = matSeed[x];
        sBlosum[x] = matBlosum[x];
    }
    __syncthreads();
    const uint64_t g = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const uint32_t s = seqOf(g, off, n);
    const int N = (int) (off[s + 1] - off[s]);
    const int i = (int) (g - off[s]);
    const uint8_t *seq = res + off[s];
    const int minPos = max(0, i - 20), maxPos = min(N, i + 20);
    const int w = maxPos - minPos;
    const int r = seq[i];
    int sumS = 0, sumB = 0;
    for (int j = minPos; j < maxPos; j++) {
        const int c = seq[j];
        sumS += sSeed[r * 21 + c];
        sumB += sBlosum[r * 21 + c];
    }
    sumS -= sSeed[r * 21 + r];
    sumB -= sBlosum[r * 21 + r];
    cbSeed[g] = tabSeed[((size_t) r * 41 + w) * spanSeed + (sumS - loSeed)];
    cbBlosum[g] = tabBlosum[((size_t) r * 41 + w) * spanBlosum + (sumB - loBlosum)];
}

__global__ void __launch_bounds__(256)
bias_round_kernel(uint64_t total, const uint64_t *__restrict__ off, uint32_t n, const float *__restrict__ cbSeed,
                  const float *__restrict__ cbBlosum, int k, int span, int8_t *__restrict__ sw8, int8_t *__restrict__ dg8,
                  int16_t *__restrict__ km16) {
    __shared__ uint8_t seedPos[8];
    if (threadIdx.x < 8) {
        const uint8_t s6[8] = {0, 1, 3, 5, 8, 9, 0, 0}, s7[8] = {0, 1, 3, 5, 6, 9, 10, 0};
        seedPos[threadIdx.x] = k == 6 ? s6[threadIdx.x] : s7[threadIdx.x];
    }
    __syncthreads();
    const uint64_t g = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const uint32_t s = seqOf(g, off, n);
    const int N = (int) (off[s + 1] - off[s]);
    const int i = (int) (g - off[s]);
    {   // profile->composition_bias[i] = (int8_t) (cb < 0.0) ? cb - 0.5 : cb + 0.5   (a double expression)
        const float cb = cbBlosum[g];
        const double dv = (cb < 0.0) ? (double) cb - 0.5 : (double) cb + 0.5;
        sw8[g] = (int8_t) dv;
    }
    {   // float aaCorrBias = (a < 0.0) ? a/4 - 0.5 : a/4 + 0.5; aaCorrectionScore = static_cast<char>(aaCorrBias)
        const float a = cbSeed[g];
        const float r = (float) ((a < 0.0) ? (double) (a / 4) - 0.5 : (double) (a / 4) + 0.5);
        dg8[g] = (int8_t) (signed char) r;
    }
    int16_t kb = 0;
    if (i + span <= N) {   // biasCorrection += compositionBias[i + pos[p]] in seed order, then rounded half away from zero
        float b = 0;
        for (int p = 0; p < k; p++) b += cbSeed[g + seedPos[p]];
        kb = (int16_t) ((b < 0.0) ? (double) b - 0.5 : (double) b + 0.5);
    }
    km16[g] = kb;
}

}  // namespace

extern "C" int sd_comp_bias_batch(sd_ctx *ctx, sd_host *h, const uint8_t *residues, const uint64_t *offsets, uint32_t n, int kmerSize,
                                  int8_t *swBias, int8_t *diagBias, int16_t *kmerBias) {
    if (!ctx || !h || !residues || !offsets || !swBias || !diagBias || !kmerBias) return SD_EINVAL;
    if (kmerSize != 6 && kmerSize != 7) return SD_EINVAL;
    (void) hipSetDevice(ctx->device);
    sdD2HReset(ctx);   // (reads an earlier, failed call left pending)
    const uint64_t total = offsets[n];
    if (total == 0) return SD_OK;
    if (h->biasTabSeed.empty()) {   // once per host handle (a few 10^6 table entries)
        sd::biasTableFull(h->seed8, h->biasTabSeed, h->biasLoSeed, h->biasSpanSeed);
        sd::biasTableFull(h->blosum2, h->biasTabBlosum, h->biasLoBlosum, h->biasSpanBlosum);
    }
    float *dTabS = nullptr, *dTabB = nullptr, *dCbS = nullptr, *dCbB = nullptr;
    int16_t *dMatS = nullptr, *dMatB = nullptr, *dKm = nullptr;
    uint8_t *dRes = nullptr;
    uint64_t *dOff = nullptr;
    int8_t *dSw = nullptr, *dDg = nullptr;
    SD_HIP(ctx, wsGet(ctx, "bias.tabS", h->biasTabSeed.size(), &dTabS));
    SD_HIP(ctx, wsGet(ctx, "bias.tabB", h->biasTabBlosum.size(), &dTabB));
    SD_HIP(ctx, wsGet(ctx, "bias.matS", 441, &dMatS));
    SD_HIP(ctx, wsGet(ctx, "bias.matB", 441, &dMatB));
    if (!ctx->biasTablesUploaded) {
        std::vector<int16_t> ms(441), mb(441);
        for (int a = 0; a < 21; a++)
            for (int b = 0; b < 21; b++) {
                ms[a * 21 + b] = h->seed8.sub[a][b];
                mb[a * 21 + b] = h->blosum2.sub[a][b];
            }
        SD_HIP(ctx, hipMemcpy(dTabS, h->biasTabSeed.data(), h->biasTabSeed.size() * sizeof(float), hipMemcpyHostToDevice));
        SD_HIP(ctx, hipMemcpy(dTabB, h->biasTabBlosum.data(), h->biasTabBlosum.size() * sizeof(float), hipMemcpyHostToDevice));
        SD_HIP(ctx, hipMemcpy(dMatS, ms.data(), 441 * sizeof(int16_t), hipMemcpyHostToDevice));
        SD_HIP(ctx, hipMemcpy(dMatB, mb.data(), 441 * sizeof(int16_t), hipMemcpyHostToDevice));
        ctx->biasTablesUploaded = true;
    }
    SD_HIP(ctx, wsGet(ctx, "bias.res", total + 64, &dRes));
    SD_HIP(ctx, wsGet(ctx, "bias.off", (size_t) n + 1, &dOff));
    SD_HIP(ctx, wsGet(ctx, "bias.cbS", total + 64, &dCbS));
    SD_HIP(ctx, wsGet(ctx, "bias.cbB", total + 64, &dCbB));
    SD_HIP(ctx, wsGet(ctx, "bias.sw", total + 64, &dSw));
    SD_HIP(ctx, wsGet(ctx, "bias.dg", total + 64, &dDg));
    SD_HIP(ctx, wsGet(ctx, "bias.km", total + 64, &dKm));
    SD_HIP(ctx, hipMemcpyAsync(dRes, residues, total, hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipMemcpyAsync(dOff, offsets, ((size_t) n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    const unsigned grid = (unsigned) ((total + 255) / 256);
    const int span = kmerSize == 6 ? 10 : 11;
    {
        ProfScope ps(ctx, "comp_bias");
        hipLaunchKernelGGL(bias_cb_kernel, dim3(grid), dim3(256), 0, ctx->stream, total, dRes, dOff, n, dMatS, dMatB, dTabS, h->biasLoSeed,
                           h->biasSpanSeed, dTabB, h->biasLoBlosum, h->biasSpanBlosum, dCbS, dCbB);
        hipLaunchKernelGGL(bias_round_kernel, dim3(grid), dim3(256), 0, ctx->stream, total, dOff, n, dCbS, dCbB, kmerSize, span, dSw, dDg, dKm);
    }
    SD_HIP(ctx, hipGetLastError());
    SD_HIP(ctx, sdD2H(ctx, swBias, dSw, total));   // (the caller's arrays are pageable: see sdD2H)
    SD_HIP(ctx, sdD2H(ctx, diagBias, dDg, total));
    SD_HIP(ctx, sdD2H(ctx, kmerBias, dKm, total * sizeof(int16_t)));
    SD_HIP(ctx, sdStreamSync(ctx));
    return SD_OK;
}
