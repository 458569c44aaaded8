// tantan repeat masking, restated for the one configuration the prefilter uses
// (M/src/commons/Masker.cpp:20-32: maxCycleLength 50, repeatProb 0.005, repeatEndProb 0.05,
// repeatOffsetProbDecay 0.9, no gaps, minMaskProb 0.9, mask letter X).
// Algorithm: forward/backward HMM of M/lib/tantan/tantan.cpp:308-460 (the endGapProb == 0
// branches).  The summation order of the reference's AVX2 build is kept (4 running partial
// sums combined as (s0+s2)+(s1+s3), tantan.cpp:320-352 with mcf_simd.h:175-179) so the float
// posteriors agree bit for bit with that build.
#include "sd_host.h"

#include <cmath>
#include <vector>

namespace sd {

void initMaskCtx(const SubMat &m, MaskCtx &ctx) {
    // ProbabilityMatrix (M/src/commons/BaseMatrix.h:85-96)
    for (int i = 0; i < ALPH; i++)
        for (int j = 0; j < ALPH; j++) ctx.lr[i][j] = m.probMatrix[i][j] / (m.pBack[i] * m.pBack[j]);
}

namespace {
const int MAX_OFF = 50;
const int SCALE_STEP = 16;

double firstRepeatOffsetProb(double probMult, int maxRepeatOffset) {
    if (probMult < 1 || probMult > 1) return (1 - probMult) / (1 - std::pow(probMult, maxRepeatOffset));
    return 1.0 / maxRepeatOffset;
}
}  // namespace

int tantanMask(const MaskCtx &ctx, uint8_t *seq, int L, double minMaskProb) {
    if (L <= 0) return 0;
    const double repeatProb = 0.005, repeatEndProb = 0.05, decay = 0.9;
    const double b2b = 1 - repeatProb;
    const double f2b = repeatEndProb;
    const double f2f0 = 1 - repeatEndProb;
    const double b2fFirst = repeatProb * firstRepeatOffsetProb(decay, MAX_OFF);
    double b2f[MAX_OFF];
    {
        double p = b2fFirst;
        for (int i = 0; i < MAX_OFF; i++) {
            b2f[i] = p;
            p *= decay;
        }
    }
    double fg[MAX_OFF];
    for (int i = 0; i < MAX_OFF; i++) fg[i] = 0.0;
    double bg = 1.0;
    std::vector<double> scale(L / SCALE_STEP + 1, 0.0);
    std::vector<float> prob(L);

    // forward
    for (int pos = 0; pos < L; pos++) {
        const double *lr = ctx.lr[seq[pos]];
        const int maxOffset = pos < MAX_OFF ? pos : MAX_OFF;
        const double b = bg;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int i = 0;
        for (; i <= maxOffset - 4; i += 4) {
            double f0 = fg[i], f1 = fg[i + 1], f2 = fg[i + 2], f3 = fg[i + 3];
            s0 += f0; s1 += f1; s2 += f2; s3 += f3;
            fg[i]     = (b * b2f[i]     + f0 * f2f0) * lr[seq[pos - i - 1]];
            fg[i + 1] = (b * b2f[i + 1] + f1 * f2f0) * lr[seq[pos - i - 2]];
            fg[i + 2] = (b * b2f[i + 2] + f2 * f2f0) * lr[seq[pos - i - 3]];
            fg[i + 3] = (b * b2f[i + 3] + f3 * f2f0) * lr[seq[pos - i - 4]];
        }
        double fromFg = (s0 + s2) + (s1 + s3);
        for (; i < maxOffset; i++) {
            double f = fg[i];
            fromFg += f;
            fg[i] = (b * b2f[i] + f * f2f0) * lr[seq[pos - i - 1]];
        }
        bg = b * b2b + fromFg * f2b;
        if (pos % SCALE_STEP == SCALE_STEP - 1) {
            double sc = 1 / bg;
            scale[pos / SCALE_STEP] = sc;
            bg *= sc;
            for (int j = 0; j < MAX_OFF; j++) fg[j] *= sc;
        }
        prob[pos] = static_cast<float>(bg);
    }
    double z;
    {
        double fromFg = 0.0;
        for (int j = 0; j < MAX_OFF; j++) fromFg += fg[j];
        z = bg * b2b + fromFg * f2b;
    }
    // backward
    bg = b2b;
    for (int j = 0; j < MAX_OFF; j++) fg[j] = f2b;
    for (int pos = L - 1; pos >= 0; pos--) {
        double nonRepeat = prob[pos] * bg / z;
        prob[pos] = 1 - static_cast<float>(nonRepeat);
        if (pos % SCALE_STEP == SCALE_STEP - 1) {
            double sc = scale[pos / SCALE_STEP];
            bg *= sc;
            for (int j = 0; j < MAX_OFF; j++) fg[j] *= sc;
        }
        const double *lr = ctx.lr[seq[pos]];
        const int maxOffset = pos < MAX_OFF ? pos : MAX_OFF;
        const double toBg = f2b * bg;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int i = 0;
        for (; i <= maxOffset - 4; i += 4) {
            double f0 = fg[i] * lr[seq[pos - i - 1]];
            double f1 = fg[i + 1] * lr[seq[pos - i - 2]];
            double f2 = fg[i + 2] * lr[seq[pos - i - 3]];
            double f3 = fg[i + 3] * lr[seq[pos - i - 4]];
            s0 += b2f[i] * f0; s1 += b2f[i + 1] * f1; s2 += b2f[i + 2] * f2; s3 += b2f[i + 3] * f3;
            fg[i] = toBg + f2f0 * f0;
            fg[i + 1] = toBg + f2f0 * f1;
            fg[i + 2] = toBg + f2f0 * f2;
            fg[i + 3] = toBg + f2f0 * f3;
        }
        double toFg = (s0 + s2) + (s1 + s3);
        for (; i < maxOffset; i++) {
            double f = fg[i] * lr[seq[pos - i - 1]];
            toFg += b2f[i] * f;
            fg[i] = toBg + f2f0 * f;
        }
        bg = b2b * bg + toFg;
    }
    int masked = 0;
    for (int pos = 0; pos < L; pos++) {
        if (prob[pos] >= minMaskProb) {   // float vs double compare, tantan.cpp:527
            seq[pos] = X_CODE;
            masked++;
        }
    }
    return masked;
}

}  // namespace sd
