// TEST INFRASTRUCTURE ONLY (never linked into or loaded by the product).
//
// oracle/_ref/libsdref_r2p.so: the reference's own result2profile pipeline for ONE centre sequence behind flat arrays --
// MultipleAlignment::computeMSA, MsaFilter::filter, PSSMCalculator::computePSSMFromMSA,
// SubstitutionMatrix::calcGlobalAaBiasCorrection, Masker::maskPssm and Profile::toBuffer are the reference's classes,
// compiled where they lie (oracle/Makefile; nothing copied); what is restated here is the 40-line driver around them,
// M/src/util/result2profile.cpp:123-277 (the module itself needs DBReader / Parameters.cpp -> cmake-generated headers).
#include "MsaFilter.h"
#include "MultipleAlignment.h"
#include "PSSMCalculator.h"
#include "Masker.h"
#include "Matcher.h"
#include "Sequence.h"
#include "SubstitutionMatrix.h"
#include "Parameters.h"
#include "Debug.h"

#include <algorithm>
#include <string>
#include <vector>

extern "C" {

// centre: ASCII letters (sequence DB entry) when centreProfile == NULL, else a profile DB entry of L * 25 bytes.
// edges: ASCII target sequences; alignment e = (qStart, tStart, expanded backtrace).  out: L * 25 bytes.
int ref_result2profile(const char *blosumOut, const char *centreSeq, const char *centreProfile, unsigned L, unsigned nEdges,
                       const char *const *edgeSeq, const unsigned *edgeLen, const int *qStart, const int *tStart,
                       const char *const *backtrace, float pca, float pcb, int wg, int filterMsa, float covMSAThr, const char *qid,
                       float qsc, float filterMaxSeqId, int Ndiff, int filterMinEnable, int compBiasCorr, int maskProfile,
                       float maskProb, char *out) {
    Debug::setDebugLevel(1);
    SubstitutionMatrix subMat(blosumOut, 2.0f, -0.2f);
    unsigned maxLen = L;
    for (unsigned e = 0; e < nEdges; e++) maxLen = std::max(maxLen, edgeLen[e]);
    const size_t maxSetSize = nEdges + 1;
    std::vector<int> qid_vec;
    {
        std::vector<std::string> v = Util::split(qid, ",");
        for (size_t i = 0; i < v.size(); i++) qid_vec.push_back(static_cast<int>((float) strtod(v[i].c_str(), NULL) * 100));
        std::sort(qid_vec.begin(), qid_vec.end());
    }
    Masker masker(subMat);
    MultipleAlignment aligner(maxLen, &subMat);
    MultiParam<PseudoCounts> mpca(PseudoCounts(pca, 1.4f)), mpcb(PseudoCounts(pcb, 5.8f));
    PSSMCalculator calculator(&subMat, maxLen, maxSetSize, Parameters::PCMODE_SUBSTITUTION_SCORE, mpca, mpcb);
    MsaFilter filter(maxLen, maxSetSize, &subMat, 11, 1);
    const int centreType = centreProfile ? Parameters::DBTYPE_HMM_PROFILE : Parameters::DBTYPE_AMINO_ACIDS;
    Sequence centerSequence(maxLen, centreType, &subMat, 0, false, compBiasCorr != 0);
    Sequence edgeSequence(maxLen, Parameters::DBTYPE_AMINO_ACIDS, &subMat, 0, false, false);
    centerSequence.mapSequence(0, 0, centreProfile ? centreProfile : centreSeq, L);
    std::vector<Matcher::result_t> alnResults;
    std::vector<std::vector<unsigned char> > seqSet;
    for (unsigned e = 0; e < nEdges; e++) {
        edgeSequence.mapSequence(e + 1, e + 1, edgeSeq[e], edgeLen[e]);
        seqSet.emplace_back(std::vector<unsigned char>(edgeSequence.numSequence, edgeSequence.numSequence + edgeSequence.L));
        Matcher::result_t r;
        r.dbKey = e + 1;
        r.qStartPos = qStart[e];
        r.dbStartPos = tStart[e];
        r.backtrace = backtrace[e];
        alnResults.push_back(r);
    }
    MultipleAlignment::MSAResult res = aligner.computeMSA(&centerSequence, seqSet, alnResults, true);
    size_t filteredSetSize = filterMsa ? filter.filter(res, alnResults, (int) (covMSAThr * 100), qid_vec, qsc, (int) (filterMaxSeqId * 100),
                                                       Ndiff, filterMinEnable)
                                       : res.setSize;
    PSSMCalculator::Profile pssmRes = calculator.computePSSMFromMSA(filteredSetSize, res.centerLength, (const char **) res.msaSequence, wg != 0, 0.0);
    std::vector<float> pNull(maxLen + 1);
    if (compBiasCorr) SubstitutionMatrix::calcGlobalAaBiasCorrection(&subMat, pssmRes.pssm, pNull.data(), Sequence::PROFILE_AA_SIZE, res.centerLength);
    if (maskProfile) masker.maskPssm(centerSequence, maskProb, pssmRes);
    std::string result;
    pssmRes.toBuffer(centerSequence, subMat, result);
    memcpy(out, result.data(), result.size());
    MultipleAlignment::deleteMSA(&res);
    return (int) result.size();
}

}  // extern "C"

// the reference's pseudo count matrix P(a|b) and background (for pinning the product's tables)
extern "C" void ref_r2p_tables(const char *blosumOut, float *R441, double *pBack21) {
    Debug::setDebugLevel(1);
    SubstitutionMatrix subMat(blosumOut, 2.0f, -0.2f);
    for (int i = 0; i < 21; i++) {
        pBack21[i] = subMat.pBack[i];
        for (int j = 0; j < 21; j++) R441[i * 21 + j] = subMat.subMatrixPseudoCounts[i][j];
    }
}
