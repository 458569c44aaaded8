// Host restatements of the small glue modules the clustersearch workflow runs between `align` and `clusterhits`, and
// after it (R/data/clustersearch.sh:121-151; SURVEY.md section 7 step 7: the GPU box has no reference binary to run
// them), plus a FASTA -> setDB writer in the createsetdb layout (R/data/createsetdb.sh:119-180):
//   prefixid            M/src/util/prefixid.cpp:12-82
//   besthitbyset        R/src/util/besthitbyset.cpp:41-144 over R/src/util/Aggregation.cpp:26-98
//   mergeresultsbyset   M/src/util/mergeresultsbyset.cpp:49-70
//   combinehits         R/src/util/combinehits.cpp:74-234 over Aggregation.cpp:100-165, M/src/multihit/combinepvalperset.cpp:12-27
//   summarizeresults    R/src/util/SummarizeResults.cpp:61-117
// Text in, text out; the double arithmetic keeps the reference's expression order because its %.3E output is what the next
// module parses.  (The fused in-process form of the same chain is sd_agg_* in libsdgpu.so.)
#include "sd_cli.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace sdcli {

namespace {

std::vector<std::string> splitTabs(const std::string &line) {
    std::vector<std::string> out;
    size_t p = 0;
    for (;;) {
        const size_t e = line.find('\t', p);
        if (e == std::string::npos) {
            out.push_back(line.substr(p));
            break;
        }
        out.push_back(line.substr(p, e - p));
        p = e + 1;
    }
    return out;
}

std::string fmt3E(double v) {
    char b[40];
    snprintf(b, sizeof(b), "%.3E", v);
    return b;
}

// Aggregation::buildMap (Aggregation.cpp:26-49): lines of one entry grouped by the set of their target column -- the
// reference's std::map<set key, vector of column vectors>, here without a copy of anything: lines and columns are views into
// the entry, `order` lists the lines by ascending set key, input order inside a set
struct EntryLines {
    struct Line {
        const char *s, *e;      // the line without its newline
        const char *c2b, *c2e;  // column 2 (the P-value column the modules rewrite); NULL when the line has fewer columns
        const char *c4b;        // start of column 4 (the E-value); NULL when absent
        uint32_t nCols;
        unsigned setKey;
    };
    std::vector<Line> lines;
    std::vector<const char *> lineStart;
    std::vector<uint32_t> order;
    // strtod of a column (an empty column is 0, as it is for the reference's std::string columns)
    static double num(const char *b, const char *lineEnd) { return (!b || b == lineEnd || *b == '\t') ? 0.0 : strtod(b, nullptr); }
    // the line with column 2 replaced (every other column as it is)
    static void appendWith(std::string &to, const Line &l, const std::string &col2) {
        if (!l.c2b) {
            to.append(l.s, l.e - l.s);
            return;
        }
        to.append(l.s, l.c2b - l.s);
        to.append(col2);
        to.append(l.c2e, l.e - l.c2e);
    }
};
bool buildMap(const char *data, const sddb::Reader &memberToSet, EntryLines &out, std::string *err) {
    out.lineStart.clear();
    while (*data != '\0') {
        const char *s = data;
        const char *nl = strchr(data, '\n');
        data = nl ? nl + 1 : s + strlen(s);
        if ((nl ? nl : data) != s) out.lineStart.push_back(s);
    }
    const size_t n = out.lineStart.size();
    out.lines.resize(n);
    bool failed = false;
    std::string firstErr;
#pragma omp parallel for schedule(static) if (n > 20000)
    for (size_t x = 0; x < n; x++) {
        EntryLines::Line &l = out.lines[x];
        l.s = out.lineStart[x];
        l.c2b = l.c2e = l.c4b = nullptr;
        const char *c1b = nullptr;
        uint32_t cols = 1;
        const char *c = l.s;
        for (; *c != '\n' && *c != '\0'; c++)
            if (*c == '\t') {
                if (cols == 1) c1b = c + 1;
                else if (cols == 2) l.c2b = c + 1;
                else if (cols == 3) l.c2e = c;
                if (cols == 4) l.c4b = c + 1;   // (cols counts the columns seen so far: this tab ends column 3)
                cols++;
            }
        l.e = c;
        if (l.c2b && !l.c2e) l.c2e = l.e;
        l.nCols = cols;
        std::string e;
        if (cols < 2) e = "Invalid result record \"" + std::string(l.s, l.e - l.s) + "\"";
        else {
            const char *k1 = l.c2b ? l.c2b - 1 : l.e;
            const unsigned tKey = c1b == k1 ? 0u : (unsigned) strtoul(c1b, nullptr, 10);
            const size_t id = memberToSet.idOfKey(tKey);
            if (id == SIZE_MAX) e = "Invalid target database key " + std::string(c1b, k1 - c1b) + ".";
            else l.setKey = (unsigned) strtoul(memberToSet.data(id), nullptr, 10);
        }
        if (!e.empty()) {
#pragma omp critical(sd_glue_map)
            if (!failed) {
                failed = true;
                firstErr = e;
            }
        }
    }
    if (failed) {
        if (err) *err = firstErr;
        return false;
    }
    out.order.resize(n);
    for (uint32_t i = 0; i < n; i++) out.order[i] = i;
    std::stable_sort(out.order.begin(), out.order.end(), [&](uint32_t x, uint32_t y) { return out.lines[x].setKey < out.lines[y].setKey; });
    return true;
}

// entries [first, first + n) through `work` on all threads, their outputs handed to `emit` in entry order
template <typename Out, typename Work, typename Emit>
bool forEntriesInOrder(size_t nEntries, Work work, Emit emit, std::string *err) {
    const size_t block = 2048;
    std::vector<Out> outs(block);
    std::string firstErr;
    bool failed = false;
    for (size_t b0 = 0; b0 < nEntries && !failed; b0 += block) {
        const size_t b1 = std::min(nEntries, b0 + block);
#pragma omp parallel
        {
            EntryLines m;
#pragma omp for schedule(dynamic, 8)
            for (size_t i = b0; i < b1; i++) {
                std::string e;
                if (!work(i, m, outs[i - b0], &e)) {
#pragma omp critical(sd_glue_err)
                    if (!failed) {
                        failed = true;
                        firstErr = e;
                    }
                }
            }
        }
        if (failed) break;
        for (size_t i = b0; i < b1; i++)
            if (!emit(i, outs[i - b0])) {
                if (err) *err = "";
                return false;
            }
    }
    if (failed) {
        if (err) *err = firstErr;
        return false;
    }
    return true;
}

double computeLogPval(double eval, double logCalibration) {   // besthitbyset.cpp:10-20
    if (eval == 0) return log(DBL_MIN) - logCalibration;
    else if (eval > 0 && eval < 10e-4) return log(eval) - logCalibration;
    else return log(1 - exp(-eval)) - logCalibration;
}

unsigned sizeOfSet(const sddb::Reader &sz, unsigned setKey, bool *ok) {
    const size_t id = sz.idOfKey(setKey);
    if (id == SIZE_MAX) {
        *ok = false;
        return 0;
    }
    return (unsigned) strtoul(sz.data(id), nullptr, 10);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
int prefixidModule(const Args &a) {
    if (a.pos.size() != 2) return fail("usage: prefixid <resultDB> <resultDB> [--tsv]");
    std::string err;
    sddb::Reader in;
    if (!in.open(a.pos[0], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, &err)) return fail(err);
    if (a.has("--mapping-file") && !a.str("--mapping-file", "").empty()) return fail("--mapping-file is not supported");
    const bool tsv = a.flag("--tsv", false);
    const std::string user = a.str("--prefix", "");
    sddb::Writer out;
    FILE *flat = nullptr;
    if (tsv) {
        sddb::removeDb(a.pos[1]);
        flat = fopen(a.pos[1].c_str(), "wb");
        if (!flat) return fail("cannot create " + a.pos[1]);
    } else if (!out.open(a.pos[1], in.dbtype(), &err)) {
        return fail(err);
    }
    std::string buf;
    for (size_t i = 0; i < in.size(); i++) {
        const unsigned key = in.key(i);
        const std::string pre = user.empty() ? std::to_string(key) : user;
        buf.clear();
        const char *d = in.data(i);
        while (*d != '\0') {   // std::getline over the entry
            const char *s = d;
            while (*d != '\n' && *d != '\0') d++;
            buf.append(pre).append("\t").append(s, d - s).append("\n");
            if (*d == '\n') d++;
        }
        if (tsv) fwrite(buf.data(), 1, buf.size(), flat);
        else if (!out.write(key, buf.data(), buf.size())) return fail("cannot write " + a.pos[1]);
    }
    if (tsv) fclose(flat);
    else if (!out.close(&err)) return fail(err);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
int besthitbysetModule(const Args &a) {
    if (a.pos.size() != 4) return fail("usage: besthitbyset <querySetDB> <targetSetDB> <resultDB> <outDB>");
    const bool simple = a.flag("--simple-best-hit", true);
    const int subopt = (int) a.integer("--suboptimal-hits", 0);
    std::string err;
    sddb::Reader memberToSet, setSize, in;
    if (!memberToSet.open(a.pos[1] + "_member_to_set", sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) return fail(err);
    if (!setSize.open(a.pos[1] + "_set_size", sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) return fail(err);
    if (!in.open(a.pos[2], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, &err)) return fail(err);
    sddb::Writer out;
    if (!out.open(a.pos[3], sddb::DBTYPE_ALIGNMENT_RES, &err)) return fail(err);
    auto work = [&](size_t i, EntryLines &m, std::string &buffer, std::string *e) {
        if (!buildMap(in.data(i), memberToSet, m, e)) return false;
        buffer.clear();
        for (size_t g0 = 0; g0 < m.order.size();) {
            size_t g1 = g0;
            const unsigned setKey = m.lines[m.order[g0]].setKey;
            while (g1 < m.order.size() && m.lines[m.order[g1]].setKey == setKey) g1++;
            const size_t nRows = g1 - g0;
            bool ok = true;
            const unsigned nbrGenes = sizeOfSet(setSize, setKey, &ok);
            if (!ok) {
                *e = "Invalid target size database key " + std::to_string(setKey) + ".";
                return false;
            }
            (void) nbrGenes;
            double bestScore = -DBL_MAX, secondBestScore = -DBL_MAX, bestEval = DBL_MAX;
            const double logCal = log(1);
            const EntryLines::Line *best = nullptr;
            for (size_t r = g0; r < g1; r++) {
                const EntryLines::Line &row = m.lines[m.order[r]];
                if (row.nCols < 5) {
                    *e = "Invalid alignment result record";
                    return false;
                }
                const double eval = EntryLines::num(row.c4b, row.e);
                const double score = std::min(DBL_MAX, -log(eval));
                if (simple || nRows < 2) {
                    if (eval < bestEval) {
                        bestEval = eval;
                        best = &row;
                    }
                } else if (score >= bestScore) {
                    secondBestScore = bestScore;
                    bestScore = score;
                    best = &row;
                } else if (score > secondBestScore) {
                    secondBestScore = score;
                }
            }
            std::vector<const EntryLines::Line *> all;
            std::vector<double> evals, logP;
            if (subopt > 0 && simple && nRows > 1) {
                const double thr = bestEval * subopt;
                for (size_t r = g0; r < g1; r++) {
                    const EntryLines::Line &row = m.lines[m.order[r]];
                    const double eval = EntryLines::num(row.c4b, row.e);
                    if (eval <= thr) {
                        all.push_back(&row);
                        evals.push_back(eval);
                    }
                }
            } else {
                all.push_back(best);
            }
            if (all.size() > 1) {
                for (size_t r = 0; r < all.size(); r++) logP.push_back(computeLogPval(evals[r], logCal));
            } else if (simple || nRows < 2) {
                logP.push_back(computeLogPval(bestEval, logCal));
            } else {
                logP.push_back(secondBestScore - bestScore);
            }
            if (best != nullptr) {
                for (size_t j = 0; j < all.size(); j++) {
                    EntryLines::appendWith(buffer, *all[j], fmt3E(logP[j]));
                    if (j + 1 != all.size()) buffer.push_back('\n');
                }
            }
            buffer.push_back('\n');
            g0 = g1;
        }
        return true;
    };
    auto emit = [&](size_t i, const std::string &buffer) { return out.write(in.key(i), buffer.data(), buffer.size()); };
    Lap lap("besthitbyset");
    if (!forEntriesInOrder<std::string>(in.size(), work, emit, &err)) return fail(err.empty() ? "cannot write " + a.pos[3] : err);
    lap.mark("entries");
    if (!out.close(&err)) return fail(err);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
int mergeresultsbysetModule(const Args &a) {
    if (a.pos.size() != 3) return fail("usage: mergeresultsbyset <setDB> <resultDB> <outDB>");
    std::string err;
    sddb::Reader sets, res;
    if (!sets.open(a.pos[0], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, &err)) return fail(err);
    if (!res.open(a.pos[1], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) return fail(err);
    sddb::Writer out;
    if (!out.open(a.pos[2], sddb::withExtended(res.dbtype(), sddb::EXT_INDEX_NEED_SRC), &err)) return fail(err);
    std::string buffer;
    for (size_t i = 0; i < sets.size(); i++) {
        buffer.clear();
        const char *d = sets.data(i);
        while (*d != '\0') {
            const unsigned key = (unsigned) strtoul(d, nullptr, 10);
            const size_t id = res.idOfKey(key);
            if (id == SIZE_MAX) return fail("Invalid key " + std::to_string(key) + " in entry " + std::to_string(i) + ".");
            buffer.append(res.data(id));
            while (*d != '\n' && *d != '\0') d++;
            if (*d == '\n') d++;
        }
        if (!out.write(sets.key(i), buffer.data(), buffer.size())) return fail("cannot write " + a.pos[2]);
    }
    if (!out.close(&err)) return fail(err);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
int combinehitsModule(const Args &a) {
    if (a.pos.size() != 5) return fail("usage: combinehits <querySetDB> <targetSetDB> <resultDB> <outDB> <tmpDir>");
    const double alpha = (double) (float) a.real("--alpha", 1.0);   // PvalueAggregator takes a float
    const int mode = (int) a.integer("--aggregation-mode", 0);
    const bool filterSelf = a.flag("--filter-self-match", false);
    std::string err;
    sddb::Reader memberToSet, qSize, tSize, in;
    if (!memberToSet.open(a.pos[1] + "_member_to_set", sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) return fail(err);
    if (!qSize.open(a.pos[0] + "_set_size", sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) return fail(err);
    if (!tSize.open(a.pos[1] + "_set_size", sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) return fail(err);
    if (!in.open(a.pos[2], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, &err)) return fail(err);
    unsigned maxOrf = 0;
    for (size_t i = 0; i < qSize.size(); i++) maxOrf = std::max(maxOrf, (unsigned) strtoul(qSize.data(i), nullptr, 10));
    std::vector<double> lGamma((size_t) maxOrf + 2), logB((size_t) std::max(maxOrf, 1u));
    for (size_t i = 0; i < lGamma.size(); i++) lGamma[i] = lgamma((double) i);
    const size_t numTargetSets = tSize.size();
    sddb::Writer out, outH;
    if (!out.open(a.pos[3], sddb::DBTYPE_ALIGNMENT_RES, &err)) return fail(err);
    if (!outH.open(a.pos[3] + "_h", sddb::DBTYPE_GENERIC_DB, &err)) return fail(err);
    Lap lap("combinehits");
    unsigned matchIdx = 0;
    EntryLines m;
    std::vector<std::pair<size_t, size_t> > groups;
    std::vector<std::string> headers, bodies;
    for (size_t i = 0; i < in.size(); i++) {
        // one entry per query set (mergeresultsbyset): its lines parsed once, its target sets worked on by all threads
        const unsigned qSetKey = in.key(i);
        if (!buildMap(in.data(i), memberToSet, m, &err)) return fail(err);
        bool okQ = true;
        const unsigned orfCount = sizeOfSet(qSize, qSetKey, &okQ);
        if (!okQ) return fail("Invalid query size database key " + std::to_string(qSetKey) + ".");
        {   // precomputeLogB (combinepvalperset.cpp:17-27) with pvalThreshold = alpha / (orfCount + 1)
            const double thr = alpha / (orfCount + 1);
            const double logThr = log(thr), log1m = log(1 - thr);
            if (orfCount > 0) {
                logB[orfCount - 1] = orfCount * logThr;
                for (int x = (int) orfCount - 2; x >= 0; x--) {
                    const int k = x + 1;
                    const double lbin = lGamma[orfCount + 1] - lGamma[orfCount - k + 1] - lGamma[k + 1];
                    const double newTerm = lbin + k * logThr + (orfCount - k) * log1m;
                    logB[x] = logB[x + 1] + log(1 + exp(newTerm - logB[x + 1]));
                }
            }
        }
        groups.clear();
        for (size_t g0 = 0, g1 = 0; g0 < m.order.size(); g0 = g1) {
            const unsigned tSetKey = m.lines[m.order[g0]].setKey;
            g1 = g0;
            while (g1 < m.order.size() && m.lines[m.order[g1]].setKey == tSetKey) g1++;
            groups.push_back(std::make_pair(g0, g1));
        }
        headers.assign(groups.size(), std::string());
        bodies.assign(groups.size(), std::string());
        bool failed = false;
        std::string firstErr;
        // one (query set, target set) match; an empty header = no match written
        auto work = [&](size_t g, std::string &header, std::string &body, std::string *e) {
            const size_t g0 = groups[g].first, g1 = groups[g].second, nRows = g1 - g0;
            const unsigned tSetKey = m.lines[m.order[g0]].setKey;
            bool ok = true;
            std::vector<const EntryLines::Line *> entries;
            header.clear();
            auto none = [&]() {
                header.clear();
                return true;
            };
            if (filterSelf && qSetKey == tSetKey) return none();
            const unsigned tOrf = sizeOfSet(tSize, tSetKey, &ok);
            if (!ok) {
                *e = "Invalid target size database key " + std::to_string(tSetKey) + ".";
                return false;
            }
            header = std::to_string(qSetKey) + "\t" + std::to_string(tSetKey) + "\t" + std::to_string(orfCount) + "\t" +
                     std::to_string(tOrf) + "\t";
            entries.clear();
            if (mode == 0) {   // AGGREGATION_MODE_MULTIHIT (combinehits.cpp:97-153)
                const double pvalThreshold = 10e-7;
                size_t k = 0;
                double r = 0;
                const double logPvalThr = log(pvalThreshold);
                for (size_t x = g0; x < g1; x++) {
                    const EntryLines::Line &row = m.lines[m.order[x]];
                    if (row.nCols < 3) {
                        *e = "Invalid alignment result record";
                        return false;
                    }
                    const double lp = EntryLines::num(row.c2b, row.e);
                    if (lp < logPvalThr) {
                        k++;
                        r -= lp - logPvalThr;
                        entries.push_back(&row);
                    }
                }
                if (r == 0 || k == 0) return none();
                header += std::to_string(k) + "\t";
                const double expMinusR = exp(-r);
                if (std::isinf(r) || expMinusR == 0) {
                    header += fmt3E(0.0);
                } else {
                    double truncatedFisher = 0;
                    const double logR = log(r);
                    for (size_t x = 0; x < orfCount; x++) truncatedFisher += exp(x * logR - lGamma[x + 1] + logB[x]);
                    const double updatedPval = expMinusR * truncatedFisher;
                    const double updatedEval = updatedPval * numTargetSets;
                    header += fmt3E(updatedEval);
                }
            } else if (mode == 2) {   // AGGREGATION_MODE_PRODUCT
                if (nRows == 0) return none();
                double sum = 0;
                for (size_t x = g0; x < g1; x++) {
                    const EntryLines::Line &row = m.lines[m.order[x]];
                    if (row.nCols < 3) {
                        *e = "Invalid alignment result record";
                        return false;
                    }
                    sum += EntryLines::num(row.c2b, row.e);
                    entries.push_back(&row);
                }
                header += std::to_string(nRows) + "\t" + fmt3E(exp(sum) * numTargetSets);
            } else if (mode == 3) {   // AGGREGATION_MODE_TRUNCATED_PRODUCT: header only (combinehits.cpp:183-199)
                const double thr = log(alpha / (orfCount + 1));
                double sum = 0;
                size_t k = 0;
                for (size_t x = g0; x < g1; x++) {
                    const EntryLines::Line &row = m.lines[m.order[x]];
                    if (row.nCols < 3) {
                        *e = "Invalid alignment result record";
                        return false;
                    }
                    const double lp = EntryLines::num(row.c2b, row.e);
                    if (lp < thr) {
                        sum += lp;
                        k++;
                    }
                }
                if (k == 0) return none();
                header += std::to_string(k) + "\t" + fmt3E(exp(sum));
            } else {
                *e = "Invalid aggregation function!";
                return false;
            }
            header.push_back('\n');
            body.clear();
            for (size_t j = 0; j < entries.size(); j++) {
                EntryLines::appendWith(body, *entries[j], fmt3E(exp(EntryLines::num(entries[j]->c2b, entries[j]->e))));
                if (j + 1 != entries.size()) body.push_back('\n');
            }
            body.push_back('\n');
            return true;
        };
#pragma omp parallel for schedule(dynamic, 4)
        for (size_t g = 0; g < groups.size(); g++) {
            std::string e;
            if (!work(g, headers[g], bodies[g], &e)) {
#pragma omp critical(sd_glue_err)
                if (!failed) {
                    failed = true;
                    firstErr = e;
                }
            }
        }
        if (failed) return fail(firstErr);
        for (size_t g = 0; g < groups.size(); g++) {
            if (headers[g].empty()) continue;
            if (!outH.write(matchIdx, headers[g].data(), headers[g].size()) || !out.write(matchIdx, bodies[g].data(), bodies[g].size()))
                return fail("cannot write " + a.pos[3]);
            matchIdx++;
        }
    }
    lap.mark("entries");
    if (!out.close(&err) || !outH.close(&err)) return fail(err);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
int summarizeresultsModule(const Args &a) {
    if (a.pos.size() != 4) return fail("usage: summarizeresults <querySetDB> <targetSetDB> <clustersDB> <out.tsv>");
    std::string err;
    Lap lap("summarizeresults");
    const std::shared_ptr<const SetInfo> qsP = loadSetInfo(a.pos[0], true, &err);
    if (!qsP) return fail(err);
    const bool sameDb = a.pos[0] == a.pos[1];
    const std::shared_ptr<const SetInfo> tsP = sameDb ? qsP : loadSetInfo(a.pos[1], true, &err);
    if (!tsP) return fail(err);
    const SetInfo &qs = *qsP, &ts = *tsP;
    lap.mark("set info");
    sddb::Reader hdr, aln;
    if (!hdr.open(a.pos[2] + "_h", sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, &err)) return fail(err);
    if (!aln.open(a.pos[2], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) return fail(err);
    const bool dbOut = a.flag("--db-output", false);
    sddb::Writer out;
    FILE *flat = nullptr;
    if (dbOut) {
        if (!out.open(a.pos[3], sddb::DBTYPE_GENERIC_DB, &err)) return fail(err);
    } else {
        sddb::removeDb(a.pos[3]);
        flat = fopen(a.pos[3].c_str(), "wb");
        if (!flat) return fail("cannot create " + a.pos[3]);
    }
    // a block of clusters formatted on all threads, written in order
    std::vector<std::string> texts;
    const size_t block = 16384;
    for (size_t b0 = 0; b0 < hdr.size(); b0 += block) {
        const size_t b1 = std::min(hdr.size(), b0 + block);
        texts.resize(b1 - b0);
        bool failed = false;
        std::string firstErr;
#pragma omp parallel for schedule(dynamic, 64)
        for (size_t id = b0; id < b1; id++) {
            std::string &buffer = texts[id - b0];
            buffer.clear();
            std::string e;
            const unsigned matchKey = hdr.key(id);
            const size_t alnId = aln.idOfKey(matchKey);
            if (alnId == SIZE_MAX) e = "cluster " + std::to_string(matchKey) + " has a header but no entry";
            const char *h = hdr.data(id);
            while (e.empty() && *h != '\0') {
                const char *s = h;
                while (*h != '\n' && *h != '\0') h++;
                std::vector<std::string> cols = splitTabs(std::string(s, h - s));
                if (*h == '\n') h++;
                if (cols.size() < 5) {
                    e = "Invalid alignment result record";
                    break;
                }
                const unsigned qset = (unsigned) strtoul(cols[0].c_str(), nullptr, 10);
                const unsigned tset = (unsigned) strtoul(cols[1].c_str(), nullptr, 10);
                buffer.append("#").append(std::to_string(matchKey)).append("\t");
                buffer.append(qset < qs.sourceOfSet.size() ? qs.sourceOfSet[qset] : std::string()).append("\t");
                buffer.append(tset < ts.sourceOfSet.size() ? ts.sourceOfSet[tset] : std::string()).append("\t");
                buffer.append(cols[2]).append("\t").append(cols[3]).append("\t").append(cols[4]).append("\n");
                const char *d = aln.data(alnId);
                while (*d != '\0') {
                    const char *ls = d;
                    while (*d != '\n' && *d != '\0') d++;
                    if (*d == '\n') d++;
                    // first two columns are keys -> lookup names; the rest of the line is copied
                    char *e2;
                    const unsigned long qid = strtoul(ls, &e2, 10);
                    const unsigned long tid = strtoul(e2, &e2, 10);
                    int tabs = 0;
                    for (const char *c = ls; c < d; c++) tabs += (*c == '\t');
                    if (tabs < 9) {
                        e = "Invalid alignment result record";
                        break;
                    }
                    if (qid >= qs.nameOfKey.size() || tid >= ts.nameOfKey.size()) {
                        e = "alignment key without lookup entry";
                        break;
                    }
                    const char *rest = e2;
                    while (rest < d && (*rest == '\t' || *rest == ' ')) rest++;
                    buffer.append(">").append(qs.nameOfKey[qid]).append("\t").append(ts.nameOfKey[tid]).append("\t").append(rest, d - rest);
                }
            }
            if (!e.empty()) {
#pragma omp critical(sd_glue_err)
                if (!failed) {
                    failed = true;
                    firstErr = e;
                }
            }
        }
        if (failed) return fail(firstErr);
        for (size_t id = b0; id < b1; id++) {
            const std::string &buffer = texts[id - b0];
            if (dbOut) {
                if (!out.write(hdr.key(id), buffer.data(), buffer.size())) return fail("cannot write " + a.pos[3]);
            } else {
                fwrite(buffer.data(), 1, buffer.size(), flat);
            }
        }
    }
    lap.mark("entries");
    if (dbOut) {
        if (!out.close(&err)) return fail(err);
    } else {
        fclose(flat);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// FASTA (Prodigal headers) -> the DB family createsetdb leaves behind for amino acid input
// (R/src/workflow/createsetdb.cpp:14-18 forces --shuffle 0; R/data/createsetdb.sh:103-180): NAME, NAME_h, NAME.lookup,
// NAME.source, NAME_member_to_set, NAME_set_to_member, NAME_set_size.  One input file = one set.
int createsetdbModule(const Args &a) {
    if (a.pos.size() < 3) return fail("usage: createsetdb <fasta> [<fasta> ...] <setDB> <tmpDir>");
    const std::string outDb = a.pos[a.pos.size() - 2];
    std::string err;
    sddb::Writer seqW, hdrW;
    if (!seqW.open(outDb, sddb::DBTYPE_AMINO_ACIDS, &err)) return fail(err);
    if (!hdrW.open(outDb + "_h", sddb::DBTYPE_GENERIC_DB, &err)) return fail(err);
    FILE *lookup = fopen((outDb + ".lookup").c_str(), "wb");
    FILE *source = fopen((outDb + ".source").c_str(), "wb");
    if (!lookup || !source) return fail("cannot create " + outDb + ".lookup / .source");
    uint32_t key = 0;
    std::vector<uint32_t> setOf;
    std::vector<uint32_t> setCount;
    for (size_t f = 0; f + 2 < a.pos.size(); f++) {
        std::ifstream in(a.pos[f]);
        if (!in) return fail("cannot open " + a.pos[f]);
        const size_t slash = a.pos[f].find_last_of('/');
        fprintf(source, "%zu\t%s\n", f, a.pos[f].substr(slash == std::string::npos ? 0 : slash + 1).c_str());
        std::string line, header, seq;
        uint32_t inSet = 0;
        bool have = false;
        auto flush = [&]() -> bool {
            if (!have) return true;
            seq.push_back('\n');
            std::string h = header + "\n";
            if (!seqW.write(key, seq.data(), seq.size()) || !hdrW.write(key, h.data(), h.size())) return false;
            // lookup name: header without blanks, split at '#': accession, start, end, strand (createsetdb.sh:119-131)
            std::string nb;
            for (char c : header)
                if (c != ' ') nb.push_back(c);
            std::vector<std::string> fld;
            size_t p = 0;
            for (;;) {
                const size_t e = nb.find('#', p);
                if (e == std::string::npos) {
                    fld.push_back(nb.substr(p));
                    break;
                }
                fld.push_back(nb.substr(p, e - p));
                p = e + 1;
            }
            std::string start = fld.size() > 1 ? fld[1] : "", end = fld.size() > 2 ? fld[2] : "";
            if (fld.size() > 3 && fld[3] == "-1") std::swap(start, end);
            fprintf(lookup, "%u\t%s_%u_%s_%s\t%zu\n", key, fld[0].c_str(), inSet, start.c_str(), end.c_str(), f);
            setOf.push_back((uint32_t) f);
            key++;
            inSet++;
            return true;
        };
        while (std::getline(in, line)) {
            if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
            if (!line.empty() && line[0] == '>') {
                if (!flush()) return fail("cannot write " + outDb);
                header = line.substr(1);
                seq.clear();
                have = true;
            } else if (have) {
                seq.append(line);
            }
        }
        if (!flush()) return fail("cannot write " + outDb);
        setCount.push_back(inSet);
    }
    fclose(lookup);
    fclose(source);
    if (!seqW.close(&err) || !hdrW.close(&err)) return fail(err);
    sddb::Writer m2s, s2m, ssz;
    if (!m2s.open(outDb + "_member_to_set", sddb::DBTYPE_ALIGNMENT_RES, &err)) return fail(err);
    if (!s2m.open(outDb + "_set_to_member", sddb::DBTYPE_ALIGNMENT_RES, &err)) return fail(err);
    if (!ssz.open(outDb + "_set_size", sddb::DBTYPE_GENERIC_DB, &err)) return fail(err);
    for (uint32_t k = 0; k < key; k++) {
        const std::string s = std::to_string(setOf[k]) + "\n";
        m2s.write(k, s.data(), s.size());
    }
    uint32_t k0 = 0;
    for (size_t s = 0; s < setCount.size(); s++) {
        std::string members;
        for (uint32_t k = k0; k < k0 + setCount[s]; k++) members.append(std::to_string(k)).append("\n");
        s2m.write((uint32_t) s, members.data(), members.size());
        const std::string c = std::to_string(setCount[s]) + "\n";
        ssz.write((uint32_t) s, c.data(), c.size());
        k0 += setCount[s];
    }
    if (!m2s.close(&err) || !s2m.close(&err) || !ssz.close(&err)) return fail(err);
    info(a, "%u sequences in %zu sets written to %s\n", key, setCount.size(), outDb.c_str());
    return 0;
}

}  // namespace sdcli
