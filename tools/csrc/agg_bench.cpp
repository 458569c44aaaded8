// tools/csrc/agg_bench.cpp -- measurement tool, not part of the product (not built by build.py).
// The aggregation stage (sd_agg_add / finish / records of csrc/host/sd_glue.cpp) on records shaped like a 1 000-proteome step of bench.py:
// 12 000 queries x 217 best hits (one per target set), 300-letter backtraces with 2 % gap letters, handed over in chunks of 2 500 queries.
// No GPU involved: the stage takes plain arrays.  (profiles/r05_experiments.txt items 17, 18.)
//   g++ -O2 -fopenmp -Iinclude -o /tmp/agg_bench tools/csrc/agg_bench.cpp -Lspacedust_amd -lsdgpu -Wl,-rpath,$PWD/spacedust_amd
//   OMP_NUM_THREADS=2 OMP_WAIT_POLICY=passive /tmp/agg_bench [hits per query]
#include "spacedust_gpu.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include <sys/resource.h>
static double cpuSec() { rusage u; getrusage(RUSAGE_SELF, &u); return u.ru_utime.tv_sec + u.ru_utime.tv_usec * 1e-6 + u.ru_stime.tv_sec + u.ru_stime.tv_usec * 1e-6; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const uint32_t nQ = 12000, nSets = 1000, perSet = 3000, nT = nSets * perSet, per = argc > 1 ? atoi(argv[1]) : 217;
    std::vector<uint32_t> qSetOf(nQ), tSetOf(nT);
    std::vector<int32_t> qLen(nQ, 300), tLen(nT, 300);
    for (uint32_t i = 0; i < nQ; i++) qSetOf[i] = i / perSet;
    for (uint32_t i = 0; i < nT; i++) tSetOf[i] = i / perSet;
    sd_agg *a = nullptr;
    if (sd_agg_create(qSetOf.data(), qLen.data(), nQ, tSetOf.data(), tLen.data(), nT, 4, nSets, 10.0, 2, 0.8f, 30, 1, &a)) return 1;
    std::mt19937 rng(1);
    std::vector<uint32_t> pq, pt;
    std::vector<sd_sw_result> res;
    std::vector<uint8_t> ident;
    std::string pool;
    for (uint32_t q = 0; q < nQ; q++)
        for (uint32_t x = 0; x < per; x++) {
            const uint32_t set = 4 + (x * 4 + rng() % 4) % (nSets - 4);
            pq.push_back(q);
            pt.push_back(set * perSet + rng() % perSet);
            sd_sw_result r;
            r.score = 200 + rng() % 400;
            r.qStart = 2; r.qEnd = 290; r.tStart = 3; r.tEnd = 291;
            r.flags = 0;
            r.evalue = 1e-30 * (1 + rng() % 1000) * 1e-3;
            r.btOffset = pool.size();
            int len = 0, idn = 0;
            for (int c = 0; c < 289; c++) {
                const uint32_t u = rng() % 100;
                if (u < 1) { pool.append(1 + rng() % 3, 'I'); }
                else if (u < 2) { pool.append(1 + rng() % 3, 'D'); }
                else { pool.push_back('M'); idn += (rng() & 1); }
            }
            len = (int) (pool.size() - r.btOffset);
            r.btLen = len;
            r.identical = idn;
            res.push_back(r);
            ident.push_back(0);
        }
    printf("records %zu pool %.1f MB\n", res.size(), pool.size() / 1e6);
    for (int rep = 0; rep < 3; rep++) {
        sd_agg *b = nullptr;
        sd_agg_create(qSetOf.data(), qLen.data(), nQ, tSetOf.data(), tLen.data(), nT, 4, nSets, 10.0, 2, 0.8f, 30, 1, &b);
        const double t0 = now(), c0 = cpuSec();
        // chunks of 2 500 queries as the pipeline hands them over
        size_t i0 = 0;
        for (uint32_t c = 0; c < nQ; c += 2500) {
            size_t i1 = i0;
            while (i1 < pq.size() && pq[i1] < c + 2500) i1++;
            std::vector<uint32_t> lq(pq.begin() + i0, pq.begin() + i1);
            for (auto &v : lq) v -= c;
            sd_agg_add(b, (uint32_t) (i1 - i0), c, lq.data(), pt.data() + i0, res.data() + i0, ident.data() + i0, pool.data());
            i0 = i1;
        }
        const double t1 = now(), c1 = cpuSec();
        uint64_t ne = 0, nh = 0;
        sd_agg_finish(b, &ne, &nh);
        const double t2 = now(), c2 = cpuSec();
        {   // cluster records of the finished aggregation: clusters of three consecutive hits, every fourth hit in none
            std::vector<uint64_t> eo(ne + 1); std::vector<uint32_t> eq(ne), et(ne), hq(nh), ht(nh); std::vector<double> pv(nh);
            sd_agg_get(b, eo.data(), eq.data(), et.data(), hq.data(), ht.data(), pv.data());
            std::vector<uint32_t> clu(nh, UINT32_MAX), rank(nh, 0), ncl(ne, 0), size(nh, 0); std::vector<double> pco(nh, 0.5), pmh(nh, 0.25);
            for (uint64_t e = 0; e < ne; e++) {
                uint32_t c = 0;
                for (uint64_t h = eo[e]; h + 4 <= eo[e + 1]; h += 4, c++) {
                    for (int x = 0; x < 3; x++) { clu[h + x] = c; rank[h + x] = x; }
                    size[eo[e] + c] = 3;
                }
                ncl[e] = c;
            }
            const double r0 = now();
            uint64_t bytes = 0;
            sd_agg_records(b, clu.data(), rank.data(), ncl.data(), pco.data(), pmh.data(), size.data(), nullptr, 0, &bytes);
            const double r1a = now();
            std::vector<char> out(bytes);
            const double r1 = now();
            printf("alloc %.3f ", r1 - r1a);
            sd_agg_records(b, clu.data(), rank.data(), ncl.data(), pco.data(), pmh.data(), size.data(), out.data(), bytes, &bytes);
            printf("records: size probe %.3f s, build %.3f s, %.1f MB\n", r1a - r0, now() - r1, bytes / 1e6);
        }
        printf("add: wall %.3f s cpu %.3f s | finish: wall %.3f cpu %.3f | entries %llu hits %llu\n", t1 - t0, c1 - c0, t2 - t1, c2 - c1,
               (unsigned long long) ne, (unsigned long long) nh);
        sd_agg_destroy(b);
    }
    return 0;
}
