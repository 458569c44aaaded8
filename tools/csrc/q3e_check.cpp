// tools/csrc/q3e_check.cpp -- test tool, not part of the product (not built by build.py).
// quantise3E of csrc/host/sd_glue.cpp (the "%.3E" + strtod round trip every matched hit makes three times) against snprintf / strtod on
// 2e7 random doubles over the whole exponent range, three doubles either side of every rounding boundary (m + 0.5) * 10^k and every m * 10^k
// of a dense sample of (m, k), and special values: 3.7e7 values, text and bit pattern.  tests/test_host_aggregation.py runs 2.2e5 of them
// through the C ABI (sd_host_quantise_3e) in the suite; this is the long form.
//   g++ -std=c++17 -O3 -mavx2 -mfma -ffp-contract=fast -fopenmp -Iinclude -Ispacedust_amd/csrc/host -Ispacedust_amd/csrc/hip \
//       tools/csrc/q3e_check.cpp -o /tmp/q3e_check -Lspacedust_amd -lsdgpu -Wl,-rpath,$PWD/spacedust_amd && /tmp/q3e_check
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include <cstdint>
#define main glue_main_unused
#include "../../spacedust_amd/csrc/host/sd_glue.cpp"
#undef main
static long bad = 0, n = 0;
static void check(double v) {
    char a[64], b[64];
    const double qa = quantise3E(v, a);
    snprintf(b, sizeof(b), "%.3E", v);
    const double qb = strtod(b, nullptr);
    n++;
    if (strcmp(a, b) != 0 || memcmp(&qa, &qb, 8) != 0) {
        if (bad++ < 20) printf("MISMATCH %.17g: '%s' %.17g vs '%s' %.17g\n", v, a, qa, b, qb);
    }
}
int main() {
    std::mt19937_64 rng(11);
    for (int i = 0; i < 20000000; i++) {   // random bit patterns over the exponent range E-values, log P and P-values live in
        const double mant = 1.0 + (double) (rng() >> 11) / 9007199254740992.0;
        const int e = (int) (rng() % 2040) - 1020;
        double v = ldexp(mant, e);
        if (rng() & 1) v = -v;
        check(v);
    }
    for (int k = -320; k <= 305; k++)   // around every boundary (m + 0.5) * 10^k and every m * 10^k
        for (int m = 1000; m <= 9999; m += (k % 7 == 0 ? 1 : 37)) {
            char t[64];
            for (int half = 0; half < 2; half++) {
                snprintf(t, sizeof(t), "%d%se%d", m, half ? ".5" : "", k);
                double v = strtod(t, nullptr);
                double lo = v, hi = v;
                for (int s = 0; s < 3; s++) {
                    check(lo); check(hi); check(-lo);
                    lo = nextafter(lo, 0.0);
                    hi = nextafter(hi, INFINITY);
                }
            }
        }
    const double sp[] = {0.0, -0.0, 1.0, 10.0, 1000.5, 0.5, 9999.5, 99995.0, 999.95, 9.9995, 1e22, 1e23, 1e-22, 1e-23, 5e-324, 1e-310, DBL_MIN, DBL_MAX, 1.7e308,
                         INFINITY, 1.0005, 2.5e-7, 1.1e-6, 10e-7, 13.815510557964274, -13.815510557964274, -708.3964185322641};
    for (double v : sp) { check(v); check(-v); }
    printf("%ld values, %ld mismatches\n", n, bad);
    return bad != 0;
}
