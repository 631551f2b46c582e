// Test program (tests/test_py2_numbers_cpu.py): the arithmetic round(x, 2) / str(float) of the native record layer against their
// text-based definitions (glibc "%.2f" / "%.12g" on the exact value), on random doubles, ties and their neighbours.
#include "records.hpp"
#include <random>
#include <cstring>
using namespace plathost;
int main(int argc, char** argv) {
    const long loops = argc > 1 ? atol(argv[1]) : 200000;
    std::mt19937_64 g(12345);
    long bad = 0, badS = 0, n = 0;
    auto check = [&](double x) {
        ++n;
        const double a = py2_round2(x), b = py2_round2_slow(x);
        if (memcmp(&a, &b, 8) != 0 && !(std::isnan(a) && std::isnan(b))) { if (bad++ < 10) printf("round2 %.17g: %.17g vs %.17g\n", x, a, b); }
        if (py2_str(x) != py2_str_slow(x)) { if (badS++ < 10) printf("str %.17g: %s vs %s\n", x, py2_str(x).c_str(), py2_str_slow(x).c_str()); }
        if (py2_str(a) != py2_str_slow(a)) { if (badS++ < 10) printf("str(r) %.17g: %s vs %s\n", a, py2_str(a).c_str(), py2_str_slow(a).c_str()); }
    };
    for (long i = 0; i < loops; ++i) {
        uint64_t bits = g(); double x; memcpy(&x, &bits, 8); check(x);                       // any double
        const double u = (double)(g() >> 11) / 9007199254740992.0;
        check(u); check(-u); check(u * 1000); check(-u * 300); check(u * 1e-3); check(std::log10(std::max(u, 1e-300)));
        const long k = (long)(g() % 2000001) - 1000000;
        check(k / 200.0); check(k / 8.0); check(k / 100.0); check(k / 1000.0); check(std::nextafter(k / 200.0, 1e9)); check(std::nextafter(k / 200.0, -1e9));
        check((double)(g() % 100000000000ull) / 100.0); check((double)(g() % 100000000000000ull) / 100.0);
        // the arithmetic "%.12g" (round 6): any magnitude it takes, thirteen-digit values ending in 5 (ties and near-ties at the twelfth digit) and their neighbours
        const double mag = std::pow(10.0, (double)((long)(g() % 17) - 5));
        check(u * mag); check(-u * mag); check((1.0 + u) * mag);
        const unsigned long long n13 = 1000000000000ull + (g() % 9000000000000ull) / 10 * 10 + 5;
        for (int j = 0; j <= 16; j += 4) { const double t = (double)n13 / std::pow(10.0, (double)j); check(t); check(std::nextafter(t, 0.0)); check(std::nextafter(t, 1e300)); }
        check((double)(long)(g() % 2000) + 0.5); check(std::ldexp((double)(g() % (1ull << 40)) + 0.5, -(int)(g() % 30)));
    }
    for (int e = -6; e <= 13; ++e) { const double t = std::pow(10.0, e); check(t); check(std::nextafter(t, 0.0)); check(std::nextafter(t, 1e300)); check(t * 9.99999999999949); check(t * 9.9999999999995); check(t * 9.99999999999951); }
    check(0.0); check(-0.0); check(0.005); check(0.015); check(1e9); check(99999999.99); check(1e13); check(9.999999999999e12); check(INFINITY); check(-INFINITY); check(NAN);
    printf("checked %ld values: round2 differs %ld, str differs %ld\n", n, bad, badS);
    return bad || badS;
}
