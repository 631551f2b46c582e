// records.hpp -- the INFO / FILTER arithmetic and the VCF record text, native (libplat_caller.so).
//
//   logFactorial, logBetaFunction, threeFTwo, betaBinomialCDF         src/cython/platypusutils.pyx:178-315
//   computeAlleleBiasPValue, computeStrandBiasPValue                   src/cython/vcfutils.pyx:1156-1222
//   the INFO fields derived from vcfINFO's per-read loop               src/cython/vcfutils.pyx:1392-1440
//   homopolymerLengthForOneVariant, getSequenceContext                 src/cython/chaplotype.pyx:462-506
//   computeSCValue, vcfFILTER                                          src/cython/vcfutils.pyx:1480-1627
//   refAndAlt, trimLeftPadding, outputCallToVCF                        src/cython/vcfutils.pyx:338-599,796-897
//   VCF.write_data / format_formatdata                                 src/python/vcf.py:297-329,710-739
//
// The reference runs under Python 2; three of its behaviours reach the text and are restated here exactly as in
// platypus_amd/vcfrecords.py: round() (ties away from zero on the exact binary value), str(float) ("%.12g") and the
// iteration order of a set of filter names.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "variants.hpp"

namespace plathost {

// ---- Python 2 semantics ---------------------------------------------------------------------------------------------------
inline std::string py2_str_slow(double x) {                              // str(float) of Python 2: "%.12g", ".0" for integral text
    char buf[64];
    snprintf(buf, sizeof buf, "%.12g", x);
    std::string t(buf);
    size_t i = (!t.empty() && t[0] == '-') ? 1 : 0;
    while (i < t.size() && t[i] == '-') ++i;
    bool digits = i < t.size();
    for (size_t k = i; k < t.size(); ++k) if (t[k] < '0' || t[k] > '9') { digits = false; break; }
    return digits ? t + ".0" : t;
}
// decimal digits of v at p, returns the end.  Two digits per division from a table; 32-bit arithmetic for the values that fit (nearly all
// numbers of a record: positions, counts, phred values)
static const char DIGIT_PAIRS[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
inline char* put_u32(char* p, uint32_t v) {
    char tmp[12];
    int n = 0;
    while (v >= 100) { const uint32_t q = v / 100, r = v - q * 100; tmp[n++] = DIGIT_PAIRS[2 * r + 1]; tmp[n++] = DIGIT_PAIRS[2 * r]; v = q; }
    if (v >= 10) { tmp[n++] = DIGIT_PAIRS[2 * v + 1]; tmp[n++] = DIGIT_PAIRS[2 * v]; }
    else tmp[n++] = (char)('0' + v);
    while (n) *p++ = tmp[--n];
    return p;
}
inline char* put_uint(char* p, unsigned long long v) {
    if (v <= 0xFFFFFFFFull) return put_u32(p, (uint32_t)v);
    char tmp[24];
    int n = 0;
    while (v >= 100) { const unsigned long long q = v / 100; const unsigned r = (unsigned)(v - q * 100); tmp[n++] = DIGIT_PAIRS[2 * r + 1]; tmp[n++] = DIGIT_PAIRS[2 * r]; v = q; }
    if (v >= 10) { tmp[n++] = DIGIT_PAIRS[2 * v + 1]; tmp[n++] = DIGIT_PAIRS[2 * v]; }
    else tmp[n++] = (char)('0' + v);
    while (n) *p++ = tmp[--n];
    return p;
}
// "%.12g" of Python 2's str(float) WITHOUT printf for 1e-4 <= |x| < 1e12 (fixed notation there): the twelve significant digits of the
// double's EXACT value, round half to even.  x * 10^(11 - k) (k = the decimal exponent; the power is exact) is formed with its rounding
// error (fma): prod + err is the exact product, prod < 2^40 so its fraction is exact; the shortcut is taken only when the exact
// fraction is farther than 1e-3 from a half (err < 1e-4), everything else goes through snprintf.  Returns nullptr when it does not apply.
inline char* put_g12_fast(char* p, double x) {
    static const double P10[27] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22,
                                   1e-1, 1e-2, 1e-3, 1e-4};
    const double ax = fabs(x);
    if (!(ax >= 1e-4 && ax < 1e12)) return nullptr;
    int k;                                                                  // 10^k <= ax < 10^(k+1)
    if (ax >= 1.0) { k = 0; while (k < 11 && ax >= P10[k + 1]) ++k; }
    else { k = -1; while (k > -4 && ax < P10[22 - k]) --k; }                // P10[23] = 1e-1 ... P10[26] = 1e-4
    const double sc = P10[11 - k];                                          // 10^(11 - k), exact (exponent 0 .. 15)
    const double prod = ax * sc, err = fma(ax, sc, -prod);
    if (!(prod >= 1e11 * 0.999999 && prod < 1e12 * 1.000001)) return nullptr;
    double fl = floor(prod);
    const double frac = (prod - fl) + err;                                  // exact fraction of the exact product (|err| < 1e-4)
    if (fabs(frac - 0.5) < 1e-3 || frac < -0.25 || frac > 1.25) return nullptr;
    unsigned long long n = (unsigned long long)fl + (frac > 0.5 ? 1ull : 0ull);
    if (n < 100000000000ull) return nullptr;                                // (k was one too large: cannot happen with exact comparisons; be safe)
    if (n >= 1000000000000ull) { n /= 10; ++k; if (k >= 12) return nullptr; }   // rounded up to the next power of ten
    char d[12];
    for (int i = 11; i >= 0; --i) { d[i] = (char)('0' + n % 10); n /= 10; }
    int last = 11;
    while (last > 0 && d[last] == '0') --last;                              // %g strips trailing zeros
    if (std::signbit(x)) *p++ = '-';
    if (k >= 0) {
        for (int i = 0; i <= k; ++i) *p++ = d[i];
        if (last > k) { *p++ = '.'; for (int i = k + 1; i <= last; ++i) *p++ = d[i]; }
        else { *p++ = '.'; *p++ = '0'; }                                    // digits only: str(float) appends ".0"
    } else {
        *p++ = '0'; *p++ = '.';
        for (int i = -1; i > k; --i) *p++ = '0';
        for (int i = 0; i <= last; ++i) *p++ = d[i];
    }
    return p;
}
// The same text appended to `out`.  Nearly every float that reaches a record was rounded to two decimals first: a double that IS the
// double nearest to n/100 with n < 10^11 prints under "%.12g" as the decimal n/100 without its trailing zeros (the twelve significant
// digits of its exact value round to those of n/100, which has at most eleven), so that text is written straight from n.
inline void append_py2_str(std::string& out, double x) {
    const double ax = fabs(x);
    if (ax < 1e9) {
        const unsigned long long n = (unsigned long long)(ax * 100.0 + 0.5);
        if ((double)n / 100.0 == ax) {
            char buf[32], *p = buf;
            if (std::signbit(x)) *p++ = '-';
            p = put_uint(p, n / 100);
            const unsigned f = (unsigned)(n % 100);
            *p++ = '.';
            *p++ = (char)('0' + f / 10);
            if (f % 10) *p++ = (char)('0' + f % 10);
            out.append(buf, (size_t)(p - buf));
            return;
        }
    }
    char buf[40];
    if (char* q = put_g12_fast(buf, x)) { out.append(buf, (size_t)(q - buf)); return; }
    out += py2_str_slow(x);
}
inline std::string py2_str(double x) { std::string t; append_py2_str(t, x); return t; }
// the same writers on a character pointer (the record line is written into space reserved for it); each returns the new end
inline char* put_chars(char* p, const char* s, size_t n) { memcpy(p, s, n); return p + n; }
inline char* put_str(char* p, const std::string& s) { memcpy(p, s.data(), s.size()); return p + s.size(); }
template <size_t N> inline char* put_lit(char* p, const char (&s)[N]) { memcpy(p, s, N - 1); return p + (N - 1); }
inline char* put_int(char* p, long long v) {
    unsigned long long u = (unsigned long long)v;
    if (v < 0) { *p++ = '-'; u = 0ull - u; }
    return put_uint(p, u);
}
inline char* put_py2_str(char* p, double x) {                             // at most 32 characters
    const double ax = fabs(x);
    if (ax < 1e9) {
        const unsigned long long n = (unsigned long long)(ax * 100.0 + 0.5);
        if ((double)n / 100.0 == ax) {
            if (std::signbit(x)) *p++ = '-';
            p = put_uint(p, n / 100);
            const unsigned f = (unsigned)(n % 100);
            *p++ = '.';
            *p++ = (char)('0' + f / 10);
            if (f % 10) *p++ = (char)('0' + f % 10);
            return p;
        }
    }
    if (char* q = put_g12_fast(p, x)) return q;
    const std::string t = py2_str_slow(x);
    return put_str(p, t);
}
// "%.0f" % x and "%1.4f" % x (the PP and FR fields) without printf for the values that occur: both print the decimal nearest to the
// double's EXACT value (ties to even).  x * 10^d is computed with its rounding error (fma); the shortcut is taken only when the product is
// far enough from a tie for that error not to matter, everything else goes through snprintf.
inline void append_fixed(std::string& out, double x, int decimals) {     // decimals 0 or 4
    if (!std::signbit(x) && x < (decimals == 4 ? 1e5 : 1e9)) {        // (x * 1e4 < 1e9: its rounding error stays below 6e-8)
        const double scale = decimals == 4 ? 1e4 : 1.0;
        const double y = x * scale, err = fma(x, scale, -y);              // x * scale == y + err exactly
        const double fl = floor(y), frac = y - fl;                        // (exact: y < 2^44)
        const double d = fabs(frac - 0.5);
        if (d > 1e-6 || decimals == 0) {
            unsigned long long n = (unsigned long long)fl;
            if (decimals == 0) {                                          // y == x: a tie is exact, round half to even
                if (frac > 0.5 || (frac == 0.5 && (n & 1ull))) ++n;
            } else if (frac > 0.5) ++n;
            (void)err;
            char buf[40], *q = buf;
            if (decimals == 0) q = put_uint(q, n);
            else {
                q = put_uint(q, n / 10000);
                unsigned f = (unsigned)(n % 10000);
                *q++ = '.';
                *q++ = (char)('0' + f / 1000); f %= 1000;
                *q++ = (char)('0' + f / 100); f %= 100;
                *q++ = (char)('0' + f / 10);
                *q++ = (char)('0' + f % 10);
            }
            out.append(buf, (size_t)(q - buf));
            return;
        }
    }
    char buf[400];
    snprintf(buf, sizeof buf, decimals == 4 ? "%1.4f" : "%.0f", x);
    out += buf;
}
inline void append_int(std::string& out, long long v) {
    char buf[24], *p = buf;
    unsigned long long u = (unsigned long long)v;
    if (v < 0) { *p++ = '-'; u = 0ull - u; }
    p = put_uint(p, u);
    out.append(buf, (size_t)(p - buf));
}
// round(x, 2) of Python 2: the exact binary value rounded to two decimals, ties away from zero; as a double.
// A double is an exact tie at two decimals only when it is an odd multiple of 1/8 (x = k/200 dyadic => 25 | k).
inline double py2_round2_slow(double x) {
    if (std::isnan(x) || std::isinf(x)) return x;
    const double ax = fabs(x);
    char buf[400];
    if (ax < 4503599627370496.0) {
        const double y = ax * 8.0;                                       // exact
        if (y == floor(y) && fmod(y, 2.0) == 1.0) {
            const double up = floor(ax * 100.0) + 1.0;                   // ax * 100 = 12.5 * odd: exact
            snprintf(buf, sizeof buf, "%s%.0f", x < 0 ? "-" : "", up);
            std::string s(buf);
            const size_t neg = x < 0 ? 1 : 0;
            while (s.size() - neg < 3) s.insert(neg, "0");
            s.insert(s.size() - 2, ".");
            return strtod(s.c_str(), nullptr);
        }
    }
    snprintf(buf, sizeof buf, "%.2f", x);                               // glibc: correctly rounded on the exact value
    return strtod(buf, nullptr);
}
// The same without going through text, for |x| < 10^13: 100 |x| = y + e exactly (y the rounded product, e its error from one fused
// multiply-add); n = y rounded to an integer, ties away from zero.  y - n is exact and a multiple of ulp(y), so unless y sits exactly on
// a half, e (at most half an ulp) cannot move the exact product across one; when it does sit on a half, the sign of e says on which side
// the exact product lies (e = 0: a true tie, away from zero).  The result is the double nearest to n/100 -- one correctly rounded division,
// which is what strtod makes of the decimal text.
inline double py2_round2(double x) {
    const double ax = fabs(x);
    if (!(ax < 1e13)) return py2_round2_slow(x);
    const double y = ax * 100.0, e = fma(ax, 100.0, -y);
    double n = floor(y);
    const double d = y - n;                                              // exact
    if (d > 0.5 || (d == 0.5 && e >= 0.0)) n += 1.0;
    const double r = n / 100.0;
    return std::signbit(x) ? -r : r;
}
inline double py2_round0(double x) { return round(x); }                  // ties away from zero

inline uint64_t py2_string_hash(const std::string& b) {                  // stringobject.c (2.7, 64-bit)
    if (b.empty()) return 0;
    uint64_t x = ((uint64_t)(unsigned char)b[0]) << 7;
    for (unsigned char c : b) x = (1000003ull * x) ^ c;
    x ^= (uint64_t)b.size();
    return x == ~0ull ? ~0ull - 1 : x;
}
// list(set(names)) under CPython 2.7 (setobject.c): open addressing (i = 5i + perturb + 1, perturb >>= 5) in a table of 8
// slots that is rebuilt four times larger when two thirds full; iteration = slot order
inline std::vector<std::string> py2_set_order(const std::vector<std::string>& names) {
    std::vector<const std::string*> table(8, nullptr);
    auto slot = [](const std::vector<const std::string*>& t, const std::string& key, uint64_t h) -> size_t {
        const uint64_t mask = t.size() - 1;
        uint64_t i = h & mask, perturb = h;
        while (t[i & mask] && *t[i & mask] != key) { i = 5 * i + perturb + 1; perturb >>= 5; }
        return (size_t)(i & mask);
    };
    size_t used = 0;
    for (const std::string& key : names) {
        const size_t j = slot(table, key, py2_string_hash(key));
        if (table[j]) continue;
        table[j] = &key;
        ++used;
        if (used * 3 >= table.size() * 2) {
            size_t size = 8;
            while (size <= used * (used > 50000 ? 2 : 4)) size <<= 1;
            std::vector<const std::string*> grown(size, nullptr);
            for (const std::string* k : table) if (k) grown[slot(grown, *k, py2_string_hash(*k))] = k;
            table.swap(grown);
        }
    }
    std::vector<std::string> out;
    for (const std::string* k : table) if (k) out.push_back(*k);
    return out;
}

// The same for a short list of C-string names (the FILTER column: at most eight distinct names, repeated per variant of the record), without
// a heap allocation: distinct names in the order list(set(names)) gives.  out must hold n pointers; returns the number written.
inline int py2_set_order_names(const char* const* names, int n, const char** out) {
    const char* table[64];                                                // 8 slots, rebuilt 4 x used when two thirds full: 8 distinct names never pass 32 slots
    uint64_t hashes[64];
    int size = 8, used = 0;
    for (int k = 0; k < size; ++k) table[k] = nullptr;
    auto slot = [&](const char* const* t, int sz, const char* key, uint64_t h) -> int {
        const uint64_t mask = (uint64_t)sz - 1;
        uint64_t i = h & mask, perturb = h;
        while (t[i & mask] && strcmp(t[i & mask], key) != 0) { i = 5 * i + perturb + 1; perturb >>= 5; }
        return (int)(i & mask);
    };
    for (int q = 0; q < n; ++q) {
        const char* key = names[q];
        const uint64_t h = py2_string_hash(std::string(key));             // (names of at most ten characters: in-object strings)
        const int j = slot(table, size, key, h);
        if (table[j]) continue;
        table[j] = key; hashes[j] = h;
        ++used;
        if (used * 3 >= size * 2) {
            int nsize = 8;
            while (nsize <= used * 4) nsize <<= 1;
            if (nsize > 64) nsize = 64;                                   // (cannot happen with the header's eight filter names)
            const char* grown[64];
            uint64_t gh[64];
            for (int k = 0; k < nsize; ++k) grown[k] = nullptr;
            for (int k = 0; k < size; ++k) if (table[k]) { const int g = slot(grown, nsize, table[k], hashes[k]); grown[g] = table[k]; gh[g] = hashes[k]; }
            for (int k = 0; k < nsize; ++k) { table[k] = grown[k]; hashes[k] = gh[k]; }
            size = nsize;
        }
    }
    int m = 0;
    for (int k = 0; k < size; ++k) if (table[k]) out[m++] = table[k];
    return m;
}

// iteration order of a Python-2 dict holding these integer keys, inserted in this order (dictobject.c: the same open addressing and
// growth as the set above; hash(int) is the int, -1 -> -2): what `for pos, vars in pop.varsByPos.iteritems()` walks (variantcaller.pyx:584-603)
inline std::vector<int> py2_int_dict_order(const std::vector<int>& keys) {
    std::vector<long long> table(8, 0);
    std::vector<char> usedSlot(8, 0);
    auto slot = [](const std::vector<long long>& t, const std::vector<char>& u, long long key) -> size_t {
        const uint64_t h = (uint64_t)(key == -1 ? -2 : key), mask = t.size() - 1;
        uint64_t i = h & mask, perturb = h;
        while (u[i & mask] && t[i & mask] != key) { i = 5 * i + perturb + 1; perturb >>= 5; }
        return (size_t)(i & mask);
    };
    size_t used = 0;
    for (int key : keys) {
        const size_t j = slot(table, usedSlot, key);
        if (usedSlot[j]) continue;
        table[j] = key; usedSlot[j] = 1;
        ++used;
        if (used * 3 >= table.size() * 2) {
            size_t size = 8;
            while (size <= used * (used > 50000 ? 2 : 4)) size <<= 1;
            std::vector<long long> gt(size, 0);
            std::vector<char> gu(size, 0);
            for (size_t k = 0; k < table.size(); ++k) if (usedSlot[k]) { const size_t q = slot(gt, gu, table[k]); gt[q] = table[k]; gu[q] = 1; }
            table.swap(gt); usedSlot.swap(gu);
        }
    }
    std::vector<int> out;
    for (size_t k = 0; k < table.size(); ++k) if (usedSlot[k]) out.push_back((int)table[k]);
    return out;
}

// ---- beta-binomial p-values -------------------------------------------------------------------------------------------------
inline double logFactorialUncached(long x);
// (memoised per thread: the beta-binomial p-values of a record ask for a dozen log-factorials of read counts, each five pow() calls;
//  the same function of the same integer gives the same double)
inline double logFactorial(long x) {
    constexpr long N = 4096;
    static thread_local double cache[N];
    static thread_local unsigned char have[N];
    if (x < 0 || x >= N) return logFactorialUncached(x);
    if (!have[x]) { cache[x] = logFactorialUncached(x); have[x] = 1; }
    return cache[x];
}
inline double logFactorialUncached(long x) {                             // platypusutils.pyx:178-191
    if (x < 15) {
        double ans = 0.0;
        for (long i = 1; i <= x; ++i) ans += log((double)i);
        return ans;
    }
    const double y = (double)x;
    return (y * log(y) + log(2.0 * M_PI * y) / 2 - y + (pow(y, -1)) / 12 - (pow(y, -3)) / 360 + (pow(y, -5)) / 1260 - (pow(y, -7)) / 1680 +
            (pow(y, -9)) / 1188);
}
inline double logBetaFunction(long x, long y) { return (logFactorial(x - 1) + logFactorial(y - 1)) - logFactorial(x + y - 1); }
inline double threeFTwo(long k, long n, long alpha, long beta) {         // :267-295
    const double a_2 = alpha + k + 1.0, a_3 = k - n + 1.0, b_1 = k + 2.0, b_2 = -beta - n + k + 2.0;
    double theSum = 1.0, lastTerm = 1.0;
    const long m = labs(k - n + 1);
    for (long i = 1; i <= m; ++i) {
        const double newTerm = lastTerm * (a_2 + i - 1) * (a_3 + i - 1) / ((b_1 + i - 1) * (b_2 + i - 1));
        theSum += newTerm;
        lastTerm = newTerm;
    }
    return theSum;
}
inline double betaBinomialCDF(long k, long n, long alpha, long beta) {   // :306-315
    if (k == n) return 1.0;
    const double numerator = logBetaFunction(beta + n - k - 1, alpha + k + 1) + log(threeFTwo(k, n, alpha, beta));
    const double denominator = logBetaFunction(alpha, beta) + logBetaFunction(n - k, k + 2) + log((double)(n + 1));
    return std::max(1e-30, 1.0 - exp(numerator - denominator));
}
inline double computeAlleleBiasPValue(long totalReads, long variantReads) {   // vcfutils.pyx:1156-1173
    if (totalReads > 0 && (double)variantReads / (double)totalReads >= 0.5) return 1.0;
    if (totalReads == 0) return 1.0;
    const double p = betaBinomialCDF(variantReads, totalReads, 20, 20);
    return std::min(p, 1.0 - p);
}
inline double computeStrandBiasPValue(long nFwdReads, long nRevReads, long nFwdVarReads, long nRevVarReads) {   // :1177-1222
    if (nFwdReads == 0 || nRevReads == 0) return 1.0;
    const bool useForward = !(nFwdReads < nRevReads);
    if (nFwdReads + nRevReads > 0 && nFwdVarReads + nRevVarReads > 0) {
        const double freq = (double)(useForward ? nFwdReads : nRevReads) / (double)(nFwdReads + nRevReads);
        long alpha, beta;
        if (freq < 0.5) { alpha = 20; beta = (long)((double)alpha / freq - alpha); }
        else if (freq > 0.5) { beta = 20; alpha = (long)(beta * freq / (1.0 - freq)); }
        else alpha = beta = 20;
        return betaBinomialCDF(useForward ? nFwdVarReads : nRevVarReads, nFwdVarReads + nRevVarReads, alpha, beta);
    }
    return 1.0;
}

// ---- INFO -------------------------------------------------------------------------------------------------------------------
// hash(tuple) of CPython 2.7 (tupleobject.c tuplehash) from the items' hashes, as the unsigned value a dict's table is indexed with
inline uint64_t py2_tuple_hash(const uint64_t* itemHashes, int n) {
    uint64_t x = 0x345678ull, mult = 1000003ull;
    for (int i = 0; i < n; ++i) {
        const int left = n - 1 - i;
        x = (x ^ itemHashes[i]) * mult;
        mult += (uint64_t)(82520ll + left + left);
    }
    x += 97531ull;
    return x == ~0ull ? ~0ull - 1 : x;
}
// hash(Variant) there: hash((refName, refPos, removed, added)), variant.pyx:270-280 (refPos >= 0: hash(int) is the int) -- kept in
// `public int hashValue` (variant.pxd:31): Cython's hash() is PyObject_Hash -> Py_hash_t, the assignment narrows it to a C int without
// a check, __hash__ hands the int back (a -1 becomes -2 in the tp_hash slot) and the dictionary probes with the SIGN-EXTENDED low 32 bits
inline uint64_t py2_variant_hash(uint64_t refNameHash, long long refPos, const char* removed, size_t nRemoved, const char* added, size_t nAdded) {
    const uint64_t h[4] = {refNameHash, (uint64_t)(refPos == -1 ? -2 : refPos), py2_string_hash(std::string(removed, nRemoved)),
                           py2_string_hash(std::string(added, nAdded))};
    int64_t narrowed = (int64_t)(int32_t)(uint32_t)py2_tuple_hash(h, 4);
    if (narrowed == -1) narrowed = -2;
    return (uint64_t)narrowed;
}
// Iteration order of a Python-2 dict into which DISTINCT keys with these hashes were inserted in this order and never deleted
// (dictobject.c: a new key takes the first empty slot of its probe sequence i = 5 i + perturb + 1, perturb >>= 5; the table of 8 slots
// is rebuilt 4 x used (2 x above 50 000 keys) slots large, in slot order, when it is two thirds full): indices into `hashes`
inline std::vector<int> py2_dict_slot_order(const std::vector<uint64_t>& hashes) {
    std::vector<int> table(8, -1);
    auto place = [&](std::vector<int>& t, int key) {
        const uint64_t h = hashes[(size_t)key], mask = t.size() - 1;
        uint64_t i = h & mask, perturb = h;
        while (t[i & mask] >= 0) { i = 5 * i + perturb + 1; perturb >>= 5; }
        t[i & mask] = key;
    };
    size_t used = 0;
    for (int key = 0; key < (int)hashes.size(); ++key) {
        place(table, key);
        ++used;
        if (used * 3 >= table.size() * 2) {
            size_t size = 8;
            while (size <= used * (used > 50000 ? 2 : 4)) size <<= 1;
            std::vector<int> grown(size, -1);
            for (int k : table) if (k >= 0) place(grown, k);
            table.swap(grown);
        }
    }
    std::vector<int> out;
    out.reserve(hashes.size());
    for (int k : table) if (k >= 0) out.push_back(k);
    return out;
}

struct Num {                                                              // a Python number as it reaches the text: int or float
    bool isInt = true; long long i = 0; double d = 0.0;
    static Num I(long long v) { Num n; n.isInt = true; n.i = v; n.d = (double)v; return n; }
    static Num D(double v) { Num n; n.isInt = false; n.d = v; return n; }
    double value() const { return isInt ? (double)i : d; }
    std::string text() const {                                            // py2_str; a value equal to the field's missing value (-1) is "."
        if (value() == -1.0) return ".";
        if (isInt) return std::to_string(i);
        return py2_str(d);
    }
    void appendTo(std::string& out) const {
        if (value() == -1.0) { out += '.'; return; }
        if (isInt) { append_int(out, i); return; }
        append_py2_str(out, d);
    }
    char* put(char* p) const {                                            // the same characters at p (at most 32)
        if (value() == -1.0) { *p++ = '.'; return p; }
        if (isInt) return put_int(p, i);
        return put_py2_str(p, d);
    }
};

// a short text held inside the object (the 21-base sequence context of a variant: one heap block per called variant otherwise)
struct InlineStr {
    char c[31]; uint8_t n = 0;
    size_t size() const { return n; }
    const char* data() const { return c; }
    const char* begin() const { return c; }
    const char* end() const { return c + n; }
    void assign(const char* s, size_t len) { n = (uint8_t)std::min<size_t>(len, sizeof c); memcpy(c, s, n); }
};
struct VarInfo {                                                          // vcfInfo[variant]
    Variant* var = nullptr;
    int HP = 0;
    InlineStr SC;
    std::string PP, FRtext;
    double PPnum = 0.0; int PPint = 0;                                    // float(PP) and int(float(PP)) as the text gives them: parsed once (setPP)
    void setPP(double posterior) { PP.clear(); append_fixed(PP, posterior, 0); PPnum = strtod(PP.c_str(), nullptr); PPint = atoi(PP.c_str()); }   // "%.0f"
    double FRsum = 0.0;
    Num ABPV, SbPval, BRF, MQ, QD;
    long long TR = 0, NF = 0, NR = 0, TC = 0, TCR = 0, TCF = 0;
    int MMLQ = 100, HapScore = 0;
    SmallVec<int, 4> nReadsPerSample, nVarReadsPerSample;
    SmallVec<const char*, 4> Source;                                      // (string literals)
    SmallVec<const char*, 8> filters;                                     // vcfFilter[variant] (string literals)
};

// chaplotype.pyx:462-498
inline int homopolymerLengthForOneVariant(const Variant& v, const Fasta& fa) {
    // the two getSequence intervals [refPos - 20, refPos) and [refPos + 1, refPos + 21) as views of the reference (clamped and checked as
    // getSequence does, fastafile.pyx:173-207)
    const int64_t lb = std::max<int64_t>(0, (int64_t)v.refPos - 20), le = std::min<int64_t>(fa.len - 1, v.refPos);
    const int64_t rb = std::max<int64_t>(0, (int64_t)v.refPos + 1), re = std::min<int64_t>(fa.len - 1, (int64_t)v.refPos + 21);
    if (le < lb || re < rb) throw WindowError("Cannot have beginPos > endPos in getSequence");
    const char* left = (const char*)fa.seq + lb; const char* right = (const char*)fa.seq + rb;
    const int64_t nL = le - lb, nR = re - rb;
    if (nL == 0 || nR == 0) return 0;
    int nl = 0, nr = 0;
    for (int64_t i = nL; i-- > 0 && left[i] == left[nL - 1];) ++nl;
    for (int64_t i = 0; i < nR && right[i] == right[0]; ++i) ++nr;
    return left[nL - 1] != right[0] ? std::max(nl, nr) : nl + nr;
}
inline std::string getSequenceContext(const Variant& v, const Fasta& fa) { return fa.getSequence(v.refPos - 10, v.refPos + 11); }   // :500-506
inline void getSequenceContext(const Variant& v, const Fasta& fa, InlineStr& out) {       // the same interval, clamped and checked as getSequence does
    const int64_t b = std::max<int64_t>(0, (int64_t)v.refPos - 10), e = std::min<int64_t>(fa.len - 1, (int64_t)v.refPos + 11);
    if (e < b) throw WindowError("Cannot have beginPos > endPos in getSequence");
    out.assign((const char*)fa.seq + b, (size_t)(e - b));
}

template <class Str> inline double computeSCValue(const Str& sequence) {  // vcfutils.pyx:1480-1498
    // the two most frequent characters of the 21-base context: counted among the characters that occur (a table of 256 counters was
    // cleared and scanned per called position)
    static thread_local int counts[256];                                  // all zero between calls
    for (unsigned char c : sequence) ++counts[c];
    int best = 0, second = 0;
    for (unsigned char c : sequence) {
        const int n = counts[c];
        if (n == 0) continue;                                             // this character was taken already
        counts[c] = 0;
        if (n > best) { second = best; best = n; }
        else if (n > second) second = n;
    }
    return (double)(best + second) / (double)sequence.size();
}

// the INFO fields vcfINFO derives from its per-read loop (vcfutils.pyx:1392-1440), from the counters of
// plat_variant_read_stats_batch: counts[16] = TC, TC_bad, TR, TC_ab, TR_ab, NR_sb, NF_sb, TCR, TCF, TCR_sb, TCF_sb, NR, NF, nGood, nBad, sumsq
// terms != nullptr: the loops were done on the device (plat_variant_info_batch: per p-value {state, logBeta term, threeFTwo, denominator});
// the libm calls of betaBinomialCDF stay here, so the doubles are the ones the functions above give
inline double cdfFromTerms(const double* t) { return std::max(1e-30, 1.0 - exp((t[1] + log(t[2])) - t[3])); }
inline void infoFieldsFromReadStats(VarInfo& d, const int64_t* c, const int32_t* perSample, int nInd, const int32_t* minq, int nminq,
                                    const double* terms = nullptr, int mmlqDevice = -1) {
    const long long TC = c[0], TC_bad = c[1], TR = c[2], TC_ab = c[3], TR_ab = c[4], NR_sb = c[5], NF_sb = c[6], TCR = c[7], TCF = c[8],
                    TCR_sb = c[9], TCF_sb = c[10], NR = c[11], NF = c[12], nGood = c[13], nBad = c[14], sumsq = c[15];
    double abpv, sbpv;
    if (terms && terms[0] == 0.0) abpv = terms[1];
    else if (terms && terms[0] == 1.0) { const double p = cdfFromTerms(terms); abpv = std::min(p, 1.0 - p); }
    else abpv = computeAlleleBiasPValue(TC_ab, TR_ab);
    if (terms && terms[4] == 0.0) sbpv = terms[5];
    else if (terms && terms[4] == 1.0) sbpv = cdfFromTerms(terms + 4);
    else sbpv = computeStrandBiasPValue(TCF_sb, TCR_sb, NF_sb, NR_sb);
    d.ABPV = Num::D(py2_round2(abpv));
    d.SbPval = Num::D(py2_round2(sbpv));
    d.TR = TR; d.NF = NF; d.NR = NR; d.TC = TC; d.TCR = TCR; d.TCF = TCF;
    d.BRF = Num::D(py2_round2((double)nBad / (double)(nGood + nBad)));
    d.nReadsPerSample.resize(nInd); d.nVarReadsPerSample.resize(nInd);
    for (int i = 0; i < nInd; ++i) { d.nReadsPerSample[i] = perSample[2 * i]; d.nVarReadsPerSample[i] = perSample[2 * i + 1]; }
    const float rms = (float)sumsq;                                       // `cdef float RMSMQ`: the quotient is a C float too
    if (TC + TC_bad > 0 && rms > 0) d.MQ = Num::D(py2_round2(sqrt((double)(rms / (float)(TC + TC_bad)))));
    else d.MQ = Num::I(0);
    if (mmlqDevice >= 0) d.MMLQ = mmlqDevice;
    else if (nminq > 0) {                                                 // sorted(...)[n // 2]: the element a full sort would leave there
        int small[256];
        std::vector<int> big;
        int* q = small;
        if (nminq > 256) { big.assign(minq, minq + nminq); q = big.data(); }
        else memcpy(small, minq, (size_t)nminq * sizeof(int));
        std::nth_element(q, q + nminq / 2, q + nminq);
        d.MMLQ = q[nminq / 2];
    } else d.MMLQ = 100;
}

// ---- REF / ALT ------------------------------------------------------------------------------------------------------------------
inline void refAndAlt(int POS, const VarList& variants, const Fasta& fa, std::string& REF, std::vector<std::string>& ALT) {   // vcfutils.pyx:843-897
    bool onlySnps = true, indel = false;
    int span = 0;
    for (const Variant* v : variants) {
        onlySnps = onlySnps && v->nRemoved == 1 && v->nAdded == 1;
        indel = indel || v->nRemoved != v->nAdded;
        span = std::max(span, v->nRemoved);
    }
    ALT.clear();
    if (onlySnps) {
        REF = std::string(1, fa.getCharacter(POS));
        for (const Variant* v : variants) ALT.push_back(v->added);
        return;
    }
    REF = fa.getSequence(POS, POS + span + (indel ? 1 : 0));
    for (const Variant* v : variants) {
        // Python list slice assignment: seq[a:b] = added (a, b clipped to the list)
        const size_t a = std::min<size_t>(v->nRemoved == v->nAdded ? 0 : 1, REF.size());
        const size_t b = std::min<size_t>(v->nRemoved == v->nAdded ? (size_t)v->nAdded : (size_t)(1 + v->nRemoved), REF.size());
        ALT.push_back(REF.substr(0, a) + v->added + REF.substr(std::max(a, b)));
    }
}

inline void trimLeftPadding(int& pos, std::string& ref, std::vector<std::string>& alt) {     // vcfutils.pyx:796-839
    if (alt.empty()) return;
    size_t shortest = ref.size();
    bool lengthsDiffer = false;
    for (const std::string& a : alt) { shortest = std::min(shortest, a.size()); lengthsDiffer = lengthsDiffer || a.size() != ref.size(); }
    for (size_t it = 1; it < shortest; ++it) {
        bool firstSame = true, secondSame = true, refInFirst = false, refInSecond = false, anySecond = false;
        const char f0 = (char)toupper((unsigned char)alt[0][0]);
        char s0 = 0;
        for (const std::string& a : alt) {
            const char f = (char)toupper((unsigned char)a[0]);
            if (f != f0) firstSame = false;
            if (f == (char)toupper((unsigned char)ref[0])) refInFirst = true;
            if (a.size() > 1) {
                const char s = (char)toupper((unsigned char)a[1]);
                if (!anySecond) { s0 = s; anySecond = true; }
                else if (s != s0) secondSame = false;
                if (s == ref[1]) refInSecond = true;                         // (the reference compares ref[1] as it is)
            }
        }
        if (!firstSame || !refInFirst) break;
        if (lengthsDiffer && (!secondSame || !refInSecond)) break;
        ref = ref.substr(1);
        for (std::string& a : alt) a = a.substr(1);
        ++pos;
    }
}

// one REFCALL line (outputRefCall, variantcaller.pyx:764-867; the text of VCF.write_data for its dictionary): every INFO field of the
// header missing but END and Size, QUAL as computed by the caller, per sample "./." with the number of reads in the sample's window
inline void writeRefCallLine(std::string& out, const char* chrom, int windowStart, int windowEnd, char refBase, int qual, const std::vector<int>& nReads) {
    out += chrom; out += '\t'; out += std::to_string(windowStart + 1); out += "\t.\t";
    out += refBase; out += '\t'; out += (refBase == 'N' ? 'T' : 'N'); out += '\t';
    out += std::to_string(qual);
    out += "\tREFCALL\tBRF=.;END="; out += std::to_string(windowEnd);
    out += ";FR=.;FS=.;HP=.;HapScore=.;MGOF=.;MMLQ=.;MQ=.;NF=.;NR=.;PP=.;QD=.;ReadPosRankSum=.;SC=.;START=.;SbPval=.;Size=";
    out += std::to_string(windowEnd - windowStart);
    out += ";Source=.;TC=.;TCF=.;TCR=.;TR=.;WE=.;WS=.\tGT:GL:GOF:GQ:NR:NV";
    for (int n : nReads) { out += "\t./.:-1,-1,-1:-1:-1:"; out += std::to_string(n); out += ":0"; }
    out += '\n';
}

inline int phred(double p) {                                              // vcfrecords._phred
    const double v = py2_round0(-10.0 * log10(std::max(1e-10, 1.0 - p)));
    return (int)std::min(99.0, v);
}

}  // namespace plathost
