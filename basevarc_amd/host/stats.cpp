#include "stats.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>

namespace bvchost {

// ---- htslib kfunc.c restatements ---------------------------------------------------------------------
// Attribution: kf_erfc, kf_lgamma / kf_gammaq and kt_fisher_exact (hypergeo_acc) below restate, nearly statement for statement,
// the numerical routines of htslib's kfunc.c (https://github.com/samtools/htslib, MIT/Expat licence, (c) Genome Research Ltd. and
// Attractive Chaos), which the reference links through SeqLib (src/Algorithm.cpp:3-25; .gitmodules:1-3, submodule absent from the
// reference tree).  The arithmetic has to be htslib's for the outputs to match the reference's; the constants are htslib's.
// erfc by W. J. Cody / Hart-style rational approximation as published in kfunc.c (kf_erfc): a degree-6 over
// degree-7 rational in z = |x| sqrt(2) for z < 10/sqrt(2), a continued fraction beyond, 0/2 past z = 37.
double kf_erfc(double x)
{
    static const double p[7] = {220.2068679123761, 221.2135961699311, 112.0792914978709, 33.912866078383,
                                6.37396220353165, .7003830644436881, .03526249659989109};
    static const double q[8] = {440.4137358247522, 793.8265125199484, 637.3336333788311, 296.5642487796737,
                                86.78073220294608, 16.06417757920695, 1.755667163182642, .08838834764831844};
    const double z = std::fabs(x) * M_SQRT2;
    if (z > 37.) return x > 0. ? 0. : 2.;
    const double expntl = std::exp(z * z * -.5);
    double r;
    if (z < 10. / M_SQRT2) {
        double num = p[6], den = q[7];
        for (int k = 5; k >= 0; --k) num = num * z + p[k];
        for (int k = 6; k >= 0; --k) den = den * z + q[k];
        r = expntl * num / den;
    } else {
        r = expntl / 2.506628274631001 / (z + 1. / (z + 2. / (z + 3. / (z + 4. / (z + .65)))));
    }
    return x > 0. ? 2. * r : 2. * (1. - r);
}

namespace {

double lbinom(int n, int k)
{
    if (k == 0 || n == k) return 0;
    return std::lgamma(n + 1) - std::lgamma(k + 1) - std::lgamma(n - k + 1);
}

// P(n11 | margins) of the hypergeometric distribution
double hypergeo(int n11, int n1_, int n_1, int n)
{
    return std::exp(lbinom(n1_, n11) + lbinom(n - n1_, n_1 - n11) - lbinom(n, n_1));
}

struct HgAcc { int n11, n1_, n_1, n; double p; };

// incremental evaluation along n11 with the margins fixed (kfunc.c's hypergeo_acc: every 11th value and the
// boundary are recomputed from scratch to stop error growth)
double hypergeo_acc(int n11, int n1_, int n_1, int n, HgAcc *a)
{
    if (n1_ || n_1 || n) {
        a->n11 = n11; a->n1_ = n1_; a->n_1 = n_1; a->n = n;
    } else {
        if (n11 % 11 && n11 + a->n - a->n1_ - a->n_1) {
            if (n11 == a->n11 + 1) {
                a->p *= (double)(a->n1_ - a->n11) / n11 * (a->n_1 - a->n11) / (n11 + a->n - a->n1_ - a->n_1);
                a->n11 = n11;
                return a->p;
            }
            if (n11 == a->n11 - 1) {
                a->p *= (double)a->n11 / (a->n1_ - n11) * (a->n11 + a->n - a->n1_ - a->n_1) / (a->n_1 - n11);
                a->n11 = n11;
                return a->p;
            }
        }
        a->n11 = n11;
    }
    a->p = hypergeo(a->n11, a->n1_, a->n_1, a->n);
    return a->p;
}

}  // namespace

double kt_fisher_exact(int n11, int n12, int n21, int n22, double *_left, double *_right, double *two)
{
    HgAcc aux;
    const int n1_ = n11 + n12, n_1 = n11 + n21, n = n11 + n12 + n21 + n22;
    const int max = (n_1 < n1_) ? n_1 : n1_;          // largest possible n11
    int min = n1_ + n_1 - n;                          // smallest possible n11
    if (min < 0) min = 0;
    *two = *_left = *_right = 1.;
    if (min == max) return 1.;
    const double q = hypergeo_acc(n11, n1_, n_1, n, &aux);      // probability of the observed table
    int i, j;
    double p, left, right;
    p = hypergeo_acc(min, 0, 0, 0, &aux);
    for (left = 0., i = min + 1; p < 0.99999999 * q && i <= max; ++i) left += p, p = hypergeo_acc(i, 0, 0, 0, &aux);
    --i;
    if (p < 1.00000001 * q) left += p;
    else --i;
    p = hypergeo_acc(max, 0, 0, 0, &aux);
    for (right = 0., j = max - 1; p < 0.99999999 * q && j >= 0; --j) right += p, p = hypergeo_acc(j, 0, 0, 0, &aux);
    ++j;
    if (p < 1.00000001 * q) right += p;
    else ++j;
    *two = left + right;
    if (*two > 1.) *two = 1.;
    if (std::abs(i - n11) < std::abs(j - n11)) right = 1. - left + q;
    else left = 1.0 - right + q;
    *_left = left;
    *_right = right;
    return q;
}

// ---- reference's own wrappers ------------------------------------------------------------------------
double normsf(double x) { return kf_erfc(x / std::sqrt(2.0)) / 2.0; }

double bt_fisher_exact(int n11, int n12, int n21, int n22)
{
    double l, r, two;
    kt_fisher_exact(n11, n12, n21, n22, &l, &r, &two);
    double p = -10 * std::log10(two);
    if (std::isinf(p)) p = 10000.0;
    else if (p == 0) p = 0.0;
    return p;
}

// Rank sum of the first sample in the pooled data (what rankR1 returns, src/Algorithm.cpp:27-53): values ranked in
// DESCENDING order (the reference sorts with '>', src/BaseVarUtils.h:31-36, so rank 1 is the largest), every run of
// equal values sharing the mean of the ranks it spans.  Computed run by run: a run occupying ranks lo..hi gives each
// of its members (lo + hi) / 2, and only the members that came from the first sample (pooled index < n1) are summed.
static double rank_r1(const std::vector<double> &x, size_t n1)
{
    const size_t n = x.size();
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&x](size_t a, size_t b) { return x[a] > x[b]; });
    double r1 = 0.0;
    for (size_t lo = 0; lo < n;) {
        size_t hi = lo;                                         // [lo, hi] = one run of equal values (0-based places)
        size_t from_first = order[lo] < n1 ? 1 : 0;
        while (hi + 1 < n && x[order[hi + 1]] == x[order[lo]]) { ++hi; from_first += order[hi] < n1 ? 1 : 0; }
        // ranks lo+1 .. hi+1: their sum is an integer, divided once by the run length as the reference does
        // (src/Algorithm.cpp:41: static_cast<double>(k) / (n + 1)).  DELIBERATE DIVERGENCE: the reference keeps that sum
        // in an int32_t `k` (:31), which wraps -- undefined behaviour -- once a run of equal values is longer than about
        // 65,536 (sum of its ranks > 2^31); at CMDB depth (1e5 observations per site, most with mapping quality 60) its
        // *RankSum fields are then garbage.  Here the sum is exact (size_t), so the two agree wherever the reference is
        // defined and this one stays the rank-sum statistic beyond (tests/test_host.py).
        const size_t len = hi - lo + 1;
        const double shared = len == 1 ? (double)(lo + 1) : (double)((lo + 1 + hi + 1) * len / 2) / (double)len;
        for (size_t t = 0; t < from_first; ++t) r1 += shared;   // added one member at a time, like the reference's loop
        lo = hi + 1;
    }
    return r1;
}

double RankSumTest(std::vector<double> &x, std::vector<double> &y)
{
    const size_t n1 = x.size(), n2 = y.size();
    x.insert(x.end(), y.begin(), y.end());
    const double r1 = rank_r1(x, n1);
    const double expected = (double)(n1 * (n1 + n2 + 1)) / 2.0;
    const double z = (r1 - expected) / std::sqrt((double)(n1 * n2 * (n1 + n2 + 1)) / 12.0);
    double p = -10 * std::log10(2 * normsf(std::abs(z)));
    if (std::isinf(p)) p = 10000.0;
    else if (p == 0) p = 0.0;
    return p;
}

}  // namespace bvchost
