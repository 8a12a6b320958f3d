// stats.h -- annotation statistics used by the CVG/VCF writers (host side, scalar).
// Counterparts of /root/reference/src/Algorithm.cpp:9-67 (normsf, bt_fisher_exact, rankR1, RankSumTest).
// The reference takes kf_erfc and kt_fisher_exact from htslib's kfunc.c (SeqLib submodule, absent from the
// reference tree, version unpinned); both are restated here from the published algorithms.
#ifndef BVC_HOST_STATS_H
#define BVC_HOST_STATS_H

#include <vector>

namespace bvchost {

double kf_erfc(double x);                                     // complementary error function (Hart, ~1e-15)
double kt_fisher_exact(int n11, int n12, int n21, int n22, double *left, double *right, double *two);

double normsf(double x);                                      // src/Algorithm.cpp:9-14
double bt_fisher_exact(int n11, int n12, int n21, int n22);   // src/Algorithm.cpp:16-25: phred of the two-sided p
double RankSumTest(std::vector<double> &x, std::vector<double> &y);   // src/Algorithm.cpp:55-67 (x is extended by y)

}  // namespace bvchost
#endif
