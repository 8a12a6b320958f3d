// Caller written the way bt_f uses BaseType (/root/reference/src/BaseVarC.cpp:612-615, 642-652),
// compiled against include/bvc_basetype.hpp.  Reads sites from stdin:
//   n ref min_af  b0 q0 b1 q1 ...        (one site per line)
// and prints one line per site: called n_alt alt... af... var_qual depth[4] | group call on the first half.
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>

#include "bvc_basetype.hpp"

int main()
{
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream in(line);
        int n, ref;
        double min_af;
        if (!(in >> n >> ref >> min_af)) continue;
        bvc::BaseV bases, quals;
        for (int i = 0; i < n; ++i) {
            int b, q;
            in >> b >> q;
            bases.push_back(static_cast<int8_t>(b));
            quals.push_back(static_cast<int8_t>(q));
        }
        bvc::BaseType bt(bases, quals, static_cast<int8_t>(ref), min_af);
        const bool bt_success = bt.LRT();
        std::printf("%d %zu", bt_success ? 1 : 0, bt.alt_bases.size());
        for (auto b : bt.alt_bases) std::printf(" %d %.17g", b, bt.af_lrt[b]);
        std::printf(" %.17g %d %d %d %d %.17g", bt.var_qual, bt.depth[0], bt.depth[1], bt.depth[2], bt.depth[3], bt.depth_total);
        // group call on the first half of the samples, candidates {ref} + alt_bases (src/BaseVarC.cpp:614-615, 642-644)
        bvc::BaseV base_comb{static_cast<int8_t>(ref)};
        base_comb.insert(base_comb.end(), bt.alt_bases.begin(), bt.alt_bases.end());
        bvc::BaseV gb(bases.begin(), bases.begin() + n / 2), gq(quals.begin(), quals.begin() + n / 2);
        if (bt_success && !gb.empty()) {
            bvc::BaseType gr_bt(gb, gq, static_cast<int8_t>(ref), min_af);
            gr_bt.SetBase(base_comb);
            gr_bt.LRT();
            std::printf(" |");
            for (auto b : bt.alt_bases) {
                if (gr_bt.af_lrt.count(b)) std::printf(" %.6f", gr_bt.af_lrt[b]);   // {:.6f}, :648
                else std::printf(" 0");                                           // literal 0, :650
            }
        }
        std::printf("\n");
    }
    return 0;
}
