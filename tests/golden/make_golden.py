#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (run from the repo root: python -m tests.golden.make_golden).

Sets follow SURVEY.md 8c: (1) random sites, nind in {1,2,3,5,12,60,500,5e3,5e4}, AF in {0, 1e-3 .. 0.5},
Q in [2,41]; (2) hand-built edge cases.
"""
import numpy as np

from oracle import orc
from tests.golden.golden_io import save_golden
from tests.sitegen import caller_min_af, random_site


def random_set():
    rng = np.random.default_rng(20261004)
    sites, mafs = [], []
    for nind, reps in ((1, 8), (2, 8), (3, 8), (5, 8), (12, 8), (60, 6), (500, 4), (5000, 2), (50000, 1)):
        for af, af2 in ((0.0, 0.0), (1e-3, 0.0), (5e-3, 0.0), (0.02, 0.0), (0.1, 0.0), (0.3, 0.05), (0.5, 0.0)):
            for _ in range(reps if nind < 50000 else (1 if af in (0.0, 0.02) else 0)):
                sites.append(random_site(rng, nind, af=af, second_af=af2, qlo=2, qhi=41))
                mafs.append(caller_min_af(nind))
    return sites, mafs


def edge_set():
    A = lambda x: np.array(x, dtype=np.int8)
    cases = [
        (A([]), A([]), 0, 0.001),
        (A([2] * 40), A([30] * 40), 2, 0.001),
        (A([1] * 11), A([30] * 11), 0, 0.001),
        (A([1] * 10), A([30] * 10), 0, 0.001),
        (A(np.repeat([0, 1, 2, 3], 50)), A([35] * 200), 0, 0.001),
        (A([0] * 30 + [1] * 4 + [2] * 4), A([30] * 38), 0, 0.001),
        (A([1, 1, 1]), A([0, 0, 0]), 0, 0.001),
        (A([0, 0, 0, 1]), A([0, 5, 0, 0]), 0, 0.001),
        (A([3]), A([40]), 3, 0.001), (A([3]), A([40]), 0, 0.001),
        (A([0, 1]), A([93, 93]), 0, 0.001),
        (A([0] * 5 + [1] * 5), A([127] * 10), 0, 0.001),
        (A([0] * 995 + [1] * 5), A([30] * 1000), 0, 0.01),      # ALT below min_af: filtered before the EM
        (A([0] * 990 + [1] * 10), A([30] * 1000), 0, 0.01),     # ALT exactly at min_af: kept (>=)
        (A([0] * 50 + [1] * 50), A([20] * 100), 2, 0.001),      # reference base never observed: two ALTs
    ]
    rng = np.random.default_rng(7)
    for _ in range(12):                                          # chi around the threshold 24
        b, q, r = random_site(rng, 300, af=0.012)
        cases.append((b, q, r, caller_min_af(300)))
    return [(b, q, r) for b, q, r, _ in cases], [m for _, _, _, m in cases]


def main():
    for name, (sites, mafs) in (("basetype_random.npz", random_set()), ("basetype_edge.npz", edge_set())):
        exp = [orc.basetype_lrt(b, q, r, m) for (b, q, r), m in zip(sites, mafs)]
        save_golden(name, sites, mafs, exp)
        print(name, len(sites), "sites,", sum(len(b) for b, _, _ in sites), "observations,",
              sum(e["called"] for e in exp), "called")


if __name__ == "__main__":
    main()
