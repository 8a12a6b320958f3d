"""The bound behind the subset a level does not run (basevarc_amd/csrc/em_items.hip site_decide, include/bvc.h "em_prune"),
checked on the CPU against the numpy restatement of the reference, which runs every subset.

Claim: at a nested level of BaseType::LRT (src/BaseType.cpp:93-110) the subset c without the deepest candidate satisfies
    chi_c = 2 (lr_alt - loglik_c)  >=  2 (lr_alt - U_c),   U_c = sum over the alleles b outside c of sum_i log(eps_i / 3)
(every marginal is at most 1; an observation of an allele outside the subset has marginal eps / 3 exactly), so when the right-hand
side exceeds the minimum chi of the level's other subsets by more than the slack, std::min_element (:99) cannot pick c and the
level's outcome does not depend on loglik_c.  Also: the C oracle's pruned counts are what this rule gives.
"""
import itertools

import numpy as np

from oracle import np_restatement as R
from oracle import orc
from tests.sitegen import random_site


def _levels(bases, quals, min_af):
    """Every nested level of the reference on one site: (candidates, deepest, [(subset, loglik, chi, passes)], lr_alt, U per subset)."""
    bases = np.asarray(bases, dtype=np.int64); quals = np.asarray(quals, dtype=np.int64)
    n = len(bases)
    eps = np.exp(R.MLN10TO10 * quals.astype(np.float64))
    lik = np.repeat((eps / 3.0)[:, None], 4, axis=1)
    lik[np.arange(n), bases] = 1.0 - eps
    depth = np.bincount(bases, minlength=4)[:4]
    cand = [b for b in range(4) if depth[b] / float(depth.sum()) >= min_af]
    lmis = np.log(eps / 3.0)

    def fit(c):
        f = np.zeros(4)
        s = sum(int(depth[b]) for b in c)
        for b in c:
            f[b] = depth[b] / s
        marg, _, p = R._em(f, lik)
        return float(np.log(marg).sum()), p

    out = []
    lr_alt, _ = fit(tuple(cand))
    while len(cand) >= 2:
        k = len(cand) - 1
        deepest = max(cand, key=lambda b: (depth[b], -cand.index(b)))
        fits = []
        for c in itertools.combinations(cand, k):
            ll, p = fit(c)
            u = float(lmis[~np.isin(bases, list(c))].sum())
            fits.append((c, ll, 2.0 * (lr_alt - ll), p, u))
        out.append((list(cand), deepest, fits, lr_alt))
        chis = [f[2] for f in fits]
        i_min = int(np.argmin(chis))                              # first minimum
        lr_alt = fits[i_min][1]
        if chis[i_min] < R.LRT_THRESHOLD:
            cand = list(fits[i_min][0])
        else:
            break
    return out


def test_the_bound_holds_and_a_ruled_out_subset_is_never_the_minimum():
    rng = np.random.default_rng(77)
    ruled_out = kept = 0
    for s in range(160):
        nind = int(rng.choice([2, 3, 5, 8, 13, 30, 100, 600, 2500]))
        af = float(rng.choice([0.0, 0.0, 0.02, 0.2, 0.5]))
        b, q, r = random_site(rng, nind, af=af, second_af=(af / 2 if s % 4 == 0 else 0.0), qlo=2, qhi=41)
        if s % 9 == 0 and nind >= 2:
            b = np.array([1, 3] * (nind // 2), dtype=np.int8); q = q[:len(b)]
        expect_skipped_fits = expect_skipped_passes = 0
        for cand, deepest, fits, lr_alt in _levels(b, q, 0.001):
            (last,) = [f for f in fits if deepest not in f[0]]
            others = [f for f in fits if deepest in f[0]]
            bound = 2.0 * (lr_alt - last[4])
            # the bound itself (rounding of two sums of |values| up to |U|)
            assert last[2] >= bound - 1e-9 * max(1.0, abs(last[4])), (s, cand, last, bound)
            if len(cand) - 1 >= 2 and bound > min(f[2] for f in others) + 1.0 + 1e-6 * abs(last[4]):
                ruled_out += 1
                expect_skipped_fits += 1; expect_skipped_passes += last[3]
                assert int(np.argmin([f[2] for f in fits])) != fits.index(last), (s, cand)
            else:
                kept += 1
        o = orc.basetype_lrt(b, q, r, 0.001)
        if o["prune_edge"] > 1e-6 and o["tie_gap"] > 1e-9 * max(1.0, abs(o["lr_alt"])):
            assert o["n_fits"] - o["n_fits_pruned"] == expect_skipped_fits, (s, o)
            assert o["n_passes"] - o["n_passes_pruned"] == expect_skipped_passes, (s, o)
    assert ruled_out > 60 and kept > 60, (ruled_out, kept)


def test_pruned_counts_of_the_two_oracle_forms_agree():
    rng = np.random.default_rng(78)
    for s in range(60):
        nind = int(rng.choice([1, 4, 20, 300, 4000]))
        b, q, r = random_site(rng, nind, af=float(rng.choice([0.0, 0.05, 0.4])))
        f = orc.basetype_lrt(b, q, r, 0.001)
        h = orc.hist_lrt(orc.dense_hist(b, q), r, 0.001)
        assert (f["n_fits_pruned"], f["n_passes_pruned"]) == (h["n_fits_pruned"], h["n_passes_pruned"]), s
        assert f["n_fits_pruned"] <= f["n_fits"] and f["n_passes_pruned"] <= f["n_passes"]
        assert (f["n_fits"] - f["n_fits_pruned"]) <= 2                       # at most one subset per EM level (n = 4: two levels)
