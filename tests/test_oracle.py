"""CPU tests of the oracle itself (no GPU): known answers, cross-restatement agreement, edge cases.

The reference ships no golden vectors for this path and cannot be built here, so
the oracle is checked against (1) scipy / closed forms, (2) an independently written numpy
restatement, (3) the pass/fit counts SURVEY.md section 6 measured on the real reference binary.
"""
import math

import numpy as np
import pytest
from scipy import special

from oracle import np_restatement as npr
from oracle import orc
from tests.sitegen import caller_min_af, random_site


def test_kf_lgamma_known_answers():
    L = orc.lib()
    for z in (0.5, 1.0, 1.5, 2.0, 3.5, 10.0):
        assert L.orc_kf_lgamma(z) == pytest.approx(math.lgamma(z), abs=5e-13)


def test_chisf_matches_scipy_over_sweep():
    # chisf(x, 1) = Q(1/2, x/2) = erfc(sqrt(x/2)); both branches of kf_gammaq (series <= 2 < continued fraction)
    for x in np.concatenate([np.geomspace(1e-3, 1400.0, 300), [1.999999, 2.0, 2.000001, 24.0]]):
        a = orc.chisf(float(x))
        b = float(special.gammaincc(0.5, x / 2.0))
        assert a == pytest.approx(b, rel=1e-12), x
        assert a == pytest.approx(math.erfc(math.sqrt(x / 2.0)), rel=1e-12), x
    assert orc.chisf(1600.0) == 0.0          # underflow -> caller writes var_qual = 10000


SIZES = [1, 2, 3, 5, 12, 60, 500, 5000]


@pytest.mark.parametrize("nind", SIZES)
def test_c_oracle_agrees_with_numpy_restatement(nind):
    rng = np.random.default_rng(1000 + nind)
    for af in (0.0, 1e-3, 0.02, 0.2, 0.5):
        for _ in range(3 if nind <= 500 else 1):
            b, q, ref = random_site(rng, nind, af=af, qlo=2, qhi=41)
            m = caller_min_af(nind)
            a = orc.basetype_lrt(b, q, ref, m)
            p = npr.basetype_lrt(b, q, ref, m)
            assert a["status"] == 0
            assert a["called"] == p["called"]
            assert a["alt_base"] == p["alt_base"]
            assert a["depth"] == p["depth"]
            assert a["n_passes"] == p["n_passes"]
            np.testing.assert_allclose(a["af"], p["af"], rtol=0, atol=1e-10)
            if a["called"]:
                assert a["chi"] == pytest.approx(p["chi"], rel=1e-9, abs=1e-9)
                assert a["var_qual"] == pytest.approx(p["var_qual"], rel=1e-9, abs=1e-9)


@pytest.mark.parametrize("nind", SIZES + [50000])
def test_histogram_form_equals_per_sample_form(nind):
    """The (base, qual) count histogram carries all the information of the per-sample vectors."""
    rng = np.random.default_rng(7 + nind)
    for af, af2 in ((0.0, 0.0), (2e-3, 0.0), (0.05, 0.0), (0.3, 0.05)):
        b, q, ref = random_site(rng, nind, af=af, second_af=af2)
        m = caller_min_af(nind)
        a = orc.basetype_lrt(b, q, ref, m)
        h = orc.hist_lrt(orc.dense_hist(b, q), ref, m)
        assert (a["called"], a["alt_base"], a["kept"], a["depth"]) == (h["called"], h["alt_base"], h["kept"], h["depth"])
        assert a["n_fits"] == h["n_fits"] and a["n_passes"] == h["n_passes"]
        np.testing.assert_allclose(a["af"], h["af"], rtol=0, atol=1e-12)
        assert a["chi"] == pytest.approx(h["chi"], rel=1e-9, abs=1e-8)
        assert a["var_qual"] == pytest.approx(h["var_qual"], rel=1e-9, abs=1e-8)


def test_survey_measured_pass_counts():
    """SURVEY.md section 6 / BASELINE.md: on the real reference, nind = 1e4, AF 0.01, Q 10..40 gives
    10 EM fits and ~350 singleEM passes per site (350-750 at larger depth).  A weak anchor (the count
    moves with the site), but it was measured on the real binary."""
    rng = np.random.default_rng(5)
    passes = []
    for _ in range(6):
        b, q, ref = random_site(rng, 10000, af=0.01)
        r = orc.basetype_lrt(b, q, ref, caller_min_af(10000))
        assert r["n_fits"] == 10
        passes.append(r["n_passes"])
    assert 250 <= np.mean(passes) <= 650 and min(passes) >= 150 and max(passes) <= 10 * 101


# ----------------------------------------------------------------------------- edge cases
def test_zero_depth_and_empty_candidates():
    r = orc.basetype_lrt([], [], 0, 0.001)
    assert r["called"] == 0 and r["depth_total"] == 0
    # base_comb restricted to bases the group never saw -> no candidate passes min_af -> no call
    r = orc.basetype_lrt([3, 3, 3], [30, 30, 30], 0, 0.001, base_comb=[0, 1])
    assert r["called"] == 0 and r["n_fits"] == 0


def test_only_reference_base_is_not_called():
    r = orc.basetype_lrt([2] * 40, [30] * 40, 2, 0.001)
    assert r["called"] == 0 and r["kept"] == [2] and r["n_alt"] == 0


def test_mono_allelic_non_reference():
    # src/BaseType.cpp:120: one allele left, depth_total > 10 and r > 0.5 -> 5000
    r = orc.basetype_lrt([1] * 11, [30] * 11, 0, 0.001)
    assert r["called"] == 1 and r["alt_base"] == [1] and r["var_qual"] == 5000.0
    assert r["af"][0] == pytest.approx(1.0, abs=1e-12)
    # depth_total <= 10: falls through to the chi branch; the single-base site never enters the k loop -> chi = 0
    r = orc.basetype_lrt([1] * 10, [30] * 10, 0, 0.001)
    assert r["called"] == 1 and r["alt_base"] == [1] and r["var_qual"] == 0.0 and r["chi"] == 0.0


def test_all_four_bases_kept():
    b = np.repeat([0, 1, 2, 3], 50).astype(np.int8)
    q = np.full(200, 35, dtype=np.int8)
    r = orc.basetype_lrt(b, q, 0, 0.001)
    assert r["called"] == 1 and r["kept"] == [0, 1, 2, 3] and r["alt_base"] == [1, 2, 3]
    np.testing.assert_allclose(r["af"], [0.25] * 3, atol=1e-6)
    assert r["chi"] >= 24.0 and r["n_fits"] == 1 + 4


def test_tie_takes_first_minimum():
    # two ALT alleles with identical evidence: removing either gives the same chi; '<' keeps the first
    b = np.array([0] * 30 + [1] * 4 + [2] * 4, dtype=np.int8)
    q = np.full(len(b), 30, dtype=np.int8)
    r = orc.basetype_lrt(b, q, 0, 0.001)
    assert r["called"] == 1 and r["alt_base"] == [1, 2]


def test_qual_zero_on_matching_base_gives_nan_not_crash():
    # src/Algorithm.cpp:81-82: no guard for marginal == 0 -> log(0), 0/0 follow IEEE rules
    r = orc.basetype_lrt([1, 1, 1], [0, 0, 0], 0, 0.001)
    assert r["status"] == 0 and r["depth"] == [0, 3, 0, 0]
    assert r["called"] == 1 and r["alt_base"] == [1] and math.isnan(r["af"][0])
    p = npr.basetype_lrt([1, 1, 1], [0, 0, 0], 0, 0.001)
    assert p["called"] == 1 and math.isnan(p["af"][0])


def test_chi_near_threshold_and_saturation():
    rng = np.random.default_rng(11)
    seen_low = seen_high = False
    for _ in range(60):
        b, q, ref = random_site(rng, 300, af=0.012)
        r = orc.basetype_lrt(b, q, ref, caller_min_af(300))
        if r["called"] and r["var_qual"] not in (5000.0, 10000.0) and not math.isnan(r["var_qual"]):
            p = special.gammaincc(0.5, r["chi"] / 2.0)
            assert r["var_qual"] == pytest.approx(-10 * math.log10(p), rel=1e-10)
            seen_low |= r["chi"] < 60
            seen_high |= r["chi"] >= 24
    assert seen_low and seen_high
    b, q, ref = random_site(rng, 20000, af=0.4)
    r = orc.basetype_lrt(b, q, ref, caller_min_af(20000))
    assert r["called"] == 1 and r["var_qual"] == 10000.0 and r["chi"] > 1500


def test_group_loop_literal_zero_when_group_lacks_alt():
    rng = np.random.default_rng(3)
    n = 600
    b, q, ref = random_site(rng, n, af=0.0, qlo=30, qhi=40)
    grp = (np.arange(n) % 3).astype(np.uint8)
    alt = (ref + 1) % 4
    b[(grp == 0) & (np.arange(n) < 200)] = alt          # ALT only in group 0
    grp[-10:] = 255                                      # ungrouped samples
    o, gd, ga, ran, _ = orc.dense_site_groups(b, q, ref, caller_min_af(n), grp, 3)
    assert o["called"] == 1 and o["alt_base"] == [alt]
    assert ran.tolist() == [1, 1, 1]
    assert ga[0, 0] > 0.2 and ga[1, 0] == 0.0 and ga[2, 0] == 0.0
    assert gd.sum() == n - 10
    oh, gdh, gah, _, _ = orc.dense_site_groups(b, q, ref, caller_min_af(n), grp, 3, use_hist=True)
    assert np.array_equal(gd, gdh)
    np.testing.assert_allclose(ga, gah, atol=1e-12)


# ----------------------------------------------------------------------------- synthetic generator
def test_synth_generator_is_deterministic_and_plausible():
    b1, q1, r1 = orc.synth_tile(1, 100, 6, 20000)
    b2, q2, r2 = orc.synth_tile(1, 102, 2, 20000)
    assert np.array_equal(b1[2:4], b2) and np.array_equal(q1[2:4], q2) and np.array_equal(r1[2:4], r2)
    assert q1.min() == 10 and q1.max() == 40 and b1.min() >= 0 and b1.max() <= 3
    b3, _, _ = orc.synth_tile(2, 100, 6, 20000)
    assert not np.array_equal(b1, b3)
    # error rate ~ mean(10^(-Q/10)) over Q = 10..40 on monomorphic sites
    mono = [s for s in range(6) if (b1[s] == r1[s]).mean() > 0.9]
    assert mono
    exp_err = np.mean(10.0 ** (-np.arange(10, 41) / 10.0))
    obs_err = np.mean([(b1[s] != r1[s]).mean() for s in mono])
    assert obs_err == pytest.approx(exp_err, rel=0.25)
    bs, qs, _ = orc.synth_tile(1, 0, 2, 20000, cov_thr16=6554)      # ~10 % coverage
    assert 0.07 < (bs >= 0).mean() < 0.13 and (bs[bs < 0] == -1).all()


def test_synth_site_mixture():
    _, _, _ = orc.synth_tile(1, 0, 1, 16)
    poly = 0
    for s in range(200):
        b, _, r = orc.synth_tile(1, s, 1, 4000)
        poly += (b[0] != r[0]).mean() > 0.05
    assert 5 <= poly <= 60          # 20 % polymorphic, AF log-uniform: a minority shows AF > ~4 %


def test_compensated_sums_expose_the_drift_of_the_per_sample_sum():
    """the reference adds N per-sample logs one by one in a double; that sum drifts with N
    while the histogram sum of <= 512 terms does not.  ORC_MODE_COMPENSATED keeps the per-sample loops but
    accumulates in long double: it must land on the histogram form (to rounding of the final doubles), and the
    faithful form's distance from both must be the same number -- i.e. the drift is a property of the reference's
    accumulator, not of regrouping equal terms."""
    n = 200_000
    b, q, r = orc.synth_tile(1, 0, 3, n)
    m = caller_min_af(n)
    for s in range(3):
        cnt = orc.dense_hist(b[s], q[s])
        f = orc.basetype_lrt(b[s], q[s], r[s], m)
        c = orc.basetype_lrt(b[s], q[s], r[s], m, compensated=True)
        h = orc.hist_lrt(cnt, r[s], m)
        hc = orc.hist_lrt(cnt, r[s], m, compensated=True)
        ulp_ll = abs(f["lr_alt"]) * 2.0 ** -52
        assert (f["n_fits"], f["n_passes"]) == (c["n_fits"], c["n_passes"]) == (h["n_fits"], h["n_passes"])
        # compensated per-sample == histogram form up to a few ulp of the log-likelihoods chi is a difference of
        assert abs(c["chi"] - h["chi"]) <= 128 * ulp_ll, (c["chi"], h["chi"])
        assert abs(hc["chi"] - h["chi"]) <= 128 * ulp_ll      # <= 512 class terms, each rounded once
        assert abs(hc["chi"] - c["chi"]) <= 128 * ulp_ll
        # the faithful sum is farther from both than they are from each other, and within the N*u*|loglik| bound
        drift = abs(f["chi"] - c["chi"])
        assert drift <= 2 * n * 2.0 ** -53 * abs(f["lr_alt"])
        assert abs(abs(f["chi"] - h["chi"]) - drift) <= 128 * ulp_ll
        assert drift > 1000 * ulp_ll                        # ... and it is far above the rounding of one sum
        np.testing.assert_allclose(c["af"], h["af"], rtol=0, atol=1e-13)
