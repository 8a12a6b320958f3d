"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
include/bvc.h, against the CPU oracle on the same seeded inputs.

Bars: class counts, depths, calls, ALT lists and EM pass counts bit-exact; AF within 1e-6 absolute;
chi / var_qual within 1e-6 relative (BASELINE.json north_star: "identical ref/alt calls and AF/LRT
within 1e-6").
"""
import math

import numpy as np
import pytest

from oracle import orc
from tests.sitegen import caller_min_af, random_site

pytestmark = pytest.mark.gpu

AF_ATOL = 1e-6
QUAL_RTOL = 1e-6


@pytest.fixture(scope="module")
def ctx():
    from basevarc_amd import Context
    c = Context(0)
    yield c
    c.close()


def assert_site_matches(rec, exp, where="", path_strict=True, faithful_drift=False, counts="default"):
    """rec: numpy record of bvc_site_result; exp: oracle dict.

    path_strict also compares the diagnostics n_fits / n_passes.  They are bit-exact except when two
    subsets of a level tie in chi to rounding (two pure-error alleles whose fitted frequencies both reach
    ~1e-17: which one the nested test drops first is decided by the last bits of a sum over samples).
    Either order ends in the same model, calls and AF; only the diagnostics differ.  The tile-sized tests
    pass path_strict=False and hand every such site to assert_path_difference_is_a_tie.

    faithful_drift: only for comparisons with the FAITHFUL per-sample oracle at N >= 1e5.  chi is a difference
    of two log-likelihoods; the reference adds N per-sample logs one by one in a double, which drifts by up to
    N*u*|loglik| (measured 2.4e-6 on chi at N = 1e6), while a sum of <= 512 class terms does not.  That the drift
    is the reference's accumulator and nothing else is shown by test_chi_at_1e6_matches_the_compensated_oracle
    (GPU == per-sample oracle with long-double sums to 1e-9) and tests/test_oracle.py (CPU, N = 2e5).  With the
    flag the absolute floor grows by 2e-10*|loglik| (about 1.6e-5 at N = 1e6); without it the floor is the
    north_star's 1e-6."""
    assert int(rec["status"]) == exp["status"], where
    assert [int(x) for x in rec["depth"]] == exp["depth"], where
    assert float(rec["depth_total"]) == exp["depth_total"], where
    assert int(rec["called"]) == exp["called"], where
    assert int(rec["n_alt"]) == exp["n_alt"], where
    assert [int(rec["alt_base"][i]) for i in range(rec["n_alt"])] == exp["alt_base"], where
    assert [int(rec["kept"][i]) for i in range(rec["n_kept"])] == exp["kept"], where
    if path_strict:
        assert path_counts_match(rec, exp, counts), (where, int(rec["n_fits"]), int(rec["n_passes"]), exp)
    for i in range(exp["n_alt"]):
        a, b = float(rec["af"][i]), exp["af"][i]
        assert (math.isnan(a) and math.isnan(b)) or abs(a - b) <= AF_ATOL, (where, a, b)
    floor = 1e-6 + (2e-10 * abs(exp["lr_alt"]) if faithful_drift else 0.0)
    for name in ("chi", "var_qual"):
        a, b = float(rec[name]), exp[name]
        if math.isnan(b):
            assert math.isnan(a), (where, name, a, b)
        else:
            assert a == pytest.approx(b, rel=QUAL_RTOL, abs=floor), (where, name)


# What the item engine of stage 2 leaves to the one-wavefront-per-site engine (basevarc_amd/csrc/em_items.hip site_classes: more
# than kWide = 48 quality values on an allele, an observation of quality 0 or 1, a SetBase list that repeats a base).
ITEM_ENGINE_MAX_QUALS = 48


def item_engine_takes(exp):
    return exp["max_quals"] <= ITEM_ENGINE_MAX_QUALS and exp["min_qual"] >= 2 and not exp["dup_candidate"]


def path_counts_match(rec, exp, counts="default"):
    """n_fits / n_passes count the EM() calls and passes the library RAN.  `counts` says which pair the record must hold:

    "reference": the reference's own counts, strictly -- the wave engine (em_kernel.hip: em_engine = 1, and the sites the item
                 engine leaves to it) and the item engine with em_prune = 0 run what the reference runs.
    "default":   the library as a new context runs it.  A site the item engine takes must hold the PRUNED pair: the engine does
                 not run a level's subset without the deepest allele when a bound rules it out as the level's minimum
                 (include/bvc.h "em_prune"); the oracle applies the same test to its own sums and reports those counts beside the
                 reference's (n_fits_pruned / n_passes_pruned), with the distance of the closest such test from its threshold
                 (prune_edge: at rounding level either outcome of the test is legitimate, so either pair is).  A site the item
                 engine leaves to the wave engine (item_engine_takes) must hold the reference's pair.
    The pruned pair mirrors the product's own rule (a self-comparison for these two diagnostics); that a
    record's reference-defined fields do not depend on it is what tests/test_gpu_round4.py checks byte for byte."""
    got = (int(rec["n_fits"]), int(rec["n_passes"]))
    ref = (exp["n_fits"], exp["n_passes"])
    if counts == "reference" or not item_engine_takes(exp):
        return got == ref
    assert counts == "default", counts
    pruned = (exp["n_fits_pruned"], exp["n_passes_pruned"])
    if exp["prune_edge"] < 1e-6:
        return got in (ref, pruned)
    return got == pruned


# A tie: the runner-up subset of some nested level is within this many ulps of the log-likelihood of the best one
# (chi = 2 * (lr_alt - lr): a difference of sums of up to 512 rounded terms; the GPU and the CPU evaluate log() with
# different last-bit errors and add in different orders).
TIE_ULPS = 4096


def assert_path_difference_is_a_tie(rec, exp, where="", faithful_n=0, counts="default"):
    """n_passes / n_fits may differ from the oracle's only where std::min_element's choice among
    the subsets of a level (src/BaseType.cpp:99) hangs on rounding.  For a site whose diagnostics differ, the oracle's
    own record must show such a tie; the results proper were already compared by assert_site_matches.
    faithful_n: the oracle record comes from the faithful per-sample form over that many samples, whose log-likelihoods
    (and so its tie gap) carry the drift of a one-by-one double sum, up to N*u*|loglik| each."""
    if path_counts_match(rec, exp, counts):
        return 0
    tol = (TIE_ULPS * 2.0 ** -52 + 2 * faithful_n * 2.0 ** -53) * max(1.0, abs(exp["lr_alt"]))
    assert exp["tie_gap"] <= tol, (where, "pass count differs without a tie", int(rec["n_passes"]), exp["n_passes"],
                                   exp["tie_gap"], tol)
    return 1


def pad_rows(sites, width=None, fill=-1):
    width = width or max(len(b) for b, _, _ in sites)
    B = np.full((len(sites), width), fill, dtype=np.int8)
    Q = np.zeros((len(sites), width), dtype=np.int8)
    R = np.zeros(len(sites), dtype=np.int8)
    for s, (b, q, r) in enumerate(sites):
        B[s, :len(b)] = b
        Q[s, :len(q)] = q
        R[s] = r
    return B, Q, R


# ------------------------------------------------------------------ stage 1: histogram, bit-exact
@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 4096, 8192 + 5, 70000, 262144 + 48])
def test_hist_exact(ctx, n):
    rng = np.random.default_rng(n)
    ns = 5
    B = rng.integers(0, 4, (ns, n)).astype(np.int8)
    Q = rng.integers(0, 128, (ns, n)).astype(np.int8)
    # uncovered / invalid bytes of every kind in some rows: base 4 (N), -1, 127; negative quals
    for s, frac in enumerate([0.0, 0.001, 0.5, 0.999, 0.0]):
        m = rng.random(n) < frac
        B[s, m] = rng.choice(np.array([4, -1, 127, -128], dtype=np.int8), m.sum())
        m2 = rng.random(n) < frac / 2
        Q[s, m2] = rng.integers(-128, 0, m2.sum())
    got = ctx.hist_dense(B, Q)
    for s in range(ns):
        exp = orc.dense_hist(B[s], Q[s])
        assert np.array_equal(got[s], exp), (n, s)


def test_hist_exact_skewed_keys(ctx):
    """Real pileups: >90 % one base, few quality values -- the worst case for histogram contention."""
    n = 300000
    B = np.zeros((3, n), dtype=np.int8)
    Q = np.full((3, n), 37, dtype=np.int8)
    B[1, ::1000] = 2
    Q[2, :] = np.where(np.arange(n) % 7 == 0, 11, 37)
    got = ctx.hist_dense(B, Q)
    for s in range(3):
        assert np.array_equal(got[s], orc.dense_hist(B[s], Q[s]))
    assert got[0, 37] == n


# ------------------------------------------------------------------ whole path vs faithful oracle
@pytest.mark.parametrize("nind", [1, 2, 3, 5, 12, 60, 500, 5000])
def test_lrt_matches_oracle_random_sites(ctx, nind):
    rng = np.random.default_rng(100 + nind)
    sites = []
    for af, af2 in ((0.0, 0.0), (1e-3, 0.0), (0.01, 0.0), (0.05, 0.0), (0.2, 0.02), (0.5, 0.1)):
        for _ in range(4 if nind <= 500 else 2):
            sites.append(random_site(rng, nind, af=af, second_af=af2, qlo=2, qhi=41))
    B, Q, R = pad_rows(sites)
    m = caller_min_af(nind)
    got = ctx.lrt_dense(B, Q, R, m)
    for s, (b, q, r) in enumerate(sites):
        assert_site_matches(got[s], orc.basetype_lrt(b, q, r, m), where=f"nind={nind} site={s}")


@pytest.mark.parametrize("qlo,qhi,nind", [(1, 127, 3000), (0, 127, 3000), (20, 90, 40000), (2, 60, 800), (30, 33, 5000)])
def test_lrt_wide_quality_ranges(ctx, qlo, qhi, nind):
    """More than 32 / 64 distinct quality values per base: the NS = 4 and NS = 8 kernel variants."""
    rng = np.random.default_rng(qhi * 1000 + nind)
    sites = [random_site(rng, nind, af=af, second_af=af2, qlo=qlo, qhi=qhi)
             for af, af2 in ((0.0, 0.0), (0.01, 0.0), (0.1, 0.0), (0.3, 0.1), (0.0, 0.0), (0.05, 0.0))]
    B, Q, R = pad_rows(sites)
    m = caller_min_af(nind)
    got = ctx.lrt_dense(B, Q, R, m)
    for s, (b, q, r) in enumerate(sites):
        assert_site_matches(got[s], orc.basetype_lrt(b, q, r, m), where=f"q{qlo}-{qhi} nind={nind} site={s}",
                            path_strict=(qlo > 0))


def test_lrt_edge_cases(ctx):
    cases = [
        ([], [], 0),                                             # depth_total == 0
        ([2] * 40, [30] * 40, 2),                                # only the reference base
        ([1] * 11, [30] * 11, 0),                                # mono-allelic non-ref, depth > 10 -> 5000
        ([1] * 10, [30] * 10, 0),                                # depth <= 10 -> chi branch, chi = 0
        (list(np.repeat([0, 1, 2, 3], 50)), [35] * 200, 0),      # all four bases kept
        ([0] * 30 + [1] * 4 + [2] * 4, [30] * 38, 0),            # tie in chi: first minimum
        ([1, 1, 1], [0, 0, 0], 0),                               # Q = 0 on a matching base: NaN propagates
        ([0, 0, 0, 1], [0, 5, 0, 0], 0),                         # Q = 0 mixed with a second base
        ([3], [40], 3), ([3], [40], 0),                          # single observation
        ([0, 1], [93, 93], 0),                                   # highest BAM quality
        ([0] * 5 + [1] * 5, [127] * 10, 0),                      # highest int8 quality
    ]
    sites = [(np.array(b, dtype=np.int8), np.array(q, dtype=np.int8), r) for b, q, r in cases]
    B, Q, R = pad_rows(sites, width=256)
    got = ctx.lrt_dense(B, Q, R, 0.001)
    for s, (b, q, r) in enumerate(sites):
        assert_site_matches(got[s], orc.basetype_lrt(b, q, r, 0.001), where=f"edge case {s}")


def test_chi_sweep_and_saturation(ctx):
    """var_qual over the whole chi range: below 24, near 24, large, and the 10000 saturation."""
    rng = np.random.default_rng(5)
    sites = []
    for nind, af in ((300, 0.012), (300, 0.02), (2000, 0.01), (20000, 0.05), (20000, 0.4), (60000, 0.3)):
        for _ in range(6 if nind <= 2000 else 2):
            sites.append(random_site(rng, nind, af=af))
    B, Q, R = pad_rows(sites)
    got = ctx.lrt_dense(B, Q, R, 0.001)
    seen = set()
    for s, (b, q, r) in enumerate(sites):
        exp = orc.basetype_lrt(b, q, r, 0.001)
        assert_site_matches(got[s], exp, where=f"chi sweep {s}")
        if exp["called"]:
            seen.add("sat" if exp["var_qual"] == 10000.0 else ("mid" if exp["var_qual"] < 5000 else "other"))
    assert {"sat", "mid"} <= seen


def test_set_base_and_min_af_filter(ctx):
    """SetBase-restricted candidate lists (the group call) and the min_af filter, through bvc_lrt_hist."""
    rng = np.random.default_rng(9)
    counts, refs, combs, ncs, exps = [], [], [], [], []
    for _ in range(24):
        nind = int(rng.choice([30, 400, 3000]))
        b, q, ref = random_site(rng, nind, af=float(rng.choice([0.0, 0.05, 0.3])), second_af=0.03)
        k = int(rng.integers(1, 5))
        comb = [ref] + [x for x in rng.permutation(4) if x != ref][:k - 1]
        m = float(rng.choice([0.001, 0.02, 0.2]))
        counts.append(orc.dense_hist(b, q)); refs.append(ref)
        combs.append(comb + [0] * (4 - len(comb))); ncs.append(len(comb))
        exps.append((orc.basetype_lrt(b, q, ref, m, base_comb=comb), m))
    for m in (0.001, 0.02, 0.2):
        idx = [i for i, e in enumerate(exps) if e[1] == m]
        got = ctx.lrt_hist(np.array([counts[i] for i in idx]), [refs[i] for i in idx], m,
                           np.array([combs[i] for i in idx], dtype=np.int8), [ncs[i] for i in idx])
        for j, i in enumerate(idx):
            assert_site_matches(got[j], exps[i][0], where=f"setbase {i}")


def test_min_af_zero_follows_the_reference_quirks(ctx):
    """--maf 0: zero-depth bases pass the filter, UpdateF skips zero-coverage subsets while
    `bc` keeps them (the index shift of src/BaseType.cpp:54 vs :104), and a candidate list without any coverage
    makes the reference index an empty vector -- reported as status = 1, no call."""
    rng = np.random.default_rng(41)
    cases = []
    for _ in range(40):
        nind = int(rng.choice([6, 40, 500]))
        b, q, ref = random_site(rng, nind, af=float(rng.choice([0.0, 0.1, 0.4])), qlo=20, qhi=40)
        present = sorted(set(int(x) for x in b))
        k = int(rng.integers(1, 5))
        comb = [int(x) for x in rng.permutation(4)[:k]]
        cases.append((b, q, ref, comb))
    cases.append((np.array([3, 3, 3], dtype=np.int8), np.array([30, 30, 30], dtype=np.int8), 0, [0, 1]))   # no coverage at all
    counts = np.array([orc.dense_hist(b, q) for b, q, _, _ in cases])
    combs = np.array([c + [0] * (4 - len(c)) for _, _, _, c in cases], dtype=np.int8)
    got = ctx.lrt_hist(counts, [r for _, _, r, _ in cases], 0.0, combs, [len(c) for _, _, _, c in cases])
    n_status = 0
    for s, (b, q, r, comb) in enumerate(cases):
        e = orc.basetype_lrt(b, q, r, 0.0, base_comb=comb)
        assert_site_matches(got[s], e, where=f"maf0 case {s} comb={comb}", path_strict=False)
        n_status += e["status"]
    assert n_status >= 1 and int(got[-1]["status"]) == 1 and int(got[-1]["called"]) == 0


def test_basetype_facade_reads_like_the_reference(ctx):
    from basevarc_amd import BaseType
    rng = np.random.default_rng(2)
    b, q, ref = random_site(rng, 800, af=0.1)
    bt = BaseType(b, q, ref, 0.001, ctx=ctx)
    ok = bt.LRT()
    exp = orc.basetype_lrt(b, q, ref, 0.001)
    assert ok == bool(exp["called"]) and bt.alt_bases == exp["alt_base"]
    assert bt.depth == {j: exp["depth"][j] for j in range(4)} and bt.depth_total == exp["depth_total"]
    for a, f in zip(exp["alt_base"], exp["af"]):
        assert bt.af_lrt[a] == pytest.approx(f, abs=AF_ATOL)
    assert bt.var_qual == pytest.approx(exp["var_qual"], rel=QUAL_RTOL)
    gr = BaseType(b[:300], q[:300], ref, 0.001, ctx=ctx)
    gr.SetBase([ref] + bt.alt_bases)
    gr.LRT()
    expg = orc.basetype_lrt(b[:300], q[:300], ref, 0.001, base_comb=[ref] + exp["alt_base"])
    assert gr.alt_bases == expg["alt_base"]
    with pytest.raises(RuntimeError):
        gr.LRT()


def test_cpp_facade_caller(tmp_path):
    """A bt_f-style C++ caller built on include/bvc_basetype.hpp, run as its own process (system HIP runtime)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "basevarc_amd")
    exe = tmp_path / "facade_demo"
    subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "facade_demo.cpp"), "-L", libdir, "-lbvc",
                           "-Wl,-rpath," + libdir, "-o", str(exe)])
    rng = np.random.default_rng(77)
    sites = [random_site(rng, n, af=af) for n, af in ((40, 0.2), (400, 0.05), (3000, 0.0), (3000, 0.3), (1, 0.0))]
    text = "".join(f"{len(b)} {r} 0.001 " + " ".join(f"{x} {y}" for x, y in zip(b, q)) + "\n" for b, q, r in sites)
    out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == len(sites)
    for line, (b, q, r) in zip(out, sites):
        e = orc.basetype_lrt(b, q, r, 0.001)
        main, _, grp = line.partition("|")
        f = main.split()
        assert int(f[0]) == e["called"] and int(f[1]) == e["n_alt"]
        for i in range(e["n_alt"]):
            assert int(f[2 + 2 * i]) == e["alt_base"][i]
            assert float(f[3 + 2 * i]) == pytest.approx(e["af"][i], abs=AF_ATOL)
        k = 2 + 2 * e["n_alt"]
        assert float(f[k]) == pytest.approx(e["var_qual"], rel=QUAL_RTOL, abs=1e-6)
        assert [int(x) for x in f[k + 1:k + 5]] == e["depth"]
        if e["called"] and len(b) // 2 > 0:
            h = len(b) // 2
            g = orc.basetype_lrt(b[:h], q[:h], r, 0.001, base_comb=[r] + e["alt_base"])
            want = [(f"{g['af'][g['alt_base'].index(a)]:.6f}" if a in g["alt_base"] else "0") for a in e["alt_base"]]
            assert grp.split() == want


def test_csr_matches_dense(ctx):
    rng = np.random.default_rng(21)
    sites = [random_site(rng, int(n), af=0.05) for n in rng.integers(0, 900, 40)]
    offs = np.zeros(len(sites) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(b) for b, _, _ in sites])
    B = np.concatenate([b for b, _, _ in sites]); Q = np.concatenate([q for _, q, _ in sites])
    R = np.array([r for _, _, r in sites], dtype=np.int8)
    got = ctx.lrt_csr(offs, B, Q, R, 0.001)
    for s, (b, q, r) in enumerate(sites):
        assert_site_matches(got[s], orc.basetype_lrt(b, q, r, 0.001), where=f"csr {s}")


def test_csr_mixed_lengths_alignments_and_device_pointers(ctx):
    """bvc_lrt_csr takes its sites by length: shorter than 4096 observations one wavefront per site, longer ones one
    workgroup per site with 16-byte loads over the aligned middle of the range.  Lengths either side of the switch
    and of the 16-byte grid, in one call, through host pointers, device pointers (arrays starting off the 16-byte
    grid: byte path) and overlap mode; records against the oracle's histogram form of each site."""
    import torch
    from basevarc_amd.lib import results_from_tensor
    rng = np.random.default_rng(55)
    lens = [0, 1, 15, 16, 17, 63, 64, 65, 1000, 4095, 4096, 4097, 4111, 8192, 10000, 70001, 3, 0, 5000, 131072 + 7]
    rng.shuffle(lens)
    sites = []
    for n in lens:
        b, q, r = random_site(rng, n, af=float(rng.choice([0.0, 0.02, 0.3])), qlo=2, qhi=60)
        if n > 100:
            b[rng.random(n) < 0.01] = 4                         # a few N bases inside (skipped)
        sites.append((b, q, r))
    offs = np.zeros(len(sites) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(b) for b, _, _ in sites])
    B = np.concatenate([b for b, _, _ in sites]); Q = np.concatenate([q for _, q, _ in sites])
    R = np.array([r for _, _, r in sites], dtype=np.int8)
    m = 0.001
    exp = []
    for b, q, r in sites:
        ok = (b >= 0) & (b < 4)
        exp.append(orc.hist_lrt(np.bincount(b[ok].astype(np.int64) * 128 + q[ok], minlength=512).astype(np.uint32), r, m))
    got = ctx.lrt_csr(offs, B, Q, R, m)
    for s in range(len(sites)):
        assert_site_matches(got[s], exp[s], where=f"csr host site {s} len {lens[s]}")
    ot = torch.from_numpy(offs).cuda()
    rt = torch.from_numpy(R).cuda()
    for shift in (0, 5):                                         # 5: arrays start off the 16-byte grid
        fb = torch.zeros(len(B) + 64, dtype=torch.int8, device="cuda")
        fq = torch.zeros_like(fb)
        fb[shift:shift + len(B)] = torch.from_numpy(B).cuda()
        fq[shift:shift + len(B)] = torch.from_numpy(Q).cuda()
        for overlap in (False, True):
            ctx.set_overlap(overlap)
            try:
                outs = [ctx.lrt_csr_device(ot, fb[shift:], fq[shift:], rt, m) for _ in range(3)]
                ctx.join()
                ctx.synchronize()
            finally:
                ctx.set_overlap(False)
            for o in outs:
                res = results_from_tensor(o)
                assert res.tobytes() == got.tobytes(), (shift, overlap)


def test_short_dense_rows_take_the_wave_kernel(ctx):
    """Dense tiles with rows of at most 16384 samples and thousands of sites go one wavefront per site
    (hist_wave_kernel): counts against numpy for row lengths either side of the switch, aligned and not."""
    import torch
    rng = np.random.default_rng(66)
    for n, stride, off in ((1, 16, 0), (17, 32, 0), (1000, 1000, 3), (16384, 16384, 0), (16385, 16400, 0), (4099, 4112, 16)):
        ns = 4100
        bh = rng.integers(-1, 5, (ns, stride)).astype(np.int8)
        qh = rng.integers(-3, 64, (ns, stride)).astype(np.int8)
        flat_b = torch.zeros(off + ns * stride + 64, dtype=torch.int8, device="cuda")
        flat_q = torch.zeros_like(flat_b)
        flat_b[off:off + ns * stride] = torch.from_numpy(bh.reshape(-1)).cuda()
        flat_q[off:off + ns * stride] = torch.from_numpy(qh.reshape(-1)).cuda()
        bt = flat_b[off:off + ns * stride].view(ns, stride)[:, :n]
        qt = flat_q[off:off + ns * stride].view(ns, stride)[:, :n]
        counts = ctx.hist_dense_device(bt, qt).cpu().numpy().view(np.uint32)
        ctx.synchronize()
        for s in rng.choice(ns, 200, replace=False):
            b, q = bh[s, :n], qh[s, :n]
            ok = (b >= 0) & (b < 4) & (q >= 0)
            want = np.bincount(b[ok].astype(np.int64) * 128 + q[ok], minlength=512).astype(np.uint32)
            assert np.array_equal(counts[s], want), (n, stride, off, s)


def test_group_mode_matches_callers_group_loop(ctx):
    """bvc_lrt_dense_groups vs the caller's --group loop (src/BaseVarC.cpp:617-661) restated in the oracle."""
    rng = np.random.default_rng(33)
    for n, k in ((600, 3), (5000, 5), (40000 + 7, 5), (3000, 1), (2000, 32)):
        ns = 12
        grp = (rng.integers(0, k, n)).astype(np.uint8)
        grp[rng.random(n) < 0.1] = 255                       # ~10 % of samples in no group
        sites = []
        for s in range(ns):
            b, q, r = random_site(rng, n, af=[0.0, 0.02, 0.3][s % 3], second_af=0.05 if s % 4 == 0 else 0.0)
            if s % 3 == 1:                                   # ALT confined to group 0 -> literal 0 elsewhere
                alt = (r + 1) % 4
                b = np.where((b != r) & (grp != 0), r, b).astype(np.int8)
                b[(grp == 0) & (rng.random(n) < 0.3)] = alt
            if s == 5:
                b[rng.random(n) < 0.5] = -1                  # uncovered samples
            sites.append((b, q, r))
        B, Q, R = pad_rows(sites)
        m = caller_min_af(n)
        res, gres = ctx.lrt_dense_groups(B, Q, R, m, grp, k)
        for s, (b, q, r) in enumerate(sites):
            o, gd, ga, ran, pres = orc.dense_site_groups(b, q, r, m, grp, k)
            assert_site_matches(res[s], o, where=f"groups overall n={n} site={s}")
            assert np.array_equal(gres[s]["depth"], gd), (n, s)
            assert np.array_equal(gres[s]["ran"], ran), (n, s)
            assert np.array_equal(gres[s]["present"], pres), (n, s)
            np.testing.assert_allclose(gres[s]["af"], ga, rtol=0, atol=AF_ATOL, err_msg=f"n={n} site={s}")


def test_group_mode_samples_ordered_by_group(ctx):
    """Samples ordered by group take the column-range histogram kernel (decided on the device): the records must
    equal, byte for byte, those of the same columns in shuffled order (general kernel), and match the oracle.
    Covers empty groups, ungrouped samples last, ranges that start and end off the 16-byte grid, one-sample runs."""
    rng = np.random.default_rng(34)
    cases = [(5000, 5, [0.3, 0.2, 0.2, 0.1, 0.1, 0.1]),        # (n, k, fractions of groups 0..k-1 and ungrouped)
             (40000 + 7, 5, [0.2, 0.0, 0.4, 0.2, 0.2, 0.0]),   # group 1 empty, nobody ungrouped
             (48000, 5, [0.31, 0.17, 0.23, 0.09, 0.11, 0.09]),  # 16-byte aligned rows: vector body + ragged edges
             (16 * 4096 + 16, 2, [0.5, 0.5, 0.0]),
             (3000, 1, [0.9, 0.1]),
             (2100, 32, [1 / 33.0] * 33),
             (700, 3, [0.0, 0.0, 0.0, 1.0]),                    # everybody ungrouped
             (64, 4, [0.25, 0.25, 0.25, 0.25, 0.0])]
    for n, k, frac in cases:
        counts = np.floor(np.array(frac) * n).astype(int)
        counts[int(np.argmax(counts))] += n - counts.sum()
        labels = np.concatenate([np.full(c, (g if g < k else 255), dtype=np.uint8) for g, c in enumerate(counts)])
        assert labels.size == n and np.all(np.diff(np.minimum(labels, k).astype(int)) >= 0)
        ns = 10
        sites = []
        for s in range(ns):
            b, q, r = random_site(rng, n, af=[0.0, 0.02, 0.3][s % 3], second_af=0.05 if s % 4 == 0 else 0.0)
            if s == 5:
                b[rng.random(n) < 0.5] = -1
            sites.append((b, q, r))
        B, Q, R = pad_rows(sites)
        m = caller_min_af(n)
        res, gres = ctx.lrt_dense_groups(B, Q, R, m, labels, k)
        perm = rng.permutation(n)                               # same columns, shuffled: the general kernel
        Bp, Qp = B.copy(), Q.copy()
        Bp[:, :n], Qp[:, :n] = B[:, perm], Q[:, perm]
        res_p, gres_p = ctx.lrt_dense_groups(Bp, Qp, R, m, labels[perm], k)
        assert np.array_equal(gres["depth"], gres_p["depth"]) and np.array_equal(res["depth"], res_p["depth"])
        assert res.tobytes() == res_p.tobytes() and gres.tobytes() == gres_p.tobytes(), (n, k)
        for s, (b, q, r) in enumerate(sites):
            o, gd, ga, ran, pres = orc.dense_site_groups(b, q, r, m, labels, k)
            assert_site_matches(res[s], o, where=f"ordered groups n={n} site={s}")
            assert np.array_equal(gres[s]["depth"], gd), (n, s)
            assert np.array_equal(gres[s]["ran"], ran) and np.array_equal(gres[s]["present"], pres), (n, s)
            np.testing.assert_allclose(gres[s]["af"], ga, rtol=0, atol=AF_ATOL, err_msg=f"n={n} site={s}")


def test_random_shapes_strides_and_offsets_on_device(ctx):
    """Seeded fuzz over the device-pointer entry points: random site and sample counts, row strides wider than the
    row, rows that start off the 16-byte grid (tensor views at odd offsets), random coverage, junk in the padding.
    Class counts against numpy, records against the oracle's histogram form."""
    import torch
    from basevarc_amd.lib import results_from_tensor
    rng = np.random.default_rng(2024)
    for case in range(40):
        ns = int(rng.integers(1, 40))
        n = int(rng.choice([0, 1, 7, 15, 16, 17, 100, 1023, 1024, 1025, 4099, int(rng.integers(2, 30000))]))
        pad = int(rng.choice([0, 0, 1, 5, 16, 48]))
        off = int(rng.choice([0, 0, 3, 16, 21]))
        stride = n + pad
        bh = rng.integers(0, 4, (ns, stride)).astype(np.int8)
        qh = rng.integers(0, 60, (ns, stride)).astype(np.int8)
        cov = float(rng.choice([1.0, 1.0, 0.5, 0.05]))
        bh[rng.random((ns, stride)) > cov] = -1                   # uncovered samples
        if stride > n:
            bh[:, n:] = rng.integers(-128, 127, (ns, stride - n))  # junk in the padding must never be read as data
            qh[:, n:] = rng.integers(-128, 127, (ns, stride - n))
        rh = rng.integers(0, 4, ns).astype(np.int8)
        flat_b = torch.zeros(off + ns * stride + 64, dtype=torch.int8, device="cuda")
        flat_q = torch.zeros_like(flat_b)
        flat_b[off:off + ns * stride] = torch.from_numpy(bh.reshape(-1)).cuda()
        flat_q[off:off + ns * stride] = torch.from_numpy(qh.reshape(-1)).cuda()
        bt = flat_b[off:off + ns * stride].view(ns, stride)[:, :n] if n else flat_b[off:off + ns * stride].view(ns, stride)[:, :0]
        qt = flat_q[off:off + ns * stride].view(ns, stride)[:, :n] if n else flat_q[off:off + ns * stride].view(ns, stride)[:, :0]
        rt = torch.from_numpy(rh).cuda()
        m = caller_min_af(max(n, 1))
        counts = ctx.hist_dense_device(bt, qt).cpu().numpy().view(np.uint32)
        res = results_from_tensor(ctx.lrt_dense_device(bt, qt, rt, m))
        ctx.synchronize()
        for s in range(ns):
            b, q = bh[s, :n], qh[s, :n]
            ok = (b >= 0) & (b < 4) & (q >= 0)
            exp = np.bincount(b[ok].astype(np.int64) * 128 + q[ok], minlength=512).astype(np.uint32)
            assert np.array_equal(counts[s], exp), (case, s, ns, n, pad, off)
            assert_site_matches(res[s], orc.hist_lrt(exp, int(rh[s]), m), where=f"fuzz case {case} site {s} n={n}")


def test_zero_samples_is_a_no_call_not_an_error(ctx):
    """Sites with no sample at all (empty vectors into BaseType): depth 0, no call, through host and CSR entry points."""
    B = np.zeros((3, 0), dtype=np.int8)
    R = np.array([0, 1, 2], dtype=np.int8)
    for res in (ctx.lrt_dense(B, B.copy(), R, 0.001),
                ctx.lrt_csr(np.zeros(4, dtype=np.int64), np.zeros(0, dtype=np.int8), np.zeros(0, dtype=np.int8), R, 0.001)):
        assert res["called"].tolist() == [0, 0, 0] and res["depth_total"].tolist() == [0.0, 0.0, 0.0]
        assert (res["depth"] == 0).all() and (res["n_alt"] == 0).all()
    res, gres = ctx.lrt_dense_groups(B, B.copy(), R, 0.001, np.zeros(0, dtype=np.uint8), 2)
    assert res["called"].tolist() == [0, 0, 0] and (gres["depth"] == 0).all() and (gres["ran"] == 0).all()


# ------------------------------------------------------------------ golden fixtures (tests/golden)
def _with_pruned_counts(exp, b, q, r, m):
    """The fixtures hold the reference's counts of EM() calls and passes; what the item engine runs by default (path_counts_match)
    comes from the oracle on the fixture's own inputs -- after checking that the oracle still reproduces the fixture's counts."""
    o = orc.basetype_lrt(b, q, r, m)
    assert (o["n_fits"], o["n_passes"]) == (exp["n_fits"], exp["n_passes"])
    return dict(exp, n_fits_pruned=o["n_fits_pruned"], n_passes_pruned=o["n_passes_pruned"], prune_edge=o["prune_edge"],
                max_quals=o["max_quals"], min_qual=o["min_qual"], dup_candidate=o["dup_candidate"])


def test_golden_fixtures(ctx):
    from tests.golden.golden_io import load_golden
    for name in ("basetype_random.npz", "basetype_edge.npz"):
        g = load_golden(name)
        for m in np.unique(g["min_af"]):
            idx = np.nonzero(g["min_af"] == m)[0]
            sites = [(g["bases"][g["offsets"][i]:g["offsets"][i + 1]],
                      g["quals"][g["offsets"][i]:g["offsets"][i + 1]], int(g["ref"][i])) for i in idx]
            B, Q, R = pad_rows(sites, width=max(1, max(len(b) for b, _, _ in sites)))
            got = ctx.lrt_dense(B, Q, R, float(m))
            for j, i in enumerate(idx):
                assert_site_matches(got[j], _with_pruned_counts(g["expected"][i], *sites[j], float(m)), where=f"{name}[{i}]")


def test_reference_test_data_pileup(ctx):
    """BASELINE configs[0] at the BaseType boundary: every covered position of the reference's own test data
    (test/test.sh:3 -- 100 BAMs, chr17:41197700-41276155, -q 20; 66,614 sites, depth 0..28), pileup columns
    extracted by tests/golden/make_testdata_pileup.py.  north_star: "identical ref/alt calls and AF/LRT within
    1e-6 on the test/ data"."""
    from tests.golden.golden_io import load_golden
    g = load_golden("testdata_pileup.npz")
    n = len(g["ref"])
    assert n == 66614
    got = ctx.lrt_csr(g["offsets"], g["bases"], g["quals"], g["ref"], float(g["min_af"][0]))
    called = 0
    for i in range(n):
        lo, hi = int(g["offsets"][i]), int(g["offsets"][i + 1])
        exp = _with_pruned_counts(g["expected"][i], g["bases"][lo:hi], g["quals"][lo:hi], int(g["ref"][i]), float(g["min_af"][0]))
        assert_site_matches(got[i], exp, where=f"test data site {i}")
        called += g["expected"][i]["called"]
    assert called == int(got["called"].sum()) == 76


# ------------------------------------------------------------------ synthetic generator + configs
def test_device_generator_is_bit_identical_to_cpu(ctx):
    import torch
    for n, cov in ((4096, 65536), (10000, 65536), (1000 + 7, 6554)):
        stride = (n + 15) // 16 * 16
        b = torch.empty((6, stride), dtype=torch.int8, device="cuda")
        q = torch.empty((6, stride), dtype=torch.int8, device="cuda")
        r = torch.empty(6, dtype=torch.int8, device="cuda")
        ctx.synth_dense_device(3, 1000, b[:, :n], q[:, :n], r, cov_thr16=cov)
        ctx.synchronize()
        eb, eq, er = orc.synth_tile(3, 1000, 6, n, cov_thr16=cov)
        assert np.array_equal(b[:, :n].cpu().numpy(), eb)
        assert np.array_equal(q[:, :n].cpu().numpy(), eq)
        assert np.array_equal(r.cpu().numpy(), er)


def test_config2_1e4_sites_by_1e4_samples(ctx):
    """BASELINE.json configs[1]: synthetic 1e4 sites x 1e4 samples, EM to convergence, vs the CPU path."""
    import torch
    from basevarc_amd.lib import results_from_tensor
    ns, n = 10000, 10000
    m = caller_min_af(n)
    b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(1, 0, b, q, r)
    res = results_from_tensor(ctx.lrt_dense_device(b, q, r, m))
    hb, hq, hr = b.cpu().numpy(), q.cpu().numpy(), r.cpu().numpy()
    # every site against the histogram form of the oracle (fast) ...
    exp_h, _ = orc.dense_batch(hb, hq, hr, m, use_hist=True)
    ties = 0
    for s in range(ns):
        assert_site_matches(res[s], exp_h[s], where=f"config2 hist-oracle site {s}", path_strict=False)
        ties += assert_path_difference_is_a_tie(res[s], exp_h[s], where=f"config2 site {s}")
    assert ties < 0.01 * ns, ties
    # ... and EVERY site against the faithful per-sample oracle (about 20 ms per site and core)
    exp_f, _ = orc.dense_batch(hb, hq, hr, m, use_hist=False)
    for s in range(ns):
        assert_site_matches(res[s], exp_f[s], where=f"config2 faithful site {s}", path_strict=False)
        assert_path_difference_is_a_tie(res[s], exp_f[s], where=f"config2 faithful site {s}", faithful_n=n)
    called = int(res["called"].sum())
    assert 0.05 * ns < called < 0.5 * ns          # ~20 % polymorphic sites in the mixture


@pytest.mark.parametrize("knob,value", [("em_wpb", 1), ("em_waves_per_cu", 3), ("hist_split", 5), ("em_waves_per_cu", 32),
                                        ("em_streams", 1), ("em_streams", 3)])
def test_tuning_knobs_are_per_context_and_never_change_a_record(ctx, knob, value):
    """bvc_set_tuning acts on ONE context (no process-wide state) and only moves work around: the records of a tuned
    context equal, byte for byte, those of an untouched one, and both match the oracle."""
    import torch
    from basevarc_amd import Context
    from basevarc_amd.lib import results_from_tensor
    with Context(0) as tuned:
        tuned.set_tuning(knob, value)
        with pytest.raises(Exception):
            tuned.set_tuning("em_rows", 1)                      # removed knob: rejected, not ignored
        for n, ns in ((3000, 401), (70000, 97)):
            m = caller_min_af(n)
            b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
            q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
            r = torch.empty(ns, dtype=torch.int8, device="cuda")
            ctx.synth_dense_device(9, 777, b, q, r)
            ctx.synchronize()
            plain = ctx.lrt_dense_device(b, q, r, m)
            ctx.synchronize()
            other = tuned.lrt_dense_device(b, q, r, m)
            tuned.synchronize()
            assert np.array_equal(plain.cpu().numpy(), other.cpu().numpy()), (knob, value, n)
            tuned.set_overlap(True)                             # several calls in flight (stage 2 on the side streams)
            outs = [tuned.lrt_dense_device(b, q, r, m) for _ in range(5)]
            tuned.join(); tuned.synchronize()
            tuned.set_overlap(False)
            assert all(np.array_equal(plain.cpu().numpy(), o.cpu().numpy()) for o in outs), (knob, value, n, "overlap")
            res = results_from_tensor(other)
            exp, _ = orc.dense_batch(b.cpu().numpy(), q.cpu().numpy(), r.cpu().numpy(), m, use_hist=True)
            for s in range(ns):
                assert_site_matches(res[s], exp[s], where=f"{knob}={value} n={n} site={s}", path_strict=False)


@pytest.mark.parametrize("pattern", ["random", "mod5", "runs", "one_group", "sorted"])
def test_group_kernel_variants_give_identical_records(ctx, pattern):
    """Group mode picks its histogram kernel on the device: sorted labels -> column ranges; any other order -> the
    class-bank kernel; rows off the 16-byte grid -> the byte kernel.  Whatever runs, the records are the same bytes:
    each label pattern through the default path, with the unpipelined load schedule, and through the byte kernel
    (rows shifted by one byte), plus the oracle on a few sites."""
    import torch
    from basevarc_amd import Context
    from basevarc_amd.lib import GROUP_DTYPE, results_from_tensor
    ns, n, k = 37, 50_000 + 9, 5
    m = caller_min_af(n)
    rng = np.random.default_rng(8)
    b = torch.empty((ns, n + 23), dtype=torch.int8, device="cuda")
    q = torch.empty((ns, n + 23), dtype=torch.int8, device="cuda")
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(11, 4242, b[:, :n], q[:, :n], r, cov_thr16=60000)
    labels = {"random": rng.integers(0, 7, n),                       # labels 5, 6 = no group
              "mod5": np.arange(n) % 5,
              "runs": (np.arange(n) // 700) % 6,                     # long runs, not sorted: 32 equal labels per lane set
              "one_group": np.full(n, 2),
              "sorted": np.sort(rng.integers(0, 6, n))}[pattern].astype(np.uint8)
    g = torch.from_numpy(labels).cuda()
    ctx.synchronize()
    base = ctx.lrt_dense_groups_device(b[:, :n], q[:, :n], r, m, g, k)
    ctx.synchronize()
    ref = [t.cpu().numpy().copy() for t in base]
    for knob in ("group_pipe",):
        with Context(0) as other:
            other.set_tuning(knob, 0)
            out = other.lrt_dense_groups_device(b[:, :n], q[:, :n], r, m, g, k)
            other.synchronize()
            assert all(np.array_equal(x.cpu().numpy(), y) for x, y in zip(out, ref)), (pattern, knob)
    # the same rows shifted by one byte: unaligned -> generic byte kernel
    fb = torch.zeros(ns * (n + 23) + 1, dtype=torch.int8, device="cuda")
    fq = torch.zeros_like(fb)
    fb[1:] = b.reshape(-1); fq[1:] = q.reshape(-1)
    ub = fb[1:].view(ns, n + 23)[:, :n]; uq = fq[1:].view(ns, n + 23)[:, :n]
    out = ctx.lrt_dense_groups_device(ub, uq, r, m, g, k)
    ctx.synchronize()
    assert all(np.array_equal(x.cpu().numpy(), y) for x, y in zip(out, ref)), pattern
    res = results_from_tensor(base[0])
    gres = ref[1].view(GROUP_DTYPE).reshape(ns, k)
    hb, hq, hr = b[:, :n].cpu().numpy(), q[:, :n].cpu().numpy(), r.cpu().numpy()
    for s in (0, 5, 17, 36):
        o, gd, ga, ran, pres = orc.dense_site_groups(hb[s], hq[s], int(hr[s]), m, labels, k, use_hist=True)
        assert_site_matches(res[s], o, where=f"{pattern} site {s}", path_strict=False)
        assert np.array_equal(gres[s]["depth"], gd) and np.array_equal(gres[s]["ran"], ran), (pattern, s)
        np.testing.assert_allclose(gres[s]["af"], ga, rtol=0, atol=AF_ATOL)


def test_host_pointer_calls_pipeline_their_chunks(ctx):
    """BVC_PTR_HOST calls stage the tile through two device buffers, chunk by chunk, the upload of chunk i+1 under the
    kernels of chunk i.  With the chunk size turned down ("host_chunk_kib") a 300-site tile goes through in dozens of
    chunks, with and without groups and overlap: the records must be those of the one-chunk call, byte for byte."""
    from basevarc_amd import Context
    rng = np.random.default_rng(12)
    ns, n, k = 301, 20000, 4
    sites = [random_site(rng, n, af=[0.0, 0.05, 0.4][s % 3]) for s in range(ns)]
    B, Q, R = pad_rows(sites, width=n + 48)
    B[:, n:] = 7                                                 # junk in the row padding
    m = caller_min_af(n)
    g = rng.integers(0, k + 1, n + 48).astype(np.uint8)
    one = ctx.lrt_dense(B, Q, R, m)
    one_g = ctx.lrt_dense_groups(B, Q, R, m, g, k)
    for overlap in (False, True):
        for kib in (20, 137, 1000):                              # 1, 6 and 51 sites per chunk
            with Context(0) as c:
                c.set_tuning("host_chunk_kib", kib)
                c.set_overlap(overlap)
                assert c.lrt_dense(B, Q, R, m).tobytes() == one.tobytes(), (overlap, kib)
                res, gres = c.lrt_dense_groups(B, Q, R, m, g, k)
                assert res.tobytes() == one_g[0].tobytes() and gres.tobytes() == one_g[1].tobytes(), (overlap, kib)
    exp = orc.basetype_lrt(sites[7][0], sites[7][1], sites[7][2], m)
    assert_site_matches(one[7], exp, where="chunked host call, site 7")


def test_overlap_mode_gives_identical_records(ctx):
    """Stage 2 on the side stream under the next call's stage 1 (bvc_set_overlap): same bytes out."""
    import torch
    from basevarc_amd.lib import SITE_DTYPE
    ns, n = 300, 50000
    m = caller_min_af(n)
    tiles = []
    for t in range(5):
        b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
        q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
        r = torch.empty(ns, dtype=torch.int8, device="cuda")
        ctx.synth_dense_device(7, 1000 * t, b, q, r)
        tiles.append((b, q, r))
    ctx.synchronize()
    plain = []
    for b, q, r in tiles:
        out = ctx.lrt_dense_device(b, q, r, m)
        ctx.synchronize()
        plain.append(out.cpu().numpy().copy())
    try:
        ctx.set_overlap(True)
        outs = [torch.zeros(ns * SITE_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in tiles]
        for rep in range(3):
            for (b, q, r), o in zip(tiles, outs):
                ctx.lrt_dense_device(b, q, r, m, o)
        ctx.join()
        torch.cuda.current_stream().synchronize()
        for o, p in zip(outs, plain):
            assert np.array_equal(o.cpu().numpy(), p)
    finally:
        ctx.set_overlap(False)


def test_overlap_ring_with_growing_tiles_and_long_rows(ctx):
    """Overlap mode with rows long enough for the capped EM grid, and calls of growing size: the three ring
    buffers are re-allocated while other calls are in flight; every call still returns its plain-mode bytes."""
    import torch
    from basevarc_amd.lib import SITE_DTYPE
    n = 200_000
    m = caller_min_af(n)
    sizes = [48, 200, 96, 400, 16, 640, 320]
    tiles = []
    for t, ns in enumerate(sizes):
        b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
        q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
        r = torch.empty(ns, dtype=torch.int8, device="cuda")
        ctx.synth_dense_device(9, 5000 * t, b, q, r)
        tiles.append((b, q, r))
    ctx.synchronize()
    plain = []
    for b, q, r in tiles:
        out = ctx.lrt_dense_device(b, q, r, m)
        ctx.synchronize()
        plain.append(out.cpu().numpy().copy())
    from basevarc_amd import Context
    with Context(0) as fresh:                      # fresh context: its scratch starts empty and has to grow
        fresh.set_overlap(True)
        outs = [torch.zeros(b.shape[0] * SITE_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for b, _, _ in tiles]
        for (b, q, r), o in zip(tiles, outs):
            fresh.lrt_dense_device(b, q, r, m, o)
        fresh.join()
        fresh.synchronize()
        for o, p in zip(outs, plain):
            assert np.array_equal(o.cpu().numpy(), p)


def test_full_size_sites_1e6_samples(ctx):
    """configs[2] shape (N = 1e6 samples per site): exact histogram, and the LRT against the oracle."""
    import torch
    from basevarc_amd.lib import results_from_tensor
    ns, n = 256, 1_000_000
    m = caller_min_af(n)
    b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(2, 5000, b, q, r)
    counts = ctx.hist_dense_device(b, q).cpu().numpy().view(np.uint32)
    res = results_from_tensor(ctx.lrt_dense_device(b, q, r, m))
    hb, hq, hr = b.cpu().numpy(), q.cpu().numpy(), r.cpu().numpy()
    assert (counts.sum(axis=1) == n).all()             # checksum of every histogram = sample count
    ties = 0
    for s in range(ns):
        if s < 32:                                     # independent recount with numpy on a subset (slow in Python)
            key = hb[s].astype(np.int64) * 128 + hq[s]
            assert np.array_equal(counts[s], np.bincount(key, minlength=512).astype(np.uint32)), s
        assert np.array_equal(counts[s], orc.dense_hist(hb[s], hq[s])), s
        e = orc.hist_lrt(counts[s], hr[s], m)
        assert_site_matches(res[s], e, where=f"1e6 hist-oracle site {s}", path_strict=False)
        ties += assert_path_difference_is_a_tie(res[s], e, where=f"1e6 site {s}")
    assert ties <= 0.03 * ns, ties
    # faithful per-sample oracle on 4 sites (about 20 s each, run in parallel on the host cores)
    exp_f, _ = orc.dense_batch(hb[:4], hq[:4], hr[:4], m, use_hist=False)
    for s in range(4):
        assert_site_matches(res[s], exp_f[s], where=f"1e6 faithful site {s}", path_strict=False, faithful_drift=True)
        assert_path_difference_is_a_tie(res[s], exp_f[s], where=f"1e6 faithful site {s}", faithful_n=n)


def test_chi_at_1e6_matches_the_compensated_oracle(ctx):
    """The north_star's "AF/LRT within 1e-6" at N = 1e6, and why the comparison with the FAITHFUL oracle needs a wider
    floor there.  Four synthetic sites (one polymorphic by construction of the seed) through three CPU forms:
      faithful      per-sample loops, double accumulators            (the reference's arithmetic)
      compensated   the same loops, sums over samples in long double (ORC_MODE_COMPENSATED)
      histogram     EM on the (base, qual) counts                    (what the GPU computes, on the CPU)
    GPU chi / var_qual must equal the compensated per-sample oracle to 1e-9 (relative, plus a few ulps of the
    log-likelihood chi is a difference of); the faithful oracle's distance from the compensated one must cover
    what the widened floor of assert_site_matches(faithful_drift=True) allows, and stay inside N*u*|loglik|."""
    import torch
    from basevarc_amd.lib import results_from_tensor
    ns, n = 4, 1_000_000
    m = caller_min_af(n)
    b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(2, 5000, b, q, r)
    res = results_from_tensor(ctx.lrt_dense_device(b, q, r, m))
    hb, hq, hr = b.cpu().numpy(), q.cpu().numpy(), r.cpu().numpy()
    exp_f, _ = orc.dense_batch(hb, hq, hr, m, use_hist=False)
    exp_c, _ = orc.dense_batch(hb, hq, hr, m, use_hist=False, compensated=True)
    worst_gpu = worst_faithful = 0.0
    for s in range(ns):
        f, c, g = exp_f[s], exp_c[s], res[s]
        assert_site_matches(g, c, where=f"1e6 compensated site {s}", path_strict=False)       # 1e-6 floor, no drift term
        assert_path_difference_is_a_tie(g, c, where=f"1e6 compensated site {s}")
        ulp_ll = abs(c["lr_alt"]) * 2.0 ** -52
        for name in ("chi", "lr_alt"):
            tol = 1e-9 * max(1.0, abs(c[name])) + 256 * ulp_ll
            assert abs(float(g[name]) - c[name]) <= tol, (s, name, float(g[name]), c[name], tol)
        if c["called"] and c["var_qual"] not in (5000.0, 10000.0):
            assert float(g["var_qual"]) == pytest.approx(c["var_qual"], rel=1e-9, abs=1e-8), s
        worst_gpu = max(worst_gpu, abs(float(g["chi"]) - c["chi"]))
        drift = abs(f["chi"] - c["chi"])
        worst_faithful = max(worst_faithful, drift)
        assert drift <= 2 * n * 2.0 ** -53 * abs(f["lr_alt"]), (s, drift)            # inside the N*u*|loglik| bound
        assert drift <= 2e-10 * abs(f["lr_alt"]), (s, drift)                          # ... and inside the test floor
        np.testing.assert_allclose([float(x) for x in g["af"][:c["n_alt"]]], c["af"], rtol=0, atol=1e-10)
    # the faithful sum is orders of magnitude farther from the compensated one than the GPU is
    assert worst_faithful > 100 * worst_gpu, (worst_faithful, worst_gpu)
    print(f"max |chi_gpu - chi_compensated| = {worst_gpu:.3e}; max |chi_faithful - chi_compensated| = {worst_faithful:.3e}")


def _group_labels(n, k, layout):
    if layout == "ordered":
        return (np.arange(n) * k // n).astype(np.uint8)          # k equal groups as contiguous runs of columns
    return (np.arange(n) % k).astype(np.uint8)                   # group = sample % k


@pytest.mark.parametrize("layout", ["interleaved", "ordered"])
def test_config5_groups_at_1e6_samples(ctx, layout):
    """BASELINE.json configs[4]: --group with k = 5 populations on N = 1e6 samples (src/BaseVarC.cpp:617-661), both
    sample orders (i % 5: the any-order histogram kernel; contiguous runs: the column-range kernel).  65 sites (an odd
    count: the two-sites-per-pass kernel's short last pass) against the oracle's histogram form of the caller's group
    loop, 2 of them also against the faithful per-sample group loop."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from basevarc_amd.lib import GROUP_DTYPE, results_from_tensor
    ns, n, k = 65, 1_000_000, 5
    m = caller_min_af(n)
    stride = (n + 127) // 128 * 128
    b = torch.empty((ns, stride), dtype=torch.int8, device="cuda")[:, :n]
    q = torch.empty((ns, stride), dtype=torch.int8, device="cuda")[:, :n]
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(3, 12345, b, q, r)
    g = _group_labels(n, k, layout)
    g[::97] = 255                                                  # some samples in no group (interleaved case only)
    if layout == "ordered":
        g = np.sort(np.minimum(g, k)).astype(np.uint8)
        g[g == k] = 255
    gt = torch.from_numpy(g).cuda()
    res_t, gres_t = ctx.lrt_dense_groups_device(b, q, r, m, gt, k)
    ctx.synchronize()
    res = results_from_tensor(res_t)
    gres = gres_t.cpu().numpy().view(GROUP_DTYPE).reshape(ns, k)
    hb, hq, hr = b.cpu().numpy(), q.cpu().numpy(), r.cpu().numpy()

    def check(s, use_hist):
        o, gd, ga, ran, pres = orc.dense_site_groups(hb[s], hq[s], int(hr[s]), m, g, k, use_hist=use_hist)
        assert_site_matches(res[s], o, where=f"groups 1e6 {layout} site {s} hist={use_hist}", path_strict=False,
                            faithful_drift=not use_hist)
        assert np.array_equal(gres[s]["depth"], gd), (layout, s)
        assert np.array_equal(gres[s]["ran"], ran) and np.array_equal(gres[s]["present"], pres), (layout, s)
        np.testing.assert_allclose(gres[s]["af"], ga, rtol=0, atol=AF_ATOL, err_msg=f"{layout} site {s}")
        return int(o["called"]), int(np.sum(ran))

    with ThreadPoolExecutor(8) as pool:                            # the C calls release the GIL
        out = list(pool.map(lambda s: check(s, True), range(ns)))
        called = sum(c for c, _ in out)
        assert called >= 3 and sum(rn for _, rn in out) >= 3 * k, out    # the sample holds called sites with group runs
        # faithful per-sample group loop on two called sites (about a minute each, side by side)
        pick = [s for s, (c, _) in enumerate(out) if c][:2]
        list(pool.map(lambda s: check(s, False), pick))
    # the depth columns of the groups and of "no group" add up to the overall depth
    assert np.array_equal(gres["depth"].sum(axis=1) + np.array([np.bincount(hb[s][g == 255], minlength=4)[:4] for s in range(ns)]),
                          res["depth"])


def test_error_behaviour_of_the_c_abi(ctx):
    """SURVEY 8b, "Errors": every entry point returns a status, nothing throws across the ABI, bvc_last_error carries
    the text, "no call" is not an error, and a context stays usable after a rejected call."""
    import ctypes as C
    from basevarc_amd.lib import BvcError, SITE_DTYPE, load_library
    L = load_library()
    h = C.c_void_p()
    assert L.bvc_create(C.byref(h), 99) == -3 and not h.value          # BVC_ERR_NO_DEVICE, no context
    assert L.bvc_create(None, 0) == -1                                   # BVC_ERR_ARG
    assert L.bvc_last_error(None) == b"null context"
    B = np.zeros((2, 8), dtype=np.int8); R = np.zeros(2, dtype=np.int8)
    out = np.zeros(2, dtype=SITE_DTYPE)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    hh = ctx._h
    # n_samples > row_stride, negative sizes, null pointers
    assert L.bvc_lrt_dense(hh, 2, 8, 4, vp(B), vp(B), vp(R), 0.001, vp(out), 0) == -1
    assert b"row_stride" in L.bvc_last_error(hh)
    assert L.bvc_lrt_dense(hh, -1, 8, 8, vp(B), vp(B), vp(R), 0.001, vp(out), 0) == -1
    assert L.bvc_lrt_dense(hh, 2, 8, 8, None, vp(B), vp(R), 0.001, vp(out), 0) == -1
    assert L.bvc_lrt_dense(hh, 2, 8, 8, vp(B), vp(B), vp(R), 0.001, None, 0) == -1
    assert L.bvc_lrt_dense(hh, 0, 8, 8, None, None, None, 0.001, None, 0) == 0       # nothing to do is not an error
    # groups: n_groups out of range, null group vector
    g = np.zeros(8, dtype=np.uint8); gout = np.zeros((2, 40), dtype=np.uint8)
    for k in (0, 33):
        assert L.bvc_lrt_dense_groups(hh, 2, 8, 8, vp(B), vp(B), vp(R), 0.001, vp(g), k, vp(out), vp(gout), 0) == -1
    assert L.bvc_lrt_dense_groups(hh, 2, 8, 8, vp(B), vp(B), vp(R), 0.001, None, 2, vp(out), vp(gout), 0) == -1
    # ragged: offsets must start at 0 and not decrease
    with pytest.raises(BvcError, match="offsets"):
        ctx.lrt_csr([1, 4, 8], B.reshape(-1)[:8], B.reshape(-1)[:8], R, 0.001)
    with pytest.raises(BvcError, match="offsets"):
        ctx.lrt_csr([0, 6, 4], B.reshape(-1)[:8], B.reshape(-1)[:8], R, 0.001)
    # SetBase lists: more than four entries, entries outside 0..3, one of the two arrays missing
    cnt = np.zeros((1, 512), dtype=np.uint32)
    with pytest.raises(BvcError, match="n_comb"):
        ctx.lrt_hist(cnt, [0], 0.001, [[0, 1, 2, 3]], [5])
    with pytest.raises(BvcError, match="base_comb"):
        ctx.lrt_hist(cnt, [0], 0.001, [[0, 7, 0, 0]], [2])
    assert L.bvc_lrt_hist(hh, 1, vp(cnt), vp(R), 0.001, vp(B), None, vp(out), 0) == -1
    with pytest.raises(BvcError):
        ctx.set_tuning("no_such_knob", 1)
    with pytest.raises(BvcError):
        ctx.set_tuning("em_waves_per_cu", 1000)
    # packed entry points: the same argument rules; packing and the packed histogram take device pointers only
    P = np.full((2, 8), 0x40 | 30, dtype=np.uint8)
    bad = C.c_int64(-1)
    assert L.bvc_lrt_dense_packed(hh, 2, 8, 4, vp(P), vp(R), 0.001, vp(out), 0) == -1
    assert L.bvc_lrt_dense_packed(hh, 2, 8, 8, None, vp(R), 0.001, vp(out), 0) == -1
    assert L.bvc_lrt_dense_packed(hh, 0, 8, 8, None, None, 0.001, None, 0) == 0
    assert L.bvc_lrt_dense_packed(hh, 2, 8, 8, vp(P), vp(R), 0.001, vp(out), 0) == 0 and out["depth"].tolist() == [[0, 8, 0, 0]] * 2
    for k in (0, 33):
        assert L.bvc_lrt_dense_groups_packed(hh, 2, 8, 8, vp(P), vp(R), 0.001, vp(g), k, vp(out), vp(gout), 0) == -1
    assert L.bvc_lrt_dense_groups_packed(hh, 2, 8, 8, vp(P), vp(R), 0.001, None, 2, vp(out), vp(gout), 0) == -1
    assert L.bvc_pack_dense(hh, 2, 8, 8, vp(B), vp(B), 8, vp(P), C.byref(bad), 0) == -1 and b"device pointers" in L.bvc_last_error(hh)
    assert L.bvc_pack_dense(hh, 2, 8, 8, vp(B), vp(B), 4, vp(P), C.byref(bad), 1) == -1     # packed stride < n_samples
    assert L.bvc_hist_dense_packed(hh, 2, 8, 8, vp(P), vp(cnt), 0) == -1
    # the context is still good, and a site without a call is a record, not an error
    rec = ctx.lrt_dense(np.full((1, 50), 2, dtype=np.int8), np.full((1, 50), 30, dtype=np.int8), [2], 0.001)
    assert int(rec[0]["called"]) == 0 and int(rec[0]["status"]) == 0 and rec[0]["depth"].tolist() == [0, 0, 50, 0]


def test_tiles_beyond_4_gib_are_addressed_with_64_bits(ctx):
    """Maximum sizes: one call over a tile whose arrays are larger than 2^32 bytes (4,400 sites x 1e6 samples = 4.4 GB
    each; the bench's own tiles stop at 4.0e9).  Every byte offset inside the kernels is 64-bit: the records of the
    sites that lie beyond the 4 GiB mark must be the records the same rows give as a tile of their own, their histograms
    must match the oracle's, and every histogram of the big call must sum to N.  The same tile as ragged (CSR) sites
    (element offsets up to 4.4e9) and in group mode (both kernels)."""
    import torch
    from basevarc_amd.lib import GROUP_DTYPE, results_from_tensor
    ns, n, k = 4400, 1_000_000, 5
    m = caller_min_af(n)
    b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(7, 0, b, q, r)
    first_beyond = (1 << 32) // n + 1                              # first row that starts past the 4 GiB mark
    assert first_beyond < ns - 8
    tail = slice(ns - 8, ns)

    counts = ctx.hist_dense_device(b, q)
    assert bool((counts.view(torch.int32).sum(dim=1) == n).all())
    hb, hq = b[tail].cpu().numpy(), q[tail].cpu().numpy()
    ct = counts[tail].cpu().numpy().view(np.uint32)
    for i in range(8):
        assert np.array_equal(ct[i], orc.dense_hist(hb[i], hq[i])), i

    big = results_from_tensor(ctx.lrt_dense_device(b, q, r, m))
    small = results_from_tensor(ctx.lrt_dense_device(b[tail], q[tail], r[tail], m))
    assert big[tail].tobytes() == small.tobytes()
    assert (big["depth_total"] == n).all()
    hr = r[tail].cpu().numpy()
    for i in range(8):
        assert_site_matches(small[i], orc.hist_lrt(ct[i], hr[i], m), where=f"beyond 4 GiB site {i}", path_strict=False)

    # ragged form of the same bytes: site s = elements [s * n, (s + 1) * n) of the concatenated arrays
    offs = (torch.arange(ns + 1, dtype=torch.int64) * n).cuda()
    csr = results_from_tensor(ctx.lrt_csr_device(offs, b.reshape(-1), q.reshape(-1), r, m))
    assert csr.tobytes() == big.tobytes()

    # group mode, labels in any order and ordered by group
    for layout in ("interleaved", "ordered"):
        g = torch.from_numpy(_group_labels(n, k, layout)).cuda()
        res_t, gres_t = ctx.lrt_dense_groups_device(b, q, r, m, g, k)
        res_s, gres_s = ctx.lrt_dense_groups_device(b[tail], q[tail], r[tail], m, g, k)
        ctx.synchronize()
        assert results_from_tensor(res_t).tobytes() == big.tobytes(), layout
        gb = gres_t.cpu().numpy().view(GROUP_DTYPE).reshape(ns, k)
        gs = gres_s.cpu().numpy().view(GROUP_DTYPE).reshape(8, k)
        assert gb[tail].tobytes() == gs.tobytes(), layout
        assert (gb["depth"].sum(axis=(1, 2)) == n).all(), layout


def _fuzz_histogram(rng):
    """A random (base, qual) count histogram with the features the EM kernel's variants key on: the number of distinct
    quality values per base (slot count 1..8, with the variant edges 16/17, 32/33, 48/49, 64/65, 96/97 over-sampled),
    the depth (1 .. 2e9), the number of alleles and how far their fractions sit from min_af."""
    nq = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 95, 96, 97, 127, 128, int(rng.integers(1, 129))]))
    quals = np.sort(rng.choice(128, size=nq, replace=False))
    if quals[0] == 0 and rng.random() < 0.8:
        quals = quals[1:] if len(quals) > 1 else np.array([1])         # Q = 0 is the NaN path; keep it rare
    depth = int(10 ** rng.uniform(0, 9.3))
    ref = int(rng.integers(0, 4))
    n_alt = int(rng.choice([0, 1, 1, 2, 3]))
    fr = np.zeros(4)
    alts = rng.permutation([b for b in range(4) if b != ref])[:n_alt]
    for a in alts:
        fr[a] = 10 ** rng.uniform(-7, -0.3)
    fr[ref] = max(1.0 - fr.sum(), 0.0) if rng.random() < 0.9 else 0.0   # sometimes no reference observation at all
    if fr.sum() == 0:
        fr[ref] = 1.0
    fr /= fr.sum()
    per_base = rng.multinomial(min(depth, 2_000_000_000), fr) if depth < 2 ** 31 else None
    counts = np.zeros((4, 128), dtype=np.uint32)
    for b in range(4):
        nb = int(per_base[b])
        if nb == 0:
            continue
        use = quals if rng.random() < 0.7 else quals[rng.random(len(quals)) < 0.5]
        if len(use) == 0:
            use = quals[:1]
        w = rng.dirichlet(np.full(len(use), 0.7))
        counts[b, use] = rng.multinomial(nb, w).astype(np.uint32)
    return counts.reshape(512), ref


def test_fuzz_random_histograms_against_the_oracle(ctx):
    """3,000 random histograms through bvc_lrt_hist (stage 2 alone) against the oracle's histogram form: every kernel
    variant (2 / 4 / 8 register slots per lane and their narrow forms), depths from 1 to 2e9, fractions on both sides
    of min_af, candidate lists from SetBase."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(20261004)
    n = 3000
    hs, refs, combs, ncs = [], [], [], []
    for i in range(n):
        h, r = _fuzz_histogram(rng)
        hs.append(h); refs.append(r)
        if i % 3 == 0:                                              # SetBase: the reference base plus a random subset
            others = [b for b in range(4) if b != r and rng.random() < 0.6]
            cb = [r] + others
        else:
            cb = [0, 1, 2, 3]
        combs.append(cb + [0] * (4 - len(cb))); ncs.append(len(cb))
    H = np.stack(hs); R = np.array(refs, dtype=np.int8)
    CB = np.array(combs, dtype=np.int8); NC = np.array(ncs, dtype=np.uint8)
    for min_af in (1e-4, 0.001):
        got = ctx.lrt_hist(H, R, min_af, CB, NC)
        with ThreadPoolExecutor(8) as pool:
            exp = list(pool.map(lambda i: orc.hist_lrt(H[i], int(R[i]), min_af, base_comb=CB[i][:NC[i]]), range(n)))
        ties = 0
        for i in range(n):
            assert_site_matches(got[i], exp[i], where=f"fuzz {i} min_af {min_af}", path_strict=False)
            ties += assert_path_difference_is_a_tie(got[i], exp[i], where=f"fuzz {i}")
        assert ties <= 0.02 * n, ties
        assert sum(e["called"] for e in exp) > n // 10


def _pack_numpy(b, q):
    """The packed byte of include/bvc.h: base << 6 | qual for covered samples with qual <= 62, else 0xFF."""
    b = np.asarray(b).astype(np.uint8); q = np.asarray(q).astype(np.uint8)
    ok = (b < 4) & (q < 63)
    return np.where(ok, (b << 6) | q, 0xFF).astype(np.uint8)


@pytest.mark.parametrize("ns,n,stride_pad,offset", [(5, 1, 0, 0), (7, 15, 1, 0), (9, 16, 0, 0), (6, 4097, 3, 1),
                                                     (40, 70001, 15, 0), (3, 262144 + 48, 0, 0), (300, 33000, 128 - 33000 % 128, 0),
                                                     (2, 2_100_000, 0, 16)])
def test_packed_tiles_give_the_records_of_the_two_byte_tiles(ctx, ns, n, stride_pad, offset):
    """bvc_lrt_dense_packed / bvc_hist_dense_packed / bvc_pack_dense (one byte per sample): the same histograms and the
    same records, bit for bit, as the two-byte entry points -- aligned and unaligned rows, ragged tails, short rows,
    rows split over several workgroups, uncovered samples, host and device pointers."""
    import torch
    from basevarc_amd.lib import results_from_tensor
    rng = np.random.default_rng(ns * 1000 + n)
    stride = n + stride_pad
    flat_b = torch.full((ns * stride + offset + 64,), -1, dtype=torch.int8, device="cuda")
    flat_q = torch.zeros((ns * stride + offset + 64,), dtype=torch.int8, device="cuda")
    b = flat_b[offset:offset + ns * stride].view(ns, stride)[:, :n]
    q = flat_q[offset:offset + ns * stride].view(ns, stride)[:, :n]
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(11, 77, b, q, r, cov_thr16=int(0.8 * 65536))
    # qualities of the generator are 10..40; add some of 0..62 and sentinels of several kinds
    hb, hq = b.cpu().numpy().copy(), q.cpu().numpy().copy()
    m = rng.random(hb.shape) < 0.05
    hq[m] = rng.integers(0, 63, size=int(m.sum()))
    hb[rng.random(hb.shape) < 0.02] = 4
    b.copy_(torch.from_numpy(hb)); q.copy_(torch.from_numpy(hq))
    m_af = caller_min_af(max(n, 1000))
    want = results_from_tensor(ctx.lrt_dense_device(b, q, r, m_af))
    want_counts = ctx.hist_dense_device(b, q).cpu().numpy()
    # (1) packed on the device by the library, into a tile with its own stride / alignment
    pflat = torch.empty((ns * stride + offset + 64,), dtype=torch.uint8, device="cuda")
    p = pflat[offset:offset + ns * stride].view(ns, stride)[:, :n]
    p2, bad = ctx.pack_dense_device(b, q, p)
    assert bad == 0
    assert np.array_equal(p.cpu().numpy(), _pack_numpy(hb, hq))
    assert np.array_equal(ctx.hist_dense_packed_device(p).cpu().numpy(), want_counts)
    got = results_from_tensor(ctx.lrt_dense_packed_device(p, r, m_af))
    assert got.tobytes() == want.tobytes()
    # (2) host pointers (the producer packs on the host)
    got_h = ctx.lrt_dense_packed(_pack_numpy(hb, hq), r.cpu().numpy(), m_af)
    assert got_h.tobytes() == want.tobytes()


def test_packing_reports_qualities_that_do_not_fit(ctx):
    import torch
    b = torch.zeros((3, 1000), dtype=torch.int8, device="cuda")
    q = torch.full((3, 1000), 30, dtype=torch.int8, device="cuda")
    q[1, 10] = 63; q[2, 999] = 93; q[0, 0] = 62
    b[0, 5] = -1; q[0, 5] = 100                                    # uncovered: its quality does not matter
    p, bad = ctx.pack_dense_device(b, q)
    assert bad == 2
    hp = p.cpu().numpy()
    assert hp[1, 10] == 0xFF and hp[2, 999] == 0xFF and hp[0, 0] == 62 and hp[0, 5] == 0xFF and hp[1, 0] == 30
    # every byte whose quality bits are 63 is "no observation"
    pp = torch.tensor([[0x3F, 0x7F, 0xBF, 0xFF, 0x40 | 20, 0x80 | 62]], dtype=torch.uint8, device="cuda")
    c = ctx.hist_dense_packed_device(pp).cpu().numpy().view(np.uint32)[0]
    assert c.sum() == 2 and c[1 * 128 + 20] == 1 and c[2 * 128 + 62] == 1


def test_packed_full_size_and_overlap(ctx):
    """N = 1e6 (rows split over workgroups when sites are few; long-row overlap mode): packed records == two-byte records."""
    import torch
    from basevarc_amd.lib import results_from_tensor
    n = 1_000_000
    m = caller_min_af(n)
    for ns in (3, 600):
        stride = (n + 127) // 128 * 128
        b = torch.empty((ns, stride), dtype=torch.int8, device="cuda")[:, :n]
        q = torch.empty((ns, stride), dtype=torch.int8, device="cuda")[:, :n]
        r = torch.empty(ns, dtype=torch.int8, device="cuda")
        ctx.synth_dense_device(5, 4242, b, q, r)
        want = results_from_tensor(ctx.lrt_dense_device(b, q, r, m))
        p, bad = ctx.pack_dense_device(b, q)
        assert bad == 0
        assert results_from_tensor(ctx.lrt_dense_packed_device(p, r, m)).tobytes() == want.tobytes()
        ctx.set_overlap(True)
        try:
            outs = [ctx.lrt_dense_packed_device(p, r, m) for _ in range(4)]
            ctx.join(); ctx.synchronize()
        finally:
            ctx.set_overlap(False)
        for o in outs:
            assert results_from_tensor(o).tobytes() == want.tobytes()


def test_overlap_ring_shared_by_every_entry_point(ctx):
    """One context in overlap mode, calls of all four device-pointer entry points (dense, packed, ragged, groups) and of
    growing and shrinking sizes interleaved: they share the ring of histogram buffers and the side streams, scratch is
    re-allocated while earlier calls are still in flight, and every call must return the bytes it returns on its own."""
    import torch
    from basevarc_amd import Context
    from basevarc_amd.lib import GROUP_DTYPE, SITE_DTYPE
    n, k = 210_000, 3
    m = caller_min_af(n)
    g = torch.from_numpy((np.arange(n) % k).astype(np.uint8)).cuda()
    sizes = [64, 300, 32, 520, 128, 700, 16, 256]
    work = []
    for t, ns in enumerate(sizes):
        b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
        q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
        r = torch.empty(ns, dtype=torch.int8, device="cuda")
        ctx.synth_dense_device(13, 9000 * t, b, q, r)
        kind = ("dense", "packed", "csr", "groups")[t % 4]
        extra = None
        if kind == "packed":
            extra, bad = ctx.pack_dense_device(b, q)
            assert bad == 0
        if kind == "csr":
            extra = (torch.arange(ns + 1, dtype=torch.int64) * n).cuda()
        work.append((kind, b, q, r, extra))
    ctx.synchronize()

    def run(c, item, out, gout):
        kind, b, q, r, extra = item
        if kind == "dense":
            c.lrt_dense_device(b, q, r, m, out)
        elif kind == "packed":
            c.lrt_dense_packed_device(extra, r, m, out)
        elif kind == "csr":
            c.lrt_csr_device(extra, b.reshape(-1), q.reshape(-1), r, m, out)
        else:
            c.lrt_dense_groups_device(b, q, r, m, g, k, out, gout)

    def buffers():
        outs = [torch.zeros(it[1].shape[0] * SITE_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for it in work]
        gouts = [torch.zeros(it[1].shape[0] * k * GROUP_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for it in work]
        return outs, gouts

    plain, gplain = buffers()
    for it, o, go in zip(work, plain, gplain):
        run(ctx, it, o, go)
        ctx.synchronize()
    with Context(0) as fresh:                      # fresh context: every scratch buffer starts empty
        fresh.set_overlap(True)
        for rep in range(2):
            outs, gouts = buffers()
            for it, o, go in zip(work, outs, gouts):
                run(fresh, it, o, go)
            fresh.join()
            fresh.synchronize()
            for i, (it, o, go) in enumerate(zip(work, outs, gouts)):
                assert torch.equal(o, plain[i]), (rep, i, it[0])
                if it[0] == "groups":
                    assert torch.equal(go, gplain[i]), (rep, i)


@pytest.mark.parametrize("k", [1, 7, 31, 32])
def test_group_counts_up_to_the_maximum(ctx, k):
    """1 .. BVC_MAX_GROUPS = 32 population groups (33 histograms of one LDS copy each at the top: 66 KiB), labels in any
    order and ordered, rows of 3008 samples (16-byte aligned: the fast kernels) and of 3000 (the byte kernels): the
    records of the oracle's group loop."""
    rng = np.random.default_rng(100 + k)
    ns, m = 6, 0.001
    for n in (3008, 3000):
        sites = [random_site(rng, n, af=0.2) for _ in range(ns)]
        B, Q, R = pad_rows(sites)
        for layout in ("any", "ordered"):
            g = rng.integers(0, k + 1, size=n).astype(np.uint8)   # k = "no group" for some samples
            if layout == "ordered":
                g = np.sort(g)
            g[g == k] = 255
            res, gres = ctx.lrt_dense_groups(B, Q, R, m, g, k)
            for s in range(ns):
                o, gd, ga, ran, pres = orc.dense_site_groups(B[s], Q[s], int(R[s]), m, g, k, use_hist=True)
                assert_site_matches(res[s], o, where=f"k={k} {layout} n={n} site {s}", path_strict=False)
                assert np.array_equal(gres[s]["depth"], gd) and np.array_equal(gres[s]["ran"], ran), (k, layout, n, s)
                np.testing.assert_allclose(gres[s]["af"], ga, rtol=0, atol=AF_ATOL)


@pytest.mark.parametrize("k", [1, 5, 7, 12, 32])          # 2 / 6 / 8 / 13 / 33 histograms: every copy count of the any-order kernels
def test_packed_group_mode_gives_the_records_of_the_two_byte_group_mode(ctx, k):
    """bvc_lrt_dense_groups_packed == bvc_lrt_dense_groups on the same observations, byte for byte (site records and
    group records): labels in any order and ordered by group, with and without samples in no group, rows of 6016
    samples (16-byte aligned: the vector kernels), 6000 (byte kernels) and 70,000; host and device pointers."""
    import torch
    from basevarc_amd.lib import GROUP_DTYPE, results_from_tensor
    rng = np.random.default_rng(500 + k)
    for n, ns in ((6016, 9), (6000, 5), (70000, 40)):
        b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
        q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
        r = torch.empty(ns, dtype=torch.int8, device="cuda")
        ctx.synth_dense_device(17, 31 * n + k, b, q, r, cov_thr16=int(0.9 * 65536))
        p, bad = ctx.pack_dense_device(b, q, torch.empty((ns, n), dtype=torch.uint8, device="cuda"))
        assert bad == 0
        m = caller_min_af(max(n, 20000))
        for layout in ("any", "ordered"):
            g = rng.integers(0, k + 1, size=n).astype(np.uint8)
            if layout == "ordered":
                g = np.sort(g)
            g[g == k] = 255
            gt = torch.from_numpy(g).cuda()
            want, gwant = ctx.lrt_dense_groups_device(b, q, r, m, gt, k)
            got, ggot = ctx.lrt_dense_groups_packed_device(p, r, m, gt, k)
            ctx.synchronize()
            assert torch.equal(got, want), (k, n, layout)
            assert torch.equal(ggot, gwant), (k, n, layout)
            if layout == "any" and n != 6000:                    # the any-order kernel's other shapes: 512-thread workgroups,
                from basevarc_amd import Context                 # fewer LDS copies per histogram (the conflict-free fold's other branch)
                for knob, val in (("group_big_lds", 0), ("group_copies_log2", 1), ("group_copies_log2", 0)):
                    with Context(0) as other:
                        other.set_tuning("group_big_lds", 0)
                        other.set_tuning(knob, val)
                        o1, o2 = other.lrt_dense_groups_packed_device(p, r, m, gt, k)
                        t1, t2 = other.lrt_dense_groups_device(b, q, r, m, gt, k)
                        other.synchronize()
                        assert torch.equal(o1, want) and torch.equal(o2, gwant), (k, n, knob, val)
                        assert torch.equal(t1, want) and torch.equal(t2, gwant), (k, n, knob, val)
            if n <= 6016:
                hres, hg = ctx.lrt_dense_groups_packed(p.cpu().numpy(), r.cpu().numpy(), m, g, k)
                assert hres.tobytes() == results_from_tensor(want).tobytes()
                assert hg.tobytes() == gwant.cpu().numpy().tobytes()


def test_packed_group_mode_at_1e6_samples(ctx):
    """N = 1e6, k = 5, both label orders, overlap mode: packed group records == two-byte group records."""
    import torch
    ns, n, k = 65, 1_000_000, 5
    m = caller_min_af(n)
    stride = (n + 127) // 128 * 128
    b = torch.empty((ns, stride), dtype=torch.int8, device="cuda")[:, :n]
    q = torch.empty((ns, stride), dtype=torch.int8, device="cuda")[:, :n]
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(3, 12345, b, q, r)
    p, bad = ctx.pack_dense_device(b, q)
    assert bad == 0
    for layout in ("interleaved", "ordered"):
        g = _group_labels(n, k, layout)
        g[::97] = 255
        if layout == "ordered":
            g = np.sort(np.minimum(g, k)).astype(np.uint8)
            g[g == k] = 255
        gt = torch.from_numpy(g).cuda()
        want, gwant = ctx.lrt_dense_groups_device(b, q, r, m, gt, k)
        ctx.synchronize()
        ctx.set_overlap(True)
        try:
            outs = [ctx.lrt_dense_groups_packed_device(p, r, m, gt, k) for _ in range(3)]
            ctx.join(); ctx.synchronize()
        finally:
            ctx.set_overlap(False)
        for got, ggot in outs:
            assert torch.equal(got, want) and torch.equal(ggot, gwant), layout


def test_four_host_threads_with_their_own_contexts(ctx):
    """include/bvc.h, "Threading": one context per host thread, no process-wide mutable state.  Four threads, each with
    its own context and HIP stream, hammer different entry points (dense, packed, ragged, groups) at the same time on
    the one device; every call returns what the same call returns alone."""
    import threading
    import torch
    from basevarc_amd import Context
    from basevarc_amd.lib import GROUP_DTYPE, SITE_DTYPE
    n, ns, k = 120_000, 200, 4
    m = caller_min_af(n)
    b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(23, 600, b, q, r, cov_thr16=int(0.7 * 65536))
    p, bad = ctx.pack_dense_device(b, q)
    assert bad == 0
    g = torch.from_numpy((np.arange(n) % k).astype(np.uint8)).cuda()
    offs = (torch.arange(ns + 1, dtype=torch.int64) * n).cuda()
    want = ctx.lrt_dense_device(b, q, r, m).clone()
    want_g = ctx.lrt_dense_groups_device(b, q, r, m, g, k)[1].clone()
    ctx.synchronize()
    errors = []

    def worker(kind):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream), Context(0, stream=stream) as c:
                c.set_overlap(kind in ("dense", "packed"))
                for rep in range(12):
                    out = torch.zeros(ns * SITE_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
                    gout = torch.zeros(ns * k * GROUP_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
                    if kind == "dense":
                        c.lrt_dense_device(b, q, r, m, out)
                    elif kind == "packed":
                        c.lrt_dense_packed_device(p, r, m, out)
                    elif kind == "csr":
                        c.lrt_csr_device(offs, b.reshape(-1), q.reshape(-1), r, m, out)
                    else:
                        c.lrt_dense_groups_packed_device(p, r, m, g, k, out, gout)
                    c.join(); c.synchronize()
                    if not torch.equal(out, want) or (kind == "groups" and not torch.equal(gout, want_g)):
                        errors.append((kind, rep))
        except Exception as e:                                     # noqa: BLE001 -- reported by the main thread
            errors.append((kind, repr(e)))

    threads = [threading.Thread(target=worker, args=(kind,)) for kind in ("dense", "packed", "csr", "groups")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
