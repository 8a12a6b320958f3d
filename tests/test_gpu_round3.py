"""GPU parity tests added in round 3 (run with -m gpu on an MI355X): the item engine of stage 2 (csrc/em_items.hip)
against the one-wavefront-per-site engine and the oracle at its boundaries, the one-byte ragged entry point, the
device-pointer contract of base_comb, and var_qual through the exp() underflow of chisf."""
import math

import numpy as np
import pytest

from oracle import orc
from tests.sitegen import caller_min_af, random_site
from tests.test_gpu_parity import assert_path_difference_is_a_tie, assert_site_matches, pad_rows, _pack_numpy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from basevarc_amd import Context
    c = Context(0)
    yield c
    c.close()


INT_FIELDS = ("called", "n_alt", "alt_base", "depth", "n_kept", "kept", "status", "depth_total")


def _site_with_quals(rng, nind, quals, af=0.02):
    """A site whose observations take exactly the given quality values (every one of them on every base that occurs
    often enough)."""
    b, _, r = random_site(rng, nind, af=af)
    q = np.asarray(quals, dtype=np.int8)[np.arange(nind) % len(quals)]
    rng.shuffle(q)
    return b, q, r


def test_item_engine_agrees_with_the_wave_engine_and_the_oracle(ctx):
    """Stage 2 has two engines (include/bvc.h, "em_engine"): the item engine takes a site when no allele has more than 32
    quality values and no observation has quality 0 or 1, the one-wavefront-per-site kernels take the rest.  A tile that
    mixes both kinds, with a site count that fills the last region only partly: every integer field of the two engines'
    records is identical (pass counts too, unless the oracle shows a tie), AF / chi / var_qual agree to 1e-12 relative,
    and both match the faithful oracle."""
    from basevarc_amd import Context
    rng = np.random.default_rng(31)
    sites = []
    for s in range(151):
        kind = s % 8
        nind = int(rng.choice([3, 17, 60, 400, 3000, 20000]))
        af = float(rng.choice([0.0, 0.0, 0.004, 0.03, 0.3]))
        if kind == 5:
            sites.append(random_site(rng, max(nind, 3000), af=af, qlo=2, qhi=70))        # > 32 values: wave engine
        elif kind == 6:
            sites.append(random_site(rng, nind, af=af, qlo=0, qhi=30))                    # qualities 0 and 1: wave engine
        elif kind == 7:
            sites.append(_site_with_quals(rng, max(nind, 2000), list(range(5, 37)), af=af))  # exactly 32 values: item engine
        elif kind in (3, 4) and s >= 96:
            sites.append(_site_with_quals(rng, nind, [2, 12, 23, 37], af=af))               # binned: with its neighbours a tiny region
        else:
            sites.append(random_site(rng, nind, af=af, second_af=af / 3 if s % 16 == 3 else 0.0))
    B, Q, R = pad_rows(sites)
    m = 0.001
    recs = {}
    for engine in (0, 1):
        with Context(0) as c:
            c.set_tuning("em_engine", engine)
            recs[engine] = c.lrt_dense(B, Q, R, m)
    a, b = recs[0], recs[1]
    ties = 0
    for s, (sb, sq, sr) in enumerate(sites):
        exp = orc.basetype_lrt(sb, sq, sr, m)
        for eng in (0, 1):
            assert_site_matches(recs[eng][s], exp, where=f"engine {eng} site {s}", path_strict=False)
            ties += assert_path_difference_is_a_tie(recs[eng][s], exp, where=f"engine {eng} site {s}",
                                                    counts="reference" if eng == 1 else "default")
        for f in INT_FIELDS:
            assert np.array_equal(a[s][f], b[s][f]), (s, f)
        for f in ("af", "chi", "var_qual", "lr_alt", "base_frq"):
            x, y = np.asarray(a[s][f], dtype=float), np.asarray(b[s][f], dtype=float)
            ok = np.isclose(x, y, rtol=1e-12, atol=1e-9 if f in ("chi", "var_qual") else 1e-14, equal_nan=True)
            assert ok.all(), (s, f, x, y)
    assert ties <= 4


def test_a_sites_record_does_not_depend_on_its_region_neighbours(ctx):
    """Stage 2 takes eight consecutive sites of a call as a region, and a region with a site of 33..48 quality values goes to
    the wide kernel whole.  The narrow and the wide kernel must give a narrow site the same record, bit for bit: the same
    twelve sites as one call, with a 41-value site spliced in at different places (which moves the region boundaries and
    sends different neighbours to the wide kernel), and one by one."""
    rng = np.random.default_rng(5)
    narrow = [random_site(rng, int(n), af=af, qlo=5, qhi=36) for n, af in zip([3000, 40, 700, 9000, 12, 2500, 300, 5000, 80, 1500, 20000, 600],
                                                                              [0.0, 0.3, 0.02, 0.0, 0.5, 0.1, 0.0, 0.004, 0.0, 0.2, 0.0, 0.05])]
    wide = _site_with_quals(rng, 6000, list(range(2, 43)), af=0.05)
    m = 0.001
    B, Q, R = pad_rows(narrow)
    alone = ctx.lrt_dense(B, Q, R, m)
    one_by_one = [ctx.lrt_dense(*pad_rows([s]), m)[0] for s in narrow]
    assert all(a.tobytes() == b.tobytes() for a, b in zip(alone, one_by_one))
    for at in (0, 1, 5, 6, 7, 8, 9, 12):
        mixed = narrow[:at] + [wide] + narrow[at:]
        got = ctx.lrt_dense(*pad_rows(mixed), m)
        rest = [got[i] for i in range(len(mixed)) if i != at]
        assert all(a.tobytes() == b.tobytes() for a, b in zip(alone, rest)), at
        assert_site_matches(got[at], orc.basetype_lrt(wide[0], wide[1], wide[2], m), where=f"wide site at {at}", path_strict=False)


@pytest.mark.parametrize("n_values", [1, 4, 8, 9, 32, 33, 41, 48, 49])
def test_class_capacity_boundaries_of_the_item_engine(ctx, n_values):
    """A region of six sites belongs to a launch by the most quality values any of its sites has on an allele: <= 32 the
    narrow one (two lanes x 16 classes), 33..48 (Illumina's unbinned 41) the wide one (two lanes x 24), and -- with the
    opt-in knob "em_tiny_regions" -- <= 8 the tiny one (binned qualities; one lane per allele); a site with 49 or more goes
    to the one-wavefront-per-site kernels.  Whichever path: the record is the oracle's.  The tile mixes sites of the tested
    width with sites of 20 values, so that regions of different kinds occur in one call; both settings of the knob."""
    rng = np.random.default_rng(n_values)
    sites = []
    for i, af in enumerate((0.0, 0.01, 0.2, 0.0, 0.05, 0.5, 0.0, 0.1, 0.0, 0.3, 0.0, 0.02, 0.0, 0.0)):
        nv = n_values if i % 5 != 4 and i < 8 else 20                 # sites 8.. form a region of narrow sites only
        sites.append(_site_with_quals(rng, 6000, list(range(3, 3 + nv)), af=af))
    n_values_of = [n_values if i % 5 != 4 and i < 8 else 20 for i in range(len(sites))]
    B, Q, R = pad_rows(sites)
    m = caller_min_af(6000)
    from basevarc_amd import Context
    for tiny in (0, 1):
        with Context(0) as c:
            c.set_tuning("em_tiny_regions", tiny)
            got = c.lrt_dense(B, Q, R, m)
        for s, (b, q, r) in enumerate(sites):
            assert len(np.unique(q[b == r])) == n_values_of[s]
            exp = orc.basetype_lrt(b, q, r, m)
            assert_site_matches(got[s], exp, where=f"{n_values} values, site {s}, tiny {tiny}", path_strict=False)
            assert_path_difference_is_a_tie(got[s], exp, where=f"{n_values} values, site {s}")


def test_ragged_sites_at_one_byte_per_observation(ctx):
    """bvc_lrt_csr_packed: the records of bvc_lrt_csr on the same observations, byte for byte, from host and from device
    pointers; sites shorter and longer than the 4096 observations that separate the two ragged histogram kernels, ranges
    that start at every alignment, "no observation" bytes inside; a sample of sites against the oracle."""
    import torch
    rng = np.random.default_rng(77)
    lens = [0, 1, 15, 16, 17, 100, 4095, 4096, 4097, 9000, 70001, 3, 20000, 64, 5000]
    sites = [random_site(rng, n, af=[0.0, 0.02, 0.3][i % 3], qlo=2, qhi=40) for i, n in enumerate(lens)]
    offs = np.zeros(len(sites) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    bases = np.concatenate([s[0] for s in sites]).astype(np.int8)
    quals = np.concatenate([s[1] for s in sites]).astype(np.int8)
    ref = np.array([s[2] for s in sites], dtype=np.int8)
    hole = rng.random(len(bases)) < 0.01                         # observations the producer marked "none"
    bases[hole] = -1
    packed = _pack_numpy(bases, quals)
    assert (packed[hole] == 0xFF).all()
    m = 0.001
    want = ctx.lrt_csr(offs, bases, quals, ref, m)
    got = ctx.lrt_csr_packed(offs, packed, ref, m)
    assert got.tobytes() == want.tobytes()
    for shift in (0, 1, 7):                                      # device pointers, arrays starting off the 16-byte grid
        buf = torch.zeros(len(packed) + 32, dtype=torch.uint8, device="cuda")
        buf[shift:shift + len(packed)] = torch.from_numpy(packed).cuda()
        out = ctx.lrt_csr_packed_device(torch.from_numpy(offs).cuda(), buf[shift:shift + len(packed)], torch.from_numpy(ref).cuda(), m)
        ctx.synchronize()
        assert out.cpu().numpy().tobytes() == want.tobytes(), shift
    for s in (2, 6, 8, 10, 12):
        keep = ~hole[offs[s]:offs[s + 1]]
        exp = orc.basetype_lrt(sites[s][0][keep], sites[s][1][keep], sites[s][2], m)
        assert_site_matches(got[s], exp, where=f"packed ragged site {s}", path_strict=False)
    # overlap mode keeps the two stages of consecutive packed calls apart like every other entry point
    ctx.set_overlap(True)
    outs = [ctx.lrt_csr_packed_device(torch.from_numpy(offs).cuda(), torch.from_numpy(packed).cuda(), torch.from_numpy(ref).cuda(), m)
            for _ in range(4)]
    ctx.join(); ctx.synchronize()
    ctx.set_overlap(False)
    assert all(o.cpu().numpy().tobytes() == want.tobytes() for o in outs)


@pytest.mark.parametrize("overlap", [False, True])
def test_ragged_host_pointer_calls_go_through_in_chunks(overlap):
    """Ragged calls with host pointers stage their sites through two device buffers in chunks, the upload of chunk i + 1
    under the kernels of chunk i; a chunk keeps its element offsets (the kernels get the staging address minus the first
    offset).  With the chunk size turned down ("host_chunk_kib") 400 sites of 0..9000 observations go through in dozens of
    chunks, chunk boundaries at every alignment: the records must be those of the device-pointer call on the whole tile,
    byte for byte -- two-byte, one-byte, and with per-site candidate lists."""
    import torch
    from basevarc_amd import Context
    rng = np.random.default_rng(2024)
    lens = rng.integers(0, 9000, 400)
    lens[[0, 17, 399]] = 0
    lens[5] = 70001
    sites = [random_site(rng, int(n), af=[0.0, 0.02, 0.3][i % 3], qlo=2, qhi=40) for i, n in enumerate(lens)]
    offs = np.zeros(len(sites) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    bases = np.concatenate([s[0] for s in sites]).astype(np.int8)
    quals = np.concatenate([s[1] for s in sites]).astype(np.int8)
    ref = np.array([s[2] for s in sites], dtype=np.int8)
    packed = _pack_numpy(bases, quals)
    comb = np.tile(np.array([0, 1, 2, 3], dtype=np.int8), (len(sites), 1))
    comb[::3] = [2, 0, 3, 1]
    n_comb = (1 + np.arange(len(sites)) % 4).astype(np.uint8)
    m = 0.001
    with Context(0) as one:
        dev = lambda a: torch.from_numpy(a).cuda()
        want = one.lrt_csr_device(dev(offs), dev(bases), dev(quals), dev(ref), m).cpu().numpy().tobytes()
        want_comb = one.lrt_csr(offs, bases, quals, ref, m, comb, n_comb).tobytes()     # one chunk (default chunk size)
    with Context(0) as c:
        c.set_overlap(overlap)
        for kib in (16, 100, 4096):
            c.set_tuning("host_chunk_kib", kib)
            assert c.lrt_csr(offs, bases, quals, ref, m).tobytes() == want, kib
            assert c.lrt_csr_packed(offs, packed, ref, m).tobytes() == want, kib
            assert c.lrt_csr(offs, bases, quals, ref, m, comb, n_comb).tobytes() == want_comb, kib
        c.join(); c.synchronize()


def test_any_order_groups_on_two_byte_rows_with_qualities_the_slots_cannot_hold(ctx):
    """With 4..8 groups and labels in any order, two-byte rows are packed in registers and counted in 256 slots per histogram
    (hist_dense_groups_slots_kernel); a site with a covered sample of quality 63..127 has no place there, is flagged, and
    is redone by the general kernel.  Sites of every kind side by side -- all qualities below 63, one sample at 63 / 64 / 127,
    most samples at 63 and more, uncovered samples scattered, a row length off the 16-sample grid: the records (site and
    group) must be those of the general kernel alone ("group_big_lds" = 0), byte for byte, and agree with the oracle."""
    import torch
    from basevarc_amd import Context
    from basevarc_amd.lib import GROUP_DTYPE, results_from_tensor
    rng = np.random.default_rng(636)
    n, k = 40_000 + 7, 5
    m = caller_min_af(n)
    kinds = ["low", "one63", "one64", "one127", "mostly_high", "low", "holes", "high_holes", "low", "one100"]
    sites = []
    for i, kind in enumerate(kinds * 2):
        b, q, r = random_site(rng, n, af=[0.0, 0.03, 0.3][i % 3], qlo=5, qhi=40)
        if kind.startswith("one"):
            q[rng.integers(0, n)] = int(kind[3:])
        if kind in ("mostly_high", "high_holes"):
            sel = rng.random(n) < 0.7
            q[sel] = rng.integers(63, 128, int(sel.sum()))
        if kind in ("holes", "high_holes"):
            b[rng.random(n) < 0.2] = -1
        sites.append((b, q, r))
    B, Q, R = pad_rows(sites, width=n + 41)
    labels = rng.integers(0, k + 2, n + 41).astype(np.uint8)      # k and k + 1: in no group
    tb, tq, tr, tg = (torch.from_numpy(x).cuda() for x in (B, Q, R, labels))
    got = ctx.lrt_dense_groups_device(tb[:, :n], tq[:, :n], tr, m, tg[:n], k)
    ctx.synchronize()
    with Context(0) as plain:
        plain.set_tuning("group_big_lds", 0)
        want = plain.lrt_dense_groups_device(tb[:, :n], tq[:, :n], tr, m, tg[:n], k)
        plain.synchronize()
        assert all(torch.equal(x, y) for x, y in zip(got, want))
    res = results_from_tensor(got[0])
    gres = got[1].cpu().numpy().view(GROUP_DTYPE).reshape(len(sites), k)
    for s in (1, 3, 4, 7, 9, 14):
        o, gd, ga, ran, pres = orc.dense_site_groups(B[s, :n], Q[s, :n], int(R[s]), m, labels[:n], k, use_hist=True)
        assert_site_matches(res[s], o, where=f"site {s} ({(kinds * 2)[s]})", path_strict=False)
        assert np.array_equal(gres[s]["depth"], gd) and np.array_equal(gres[s]["ran"], ran), s
        np.testing.assert_allclose(gres[s]["af"], ga, rtol=0, atol=1e-6)


def test_base_comb_entries_outside_acgt_are_ignored_with_device_pointers(ctx):
    """include/bvc.h, base_comb contract: with host pointers an entry outside 0..3 is BVC_ERR_ARG; with device pointers
    the kernels apply the reference's rule -- such a candidate has depth 0 (src/BaseType.cpp:79) and falls to the min_af
    filter -- so the record equals the one for the list without the entry.  Both stage-2 engines."""
    import torch
    from basevarc_amd import Context
    from basevarc_amd.lib import BVC_PTR_DEVICE, BvcError, SITE_DTYPE, _dev_ptr
    rng = np.random.default_rng(9)
    sites = [random_site(rng, 800, af=af, second_af=0.05) for af in (0.0, 0.1, 0.3, 0.02)]
    counts = np.stack([orc.dense_hist(b, q) for b, q, _ in sites])
    ref = np.array([r for _, _, r in sites], dtype=np.int8)
    bad = np.array([[0, 7, 1, 2], [-3, 3, 2, 1], [2, 1, 99, 0], [1, 0, 2, 5]], dtype=np.int8)
    n_bad = np.array([4, 4, 3, 4], dtype=np.uint8)
    clean = np.array([[0, 1, 2, 0], [3, 2, 1, 0], [2, 1, 0, 0], [1, 0, 2, 0]], dtype=np.int8)
    n_clean = np.array([3, 3, 2, 3], dtype=np.uint8)
    with pytest.raises(BvcError):
        ctx.lrt_hist(counts, ref, 0.001, bad, n_bad)
    want = ctx.lrt_hist(counts, ref, 0.001, clean, n_clean)
    for engine in (0, 1):
        with Context(0) as c:
            c.set_tuning("em_engine", engine)
            want_e = c.lrt_hist(counts, ref, 0.001, clean, n_clean)          # host pointers, the clean lists
            d = [torch.from_numpy(x).cuda() for x in (counts.astype(np.uint32).view(np.int32), ref, bad, n_bad)]
            out = torch.empty(len(sites) * SITE_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
            c._check(c._L.bvc_lrt_hist(c._h, len(sites), _dev_ptr(d[0]), _dev_ptr(d[1]), 0.001, _dev_ptr(d[2]), _dev_ptr(d[3]),
                                       _dev_ptr(out), BVC_PTR_DEVICE))
            c.synchronize()
            assert out.cpu().numpy().tobytes() == want_e.tobytes(), engine
    for s, (b, q, r) in enumerate(sites):
        exp = orc.basetype_lrt(b, q, r, 0.001, base_comb=clean[s][:n_clean[s]])
        assert_site_matches(want[s], exp, where=f"comb site {s}", path_strict=False)


def test_var_qual_through_the_underflow_of_chisf(ctx):
    """src/BaseType.cpp:127-132: var_qual = -10 log10(chisf(chi, 1)) while chisf is non-zero, 10000 once it is not.
    chisf = kf_gammaq(0.5, chi / 2) ends in exp(-chi / 2 - ...), which underflows to 0 near chi = 1490.  Histograms
    whose chi walks through [1380, 1540]: var_qual equals the oracle's on both sides of the switch, and below it also
    scipy's regularised upper incomplete gamma function (the arithmetic is a third-party restatement: parity unpinned,
    not unguarded)."""
    from scipy.special import gammaincc
    cands = []
    for n in range(1900, 2300, 7):
        for k in range(106, 120):
            h = np.zeros(512, dtype=np.uint32)
            h[0 * 128 + 40] = n - k                                 # reference allele A at Q40
            h[2 * 128 + 40] = k                                     # ALT G at Q40
            cands.append(h)
    counts = np.stack(cands)
    ref = np.zeros(len(cands), dtype=np.int8)
    got = ctx.lrt_hist(counts, ref, 0.001)
    sides = {"below": 0, "above": 0}
    worst = 0.0
    for s in range(len(cands)):
        chi = float(got[s]["chi"])
        if not (1380.0 <= chi <= 1540.0):
            continue
        exp = orc.hist_lrt(counts[s], 0, 0.001)
        assert_site_matches(got[s], exp, where=f"chi {chi:.2f}")
        vq = float(got[s]["var_qual"])
        if vq == 10000.0:
            sides["above"] += 1
            assert orc.chisf(exp["chi"]) == 0.0
        else:
            sides["below"] += 1
            p = gammaincc(0.5, chi / 2.0)
            if p > 1e-300:                                       # scipy reaches denormals; compare where it is normal
                assert vq == pytest.approx(-10.0 * math.log10(p), rel=1e-9), chi
                worst = max(worst, abs(vq + 10.0 * math.log10(p)) / vq)
    assert sides["below"] >= 20 and sides["above"] >= 20, sides
    assert worst < 1e-9
