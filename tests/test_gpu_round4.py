"""GPU parity tests added in round 4 (run with -m gpu on an MI355X).

* The any-order group histogram kernels that arrived late in round 3 -- hist_dense_groups_slots_kernel<4 | 3 | 2> (two-byte
  rows packed in registers, 256 slots x 16 / 8 / 4 LDS copies per histogram) and hist_packed_groups_kernel<4 | 3 | 2, true,
  1024> -- against the ORACLE's restatement of the caller's --group loop (src/BaseVarC.cpp:617-661), not against one another:
  rows long enough for the kernels' full-block path (2 x 1024 chunks of 16 samples per block), labels in any order, samples in
  no group, uncovered samples, and covered samples of quality 63 and more (the sites the slots kernel flags and the general
  kernel redoes).
* The rewritten hist_csr_block_kernel (loads of block k + 1 in flight while block k is counted; head and tail with the first
  block): class counts of ragged sites of every length and alignment around its block size, bit-exact.
* Stage 2's launch shapes (teams of wavefronts per region, sequences of launches underneath a histogram pass) never change a
  record.
* The subset a level does not run (include/bvc.h "em_prune"): records with and without it are the same bytes in every field the
  reference defines, the counts of what was run are the oracle's own.
"""
import numpy as np
import pytest

from oracle import orc
from tests.sitegen import caller_min_af, random_site
from tests.test_gpu_parity import AF_ATOL, assert_path_difference_is_a_tie, assert_site_matches, pad_rows, path_counts_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from basevarc_amd import Context
    c = Context(0)
    yield c
    c.close()


def _long_group_tile(rng, k, ns, n, high_quals):
    """ns sites x n samples at 90 % coverage, a third of the sites polymorphic; labels 0..k-1 in any order, about one sample
    in k + 1 in no group (255); high_quals: a few covered samples of quality 63..93 in every other site."""
    sites = []
    for s in range(ns):
        b, q, r = random_site(rng, n, af=(0.2 if s % 3 == 0 else 0.0), second_af=(0.05 if s % 6 == 0 else 0.0))
        b = b.copy(); q = q.copy()
        b[rng.random(n) >= 0.9] = -1                                  # uncovered
        if high_quals and s % 2 == 1:
            at = rng.choice(n, size=7, replace=False)
            q[at] = rng.integers(63, 94, size=7).astype(np.int8)
            b[at] = np.where(b[at] < 0, r, b[at])                     # ... and covered
        sites.append((b, q, r))
    B, Q, R = pad_rows(sites)
    g = rng.integers(0, k + 1, size=n).astype(np.uint8)
    g[g == k] = 255
    return B, Q, R, g


def _check_groups_against_the_oracle(res, gres, B, Q, R, m, g, k, where):
    for s in range(B.shape[0]):
        o, gd, ga, ran, pres = orc.dense_site_groups(B[s], Q[s], int(R[s]), m, g, k, use_hist=True)
        assert_site_matches(res[s], o, where=f"{where} site {s}", path_strict=False)
        assert np.array_equal(gres[s]["depth"], gd), (where, s)
        assert np.array_equal(gres[s]["ran"], ran), (where, s)
        assert np.array_equal(gres[s]["present"], pres), (where, s)
        np.testing.assert_allclose(gres[s]["af"], ga, rtol=0, atol=AF_ATOL)


@pytest.mark.parametrize("k", [5, 12, 26, 32])        # 6 / 13 / 27 / 33 histograms: 16, 8, 4 and 4 LDS copies of 256 slots
def test_any_order_group_kernels_on_long_rows_against_the_oracle(ctx, k):
    """Two-byte rows: hist_dense_groups_slots_kernel<4> (k = 5), <3> (k = 12), <2> (k = 26, 32) on rows of 70,016 samples =
    two full blocks of 2 x 1024 chunks and a partial one; with qualities of 63 and more in every other site (flagged by the
    slots kernel, redone by hist_dense_groups_kernel).  Every site record and every group record against the oracle's group
    loop."""
    rng = np.random.default_rng(4000 + k)
    ns, n = 6, 70016
    m = caller_min_af(n)
    for high in (False, True):
        B, Q, R, g = _long_group_tile(rng, k, ns, n, high)
        res, gres = ctx.lrt_dense_groups(B, Q, R, m, g, k)
        _check_groups_against_the_oracle(res, gres, B, Q, R, m, g, k, f"two-byte k={k} high={high}")
        # the 512-thread general kernel alone on the same tile: the same records, byte for byte
        from basevarc_amd import Context
        with Context(0) as other:
            other.set_tuning("group_big_lds", 0)
            r2, g2 = other.lrt_dense_groups(B, Q, R, m, g, k)
        assert r2.tobytes() == res.tobytes() and g2.tobytes() == gres.tobytes(), (k, high)


@pytest.mark.parametrize("k", [5, 12, 26, 32])
def test_packed_any_order_group_kernels_on_long_rows_against_the_oracle(ctx, k):
    """Packed rows (one byte per sample): hist_packed_groups_kernel<4 | 3 | 2, true, 1024> on the same kind of tile (no
    quality beyond 62: such tiles stay on the two-byte entry points), against the oracle's group loop."""
    rng = np.random.default_rng(4100 + k)
    ns, n = 6, 70016
    m = caller_min_af(n)
    B, Q, R, g = _long_group_tile(rng, k, ns, n, False)
    P = np.where((B >= 0) & (B < 4), (B.astype(np.uint8) << 6) | Q.astype(np.uint8), 0xFF).astype(np.uint8)
    res, gres = ctx.lrt_dense_groups_packed(P, R, m, g, k)
    _check_groups_against_the_oracle(res, gres, B, Q, R, m, g, k, f"packed k={k}")


def test_ragged_block_kernel_around_its_block_size(ctx):
    """hist_csr_block_kernel takes the sites of 4096 observations and more in blocks of 2 x 512 chunks of 16: lengths just
    below, at and above one, two and three blocks, every start alignment 0..15 (unaligned head and tail of up to 15
    observations), uncovered observations inside, two-byte and one-byte forms, device pointers: records of the oracle, and the
    one-byte records equal to the two-byte ones byte for byte."""
    import torch
    from basevarc_amd.lib import results_from_tensor
    rng = np.random.default_rng(4200)
    block = 2 * 512 * 16
    lengths = [4096, 4097, block - 17, block, block + 1, 2 * block - 1, 2 * block + 16, 3 * block + 5, 40000, 123, 0, 15, 4095]
    sites = []
    for i, n in enumerate(lengths):
        b, q, r = random_site(rng, n, af=0.1 if i % 2 else 0.0)
        b = b.copy()
        if n:
            b[rng.random(n) >= 0.95] = -1
        sites.append((b, q, r))
    refs = np.array([r for _, _, r in sites], dtype=np.int8)
    lens = np.array([len(b) for b, _, _ in sites], dtype=np.int64)
    m = 0.001
    for lead in range(16):
        # `lead` observations that belong to no site in front: every site then starts at alignment (lead + lengths before it) mod 16
        allb = np.concatenate([np.full(lead, 2, np.int8)] + [b for b, _, _ in sites])
        allq = np.concatenate([np.full(lead, 30, np.int8)] + [q for _, q, _ in sites])
        offs = lead + np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        tb, tq = torch.from_numpy(allb).cuda(), torch.from_numpy(allq).cuda()
        to, tr = torch.from_numpy(offs).cuda(), torch.from_numpy(refs).cuda()
        rec = results_from_tensor(ctx.lrt_csr_device(to, tb, tq, tr, m))
        ctx.synchronize()
        for s in range(len(sites)):
            bs, qs = allb[offs[s]:offs[s + 1]], allq[offs[s]:offs[s + 1]]
            exp = orc.hist_lrt(orc.dense_hist(bs, qs), int(refs[s]), m)
            assert_site_matches(rec[s], exp, where=f"lead {lead} site {s} len {len(bs)}", path_strict=False)
        pk = np.where(allb >= 0, (allb.astype(np.uint8) << 6) | allq.astype(np.uint8), 0xFF).astype(np.uint8)
        rec1 = results_from_tensor(ctx.lrt_csr_packed_device(to, torch.from_numpy(pk).cuda(), tr, m))
        ctx.synchronize()
        assert rec1.tobytes() == rec.tobytes(), lead


def test_stage2_launch_shapes_never_change_a_record(ctx):
    """Stage 2 gives a region of eight sites to a team of wavefronts; how many wavefronts a call may hold per CU
    (em_waves_per_cu: the walk underneath a histogram pass is a SEQUENCE of launches) and whether it runs beside a histogram pass
    (overlap mode) are launch policies: the records of a tile are the same bytes under every one of them, for plain calls, calls
    with a site count that leaves the last region partly empty, and group calls (pseudo-sites)."""
    import torch
    from basevarc_amd import Context
    ns, n, k = 1003, 20000, 5
    m = caller_min_af(n)
    b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(11, 777, b, q, r, cov_thr16=int(0.8 * 65536))
    g = torch.from_numpy((np.arange(n) % k).astype(np.uint8)).cuda()
    want = ctx.lrt_dense_device(b, q, r, m)
    gw, ggw = ctx.lrt_dense_groups_device(b, q, r, m, g, k)
    ctx.synchronize()
    for per_cu in (1, 2, 3, 4, 8):
        for overlap in (False, True):
            with Context(0) as other:
                other.set_tuning("em_waves_per_cu", per_cu)
                other.set_overlap(overlap)
                for rep in range(2):
                    got = other.lrt_dense_device(b, q, r, m)
                    g1, g2 = other.lrt_dense_groups_device(b, q, r, m, g, k)
                other.join(); other.synchronize()
                assert torch.equal(got, want), (per_cu, overlap)
                assert torch.equal(g1, gw) and torch.equal(g2, ggw), (per_cu, overlap)


def _without_run_counts(rec):
    """A record array with the two diagnostics that count what was RUN zeroed: everything else is what the reference defines."""
    r = rec.copy()
    r["n_fits"] = 0
    r["n_passes"] = 0
    return r.tobytes()


def test_the_subset_a_level_does_not_run_never_changes_a_record(ctx):
    """A level of the likelihood-ratio test goes on with the first minimum of chi over its subsets (src/BaseType.cpp:97-105).  By
    default the item engine does not run the subset without the deepest allele when an upper bound on its log-likelihood rules
    it out as that minimum (em_prune = 1), and runs it in a second round of the level when the bound cannot (shallow sites).
    Sites of 1 .. 20,000 observations, monomorphic, bi- and tri-allelic, two alleles of equal depth: every field the reference
    defines is the same BYTES with em_prune = 0 (everything run: the counts are then the reference's, strictly) and with
    em_prune = 1 (the counts are then the oracle's pruned ones), both kinds of level outcome occur, and the sites of 3000 observations
    and more run at most 70 % of the reference's passes between them."""
    from basevarc_amd import Context
    rng = np.random.default_rng(4400)
    sites = []
    for s in range(240):
        nind = int(rng.choice([1, 2, 3, 4, 6, 9, 14, 25, 60, 400, 3000, 20000]))
        af = float(rng.choice([0.0, 0.0, 0.01, 0.1, 0.5]))
        b, q, r = random_site(rng, nind, af=af, second_af=(af / 2 if s % 5 == 0 else 0.0))
        if s % 7 == 0 and nind >= 2:                              # two alleles of exactly equal depth, nothing else
            b = np.array([0, 1] * (nind // 2), dtype=np.int8)
            q = q[:len(b)]
        sites.append((b, q, r))
    B, Q, R = pad_rows(sites)
    m = 0.001
    with Context(0) as all_run:
        all_run.set_tuning("em_prune", 0)
        full = all_run.lrt_dense(B, Q, R, m)
    pruned = ctx.lrt_dense(B, Q, R, m)
    assert _without_run_counts(full) == _without_run_counts(pruned)
    second_round = skipped = deep_run = deep_all = 0
    for s, (b, q, r) in enumerate(sites):
        o = orc.basetype_lrt(b, q, r, m)
        assert_site_matches(full[s], o, where=f"all run, site {s}", path_strict=False)
        assert_path_difference_is_a_tie(full[s], o, where=f"all run, site {s}", counts="reference")
        if o["prune_edge"] < 1e-6 or o["tie_gap"] < 1e-9 * max(1.0, abs(o["lr_alt"])):
            continue
        assert (int(full[s]["n_fits"]), int(full[s]["n_passes"])) == (o["n_fits"], o["n_passes"]), s
        assert (int(pruned[s]["n_fits"]), int(pruned[s]["n_passes"])) == (o["n_fits_pruned"], o["n_passes_pruned"]), (s, o)
        skipped += o["n_fits"] - o["n_fits_pruned"]
        second_round += (o["n_fits"] - o["n_fits_pruned"]) < _levels(o)      # some level had to run its last subset
        if len(b) >= 3000:
            deep_run += int(pruned[s]["n_passes"]); deep_all += int(full[s]["n_passes"])
    assert skipped > 100 and second_round > 10, (skipped, second_round)
    assert deep_run <= 0.7 * deep_all, (deep_run, deep_all)


def _levels(o):
    """Nested levels the oracle's site ran: n_fits = 1 + n + (n - 1) + ... over the levels it entered."""
    fits, n, levels = o["n_fits"] - 1, None, 0
    for n0 in (4, 3, 2):
        f, lv, k = 0, 0, n0
        while k >= 2 and f < fits:
            f += k; lv += 1; k -= 1
        if f == fits:
            return lv
    return 0


def test_skipped_subsets_on_every_entry_point(ctx):
    """em_prune 1 against 0 on the other entry points: group calls (pseudo-sites with SetBase candidate lists: one, two and three
    candidates), ragged sites and packed tiles -- site records and group records the same bytes apart from the two run counts."""
    import torch
    from basevarc_amd import Context
    from basevarc_amd.lib import results_from_tensor
    ns, n, k = 501, 30000, 5
    m = caller_min_af(n)
    b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(21, 4321, b, q, r, cov_thr16=int(0.7 * 65536))
    g = torch.from_numpy((np.arange(n) % (k + 1)).astype(np.uint8)).cuda()
    g[g == k] = 255
    lens = torch.full((ns,), n, dtype=torch.int64)
    offs = torch.zeros(ns + 1, dtype=torch.int64); offs[1:] = torch.cumsum(lens, 0)
    out = {}
    for prune in (1, 0):
        with Context(0) as c:
            c.set_tuning("em_prune", prune)
            res, gres = c.lrt_dense_groups_device(b, q, r, m, g, k)
            p, bad = c.pack_dense_device(b, q)
            assert bad == 0
            rp = c.lrt_dense_packed_device(p, r, m)
            rc = c.lrt_csr_device(offs.cuda(), b.reshape(-1), q.reshape(-1), r, m)
            c.synchronize()
            out[prune] = (results_from_tensor(res).copy(), gres.cpu().numpy().tobytes(), results_from_tensor(rp).copy(),
                          results_from_tensor(rc).copy())
    assert out[1][1] == out[0][1]                                  # group records: no run counts in them
    for i in (0, 2, 3):
        assert _without_run_counts(out[1][i]) == _without_run_counts(out[0][i]), i
        assert (out[1][i]["n_passes"] <= out[0][i]["n_passes"]).all() and out[1][i]["n_passes"].sum() < 0.8 * out[0][i]["n_passes"].sum(), i
