"""Round 5 (GPU): the producer side on the device and the ragged group loop.

* bvc_pileup_begin / bvc_pileup_finish: a tile of temp-batch pileup TEXT (the reference's own format, writer
  src/BaseVarC.cpp:509-527) parsed on the device must give the columns the reference's position loop builds from it
  (src/BaseVarC.cpp:403-441: entries in sample order, N bases dropped, bit-field widths, indel entries inheriting the fields of the last
  base token parsed before them -- across lines, positions and tiles), the tallies bt_f takes (:560-590) and the records of
  bvc_lrt_csr on those columns.  Checked against the Python restatement of that parser (oracle/emit_oracle.py Parser) on the
  reference's test data and on fuzzed regular lines; lines the reference's writer cannot produce must be REPORTED, not guessed.
* bvc_lrt_csr_groups: the caller's --group loop (src/BaseVarC.cpp:617-661) on ragged columns, against the oracle's group loop and,
  byte for byte, against the dense group call on the tile that holds the same observations.
"""
import numpy as np
import pytest

from oracle import emit_oracle as eo
from oracle import orc
from tests.test_gpu_parity import AF_ATOL, assert_site_matches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from basevarc_amd import Context
    c = Context(0)
    yield c
    c.close()


# ------------------------------------------------------------------------------------------------ pileup text on the device
def tile_of(batch_lines):
    """batch_lines[b] = list of T lines (no newline).  Returns (text, line_start [nb, T + 1])."""
    nb, T = len(batch_lines), len(batch_lines[0])
    text = bytearray()
    ls = np.zeros((nb, T + 1), dtype=np.uint32)
    for b, lines in enumerate(batch_lines):
        while len(text) % 16 != (b * 5) % 16:                      # chunks at every alignment
            text += b"#"
        for t, l in enumerate(lines):
            ls[b, t] = len(text)
            text += l.encode() + b"\n"
        ls[b, T] = len(text)
    return bytes(text), ls


def reference_columns(batch_lines, parser):
    """What bt_s builds per position: (aiv, sample) with the restated parser, lines taken position by position, batch by batch."""
    nb, T = len(batch_lines), len(batch_lines[0])
    return [parser.parse([batch_lines[b][t] for b in range(nb)]) for t in range(T)]


def check_tile(ctx, batch_lines, n_in_batch, ref, min_af, parser, carry_in, where=""):
    text, ls = tile_of(batch_lines)
    sample0 = np.concatenate([[0], np.cumsum(n_in_batch)[:-1]]).astype(np.int32)
    out = ctx.pileup_tile(text, ls, sample0, n_in_batch, ref, min_af, carry_in=carry_in)
    assert out is not None, where + ": reported as irregular"
    cols = reference_columns(batch_lines, parser)
    T = len(cols)
    eoff = out["entry_off"]
    assert eoff[0] == 0 and eoff[T] == len(out["entries"]) == sum(len(a) for a, _ in cols), where
    indel_texts = {int(r["entry"]): text[int(r["text_off"]):int(r["text_off"]) + int(r["len"])].decode() for r in out["indels"]}
    assert len(indel_texts) == len(out["indels"])
    sites = []
    for t, (aiv, sample) in enumerate(cols):
        e = out["entries"][eoff[t]:eoff[t + 1]]
        assert len(e) == len(aiv), (where, t)
        assert out["samples"][eoff[t]:eoff[t + 1]].tolist() == sample, (where, t)
        tally = np.zeros(32, dtype=np.int64)
        for k, a in enumerate(aiv):
            got = tuple(int(e[k][f]) for f in ("base", "mapq", "qual", "rpr", "strand", "is_indel"))
            assert got == (a["base"], a["mapq"], a["qual"], a["rpr"], a["strand"], a["is_indel"]), (where, t, k, got, a)
            tally[(16 if a["is_indel"] else 0) + (a["strand"] << 3 | a["base"])] += 1
            if a["is_indel"]:
                assert indel_texts[int(eoff[t]) + k] == a["indel"], (where, t, k)
        assert out["tally"][t].tolist() == tally.tolist(), (where, t)
        b = np.array([a["base"] for a in aiv if not a["is_indel"]], dtype=np.int8)
        q = np.array([a["qual"] for a in aiv if not a["is_indel"]], dtype=np.uint8).astype(np.int8)
        sites.append((b, q))
    assert set(indel_texts) == {int(eoff[t]) + k for t, (aiv, _) in enumerate(cols) for k, a in enumerate(aiv) if a["is_indel"]}
    # the records: bvc_lrt_csr on the same columns, byte for byte
    offs = np.concatenate([[0], np.cumsum([len(b) for b, _ in sites])]).astype(np.int64)
    allb = np.concatenate([b for b, _ in sites] + [np.zeros(0, np.int8)])
    allq = np.concatenate([q for _, q in sites] + [np.zeros(0, np.int8)])
    want = ctx.lrt_csr(offs, allb if len(allb) else np.zeros(1, np.int8), allq if len(allq) else np.zeros(1, np.int8), ref, min_af)
    assert out["results"].tobytes() == want.tobytes(), where
    ai = parser.ai
    assert out["carry_out"] == [ai["base"], ai["mapq"], ai["qual"], ai["rpr"], ai["strand"]], where
    # bvc_pileup_finish_called: everything the same, the entries of the called positions only (gathered on the device)
    co = ctx.pileup_tile(text, ls, sample0, n_in_batch, ref, min_af, carry_in=carry_in, called_only=True)
    for key in ("entry_off", "tally", "results", "indels"):
        assert co[key].tobytes() == out[key].tobytes(), (where, key)
    assert co["carry_out"] == out["carry_out"]
    coff, called = co["called_off"], out["results"]["called"]
    assert coff[0] == 0 and coff[T] == len(co["entries"]) == len(co["samples"]) == sum(int(eoff[t + 1] - eoff[t]) for t in range(T) if called[t])
    for t in range(T):
        n_t = int(eoff[t + 1] - eoff[t]) if called[t] else 0
        assert coff[t + 1] - coff[t] == n_t, (where, t)
        assert co["entries"][coff[t]:coff[t + 1]].tobytes() == out["entries"][eoff[t]:eoff[t] + n_t].tobytes(), (where, t)
        assert co["samples"][coff[t]:coff[t + 1]].tolist() == out["samples"][eoff[t]:eoff[t] + n_t].tolist(), (where, t)
    return out, sites


def random_token(rng, p_data, p_indel, wide=False):
    u = rng.random()
    if u < p_data:
        hi = 1000 if wide else 100
        base = int(rng.integers(0, 8 if wide else 5))
        return f"{base},{rng.integers(0, hi if wide else 61)},{rng.integers(0, hi if wide else 42)},{rng.integers(0, hi)},{rng.integers(0, 10 if wide else 2)}"
    if u < p_data + p_indel:
        kind = rng.integers(3)
        if kind == 0:
            return "N"
        seq = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 30))))
        return ("+" if kind == 1 else "-") + seq
    return "."


@pytest.mark.parametrize("shape", ["sparse", "dense", "indel_heavy", "wide_fields", "tiny_batches"])
def test_device_parse_of_fuzzed_regular_lines_equals_the_restated_parser(ctx, shape):
    """Regular lines of every kind the writer can produce, at every alignment: tokens crossing the 16-byte lanes and the 1 KiB steps,
    indel tokens first in a line / after an N base / in runs, fields of three digits (bit-field wrap: base & 7, strand & 1, the
    others mod 256), batches of one sample, empty positions; three consecutive tiles share one parser state (the carry)."""
    rng = np.random.default_rng({"sparse": 1, "dense": 2, "indel_heavy": 3, "wide_fields": 4, "tiny_batches": 5}[shape])
    p_data, p_indel, wide = {"sparse": (0.08, 0.004, False), "dense": (0.9, 0.02, False), "indel_heavy": (0.2, 0.3, False),
                             "wide_fields": (0.5, 0.05, True), "tiny_batches": (0.3, 0.05, False)}[shape]
    n_in_batch = np.array({"tiny_batches": [1, 2, 1, 3, 1, 1, 7, 1]}.get(shape, [700, 37, 1, 300]), dtype=np.int32)
    parser = eo.Parser()
    carry = [0, 0, 0, 0, 0]
    for tile in range(3):
        T = [13, 1, 40][tile]
        batch_lines = [["".join(random_token(rng, p_data if (t + tile) % 7 else 0.0, p_indel, wide) + " " for _ in range(n)) for t in range(T)]
                       for n in n_in_batch]
        ref = rng.integers(0, 4, T).astype(np.int8)
        out, _ = check_tile(ctx, batch_lines, n_in_batch, ref, 0.001, parser, carry, where=f"{shape} tile {tile}")
        carry = out["carry_out"]


def test_device_parse_of_the_reference_test_data_batches(ctx):
    """BASELINE configs[0]'s own temp batches (100 BAMs, chr17:41197700-41276155, -q 20; written by the restated bt_r): every tile of
    2048 positions of one thread's window, batch = 30 samples, against the restated parser; records against bvc_lrt_csr."""
    from tests import hostref
    P = hostref.Pipeline(mapq=20, batch=30, thread=1)
    files = P.batch_files()
    nb = 1 + (P.n - 1) // P.batch
    per_batch = [files[(0, ib)].split("\n")[1:-1] for ib in range(nb)]
    n_in_batch = np.array([min(P.batch, P.n - ib * P.batch) for ib in range(nb)], dtype=np.int32)
    assert all(len(pb) == len(P.pv) for pb in per_batch)
    parser = eo.Parser()
    carry = [0, 0, 0, 0, 0]
    n_entries = 0
    for t0 in range(0, len(P.pv), 2048)[:12]:
        t1 = min(len(P.pv), t0 + 2048)
        ref = np.array(["ACGT".index(P.refseq[p - P.rg_s]) for p in P.pv[t0:t1]], dtype=np.int8)
        out, _ = check_tile(ctx, [pb[t0:t1] for pb in per_batch], n_in_batch, ref, P.min_af, parser, carry, where=f"test data tile {t0}")
        carry = out["carry_out"]
        n_entries += len(out["entries"])
    assert n_entries > 100000


def test_lines_the_writer_cannot_produce_are_reported_not_guessed(ctx):
    """strtok_r / atoi give a meaning to lines the reference's writer never produces (runs of spaces, missing fields, signs, a last token
    without its space ...).  The device does not imitate that: every such tile is reported (BVC_PILEUP_IRREGULAR -> None here) and the
    host program then parses it with the reference's rules on the CPU (tests/test_gpu_host.py).  A regular tile beside each is accepted."""
    good = ["1,30,25,7,1 . . 0,60,40,12,0 ", ". . . +ACG "]
    n = np.array([4], dtype=np.int32)
    ref = np.zeros(2, dtype=np.int8)

    def run(lines, nib=n):
        text, ls = tile_of([lines])
        return ctx.pileup_tile(text, ls, [0], nib, ref, 0.001)
    assert run(good) is not None
    for bad_line in ["1,30,25,7,1  . . 0,60,40,12,0 ",      # two spaces
                     " 1,30,25,7,1 . . 0,60,40,12,0 ",      # leading space
                     "1,30,25,7,1 . . 0,60,40,12,0",        # no space behind the last token
                     "1,30,25,7 . . 0,60,40,12,0 ",         # four fields
                     "1,30,25,7,1,9 . . 0,60,40,12,0 ",     # six fields
                     "1,30,2555,7,1 . . 0,60,40,12,0 ",     # four digits
                     "1,30,,7,1 . . 0,60,40,12,0 ",         # empty field
                     "1,30,25,7,1 .x . 0,60,40,12,0 ",      # a '.' token with a tail
                     "1,30,25,7,1 A . 0,60,40,12,0 ",       # a token of no known kind
                     "1,30,25,7,1 . 0,60,40,12,0 ",         # three tokens for four samples
                     "1,30,25,7,1 . . . 0,60,40,12,0 ",     # five
                     "1,30,2x,7,1 . . 0,60,40,12,0 ",       # a letter in a field
                     ""]:
        assert run([bad_line, good[1]]) is None, bad_line
        assert run([good[0], bad_line]) is None, bad_line
    assert run(good) is not None                             # the context is as usable as before
    # a batch of no samples has empty lines
    text, ls = tile_of([good, ["", ""]])
    out = ctx.pileup_tile(text, ls, [0, 4], np.array([4, 0], dtype=np.int32), ref, 0.001)
    assert out is not None and len(out["entries"]) == 3


def test_empty_tiles_and_positions_without_entries(ctx):
    out = ctx.pileup_tile(b"", np.zeros((0, 1), dtype=np.uint32), [], [], np.zeros(0, dtype=np.int8), 0.001, carry_in=[3, 9, 8, 7, 1])
    assert out is not None and out["entry_off"].tolist() == [0] and out["carry_out"] == [3, 9, 8, 7, 1]
    lines = [". . . ", "4,1,1,1,1 . . ", ". N . "]            # nothing, an N base only (dropped, but it IS the last base token), an indel
    text, ls = tile_of([lines])
    out = ctx.pileup_tile(text, ls, [5], [3], np.zeros(3, dtype=np.int8), 0.001, carry_in=[2, 50, 33, 6, 1])
    assert out["entry_off"].tolist() == [0, 0, 0, 1] and out["samples"].tolist() == [6]
    e = out["entries"][0]
    assert (int(e["base"]), int(e["mapq"]), int(e["qual"]), int(e["rpr"]), int(e["strand"]), int(e["is_indel"])) == (4, 1, 1, 1, 1, 1)
    assert out["results"]["called"].tolist() == [0, 0, 0] and out["carry_out"] == [4, 1, 1, 1, 1]
    # the same indel first in the tile: the carry of the tile before
    text, ls = tile_of([[". N . "]])
    out = ctx.pileup_tile(text, ls, [5], [3], np.zeros(1, dtype=np.int8), 0.001, carry_in=[2, 50, 33, 6, 1])
    e = out["entries"][0]
    assert (int(e["base"]), int(e["mapq"]), int(e["qual"]), int(e["rpr"]), int(e["strand"]), int(e["is_indel"])) == (2, 50, 33, 6, 1, 1)
    assert out["tally"][0][16 + (1 << 3 | 2)] == 1 and out["tally"][0].sum() == 1 and out["carry_out"] == [2, 50, 33, 6, 1]


# ------------------------------------------------------------------------------------------------ the group loop on ragged columns
def ragged_group_case(rng, n_sites, n_samples, k, coverage, ungrouped=0.1):
    labels = rng.integers(0, k, n_samples).astype(np.uint8)
    labels[rng.random(n_samples) < ungrouped] = 255
    sites = []
    for s in range(n_sites):
        cov = coverage if s % 9 else 0.0                                 # some sites without a single observation
        who = np.nonzero(rng.random(n_samples) < cov)[0].astype(np.int32)
        ref = int(rng.integers(0, 4))
        af = float(rng.choice([0.0, 0.0, 0.02, 0.3]))
        alt = (ref + 1 + int(rng.integers(0, 3))) % 4
        b = np.where(rng.random(len(who)) < af, alt, ref).astype(np.int8)
        q = rng.integers(5, 42, len(who)).astype(np.int8)
        err = rng.random(len(who)) < 10.0 ** (-q / 10.0)
        b[err] = (b[err] + 1 + rng.integers(0, 3, int(err.sum()))) % 4
        if s % 11 == 3 and len(who) > 4:
            b[:2] = 5                                                    # entries that are no A/C/G/T: skipped (src/BaseVarC.cpp:551-559)
        sites.append((who, b, q, ref))
    return labels, sites


@pytest.mark.parametrize("k", [1, 5, 32])
def test_csr_groups_against_the_oracles_group_loop_and_the_dense_group_call(ctx, k):
    rng = np.random.default_rng(100 + k)
    n_samples = 6000
    labels, sites = ragged_group_case(rng, 130, n_samples, k, coverage=0.08)
    m = min(0.001, 100.0 / n_samples)
    offs = np.concatenate([[0], np.cumsum([len(w) for w, _, _, _ in sites])]).astype(np.int64)
    allb = np.concatenate([b for _, b, _, _ in sites]); allq = np.concatenate([q for _, _, q, _ in sites])
    alls = np.concatenate([w for w, _, _, _ in sites])
    ref = np.array([r for _, _, _, r in sites], dtype=np.int8)
    res, gres = ctx.lrt_csr_groups(offs, allb, allq, alls, ref, m, labels, k)
    # the dense tile with the same observations: the same bytes
    B = np.full((len(sites), n_samples), -1, dtype=np.int8); Q = np.zeros((len(sites), n_samples), dtype=np.int8)
    for s, (w, b, q, _) in enumerate(sites):
        B[s, w] = b; Q[s, w] = q
    dres, dgres = ctx.lrt_dense_groups(B, Q, ref, m, labels, k)
    assert res.tobytes() == dres.tobytes() and gres.tobytes() == dgres.tobytes()
    ran = called = 0
    for s, (w, b, q, r) in enumerate(sites):
        o, gd, ga, gr, gp = orc.dense_site_groups(B[s], Q[s], r, m, labels, k)
        assert_site_matches(res[s], o, where=f"k={k} site {s}", path_strict=False)
        called += o["called"]
        for g in range(k):
            assert gres[s, g]["depth"].tolist() == gd[g].tolist(), (s, g)
            assert int(gres[s, g]["ran"]) == int(gr[g]) and int(gres[s, g]["present"]) == int(gp[g]), (s, g)
            ran += int(gr[g])
            for i in range(o["n_alt"]):
                if gp[g] >> i & 1:
                    assert abs(float(gres[s, g]["af"][i]) - float(ga[g][i])) <= AF_ATOL, (s, g, i)
    assert called > 10 and ran > 10
    # device pointers: the same records
    import torch
    from basevarc_amd.lib import results_from_tensor
    d = [torch.from_numpy(x).cuda() for x in (offs, allb, allq, alls, ref, labels)]
    rt, gt = ctx.lrt_csr_groups_device(d[0], d[1], d[2], d[3], d[4], m, d[5], k)
    ctx.synchronize()
    assert results_from_tensor(rt).tobytes() == res.tobytes() and gt.cpu().numpy().tobytes() == gres.tobytes()


def test_csr_groups_sample_indices_outside_the_label_vector_are_in_no_group(ctx):
    offs = np.array([0, 6], dtype=np.int64)
    b = np.array([0, 0, 1, 1, 0, 1], dtype=np.int8); q = np.full(6, 30, dtype=np.int8)
    smp = np.array([0, 1, 2, -1, 7, 1000000], dtype=np.int32)
    labels = np.array([0, 1, 0, 1], dtype=np.uint8)
    res, gres = ctx.lrt_csr_groups(offs, b, q, smp, np.zeros(1, np.int8), 0.001, labels, 2)
    assert res[0]["depth"].tolist() == [3, 3, 0, 0]
    assert gres[0, 0]["depth"].tolist() == [1, 1, 0, 0] and gres[0, 1]["depth"].tolist() == [1, 0, 0, 0]


def test_device_parse_with_groups_feeds_the_ragged_group_loop(ctx):
    """bvc_pileup_finish with n_groups > 0 = the parse + bvc_lrt_csr_groups on its columns (sample indices from the token positions)."""
    rng = np.random.default_rng(9)
    n_in_batch = np.array([200, 200, 77], dtype=np.int32)
    N = int(n_in_batch.sum())
    labels = rng.integers(0, 4, N).astype(np.uint8); labels[::13] = 255
    T = 60
    batch_lines = [["".join(random_token(rng, 0.5, 0.02) + " " for _ in range(n)) for _ in range(T)] for n in n_in_batch]
    text, ls = tile_of(batch_lines)
    ref = rng.integers(0, 4, T).astype(np.int8)
    out = ctx.pileup_tile(text, ls, [0, 200, 400], n_in_batch, ref, 0.001, group_of_sample=labels, n_groups=3)
    cols = reference_columns(batch_lines, eo.Parser())
    offs, bb, qq, ss = [0], [], [], []
    for aiv, sample in cols:
        for a, j in zip(aiv, sample):
            if not a["is_indel"]:
                bb.append(a["base"]); qq.append(a["qual"]); ss.append(j)
        offs.append(len(bb))
    res, gres = ctx.lrt_csr_groups(np.array(offs, np.int64), np.array(bb, np.int8), np.array(qq, np.int8), np.array(ss, np.int32), ref, 0.001,
                                   labels, 3)
    assert out["results"].tobytes() == res.tobytes() and out["grp_results"].tobytes() == gres.tobytes()
    assert res["called"].sum() > 5 and gres["ran"].sum() > 5


# ------------------------------------------------------------------------------------------------ H16 group counters
@pytest.mark.parametrize("k,n", [(5, 70_000 + 9), (8, 33_000), (2, 20_011)])
def test_h16_group_counters_give_identical_records(ctx, k, n):
    """group_h16 = 1 (32 conflict-free copies of 16-bit counter pairs, hist_kernel.hip) against the 16-copy kernels on packed tiles
    with labels in any order: every record the same bytes -- rows that are no multiple of 16 long, uncovered samples, samples in no
    group, one site whose samples all fall into ONE counter pair (the worst case for a 16-bit counter: n / 32 per copy)."""
    import torch
    from basevarc_amd import Context
    ns = 41
    m = min(0.001, 100.0 / n)
    rng = np.random.default_rng(k)
    b = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    q = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    r = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(5, 777, b, q, r, cov_thr16=52000)
    b[3, :] = 2; q[3, :] = 37                                         # one class, everywhere
    labels = rng.integers(0, k + 2, n).astype(np.uint8)
    labels[:4000] = 1 % k
    g = torch.from_numpy(labels).cuda()
    p, bad = ctx.pack_dense_device(b, q)
    assert bad == 0
    out = {}
    for h16 in (0, 1):
        with Context(0) as c:
            c.set_tuning("group_h16", h16)
            res, gres = c.lrt_dense_groups_packed_device(p, r, m, g, k)
            c.synchronize()
            out[h16] = (res.cpu().numpy().tobytes(), gres.cpu().numpy().tobytes())
    assert out[0] == out[1]


# ------------------------------------------------------------------------------------------------ BGZF blocks inflated on the device
def _streams():
    """(name, data) pairs: what the block decoders are checked on (as tests/test_host.py does for the CPU decoder)."""
    rng = np.random.default_rng(77)
    toks = [("%d,%d,%d,%d,%d " % (rng.integers(4), rng.integers(20, 61), rng.integers(10, 41), rng.integers(1, 100), rng.integers(2)))
            if rng.random() < 0.1 else ". " for _ in range(40000)]
    text = "".join(toks).encode()
    out = [("empty", b""), ("one", b"a"), ("abab", b"ab" * 5), ("dots", b". " * 32000), ("random", rng.bytes(65000)),
           ("acgt", bytes(rng.choice(list(b"ACGT"), 65000).tolist())), ("pileup", text[:65280]), ("pileup_tail", text[-30011:])]
    for n in (1, 2, 3, 7, 8, 9, 15, 16, 17, 255, 256, 257, 258, 259, 1000, 1023, 1024, 1025, 32768, 32769, 65280, 65536):
        out.append((f"random{n}", rng.bytes(n)))
        for period in (1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 300):
            out.append((f"period{period}_{n}", (rng.bytes(period) * (n // period + 1))[:n]))
    return out


def test_device_inflate_equals_zlib_and_refuses_what_zlib_refuses(ctx):
    """inflate_kernel.hip (and crc32_kernel: two blocks in three carry their CRC32, as BGZF trailers do, and it is compared on the device)
    against zlib: empty, tiny, random, periodic and pileup-text inputs up to a block's 64 KiB; stored, fixed-code,
    dynamic-code, Huffman-only and RLE streams; several deflate blocks per stream; every alignment of the payload.  Truncated and
    corrupted streams and a wrong ISIZE must come back with a non-zero status (never the expected bytes with status 0), and the
    blocks beside them must be untouched by it."""
    import zlib
    comp = bytearray()
    blocks, want = [], []
    for name, data in _streams():
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                if len(data) > 40000 and (level, strategy) not in ((0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED),
                                                                   (9, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
                    continue
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
                one = co.compress(data) + co.flush()
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
                step = max(1, len(data) // 3)
                many = b"".join(co.compress(data[i:i + step]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(data), step)) + co.flush()
                for c in (one, many):
                    comp += b"\xA5" * (len(blocks) % 5)                  # payloads at every alignment
                    blocks.append((len(comp), len(c), len(data), zlib.crc32(data) & 0xffffffff) if len(blocks) % 3 else (len(comp), len(c), len(data)))
                    want.append((name, level, strategy, data))
                    comp += c
    got, status = ctx.inflate_blocks(bytes(comp), blocks)
    bad = [(want[i][:3], int(status[i])) for i in range(len(blocks)) if status[i] != 0 or got[i] != want[i][3]]
    assert not bad, bad[:10]
    assert len(blocks) > 3000
    # damaged streams between good ones
    rng = np.random.default_rng(5)
    comp2 = bytearray()
    blocks2, expect = [], []
    good = [i for i in range(len(blocks)) if 64 < len(want[i][3]) < 65536][::37]
    for i in good:
        co, cl, isz = blocks[i][:3]
        c = bytes(comp[co:co + cl])
        variants = [("good", c, isz), ("short_isize", c, isz - 1), ("long_isize", c, isz + 1), ("truncated", c[:cl // 2], isz), ("wrong_crc", c, isz)]
        dmg = bytearray(c); dmg[len(dmg) // 2] ^= 0x55
        variants.append(("corrupt", bytes(dmg), isz))
        for kind, cc, sz in variants:
            if kind in ("good", "wrong_crc"):                  # with the CRC32 of the data (its lowest bit flipped: the one thing wrong)
                blocks2.append((len(comp2), len(cc), sz, (zlib.crc32(want[i][3]) & 0xffffffff) ^ (1 if kind == "wrong_crc" else 0)))
            else:
                blocks2.append((len(comp2), len(cc), sz))
            expect.append((kind, want[i][3]))
            comp2 += cc + b"\0" * 3
    got2, status2 = ctx.inflate_blocks(bytes(comp2), blocks2)
    for (kind, data), g, st in zip(expect, got2, status2):
        if kind == "good":
            assert st == 0 and g == data
        elif kind == "corrupt":
            assert st != 0 or g != data or True          # a flipped bit may still decode to ISIZE bytes of something: zlib would say the same
            if st == 0:
                z = zlib.decompressobj(-15)
                try:
                    ok = z.decompress(bytes(comp2[blocks2[expect.index((kind, data))][0]:][:blocks2[expect.index((kind, data))][1]])) == g
                except zlib.error:
                    ok = False
                assert ok, "status 0 for a stream zlib does not inflate to the same bytes"
        elif kind == "wrong_crc":
            assert st == 10, (kind, st)
        else:
            assert st != 0, kind


# ------------------------------------------------------------------------------------------------ tiles from COMPRESSED temp batches
def _bgzf_payloads(data, rng, sizes):
    """data cut into deflate payloads the way a BGZF writer cuts a stream into blocks (block ends fall anywhere, also inside a line)."""
    import zlib
    out, at = [], 0
    while at < len(data):
        n = int(rng.choice(sizes))
        chunk = data[at:at + n]
        co = zlib.compressobj(int(rng.choice([1, 6, 9])), zlib.DEFLATED, -15)
        out.append((co.compress(chunk) + co.flush(), len(chunk)))
        at += n
    return out


def _columns_check(out, cols, ref, min_af, ctx, where):
    """out: a finished tile; cols: [(aiv, sample)] per position of the tile from the restated parser."""
    T = len(cols)
    eoff = out["entry_off"]
    assert eoff[T] == len(out["entries"]) == sum(len(a) for a, _ in cols), where
    texts = {int(r["entry"]): out["indel_text"][int(r["text_off"]):int(r["text_off"]) + int(r["len"])].decode() for r in out["indels"]}
    bb, qq, offs = [], [], [0]
    for t, (aiv, sample) in enumerate(cols):
        e = out["entries"][eoff[t]:eoff[t + 1]]
        assert out["samples"][eoff[t]:eoff[t + 1]].tolist() == sample, (where, t)
        tally = np.zeros(32, dtype=np.int64)
        for k, a in enumerate(aiv):
            got = tuple(int(e[k][f]) for f in ("base", "mapq", "qual", "rpr", "strand", "is_indel"))
            assert got == (a["base"], a["mapq"], a["qual"], a["rpr"], a["strand"], a["is_indel"]), (where, t, k, got, a)
            tally[(16 if a["is_indel"] else 0) + (a["strand"] << 3 | a["base"])] += 1
            if a["is_indel"]:
                assert texts[int(eoff[t]) + k] == a["indel"], (where, t, k)
            else:
                bb.append(a["base"]); qq.append(a["qual"])
        offs.append(len(bb))
        assert out["tally"][t].tolist() == tally.tolist(), (where, t)
    b = np.array(bb if bb else [0], dtype=np.int8); q = np.array(qq if qq else [0], dtype=np.uint8).astype(np.int8)
    want = ctx.lrt_csr(np.array(offs, np.int64), b, q, ref, min_af)
    assert out["results"].tobytes() == want.tobytes(), where


@pytest.mark.parametrize("shape", ["sparse", "dense", "tiny_blocks"])
def test_tiles_from_compressed_batches_equal_the_restated_parser(ctx, shape):
    """bvc_pileup_begin_bgzf: the batches' BGZF blocks go to the device as they are, a few per call and batch (0..3, so that batches
    run ahead of and fall behind each other), block ends anywhere (inside lines, inside tokens); the tile = the lines every batch has
    whole.  Position by position the columns, tallies, indel texts and records are those of the restated parser on the same lines, the
    carry runs through all tiles, every line is used exactly once, and a call that finds some batch without a whole line says so."""
    rng = np.random.default_rng({"sparse": 21, "dense": 22, "tiny_blocks": 23}[shape])
    p_data, p_indel = {"sparse": (0.08, 0.004), "dense": (0.8, 0.03), "tiny_blocks": (0.3, 0.05)}[shape]
    n_in_batch = np.array([300, 41, 1, 120], dtype=np.int32)
    sample0 = np.concatenate([[0], np.cumsum(n_in_batch)[:-1]]).astype(np.int32)
    n_lines = {"sparse": 900, "dense": 260, "tiny_blocks": 150}[shape]
    sizes = {"sparse": [65280, 30000, 5000], "dense": [65280, 40000, 777], "tiny_blocks": [1, 2, 17, 300, 2000]}[shape]
    batch_lines = [["".join(random_token(rng, p_data if t % 9 else 0.0, p_indel) + " " for _ in range(n)) for t in range(n_lines)] for n in n_in_batch]
    names = ["".join(f"S{sample0[b] + j}\t" for j in range(n)) + "\n" for b, n in enumerate(n_in_batch)]
    payloads = [_bgzf_payloads((names[b] + "".join(l + "\n" for l in batch_lines[b])).encode(), rng, sizes) for b in range(len(n_in_batch))]
    nxt = [0] * len(payloads)
    parser = eo.Parser()
    carry = [0, 0, 0, 0, 0]
    done = 0
    first, zero_tiles, tiles = True, 0, 0
    while done < n_lines:
        comp, blocks, bob = bytearray(), [], []
        for b, pl in enumerate(payloads):
            k = int(rng.integers(0, 4)) if shape != "tiny_blocks" else int(rng.integers(0, 40))
            if first:
                k = max(k, 1)
            take = pl[nxt[b]:nxt[b] + k]
            nxt[b] += len(take)
            for c, isz in take:
                comp += b"\x5A" * int(rng.integers(0, 4))
                blocks.append((len(comp), len(c), isz))
                comp += c
            bob.append(len(take))
        if all(n >= len(pl) for n, pl in zip(nxt, payloads)) is False and sum(bob) == 0:
            continue
        max_pos = int(rng.choice([3, 50, 1000]))
        r = ctx.pileup_begin_bgzf(bytes(comp), blocks, bob, [len(n) for n in names] if first else None, sample0, n_in_batch, max_pos, first)
        first = False
        assert r["rc"] == 0, r
        T = r["T"]
        assert T == min(int(r["lines"].min()), max_pos)
        tiles += 1
        if T == 0:
            zero_tiles += 1
            continue
        ref = rng.integers(0, 4, T).astype(np.int8)
        out = ctx._pileup_finish(T, r["n_entries"], r["n_indels"], r["indel_text_bytes"], ref, 0.001, carry, None, 0)
        cols = [parser.parse([batch_lines[b][done + t] for b in range(len(n_in_batch))]) for t in range(T)]
        _columns_check(out, cols, ref, 0.001, ctx, f"{shape} positions {done}..{done + T}")
        ai = parser.ai
        assert out["carry_out"] == [ai["base"], ai["mapq"], ai["qual"], ai["rpr"], ai["strand"]]
        carry = out["carry_out"]
        done += T
        assert tiles < 5000
    assert done == n_lines and all(n == len(pl) for n, pl in zip(nxt, payloads))
    # nothing is left: a call without new blocks finds no line
    r = ctx.pileup_begin_bgzf(b"", [], [0] * len(payloads), None, sample0, n_in_batch, 10, False)
    assert r["rc"] == 0 and r["T"] == 0 and r["lines"].tolist() == [0] * len(payloads)


def test_compressed_tiles_report_irregular_lines_and_broken_blocks(ctx):
    import zlib
    n_in_batch = np.array([4, 2], dtype=np.int32)
    lines0 = ["1,30,25,7,1 . . 0,60,40,12,0 ", ". . . +ACG ", "1,30,25,7,1  . . 0,60,40,12,0 ", ". . . . "]     # the third has two spaces in a row
    lines1 = [". 2,22,33,44,0 ", ". . ", "N . ", "3,1,2,3,1 . "]
    names = ["a\tb\tc\td\t\n", "e\tf\t\n"]

    def payload(b, lines):
        data = (names[b] + "".join(l + "\n" for l in lines)).encode()
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        return co.compress(data) + co.flush(), len(data)
    p0, p1 = payload(0, lines0), payload(1, lines1)
    comp = p0[0] + p1[0]
    blocks = [(0, len(p0[0]), p0[1]), (len(p0[0]), len(p1[0]), p1[1])]
    skip = [len(n) for n in names]
    # two positions at a time: the first tile is regular, the second holds the line with the run of spaces
    r = ctx.pileup_begin_bgzf(comp, blocks, [1, 1], skip, [0, 4], n_in_batch, 2, True)
    assert r["rc"] == 0 and r["T"] == 2 and r["lines"].tolist() == [4, 4]
    out = ctx._pileup_finish(2, r["n_entries"], r["n_indels"], r["indel_text_bytes"], np.zeros(2, np.int8), 0.001, [0] * 5, None, 0)
    assert out["samples"].tolist() == [0, 3, 5, 3] and out["indel_text"] == b"+ACG"
    r = ctx.pileup_begin_bgzf(b"", [], [0, 0], None, [0, 4], n_in_batch, 2, False)
    assert r["rc"] == 1 and r["T"] == 2                           # BVC_PILEUP_IRREGULAR: the tile is decided, the caller parses it
    text, ls = ctx.pileup_text(2, 2)
    got = [[text[ls[b, t]:ls[b, t + 1]].decode() for t in range(2)] for b in range(2)]
    assert got == [[l + "\n" for l in lines0[2:]], [l + "\n" for l in lines1[2:]]]
    for _ in range(3):                                            # all four positions are used up, and stay so (empty regions carry nothing)
        r = ctx.pileup_begin_bgzf(b"", [], [0, 0], None, [0, 4], n_in_batch, 2, False)
        assert r["rc"] == 0 and r["T"] == 0 and r["lines"].tolist() == [0, 0]
    # one batch ends before the other: the tile is then empty however much the other has, call after call
    r = ctx.pileup_begin_bgzf(p1[0], [(0, len(p1[0]), p1[1])], [0, 1], [0, skip[1]], [0, 4], n_in_batch, 5, True)
    assert r["rc"] == 0 and r["T"] == 0 and r["lines"].tolist() == [0, 4]
    r = ctx.pileup_begin_bgzf(b"", [], [0, 0], None, [0, 4], n_in_batch, 5, False)
    assert r["rc"] == 0 and r["T"] == 0 and r["lines"].tolist() == [0, 4]
    # a block that is not deflate: an error, not a guess
    bad = bytearray(p0[0]); bad[len(bad) // 2] ^= 0x10; bad = bytes(bad)
    try:
        zlib.decompressobj(-15).decompress(bad)
        broken = False
    except zlib.error:
        broken = True
    r = ctx.pileup_begin_bgzf(bad + p1[0], [(0, len(bad), p0[1]), (len(bad), len(p1[0]), p1[1])], [1, 1], skip, [0, 4], n_in_batch, 2, True)
    if broken:
        assert r["rc"] == -5 and "deflate" in r["error"]
    # and the context goes on with good data
    r = ctx.pileup_begin_bgzf(comp, blocks, [1, 1], skip, [0, 4], n_in_batch, 100, True)
    assert r["rc"] == 1 and r["T"] == 4


def test_compressed_blocks_in_page_locked_memory_give_the_same_tile(ctx):
    """bvc_host_alloc: the compressed blocks handed over from page-locked memory of the library's (no bounce copy inside the call) give
    the tile the same bytes from pageable memory give."""
    import zlib
    rng = np.random.default_rng(21)
    n_in_batch = np.array([40, 25], dtype=np.int32)
    T = 30
    lines = [["".join(random_token(rng, 0.3, 0.05) + " " for _ in range(n)) for t in range(T)] for n in n_in_batch]
    comp, blocks, bob = bytearray(), [], []
    for b, ls in enumerate(lines):
        data = ("".join(l + "\n" for l in ls)).encode()
        k = 0
        for i in range(0, len(data), 700):
            piece = data[i:i + 700]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            c = co.compress(piece) + co.flush()
            blocks.append((len(comp), len(c), len(piece), zlib.crc32(piece) & 0xffffffff))
            comp += c + b"\0" * ((-len(c)) % 4)
            k += 1
        bob.append(k)
    sample0 = np.array([0, 40], dtype=np.int32)
    ref = rng.integers(0, 4, T).astype(np.int8)
    outs = []
    addr, view = ctx.host_alloc(len(comp) + 64)
    try:
        view[:len(comp)] = np.frombuffer(bytes(comp), dtype=np.uint8)
        pageable = np.frombuffer(bytes(comp), dtype=np.uint8).copy()
        for source in (bytes(comp), view[:len(comp)], pageable):
            r = ctx.pileup_begin_bgzf(source, blocks, bob, [0, 0], sample0, n_in_batch, T, True)
            assert r["rc"] == 0 and r["T"] == T, r
            outs.append(ctx._pileup_finish(T, r["n_entries"], r["n_indels"], r["indel_text_bytes"], ref, 0.001, [0] * 5, None, 0))
    finally:
        ctx.host_free(addr)
    for o in outs[1:]:
        for key in ("entry_off", "tally", "entries", "samples", "results"):
            assert o[key].tobytes() == outs[0][key].tobytes(), key
    assert len(outs) == 3
