"""CPU tests of the host side (no GPU): annotation statistics, the temp-batch pileup text format and the
CVG/VCF writers of basevarc_amd/host (C++), against scipy and the independent Python restatement in
oracle/emit_oracle.py."""
import ctypes as C
import math
import os

import numpy as np
import pytest
from scipy import stats as sps

from oracle import emit_oracle as eo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    from basevarc_amd import build as b
    from basevarc_amd.lib import GroupResult, SiteResult
    _, lib = b.build_host()
    L = C.CDLL(lib)
    L.bvchost_erfc.restype = C.c_double; L.bvchost_erfc.argtypes = [C.c_double]
    L.bvchost_normsf.restype = C.c_double; L.bvchost_normsf.argtypes = [C.c_double]
    L.bvchost_fisher_phred.restype = C.c_double; L.bvchost_fisher_phred.argtypes = [C.c_int] * 4
    L.bvchost_fisher_two_sided.restype = C.c_double; L.bvchost_fisher_two_sided.argtypes = [C.c_int] * 4
    L.bvchost_ranksum.restype = C.c_double
    L.bvchost_ranksum.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int]
    L.bvchost_site_parse.restype = C.c_void_p; L.bvchost_site_parse.argtypes = [C.c_char_p, C.c_int32]
    L.bvchost_site_free.argtypes = [C.c_void_p]
    L.bvchost_site_size.restype = C.c_int32; L.bvchost_site_size.argtypes = [C.c_void_p]
    L.bvchost_site_field.restype = C.c_int32; L.bvchost_site_field.argtypes = [C.c_void_p, C.c_int32, C.c_int]
    L.bvchost_cvg_line.restype = C.c_size_t
    L.bvchost_cvg_line.argtypes = [C.c_void_p, C.c_char_p, C.c_int8, C.POINTER(GroupResult), C.c_int, C.c_char_p, C.c_size_t]
    L.bvchost_vcf_line.restype = C.c_size_t
    L.bvchost_vcf_line.argtypes = [C.c_void_p, C.POINTER(SiteResult), C.c_char_p, C.c_int8, C.c_int32, C.c_char_p,
                                   C.c_char_p, C.c_char_p, C.c_size_t]
    L.bvchost_format_token.restype = C.c_size_t
    L.bvchost_format_token.argtypes = [C.c_int] * 5 + [C.c_char_p, C.c_char_p, C.c_size_t]
    return L


def ranksum(H, x, y):
    a = (C.c_double * max(1, len(x)))(*x)
    b = (C.c_double * max(1, len(y)))(*y)
    return H.bvchost_ranksum(a, len(x), b, len(y))


def test_erfc_and_normsf(H):
    for x in np.concatenate([np.linspace(-6, 6, 241), [0.0, 7.5, 12.0, 26.0, 27.0, -27.0]]):
        assert H.bvchost_erfc(float(x)) == pytest.approx(math.erfc(x), rel=2e-8, abs=1e-300)  # the rational form is ~1e-16 absolute
        assert H.bvchost_normsf(float(x)) == pytest.approx(sps.norm.sf(x), rel=2e-8, abs=1e-300)


def test_fisher_exact_against_scipy_and_restatement(H):
    rng = np.random.default_rng(4)
    tables = [(0, 0, 0, 0), (1, 0, 0, 1), (3, 0, 0, 3), (10, 2, 3, 15), (100, 90, 80, 120), (0, 5, 7, 0), (5, 5, 5, 5)]
    tables += [tuple(int(v) for v in rng.integers(0, m, 4)) for m in (4, 12, 40, 300) for _ in range(40)]
    for t in tables:
        two = H.bvchost_fisher_two_sided(*t)
        assert two == pytest.approx(eo.fisher_two_sided(*t), rel=1e-12)
        if sum(t) > 0:
            assert two == pytest.approx(sps.fisher_exact([[t[0], t[1]], [t[2], t[3]]])[1], rel=1e-6, abs=1e-12)
        ph = H.bvchost_fisher_phred(*t)
        assert ph == pytest.approx(eo.bt_fisher_exact(*t), rel=1e-12, abs=1e-12) and ph >= 0.0


def test_rank_sum_test(H):
    rng = np.random.default_rng(8)
    for n1, n2, hi in ((5, 7, 4), (30, 3, 60), (1, 1, 2), (12, 12, 3), (200, 40, 50)):
        for shift in (0, 5):
            x = [float(v) for v in rng.integers(0, hi, n1)]
            y = [float(v) + shift for v in rng.integers(0, hi, n2)]
            got = ranksum(H, x, y)
            assert got == pytest.approx(eo.rank_sum_test(x, y), rel=1e-10, abs=1e-10)
            r = sps.rankdata([-v for v in x + y])                     # average ranks, descending order of value
            z = (r[:n1].sum() - n1 * (n1 + n2 + 1) / 2) / math.sqrt(n1 * n2 * (n1 + n2 + 1) / 12)
            want = -10 * math.log10(2 * sps.norm.sf(abs(z))) if z != 0 else 0.0
            assert got == pytest.approx(min(want, 10000.0), rel=1e-9, abs=1e-9)
    assert math.isnan(ranksum(H, [], [1.0, 2.0])) and math.isnan(ranksum(H, [3.0], []))



def test_rank_sum_with_a_tie_run_longer_than_65536(H):
    """The reference sums a tie run's ranks in an int32 (src/Algorithm.cpp:31-41), which overflows once a run of equal
    values exceeds about 65,536 members -- the regime of CMDB-depth sites, where most reads share mapping quality 60.
    The host library keeps the sum exact (a stated divergence: the reference's value is undefined there).  Pinned
    against the Python restatement, whose integers do not overflow, and against the closed form for one all-equal run."""
    n1, n2 = 70_000, 50_000
    x = [60.0] * n1
    y = [60.0] * (n2 - 3) + [20.0, 30.0, 61.0]
    got = ranksum(H, x, y)
    want = eo.rank_sum_test(x, y)
    assert got == pytest.approx(want, rel=1e-9, abs=1e-9) and 0.0 < got < 10000.0
    # everything tied: every rank is (n + 1) / 2, R1 equals its expectation, z = 0, phred(1) = 0
    assert ranksum(H, [60.0] * n1, [60.0] * n2) == 0.0

def random_site_lines(rng, n_samples, batch, ref, cov=0.6):
    """Batch lines of one position in the temp-file format, plus the per-sample truth."""
    toks = []
    for _ in range(n_samples):
        u = rng.random()
        if u > cov:
            toks.append(None)
        elif u < 0.04:
            toks.append(dict(is_indel=1, indel=str(rng.choice(["+A", "-CT", "+GGA", "N"])), base=5, mapq=0, qual=0, rpr=0, strand=0))
        else:
            b = ref if rng.random() < 0.8 else int(rng.integers(0, 5))
            toks.append(dict(is_indel=0, indel="", base=b, mapq=int(rng.integers(0, 61)), qual=int(rng.integers(2, 42)),
                             rpr=int(rng.integers(1, 151)), strand=int(rng.integers(0, 2))))
    lines = ["".join(eo.format_token(t) for t in toks[i:i + batch]) for i in range(0, n_samples, batch)]
    return lines


def test_token_format_and_parser_with_carry(H):
    buf = C.create_string_buffer(256)
    H.bvchost_format_token(2, 60, 37, 12, 1, None, buf, 256)
    assert buf.value == b"2,60,37,12,1 "
    H.bvchost_format_token(-1, 0, 0, 0, 0, None, buf, 256)
    assert buf.value == b". "
    H.bvchost_format_token(5, 0, 0, 0, 0, b"-ACG", buf, 256)
    assert buf.value == b"-ACG "
    rng = np.random.default_rng(12)
    H.bvchost_reset_parser()
    P = eo.Parser()
    for _ in range(60):                                        # the carry runs across positions, as in bt_s
        lines = random_site_lines(rng, 37, 10, int(rng.integers(0, 4)))
        aiv, sample = P.parse(lines)
        s = H.bvchost_site_parse("\n".join(lines).encode(), 100)
        assert H.bvchost_site_size(s) == len(aiv)
        for k, (a, j) in enumerate(zip(aiv, sample)):
            got = [H.bvchost_site_field(s, k, f) for f in range(7)]
            assert got == [a["base"], a["mapq"], a["qual"], a["rpr"], a["strand"], a["is_indel"], j]
        H.bvchost_site_free(s)


def test_cvg_and_vcf_lines_match_the_python_restatement(H):
    from basevarc_amd.lib import GroupResult, SiteResult
    from oracle import orc
    rng = np.random.default_rng(21)
    H.bvchost_reset_parser()
    P = eo.Parser()
    n_vcf = 0
    for it in range(120):
        n = int(rng.choice([8, 40, 150]))
        ref = int(rng.integers(0, 4))
        lines = random_site_lines(rng, n, 10, ref, cov=float(rng.choice([0.3, 0.9])))
        aiv, sample = P.parse(lines)
        s = H.bvchost_site_parse("\n".join(lines).encode(), 4242 + it)
        ng = int(rng.choice([0, 3]))
        gd = rng.integers(0, 50, (ng, 4)).astype(np.int32)
        grp = (GroupResult * max(1, ng))()
        for g in range(ng):
            for j in range(4):
                grp[g].depth[j] = int(gd[g, j])
        buf = C.create_string_buffer(1 << 16)
        H.bvchost_cvg_line(s, b"chr17", ref, grp if ng else None, ng, buf, len(buf))
        assert buf.value.decode() == eo.cvg_line("chr17", 4242 + it, ref, aiv, gd.tolist() if ng else None)
        b = [a["base"] for a in aiv if not a["is_indel"]]
        q = [a["qual"] for a in aiv if not a["is_indel"]]
        e = orc.basetype_lrt(b, q, ref, 0.001)
        if e["called"]:
            n_vcf += 1
            r = SiteResult()
            r.var_qual, r.depth_total, r.n_alt, r.called = e["var_qual"], e["depth_total"], e["n_alt"], 1
            for j in range(4):
                r.depth[j] = e["depth"][j]
            for i2 in range(e["n_alt"]):
                r.alt_base[i2] = e["alt_base"][i2]
                r.af[i2] = e["af"][i2]
            extra = {"EAS_AF": "0.100000,0", "AFR_AF": "0"} if ng else {}
            H.bvchost_vcf_line(s, C.byref(r), b"chr17", ref, n, ";".join(extra).encode() or None,
                               ";".join(extra.values()).encode() or None, buf, len(buf))
            assert buf.value.decode() == eo.vcf_line(e, "chr17", 4242 + it, ref, aiv, sample, n, extra)
        H.bvchost_site_free(s)
    assert n_vcf > 10


# ----------------------------------------------------------------------------- phase 1 end to end (no GPU needed)
@pytest.mark.parametrize("thread,batch", [(1, 10), (4, 10), (3, 37)])
def test_load_phase_writes_the_temp_batch_files_of_the_test_data(tmp_path, thread, batch):
    """`BaseVarC basetype --load` on the reference's test BAMs (test/test.sh:3): BGZF/BAM decoding, read
    filters, the first-usable-read pileup rule and the temp-batch text, against the Python restatement."""
    import gzip
    import subprocess
    from basevarc_amd import build as b
    from tests import hostref
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    lst = hostref.write_bam_list(str(tmp_path / "bam.list"))
    out = str(tmp_path / "test.out")
    subprocess.run([exe, "basetype", "--load", "-q", "20", "-t", str(thread), "-b", str(batch), "-i", lst, "-s",
                    hostref.REGION, "-r", fa, "-o", out], check=True, capture_output=True)
    pipe = hostref.Pipeline(mapq=20, batch=batch, thread=thread)
    want = pipe.batch_files()
    assert len(want) == thread * (1 + (pipe.n - 1) // batch)
    for (t, ib), text in want.items():
        f = f"{out}.tmp.thread.{t}/batch.{ib}"
        raw = open(f, "rb").read()
        assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC" and raw[-28:-12] == raw[-28:][:16]   # BGZF + EOF block
        assert gzip.decompress(raw).decode() == text, (t, ib)


# ----------------------------------------------------------------------------- the per-sample pileup rule (f4)
def _token(e):
    if e is None:
        return ". "
    if e["is_indel"]:
        return e["indel"] + " "
    return f"{e['base']},{e['mapq']},{e['qual']},{e['rpr']},{e['strand']} "


def _random_read(rng, pos, rich):
    """A read at 0-based `pos` with a CIGAR over M I D N S H P = X (rich) or M I D S (plain)."""
    ops = []
    if rich and rng.random() < 0.2:
        ops.append(("H", int(rng.integers(1, 6))))
    if rng.random() < 0.4:
        ops.append(("S", int(rng.integers(1, 9))))
    ops.append((str(rng.choice(list("M=X") if rich else ["M"])), int(rng.integers(1, 30))))
    for _ in range(int(rng.integers(0, 6))):
        op = str(rng.choice(list("MIDNP=X") if rich else list("MID")))
        if op == ops[-1][0]:
            continue
        ops.append((op, int(rng.integers(1, 12 if op in "MX=" else 5))))
    if ops[-1][0] not in "M=X":
        ops.append(("M", int(rng.integers(1, 20))))
    if rng.random() < 0.3:
        ops.append(("S", int(rng.integers(1, 9))))
    if rich and rng.random() < 0.15:
        ops.append(("H", int(rng.integers(1, 6))))
    qlen = sum(l for o, l in ops if o in "MIS=X")
    seq = "".join(rng.choice(list("ACGTN"), qlen, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
    qual = [int(x) for x in rng.integers(2, 42, qlen)]
    flag = int(rng.choice([0, 16, 32, 99, 147, 163]))
    return dict(pos=pos, flag=flag, mapq=int(rng.integers(20, 61)), cigar=ops, seq=seq, qual=qual)


@pytest.mark.parametrize("rich", [False, True])
def test_pileup_rule_matches_the_statement_by_statement_restatement(H, rich):
    """The host's single-pass pileup walk (prefix tables + forward cursors, basevarc_amd/host/bam.cpp) against the
    restatement of BamProcess::FindSnpAtPos / GetAllele / GetOffset that rescans every CIGAR per position
    (tests/golden/make_testdata_pileup.py, src/BamProcess.cpp:4-94, 214-261): first covering read, deletion
    fall-through to the next read, indel tokens, soft clips / pads / '=' counted the way the reference counts them."""
    from tests.golden import make_testdata_pileup as ref
    H.bvchost_pileup_tokens.restype = C.c_size_t
    H.bvchost_pileup_tokens.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_char_p, C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(77 + rich)
    checked = indels = fall = 0
    for case in range(60):
        rg_s = int(rng.integers(100, 200))
        span = int(rng.integers(50, 400))
        refseq = "".join(rng.choice(list("ACGT"), span + 1200))
        n_reads = int(rng.integers(1, 25))
        starts = np.sort(rng.integers(rg_s - 30, rg_s + span, n_reads))
        rv = [_random_read(rng, int(p), rich) for p in starts]
        pv = [p for p in range(rg_s, rg_s + span) if rng.random() < 0.9]
        try:
            want = ref.find_snp_at_pos(rv, pv, refseq=refseq, rg_s=rg_s)
        except (IndexError, AssertionError):
            continue                                  # the restatement hit the reference's out-of-range / assert case
        text = "".join(f"{r['pos']} {r['flag']} {r['mapq']} " + "".join(f"{l}{o}" for o, l in r["cigar"]) +
                       f" {r['seq']} " + ",".join(map(str, r["qual"])) + "\n" for r in rv)
        buf = C.create_string_buffer(1 << 20)
        arr = (C.c_int32 * len(pv))(*pv)
        H.bvchost_pileup_tokens(text.encode(), arr, len(pv), rg_s, refseq.encode(), buf, len(buf))
        got = buf.value.decode()
        exp = "".join(_token(want.get(p)) for p in pv)
        assert got == exp, (case, got[:300], exp[:300])
        checked += len(pv)
        indels += sum(1 for e in want.values() if e["is_indel"])
        # a position whose first covering read has a deletion there and whose entry comes from a later read
        fall += sum(1 for p, e in want.items() if not e["is_indel"] and any(
            r["pos"] + 1 <= p <= ref.end_pos(r) for r in rv) and e["mapq"] != next(
            r for r in rv if ref.end_pos(r) >= p)["mapq"])
    assert checked > 3000 and indels > 20 and fall > 0, (checked, indels, fall)


def test_truncated_and_corrupt_bam_files_are_rejected_not_read_past(tmp_path):
    """ADVICE (bam.cpp): every length field is checked against the block it sits in.  A BAM cut in the middle of a
    record, and one whose l_seq / n_cigar fields are garbage, must end `--load` with an error message and exit
    code 1 -- not a crash, not a silent short read.  (tools/sanitize_cpu.sh runs this under ASan + UBSan.)"""
    import gzip
    import shutil
    import struct
    import subprocess
    from basevarc_amd import build as b
    from tests import hostref
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    src = open(hostref.write_bam_list(str(tmp_path / "bam.list"))).read().split()[0]
    raw = gzip.decompress(open(src, "rb").read())        # BGZF is a series of gzip members

    def bgzf(data, path):
        from tools.host_bench import _bgzf_write
        tmp = str(path) + ".raw"
        open(tmp, "wb").write(data)
        _bgzf_write(tmp, str(path))
        os.remove(tmp)

    # locate the first record
    l_text = struct.unpack_from("<i", raw, 4)[0]
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, off)[0]
    off += 4
    for _ in range(n_ref):
        l = struct.unpack_from("<i", raw, off)[0]
        off += 4 + l + 4
    bs = struct.unpack_from("<i", raw, off)[0]
    cases = {
        "cut_mid_record": raw[:off + 4 + bs // 2],
        "huge_l_seq": raw[:off + 4 + 16] + struct.pack("<i", 1 << 30) + raw[off + 4 + 20:],
        "negative_l_seq": raw[:off + 4 + 16] + struct.pack("<i", -5) + raw[off + 4 + 20:],
        "huge_n_cigar": raw[:off + 4 + 12] + struct.pack("<H", 65535) + raw[off + 4 + 14:],
        "negative_l_text": raw[:4] + struct.pack("<i", -1) + raw[8:],
    }
    for name, data in cases.items():
        d = tmp_path / name
        d.mkdir()
        bad = d / "bad.bam"
        bgzf(data, bad)
        if os.path.exists(src + ".bai"):
            shutil.copy(src + ".bai", str(bad) + ".bai")
        lst = d / "bam.list"
        lst.write_text(str(bad) + "\n")
        r = subprocess.run([exe, "basetype", "--load", "-q", "20", "-t", "1", "-b", "10", "-i", str(lst), "-s",
                            hostref.REGION, "-r", fa, "-o", str(d / "out")], capture_output=True, text=True)
        assert r.returncode == 1, (name, r.returncode, r.stderr[-500:])
        assert "ERROR" in r.stderr or "can not open" in r.stderr, (name, r.stderr[-500:])


# ----------------------------------------------------------------------------- binary temp-batch form (additive)
def _decode_bin_batch(raw):
    """Python reader of the `--tmp-format bin` stream (basevarc_amd/host/pileup.h): -> (names line, n_in_batch,
    per position: list of (sample, base, mapq, qual, rpr, strand, indel or None))."""
    import struct
    assert raw[:8] == b"BVCBAT1\n"
    n_in, l = struct.unpack_from("<II", raw, 8)
    names = raw[16:16 + l].decode()
    off = 16 + l
    positions = []
    while off < len(raw):
        (n,) = struct.unpack_from("<I", raw, off)
        off += 4
        end, ents = off + n, []
        while off < end:
            j, base, mapq, qual, rpr, flags = struct.unpack_from("<IBBBBB", raw, off)
            off += 9
            indel = None
            if flags & 2:
                (k,) = struct.unpack_from("<H", raw, off)
                indel = raw[off + 2:off + 2 + k].decode()
                off += 2 + k
            ents.append((j, base, mapq, qual, rpr, flags & 1, indel))
        assert off == end
        positions.append(ents)
    return names, n_in, positions


def test_binary_temp_batches_hold_what_the_text_batches_hold(tmp_path):
    """`--load --tmp-format bin`: same file names, BGZF framing and EOF block as the text form (so --rerun's
    completeness check works on them), and position by position the entries of the text tokens."""
    import gzip
    import subprocess
    from basevarc_amd import build as b
    from tests import hostref
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    lst = hostref.write_bam_list(str(tmp_path / "bam.list"))
    thread, batch = 3, 37
    outs = {}
    for fmt in ("text", "bin", "raw"):
        outs[fmt] = str(tmp_path / f"{fmt}.out")
        subprocess.run([exe, "basetype", "--load", "-q", "20", "-t", str(thread), "-b", str(batch), "-i", lst, "-s",
                        hostref.REGION, "-r", fa, "-o", outs[fmt], "--tmp-format", fmt], check=True, capture_output=True)
    n_indel = n_base = 0
    for t in range(thread):
        for ib in range(3):
            raw_t = open(f"{outs['text']}.tmp.thread.{t}/batch.{ib}", "rb").read()
            raw_b = open(f"{outs['bin']}.tmp.thread.{t}/batch.{ib}", "rb").read()
            assert raw_b[:4] == b"\x1f\x8b\x08\x04" and raw_b[12:14] == b"BC" and raw_b[-28:] == raw_t[-28:]
            lines = gzip.decompress(raw_t).decode().split("\n")
            names, n_in, positions = _decode_bin_batch(gzip.decompress(raw_b))
            assert names == lines[0] + "\n" and n_in == lines[0].count("\t")
            body = lines[1:-1]
            assert len(positions) == len(body)
            for ents, line in zip(positions, body):
                toks = line.split(" ")[:-1]
                assert len(toks) == n_in
                want = []
                for j, tok in enumerate(toks):
                    if tok == ".":
                        continue
                    if tok[0] in "+-N":
                        want.append((j, tok))
                        n_indel += 1
                    else:
                        want.append((j,) + tuple(int(x) for x in tok.split(",")))
                        n_base += 1
                got = [(e[0], e[6]) if e[6] is not None else (e[0], e[1], e[2], e[3], e[4], e[5]) for e in ents]
                assert got == want
            # "raw": the same records in stored BGZF blocks
            raw_r = open(f"{outs['raw']}.tmp.thread.{t}/batch.{ib}", "rb").read()
            assert raw_r[-28:] == raw_t[-28:] and gzip.decompress(raw_r) == gzip.decompress(raw_b)
    assert n_base > 100000 and n_indel > 10
    assert os.path.getsize(f"{outs['bin']}.tmp.thread.0/batch.0") < os.path.getsize(f"{outs['text']}.tmp.thread.0/batch.0")


def test_binary_and_text_parsers_build_the_same_site_columns(H):
    """parse_pileup_bin reproduces the text parser, including the one long-lived AlleleInfo whose fields an indel
    entry inherits across samples, batches and positions (src/BaseVarC.cpp:392, 407-440) and the dropped N bases."""
    import struct
    H.bvchost_site_parse_bin.restype = C.c_void_p
    H.bvchost_site_parse_bin.argtypes = [C.c_char_p, C.c_size_t, C.c_int32, C.c_int32]
    rng = np.random.default_rng(3)
    for trial in range(30):
        H.bvchost_reset_parser()
        text_sites, bin_sites = [], []
        script = []                                               # the same entries for both parsers, position by position
        for p in range(12):
            n_batches = int(rng.integers(1, 4))
            batches = []
            for _ in range(n_batches):
                n_in = int(rng.integers(1, 9))
                ents = []
                for j in range(n_in):
                    u = rng.random()
                    if u < 0.35:
                        continue
                    if u < 0.5:
                        ents.append((j, None, str(rng.choice(["+AC", "-T", "+GGGT", "-"]))))
                    else:
                        ents.append((j, (int(rng.integers(0, 5)), int(rng.integers(0, 61)), int(rng.integers(0, 42)),
                                         int(rng.integers(1, 151)), int(rng.integers(0, 2))), None))
                batches.append((n_in, ents))
            script.append(batches)
        for fmt in ("text", "bin"):
            H.bvchost_reset_parser()
            for p, batches in enumerate(script):
                if fmt == "text":
                    lines = []
                    for n_in, ents in batches:
                        d = {j: (b, i) for j, b, i in ents}
                        lines.append("".join((". " if j not in d else (d[j][1] + " " if d[j][0] is None else
                                              ",".join(map(str, d[j][0])) + " ")) for j in range(n_in)))
                    h = H.bvchost_site_parse("\n".join(lines).encode(), p)
                else:
                    blob, sizes = b"", []
                    for n_in, ents in batches:
                        pay = b""
                        for j, bq, indel in ents:
                            if indel is None:
                                pay += struct.pack("<IBBBBB", j, bq[0], bq[1], bq[2], bq[3], bq[4])
                            else:
                                pay += struct.pack("<IBBBBBH", j, 5, 0, 0, 0, 2, len(indel)) + indel.encode()
                        blob += struct.pack("<II", n_in, len(pay)) + pay
                    h = H.bvchost_site_parse_bin(blob, len(blob), len(batches), p)
                n = H.bvchost_site_size(h)
                rec = [tuple(H.bvchost_site_field(h, k, f) for f in range(7)) for k in range(n)]
                (text_sites if fmt == "text" else bin_sites).append(rec)
                H.bvchost_site_free(h)
        assert text_sites == bin_sites, trial


# ----------------------------------------------------------------------------- --rerun (src/BaseVarC.cpp:218-246)
def test_rerun_resumes_extraction_at_the_first_incomplete_batch(tmp_path):
    """--rerun looks at every <out>.tmp.thread.<t>/batch.<b>: a batch counts as done only when all its per-thread
    files are BGZF with the EOF block.  Extraction restarts at the FIRST incomplete batch and redoes everything from
    there (src/BaseVarC.cpp:218-246); with every file complete nothing is extracted."""
    import gzip
    import subprocess
    import time
    from basevarc_amd import build as b
    from tests import hostref
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    lst = hostref.write_bam_list(str(tmp_path / "bam.list"))
    out = str(tmp_path / "test.out")
    thread, batch = 2, 10
    base = [exe, "basetype", "-q", "20", "-t", str(thread), "-b", str(batch), "-i", lst, "-s", hostref.REGION, "-r", fa,
            "-o", out, "--load"]
    subprocess.run(base, check=True, capture_output=True)
    files = {(t, ib): f"{out}.tmp.thread.{t}/batch.{ib}" for t in range(thread) for ib in range(10)}
    good = {k: open(f, "rb").read() for k, f in files.items()}
    # every file complete: --rerun extracts nothing
    r = subprocess.run(base + ["--rerun"], check=True, capture_output=True, text=True)
    assert "begin to extract reads from bam" not in r.stderr
    # damage: batch 3 of thread 0 loses its EOF block (a writer that was killed), batch 7 of thread 1 is gone
    open(files[(0, 3)], "wb").write(good[(0, 3)][:-28])
    os.remove(files[(1, 7)])
    old = time.time() - 3600
    for f in files.values():
        if os.path.exists(f):
            os.utime(f, (old, old))
    r = subprocess.run(base + ["--rerun"], check=True, capture_output=True, text=True)
    assert "begin to extract reads from bam" in r.stderr
    for (t, ib), f in files.items():
        assert open(f, "rb").read() == good[(t, ib)], (t, ib)              # everything is whole again
        rewritten = os.path.getmtime(f) > old + 1800
        assert rewritten == (ib >= 3), (t, ib, rewritten)                  # batches 0-2 untouched, 3-9 redone
    # without --rerun everything is extracted again, whatever is there
    for f in files.values():
        os.utime(f, (old, old))
    subprocess.run(base, check=True, capture_output=True)
    assert all(os.path.getmtime(f) > old + 1800 for f in files.values())


@pytest.mark.parametrize("background", [0, 1, 3])
def test_bgzf_writer_with_its_own_deflate_thread_writes_the_same_stream(H, tmp_path, background):
    """The compute phase's output writers deflate on a thread of their own, the VCF writer on three (BgzfWriter, background mode:
    `background` = the number of deflating threads; blocks reach the file in the order they were handed over).  Whatever the
    piece size of the writes, the file is valid BGZF with the EOF block and inflates to exactly the bytes written; the
    background and the foreground writer produce identical files (blocks are cut at the same 0xff00-byte marks)."""
    import gzip
    rng = np.random.default_rng(3)
    text = ("\t".join(str(x) for x in rng.integers(0, 50, 400000)) + "\n").encode() * 3          # ~3.4 MB, compressible
    blob = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()                                   # incompressible tail
    data = text + blob
    H.bvchost_bgzf_write.restype = C.c_int
    H.bvchost_bgzf_write.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int]
    files = []
    for piece in (1 << 20, 70001, 513):
        f = str(tmp_path / f"w{background}_{piece}.gz")
        assert H.bvchost_bgzf_write(f.encode(), data, len(data), piece, 6, background) == 1
        raw = open(f, "rb").read()
        assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC" and raw[-28:-12] == raw[-28:][:16]
        assert gzip.decompress(raw) == data
        files.append(raw)
    assert files[0] == files[1] == files[2]
    ref = str(tmp_path / "fg.gz")
    assert H.bvchost_bgzf_write(ref.encode(), data, len(data), 99999, 6, 0) == 1
    assert open(ref, "rb").read() == files[0]


def test_bgzf_readers_with_read_ahead_give_the_same_bytes(H, tmp_path):
    """The position loop's temp-batch readers have their next block inflated ahead by a small thread pool (InflatePool).
    Several files read round robin -- by line and in odd-sized pieces, with a seek back to the start in the middle -- must
    give exactly the bytes of the readers without read-ahead, in the same order, for 1, 2 and 5 pool threads."""
    rng = np.random.default_rng(9)
    H.bvchost_bgzf_write.restype = C.c_int
    H.bvchost_bgzf_write.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int]
    H.bvchost_bgzf_read_hash.restype = C.c_uint64
    H.bvchost_bgzf_read_hash.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                         C.POINTER(C.c_int64)]
    paths, total = [], 0
    for i, n_lines in enumerate((30000, 0, 1, 700, 9000, 25000, 3)):
        lines = [" ".join(("." if rng.random() < 0.9 else "0,40,%d,7,1" % rng.integers(10, 41)) for _ in range(int(rng.integers(1, 60))))
                 for _ in range(n_lines)]
        data = ("".join(l + "\n" for l in lines)).encode()
        f = str(tmp_path / f"b{i}.gz")
        assert H.bvchost_bgzf_write(f.encode(), data, len(data), 65537, [6, 0][i % 2], 0) == 1
        paths.append(f.encode()); total += len(data)
    arr = (C.c_char_p * len(paths))(*paths)
    for by_line, piece in ((1, 0), (0, 777), (0, 200001)):
        want_bytes = C.c_int64(0)
        want = H.bvchost_bgzf_read_hash(arr, len(paths), 0, by_line, piece, 5, C.byref(want_bytes))
        assert want_bytes.value > total                    # file 0.. read once, the start of file 0 twice (the seek)
        for threads in (1, 2, 5):
            got_bytes = C.c_int64(0)
            assert H.bvchost_bgzf_read_hash(arr, len(paths), threads, by_line, piece, 5, C.byref(got_bytes)) == want
            assert got_bytes.value == want_bytes.value


def test_block_inflate_equals_zlib_and_declines_bad_streams(H, tmp_path):
    """inflate.cpp (the decoder the BGZF readers use for whole blocks) against zlib: empty, tiny, random, periodic and
    pileup-text inputs; stored, fixed-code, dynamic-code, Huffman-only and RLE streams; several blocks per stream.  With too
    little room, truncated or corrupted input it must return -1 or a wrong byte count -- never the expected one, never touch
    memory outside its buffers (the sanitizer build, tools/sanitize_cpu.sh, runs this test too).  And a file written by the
    BGZF writer is read back without one block falling back to zlib."""
    import zlib
    H.bvchost_fast_inflate.restype = C.c_long
    H.bvchost_fast_inflate.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    H.bvchost_zlib_fallbacks.restype = C.c_long

    def inflate(comp, cap):
        out = C.create_string_buffer(max(cap, 1) + 64)
        guard = b"\xA5" * 64
        out[cap:cap + 64] = guard
        r = H.bvchost_fast_inflate(comp, len(comp), out, cap)
        assert out.raw[cap:cap + 64] == guard, "wrote past its output buffer"
        return r, out.raw[:max(r, 0)]

    rng = np.random.default_rng(77)
    toks = [("%d,%d,%d,%d,%d " % (rng.integers(4), rng.integers(20, 61), rng.integers(10, 41), rng.integers(1, 100), rng.integers(2)))
            if rng.random() < 0.1 else ". " for _ in range(30000)]
    inputs = [b"", b"a", b"ab" * 5, b". " * 32000, rng.bytes(70000), bytes(rng.choice(list(b"ACGT"), 65000).tolist()),
              "".join(toks).encode()[:65280]]
    for n in (1, 2, 3, 7, 8, 9, 15, 16, 17, 255, 256, 257, 258, 259, 1000, 65280):
        inputs.append(rng.bytes(n))
        for period in (1, 2, 3, 4, 5, 7, 8, 9):
            inputs.append((rng.bytes(period) * (n // period + 1))[:n])
    checked = 0
    for data in inputs:
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
                one = co.compress(data) + co.flush()
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
                step = max(1, len(data) // 3)
                many = b"".join(co.compress(data[i:i + step]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(data), step)) + co.flush()
                for comp in (one, many):
                    r, out = inflate(comp, len(data))
                    assert r == len(data) and out == data, (len(data), level, strategy)
                    checked += 1
                    if len(data) > 4:
                        assert inflate(comp, len(data) - 1)[0] != len(data)                   # too little room
                        assert inflate(comp[:len(comp) // 2], len(data))[0] != len(data) or comp[:len(comp) // 2] == comp
                        bad = bytearray(comp); bad[len(bad) // 2] ^= 0x55
                        r2, out2 = inflate(bytes(bad), len(data))                              # corrupted: whatever it says, no crash,
                        assert r2 <= len(data)                                                 # ... and never more than the room it was given
    assert checked > 2000
    # through the readers: nothing falls back to zlib
    H.bvchost_bgzf_write.restype = C.c_int
    H.bvchost_bgzf_write.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int]
    H.bvchost_bgzf_read_hash.restype = C.c_uint64
    H.bvchost_bgzf_read_hash.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                         C.POINTER(C.c_int64)]
    data = "".join(toks).encode() * 4
    f = str(tmp_path / "t.gz").encode()
    assert H.bvchost_bgzf_write(f, data, len(data), 65537, 6, 0) == 1
    before = H.bvchost_zlib_fallbacks()
    nbytes = C.c_int64(0)
    H.bvchost_bgzf_read_hash((C.c_char_p * 1)(f), 1, 2, 0, 4096, 0, C.byref(nbytes))
    assert nbytes.value >= len(data) and H.bvchost_zlib_fallbacks() == before


def test_bgzf_blocks_are_crc_checked_and_incomplete_code_sets_declined(H, tmp_path):
    """htslib compares the CRC32 of every BGZF block it inflates (the reference reads its temp batches through bgzf_getline,
    src/BaseVarC.cpp:406); so do the readers here.  (1) bgzf_crc32 -- PCLMULQDQ folding on x86-64 -- equals zlib's crc32 for every
    length around its 16- and 64-byte steps; (2) a block whose payload still inflates to ISIZE bytes but whose trailer CRC is wrong
    ends the read and is counted; (3) the block decoder declines the code sets zlib rejects as incomplete (a literal/length code
    that leaves patterns unused and is longer than one bit), so that such a stream reaches zlib, which refuses it."""
    import struct
    import zlib
    H.bvchost_crc32.restype = C.c_uint32
    H.bvchost_crc32.argtypes = [C.c_char_p, C.c_size_t]
    H.bvchost_crc_errors.restype = C.c_long
    rng = np.random.default_rng(5)
    for n in list(range(0, 200)) + [255, 256, 1000, 4095, 4096, 65279, 65280, 65536, 100001]:
        d = rng.bytes(n)
        assert H.bvchost_crc32(d, n) == (zlib.crc32(d) & 0xffffffff), n

    def block(payload, crc=None):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(payload) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(payload) & 0xffffffff if crc is None else crc, len(payload)))
    eof = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    H.bvchost_bgzf_read_hash.restype = C.c_uint64
    H.bvchost_bgzf_read_hash.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                         C.POINTER(C.c_int64)]
    a, b, c = (("line %d of block %s\n" % (i, t)) * 50 for i, t in enumerate("abc"))
    good = str(tmp_path / "good.gz").encode()
    open(good, "wb").write(block(a.encode()) + block(b.encode()) + block(c.encode()) + eof)
    bad = str(tmp_path / "bad.gz").encode()
    open(bad, "wb").write(block(a.encode()) + block(b.encode(), crc=(zlib.crc32(b.encode()) ^ 1) & 0xffffffff) + block(c.encode()) + eof)
    for threads in (0, 2):
        before = H.bvchost_crc_errors()
        n_good, n_bad = C.c_int64(0), C.c_int64(0)
        H.bvchost_bgzf_read_hash((C.c_char_p * 1)(good), 1, threads, 0, 4096, 0, C.byref(n_good))
        assert n_good.value == len(a) + len(b) + len(c) and H.bvchost_crc_errors() == before
        H.bvchost_bgzf_read_hash((C.c_char_p * 1)(bad), 1, threads, 0, 4096, 0, C.byref(n_bad))
        assert n_bad.value == len(a) and H.bvchost_crc_errors() == before + 1      # the read ends at the damaged block

    # (3) a dynamic block whose literal/length code is incomplete: lengths 2, 2, 2 for the symbols 'a', 'b' and 256 (Kraft sum 3/4);
    # the same block with a fourth two-bit symbol ('c') is complete and must decode
    class W:
        def __init__(self): self.v = 0; self.n = 0; self.out = bytearray()
        def put(self, val, bits):
            self.v |= val << self.n; self.n += bits
            while self.n >= 8: self.out.append(self.v & 0xff); self.v >>= 8; self.n -= 8
        def done(self):
            if self.n: self.out.append(self.v & 0xff)
            return bytes(self.out)

    def stream(complete):
        w = W()
        w.put(1, 1); w.put(2, 2)                            # last block, dynamic codes
        w.put(0, 5); w.put(0, 5); w.put(15, 4)              # HLIT 257, HDIST 1, HCLEN 19
        # code-length code (complete): symbol 0 -> '0', 2 -> '10', 18 -> '11' (codes are sent bit-reversed)
        order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
        cl = {0: 1, 2: 2, 18: 2}
        for s in order: w.put(cl.get(s, 0), 3)
        sym0 = lambda: w.put(0, 1)
        sym2 = lambda: w.put(0b01, 2)
        def sym18(rep): w.put(0b11, 2); w.put(rep - 11, 7)
        sym18(97)                                            # symbols 0..96: length 0
        sym2(); sym2()                                       # 'a' (97), 'b' (98): length 2
        if complete:
            sym2(); sym18(138); sym18(18)                    # 'c' too; 100..255: length 0
        else:
            sym18(138); sym18(19)                            # 99..255: length 0
        sym2()                                               # 256: length 2
        sym0()                                               # the one distance code: length 0
        if complete:                                         # canonical codes a = 00, b = 01, c = 10, 256 = 11
            w.put(0b00, 2); w.put(0b10, 2); w.put(0b11, 2)   # a b end-of-block
        else:                                                # a = 00, b = 01, 256 = 10; 11 is unused
            w.put(0b00, 2); w.put(0b10, 2); w.put(0b01, 2)
        return w.done()
    H.bvchost_fast_inflate.restype = C.c_long
    H.bvchost_fast_inflate.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    out = C.create_string_buffer(16)
    assert zlib.decompressobj(-15).decompress(stream(True)) == b"ab"
    assert H.bvchost_fast_inflate(stream(True), len(stream(True)), out, 2) == 2 and out.raw[:2] == b"ab"
    with pytest.raises(zlib.error, match="invalid literal/lengths set"):
        zlib.decompressobj(-15).decompress(stream(False))
    assert H.bvchost_fast_inflate(stream(False), len(stream(False)), out, 2) == -1


@pytest.mark.parametrize("thread", [1, 3, 8, 16])
def test_thread_windows_device_map_and_merge_order_give_position_ordered_output(H, tmp_path, thread):
    """configs[3] on the host program's side (reference: per-thread windows src/BaseVarC.cpp:399-403, sub-file merge
    :274-295), with the functions main.cpp itself calls: (1) the windows of T threads are disjoint, ascending, cover the position
    list and agree with the load phase's rule for which temp file a position goes to; (2) thread i works on device i mod G for G in
    {1, 2, 4, 8} devices present, `--gpus` only ever lowering G, every device taking floor or ceil of T / G threads; (3) sub-files
    written per thread and merged in thread order hold the positions in ascending order, and the sub-files are gone afterwards.
    No GPU is involved: which device a window is computed on cannot change its lines, because a site needs no other site's data."""
    import gzip
    H.bvchost_thread_window.restype = None
    H.bvchost_thread_window.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    H.bvchost_device_of_thread.restype = C.c_int32
    H.bvchost_device_of_thread.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    H.bvchost_merge_subfiles.restype = C.c_int32
    H.bvchost_merge_subfiles.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
    H.bvchost_bgzf_write.restype = C.c_int
    H.bvchost_bgzf_write.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int]

    def windows(psize):
        out = []
        for i in range(thread):
            lo, hi = C.c_int64(), C.c_int64()
            H.bvchost_thread_window(psize, thread, i, C.byref(lo), C.byref(hi))
            out.append((lo.value, hi.value))
        return out
    for psize in (0, 1, 5, 15, 16, 17, 31, 100, 1001, 78455):
        w = windows(psize)
        assert w[0][0] == 0 and w[-1][1] == psize and all(lo <= hi for lo, hi in w)
        assert all(w[i][1] == w[i + 1][0] for i in range(thread - 1)), (psize, w)
        window = psize % thread + psize // thread                      # the load phase's rule (bt_r, src/BaseVarC.cpp:497, 523)
        for i in (0, psize // 3, psize // 2, psize - 1):
            if 0 <= i < psize:
                owner = min(i // window, thread - 1)
                assert w[owner][0] <= i < w[owner][1], (psize, i, owner, w)
    for present in (1, 2, 4, 8):
        for opt in (0, 1, 2, 4, 8, 16):
            g = present if opt <= 0 or opt >= present else opt
            devs = [H.bvchost_device_of_thread(i, present, opt) for i in range(thread)]
            assert devs == [i % g for i in range(thread)]
            per = [devs.count(d) for d in range(g)]
            assert max(per) - min(per) <= 1 or thread < g
    # (3) T sub-files of a 78,455-position region (the test data's), merged
    psize, start = 78455, 41197700
    out = str(tmp_path / f"m{thread}")
    for i, (lo, hi) in enumerate(windows(psize)):
        for suffix in (".vcf.gz", ".cvg.gz"):
            lines = "".join(f"chr17\t{start + p}\t{suffix}\n" for p in range(lo, hi) if suffix == ".cvg.gz" or p % 97 == 0)
            if i == 0:
                lines = "#header\n" + lines
            data = lines.encode()
            assert H.bvchost_bgzf_write(f"{out}.{i}{suffix}".encode(), data, len(data), 1 << 20, 6, 0) == 1
    for suffix in (".vcf.gz", ".cvg.gz"):
        assert H.bvchost_merge_subfiles(out.encode(), suffix.encode(), thread) == 1
        raw = open(out + suffix, "rb").read()
        assert raw[-28:-12] == raw[-28:][:16]                            # ends with the BGZF EOF block
        text = gzip.decompress(raw).decode().splitlines()
        assert text[0] == "#header"
        pos = [int(l.split("\t")[1]) for l in text[1:]]
        assert pos == sorted(pos) and len(set(pos)) == len(pos)
        assert len(pos) == (psize if suffix == ".cvg.gz" else len(range(0, psize, 97)))
        assert not any(os.path.exists(f"{out}.{i}{suffix}") for i in range(thread))


def test_read_lines_hands_over_whole_lines_with_their_starts(H, tmp_path):
    """BgzfReader::read_lines (what the tiles the device parses are filled with): n lines per call, appended with their newlines,
    the start of every line noted -- across BGZF block ends, for lines longer than a block, empty lines, a last line without a
    newline (not returned), with and without read-ahead: the concatenation is the file up to its last newline."""
    H.bvchost_bgzf_write.restype = C.c_int
    H.bvchost_bgzf_write.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int]
    H.bvchost_bgzf_read_lines.restype = C.c_int64
    H.bvchost_bgzf_read_lines.argtypes = [C.c_char_p, C.c_int64, C.c_int32, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]
    rng = np.random.default_rng(12)
    cases = []
    lines = [". " * int(rng.integers(0, 4000)) for _ in range(300)] + ["", "", "x" * 200000, ""] + ["1,2,3,4,5 " * 10] * 50
    cases.append(("".join(l + "\n" for l in lines)).encode())
    cases.append(cases[0] + b"a last line without its newline")
    cases.append(b"")
    cases.append(b"\n")
    cases.append(b"no newline at all")
    for ci, data in enumerate(cases):
        f = str(tmp_path / f"rl{ci}.gz").encode()
        assert H.bvchost_bgzf_write(f, data, len(data), 65537, 6, 0) == 1
        want = data[:data.rfind(b"\n") + 1]
        for n_per_call in (1, 7, 1000):
            for threads in (0, 2):
                out = C.create_string_buffer(len(data) + 16)
                nl = C.c_int64(0)
                got = H.bvchost_bgzf_read_lines(f, n_per_call, threads, out, len(data) + 16, C.byref(nl))
                assert got == len(want) and out.raw[:got] == want and nl.value == want.count(b"\n"), (ci, n_per_call, threads, got)
