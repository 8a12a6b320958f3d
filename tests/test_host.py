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
