"""GPU end-to-end test of the host program: `BaseVarC basetype` (basevarc_amd/BaseVarC, C++ over libbvc) on the
reference's own test data, with the reference's own command line (test/test.sh:3), against the Python pipeline
(tests/hostref.py: pileup restatement + CPU oracle + emit restatement).  BASELINE configs[0] at file level."""
import gzip
import subprocess

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("thread,grouped,tmp_format", [(1, False, "text"), (4, False, "text"), (2, True, "text"),
                                                       (3, False, "bin"), (2, True, "bin"), (2, False, "raw"),
                                                       (2, True, "two_byte_tiles")])
def test_basetype_command_on_reference_test_data(tmp_path, thread, grouped, tmp_format):
    from basevarc_amd import build as b
    from tests import hostref
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    lst = hostref.write_bam_list(str(tmp_path / "bam.list"))
    out = str(tmp_path / "test.out")
    pipe = hostref.Pipeline(mapq=20, batch=10, thread=thread)
    cmd = [exe, "basetype", "--rerun", "-q", "20", "-t", str(thread), "-b", "10", "-i", lst, "-s", hostref.REGION,
           "-r", fa, "-o", out]
    env = None
    if tmp_format == "two_byte_tiles":                            # --group tiles go to libbvc packed (one byte per sample)
        import os                                                 # unless a quality does not fit; this forces the two-byte tiles
        env = dict(os.environ, BVC_HOST_TWO_BYTE_TILES="1", BVC_HOST_DEVICE_PARSE="0")      # (dense tiles are the CPU parser's path)
    elif tmp_format != "text":                                    # additive: binary temp batches, same outputs
        cmd += ["--tmp-format", tmp_format]
    group_of = None
    if grouped:                                                   # --group <SampleID Group>; 7 samples left ungrouped
        group_of = {n: ["EAS", "AFR", "EUR"][i % 3] for i, n in enumerate(pipe.names) if i % 14 != 5}
        gf = tmp_path / "groups.txt"
        gf.write_text("".join(f"{k} {v}\n" for k, v in group_of.items()))
        cmd += ["-g", str(gf)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "basetype done" in r.stdout
    vcf_body, cvg_body = pipe.outputs(group_of)
    vcf_head, cvg_head = hostref.headers(fa, pipe.names, sorted(set(group_of.values())) if grouped else ())
    got_vcf = gzip.decompress(open(out + ".vcf.gz", "rb").read()).decode()
    got_cvg = gzip.decompress(open(out + ".cvg.gz", "rb").read()).decode()
    assert got_cvg.count("\n") == 66614 + 3                      # every covered position + 3 header lines
    assert got_cvg == cvg_head + cvg_body
    want_vcf = vcf_head + vcf_body
    assert got_vcf.split("\n")[:20] == want_vcf.split("\n")[:20]
    gl, wl = got_vcf.split("\n"), want_vcf.split("\n")
    assert len(gl) == len(wl) and sum(1 for l in gl if l and l[0] != "#") == 76
    for g, w in zip(gl, wl):
        if g == w:
            continue
        # device and host libm may differ in the last printed digit of a float field: compare numerically
        gf, wf = g.split("\t"), w.split("\t")
        assert gf[:5] == wf[:5] and gf[6] == wf[6] and gf[8:] == wf[8:], (g[:200], w[:200])
        assert abs(float(gf[5]) - float(wf[5])) <= 0.011
        for a, c in zip(gf[7].split(";"), wf[7].split(";")):
            ka, va = a.split("="); kc, vc = c.split("=")
            assert ka == kc
            for x, y in zip(va.split(","), vc.split(",")):
                assert x == y or abs(float(x) - float(y)) <= 2e-6 * max(1.0, abs(float(y))) + 1.1e-3, (ka, x, y)
    # temp files and directories are gone (no --keep_tmp), sub-files merged
    import os
    assert not os.path.exists(out + ".tmp.thread.0") and not os.path.exists(out + ".0.vcf.gz")


def test_keep_tmp_leaves_the_batches_and_rerun_reuses_them(tmp_path):
    """--keep_tmp (src/BaseVarC.cpp:458-462): the temp directories and batch files survive the run; a second run with
    --rerun finds them complete, extracts nothing and writes the same outputs."""
    import os
    from basevarc_amd import build as b
    from tests import hostref
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    lst = hostref.write_bam_list(str(tmp_path / "bam.list"))
    out = str(tmp_path / "test.out")
    cmd = [exe, "basetype", "-q", "20", "-t", "2", "-b", "25", "-i", lst, "-s", hostref.REGION, "-r", fa, "-o", out, "--keep_tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    files = [f"{out}.tmp.thread.{t}/batch.{ib}" for t in range(2) for ib in range(4)]
    assert all(os.path.exists(f) for f in files)
    first = {k: open(out + k, "rb").read() for k in (".vcf.gz", ".cvg.gz")}
    r = subprocess.run(cmd + ["--rerun"], capture_output=True, text=True)
    assert r.returncode == 0 and "begin to extract reads from bam" not in r.stderr
    assert all(gzip.decompress(open(out + k, "rb").read()) == gzip.decompress(v) for k, v in first.items())
    r = subprocess.run(cmd[:-1] + ["--rerun"], capture_output=True, text=True)     # without --keep_tmp: cleaned up
    assert r.returncode == 0 and not any(os.path.exists(f) for f in files) and not os.path.exists(out + ".tmp.thread.0")


def test_a_failing_worker_thread_ends_the_program_with_its_message(tmp_path):
    """ADVICE (main.cpp): an error inside the phase-2 worker threads (here: more than 32 population groups, thrown by
    every bt_s thread) must come out as exit code 1 with the message on stderr -- not as std::terminate from unwinding
    past joinable threads -- and must leave the temp batches in place for --rerun."""
    import os
    from basevarc_amd import build as b
    from tests import hostref
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    lst = hostref.write_bam_list(str(tmp_path / "bam.list"))
    out = str(tmp_path / "test.out")
    names = hostref.Pipeline(mapq=20, batch=25, thread=1).names
    gf = tmp_path / "groups.txt"
    gf.write_text("".join(f"{n} G{i % 40:02d}\n" for i, n in enumerate(names)))          # 40 groups > BVC_MAX_GROUPS
    r = subprocess.run([exe, "basetype", "-q", "20", "-t", "4", "-b", "25", "-i", lst, "-s", hostref.REGION, "-r", fa,
                        "-o", out, "-g", str(gf)], capture_output=True, text=True)
    assert r.returncode == 1, (r.returncode, r.stderr[-500:])
    assert "more than 32 population groups" in r.stderr
    assert os.path.exists(f"{out}.tmp.thread.0/batch.0") and not os.path.exists(out + ".0.vcf.gz")


def test_more_gpus_requested_than_present_degrades_to_the_devices_there(tmp_path):
    """`--gpus 2 -t 4` on a box with one MI355X: threads map to device i mod min(--gpus, devices present) =
    device 0 (host/main.cpp), the run succeeds and writes what a plain run writes."""
    from basevarc_amd import build as b
    from tests import hostref
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    lst = hostref.write_bam_list(str(tmp_path / "bam.list"))
    outs = []
    for extra, name in ((["--gpus", "2"], "g2"), ([], "plain")):
        out = str(tmp_path / name)
        r = subprocess.run([exe, "basetype", "-q", "20", "-t", "4", "-b", "25", "-i", lst, "-s", hostref.REGION, "-r", fa,
                            "-o", out] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([gzip.decompress(open(out + k, "rb").read()) for k in (".vcf.gz", ".cvg.gz")])
    assert outs[0] == outs[1]


def _run(exe, out, lst, fa, extra=(), env=None, thread=2, batch=25):
    from tests import hostref
    r = subprocess.run([exe, "basetype", "-q", "20", "-t", str(thread), "-b", str(batch), "-i", lst, "-s", hostref.REGION,
                        "-r", fa, "-o", out] + list(extra), capture_output=True, text=True, env=env)
    return r


@pytest.mark.parametrize("grouped", [False, True])
def test_tiles_with_qualities_of_63_and_more_fall_back_to_two_bytes_by_themselves(tmp_path, grouped):
    """The host program sends a tile to libbvc at one byte per observation (bvc_lrt_csr_packed, or
    bvc_lrt_dense_groups_packed with --group) as long as every base quality of the tile is below 63, and as two bytes
    otherwise -- tile by tile, without being told.  The test data stop at quality 39 (0.5 % of the observations), so a test
    hook shifts every quality by 24 as it is read (Q39 becomes 63): with tiles of 32 positions some fit, some do not.  The outputs must be those of a run that is forced onto
    two-byte tiles throughout (BVC_HOST_TWO_BYTE_TILES=1), byte for byte, and the automatic run must have used both forms."""
    import os
    import re
    from basevarc_amd import build as b
    from tests import hostref
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    lst = hostref.write_bam_list(str(tmp_path / "bam.list"))
    extra = ["--tile", "32"]
    if grouped:
        names = hostref.Pipeline(mapq=20, batch=25, thread=1).names
        gf = tmp_path / "groups.txt"
        gf.write_text("".join(f"{n} {['EAS', 'AFR', 'EUR'][i % 3]}\n" for i, n in enumerate(names) if i % 14 != 5))
        extra += ["-g", str(gf)]
    outs = {}
    for name, env_extra in (("auto", {}), ("two", {"BVC_HOST_TWO_BYTE_TILES": "1"})):
        out = str(tmp_path / name)
        env = dict(os.environ, BVC_HOST_QUAL_SHIFT="24", BVC_HOST_PROFILE="1", **env_extra)
        r = _run(exe, out, lst, fa, extra, env)
        assert r.returncode == 0, r.stderr[-2000:]
        forms = [(int(a), int(c)) for a, c in re.findall(r"one-byte tiles (\d+), on two-byte tiles (\d+)", r.stderr)]
        outs[name] = ([gzip.decompress(open(out + k, "rb").read()) for k in (".vcf.gz", ".cvg.gz")], forms)
    assert outs["auto"][0] == outs["two"][0]
    one = sum(a for a, _ in outs["auto"][1]); two = sum(c for _, c in outs["auto"][1])
    assert one > 0 and two > 0, outs["auto"][1]
    assert sum(a for a, _ in outs["two"][1]) == 0
    assert outs["auto"][0][0].count(b"\n") > 30                 # a VCF with records, not an empty run


def test_a_temp_batch_cut_short_is_an_error_in_both_forms(tmp_path):
    """A temp batch holds one record per position of its thread's window.  One that ends early (a valid BGZF file with
    its EOF block, but fewer records) used to leave the remaining positions without that batch's samples, silently; it
    is an error now, in the text and in the binary form (exit code 1, message on stderr)."""
    import os
    from basevarc_amd import build as b
    from tests import hostref
    from tools.host_bench import _bgzf_write
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    lst = hostref.write_bam_list(str(tmp_path / "bam.list"))
    for fmt in ("text", "bin"):
        out = str(tmp_path / f"cut_{fmt}")
        extra = ["--keep_tmp"] + ([] if fmt == "text" else ["--tmp-format", "bin"])
        r = _run(exe, out, lst, fa, extra + ["--load"])
        assert r.returncode == 0, r.stderr[-1000:]
        victim = f"{out}.tmp.thread.1/batch.2"
        raw = gzip.decompress(open(victim, "rb").read())
        cut = raw[:raw.rfind(b"\n", 0, len(raw) * 2 // 3) + 1] if fmt == "text" else raw[:16 + (len(raw) - 16) // 2]
        tmp = str(tmp_path / "cut.raw")
        open(tmp, "wb").write(cut)
        _bgzf_write(tmp, victim)
        r = _run(exe, out, lst, fa, extra + ["--rerun"])
        assert r.returncode == 1, (fmt, r.returncode, r.stderr[-500:])
        assert "truncated temp batch" in r.stderr or "malformed temp batch" in r.stderr, r.stderr[-500:]
        assert os.path.exists(victim)


@pytest.mark.parametrize("grouped", [False, True])
def test_device_parsed_tiles_write_what_cpu_parsed_tiles_write(tmp_path, grouped):
    """The text batches of the reference's test data through the feeds of the compute phase: the BGZF blocks inflated AND parsed on the
    device (bvc_pileup_begin_bgzf, the default), tiles of text inflated on the CPU and parsed on the device (BVC_HOST_DEVICE_INFLATE=0,
    bvc_pileup_begin; with --group both use the ragged group call), and the CPU parser (BVC_HOST_DEVICE_PARSE=0; with --group dense tiles).  Same VCF and CVG, byte for byte, whatever the tile size (tiles of 1, 37
    and the default), and the device run really parsed on the device.  One batch file then gets a line the reference's writer never
    produces (two spaces in a row, which strtok_r skips): that tile -- and only that tile -- goes through the CPU parser, and the
    outputs are those of the CPU feed on the same files."""
    import os
    import re
    from basevarc_amd import build as b
    from tests import hostref
    from tools.host_bench import _bgzf_write
    exe, _ = b.build_host()
    fa = hostref.write_fasta(str(tmp_path / "chr17.fa"))
    lst = hostref.write_bam_list(str(tmp_path / "bam.list"))
    extra = []
    if grouped:
        names = hostref.Pipeline(mapq=20, batch=25, thread=1).names
        gf = tmp_path / "groups.txt"
        gf.write_text("".join(f"{n} {['EAS', 'AFR', 'EUR'][i % 3]}\n" for i, n in enumerate(names) if i % 14 != 5))
        extra += ["-g", str(gf)]
    base = str(tmp_path / "base")
    r = _run(exe, base, lst, fa, extra + ["--keep_tmp"], dict(os.environ, BVC_HOST_DEVICE_PARSE="0", BVC_HOST_PROFILE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert sum(int(x) for x in re.findall(r"parsed on the device (\d+)", r.stderr)) == 0
    want = [gzip.decompress(open(base + k, "rb").read()) for k in (".vcf.gz", ".cvg.gz")]
    assert want[0].count(b"\n") > 30

    def rerun(tile, env_extra=None):
        env = dict(os.environ, BVC_HOST_PROFILE="1", **(env_extra or {}))
        rr = _run(exe, base, lst, fa, extra + ["--keep_tmp", "--rerun"] + (["--tile", str(tile)] if tile else []), env)
        assert rr.returncode == 0, rr.stderr[-2000:]
        dev = sum(int(x) for x in re.findall(r"parsed on the device (\d+)", rr.stderr))
        cpu = sum(int(x) for x in re.findall(r"handed back to the CPU parser (\d+)", rr.stderr))
        return [gzip.decompress(open(base + k, "rb").read()) for k in (".vcf.gz", ".cvg.gz")], dev, cpu
    # the device feed in both its forms: blocks inflated on the device too (the default), and inflated on the CPU (tiles of text)
    for tile in (0, 37, 1):
        for inflate in ("1", "0"):
            got, dev, cpu = rerun(tile, {"BVC_HOST_DEVICE_INFLATE": inflate})
            assert got == want, (tile, inflate)
            assert dev > 0 and cpu == 0, (tile, inflate, dev, cpu)
    got, dev, cpu = rerun(0, {"BVC_HOST_TILE_MB": "1"})              # small tiles by bytes: many calls, batches at different paces
    assert got == want and dev > 0 and cpu == 0
    # a line with a run of two spaces in batch 1 of thread 0: the same columns to strtok_r, not a line the device parses
    victim = f"{base}.tmp.thread.0/batch.1"
    raw = gzip.decompress(open(victim, "rb").read()).split(b"\n")
    k = next(i for i, l in enumerate(raw) if i > 2000 and l.count(b",") >= 8)
    raw[k] = raw[k].replace(b" ", b"  ", 1)
    tmp = str(tmp_path / "victim.raw")
    open(tmp, "wb").write(b"\n".join(raw))
    _bgzf_write(tmp, victim)
    for inflate in ("1", "0"):
        got, dev, cpu = rerun(64, {"BVC_HOST_DEVICE_INFLATE": inflate})
        assert got == want, inflate
        assert cpu == 1 and dev > 10, (inflate, dev, cpu)
