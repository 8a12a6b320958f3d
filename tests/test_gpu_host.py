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
        env = dict(os.environ, BVC_HOST_TWO_BYTE_TILES="1")
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
