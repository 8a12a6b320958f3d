"""The one-line JSON contract of bench.py (driver-facing), on a reduced workload so that it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "1",
                        "--samples", "200000", "--total-sites", "8000", "--tile-sites", "2000", "--cpu-sites", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "none" and d["vs_baseline"] is None           # one GPU: nothing is scaled and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # (outside the timed region) the same steps with every subset of every level run; at BASELINE size the two are equal (the histogram
    # pass bounds the call), on this small workload stage 2 shows
    assert 0 < d["value_with_every_subset_run"] and (d.get("diagnostic_build") or d["value_with_every_subset_run"] <= 1.05 * d["value"])
    # a step is one pass over the whole resident workload (8000 sites here, in calls of 2000)
    assert d["config"]["sites_per_step"] == 8000 and d["config"]["sites_per_call"] == 2000
    assert d["value"] == pytest.approx(8000 / (d["ms_per_step"] / 1e3), rel=1e-3)
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s"
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3) and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["cpu_model"] and cb["single_thread"]["cores"] == 1 and cb["single_thread"]["value"] > 0
    assert cb["gpu_check_same_sites"]["mismatches"] == 0 and cb["gpu_check_hist_form"]["mismatches"] == 0
    # the parallelism the CPU leg GOT (not just what it was given), and the rates BASELINE.md section 4 asks for
    assert 0 < cb["effective_parallelism"] <= cb["cores"] * 1.2 and "cgroup_cpu_quota" in cb
    for k in ("faithful_port_N1e4", "histogram_form"):
        assert cb[k]["one_thread"] > 0 and cb[k]["all_threads"] > 0 and cb[k]["threads"] >= 1, k
    # the streaming read (best of fifteen passes over three tiles, before and after the warm-up) is a ceiling: the histogram kernel
    # does not read faster, to within the 1 % by which two measurements of the same read differ
    assert rf["frac_of_empirical"] <= 1.01 and rf["empirical_stream_read_median_GBs"] <= rf["empirical_stream_read_GBs"]
    assert "traffic_note" in rf
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["sites"] == 8000 and d["per_rank"][0]["calls_per_step"] == 4
    # every leg's rate and roofline fraction once more as the LAST key of the line (a reader that keeps the tail of stdout sees them)
    assert list(d.keys())[-1] == "legs_summary" and len(json.dumps(d["legs_summary"])) < 800
    assert set(d["legs_summary"]) == set(d["legs"])
    for name, (value, frac) in d["legs_summary"].items():
        assert value > 0 and 0 < frac < 1, name
        if name != "host_pointer_one_byte":
            assert value == pytest.approx(d["legs"][name]["value"], rel=1e-3) and frac == pytest.approx(d["legs"][name]["roofline"]["frac"], rel=1e-3)
    # the other single-GPU configurations ride along as sub-records with their own roofline
    legs = d["legs"]
    assert legs["csr_groups5_coverage10pct"]["overall_records_identical_to_bvc_lrt_csr"] is True
    for name in ("config1_1e4x1e4", "config4_groups5_interleaved", "config4_groups5_ordered", "csr_coverage10pct", "csr_groups5_coverage10pct"):
        assert legs[name]["value"] > 0 and 0 < legs[name]["roofline"]["frac"] < 1, name
    # the additive packed layout (one byte per sample) rides along too, and must give the two-byte path's records
    pk = legs["packed_1_byte_per_sample"]
    assert pk["value"] > 0 and pk["records_identical_to_two_byte_path"] is True and pk["roofline"]["bytes_per_sample"] == 1
    for name in ("packed_groups5_interleaved", "packed_groups5_ordered"):
        assert legs[name]["value"] > 0 and legs[name]["records_identical_to_two_byte_path"] is True, name
    # the producer of the path's input on the device, tile by tile as the host program calls it, and its inflate kernel alone
    pr = legs["producer_bgzf_1e5_coverage10pct"]
    assert pr["unit"] == "positions/s" and pr["value"] > 0 and pr["called_positions"] > 0 and pr["entries_parsed"] > 0
    assert pr["roofline"]["text_GBs"] > 0 and pr["text_GBs"] > 0 and 0 < pr["roofline"]["frac"] < 1
    assert pr["cpu_path"]["cores"] == 1 and pr["cpu_path"]["value"] > 0                          # the same tiles on one CPU core
    if not d.get("diagnostic_build"):                            # (rates against rates: not under the poisoned, bound-checked build)
        assert pr["roofline"]["text_GBs"] > pr["text_GBs"] and pr["cpu_path"]["value"] < pr["value"]
    hp = legs["host_pointer_one_byte"]
    assert hp["bound"] == "pcie" and 0 < hp["ragged_one_byte"]["frac"] < 1 and 0 < hp["dense_one_byte"]["frac"] < 1
    assert hp["ragged_one_byte"]["records_identical_to_device_pointer_call"] is True
    assert legs["config1_1e4x1e4"]["roofline"]["bound"] == "fp64_valu_issue"
    # the EM-bound legs are also timed with every subset of every level run (em_prune = 0): what the reference defines of a record
    # is the same bytes, the engine runs fewer passes than the reference counts, and is not slower for it
    for name in ("config1_1e4x1e4", "csr_coverage10pct"):
        leg = legs[name]
        assert leg["records_identical_with_every_subset_run_except_the_run_counts"] is True, name
        assert leg["roofline"]["em_passes_per_site"] < leg["em_passes_per_site_of_the_reference"], name
        assert leg["value_with_every_subset_run"] > 0, name
        if not d.get("diagnostic_build"):                        # (a rate against a rate: not under the poisoned, bound-checked build)
            assert leg["value"] > 0.95 * leg["value_with_every_subset_run"], name
    assert legs["csr_coverage10pct"]["roofline"]["bound"] == "fp64_valu_issue" and legs["csr_coverage10pct"]["hist_roofline"]["bound"] == "hbm"
    # the ragged histogram pass is reported twice: underneath stage 2 and with the chip to itself (faster alone)
    hr = legs["csr_coverage10pct"]["hist_roofline"]
    assert 0 < hr["frac"] <= hr["alone"]["frac"] * 1.05 and hr["alone"]["launches_timed"] >= 8


def test_bench_under_the_drivers_multi_gpu_launcher_two_ranks_sharing_the_card():
    """The driver's N > 1 command line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...) with N = 2 on this one-GPU box: BVC_BENCH_BACKEND=gloo lets the
    ranks share device 0 (a rehearsal of the launch, rendezvous, sharding, barrier and max-over-ranks path -- not a
    scaling measurement).  configs[3] as written: ONE workload split by site (strong scaling), rank 0 prints the line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BVC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--samples", "200000", "--total-sites", "8000",
                        "--tile-sites", "2000", "--cpu-sites", "0", "--no-legs"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong"
    # the whole job's 8000 sites per step, 4000 on each rank
    assert d["config"]["sites_per_step"] == 8000
    assert d["value"] == pytest.approx(8000 / (d["ms_per_step"] / 1e3), rel=1e-3)
    assert "gloo" in d["config"]["sharding"] and "strong" in d["config"]["sharding"]
    # every rank's own clock and kernel times; each rank cuts its 4000 sites into 8 equal calls (no short tail call)
    pr = d["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1] and all(p["sites"] == 4000 and p["calls_per_step"] == 8 for p in pr)
    assert all(p["ms_per_step"] > 0 and p["hist_ms_per_call"] > 0 for p in pr) and d["config"]["sites_per_call"] == 500
