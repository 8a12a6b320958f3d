"""profiles/stage2_valu.json: VALU instructions stage 2 executes per site-pass, from rocprofv3 --pmc SQ_INSTS_VALU passes over
tools/em_stage2.py (tools/_pmc4.sh), stamped with the hash of the em_items.hip they were taken on (bench.py: em_roofline).
usage: python tools/stage2_valu.py gpurun_out/<tag>      (expects sq1e4/ sq1e5/ sq1e6/ and their .log files in there)"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basevarc_amd.build import code_sha16

d = sys.argv[1]
out = {}
out_ref = {}
all_run = {}
detail = {}
for name, depth in (("sq1e4", 10_000), ("sq1e5", 100_000), ("sq1e6", 1_000_000)):
    f = max(glob.glob(os.path.join(d, name, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    rows = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == "SQ_INSTS_VALU" and re.search(r"region_kernel<32>", r["Kernel_Name"])]
    rows.sort()
    # tools/em_stage2.py runs the item engine twice: first with em_prune = 0 (every subset run), then as the library ships it
    half = len(rows) // 2
    vals_all, vals = [v for _, v in rows[:half]], [v for _, v in rows[half:]]
    log = open(os.path.join(d, name + ".log")).read()
    m = re.search(r"engine 0: .* per call of (\d+) sites x N=(\d+) .*passes/site ([\d.]+)", log)
    ma = re.search(r"engine 0a: .* per call of (\d+) sites x N=(\d+) .*passes/site ([\d.]+)", log)
    sites, passes, ref_passes = int(m.group(1)), float(m.group(3)), float(ma.group(3))
    per_launch, per_launch_all = sum(vals) / len(vals), sum(vals_all) / len(vals_all)
    out[str(depth)] = round(per_launch / (sites * passes), 2)
    out_ref[str(depth)] = round(per_launch / (sites * ref_passes), 2)
    all_run[str(depth)] = round(per_launch_all / (sites * ref_passes), 2)
    detail[str(depth)] = {"SQ_INSTS_VALU_per_launch": per_launch, "SQ_INSTS_VALU_per_launch_every_subset_run": per_launch_all, "sites": sites,
                          "passes_run_per_site": passes, "passes_of_the_reference_per_site": ref_passes, "launches": len(vals)}
j = {"em_items_sha16": code_sha16(os.path.join(ROOT, "basevarc_amd", "csrc", "em_items.hip")),
     "needed_per_pass": 16.4,
     "needed_note": "issue slots an E+M pass of one fit needs with every lane group of its wavefront busy: a pass of eight four-allele fits "
                    "(two lanes x 16 classes per allele) is 150 slots = 18.75 per fit, of eight two-allele fits (four lanes x 8 classes) half "
                    "that; the passes the engine RUNS are 75 % four-allele and 25 % two-allele on the synthetic workload (N = 1e4; the subsets "
                    "it does not run were 36 % / 58 % of the two kinds): 0.75 x 18.75 + 0.25 x 9.4",
     "executed_per_pass": out,
     "executed_per_pass_note": "VALU instructions of region_kernel<32> per E+M pass the engine RAN (record field n_passes)",
     "executed_per_reference_pass": out_ref,
     "executed_per_reference_pass_note": "the same instructions over the passes the REFERENCE runs on those sites (em_prune = 0 counts them): "
                                         "the unit of rounds 3-4's figures (26.8 / 22.7 / 18.9 before the last-resort subsets were skipped)",
     "every_subset_run_per_reference_pass": all_run,
     "detail": detail,
     "summary": f"rocprofv3 --pmc SQ_INSTS_VALU over tools/em_stage2.py ({d}); region_kernel<32> only"}
json.dump(j, open(os.path.join(ROOT, "profiles", "stage2_valu.json"), "w"), indent=1)
print(json.dumps(j["executed_per_pass"]), json.dumps(out_ref), json.dumps(all_run), j["em_items_sha16"])
