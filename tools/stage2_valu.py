"""profiles/stage2_valu.json: VALU instructions stage 2 executes per site-pass, from rocprofv3 --pmc SQ_INSTS_VALU passes over
tools/em_stage2.py (tools/_pmc4.sh), stamped with the hash of the em_items.hip they were taken on (bench.py: em_roofline).
usage: python tools/stage2_valu.py gpurun_out/<tag>      (expects sq1e4/ sq1e5/ sq1e6/ and their .log files in there)"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basevarc_amd.build import code_sha16

d = sys.argv[1]
out = {}
detail = {}
for name, depth in (("sq1e4", 10_000), ("sq1e5", 100_000), ("sq1e6", 1_000_000)):
    f = max(glob.glob(os.path.join(d, name, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == "SQ_INSTS_VALU" and re.search(r"region_kernel<32>", r["Kernel_Name"])]
    log = open(os.path.join(d, name + ".log")).read()
    m = re.search(r"engine 0: .* per call of (\d+) sites x N=(\d+) .*passes/site ([\d.]+)", log)
    sites, passes = int(m.group(1)), float(m.group(3))
    per_launch = sum(vals) / len(vals)
    out[str(depth)] = round(per_launch / (sites * passes), 2)
    detail[str(depth)] = {"SQ_INSTS_VALU_per_launch": per_launch, "sites": sites, "passes_per_site": passes, "launches": len(vals)}
j = {"em_items_sha16": code_sha16(os.path.join(ROOT, "basevarc_amd", "csrc", "em_items.hip")),
     "needed_per_pass": 15.4,
     "needed_note": "issue slots an E+M pass of one fit needs with every lane group of its wavefront busy: a pass of eight four-allele fits "
                    "(two lanes x 16 classes per allele) is 150 slots = 18.75 per fit, of eight two-allele fits (four lanes x 8 classes) half "
                    "that; a site's passes are 64 % four-allele and 36 % two-allele on the synthetic workload: 0.64 x 18.75 + 0.36 x 9.4",
     "executed_per_pass": out, "detail": detail,
     "summary": f"rocprofv3 --pmc SQ_INSTS_VALU over tools/em_stage2.py ({d}); region_kernel<32> only"}
json.dump(j, open(os.path.join(ROOT, "profiles", "stage2_valu.json"), "w"), indent=1)
print(json.dumps(j["executed_per_pass"]), j["em_items_sha16"])
