"""Reads a rocprofv3 kernel trace (csv) and prints, per kernel name, count / mean duration, and how much of each region kernel's
lifetime another launch of the same kernel was running too."""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
ev = [(r["Kernel_Name"].split("(")[0].replace("bvc::(anonymous namespace)::", "").replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
by = collections.defaultdict(list)
for n, s, e in ev: by[n].append((s, e))
for n, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:12]:
    d = [e - s for s, e in v]
    # overlap with other launches of the same name
    ov = 0
    vs = sorted(v)
    for i, (s, e) in enumerate(vs):
        for s2, e2 in vs[i + 1:]:
            if s2 >= e: break
            ov += min(e, e2) - s2
    print(f"{n[:60]:60s} n={len(v):4d} mean={sum(d) / len(d) / 1e3:8.1f} us  self-overlap={ov / max(1, sum(d)):.2f}")
t0 = min(s for _, s, _ in ev[len(ev) // 2:]); 
last = sorted(ev, key=lambda x: x[1])[-24:]
for n, s, e in last: print(f"  {n[:40]:40s} start {(s - last[0][1]) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us")
