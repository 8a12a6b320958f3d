#!/usr/bin/env python3
"""How long the EM fits of a site run, fit by fit, and what that means for stage 2's lock-step slots.

CHECKER-SIDE analysis, not a test and not product code: it compiles a COPY of the CPU restatement (oracle/basetype_oracle.c) in a
temporary directory with one line added to update_f -- print (site, n, k, the allele the subset leaves out, depths, passes) -- and
runs it on the bench's synthetic sites (histogram form: the same fits, class by class).  Then, from those pass counts alone:
  * what the item engine runs per region of eight sites (the subset without the deepest allele is not run: em_prune), list by list;
  * the passes its slots of eight fits last in lock step (a slot lasts as long as its slowest fit), against the sum of the fits;
  * the same with the region's fits ordered by their pass counts (an ORACLE order: nothing known before the fit predicts them) and
    with fit-granular refill inside the team (16 places pull the next fit as they finish; the level still ends with its last fit).
usage: python tests/analysis/fit_passes.py [n_sites=800] [n_samples=100000] [coverage=1.0]"""
import os
import subprocess
import sys
import tempfile
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.join(HERE, "..", "..", "oracle")
n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 800
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
cov = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0

src = open(os.path.join(ORACLE, "basetype_oracle.c")).read()
mark = "        fs->passes[fs->n_fit] += sm->n_passes;"
assert mark in src
src = src.replace(mark, mark + r'''
        if (g_trace) {
            int drop = -1, tt, bb;
            for (bb = 0; bb < n; ++bb) { int in = 0; for (tt = 0; tt < k; ++tt) if (fs->comb[c][tt] == bases[bb]) in = 1; if (!in) { drop = bases[bb]; break; } }
            fprintf(g_trace, "%ld %d %d %d %d %d %d %d %d %d\n", g_site, n, k, c, drop, sm->depth[0], sm->depth[1], sm->depth[2], sm->depth[3], fs->passes[fs->n_fit]);
        }''', 1)
src = src.replace("typedef struct fit_set {", "#include <stdio.h>\nstatic FILE *g_trace; static long g_site;\ntypedef struct fit_set {", 1)
src += r'''
int main(int argc, char **argv)
{
    long n_sites = atol(argv[1]), n = atol(argv[2]);
    double cov = atof(argv[3]);
    int8_t *b = malloc(n), *q = malloc(n), ref;
    orc_result r;
    g_trace = stdout;
    for (g_site = 0; g_site < n_sites; ++g_site) {
        double min_af = 100.0 / n; if (min_af > 0.001) min_af = 0.001;
        orc_synth_site(1, g_site, n, (uint32_t)(cov * 65536 + 0.5), b, q, &ref);
        dense_site_subset(n, b, q, NULL, -1, ref, min_af, NULL, 0, ORC_MODE_HIST, &r);
    }
    return 0;
}
'''
with tempfile.TemporaryDirectory() as d:
    for f in os.listdir(ORACLE):
        if f.endswith((".h", ".inc")):
            open(os.path.join(d, f), "w").write(open(os.path.join(ORACLE, f)).read())
    open(os.path.join(d, "trace.c"), "w").write(src)
    subprocess.run(["gcc", "-O2", "-w", "-o", os.path.join(d, "trace"), os.path.join(d, "trace.c"), "-lm"], check=True)
    out = subprocess.run([os.path.join(d, "trace"), str(n_sites), str(n_samples), str(cov)], check=True, capture_output=True, text=True).stdout
rows = [tuple(map(int, l.split())) for l in out.splitlines() if len(l.split()) == 10]
sites = defaultdict(list)
for r in rows:
    sites[r[0]].append(r)


def pct(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(p / 100.0 * len(v)))]


print(f"{len(sites)} sites of {n_samples} samples at coverage {cov:g}: {len(rows)} fits of the reference, {sum(r[9] for r in rows)} passes")
by = defaultdict(list)
for r in rows:
    by[(r[1], r[2])].append(r[9])
for key, v in sorted(by.items()):
    print(f"  fits of {key[1]} out of {key[0]} candidates: {len(v):6d}  mean {sum(v) / len(v):6.1f}  p10 {pct(v, 10):3d}  p50 {pct(v, 50):3d}  p90 {pct(v, 90):3d}  max {max(v)}")
# what the item engine runs: per site and level the fits of the model that level is at; the (n-1)-subset without the deepest allele is
# left out (its log-likelihood bound proves it cannot be the level's first minimum on this workload); one-allele models are closed-form
levels = defaultdict(lambda: defaultdict(list))          # region -> (round, rows of the item) -> passes
for s, fits in sites.items():
    region = s // 8
    for r in fits:
        n, k, drop, dep, passes = r[1], r[2], r[4], r[5:9], r[9]
        if k <= 1:
            continue
        if k == n - 1 and drop >= 0 and dep[drop] == max(dep):
            continue
        rows_of_item = 4 if k >= 3 else 2
        # level order: the full model and the 3-subsets of four candidates in round 0, 2-subsets of three in round 1, ...
        rnd = {(4, 4): 0, (4, 3): 0, (3, 3): 0, (3, 2): 1, (2, 2): 0}.get((n, k), 1)
        levels[region][(rnd, rows_of_item)].append(passes)
lock = work = ordered = refill = 0
for region, lists in levels.items():
    for key, p in lists.items():
        work += sum(p)
        slots = [p[i:i + 8] for i in range(0, len(p), 8)]
        # two wavefronts of a team take slots from a counter: the level lasts as long as the busier wavefront
        w = [0, 0]
        for sl in slots:
            w[w.index(min(w))] += max(sl)
        lock += 2 * max(w)                                # wavefront-passes the level occupies (both wavefronts are held to its end)
        q = sorted(p, reverse=True)
        w = [0, 0]
        for sl in [q[i:i + 8] for i in range(0, len(q), 8)]:
            w[w.index(min(w))] += max(sl)
        ordered += 2 * max(w)
        places = [0] * 16
        for x in p:                                       # in list order, the next fit to the place that is free first
            places[places.index(min(places))] += x
        refill += 2 * max(places)
per_fit = work / 8.0
print(f"  the engine's fits: {work} passes = {per_fit:.0f} wavefront-passes of eight fits each")
print(f"  lock-step slots as built:                         {lock} wavefront-passes ({per_fit / lock:.2f} of them useful)")
print(f"  ... with the level's fits ordered by pass count:  {ordered} ({per_fit / ordered:.2f}; an oracle order)")
print(f"  ... with fit-granular refill inside the team:     {refill} ({per_fit / refill:.2f})")
