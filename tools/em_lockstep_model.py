"""Model of the item engine's lockstep loss on synthetic sites (CPU only; uses the numpy restatement for per-fit pass counts).
A region = 6 consecutive sites; per level the fits go to four lists (3-4 units and 1-2 units, 8 fits per wavefront either way;
"slow" = the subset leaves out the deepest candidate) and a wavefront runs as many passes as its longest fit.
Prints the wave-passes of (a) site order, (b) sites ordered within the region by a proxy, (c) by the true pass count.
usage: python tools/em_lockstep_model.py [n_samples] [n_sites]"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import np_restatement as R   # noqa: E402
from oracle import orc                    # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 600

fits = []   # per site: list of levels; level = list of (n_units, slow, passes)
_em = R._em


def lrt_levels(bases, quals, min_af):
    bases = np.asarray(bases, dtype=np.int64); quals = np.asarray(quals, dtype=np.int64)
    ok = (bases >= 0) & (bases < 4)
    bases, quals = bases[ok], quals[ok]
    n = len(bases)
    eps = np.exp(R.MLN10TO10 * quals.astype(np.float64))
    lik = np.repeat((eps / 3.0)[:, None], 4, axis=1)
    lik[np.arange(n), bases] = 1.0 - eps
    depth = np.bincount(bases, minlength=4)[:4]
    total = float(depth.sum())
    levels = []
    if total == 0:
        return levels, depth
    cand = [b for b in range(4) if depth[b] / total >= min_af]
    if not cand:
        return levels, depth
    deepest = max(cand, key=lambda b: (depth[b], -cand.index(b)))

    def update_f(cur, k):
        combs = list(itertools.combinations(cur, k)); lr = []; bp = []; lev = []
        for c in combs:
            f = np.zeros(4); s = sum(int(depth[b]) for b in c)
            for b in c:
                f[b] = depth[b] / s
            marg, expect, p = _em(f, lik)
            lr.append(float(np.log(marg).sum())); bp.append(expect)
            lev.append((len(c), deepest not in c, p))
        return combs, lr, bp, lev

    combs, lr, bp, lev = update_f(cand, len(cand))
    first = lev
    lr_alt = lr[0]
    for k in range(len(cand) - 1, 0, -1):
        combs, lr, bp, lev = update_f(cand, k)
        if first is not None:
            levels.append(first + lev); first = None          # the engine runs the full model with its (n-1)-subsets
        else:
            levels.append(lev)
        chis = [2.0 * (lr_alt - x) for x in lr]
        i_min = int(np.argmin(chis))
        for i in range(len(chis)):
            if chis[i] < chis[i_min]:
                i_min = i
        lr_alt = lr[i_min]
        if chis[i_min] < R.LRT_THRESHOLD:
            cand = list(combs[i_min])
        else:
            break
    if first is not None:
        levels.append(first)
    return levels, depth


b, q, r = orc.synth_tile(1, 0, S, N)
min_af = min(0.001, 100.0 / N)
site_levels = []; proxy = []
for s in range(S):
    lv, depth = lrt_levels(b[s], q[s], min_af)
    site_levels.append(lv)
    d = np.sort(depth)[::-1]
    proxy.append(d[1] / max(1, d.sum()))
    if s % 100 == 99:
        print("site", s + 1, file=sys.stderr, flush=True)


def cost(order_fn, region=6, by_level=None):
    tot = 0; work = 0
    for r0 in range(0, S, region):
        sites = list(range(r0, min(S, r0 + region)))
        for level in range(3):
            order = order_fn(sites, level)
            lists = {0: [], 1: [], 2: [], 3: []}
            for s in order:
                if level < len(site_levels[s]):
                    for (nu, slow, p) in site_levels[s][level]:
                        lists[(0 if nu >= 3 else 2) + (1 if slow else 0)].append(p)
                        work += p * (4 if nu >= 3 else 2)
                        if by_level is not None: by_level[level][0] += p * (4 if nu >= 3 else 2)
            for l, items in lists.items():
                per = 8                                     # 8 items per wavefront in both shapes (2 lanes x 4 units, 4 lanes x 2 units)
                for i in range(0, len(items), per):
                    tot += max(items[i:i + per]) * 32           # a wavefront pass costs the same whatever it holds: 32 rows of lanes
                    if by_level is not None: by_level[level][1] += max(items[i:i + per]) * 32; by_level[level][2 + l] += max(items[i:i + per]) * 32
    return tot, work


def first_passes(s, level):
    lv = site_levels[s]
    if level < len(lv):
        fast = [p for (nu, slow, p) in lv[level] if not slow]
        return max(fast) if fast else 0
    return 0


a, w = cost(lambda sites, level: sites)
bb, _ = cost(lambda sites, level: sorted(sites, key=lambda s: proxy[s]))
c, _ = cost(lambda sites, level: sorted(sites, key=lambda s: first_passes(s, level)))
print(f"N={N} sites={S}: useful row-passes {w}; wave row-passes site order {a} ({w / a:.3f} useful), by proxy (second depth fraction) {bb} ({a / bb:.3f}x), by true passes {c} ({a / c:.3f}x)")
for rg in (6, 8, 16):
    bl = [[0] * 6 for _ in range(3)]
    cost(lambda sites, level: sites, rg, bl)
    print('region', rg, 'per level [useful, wave cost, cost of lists fast4 slow4 fast2 slow2]:', bl)
for rg in (4, 6, 8, 12, 16, 24, 48):
    t, _ = cost(lambda sites, level: sites, rg)
    print(f"region of {rg} sites: wave row-passes {t} ({w / t:.3f} useful)")
p_all = [p for lv in site_levels for L in lv for (_, slow, p) in L if not slow]
print("fast fits: passes percentiles 10/50/90", np.percentile(p_all, [10, 50, 90]), "corr(proxy, passes of the site's first level)",
      np.corrcoef(proxy, [first_passes(s, 0) for s in range(S)])[0, 1])
