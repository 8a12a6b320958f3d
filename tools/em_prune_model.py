"""CPU model of the subset a level need not run (em_items.hip site_decide; include/bvc.h "em_prune"), on the numpy restatement of the
reference: per level, the subset without the deepest candidate against the bound 2 (lr_alt - U) > min chi of the others + slack,
U = sum over the alleles outside the subset of their observations' log(eps / 3).  Prints the share of E+M passes that are never
needed, checks that the bound is never violated by the subset's true chi, and what an oracle that knew the minimum in advance could
skip on top ("ideal": any subset).   usage: python tools/em_prune_model.py n_samples n_sites [coverage]"""
import itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import np_restatement as R
from oracle import orc
N = int(sys.argv[1]); S = int(sys.argv[2]); cov = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
_em = R._em
b, q, r = orc.synth_tile(1, 0, S, N, cov_thr16=int(cov * 65536))
min_af = min(0.001, 100.0 / N)
tot_p = 0; pruned_p = 0; nprune = 0; nslow = 0; minmargin = 1e300; bad = 0; allprune_p = 0
for s in range(S):
    bases = np.asarray(b[s], dtype=np.int64); quals = np.asarray(q[s], dtype=np.int64)
    ok = (bases >= 0) & (bases < 4); bases, quals = bases[ok], quals[ok]
    n = len(bases)
    if n == 0: continue
    eps = np.exp(R.MLN10TO10 * quals.astype(np.float64))
    lik = np.repeat((eps / 3.0)[:, None], 4, axis=1); lik[np.arange(n), bases] = 1.0 - eps
    depth = np.bincount(bases, minlength=4)[:4]; total = float(depth.sum())
    cand = [x for x in range(4) if depth[x] / total >= min_af]
    if not cand: continue
    lmax = np.log(np.maximum(1.0 - eps, eps / 3.0)); lmis = np.log(eps / 3.0)
    def U(c):
        inc = np.isin(bases, list(c))
        return float(lmax[inc].sum() + lmis[~inc].sum())
    def update_f(cur, k):
        combs = list(itertools.combinations(cur, k)); out = []
        for c in combs:
            f = np.zeros(4); sm = sum(int(depth[x]) for x in c)
            for x in c: f[x] = depth[x] / sm
            marg, expect, p = _em(f, lik)
            out.append((c, float(np.log(marg).sum()), p))
        return out
    full = update_f(cand, len(cand)); tot_p += full[0][2]
    lr_alt = full[0][1]
    for k in range(len(cand) - 1, 0, -1):
        deepest = max(cand, key=lambda x: (depth[x], -cand.index(x)))
        fits = update_f(cand, k)
        chis = [2.0 * (lr_alt - x[1]) for x in fits]
        for (c, lr, p) in fits: tot_p += p
        # prune rule: slow = subset without deepest; others run first
        fast = [i for i, (c, lr, p) in enumerate(fits) if deepest in c]
        slow = [i for i, (c, lr, p) in enumerate(fits) if deepest not in c]
        if fast and k >= 1:
            cmin = min(chis[i] for i in fast)
            for i in slow:
                nslow += 1
                lb = 2.0 * (lr_alt - U(fits[i][0]))
                if not (chis[i] >= lb - 1e-9 * abs(lb)): bad += 1
                margin = 1e-6 * (abs(lr_alt) + abs(U(fits[i][0]))) + 1.0
                if lb > cmin + margin:
                    nprune += 1; pruned_p += fits[i][2]; minmargin = min(minmargin, lb - cmin)
        # ideal: all subsets that bound says are not min, given the true min
        cm = min(chis)
        for i, (c, lr, p) in enumerate(fits):
            if 2.0 * (lr_alt - U(c)) > cm + 1.0 and chis[i] != cm: allprune_p += p
        i_min = 0
        for i in range(1, len(chis)):
            if chis[i] < chis[i_min]: i_min = i
        lr_alt = fits[i_min][1]
        if chis[i_min] < 24.0: cand = list(fits[i_min][0])
        else: break
print(f"N={N} cov={cov} sites={S}: passes/site {tot_p / S:.1f}; slow fits {nslow}, pruned {nprune} ({pruned_p / S:.1f} passes/site = {pruned_p / tot_p:.3f}); bound violations {bad}; min gap {minmargin:.3g}; ideal prune (any subset) {allprune_p / tot_p:.3f}")
