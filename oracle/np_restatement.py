"""Second, independent restatement of the basetype path in numpy (vectorised over samples).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED.  Written separately from basetype_oracle.c so that the
two restatements check each other (tests/test_oracle.py).  Follows /root/reference:
src/BaseType.cpp:5-139, 237-255 and src/Algorithm.cpp:3-7, 69-130.  Sums are numpy pairwise sums,
so agreement with the C oracle is to rounding (1e-12), not bit for bit.
"""
import itertools
import math

import numpy as np
from scipy import special

MLN10TO10 = -0.23025850929940458   # src/BaseType.h:10
LRT_THRESHOLD = 24.0               # src/BaseType.h:9


def _single_em(freq, lik):          # src/Algorithm.cpp:69-93
    joint = lik * freq[None, :]
    marg = joint.sum(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        post = joint / marg[:, None]
    return marg, post.sum(axis=0) / lik.shape[0]


def _em(freq, lik, iters=100, eps=0.001):   # src/Algorithm.cpp:115-130
    marg, expect = _single_em(freq, lik)
    passes = 1
    for _ in range(iters):
        freq = expect
        nxt, expect = _single_em(freq, lik)
        passes += 1
        with np.errstate(divide="ignore", invalid="ignore"):
            delta = np.abs(np.log(nxt) - np.log(marg)).sum()
        marg = nxt
        if delta < eps:
            break
    return marg, expect, passes


def basetype_lrt(bases, quals, ref_base, min_af, base_comb=(0, 1, 2, 3)):
    bases = np.asarray(bases, dtype=np.int64)
    quals = np.asarray(quals, dtype=np.int64)
    n = len(bases)
    eps = np.exp(MLN10TO10 * quals.astype(np.float64))
    lik = np.repeat((eps / 3.0)[:, None], 4, axis=1)             # src/BaseType.cpp:10-17
    lik[np.arange(n), bases] = 1.0 - eps
    depth = np.bincount(bases, minlength=4)[:4]
    total = float(depth.sum())
    out = dict(called=0, alt_base=[], af=[], var_qual=0.0, chi=0.0, depth=depth.tolist(), n_passes=0)
    if total == 0:
        return out
    cand = [b for b in base_comb if depth[b] / total >= min_af]  # :77-83
    if not cand:
        return out

    def update_f(cur, k):                                         # :41-71
        combs = list(itertools.combinations(cur, k))              # lexicographic by position (:237-255)
        lr, bp = [], []
        for c in combs:
            f = np.zeros(4)
            s = sum(int(depth[b]) for b in c)
            if s > 0:
                for b in c:
                    f[b] = depth[b] / s
            if f.sum() == 0:
                continue
            marg, expect, p = _em(f, lik)
            out["n_passes"] += p
            with np.errstate(divide="ignore", invalid="ignore"):
                lr.append(float(np.log(marg).sum()))
            bp.append(expect)
        return combs, lr, bp

    combs, lr, bp = update_f(cand, len(cand))
    frq, lr_alt, chi = bp[0], lr[0], 0.0
    for k in range(len(cand) - 1, 0, -1):                         # :93-110
        combs, lr, bp = update_f(cand, k)
        chis = [2.0 * (lr_alt - x) for x in lr]
        i_min = 0
        for i in range(1, len(chis)):
            if chis[i] < chis[i_min]:
                i_min = i
        lr_alt, chi = lr[i_min], chis[i_min]
        if chi < LRT_THRESHOLD:
            cand, frq = list(combs[i_min]), bp[i_min]
        else:
            break
    out["chi"] = chi
    out["kept"] = [int(b) for b in cand]
    for b in cand:                                                # :111-116
        if b != ref_base:
            out["alt_base"].append(int(b))
            out["af"].append(float(frq[b]))
    if out["alt_base"]:                                           # :117-135
        out["called"] = 1
        r = depth[cand[0]] / total
        if len(cand) == 1 and total > 10 and r > 0.5:
            out["var_qual"] = 5000.0
        elif chi <= 0:
            out["var_qual"] = 0.0
        else:
            p = float(special.gammaincc(0.5, chi / 2.0))          # chisf(chi, 1): scipy, not kfunc
            out["var_qual"] = (-10 * math.log10(p)) if p else 10000.0
            if out["var_qual"] == 0:
                out["var_qual"] = 0.0
    return out
