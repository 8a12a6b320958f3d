"""Differential run of the skipped subsets (include/bvc.h "em_prune") on random shallow-to-deep sites (GPU): every record with
em_prune = 1 against em_prune = 0, byte for byte apart from the two run counts; counts how often a level's last subset was skipped
and how often it had to be run.  usage: python tools/prune_fuzz.py [tiles=40]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from basevarc_amd import Context

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2026)
S, W = 4096, 768
skipped = ran_all = sites = called = 0
with Context(0) as c1, Context(0) as c0:
    c0.set_tuning("em_prune", 0)
    for t in range(tiles):
        depth = rng.integers(0, W + 1, size=S) if t % 2 else np.minimum(W, rng.geometric(0.08, size=S))
        # allele mixtures: up to four alleles with random weights, some sites with two alleles of equal weight
        w = rng.dirichlet(rng.choice([0.05, 0.3, 1.0]) * np.ones(4), size=S)
        eq = rng.random(S) < 0.1
        w[eq] = np.array([0.5, 0.5, 0.0, 0.0])
        B = np.full((S, W), -1, dtype=np.int8)
        cum = np.cumsum(w, axis=1)
        u = rng.random((S, W))
        base = (u[:, :, None] > cum[:, None, :]).sum(axis=2).clip(0, 3).astype(np.int8)
        perm = np.argsort(rng.random((S, 4)), axis=1).astype(np.int8)        # which base is which
        base = np.take_along_axis(perm, base.astype(np.int64), axis=1)
        covered = np.arange(W)[None, :] < depth[:, None]
        B[covered] = base[covered]
        qlo = rng.choice([2, 10, 20])
        Q = rng.integers(qlo, rng.choice([30, 41, 60]) + 1, size=(S, W)).astype(np.int8)
        R = rng.integers(0, 4, size=S).astype(np.int8)
        m = float(rng.choice([0.001, 0.01, 0.05, 0.2]))
        a = c1.lrt_dense(B, Q, R, m); b = c0.lrt_dense(B, Q, R, m)
        x = a.copy(); x["n_fits"] = b["n_fits"]; x["n_passes"] = b["n_passes"]
        if x.tobytes() != b.tobytes():
            bad = np.nonzero([x[i].tobytes() != b[i].tobytes() for i in range(S)])[0]
            print("MISMATCH tile", t, "sites", bad[:10], a[bad[0]], b[bad[0]]); sys.exit(1)
        skipped += int((b["n_fits"].astype(int) - a["n_fits"].astype(int)).sum())
        levels = ((b["n_fits"] >= 3).astype(int) + (b["n_fits"] >= 8).astype(int))     # EM levels entered (roughly)
        ran_all += int(((b["n_fits"] == a["n_fits"]) & (b["n_fits"] >= 3)).sum())
        sites += S; called += int(b["called"].sum())
        print(f"tile {t}: min_af {m} identical; fits run {int(a['n_fits'].sum())} of {int(b['n_fits'].sum())}, passes {int(a['n_passes'].sum())} of {int(b['n_passes'].sum())}", flush=True)
print(f"{sites} sites ({called} called): every field but the run counts identical; {skipped} subsets skipped; {ran_all} sites with nested levels ran every subset")
