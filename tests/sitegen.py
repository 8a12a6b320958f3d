"""Seeded random pileup columns for the parity tests (numpy; independent of the C generator)."""
import numpy as np


def random_site(rng, nind, af=0.0, qlo=10, qhi=40, ref=None, second_af=0.0):
    ref = int(rng.integers(0, 4)) if ref is None else ref
    alt = (ref + 1 + int(rng.integers(0, 3))) % 4
    alt2 = next(b for b in range(4) if b not in (ref, alt))
    u = rng.random(nind)
    true = np.where(u < af, alt, np.where(u < af + second_af, alt2, ref))
    q = rng.integers(qlo, qhi + 1, nind)
    err = rng.random(nind) < 10.0 ** (-q / 10.0)
    obs = np.where(err, (true + 1 + rng.integers(0, 3, nind)) % 4, true)
    return obs.astype(np.int8), q.astype(np.int8), ref


def caller_min_af(n_total, maf=0.001):
    """min_af as the caller picks it: src/BaseVarC.cpp:541-543."""
    m = 100.0 / n_total
    if m > 0.001:
        m = 0.001
    if maf < m:
        m = maf
    return m
