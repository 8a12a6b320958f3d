"""Reader/writer of the golden fixtures (inputs + expected outputs of the basetype path).

The reference ships no golden outputs and cannot be built in this image ("Oracle"), so
these vectors were produced by oracle/basetype_oracle.c via tests/golden/make_golden.py -- PARITY
UNPINNED.  They are data only: concatenated per-site base/qual vectors and the expected BaseType fields.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS_I = ["called", "n_alt", "n_kept", "n_fits", "n_passes", "status"]
FIELDS_F = ["var_qual", "chi", "depth_total", "lr_alt"]


def save_golden(name, sites, min_afs, expected):
    offs = np.zeros(len(sites) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(b) for b, _, _ in sites])
    arrs = dict(
        offsets=offs,
        bases=np.concatenate([np.asarray(b, dtype=np.int8) for b, _, _ in sites]) if offs[-1] else np.zeros(0, np.int8),
        quals=np.concatenate([np.asarray(q, dtype=np.int8) for _, q, _ in sites]) if offs[-1] else np.zeros(0, np.int8),
        ref=np.array([r for _, _, r in sites], dtype=np.int8),
        min_af=np.array(min_afs, dtype=np.float64),
        alt_base=np.array([e["alt_base"] + [0] * (3 - len(e["alt_base"])) for e in expected], dtype=np.int8),
        af=np.array([e["af"] + [0.0] * (3 - len(e["af"])) for e in expected], dtype=np.float64),
        kept=np.array([e["kept"] + [0] * (4 - len(e["kept"])) for e in expected], dtype=np.int8),
        depth=np.array([e["depth"] for e in expected], dtype=np.int32),
    )
    for f in FIELDS_I:
        key = f if f != "n_kept" else None
        arrs[f] = np.array([len(e["kept"]) if f == "n_kept" else e[key] for e in expected], dtype=np.int32)
    for f in FIELDS_F:
        arrs[f] = np.array([e[f] for e in expected], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name), **arrs)


def load_golden(name):
    with np.load(os.path.join(HERE, name)) as zf:
        z = {k: zf[k] for k in zf.files}            # decompress each array once, not per access
    n = len(z["ref"])
    exp = []
    for i in range(n):
        na, nk = int(z["n_alt"][i]), int(z["n_kept"][i])
        e = dict(alt_base=[int(x) for x in z["alt_base"][i][:na]], af=[float(x) for x in z["af"][i][:na]],
                 kept=[int(x) for x in z["kept"][i][:nk]], depth=[int(x) for x in z["depth"][i]])
        for f in FIELDS_I:
            if f != "n_kept":
                e[f] = int(z[f][i])
        for f in FIELDS_F:
            e[f] = float(z[f][i])
        exp.append(e)
    return dict(offsets=z["offsets"], bases=z["bases"], quals=z["quals"], ref=z["ref"], min_af=z["min_af"],
                expected=exp)
