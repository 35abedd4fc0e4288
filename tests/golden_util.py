"""Loader for the fixtures written by tests/golden/make_golden.py."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def model_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g[0-57-9]*.npz")))


def grad_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "h[1234]_grad_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out/")}
    return meta, sd, z["in/ids"], z["in/vals"], out


def load_entmax():
    z = np.load(os.path.join(GOLDEN, "g6_entmax.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return [(m, z["X/" + m["key"]], z["P/" + m["key"]]) for m in meta]


def load_entmax_row_alpha():
    """round 6: utils/entmax.py:31-36 with a tensor alpha (one per row) — [(meta, X, alpha, P)] captured from the reference"""
    z = np.load(os.path.join(GOLDEN, "g6c_entmax_row_alpha.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return [(m, z["X/" + m["key"]], z["A/" + m["key"]], z["P/" + m["key"]]) for m in meta]


def load_entmax_row_alpha_grads():
    """... and the reference's gradients for a random dY: [(meta, X, alpha, dY, dX, dalpha)]"""
    z = np.load(os.path.join(GOLDEN, "g6c_entmax_row_alpha.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return [(m, z["X/" + m["key"]], z["A/" + m["key"]], z["dY/" + m["key"]], z["dX/" + m["key"]], z["dA/" + m["key"]]) for m in meta]
