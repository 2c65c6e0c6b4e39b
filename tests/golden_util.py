"""Fixture format shared by tests/golden/make_golden.py and the tests.

A fixture is DATA: for every named tensor of a case we keep its shape, a strided
sample of at most MAX_SAMPLES elements of the flattened tensor and three float64
moments (sum, sum|x|, sum x^2).  Small tensors are therefore stored in full.
"""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# the generators write here; tests/test_golden_recipe.py points it at a scratch directory to check that the committed
# recipe still reproduces the committed fixtures
OUT_DIR = os.environ.get("CMR_GOLDEN_OUT", GOLDEN_DIR)
MAX_SAMPLES = 4096


def _np(t):
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu()
        if t.dtype == torch.bool:
            t = t.to(torch.uint8)
        return t.numpy()
    return np.asarray(t)


def summarize(t):
    a = _np(t)
    flat = a.reshape(-1)
    stride = max(1, -(-flat.size // MAX_SAMPLES))
    f64 = flat.astype(np.float64)
    return dict(shape=np.array(a.shape, dtype=np.int64), stride=np.int64(stride),
                sample=np.ascontiguousarray(flat[::stride]),
                stats=np.array([f64.sum(), np.abs(f64).sum(), (f64 * f64).sum()]))


def pack(named):
    out = {}
    for name, t in named.items():
        for k, v in summarize(t).items():
            out["%s::%s" % (name, k)] = v
    return out


def save_case(case, named):
    np.savez_compressed(os.path.join(OUT_DIR, case + ".npz"), **pack(named))


def load_case(case):
    z = np.load(os.path.join(GOLDEN_DIR, case + ".npz"))
    names = sorted({k.split("::")[0] for k in z.files})
    return {n: dict(shape=z[n + "::shape"], stride=int(z[n + "::stride"]), sample=z[n + "::sample"],
                    stats=z[n + "::stats"]) for n in names}


def compare(name, got, fx, atol, rtol, exact_frac=1.0):
    """Returns an error string or None.  For integer / bool fixtures `exact_frac` is
    the minimum fraction of sampled elements that must be equal (1.0 = bit exact)."""
    a = _np(got)
    if tuple(a.shape) != tuple(int(s) for s in fx["shape"]):
        return "%s: shape %s != %s" % (name, tuple(a.shape), tuple(fx["shape"]))
    s = a.reshape(-1)[::fx["stride"]]
    ref = fx["sample"]
    if ref.dtype.kind in "iub":
        eq = float((s.astype(np.int64) == ref.astype(np.int64)).mean()) if ref.size else 1.0
        if eq < exact_frac:
            return "%s: only %.5f of sampled integers equal (need %.5f)" % (name, eq, exact_frac)
        return None
    s = s.astype(np.float64)
    ref = ref.astype(np.float64)
    err = np.abs(s - ref)
    tol = atol + rtol * np.abs(ref)
    if not np.all(err <= tol):
        i = int(np.argmax(err - tol))
        return "%s: max|d|=%.3e at sample %d (got %.6g ref %.6g, tol %.3e); %d/%d out of tol" % (
            name, err.max(), i, s[i], ref[i], tol[i], int((err > tol).sum()), err.size)
    # moments catch errors between the sampled positions
    f64 = a.reshape(-1).astype(np.float64)
    n = max(1, f64.size)
    got_stats = np.array([f64.sum(), np.abs(f64).sum(), (f64 * f64).sum()])
    scale = fx["stats"][1] / n + atol
    if abs(got_stats[1] - fx["stats"][1]) / n > 10 * (atol + rtol * scale):
        return "%s: mean|x| differs: got %.8g ref %.8g" % (name, got_stats[1] / n, fx["stats"][1] / n)
    return None


def assert_case(case, named, atol, rtol, exact_frac=1.0, only=None):
    fx = load_case(case)
    errs = []
    for name, t in named.items():
        if only is not None and name not in only:
            continue
        if name not in fx:
            errs.append("%s: not in fixture %s" % (name, case))
            continue
        e = compare(name, t, fx[name], atol, rtol, exact_frac)
        if e:
            errs.append(e)
    assert not errs, "fixture %s mismatches:\n  " % case + "\n  ".join(errs)
