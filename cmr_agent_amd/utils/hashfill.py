"""Closed-form, machine-independent pseudo-random fills.

The reference's checkpoints are absent (/root/reference/.MISSING_LARGE_BLOBS) and
there is no network, so weights and synthetic inputs are produced by a pure
uint64 hash (splitmix64) evaluated in numpy.  The same (name, index) always gives
the same float on any machine, with no dependence on torch's RNG streams.

Used by: tests/golden/make_golden.py (reference side), the oracle tests, the GPU
parity tests and bench.py, so that all of them see identical parameters.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def name_seed(name: str) -> int:
    """FNV-1a 64-bit hash of a key string (order independent seeding)."""
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def uniform(name: str, shape, lo=-1.0, hi=1.0) -> np.ndarray:
    """float64 array of `shape`, U[lo, hi), element i keyed on (name, i)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(name_seed(name))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = _splitmix64(_splitmix64(idx ^ seed) + seed)
    u01 = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u01).reshape(shape)


_ALIASES = (("embeddings.embedding_layers.0.", "embeddings.mini_resnet."),
            ("embeddings.embedding_layers.1.", "embeddings.patch_embeddings."))
_SKIP = ("num_batches_tracked", "position_embeddings")


def canonical_key(key: str) -> str:
    """ImageViT.py:17-23 registers mini_resnet / patch_embeddings a second time under
    embedding_layers.{0,1}; both names must receive the same values."""
    for a, b in _ALIASES:
        key = key.replace(a, b)
    return key


def fill_value(key: str, shape, tag="") -> np.ndarray:
    """float32 values for one state_dict entry (SURVEY.md 8c recipe, keyed by NAME):
    >=2-D: U(-1,1)*sqrt(3/fan_in); 1-D '*.weight' (BN/LN gamma): 1+0.1u; other 1-D
    (bias/beta): 0.05u; running_mean: 0.1u; running_var: 1+0.25|u|."""
    key = canonical_key(key)
    u = uniform(tag + key, tuple(shape))
    if len(shape) >= 2:
        v = u * np.sqrt(3.0 / int(np.prod(shape[1:])))
    elif key.endswith("running_mean"):
        v = 0.1 * u
    elif key.endswith("running_var"):
        v = 1.0 + 0.25 * np.abs(u)
    elif key.endswith(".weight"):
        v = 1.0 + 0.1 * u
    else:
        v = 0.05 * u
    return v.astype(np.float32)


def fill_state_dict(sd, tag=""):
    """In-place deterministic fill of a torch state_dict; integer entries
    (num_batches_tracked) and position_embeddings are left untouched."""
    import torch
    for key, t in sd.items():
        if key.endswith(_SKIP) or not torch.is_floating_point(t):
            continue
        t.copy_(torch.from_numpy(fill_value(key, tuple(t.shape), tag)).reshape(t.shape))
    return sd


def make_state_dict(spec, tag=""):
    """{key: shape} -> {key: float32 tensor}; same values fill_state_dict would write."""
    import torch
    out = {}
    for key, shape in spec.items():
        if key.endswith(_SKIP):
            continue
        out[key] = torch.from_numpy(fill_value(key, tuple(shape), tag)).reshape(tuple(shape))
    return out
