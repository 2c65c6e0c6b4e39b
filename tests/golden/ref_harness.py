"""Container-only harness that imports the *reference* (read-only, /root/reference)
on CPU so that golden fixtures can be generated from its own code.

This file is the build's own code: it contains no reference source.  It is only
usable where /root/reference exists (the authoring container); on the GPU box the
committed fixtures under tests/golden/ are used instead and this module is never
imported.  Recipe from SURVEY.md §8(c):

  * stub modules for cv2 / open3d / tensorboardX / torchvision (imported at module
    top in the reference but unused on the hot path),
  * a `torch_scatter` stand-in implementing its *documented* semantics with core
    torch ops (third-party arithmetic, version unpinned in the reference),
  * Tensor.cuda -> identity because reference forwards hard-code `.cuda()`
    (IMGPCEncoder.py:130-134, MultiHeadModel.py:68).
"""
import os
import sys
import types

REF_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _scatter_index_dimsize(index, dim_size):
    if dim_size is None:
        dim_size = int(index.max()) + 1
    return dim_size


def _make_torch_scatter():
    import torch
    m = types.ModuleType("torch_scatter")

    def _out_shape(src, dim, size):
        shp = list(src.shape)
        shp[dim] = size
        return shp

    def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
        size = _scatter_index_dimsize(index, dim_size)
        res = torch.zeros(_out_shape(src, dim, size), dtype=src.dtype, device=src.device)
        return res.scatter_add_(dim, index, src)

    def scatter_max(src, index, dim=-1, out=None, dim_size=None):
        size = _scatter_index_dimsize(index, dim_size)
        res = torch.zeros(_out_shape(src, dim, size), dtype=src.dtype, device=src.device)
        res = res.scatter_reduce(dim, index, src, reduce="amax", include_self=False)
        return res, None

    def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
        size = _scatter_index_dimsize(index, dim_size)
        tot = scatter_sum(src, index, dim, dim_size=size)
        cnt = scatter_sum(torch.ones_like(src), index, dim, dim_size=size)
        return tot / cnt.clamp(min=1)

    m.scatter_sum = scatter_sum
    m.scatter_add = scatter_sum
    m.scatter_max = scatter_max
    m.scatter_mean = scatter_mean
    return m


_loaded = {}


def load_reference():
    """Returns a namespace with the reference's modules (models, env, config, utils)."""
    if _loaded:
        return _loaded["ns"]
    if not reference_available():
        raise RuntimeError("reference tree not present; use the committed fixtures")
    import torch
    sys.dont_write_bytecode = True
    for name in ("cv2", "open3d", "tensorboardX", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torch_scatter"] = _make_torch_scatter()
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    # the reference uses top-level package names `models`, `utils`, `config`,
    # `environment`; make sure ours (if any were imported) do not shadow them.
    for name in list(sys.modules):
        if name.split(".")[0] in ("models", "utils", "config", "environment"):
            del sys.modules[name]
    sys.path.insert(0, REF_ROOT)
    try:
        import models as ref_models
        import utils as ref_utils
        from environment import environment as ref_env
        from environment import buffer as ref_buffer
        import config as ref_config
        # models/__init__.py re-exports classes under the module names, so fetch the
        # sub-MODULES from sys.modules
        sm = lambda n: sys.modules["models." + n]
        ref_pnu, ref_pointnn, ref_resnet = sm("pointnet_util"), sm("PointNN"), sm("ImageResNet")
        ref_la, ref_enc, ref_ivit, ref_pvit = sm("LinearAttention"), sm("IMGPCEncoder"), sm("ImageViT"), sm("PointViT")
    finally:
        sys.path.remove(REF_ROOT)
    ns = types.SimpleNamespace(models=ref_models, utils=ref_utils, pnu=ref_pnu, pointnn=ref_pointnn,
                               resnet=ref_resnet, la=ref_la, enc=ref_enc, ivit=ref_ivit, pvit=ref_pvit,
                               env=ref_env, buffer=ref_buffer, config=ref_config)
    _loaded["ns"] = ns
    return ns


def load_dataset_module():
    """dataset/KittiDataset.py (for FarthestSampler + the cKDTree assignment)."""
    load_reference()
    sys.path.insert(0, REF_ROOT)
    try:
        import dataset  # noqa: F401
    finally:
        sys.path.remove(REF_ROOT)
    return sys.modules["dataset.KittiDataset"]
