"""Dispatch-mode logger shared by reg_torch_ops.py / geo_torch_ops.py: counts the device-side aten ops by (op, innermost package line)."""
import collections, os, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode

sites = collections.Counter()
clone_bytes = collections.Counter()          # bytes produced by clone() per site
VIEW = ("view", "reshape", "expand", "permute", "transpose", "t.default", "unsqueeze", "squeeze", "slice", "select", "as_strided", "alias", "detach",
        "_unsafe_view", "unbind", "split", "narrow", "empty", "sym_", "size", "stride", "is_", "_local_scalar", "lift_fresh", "unfold")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name for v in VIEW):
            return out
        flat = list(args) + list((kwargs or {}).values()) + (list(out) if isinstance(out, (tuple, list)) else [out])
        if any(torch.is_tensor(x) and x.is_cuda for x in flat):
            fr = [x for x in traceback.extract_stack()[:-1] if "/cmr_agent_amd/" in x.filename or x.filename.endswith("bench.py")]
            site = "%s:%d  %s" % (os.path.basename(fr[-1].filename), fr[-1].lineno, (fr[-1].line or "")[:80]) if fr else "?"
            sites[(name.replace("aten.", ""), site)] += 1
            if "clone" in name and torch.is_tensor(out):
                clone_bytes[site] += out.numel() * out.element_size()
        return out


