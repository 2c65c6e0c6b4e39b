"""Debug aid: which Python lines of the geometric-model update issue torch copies / fills (aten::copy_, aten::fill_, aten::zero_)."""
import collections, json, os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases as C, golden_util as G
from cmr_agent_amd.models import MultiHeadModel
from cmr_agent_amd.train.geo_update import GeoUpdate
from cmr_agent_amd.utils.checkpoint import load_checked
SPECS = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
cfg = C.e2e_config("e2e_small"); batch = C.e2e_batch("e2e_small")
geo_sd, _ = C.e2e_state_dicts(SPECS)
model = MultiHeadModel(cfg); load_checked(model, geo_sd); model = model.to("cuda")
up = GeoUpdate(model, cfg, dropout=False)
data = {k: (v.to("cuda") if torch.is_tensor(v) else v) for k, v in batch.items()}
up.step(data)
sites = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if self.is_cuda or any(torch.is_tensor(x) and x.is_cuda for x in a):
            fr = [x for x in traceback.extract_stack()[:-1] if "cmr_agent_amd" in x.filename]
            if fr:
                sites[(name, os.path.basename(fr[-1].filename), fr[-1].lineno, fr[-1].line[:70])] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ("copy_", "contiguous", "clone", "zero_", "fill_", "to"):
    wrap(n)
for fn in ("zeros", "ones", "full", "zeros_like", "arange"):
    orig = getattr(torch, fn)
    def mk(orig, fn):
        def f(*a, **k):
            fr = [x for x in traceback.extract_stack()[:-1] if "cmr_agent_amd" in x.filename]
            if fr:
                sites[(fn, os.path.basename(fr[-1].filename), fr[-1].lineno, fr[-1].line[:70])] += 1
            return orig(*a, **k)
        return f
    setattr(torch, fn, mk(orig, fn))
up.step(data)
torch.cuda.synchronize()
for k, v in sites.most_common(40):
    print(v, k)
