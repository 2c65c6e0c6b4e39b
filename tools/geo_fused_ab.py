"""A/B of the geometric update's gradients with the transformer blocks on the fused train-mode kernels (GeoUpdate.FUSED_VIT) and on the op-by-op
tape, both against oracle autograd (tests' GEO_TRAIN_CASE, dropout off): per-tensor max error relative to the tensor's own scale and to the
model's largest gradient entry; the sign-stable share after one Adam step; the losses of two free-running steps."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases as C, golden_util as G
from oracle import train_oracle as TO
from cmr_agent_amd.models import MultiHeadModel
from cmr_agent_amd.train import GeoUpdate
from cmr_agent_amd.utils.checkpoint import load_checked

def main():
    dev = "cuda"
    specs = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
    cfg = C.e2e_config(C.GEO_TRAIN_CASE)
    geo_sd, _ = C.e2e_state_dicts(specs)
    sd0 = {k: v for k, v in geo_sd.items() if not k.endswith("num_batches_tracked")}
    batches = C.geo_train_batches()
    todev = lambda b: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    with torch.enable_grad():
        out, og = TO.geo_forward_backward({k: x.clone() for k, x in sd0.items()}, batches[0], cfg, True)
        osd, ohist = TO.geo_adam_train(sd0, batches, cfg, True)
    gmax = max(float(g.abs().max()) for g in og.values())
    res = {}
    for fused in (False, True):
        GeoUpdate.FUSED_VIT = fused
        m = MultiHeadModel(cfg); load_checked(m, geo_sd); m = m.to(dev)
        up = GeoUpdate(m, cfg, dropout=False)
        up.forward_backward(todev(batches[0])); torch.cuda.synchronize()
        named = dict(m.named_parameters(remove_duplicate=False))
        lg = {k: up.bucket.by_id[id(p)].view(up.bucket.grads).detach().cpu().double() for k, p in named.items() if p.requires_grad}
        rows = []
        for k, g in og.items():
            if k not in lg: continue
            d = float((lg[k].reshape(g.shape) - g.double()).abs().max()); mx = float(g.abs().max())
            rows.append((d / gmax, d / max(mx, 1e-30), mx / gmax, k))
        res[fused] = (lg, rows)
        rows.sort(reverse=True)
        print("FUSED_VIT", fused, ": worst by model scale", [(round(a, 6), k[-60:]) for a, _, _, k in rows[:4]])
        rows2 = sorted([r for r in rows if r[2] > 1e-4], key=lambda r: -r[1])
        print("    worst by own scale (tensors above 1e-4 of the model's largest entry)", [(round(b, 4), round(c, 6), k[-60:]) for _, b, c, k in rows2[:6]])
        m2 = MultiHeadModel(cfg); load_checked(m2, geo_sd); m2 = m2.to(dev)
        up2 = GeoUpdate(m2, cfg, dropout=False)
        hist = [{k: float(v) for k, v in up2.step(todev(b)).items()} for b in batches]
        for i in range(2):
            print("    step", i, {k: (round(hist[i][k], 7), round(float(ohist[i][k]), 7)) for k in C.LOSS_KEYS})
    a, b = res[False][0], res[True][0]
    diff = sorted(((float((a[k] - b[k]).abs().max()) / gmax, k) for k in a), reverse=True)
    print("fused vs op-by-op, max |d grad| / model scale:", [(round(x, 7), k[-60:]) for x, k in diff[:8]])

main()
