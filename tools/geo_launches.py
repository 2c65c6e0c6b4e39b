"""Launch census of ONE GeoUpdate step (Train_Geo.py:166-174) at --num-pt points: C-ABI calls by entry point (count, summed HIP-event ms), eager,
side streams off.  python tools/geo_launches.py [num_pt [HxW]]"""
import os, sys, json, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CMR_STREAMS", "0")
import bench as BM
from cmr_agent_amd import _lib
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.models import MultiHeadModel
from cmr_agent_amd.train import GeoUpdate
from cmr_agent_amd.utils import hashfill, synthetic
from cmr_agent_amd.utils.checkpoint import load_checked
from cmr_agent_amd.utils.workmodel import CallTimer

def main():
    dev = torch.device("cuda", 0)
    npt = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    kw = {}
    if len(sys.argv) > 2:                                   # HxW, e.g. 352x1216
        H, W = (int(v) for v in sys.argv[2].lower().split("x"))
        kw = dict(cropped_img_H=H, cropped_img_W=W)
    cfg = KittiConfiguration(device=dev, num_pt=npt, **kw)
    spec = json.load(open(os.path.join(BM.ROOT, "tests", "golden", "specs.json")))
    model = MultiHeadModel(cfg); load_checked(model, hashfill.make_state_dict(spec["geo"], BM.GEO_TAG)); model = model.to(dev)
    up = GeoUpdate(model, cfg)
    batch = synthetic.make_batch(cfg.train_batch_size, cfg.num_pt, cfg.cropped_img_H, cfg.cropped_img_W, cfg.num_node, BM.hip_fps(dev), BM.hip_nearest(dev),
                                 seed=cfg.seed, n_circle=512, device=dev)
    up.step(batch); torch.cuda.synchronize()
    # second hook: (entry point, rows, C) of every call with its HIP-event time -> which passes run over the big row maps
    protos, recs, orig = _lib.prototypes(), [], _lib.call
    def hook(name, *args, **kw):
        names = protos[name][2]
        a = dict(zip(names, args))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = orig(name, *args, **kw); e1.record()
        tag = name
        if name == "cmr_linear_f32":       # which row GEMMs are one-source K = 64 (the register-weights kernel's shape)
            tag = "%s[k1=%d%s]" % (name, int(a.get("k1", 0)), "+x2" if a.get("x2") else "")
        recs.append((tag, int(a.get("rows", a.get("nseg", 0)) or 0), int(a.get("C", a.get("n_out", a.get("n", 0))) or 0), e0, e1))
        return rc
    _lib.call = hook
    up.step(batch); torch.cuda.synchronize()
    _lib.call = orig
    import collections
    agg = collections.OrderedDict()
    for name, rows, C, e0, e1 in recs:
        d = agg.setdefault((name, rows, C), [0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1)
    print("entry point (rows, C)                                   calls      ms")
    for (name, rows, C), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%-40s %8d %5d %5d %7.2f" % (name, rows, C, n, ms))
    with CallTimer() as ct:
        up.step(batch); torch.cuda.synchronize()
    tab = ct.table()
    print("entry point                         calls      ms   us/call")
    for d in tab:
        print("%-34s %6d %7.2f %8.1f" % (d["name"], d["calls"], d["ms"], 1e3 * d["ms"] / d["calls"]))
    print("TOTAL calls %d, summed ms %.1f" % (sum(d["calls"] for d in tab), sum(d["ms"] for d in tab)))

main()
