"""Which tape sites issue the big elementwise passes (cmr_affine_act_f32 / cmr_axpy_f32 / cmr_act_bwd_f32 on >= 500 000-row maps) of ONE GeoUpdate
step: call-site census (file:line of the innermost cmr_agent_amd/train frame).  python tools/geo_affine_sites.py [num_pt]"""
import os, sys, json, collections, traceback
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


def main():
    dev = torch.device("cuda", 0)
    npt = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    cfg = KittiConfiguration(device=dev, num_pt=npt)
    spec = json.load(open(os.path.join(BM.ROOT, "tests", "golden", "specs.json")))
    model = MultiHeadModel(cfg); load_checked(model, hashfill.make_state_dict(spec["geo"], BM.GEO_TAG)); model = model.to(dev)
    up = GeoUpdate(model, cfg)
    batch = synthetic.make_batch(cfg.train_batch_size, cfg.num_pt, cfg.cropped_img_H, cfg.cropped_img_W, cfg.num_node, BM.hip_fps(dev), BM.hip_nearest(dev),
                                 seed=cfg.seed, n_circle=512, device=dev)
    up.step(batch); torch.cuda.synchronize()
    protos, sites, orig = _lib.prototypes(), collections.Counter(), _lib.call
    def hook(name, *args, **kw):
        if name in ("cmr_affine_act_f32", "cmr_axpy_f32", "cmr_act_bwd_f32", "cmr_concat_rows_f32", "cmr_gather_rows_f32", "cmr_dropout_f32"):
            a = dict(zip(protos[name][2], args))
            rows = int(a.get("rows", 0) or 0)
            if rows >= 500000:
                fr = [f for f in traceback.extract_stack() if "cmr_agent_amd/train" in f.filename]
                where = " <- ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in fr[-3:][::-1])
                sites[(name, where)] += 1
        return orig(name, *args, **kw)
    _lib.call = hook
    up.step(batch); torch.cuda.synchronize()
    _lib.call = orig
    for (name, where), n in sites.most_common():
        print("%3d  %-22s %s" % (n, name, where))

main()
