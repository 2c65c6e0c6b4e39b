#!/usr/bin/env python3
"""Shapes of every cmr_linear_f32 call of one registration iteration at BASELINE configs[1] (geo forward + one agent step), with the
HIP-event time of each: which calls a specialised streaming kernel would serve.  python tools/linear_shapes.py"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BM
from cmr_agent_amd import _lib
from cmr_agent_amd.environment import environment as env
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.utils import synthetic


def main():
    dev = torch.device("cuda", 0); w = BM.WORKLOAD
    cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
    geo, agent, _ = BM.load_models(cfg, dev)
    batch = synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], BM.hip_fps(dev), BM.hip_nearest(dev), seed=cfg.seed, n_circle=16, device=dev)
    recs = []
    orig = _lib.call

    def hook(name, *args, **kw):
        if name != "cmr_linear_f32":
            return orig(name, *args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = orig(name, *args, **kw); e1.record()
        recs.append((args, e0, e1))
        return rc
    with torch.no_grad():
        data = dict(batch); geo(data)
        pose, target = env.init(data)
        s2, s3 = env.observation_from_a_pose(data, pose); agent(s2, s3)
        torch.cuda.synchronize()
        _lib.call = hook
        data = dict(batch); geo(data)
        pose, target = env.init(data)
        s2, s3 = env.observation_from_a_pose(data, pose); agent(s2, s3)
        torch.cuda.synchronize()
        _lib.call = orig
    agg = collections.OrderedDict()
    for args, e0, e1 in recs:
        # cmr_linear_f32(x1, ld1, k1, x2, ld2, k2, idx2, div2, w, ldw, bias, res, ldres, res_mod, y, ldy, rows, n_out, act, act_param, stream)
        x1, ld1, k1, x2, ld2, k2, idx2, div2, wt, ldw, bias, res, ldres, res_mod, y, ldy, rows, n_out, act = args[:19]
        key = (rows, k1, k2 if x2 else 0, n_out, ld1, ldy, bool(idx2), bool(res), ldres if res else 0, res_mod if res else 0, act)
        d = agg.setdefault(key, [0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1)
    print("rows k1 k2 n_out ld1 ldy gather res ldres res_mod act : calls, total us")
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(k, n, "%.1f" % (1e3 * ms))


main()
