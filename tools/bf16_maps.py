"""Round 6: the bf16 3x3 convolutions of ONE eager registration iteration, launch by launch, with the map each one runs on.

    python3 tools/bf16_maps.py --workload c3 [--out gpurun_out/bf16_maps_c3.json] [--iters 3]

Every branch on one stream (CMR_STREAMS=0), so that a launch's HIP-event time and its PMC counters are its own.  The LAST iteration's
ordered call list (entry point, B, H, W, Cin, Cout, stride, pool, stored types, algorithmic bytes and FLOPs, microseconds alone) is written as
JSON; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (tools/_pmc_path_bf16.sh) the same program gives the counters of
the same launches, and tools/bf16_maps_join.py joins the two by launch order into the per-map-size table."""
import argparse
import json
import os
import sys

os.environ["CMR_STREAMS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench as BM  # noqa: E402
from cmr_agent_amd import _lib, ops  # noqa: E402
from cmr_agent_amd.config import KittiConfiguration, NuScenesConfiguration  # noqa: E402
from cmr_agent_amd.utils import synthetic, workmodel  # noqa: E402

CONVS = ("cmr_conv3x3_bf16_nhwc_f32", "cmr_conv3x3_bf16io_nhwc", "cmr_conv3x3_bf16_pro_nhwc_f32")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=sorted(BM.WORKLOADS))
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    w = BM.WORKLOADS[args.workload]
    ops.CONV_BF16 = True
    Cfg = NuScenesConfiguration if w["cfg"] == "nuscenes" else KittiConfiguration
    cfg = Cfg(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
    geo, agent, _ = BM.load_models(cfg, dev)
    batch = synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], BM.hip_fps(dev), BM.hip_nearest(dev), seed=cfg.seed, n_circle=16, device=dev)
    names = {n: _lib.prototypes()[n][2] for n in CONVS}
    orig = _lib.call
    log = []

    def logged(name, *a, allow_unsupported=False, work_extra=None):
        if name not in CONVS:
            return orig(name, *a, allow_unsupported=allow_unsupported, work_extra=work_extra)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = orig(name, *a, allow_unsupported=allow_unsupported, work_extra=work_extra)
        e1.record()
        if rc != _lib.UNSUPPORTED:
            d = dict(zip(names[name], a))
            fl, by = workmodel.work(name, a, work_extra)
            log.append(dict(entry=name, B=d["B"], H=d["H"], W=d["W"], Cin=d["Cin"], Cout=d["Cout"], stride=d.get("stride", 1), pool=d.get("pool", 1),
                            res=bool(d.get("res")), post=bool(d.get("post")), x_bf16=int(d.get("x_bf16", 0) or 0), y_bf16=int(d.get("y_bf16", 0) or 0),
                            flops=fl, bytes=by, _ev=(e0, e1)))
        return rc

    _lib.call = logged
    with torch.no_grad():
        for it in range(args.iters):
            del log[:]
            BM.registration_step(geo, agent, cfg, batch)
            torch.cuda.synchronize()
    _lib.call = orig
    for d in log:
        e0, e1 = d.pop("_ev")
        d["us"] = 1e3 * e0.elapsed_time(e1)
    out = dict(workload=args.workload, name=w["name"], launches=len(log), calls=log)
    path = args.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bf16_maps_%s.json" % args.workload)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"))
    print("%s: %d bf16 convolution launches per iteration, %.3f ms, %.1f MB algorithmic -> %s" % (
        args.workload, len(log), sum(d["us"] for d in log) / 1e3, sum(d["bytes"] for d in log) / 1e6, path))


if __name__ == "__main__":
    main()
