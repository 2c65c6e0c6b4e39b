"""Wall time of the two phases of a registration iteration (geo forward / the agent loop), each captured in its own
hipGraph and replayed -- where the remaining time is, without profiler distortion."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BM
from cmr_agent_amd.environment import environment as env
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.utils import synthetic

def graph_time(fn, reps=5):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, out

def main():
    from cmr_agent_amd import ops
    ops.CONV_BF16 = "bf16" in sys.argv            # python tools/phases.py sub bf16
    from cmr_agent_amd import _lib
    for a in sys.argv:
        if a.startswith("cus="):
            ops._policy.cu_budget = int(a[4:])          # `cu_budget` argument of every convolution call from here on
    dev = torch.device("cuda", 0); w = BM.WORKLOAD
    cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
    geo, agent, _ = BM.load_models(cfg, dev)
    batch = synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], BM.hip_fps(dev), BM.hip_nearest(dev), seed=cfg.seed, n_circle=16, device=dev)
    data = dict(batch)
    with torch.no_grad():
        geo(data)
    def geo_fn():
        d = dict(batch); geo(d); return d
    def agent_fn():
        pose, target = env.init(data)
        env.to_disentangled(target, data['pc'])
        for _ in range(cfg.action_num):
            s2, s3 = env.observation_from_a_pose(data, pose, materialize_state_2d=False)
            r, t, _ = agent(s2, s3)
            ar, at = agent.action_from_logits(r, t, deterministic=True)
            pose = env.step(ar, at, pose, cfg)
        return pose
    tg, _ = graph_time(geo_fn)
    ta, _ = graph_time(agent_fn)
    print("geo forward %.2f ms   agent loop (%d steps) %.2f ms   sum %.2f ms" % (tg, cfg.action_num, ta, tg + ta))
    if len(sys.argv) > 1:     # sub-phases of one agent step
        pose, target = env.init(data)
        s2, s3 = env.observation_from_a_pose(data, pose)
        B, N = s3.shape[0], s3.shape[2]
        from cmr_agent_amd.models.ImageResNet import to_nhwc
        s3r = torch.as_strided(s3, (B * N, 8), (8, 1)) if s3.stride(1) == 1 else None
        split = getattr(s2, "_cmr_split", None)
        to, _ = graph_time(lambda: env.observation_from_a_pose(data, pose, materialize_state_2d=False), 20)
        t2, _ = graph_time(lambda: agent._embed_2d(to_nhwc(s2), B, split), 20)
        t3, _ = graph_time(lambda: agent._embed_3d_any(s3r, B, N), 20)
        tf, _ = graph_time(lambda: agent(s2, s3), 20)
        print("agent step: observation %.3f   2-D embed %.3f   3-D embed %.3f   full forward %.3f ms" % (to, t2, t3, tf))
    if len(sys.argv) > 1:     # sub-phases of the geo forward
        ed = geo.encoder_decoder; enc = ed.encoder
        img = batch['img'].contiguous()
        t1, _ = graph_time(lambda: enc.img_transformer.forward_cl(img))
        from cmr_agent_amd.models.PointViT import PointGeometry
        t2, _ = graph_time(lambda: enc.pt_transformer.forward_cl(PointGeometry(batch['pc'], batch['node'], batch['pt2node'])))
        t3, _ = graph_time(lambda: enc.forward_cl(dict(batch)))
        t4, _ = graph_time(lambda: ed.forward_cl(dict(batch)))
        print("image tower %.2f   point tower %.2f   encoder %.2f   encoder+decoder %.2f   full geo %.2f" % (t1, t2, t3, t4, tg))
        with torch.no_grad():
            cl = enc.forward_cl(dict(batch))
        B, T, Q = cl["B"], cl["T"], cl["Q"]
        ip0, pp0 = cl["img_proxy"].clone(), cl["pt_proxy"].clone()
        from cmr_agent_amd.utils.streams import fork_join
        def coarse():
            ip, pp = ip0, pp0
            for i in range(cfg.num_ca_layer_coarse):
                ip = enc.p2i_ca_layers[i].rows(ip, pp, B, T, Q)
                pp = enc.i2p_ca_layers[i].rows(pp, ip, B, Q, T)
                a, b = ip, pp
                pp, ip = fork_join(lambda: enc.pt_sa_layers[i].rows(b, None, B, Q, Q), lambda: enc.img_sa_layers[i].rows(a, None, B, T, T))
            return ip, pp
        t5, _ = graph_time(coarse, 10)
        t6, _ = graph_time(lambda: enc.p2i_ca_layers[0].rows(ip0, pp0, B, T, Q), 20)
        print("coarse matcher (6 layers) %.2f ms; one cross block %.3f ms" % (t5, t6))

if __name__ == "__main__":
    main()
