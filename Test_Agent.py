#!/usr/bin/env python3
"""Evaluation entry point with the reference's structure and flag (Test_Agent.py:108-206:
`python Test_Agent.py --dataset kitti|nuscenes`), running the HIP path.

There are no KITTI / nuScenes files and no checkpoints in this environment, so the loader is the
synthetic generator (cmr_agent_amd.utils.synthetic) and the weights are the deterministic hash
fill unless --geo-ckpt / --agent-ckpt point at reference-format state_dicts.  Metrics are the
reference's: RTE / RRE per pair and registration recall (RTE < 5 m and RRE < 10 deg, :198)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL on this host driver): read at HSA init, so set before any GPU call
os.environ.setdefault("ROC_CPU_WAIT_FOR_SIGNAL", "1")        # HIP runtime: cross-queue waits resolved on the host; replayed registration / agent update - 2 to - 3 % (bench.py, profiles/r06_ab_cpuwait.txt)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cmr_agent_amd.dataset.sampling import hip_fps, hip_nearest  # noqa: E402
from cmr_agent_amd.config import KittiConfiguration, NuScenesConfiguration  # noqa: E402
from cmr_agent_amd.environment import environment as env  # noqa: E402
from cmr_agent_amd.models import CMRAgent, MultiHeadModel  # noqa: E402
from cmr_agent_amd.utils import hashfill, synthetic  # noqa: E402
from cmr_agent_amd.utils.checkpoint import load_checked  # noqa: E402


def get_P_diff(P_pred, P_gt):
    """Test_Agent.py:99-105 (scipy Euler 'XYZ' in degrees, summed absolute angles)."""
    from scipy.spatial.transform import Rotation
    r = Rotation.from_matrix(np.dot(P_pred[0:3, 0:3], P_gt[0:3, 0:3].T)).as_euler('XYZ', degrees=True)
    return np.linalg.norm(P_pred[0:3, 3] - P_gt[0:3, 3]), np.sum(np.abs(r))


def main():
    ap = argparse.ArgumentParser(description='Image to point Registration (MI355X HIP path)')
    ap.add_argument('--dataset', type=str, default='kitti', help=" 'kitti' or 'nuscenes' ")
    ap.add_argument('--pairs', type=int, default=4, help="number of synthetic (image, cloud) pairs")
    ap.add_argument('--num-pt', type=int, default=None)
    ap.add_argument('--img', type=str, default=None, help="HxW network input size (multiples of 32), default from the config")
    ap.add_argument('--geo-ckpt', default=None)
    ap.add_argument('--agent-ckpt', default=None)
    ap.add_argument('--data-root', default=None, help="dataset root in the reference's on-disk layout (cmr_agent_amd/dataset/loader.py): the 'test' split; "
                    "default: the synthetic generator")
    args = ap.parse_args()
    dev = torch.device("cuda")
    Cfg = {"kitti": KittiConfiguration, "nuscenes": NuScenesConfiguration}[args.dataset]
    kw = {}
    if args.img:
        kw["cropped_img_H"], kw["cropped_img_W"] = (int(v) for v in args.img.lower().split("x"))
    config = Cfg(num_pt=args.num_pt, device=dev, data_root=args.data_root, **kw)
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "specs.json")))
    geo_model, agent = MultiHeadModel(config), CMRAgent(config)
    load_checked(geo_model, torch.load(args.geo_ckpt) if args.geo_ckpt else hashfill.make_state_dict(spec["geo"], "geo4/"))
    load_checked(agent, torch.load(args.agent_ckpt) if args.agent_ckpt else hashfill.make_state_dict(spec["agent"], "agent/"))
    geo_model, agent = geo_model.to(dev).eval(), agent.to(dev).eval()

    rte, rre = [], []
    with torch.no_grad():
        if args.data_root:
            from cmr_agent_amd.dataset import FrameDataset, FrameLoader
            import itertools
            frames = itertools.islice(iter(FrameLoader(FrameDataset(args.data_root, config, 'test', device=dev), 1, shuffle=False)), args.pairs)
        else:
            frames = (synthetic.make_batch(1, config.num_pt, config.cropped_img_H, config.cropped_img_W, config.num_node,
                                           hip_fps(dev), hip_nearest(dev), seed=config.seed + i, n_circle=16, device=dev) for i in range(args.pairs))
        for data in frames:                                          # batch_size = 1 like the reference loader (:125)
            geo_model(data)
            pose_source, pose_target = env.init(data)
            pose_target = env.to_disentangled(pose_target, data['pc'], data=data)
            for _ in range(config.action_num):
                s2, s3 = env.observation_from_a_pose(data, pose_source, materialize_state_2d=False)
                r_logits, t_logits, _ = agent(s2, s3)
                action_r, action_t = agent.action_from_logits(r_logits, t_logits, deterministic=True)
                pose_source = env.step(action_r, action_t, pose_source, config)
            t_diff, r_diff = get_P_diff(pose_source[0].cpu().numpy(), pose_target[0].cpu().numpy())
            print(t_diff, r_diff)
            rte.append(t_diff)
            rre.append(r_diff)
    rte, rre = np.array(rte), np.array(rre)
    mask = (rte < 5) & (rre < 10)
    print("Registration Recall:", mask.sum() / mask.shape[0])
    if mask.any():
        print('RTE Mean:', rte[mask].mean(), 'RTE Std:', rte[mask].std())
        print('RRE Mean:', rre[mask].mean(), 'RRE Std:', rre[mask].std())


if __name__ == '__main__':
    main()
