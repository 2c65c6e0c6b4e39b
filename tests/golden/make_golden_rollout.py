#!/usr/bin/env python3
"""Generates tests/golden/rollout_ops.npz by running the REFERENCE's environment.expert / environment.reward /
buffer.discounted / buffer.advantage (imported from /root/reference on CPU through ref_harness) on
tests/cases.py:rollout_inputs, and cross-checks the oracle while doing so.

Run in the authoring container only:   python tests/golden/make_golden_rollout.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_harness  # noqa: E402
import golden_util as G  # noqa: E402
import cases as C  # noqa: E402
from oracle import cmr_oracle as O  # noqa: E402

torch.set_grad_enabled(False)


def main():
    ns = ref_harness.load_reference()
    inp = C.rollout_inputs()
    named, report = {}, {}
    for six in (False, True):
        cfg = ns.config.KittiConfiguration()
        cfg.is_6_DoF = six
        ar, at = ns.env.expert(inp["pose_source"].clone(), inp["pose_target"].clone(), cfg, None)
        tag = "6dof" if six else "3dof"
        named["expert_r_" + tag], named["expert_t_" + tag] = ar, at
        oar, oat = O.env_expert(inp["pose_source"], inp["pose_target"], cfg.r_steps, cfg.t_steps, six)
        report["expert_" + tag] = [float((oar != ar).float().mean()), float((oat != at).float().mean())]
    data = dict(pc=inp["pc"].clone(), pc_in_cam_space=inp["pc_in_cam_space"], pc_mask=inp["pc_mask"])
    r0, d0 = ns.env.reward(None, data)
    r1, d1 = ns.env.reward(None, data, prev_distance=d0 + torch.tensor([0.5, -0.5, 0.0] * 4).view(-1, 1, 1))
    named.update(reward_first=r0, distance=d0, reward_next=r1)
    o0, od0 = O.env_reward(data)
    o1, _ = O.env_reward(data, prev_distance=d0 + torch.tensor([0.5, -0.5, 0.0] * 4).view(-1, 1, 1))
    report["reward"] = [float((od0 - d0).abs().max()), float((o1 - r1).abs().max()), float((o0 - r0).abs().max())]
    named["returns"] = ns.buffer.discounted(inp["rewards"], 0.99)
    named["advantage_plain"] = ns.buffer.advantage(inp["rewards"], inp["values"], 0.99, 0)
    named["advantage_gae"] = ns.buffer.advantage(inp["rewards"], inp["values"], 0.99, 0.95)
    report["buffer"] = [float((O.discounted(inp["rewards"], 0.99) - named["returns"]).abs().max()),
                        float((O.advantage(inp["rewards"], inp["values"], 0.99, 0) - named["advantage_plain"]).abs().max()),
                        float((O.advantage(inp["rewards"], inp["values"], 0.99, 0.95) - named["advantage_gae"]).abs().max())]
    G.save_case("rollout_ops", named)
    print(json.dumps(report, indent=1))
    rp = os.path.join(G.OUT_DIR, "oracle_vs_reference.json")
    rep = json.load(open(rp))
    rep["rollout_ops"] = {"%s_%d" % (k, i): v for k, vs in report.items() for i, v in enumerate(vs)}   # flat: max |oracle - reference|
    json.dump(rep, open(rp, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
