"""GPU tier, end to end: geo model + agent loop through the HIP path vs oracle and golden."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["e2e_small", "e2e_native"])
def test_registration_iteration(case):
    import parity_e2e
    parity_e2e.run_case(case, check_golden=True, verbose=True)


def test_registration_iteration_baseline_config0():
    """BASELINE.json configs[0] sizes (batch 1, 4096 points, 176x608 -> 192x608, 1 agent step) against the oracle."""
    import parity_e2e
    parity_e2e.run_case("e2e_config0", check_golden=False, verbose=True)


def test_registration_iteration_headline_config1():
    """BASELINE.json configs[1] -- the shape the bench line is quoted on (352x1216, 16 384 points, 10 agent steps) --
    at B = 2 against the oracle, same tolerances as the small cases (parity_e2e.compare).  The oracle needs a few
    seconds per sample on the host."""
    import parity_e2e
    parity_e2e.run_case("e2e_config1", check_golden=False, verbose=True)


def test_registration_iteration_headline_shape_at_the_benchmark_batch():
    """BASELINE.json configs[1] exactly as bench.py runs it: 8 pairs of 352x1216 / 16 384 points, 10 agent steps, against the oracle
    (features, losses, per-step logits / values, actions and poses), at the tolerances of the small cases -- with one allowance: pair 5 of
    this batch has 8 of its 1 280 nodes on which the third pixel-to-node linear-attention layer is ill-conditioned (a LayerNorm over a
    nearly constant message row): the fp32 ORACLE is 2.1e-4 of the feature scale away from its own float64 evaluation on exactly those
    rows (tests/test_conditioning_cpu.py pins that), and so is any fp32 path.  The allowance is restricted to exactly that: in PAIR 5 at
    most 16 node rows of `fused_node_feat` may exceed the 1e-4 tolerance (none above 1e-3 of the scale), and the per-point outputs of the
    heads may exceed it only at points of pair 5 ASSIGNED to such a node (parity_e2e.ill_conditioned_rows); every other pair, node and
    point, the image side, every agent step's logits, actions and poses meet the usual bars (tools/e2e_per_sample.py,
    tools/decoder_trace.py)."""
    import parity_e2e
    spec = dict(pair=5, node_key="fused_node_feat", point_keys=("pc_overlap_logits", "pc_geo_feat", "pc_is_in_cam_scores"), max_nodes=16, hard=1e-3)
    parity_e2e.run_case("e2e_config1_b8", check_golden=False, verbose=True, ill_conditioned=spec)


def test_registration_iteration_nuscenes_config3_shape():
    """BASELINE.json configs[3] shape (896x1600 next to 900x1600, 32 768 points; NuScenesConfig), B = 1, 2 agent steps,
    against the oracle."""
    import parity_e2e
    parity_e2e.run_case("e2e_config3", check_golden=False, verbose=True)


def test_pipelined_graph_gives_every_batch_the_unpipelined_result():
    """cmr_agent_amd.runtime.PipelinedRegistrationGraph runs geo(batch i) beside the agent loop of batch i - 1 in one hipGraph: run(batch)
    must return, one call later, what RegistrationGraph returns for that batch (same kernels in the same order per batch), for
    alternating different batches, and flush() must deliver the last one."""
    import cases as C
    import parity_e2e
    from cmr_agent_amd.runtime import PipelinedRegistrationGraph, RegistrationGraph
    from cmr_agent_amd.utils import synthetic
    from oracle import cmr_oracle as O
    case = "e2e_small"
    c = C.E2E_CASES[case]
    cfg = C.e2e_config(case)
    cfg.r_steps, cfg.t_steps = cfg.r_steps.cuda(), cfg.t_steps.cuda()      # step tables on the device: no host copy inside the capture
    geo, agent, _, _ = parity_e2e.build_models(cfg)
    batches = [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in
                synthetic.make_batch(c["B"], c["N"], c["H"], c["W"], c["M"], O.dataset_fps, O.nearest_node, seed=sd, n_circle=c["n_circle"]).items()}
               for sd in (2023, 7, 99)]
    snap = lambda g: [g.static_pose.clone()] + [x.clone() for x in g.static_last]
    with torch.no_grad():
        plain = RegistrationGraph(geo, agent, cfg, batches[0])
        want = []
        for b in batches:
            plain.run(b)
            want.append(snap(plain))
        # the hash-filled agent's actions hardly depend on the pair, its logits / value do: that is what tells the batches apart
        assert not torch.equal(want[0][1], want[1][1]) and not torch.equal(want[1][3], want[2][3])
        pipe = PipelinedRegistrationGraph(geo, agent, cfg, batches[0])
        pipe.run(batches[0])                                   # submits batch 0 (returns the priming batch's pose)
        pipe.run(batches[1]); got0 = snap(pipe)                # agent loop of batch 0 || geo of batch 1
        pipe.run(batches[2]); got1 = snap(pipe)
        pipe.flush(); got2 = snap(pipe)
    torch.cuda.synchronize()
    # poses (products of the discrete action tables) are equal to the bit; logits / value agree to rounding only -- the observation's
    # scatter-mean uses float atomics, whose order differs from launch to launch even within one graph -- and each must match ITS batch
    scale = max(float(w_[1].abs().max()) for w_ in want)
    for i, g in enumerate((got0, got1, got2)):
        assert torch.equal(g[0], want[i][0])
        for j in range(3):
            err = max(float((x - y).abs().max()) for x, y in zip(g[1:], want[j][1:]))
            assert (err <= 1e-4 * scale) == (i == j), (i, j, err, scale)


def test_shape_only_observation_gives_the_materialised_result():
    """environment.observation_from_a_pose(materialize_state_2d=False) -- what the inference loops use -- never writes the concatenated
    128-channel map: the agent must produce the same logits / value / actions from the two halves as from the materialised observation
    (to rounding: the scatter-mean uses float atomics), the shape-only tensor must keep the reference's shape, and using it without
    its halves must fail loudly instead of reading null."""
    import cases as C
    import parity_e2e
    from cmr_agent_amd.environment import environment as env
    case = "e2e_small"
    cfg = C.e2e_config(case)
    geo, agent, _, _ = parity_e2e.build_models(cfg)
    data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in C.e2e_batch(case).items()}
    with torch.no_grad():
        geo(data)
        pose, _ = env.init(data)
        pose[:, 0, 3] += 0.3
        s2, s3 = env.observation_from_a_pose(data, pose)
        r0, t0, v0 = agent(s2, s3)
        m2, m3 = env.observation_from_a_pose(data, pose, materialize_state_2d=False)
        assert m2.device.type == "meta" and tuple(m2.shape) == tuple(s2.shape)
        r1, t1, v1 = agent(m2, m3)
        for a, b in ((r0, r1), (t0, t1), (v0, v1)):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
        assert torch.equal(agent.action_from_logits(r0, t0, True)[0], agent.action_from_logits(r1, t1, True)[0])
        assert torch.equal(s3, m3)
        bare = torch.empty(tuple(s2.shape), device="meta")
        with pytest.raises(ValueError):
            agent(bare, m3)


def test_in_place_projected_observation_over_a_pose_sequence():
    """cmr_observation_proj_f32 keeps the projected half of the observation in place and only touches the cells of the previous and of
    the current pose: over a sequence of poses (cells get vacated, re-entered, shared by several points; one pose throws every point out
    of view) the map must equal the scatter + finalize result of each pose -- to rounding of the mean (sum of pre-divided terms, float
    atomics) -- vacated cells exactly zero, and state_3d bit-identical."""
    import cases as C
    import parity_e2e
    from cmr_agent_amd.environment import environment as env
    case = "e2e_small"
    cfg = C.e2e_config(case)
    geo, agent, _, _ = parity_e2e.build_models(cfg)
    data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in C.e2e_batch(case).items()}
    with torch.no_grad():
        geo(data)
        pose0, _ = env.init(data)
        shifts = [(0.0, 0.0, 0.0), (0.4, 0.0, 0.0), (0.4, -0.3, 0.2), (0.0, 0.0, -1.0e4), (0.0, 0.0, 0.0), (-0.5, 0.2, 0.0), (-0.5, 0.2, 0.0)]
        for k, (dx, dy, dz) in enumerate(shifts):
            pose = pose0.clone()
            pose[:, 0, 3] += dx; pose[:, 1, 3] += dy; pose[:, 2, 3] += dz
            s2, s3 = env.observation_from_a_pose(data, pose)                                  # scatter + finalize, materialised
            m2, m3 = env.observation_from_a_pose(data, pose, materialize_state_2d=False)      # in place
            want = s2[:, 64:].permute(0, 2, 3, 1)
            got = m2._cmr_split[1]
            assert torch.equal(s3, m3), k
            assert torch.equal(got == 0, want == 0), k                                        # the same cells are occupied, the others exactly zero
            assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())), k
            if k == 3:
                assert float(got.abs().max()) == 0.0                                          # everything behind the camera: the map is empty again


def test_registration_iteration_op_level_paths(monkeypatch):
    """Same iteration with every layer-level fusion and the side streams switched off: the op-level composition
    must meet the same oracle / golden bars."""
    import parity_e2e
    from cmr_agent_amd.models import LinearAttention as LA, PointNN, _vit, CMRAgent
    from cmr_agent_amd.utils import streams
    monkeypatch.setattr(LA.LinearAttention, "FUSED", False)
    monkeypatch.setattr(PointNN, "FUSED_FRONT", False)
    monkeypatch.setattr(_vit.Block, "FUSED", False)
    monkeypatch.setattr(CMRAgent, "FUSED_TAIL", False)
    monkeypatch.setattr(streams, "ENABLED", False)
    parity_e2e.run_case("e2e_small", check_golden=True, verbose=False)


def test_agent_image_half_cache_is_per_registration():
    """the image half of the agent's first convolution used to be cached under (data_ptr, version,
    shape) of img_geo_feat; a second registration whose buffer landed at the same address reused the previous pair's
    result.  Two same-shape observations with different image features through ONE agent, the first dict dropped in
    between, must give what a fresh agent gives.  (Built at the environment level: with the hash-filled geo model the
    agent's logits hardly depend on the input image, so whole-pipeline batches cannot tell a stale cache.)"""
    import gc
    import parity_e2e
    import cases as C
    from cmr_agent_amd.environment import environment as env
    cfg = C.e2e_config("e2e_small")
    _, agent, _, _ = parity_e2e.build_models(cfg)
    _, agent2, _, _ = parity_e2e.build_models(cfg)
    B, N, h, w = 2, 1024, cfg.image_H, cfg.image_W
    g = torch.Generator().manual_seed(3)
    common = dict(pc=torch.rand(B, 3, N, generator=g) * 20, K=torch.tensor([[0.6 * w, 0, w / 2], [0, 0.6 * w, h / 2], [0, 0, 1.0]]).repeat(B, 1, 1),
                  pc_overlap_pred=torch.rand(B, N, generator=g) > 0.5, pc_geo_feat=torch.nn.functional.normalize(torch.rand(B, 64, N, generator=g) - 0.5, dim=1))

    def logits(ag, seed):
        gg = torch.Generator().manual_seed(seed)
        data = {k: v.to("cuda") for k, v in common.items()}
        # a seed-specific mean direction + noise: the 2-D branch ends in a global average pool, so two iid-noise maps
        # with the same statistics would give the same logits
        img = (torch.rand(1, 64, 1, 1, generator=gg) - 0.5) * 2 + 0.3 * (torch.rand(B, 64, h, w, generator=gg) - 0.5)
        data["img_geo_feat"] = torch.nn.functional.normalize(img, dim=1).to("cuda")
        with torch.no_grad():
            s2, s3 = env.observation_from_a_pose(data, torch.eye(4, device="cuda").repeat(B, 1, 1))
            r, t, v = ag(s2, s3)
        return torch.cat([r.flatten(), t.flatten(), v.flatten()]).cpu()

    l1 = logits(agent, 1)
    gc.collect()                                                             # the first dict is gone: its buffers return to the allocator
    l2 = logits(agent, 2)
    want = logits(agent2, 2)
    scale = float(want.abs().max())
    tol = 1e-5 * scale                 # not bit-exact: the projection scatter-mean accumulates with float atomics
    # the image half moves these (hash-filled) logits only slightly, but well above the tolerance: a stale cache would
    # leave l2 at l1's image contribution
    assert float((l1 - l2).abs().max()) > 5 * tol
    assert float((l2 - want).abs().max()) < tol, float((l2 - want).abs().max())


def test_entry_point_scripts_run_end_to_end(tmp_path):
    """Test_Agent.py and Train_Agent.py (the reference's entry points) on small synthetic pairs: the
    evaluation prints a recall, the training loop reaches an agent update (num_trajectory = 4 batches) with finite
    losses and writes a checkpoint with the reference's state_dict keys."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "Test_Agent.py"), "--pairs", "2", "--num-pt", "4096"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Registration Recall:" in r.stdout
    out = tmp_path / "ckpt"
    r = subprocess.run([sys.executable, os.path.join(root, "Train_Agent.py"), "--batches", "4", "--img", "96x160", "--num-pt", "2048",
                        "--batch-size", "2", "--out", str(out)], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    logs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    upd = [l for l in logs if "train_loss/BC_Loss" in l]
    assert len(upd) == 1 and upd[0]["minibatches"] == 8                 # 4 trajectories x 10 steps x 2 pairs / 10
    assert all(abs(upd[0][k]) < 1e4 for k in ("train_loss/BC_Loss", "train_loss/PPO_Loss"))
    assert upd[0]["update_mode"] == "hipGraph"
    # the same run with every launch issued from Python: the captured update (minibatches gathered straight into the graph's input
    # buffers) is the same kernels in the same order.  Two runs of EITHER mode differ in the 5th digit (the rollouts' scatter uses
    # atomics), so the comparison is to 2e-3, far below what a stale or aliased input / loss buffer would show (1.97 vs 2.78 when
    # step() still returned the graph's own loss buffer)
    r2 = subprocess.run([sys.executable, os.path.join(root, "Train_Agent.py"), "--batches", "4", "--img", "96x160", "--num-pt", "2048",
                         "--batch-size", "2", "--eager", "--out", str(tmp_path / "ckpt_eager")], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r2.returncode == 0, r2.stderr[-2000:]
    upd2 = [l for l in (json.loads(l) for l in r2.stdout.splitlines() if l.startswith("{")) if "train_loss/BC_Loss" in l]
    assert len(upd2) == 1 and upd2[0]["update_mode"] == "eager"
    for k in ("train_loss/BC_Loss", "train_loss/PPO_Loss", "train_loss/reward"):
        assert abs(upd2[0][k] - upd[0][k]) <= 2e-3 * max(1.0, abs(upd[0][k])), (k, upd[0][k], upd2[0][k])
    ck = [f for f in os.listdir(out) if f.endswith(".pth")]
    assert ck
    sd = torch.load(os.path.join(out, ck[0]), map_location="cpu")
    spec = json.load(open(os.path.join(root, "tests", "golden", "specs.json")))["agent"]
    assert set(sd) == set(spec) and all(list(sd[k].shape) == spec[k] for k in spec)


def test_train_geo_entry_point_runs_end_to_end(tmp_path):
    """Train_Geo.py on small synthetic batches: validation in eval mode through the inference path, six
    optimizer steps through GeoUpdate with finite losses, validation losses that move with the weights, checkpoints with the
    reference's state_dict keys."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    out = tmp_path / "ckpt"
    r = subprocess.run([sys.executable, os.path.join(root, "Train_Geo.py"), "--batches", "3", "--epochs", "2", "--img", "96x160", "--num-pt", "2048",
                        "--batch-size", "2", "--val-interval", "3", "--out", str(out)], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    logs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    train = [l for l in logs if "train_loss/loss" in l]
    val = [l for l in logs if "val_loss/loss" in l]
    assert len(train) == 6 and len(val) == 2
    assert all(l["train_loss/loss"] == l["train_loss/loss"] and abs(l["train_loss/loss"]) < 1e4 for l in train)
    assert val[1]["val_loss/loss"] != val[0]["val_loss/loss"]
    ck = [f for f in os.listdir(out) if f.endswith(".pth")]
    assert len(ck) == 2
    sd = torch.load(os.path.join(out, ck[0]), map_location="cpu")
    spec = json.load(open(os.path.join(root, "tests", "golden", "specs.json")))["geo"]
    want = {k for k in spec if not k.endswith("num_batches_tracked")}
    assert want <= set(sd) and all(list(sd[k].shape) == spec[k] for k in want if not k.endswith("position_embeddings"))


def test_entry_points_with_the_module_api_and_on_disk_frames(tmp_path):
    """The three entry points on frames READ FROM DISK in the reference's layout (`--data-root`, cmr_agent_amd/dataset/loader.py) and, for the
    two training scripts, through the nn.Module boundary (`--module-api`: model(data); data['loss'].backward(); torch.optim step /
    the Train_Agent minibatch body composed in torch -- cmr_agent_amd/train/bridge.py)."""
    import json
    import os
    import subprocess
    import sys
    from test_loader import _write_dataset
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.path.join(root, "tests"))
    data = tmp_path / "kitti"
    _write_dataset(str(data), seqs=(0, 9), frames=8, with_image_3=False, n_raw=6000, img_hw=(200, 340))     # half size 100 x 170 >= the 96 x 160 crop
    # (num_pt >= 8 num_node = 10 240: the node candidates are drawn without replacement, KittiDataset.py:356; the 6 000-point clouds are tiled, :186-190)
    common = ["--img", "96x160", "--num-pt", "10240", "--batch-size", "2", "--data-root", str(data)]
    out = tmp_path / "geo"
    r = subprocess.run([sys.executable, os.path.join(root, "Train_Geo.py"), "--batches", "2", "--epochs", "2", "--val-interval", "2", "--module-api",
                        "--out", str(out)] + common, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    logs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    train = [l for l in logs if "train_loss/loss" in l]
    val = [l for l in logs if "val_loss/loss" in l]
    assert len(train) == 4 and len(val) == 2
    assert all(l["train_loss/loss"] == l["train_loss/loss"] and abs(l["train_loss/loss"]) < 1e4 for l in train)
    assert val[1]["val_loss/loss"] != val[0]["val_loss/loss"]                     # torch.optim moved the weights the inference path reads
    assert "8 samples in train set..." in r.stdout and "8 samples in val set..." in r.stdout
    out = tmp_path / "agent"
    r = subprocess.run([sys.executable, os.path.join(root, "Train_Agent.py"), "--batches", "4", "--module-api", "--out", str(out)] + common,
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    logs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    upd = [l for l in logs if "train_loss/BC_Loss" in l]
    assert len(upd) == 1 and upd[0]["minibatches"] == 8
    assert all(abs(upd[0][k]) < 1e4 for k in ("train_loss/BC_Loss", "train_loss/PPO_Loss"))
    r = subprocess.run([sys.executable, os.path.join(root, "Test_Agent.py"), "--pairs", "2", "--img", "96x160", "--num-pt", "10240", "--data-root", str(data)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Registration Recall:" in r.stdout


@pytest.mark.parametrize("shape", ["kitti_headline", "small_six_dof"])
def test_agent_tail_on_transposed_weights(shape):
    """cmr_agent_heads_t_f32 (weights stored [in][out4]: a layer is one memory round trip, no cross-lane reduction) against the
    row-per-wave kernel and a float64 torch restatement of CMRAgent.py:52-56, 101-116: global pool, two 1x1 convs, the three heads, and the
    deterministic actions (CMRAgent.py:118-123) -- same logits to fp32 rounding of the sums, identical actions."""
    import cases as C
    from cmr_agent_amd import ops
    from cmr_agent_amd.config import KittiConfiguration
    from cmr_agent_amd.models import CMRAgent
    if shape == "kitti_headline":
        cfg, B = KittiConfiguration(cropped_img_H=352, cropped_img_W=1216, num_pt=1024, device="cuda"), 8
    else:
        cfg, B = KittiConfiguration(cropped_img_H=96, cropped_img_W=160, num_pt=1024, device="cuda", is_6_DoF=True), 3
    torch.manual_seed(5)
    agent = CMRAgent(cfg).to("cuda").eval()
    p = agent.plan()
    assert p["tail_t"]
    kh, kw = cfg.image_H // 8, cfg.image_W // 8
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(B * kh * kw, 128, generator=g) - 0.3).to("cuda")
    e3d = (torch.rand(B, 128, generator=g) - 0.5).to("cuda")
    names = ("policy_r", "policy_t", "value")
    acts = (cfg.num_steps, agent.degree_r, agent.degree_t)
    out_t, (ar_t, at_t) = ops.agent_heads_t(x, B, kh * kw, p["c24t"], p["c26t"], e3d, [p[n + "_t"] for n in names], 0.01, actions=acts)
    out_r, (ar_r, at_r) = ops.agent_heads(x, B, kh * kw, p["c24"], p["c26"], e3d, [p[n] for n in names], 0.01, actions=acts)
    torch.cuda.synchronize()
    lr = lambda v: torch.where(v > 0, v, 0.01 * v)
    xm = x.double().view(B, kh * kw, 128).mean(1)
    e2 = lr(xm @ p["c24"][0].double().t() + p["c24"][1].double()) @ p["c26"][0].double().t() + p["c26"][1].double()
    st = torch.cat([e2, e3d.double()], 1)
    for i, n in enumerate(names):
        (w0, b0), (w1, b1), (w2, b2) = p[n]
        want = lr(lr(st @ w0.double().t() + b0.double()) @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
        scale = float(want.abs().max())
        assert float((out_t[i].double() - want).abs().max()) <= 2e-6 * max(1.0, scale), (n, float((out_t[i].double() - want).abs().max()))
        assert float((out_t[i] - out_r[i]).abs().max()) <= 2e-6 * max(1.0, scale), n
    assert torch.equal(ar_t, ar_r) and torch.equal(at_t, at_r)
    S = cfg.num_steps
    assert torch.equal(ar_t, out_t[0][:, :agent.degree_r * S].view(B, agent.degree_r, S).argmax(-1))
