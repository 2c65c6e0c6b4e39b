"""Pose-conditioned observation and pose update on HIP kernels.  Function-level mirror of the
reference's environment/environment.py: init :129-140, to_disentangled :14-21,
observation_from_a_pose :24-126, step :179-207 (+ euler_angles_to_matrix :210-260 inside the
kernel).  `expert` / `reward` (:143-176, :263-302) are training-side host logic.

The geo model leaves its row-layout buffers in data['_cmr']; the observation is then two kernel
launches + two memsets per agent step, with no per-sample Python loop and no host sync.  A
sample with zero predicted-overlap points yields an all-zero projected half instead of the
reference's crash (environment.py:74-82)."""
import torch

from .. import ops
from ..models.ImageResNet import to_nhwc
from ..models.PointNN import rows_from_bcl


DIRECT_PROJ = True      # shape-only observations keep the projected half in place (cmr_observation_proj_f32); False: scatter + finalize sweeps


class _ObsContext:
    """Per-batch constants of the agent loop + the reusable scatter / state buffers."""

    def __init__(self, data):
        cl = data.get('_cmr')
        pc = data['pc']
        self.B, _, self.N = pc.shape
        dev = pc.device
        if cl is not None and 'geo' in cl and cl['geo'].pc4.device == dev:
            self.pc4 = cl['geo'].pc4
        else:
            self.pc4 = ops.planar_to_rows(pc.contiguous(), 4)
        self.feat = cl['pc_geo_feat'] if cl is not None and 'pc_geo_feat' in cl else rows_from_bcl(data['pc_geo_feat'])
        ov = data['pc_overlap_pred']
        self.overlap = ov.contiguous().view(torch.uint8).view(-1) if ov.dtype == torch.bool else ov.to(torch.uint8).view(-1)
        self.overlap_src = ov
        self.img = cl['img_geo_feat'] if cl is not None and 'img_geo_feat' in cl else to_nhwc(data['img_geo_feat'])
        _, self.h, self.w, c = self.img.shape
        if c != 64:
            raise ValueError("observation kernels are instantiated for 64-d geometric features")
        self.K = data['K'].to(dev).contiguous()
        cen = data.get('_cmr_centroid')
        self.mean4 = cen[1] if (cen is not None and cen[0] is pc) else _centroid(pc, self.pc4)
        # scatter-mean accumulators of the materialised path: created (zeroed once) on first use, re-zeroed sparsely by every
        # observation_finalize(clear=True); the inference loops never touch them
        self.acc = self.cnt = None
        self.first = True
        # state of the direct projected-half path (materialize_state_2d=False): the map itself, its counts and every point's cell of the
        # previous observation; created on first use
        self.proj = self.proj_cnt = self.proj_cell = None
        self.agent_cache = {}     # CMRAgent keeps the image half of its first conv here (constant over the steps of THIS registration)
        self.dirty = False        # True between a scatter and its finalize: an interrupted call must not leak accumulators


def _context(data):
    ctx = data.get('_cmr_obs')
    if ctx is None or ctx.overlap_src is not data['pc_overlap_pred'] or ctx.pc4.device != data['pc'].device:
        ctx = _ObsContext(data)
        data['_cmr_obs'] = ctx
    return ctx


def _centroid(pcd, pc4=None):
    """Per-sample centroid rows [B,4] of a cloud [B,3(+),N]."""
    B, _, N = pcd.shape
    if pc4 is None:
        pc4 = ops.planar_to_rows(pcd[:, 0:3, :].contiguous(), 4)
    return ops.colmean(pc4, B, N)


@torch.no_grad()
def to_disentangled(poses, pcd, data=None):
    """t <- t - mu + R mu with mu the centroid of the cloud; mutates and returns `poses`.  data (optional, the batch dict `pcd` belongs
    to): the centroid is left there for the observations of the same registration, which need it too (and the row layout of the
    cloud the geo model left there is used instead of converting it again)."""
    pc4 = None
    if data is not None and data.get('pc') is pcd:
        cl = data.get('_cmr')
        if cl is not None and 'geo' in cl and cl['geo'].pc4.device == pcd.device:
            pc4 = cl['geo'].pc4
    mean4 = _centroid(pcd, pc4)
    if data is not None and data.get('pc') is pcd:
        data['_cmr_centroid'] = (pcd, mean4)
    ops.to_disentangled(poses, mean4)
    return poses


@torch.no_grad()
def observation_from_a_pose(data, RT, materialize_state_2d=True):
    """-> (state_2d [B,128,h,w], state_3d [B,5,N]) as views of channels-last / row storage.
    New storage is returned on every call (the reference's replay buffer keeps them).
    materialize_state_2d=False (inference loops whose only consumer is CMRAgent): state_2d is returned as a shape-only `meta`
    tensor carrying the two halves it stands for (image features | projected point features) in `_cmr_split`, which is what
    CMRAgent.forward convolves; the concatenated 128-channel map (110 of the 275 MB this function moves per step at the headline
    shape) is then never written.  Anything that needs the values must ask for the materialised observation (the default)."""
    ctx = _context(data)
    B, N, h, w = ctx.B, ctx.N, ctx.h, ctx.w
    dev = ctx.pc4.device
    state3d = torch.empty((B * N, 8), dtype=torch.float32, device=dev)
    if not materialize_state_2d and DIRECT_PROJ:
        # only the projected half is wanted: it lives in the context and is updated per point (the cells of the previous observation are
        # zeroed, the new ones receive feat / count) instead of being rebuilt by two sweeps over the whole map.  The returned halves alias
        # the context's storage: valid until the next observation of this batch -- which is all CMRAgent.forward needs.
        if ctx.proj is None:
            ctx.proj = torch.zeros((B, h, w, 64), dtype=torch.float32, device=dev)
            ctx.proj_cnt = torch.zeros((B * h * w,), dtype=torch.float32, device=dev)
            ctx.proj_cell = torch.full((B * N,), -1, dtype=torch.int32, device=dev)
        ops.observation_proj(ctx.pc4, ctx.feat, ctx.overlap, RT.contiguous(), ctx.K, ctx.mean4, B, N, h, w, ctx.proj, ctx.proj_cnt,
                             ctx.proj_cell, state3d)
        obs2d = torch.empty((B, 128, h, w), dtype=torch.float32, device="meta")
        obs2d._cmr_split = (ctx.img, ctx.proj, ctx.agent_cache)
        return obs2d, state3d.view(B, N, 8)[:, :, :5].permute(0, 2, 1)
    state2d = torch.empty((B, h, w, 128), dtype=torch.float32, device=dev) if materialize_state_2d else None
    proj = torch.empty((B, h, w, 64), dtype=torch.float32, device=dev)
    if ctx.acc is None:
        ctx.acc = torch.zeros((B * h * w, 64), dtype=torch.float32, device=dev)
        ctx.cnt = torch.zeros((B * h * w,), dtype=torch.float32, device=dev)
    zero_first, ctx.dirty = ctx.dirty, True
    ops.project_scatter(ctx.pc4, ctx.feat, ctx.overlap, RT.contiguous(), ctx.K, ctx.mean4, B, N, h, w, ctx.acc, ctx.cnt,
                        state3d, zero_first=zero_first)
    ops.observation_finalize(ctx.img, ctx.acc, ctx.cnt, state2d, proj, B, h, w, True, clear=True)
    ctx.dirty = False
    obs2d = state2d.permute(0, 3, 1, 2) if state2d is not None else torch.empty((B, 128, h, w), dtype=torch.float32, device="meta")
    # the agent's first conv is linear in its input: hand it the two halves separately so that the image half
    # (constant over the steps of one registration) is convolved once (CMRAgent.forward_cl)
    obs2d._cmr_split = (ctx.img, proj, ctx.agent_cache)
    obs3d = state3d.view(B, N, 8)[:, :, :5].permute(0, 2, 1)
    return obs2d, obs3d


_EYES = {}


def _identity(B, dev):
    key = (B, str(dev))
    if key not in _EYES:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return torch.eye(4, device=dev).repeat(B, 1, 1)      # never cache a tensor that lives in a graph's private pool
        _EYES[key] = torch.eye(4, device=dev).repeat(B, 1, 1)
    return _EYES[key]


def init(data):
    """Identity start pose and the ground-truth pose (for the expert)."""
    dev = data['pc'].device
    B = data['pc'].shape[0]
    pose_target = data['P'].to(dev)
    pose_source = _identity(B, dev).clone()                  # one copy launch instead of eye + repeat (3)
    return pose_source, pose_target


def step(action_r, action_t, pose_source, config):
    """pose <- [E_xyz(delta_r) R | t + delta_t]; mutates and returns pose_source."""
    dev = pose_source.device
    r_steps, t_steps = config.r_steps, config.t_steps
    if r_steps.device != dev:
        r_steps, t_steps = r_steps.to(dev), t_steps.to(dev)
    ops.pose_step(pose_source, action_r.contiguous(), action_t.contiguous(), r_steps, t_steps, config.is_6_DoF)
    return pose_source


@torch.no_grad()
def expert(pose_source, targets, config, data=None):
    """Expert action of the current state (reference environment.py:143-176), on the device: the reference leaves the
    GPU for scipy's Euler decomposition at every step.  Returns (action_r, action_t) int64 like the reference."""
    dev = pose_source.device
    r_steps, t_steps = config.r_steps, config.t_steps
    if r_steps.device != dev:
        r_steps, t_steps = r_steps.to(dev), t_steps.to(dev)
    return ops.expert_action(pose_source.contiguous(), targets.to(dev).contiguous(), r_steps, t_steps, config.is_6_DoF)


@torch.no_grad()
def reward(RT, data, prev_distance=None):
    """Dense step reward (reference environment.py:263-302).  RT is accepted and ignored exactly as in the reference
    (the transformed cloud is commented out there, :276).  Returns (reward [B,1,1], p2p_distance [B,1,1])."""
    pc = data['pc']
    dev = pc.device
    cam = data['pc_in_cam_space'].to(dev).contiguous()
    mask = data['pc_mask'].to(dev).to(torch.int64).contiguous()
    prev = None if prev_distance is None else prev_distance.to(dev).reshape(-1).contiguous()
    rew, dist = ops.reward(pc.contiguous(), cam, mask, prev)
    return rew.view(-1, 1, 1), dist.view(-1, 1, 1)
