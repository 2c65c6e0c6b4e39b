"""HIP-graph execution of one registration iteration (geo model + action_num agent steps).

An eager iteration is ~1000 kernel launches of 5-500 us issued from Python through ctypes; once the
kernels are fast the host becomes the limiter.  Shapes are static per configuration, every kernel
of libcmr_hip.so is enqueued on the caller's stream without host synchronisation, and workspaces are
caller-owned, so the whole loop body of Test_Agent.py:150-170 is captured once into a hipGraph and
replayed: inputs are copied into the graph's static buffers, the final pose is read from its static
output.  Eager execution stays available (and is what the parity tests use)."""
import torch

from .environment import environment as env

INPUT_KEYS = ("img", "pc", "node", "pt2node", "K", "P")


class RegistrationGraph:
    def __init__(self, geo_model, agent, config, example_batch, warmup=2):
        self.geo, self.agent, self.cfg = geo_model, agent, config
        self.static_in = {k: example_batch[k].clone() for k in INPUT_KEYS if k in example_batch}
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # builds the plans / sets kernel attributes outside capture
                self._iteration()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_pose, self.static_data = self._iteration()

    def _iteration(self):
        data = dict(self.static_in)
        self.geo(data)
        pose, target = env.init(data)
        env.to_disentangled(target, data['pc'])
        for _ in range(self.cfg.action_num):
            s2, s3 = env.observation_from_a_pose(data, pose)
            r, t, _ = self.agent(s2, s3)
            ar, at = self.agent.action_from_logits(r, t, deterministic=True)
            pose = env.step(ar, at, pose, self.cfg)
        return pose, data

    def run(self, batch=None):
        """Replays the captured iteration; `batch` (same shapes) is copied into the static inputs first.
        Returns the static output pose tensor [B,4,4] (overwritten by the next run)."""
        if batch is not None:
            for k, buf in self.static_in.items():
                if batch[k].data_ptr() != buf.data_ptr():
                    buf.copy_(batch[k], non_blocking=True)
        self.graph.replay()
        return self.static_pose
