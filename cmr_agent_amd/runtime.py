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
SEGMENTED_GRAPH = __import__("os").environ.get("CMR_SEGMENTED_GRAPH_REG", "0") == "1"


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
        if SEGMENTED_GRAPH:
            # a program of single-chain graphs on several streams instead of one graph that holds every branch (utils/seggraph.py)
            from .utils.seggraph import SegmentedGraph
            self.graph = SegmentedGraph()
            with torch.no_grad():
                self.static_pose, self.static_data = self.graph.capture(self._iteration)
        else:
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.static_pose, self.static_data = self._iteration()

    def _iteration(self):
        data = dict(self.static_in)
        self.geo(data)
        pose, target = env.init(data)
        env.to_disentangled(target, data['pc'], data=data)
        for _ in range(self.cfg.action_num):
            s2, s3 = env.observation_from_a_pose(data, pose, materialize_state_2d=False)
            r, t, v = self.agent(s2, s3)
            ar, at = self.agent.action_from_logits(r, t, deterministic=True)
            pose = env.step(ar, at, pose, self.cfg)
        self.static_last = (r, t, v)            # logits / value of the last agent step (static tensors of the graph)
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


class _Geo:            # the one attribute environment._ObsContext reads from data['_cmr']['geo']
    def __init__(self, pc4):
        self.pc4 = pc4


class PipelinedRegistrationGraph:
    """Two-stage software pipeline over CONSECUTIVE batches, one hipGraph: replay i runs the geo model on batch i and, concurrently
    on a second stream, the action_num agent steps of batch i - 1 (whose geo outputs the previous replay left in stable buffers);
    after both have finished the fresh geo outputs are copied into the stable buffers (~ 60 MB, device to device).

    Why: the two phases load the chip differently.  The geo forward is long MFMA-bound convolutions; the agent loop is ten serial
    steps whose tail (22x76 / 11x38 maps, the heads, the observation kernels, the 3-D branch) fills a fraction of the 256 CUs.  Run
    back to back each phase leaves the other's resource idle; batches are independent, so the pipeline changes no
    result -- every batch still goes through exactly Test_Agent.py:150-170 -- only WHEN its two phases run.

    The agent loop is the fork's MAIN branch: it stays on the capture's origin stream, so its per-step 2-D / 3-D fork is captured as real
    graph branches; the geo stage is the side branch and runs its own forks sequentially (a fork issued from a forked stream cannot be
    captured on this ROCm build; utils/streams.sequential_forks makes that explicit).

    Per replay the device does one geo forward and one agent loop = the work of one registration step; `run()` returns the final
    pose of the PREVIOUS batch (pipeline depth 2: call `flush()` for the last one).  Throughput, not latency: the latency of one
    batch is that of RegistrationGraph (geo + loop back to back) or a little more."""

    STABLE_KEYS = ("pc", "K", "P")

    # Measured alternative, off: one hipGraph PER STAGE replayed on two streams, so that each stage keeps its own fork_join branches.
    # 382.7 it/s fp32 / 703.9 bf16 = the unpipelined rate, against 408.4 / 804.9 for the single graph: two graph launches on two
    # streams do not overlap on this runtime (CMR_PIPE_TWO_GRAPHS=1 to repeat the measurement).
    TWO_GRAPHS = __import__("os").environ.get("CMR_PIPE_TWO_GRAPHS", "0") == "1"

    def __init__(self, geo_model, agent, config, example_batch, warmup=2):
        self.geo, self.agent, self.cfg = geo_model, agent, config
        self.static_in = {k: example_batch[k].clone() for k in INPUT_KEYS if k in example_batch}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            data = dict(self.static_in)
            self.geo(data)                                  # prime: batch "-1" = the example batch
            self.stable = self._snapshot(data)
            for _ in range(warmup):
                if self.TWO_GRAPHS:
                    self._geo_stage(); self._agent_loop()
                else:
                    self._iteration()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.TWO_GRAPHS:
            # A fork issued from a forked stream cannot be captured on this ROCm build, and both stages fork inside
            # (image / point tower, heads; the agent's 2-D / 3-D branches).  So each stage is its OWN graph, captured at nesting depth
            # 0, and the two graphs are replayed on two streams per step; the hand-over copies follow on the caller's stream.
            self.s_geo, self.s_agent = torch.cuda.Stream(), torch.cuda.Stream()
            self.graph_geo, self.graph_agent = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph_geo):
                self._fresh = self._agent_inputs(self._geo_stage())
            with torch.no_grad(), torch.cuda.graph(self.graph_agent):
                self.static_pose = self._agent_loop()
        else:
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.static_pose = self._iteration()

    @staticmethod
    def _agent_inputs(data):
        cl = data['_cmr']
        return {"pc": data['pc'], "K": data['K'], "P": data['P'], "pc_overlap_pred": data['pc_overlap_pred'],
                "pc4": cl['geo'].pc4, "pc_geo_feat": cl['pc_geo_feat'], "img_geo_feat": cl['img_geo_feat']}

    def _snapshot(self, data):
        return {k: v.clone() for k, v in self._agent_inputs(data).items()}

    def _stable_view(self):
        s = self.stable
        return {"pc": s["pc"], "K": s["K"], "P": s["P"], "pc_overlap_pred": s["pc_overlap_pred"],
                "_cmr": {"geo": _Geo(s["pc4"]), "pc_geo_feat": s["pc_geo_feat"], "img_geo_feat": s["img_geo_feat"]}}

    def _agent_loop(self):
        data = self._stable_view()
        pose, target = env.init(data)
        env.to_disentangled(target, data['pc'], data=data)
        for _ in range(self.cfg.action_num):
            s2, s3 = env.observation_from_a_pose(data, pose, materialize_state_2d=False)
            r, t, v = self.agent(s2, s3)
            ar, at = self.agent.action_from_logits(r, t, deterministic=True)
            pose = env.step(ar, at, pose, self.cfg)
        self.static_last = (r, t, v)            # logits / value of the last agent step of the batch this stage worked on
        return pose

    def _geo_stage(self):
        data = dict(self.static_in)
        self.geo(data)
        return data

    # CUs the persistent convolution kernels of each stage may occupy (0 = all): with both stages free to take every CU the two
    # streams alternate kernel by kernel; a split lets a convolution of one stage run beside the other's
    BUDGET = tuple(int(v) for v in __import__("os").environ.get("CMR_PIPE_BUDGET", "0,0").split(","))

    def _with_budget(self, cus, fn):
        from . import ops
        def run():
            with ops.conv_cu_budget(cus):
                return fn()
        return run

    def _geo_side(self):
        # the geo stage is the SIDE branch of the pipeline's fork: its own forks (towers, fuse, heads) cannot be captured from there
        # (an edge between two non-origin streams) and run sequentially -- announced, not silent
        from .utils.streams import sequential_forks
        with sequential_forks():
            return self._geo_stage()

    def _iteration(self):
        from .utils.streams import fork_join
        # MAIN branch = the agent loop: it stays on the capture's origin stream, so its per-step 2-D / 3-D fork is captured as real graph
        # branches (round 3 serialised every fork inside both stages); the geo stage is the side branch
        data, pose = fork_join(self._with_budget(self.BUDGET[0], self._geo_side), self._with_budget(self.BUDGET[1], self._agent_loop),
                               tag="pipeline")
        for k, v in self._agent_inputs(data).items():       # hand batch i over to the next replay's agent stage
            self.stable[k].copy_(v)
        return pose

    def run(self, batch=None):
        """Replays the pipeline step: geo(batch) + agent loop of the previously submitted batch.  Returns the static pose tensor
        [B,4,4] of that PREVIOUS batch (overwritten by the next run)."""
        if batch is not None:
            for k, buf in self.static_in.items():
                if batch[k].data_ptr() != buf.data_ptr():
                    buf.copy_(batch[k], non_blocking=True)
        self._replay()
        return self.static_pose

    def _replay(self):
        if not self.TWO_GRAPHS:
            self.graph.replay()
            return
        main = torch.cuda.current_stream()
        self.s_geo.wait_stream(main)
        self.s_agent.wait_stream(main)
        with torch.cuda.stream(self.s_geo):
            self.graph_geo.replay()
        with torch.cuda.stream(self.s_agent):
            self.graph_agent.replay()
        main.wait_stream(self.s_geo)
        main.wait_stream(self.s_agent)
        with torch.no_grad():
            for k, v in self._fresh.items():                # hand batch i over to the next replay's agent stage
                self.stable[k].copy_(v)

    def flush(self):
        """One more replay so that the last submitted batch's agent loop runs; returns its pose (the geo stage re-runs the last inputs)."""
        self._replay()
        return self.static_pose
