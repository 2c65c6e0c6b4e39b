"""Gradient cosines of one agent update in bf16 mode at the benchmark shape (minibatch 10, 88x304, 16 384 points) against the fp32 oracle,
with the weight gradients on the bf16 cores (ops.WGRAD_BF16, default) and in fp32."""
import json, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases as C, golden_util as G
from cmr_agent_amd import ops
from cmr_agent_amd.models import CMRAgent
from cmr_agent_amd.train import AgentUpdate
from cmr_agent_amd.utils import hashfill
from cmr_agent_amd.utils.checkpoint import load_checked
from oracle import train_oracle as TO

def main():
    dev = "cuda"
    specs = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
    case = "agent_train_full"
    cfg_d, cfg_c = C.train_config(case, device=dev), C.train_config(case)
    batch = C.train_inputs(case)[0]
    sd0 = {k: v for k, v in hashfill.make_state_dict(specs["agent"], C.AGENT_TAG).items() if not k.endswith("num_batches_tracked")}
    with torch.enable_grad():
        ol, og, _ = TO.agent_forward_backward({k: x.clone() for k, x in sd0.items()}, batch, cfg_c, True)
    gmax = max(float(g.norm()) for g in og.values())
    for label, conv, wg in (("fp32", False, False), ("bf16 convolutions, fp32 weight gradients", True, False), ("bf16 convolutions and weight gradients", True, True)):
        agent = CMRAgent(cfg_d)
        load_checked(agent, hashfill.make_state_dict(specs["agent"], C.AGENT_TAG))
        up = AgentUpdate(agent.to(dev), cfg_d)
        ops.CONV_BF16, ops.WGRAD_BF16 = conv, wg
        try:
            up.forward_backward({k: x.to(dev) for k, x in batch.items()})
            torch.cuda.synchronize()
        finally:
            ops.CONV_BF16, ops.WGRAD_BF16 = False, True
        grads = up.bucket.logical_grads()
        rows = []
        for k, ref in og.items():
            if float(ref.norm()) < 1e-3 * gmax:
                continue
            rows.append((float(F.cosine_similarity(grads[k].cpu().double().reshape(1, -1), ref.double().reshape(1, -1))), k))
        rows.sort()
        print(label, ": worst five", ["%s %.5f" % (k, c) for c, k in rows[:5]])

main()
