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
    # (label, bf16 mode, bf16 weight gradients, bf16 forward, bf16 backward, earliest data-gradient convolutions kept in fp32)
    for label, conv, wg, fw, bw, early in (("fp32", False, False, True, True, 0),
                                           ("bf16 convolutions, fp32 weight gradients", True, False, True, True, 0),
                                           ("bf16 convolutions and weight gradients", True, True, True, True, 0),
                                           ("bf16 FORWARD only (data and weight gradients fp32)", True, True, True, False, 0),
                                           ("bf16 BACKWARD only (forward fp32)", True, True, False, True, 0),
                                           ("bf16, the two earliest data gradients in fp32", True, True, True, True, 2),
                                           ("bf16, the four earliest data gradients in fp32", True, True, True, True, 4)):
        agent = CMRAgent(cfg_d)
        load_checked(agent, hashfill.make_state_dict(specs["agent"], C.AGENT_TAG))
        up = AgentUpdate(agent.to(dev), cfg_d)
        up.bf16_forward, up.bf16_backward, up.fp32_early_dgrads = fw, bw, early
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
