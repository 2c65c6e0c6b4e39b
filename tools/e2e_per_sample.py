"""Per-sample error of the headline-shape iteration at B = 8 against the oracle: which pair carries the largest deviation, and does the same
pair deviate as much when it is run in a batch of 2 (= is it the data, not the batch size)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases as C, parity_e2e

def main():
    from cmr_agent_amd.models import _vit, PointNN, LinearAttention as LA
    for a in sys.argv[1:]:                      # ablations: noqkv | novit | nofront | nola
        if a == "noqkv": _vit.Block.FUSED_QKV = False
        if a == "novit": _vit.Block.FUSED = False
        if a == "nofront": PointNN.FUSED_FRONT = False
        if a == "nola": LA.LinearAttention.FUSED = False
    case = "e2e_config1_b8"
    cfg = C.e2e_config(case)
    geo, agent, geo_sd, agent_sd = parity_e2e.build_models(cfg)
    batch = C.e2e_batch(case)
    got = parity_e2e.run_product(case, geo, agent, batch, cfg)
    ref = C.e2e_oracle(case, geo_sd, agent_sd, batch)
    B = 8
    for k in ("pt_proxy", "node_feat", "pt_feat", "pc_overlap_logits", "fused_node_feat"):
        if k not in got: continue
        g, r = got[k].detach().cpu().double(), ref[k].detach().cpu().double()
        scale = float(r.abs().max())
        per = [(float((g[b] - r[b]).abs().max()) / scale) for b in range(B)]
        print("%-20s scale %8.3f  per-sample max|d| / scale: %s" % (k, scale, " ".join("%.1e" % e for e in per)))

if "knn" not in sys.argv:
    main()


def knn_check():
    """k-NN neighbour sets of the nodes, kernel against oracle, per sample (a near-tie in squared distance is ordered by the last bit of the
    distance, which the host BLAS and the kernel need not share)."""
    from cmr_agent_amd import ops
    from oracle import cmr_oracle as O
    case = "e2e_config1_b8"
    batch = C.e2e_batch(case)
    node = batch["node"]                      # [B,3,M]
    B, _, M = node.shape
    n4 = ops.planar_to_rows4(node.cuda())
    got = ops.knn16(n4, B, M).cpu().long().view(B, M, 16) - torch.arange(B).view(B, 1, 1) * M
    xyz = node.permute(0, 2, 1)
    d = O.square_distance(xyz, xyz)
    ref = d.argsort()[:, :, :16]
    for b in range(B):
        diff = (got[b] != ref[b]).any(1).nonzero().view(-1)
        msg = "sample %d: %d of %d nodes with a different neighbour list" % (b, diff.numel(), M)
        for m in diff[:3].tolist():
            dd = d[b, m]
            msg += "\n   node %d: kernel %s\n            oracle %s\n            distances of the differing entries: %s" % (
                m, got[b, m].tolist(), ref[b, m].tolist(), ["%.9g" % float(dd[j]) for j in sorted(set(got[b, m].tolist()) ^ set(ref[b, m].tolist()))])
        print(msg)


if "knn" in sys.argv:
    knn_check()
