"""Stage-by-stage comparison of the decoder's node side (fuse blocks, then every linear-attention layer) between the HIP path and the oracle
for ONE sample of a case: python tools/decoder_trace.py <sample>"""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases as C, golden_util as G, parity_e2e
from oracle import cmr_oracle as O
from cmr_agent_amd.utils.streams import fork_join

def main():
    smp = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    case = "e2e_config1_b8"
    cfg = C.e2e_config(case)
    geo, agent, geo_sd, agent_sd = parity_e2e.build_models(cfg)
    batch = C.e2e_batch(case)
    one = {k: (v[smp:smp + 1].clone() if torch.is_tensor(v) and v.shape[0] == 8 else v) for k, v in batch.items()}
    # ---- oracle, stepwise (mirrors oracle.imgpc_endecoder)
    import torch.nn.functional as F
    ed_w = O.Weights(geo_sd).sub("encoder_decoder")
    out = O.imgpc_encoder(ed_w.sub("encoder"), dict(one), cfg)
    ref = {}
    pt_proxy = out["pt_proxy"].permute(0, 2, 1)
    n2p = out["node2proxy"]
    f = pt_proxy.shape[1]; b, n = n2p.shape
    g = torch.gather(pt_proxy, 2, n2p.unsqueeze(1).expand(b, f, n))
    fn = torch.cat([out["node_feat"], g], dim=1)
    for i in range(cfg.node_fuse_res_num):
        fn = O.conv_bn_relu_res1d(ed_w.sub("node_fuse_convs.%d" % i), fn)
        ref["fuse%d" % i] = fn.permute(0, 2, 1).reshape(-1, fn.shape[1])
    f2 = out["img_feat_2"]
    hp, wp = f2.shape[2] // cfg.patch_size, f2.shape[3] // cfg.patch_size
    up = F.interpolate(out["img_proxy"].permute(0, 2, 1).reshape(b, f, hp, wp), scale_factor=cfg.patch_size, mode="nearest")
    fi = torch.cat([f2, up], dim=1)
    for i in range(cfg.img_fuse_res_num):
        fi = O.residual_block(ed_w.sub("img_fuse_convs.%d" % i), fi, 1)
        if i == 0:
            fi = fi + O.position_encoding_sine_2d(f, f2.shape[2], f2.shape[3])
    pix = fi.view(b, f, -1).permute(0, 2, 1); nod = fn.permute(0, 2, 1)
    for i in range(cfg.linear_attention_num):
        nod = O.linear_attention(ed_w.sub("pixel_to_node_LA.%d" % i), nod, pix, cfg.LA_head_num); ref["p2n%d" % i] = nod.reshape(-1, f)
        pix = O.linear_attention(ed_w.sub("node_to_pixel_LA.%d" % i), pix, nod, cfg.LA_head_num)
        nod = O.linear_attention(ed_w.sub("node_self_LA.%d" % i), nod, nod, cfg.LA_head_num); ref["nself%d" % i] = nod.reshape(-1, f)
        pix = O.linear_attention(ed_w.sub("pixel_self_LA.%d" % i), pix, pix, cfg.LA_head_num)
    # ---- HIP, stepwise (mirrors IMGPCEnDecoder.forward_cl)
    ed = geo.encoder_decoder
    got = {}
    with torch.no_grad():
        d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in one.items()}
        cl = ed.encoder.forward_cl(d)
        B, gg, f2c = cl["B"], cl["geo"], cl["f2"]
        _, h, wd, ff = f2c.shape
        nodc = ed.node_fuse_convs[0].rows(cl["node_feat"], x2=cl["pt_proxy"], idx2=cl["node2proxy_global"]); got["fuse0"] = nodc
        for j, layer in enumerate(list(ed.node_fuse_convs)[1:-1]):
            nodc = layer.rows(nodc); got["fuse%d" % (j + 1)] = nodc
        from cmr_agent_amd import ops
        x = ops.upsample_concat(f2c, cl["img_proxy"], cfg.patch_size)
        for i, layer in enumerate(list(ed.img_fuse_convs)[:-1]):
            x = layer.forward_cl(x, post=ed._pos_table(h, wd, x.device) if i == 0 else None)
        pixc = x.view(B * h * wd, ff)
        M, L = gg.M, h * wd
        for i in range(cfg.linear_attention_num):
            nodc = ed.pixel_to_node_LA[i].rows(nodc, pixc, B, M, L); got["p2n%d" % i] = nodc
            pixc = ed.node_to_pixel_LA[i].rows(pixc, nodc, B, L, M)
            nodc = ed.node_self_LA[i].rows(nodc, nodc, B, M, M); got["nself%d" % i] = nodc
            pixc = ed.pixel_self_LA[i].rows(pixc, pixc, B, L, L)
    for k in ref:
        if k not in got: continue
        g_, r_ = got[k].cpu().double(), ref[k].double()
        e = (g_ - r_).abs()
        row = int(e.max(1)[0].argmax())
        print("%-8s scale %9.3f  max|d|/scale %.2e  (row %d; rows with error > 1e-5 of scale: %d of %d; that row's own max %.3e)" % (
            k, float(r_.abs().max()), float(e.max()) / float(r_.abs().max()), row, int((e.max(1)[0] > 1e-5 * float(r_.abs().max())).sum()), e.shape[0], float(r_[row].abs().max())))

main()
