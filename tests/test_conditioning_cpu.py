"""CPU tier: the conditioning fact behind the one allowance of tests/test_e2e_gpu.py::test_registration_iteration_headline_shape_at_the_
benchmark_batch.  For pair 5 of the e2e_config1_b8 batch, the third pixel-to-node linear-attention layer of the decoder evaluated by the
ORACLE in float32 differs from the same layer evaluated in float64 (same float32 inputs, weights cast) by ~2e-4 of the feature scale on
a handful of node rows and by < 2e-5 on all others: those rows are ill-conditioned for every float32 evaluation, the HIP path included."""
import json
import os

import torch
import torch.nn.functional as F

import cases as C
import golden_util as G
from oracle import cmr_oracle as O


def test_fp32_oracle_is_off_its_float64_self_on_the_same_few_rows():
    case, smp = "e2e_config1_b8", 5
    specs = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
    geo_sd, _ = C.e2e_state_dicts(specs)
    batch, cfg = C.e2e_batch(case), C.e2e_config(case)
    one = {k: (v[smp:smp + 1].clone() if torch.is_tensor(v) and v.shape[0] == 8 else v) for k, v in batch.items()}
    w = O.Weights(geo_sd).sub("encoder_decoder")
    out = O.imgpc_encoder(w.sub("encoder"), dict(one), cfg)
    pt_proxy, n2p = out["pt_proxy"].permute(0, 2, 1), out["node2proxy"]
    f = pt_proxy.shape[1]
    b, n = n2p.shape
    fn = torch.cat([out["node_feat"], torch.gather(pt_proxy, 2, n2p.unsqueeze(1).expand(b, f, n))], 1)
    for i in range(cfg.node_fuse_res_num):
        fn = O.conv_bn_relu_res1d(w.sub("node_fuse_convs.%d" % i), fn)
    f2 = out["img_feat_2"]
    hp, wp = f2.shape[2] // cfg.patch_size, f2.shape[3] // cfg.patch_size
    up = F.interpolate(out["img_proxy"].permute(0, 2, 1).reshape(b, f, hp, wp), scale_factor=cfg.patch_size, mode="nearest")
    fi = torch.cat([f2, up], 1)
    for i in range(cfg.img_fuse_res_num):
        fi = O.residual_block(w.sub("img_fuse_convs.%d" % i), fi, 1)
        if i == 0:
            fi = fi + O.position_encoding_sine_2d(f, f2.shape[2], f2.shape[3])
    pix, nod = fi.view(b, f, -1).permute(0, 2, 1), fn.permute(0, 2, 1)
    for i in range(2):
        nod = O.linear_attention(w.sub("pixel_to_node_LA.%d" % i), nod, pix, cfg.LA_head_num)
        pix = O.linear_attention(w.sub("node_to_pixel_LA.%d" % i), pix, nod, cfg.LA_head_num)
        nod = O.linear_attention(w.sub("node_self_LA.%d" % i), nod, nod, cfg.LA_head_num)
        pix = O.linear_attention(w.sub("pixel_self_LA.%d" % i), pix, pix, cfg.LA_head_num)
    y32 = O.linear_attention(w.sub("pixel_to_node_LA.2"), nod, pix, cfg.LA_head_num)
    sd64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in geo_sd.items()}
    y64 = O.linear_attention(O.Weights(sd64).sub("encoder_decoder").sub("pixel_to_node_LA.2"), nod.double(), pix.double(), cfg.LA_head_num)
    rows = (y32.double() - y64).abs()[0].max(1)[0] / float(y64.abs().max())
    off = int((rows > 1e-4).sum())
    assert 1 <= off <= 16, off                                     # measured: 7 rows at 1.9e-4 .. 2.1e-4
    assert float(rows.max()) < 1e-3
    assert int((rows > 2e-5).sum()) <= 16                          # ... and nothing in between: the other ~1 270 rows are at 1e-6
