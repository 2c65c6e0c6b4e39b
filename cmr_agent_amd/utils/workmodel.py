"""Algorithmic work of every C-ABI entry point on the registration path: FLOPs = 2 x MACs of the math the
call implements, bytes = each distinct input and output once, weights once (fp32).  A call is MFMA-class when
FLOPs / bytes exceeds the fp32 ridge (157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B) and its ideal time is FLOPs / peak, else
HBM-class with ideal time bytes / bandwidth.  bench.py times every call with HIP events (CallTimer) and reports
roofline.path = sum(ideal) / sum(measured) over the whole path, plus the per-entry-point table.

The numbers are ALGORITHMIC, not what the kernels issue: Winograd issues 16/36 of a convolution's multiplies, the fused
layer kernels recompute small GEMMs, the transposed-GEMM chains pad K; none of that is credited."""
import torch

from .. import _lib

FP32_MFMA_PEAK = 157.3e12      # FLOP/s, MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32, 256 CUs at 2.4 GHz)
HBM_PEAK = 8.0e12              # B/s
RIDGE = FP32_MFMA_PEAK / HBM_PEAK
F = 4                          # bytes per float


def _linear(a):
    rows, k1, n = a["rows"], a["k1"], a["n_out"]
    k2 = a["k2"] if a["x2"] else 0
    src2 = 0
    if a["x2"]:
        src2 = rows * k2 if a["idx2"] else max(rows // max(a["div2"], 1), 1) * k2
    by = F * (rows * k1 + src2 + rows * n + n * (k1 + k2) + (rows * n if a["res"] and not a["res_mod"] else 0))
    return 2.0 * rows * (k1 + k2) * n, by


def _cbr(a):
    rows, kx, ch, co, k1 = a["rows"], a["kx"], a["ch"], a["co"], a["k1"]
    w = kx * ch + ch * co + (kx * co if a["wsc"] else 0)
    src2 = 0
    if a["x2"]:
        src2 = rows * (kx - k1) if a["idx2"] else max(rows // max(a["div2"], 1), 1) * (kx - k1)
    return 2.0 * rows * w, F * (rows * k1 + src2 + (rows * co if a["y"] else 0) + w)


def _conv(a, stride=None):
    s = a.get("stride", 1) if stride is None else stride
    B, H, W, ci, co, pool = a["B"], a["H"], a["W"], a["Cin"], a["Cout"], a.get("pool", 1)
    opix = B * ((H - 1) // s + 1) * ((W - 1) // s + 1)
    by = F * (B * H * W * ci + opix * co / (pool * pool) + (opix * co if a["res"] else 0) + 9 * ci * co)
    return 2.0 * 9 * ci * co * opix, by


def _conv_io(a):
    s = a.get("stride", 1)
    B, H, W, ci, co, pool = a["B"], a["H"], a["W"], a["Cin"], a["Cout"], a.get("pool", 1)
    opix = B * ((H - 1) // s + 1) * ((W - 1) // s + 1)
    by = ((2 if a["x_bf16"] else F) * B * H * W * ci + (2 if a["y_bf16"] else F) * opix * co / (pool * pool)
          + (2 if a.get("res_bf16") else F) * (opix * co if a["res"] else 0) + F * 9 * ci * co)
    return 2.0 * 9 * ci * co * opix, by


def _heads(a):
    w = 2 * 128 * 128 + sum(256 * a[p + "_n0"] + a[p + "_n0"] * a[p + "_n1"] + a[p + "_n1"] * a[p + "_n2"] for p in "rtv")
    return 2.0 * a["B"] * w, F * (a["B"] * a["npix"] * 128 + w)


def _vecattn(a):
    rows = a["rows"]
    if a["feat"]:
        return rows * 2.0 * (3 * 4096 + 192 + 4096 + 2 * 4096), rows * (256 + 256 + 32 + 512)
    return rows * 2.0 * (192 + 4096 + 2 * 4096), rows * (512 + 256 + 32 + 512)


WORK = {
    "cmr_linear_f32": _linear,
    "cmr_cbr_block_f32": _cbr,
    "cmr_cbr_block_bf16_f32": _cbr,
    "cmr_colmax_partials_f32": lambda a: (0, F * a["B"] * a["tiles_per_batch"] * a["C"]),
    "cmr_colmax_bias2_f32": lambda a: (2.0 * a["B"] * a["C"] * (a["n1"] + a["n2"]),
                                       F * (a["B"] * a["tiles_per_batch"] * a["C"] + (a["C"] + 1 + a["B"]) * (a["n1"] + a["n2"]))),
    "cmr_layernorm64_f32": lambda a: (0, F * a["rows"] * 64 * (3 if a["res"] else 2)),
    "cmr_conv3x3_nhwc_f32": _conv,
    "cmr_conv3x3_wino_nhwc_f32": lambda a: _conv(a, 1),
    # the same launch with the BatchNorm sums from its epilogue (no residual / pool; the partial sums are a few KB)
    "cmr_conv3x3_wino_stats_nhwc_f32": lambda a: (2.0 * 9 * a["Cin"] * a["Cout"] * a["B"] * a["H"] * a["W"],
                                                  F * (a["B"] * a["H"] * a["W"] * (a["Cin"] + a["Cout"]) + 9 * a["Cin"] * a["Cout"])),
    "cmr_bn_stats_from_sums_f32": lambda a: (0, F * a["parts"] * 2 * a["C"]),
    # data gradient + the BatchNorm-backward sums: reads dy and the BatchNorm input, writes dx
    "cmr_conv3x3_wino_bnbwd_nhwc_f32": lambda a: (2.0 * 9 * a["Cin"] * a["Cout"] * a["B"] * a["H"] * a["W"],
                                                  F * (a["B"] * a["H"] * a["W"] * (a["Cin"] + 2 * a["Cout"]) + 9 * a["Cin"] * a["Cout"])),
    "cmr_bn_bwd_from_sums_f32": lambda a: (0, F * a["rows"] * a["C"] * 3),
    "cmr_conv3x3_s2_nhwc_f32": lambda a: _conv(a, 2),
    "cmr_conv3x3_bf16_nhwc_f32": _conv,
    "cmr_conv3x3_bf16io_nhwc": _conv_io,
    # BatchNorm + LeakyReLU in the staging pass: the BatchNorm input in, the convolution's output out (the activated map is never in memory)
    "cmr_conv3x3_bf16_pro_nhwc_f32": lambda a: (2.0 * 9 * a["Cin"] * a["Cout"] * a["B"] * a["H"] * a["W"],
                                                F * (a["B"] * a["H"] * a["W"] * (a["Cin"] + a["Cout"]) + 9 * a["Cin"] * a["Cout"])),
    # ResidualBlock(3 -> 64): conv3x3 3->3, conv3x3 3->64, 1x1 shortcut 3->64
    "cmr_stem_block_f32": lambda a: (2.0 * (81 + 1728 + 192) * a["B"] * a["H"] * a["W"], a["B"] * a["H"] * a["W"] * (F * 3 + (2 if a.get("out_bf16") else F) * 64)),
    "cmr_avgpool_nhwc_f32": lambda a: (0, F * a["B"] * a["H"] * a["W"] * a["C"] * (1 + 1.0 / (a["kh"] * a["kw"]))),
    "cmr_upsample_concat_f32": lambda a: (0, F * a["B"] * a["H"] * a["W"] * (2 * a["C1"] + a["C2"] * (1 + 1.0 / a["scale"] ** 2))),
    "cmr_patchify_nhwc_f32": lambda a: (0, 2 * F * a["B"] * a["H"] * a["W"] * a["C"]),
    "cmr_transpose_f32": lambda a: (0, 2 * F * a["batch"] * a["R"] * a["Cn"]),
    "cmr_linear_rows_bf16_f32": lambda a: (2.0 * a["rows"] * a["k"] * a["n_out"], F * (a["rows"] * (a["k"] + a["n_out"] * (2 if a["res"] else 1)) + a["k"] * a["n_out"])),
    "cmr_patch_embed_f32": lambda a: (2.0 * a["B"] * (a["H"] // a["P"]) * (a["W"] // a["P"]) * a["P"] * a["P"] * a["C"] * a["n_out"],
                                     F * (a["B"] * a["H"] * a["W"] * a["C"] + a["B"] * (a["H"] // a["P"]) * (a["W"] // a["P"]) * a["n_out"] + a["P"] * a["P"] * a["C"] * a["n_out"])),
    "cmr_mha_f32": lambda a: (4.0 * a["B"] * a["Tq"] * a["Tk"] * 64, F * 64 * a["B"] * (2 * a["Tq"] + 2 * a["Tk"])),
    "cmr_mha_expf_f32": lambda a: (4.0 * a["B"] * a["Tq"] * a["Tk"] * 64, F * 64 * a["B"] * (2 * a["Tq"] + 2 * a["Tk"])),
    # attention + the projections (64 -> 64 for the queries, 64 -> 128 for the source rows; algorithmic: once per row, not per workgroup)
    "cmr_mha_ln_f32": lambda a: (4.0 * a["B"] * a["Tq"] * a["Tk"] * 64 + 2.0 * 64 * a["B"] * (64 * a["Tq"] + 128 * a["Tk"]),
                                F * 64 * a["B"] * (2 * a["Tq"] + a["Tk"])),
    "cmr_la_reduce_f32": lambda a: (2.0 * a["B"] * a["S"] * 576, F * a["B"] * a["S"] * 128),
    "cmr_la_apply_f32": lambda a: (2.0 * a["B"] * a["L"] * 576, F * a["B"] * a["L"] * 128),
    "cmr_agent_heads_f32": _heads,
    "cmr_agent_heads_train_f32": _heads,
    "cmr_ln64_linear_f32": lambda a: (2.0 * 64 * (a["rows_x"] * a["n_out_x"] + (a["rows_y"] * a["n_out_y"] if a["y"] else 0)),
                                      F * (a["rows_x"] * (64 + a["n_out_x"]) + (a["rows_y"] * (64 + a["n_out_y"]) if a["y"] else 0)
                                           + 64 * (a["n_out_x"] + (a["n_out_y"] if a["y"] else 0)))),
    "cmr_ln64_linear_bf16_f32": lambda a: (2.0 * 64 * (a["rows_x"] * a["n_out_x"] + (a["rows_y"] * a["n_out_y"] if a["y"] else 0)),
                                      F * (a["rows_x"] * (64 + a["n_out_x"]) + (a["rows_y"] * (64 + a["n_out_y"]) if a["y"] else 0)
                                           + 64 * (a["n_out_x"] + (a["n_out_y"] if a["y"] else 0)))),
    "cmr_vit_out_ffn_f32": lambda a: (2.0 * a["rows"] * (4096 + 2 * 65536), F * (a["rows"] * 192 + 4096 + 2 * 65536)),
    "cmr_vit_out_ffn16_f32": lambda a: (2.0 * a["rows"] * (4096 + 2 * 65536), F * (a["rows"] * 192 + 4096 + 2 * 65536)),
    "cmr_vit_out_ffn_bf16_f32": lambda a: (2.0 * a["rows"] * (4096 + 2 * 65536), F * (a["rows"] * 192 + 4096 + 2 * 65536)),
    # LinearAttention.py:46-60: k / v projections + per-head 8x8 state (source side); q projection, application, merge,
    # MLP 128 -> 128 -> 64 (query side): 17.5 + 66.7 = 84 kFLOP per token pair, as SURVEY.md 8a row a10 counts
    "cmr_la_kv_state_f32": lambda a: (2.0 * a["B"] * a["S"] * (2 * 4096 + 576), F * (a["B"] * a["S"] * 64 + 2 * 4096)),
    "cmr_la_kv_state_bf16_f32": lambda a: (2.0 * a["B"] * a["S"] * (2 * 4096 + 576), F * (a["B"] * a["S"] * 64 + 2 * 4096)),
    "cmr_la_query_layer_f32": lambda a: (2.0 * a["B"] * a["L"] * (4096 + 576 + 4096 + 16384 + 8192),
                                         F * (a["B"] * a["L"] * 128 + 2 * 4096 + 16384 + 8192)),
    "cmr_la_query_layer_bf16_f32": lambda a: (2.0 * a["B"] * a["L"] * (4096 + 576 + 4096 + 16384 + 8192),
                                         F * (a["B"] * a["L"] * 128 + 2 * 4096 + 16384 + 8192)),
    "cmr_vecattn_front_f32": _vecattn,
    "cmr_focal_metrics_f32": lambda a: (0, a["rows"] * 16),
    "cmr_circle_loss_f32": lambda a: (2.0 * a["B"] * a["n"] * a["n"] * 64, F * a["B"] * a["n"] * 128),
    "cmr_planar_to_rows_f32": lambda a: (0, F * a["B"] * a["N"] * (a["C"] + a["Cpad"])),
    "cmr_concat_rows_f32": lambda a: (0, F * a["rows"] * (2 * a["C1"] + a["C2"])),
    "cmr_index_to_global_i32": lambda a: (0, 12 * a["B"] * a["N"]),
    "cmr_csr_build_i32": lambda a: (0, 8 * a["B"] * a["n_per_batch"] + 8 * a["B"] * a["seg_per_batch"]),
    "cmr_knn16_f32": lambda a: (8.0 * a["B"] * a["M"] * a["M"], a["B"] * a["M"] * (16 + 64)),
    "cmr_nearest_f32": lambda a: (8.0 * a["B"] * a["Nq"] * a["Nc"], a["B"] * (a["Nq"] * 28 + a["Nc"] * 16)),
    "cmr_rel_pos_f32": lambda a: (0, a["rows"] * 48),
    "cmr_vecattn_prep_f32": lambda a: (0, F * a["rows"] * 64 * 6),
    "cmr_segment_softmax_f32": lambda a: (0, F * a["_rows"] * 128 + F * a["nseg"] * 64),
    "cmr_segment_reduce_f32": lambda a: (0, F * a["_rows"] * a["C"] + F * a["nseg"] * a["C"]),
    "cmr_gather_rows_f32": lambda a: (0, a["rows"] * (2 * F * a["C"] + 4)),
    "cmr_colmax_f32": lambda a: (0, F * a["B"] * a["N"] * a["C"]),
    "cmr_colmean_f32": lambda a: (0, F * a["B"] * a["N"] * a["C"]),
    # environment.py:24-126: per point xyz0 + mask + 64-d feature in, 8-float state row out; per cell image feature +
    # accumulator + count in, [img | mean] + projected half out
    "cmr_project_scatter_f32": lambda a: (0, a["B"] * a["N"] * (16 + 1 + 256 + 32)),
    "cmr_observation_finalize_f32": lambda a: (0, a["B"] * a["h"] * a["w"] * (256 + 256 + 4 + 512 + 256)),
    "cmr_pose_step_f32": lambda a: (0, a["B"] * 128),
    "cmr_to_disentangled_f32": lambda a: (0, a["B"] * 128),
    "cmr_argmax_rows_f32": lambda a: (0, a["outer"] * a["inner"] * (F * a["n"] + 8)),
    "cmr_softmax2_f32": lambda a: (0, a["rows"] * 14),
    "cmr_l2norm64_f32": lambda a: (0, a["rows"] * 512),
    # training direction (Train_Agent.py:296-305, Train_Geo.py:166-174): weight gradients as GEMMs over the minibatch's pixels / rows
    "cmr_conv3x3_wgrad_f32": lambda a: (2.0 * 9 * a["Cin"] * a["Cout"] * a["B"] * a["H"] * a["W"],
                                        F * (a["B"] * a["H"] * a["W"] * (a["Cin"] + a["Cout"]) + 9 * a["Cin"] * a["Cout"])),
    # the same gradient in the Winograd domain (algorithmic work of the direct sum; the kernel issues 16/36 of it: ISSUED)
    "cmr_conv3x3_wgrad_wino_f32": lambda a: (2.0 * 9 * a["Cin"] * a["Cout"] * a["B"] * a["H"] * a["W"],
                                             F * (a["B"] * a["H"] * a["W"] * (a["Cin"] + a["Cout"]) + 9 * a["Cin"] * a["Cout"])),
    "cmr_conv3x3_wgrad_s2_f32": lambda a: (2.0 * 9 * a["Cin"] * a["Cout"] * a["B"] * (a["H"] // 2) * (a["W"] // 2),
                                           F * (a["B"] * a["H"] * a["W"] * (a["Cin"] + a["Cout"] / 4.0) + 9 * a["Cin"] * a["Cout"])),
    "cmr_conv3x3_wgrad_bf16_f32": lambda a: (2.0 * 9 * a["Cin"] * a["Cout"] * a["B"] * a["H"] * a["W"],
                                             F * (a["B"] * a["H"] * a["W"] * (a["Cin"] + a["Cout"]) + 9 * a["Cin"] * a["Cout"])),
    "cmr_conv3x3_wgrad_bias_bf16_f32": lambda a: (2.0 * 9 * a["Cin"] * a["Cout"] * a["B"] * a["H"] * a["W"],
                                                  F * (a["B"] * a["H"] * a["W"] * (a["Cin"] + a["Cout"]) + 9 * a["Cin"] * a["Cout"])),
    "cmr_conv3x3_wgrad_bias_bf16_pro_f32": lambda a: (2.0 * 9 * a["Cin"] * a["Cout"] * a["B"] * a["H"] * a["W"],
                                                      F * (a["B"] * a["H"] * a["W"] * (a["Cin"] + a["Cout"]) + 9 * a["Cin"] * a["Cout"])),
    "cmr_linear_bwd_rows_f32": lambda a: (4.0 * a["rows"] * a["n"] * a["k"], F * (a["rows"] * (3 * a["n"] + 3 * a["k"]) + 2 * a["n"] * a["k"])),
    "cmr_linear_wgrad_f32": lambda a: (2.0 * a["rows"] * a["n"] * a["k"], F * (a["rows"] * (a["n"] + a["k"]) + a["n"] * a["k"])),
    "cmr_adam_f32": lambda a: (0, 28 * a["n"]),
    "cmr_sgd_f32": lambda a: (0, 20 * a["n"]),
    # ---- every other entry point of the two training steps (round 4: `path` covers all kernel time).  Streaming passes: each distinct input
    # and output once, no FLOPs credited (HBM-class); the fused layer kernels: the arithmetic of the layer as the reference writes it
    "cmr_bn_stats_f32": lambda a: (0, F * a["rows"] * a["C"]),
    "cmr_affine_act_f32": lambda a: (0, F * a["rows"] * a["C"] * (2 + (1 if a["res"] else 0))),
    "cmr_bn_bwd_f32": lambda a: (0, F * a["rows"] * a["C"] * (3 + (1 if a["z"] else 0) + (1 if a["add"] else 0) + (1 if a["dzm"] else 0))),
    "cmr_linear_bn_fwd_f32": lambda a: (2.0 * a["rows"] * a["n"] * a["k"], F * (a["rows"] * (a["k"] + a["n"]) + a["n"] * a["k"])),
    "cmr_bn_bwd_coef_f32": lambda a: (0, F * a["rows"] * a["C"] * (2 + (1 if a["z"] else 0))),
    # fused layer backward: the two GEMMs of the layer (weight + data gradient), every distinct map once
    "cmr_bn_linear_bwd_f32": lambda a: (2.0 * a["rows"] * a["n"] * a["k"] * (2 if a["dx"] else 1),
                                        F * (a["rows"] * (a["n"] * (1 + (1 if (a["z"] and not a["mask_from_h"]) else 0) + (1 if a["stat"] else 0) + (1 if a["dzm"] else 0)) +
                                                          a["k"] * (1 + (1 if a["res"] else 0) + (1 if a["dx"] else 0))) + 2 * a["n"] * a["k"])),
    "cmr_vecattn_front_train_f32": lambda a: (2.0 * a["rows"] * (8 * 64 + 3 * 64 * 64), F * a["rows"] * (3 * 64 + 8 + 5 * 64)),
    "cmr_vecattn_front_kv_train_f32": lambda a: (2.0 * a["rows"] * (8 * 64 + 6 * 64 * 64), F * a["rows"] * (2 * 64 + 8 + 6 * 64)),
    "cmr_vecattn_mix_f32": lambda a: (0, 6 * F * a["rows"] * a["C"]),
    "cmr_vecattn_mix_bwd_f32": lambda a: (0, 4 * F * a["rows"] * a["C"]),
    "cmr_act_bwd_f32": lambda a: (0, F * a["rows"] * a["C"] * (3 + (1 if a["add"] else 0))),
    "cmr_pool_act_bwd_f32": lambda a: (0, F * a["B"] * a["H"] * a["W"] * a["C"] * (2 + 1.0 / (a["ph"] * a["pw"]))),
    "cmr_colsum_f32": lambda a: (0, F * a["B"] * a["N"] * a["C"]),
    "cmr_colmax_arg_f32": lambda a: (0, F * a["B"] * a["N"] * a["C"]),
    "cmr_add_at_arg_f32": lambda a: (0, 12 * a["B"] * a["C"]),
    "cmr_linear_bwd_small_f32": lambda a: (4.0 * a["rows"] * a["n"] * (a["k1"] + a["k2"]),
                                           F * (a["rows"] * (2 * a["n"] + 2 * (a["k1"] + a["k2"])) + 2 * a["n"] * (a["k1"] + a["k2"]))),
    "cmr_agent_loss_f32": lambda a: (0, 8 * a["B"] * (a["dr"] + a["dt"]) * a["S"]),
    "cmr_dropout_f32": lambda a: (0, 2 * F * a["rows"] * a["C"]),
    "cmr_act_f32": lambda a: (0, 2 * F * a["rows"] * a["C"]),
    "cmr_act_bwd_x_f32": lambda a: (0, 3 * F * a["rows"] * a["C"]),
    "cmr_axpy_f32": lambda a: (0, 3 * F * a["rows"] * a["C"]),
    "cmr_layernorm64_bwd_f32": lambda a: (0, 3 * F * a["rows"] * 64),
    "cmr_l2norm64_bwd_f32": lambda a: (0, 3 * F * a["rows"] * 64),
    "cmr_segment_softmax_bwd_f32": lambda a: (0, F * (a["_rows"] * 64 * 4 + a["nseg"] * 64)),
    # softmax attention with / without dropout on the probabilities; backward: S recomputed, dP, dV, dQ, dK = 5 products
    "cmr_mha_dropout_f32": lambda a: (4.0 * a["B"] * a["Tq"] * a["Tk"] * 64, F * 64 * a["B"] * (2 * a["Tq"] + 2 * a["Tk"])),
    "cmr_mha_bwd_f32": lambda a: (10.0 * a["B"] * a["Tq"] * a["Tk"] * 64, F * 64 * a["B"] * (4 * a["Tq"] + 4 * a["Tk"])),
    "cmr_mha_dropout_bwd_f32": lambda a: (10.0 * a["B"] * a["Tq"] * a["Tk"] * 64, F * 64 * a["B"] * (4 * a["Tq"] + 4 * a["Tk"])),
    "cmr_la_bwd_f32": lambda a: (2.0 * 576 * a["B"] * (3 * a["L"] + 2 * a["S"]), F * 64 * a["B"] * (3 * a["L"] + 4 * a["S"])),
    "cmr_circle_loss_bwd_f32": lambda a: (4.0 * a["B"] * a["n"] * a["n"] * 64, 2 * F * a["B"] * a["n"] * 128),
    "cmr_focal_bwd_f32": lambda a: (0, a["rows"] * 32),
    "cmr_pack_conv3x3_f32": lambda a: (0, F * a["Cout"] * a["Cin"] * 34),
    "cmr_pack_conv3x3_slots_f32": lambda a: (0, F * a.get("_pairs", 0) * 34),
    "cmr_pack_frags_f32": lambda a: (0, 2 * F * a.get("_elems", 0)),
    "cmr_transpose_slots_f32": lambda a: (0, 2 * F * 1024 * a["total_tiles"]),
    "cmr_zero_insert2_f32": lambda a: (0, F * a["B"] * a["C"] * (a["Ho"] * a["Wo"] + a["H"] * a["W"])),
    "cmr_im2col3_f32": lambda a: (0, F * a["B"] * a["H"] * a["W"] * 40),
    "cmr_col2im3_f32": lambda a: (0, F * a["B"] * a["H"] * a["W"] * 40),
    "cmr_upsample_bwd_f32": lambda a: (0, F * a["B"] * a["H"] * a["W"] * a["C2"] * (1 + 1.0 / a["scale"] ** 2)),
    "cmr_patchify_bwd_f32": lambda a: (0, 2 * F * a["B"] * a["H"] * a["W"] * a["C"]),
    # fused train-mode transformer block (csrc/vit_train.hip): out-projection + MLP forward; its backward (fc1 recomputed, two MLP data
    # gradients, the out-projection's); LayerNorm + projections backward; grouped weight gradients
    "cmr_vit_out_ffn16_train_f32": lambda a: (2.0 * a["rows"] * (4096 + 2 * 65536), F * (a["rows"] * 256 + 4096 + 2 * 65536)),
    "cmr_vit_ffn_bwd16_f32": lambda a: (2.0 * a["rows"] * (3 * 65536 + 4096), F * (a["rows"] * (7 * 64 + 2048) + 3 * 65536 + 4096)),
    "cmr_vit_lnqkv_bwd_f32": lambda a: (2.0 * 64 * (a["rows_x"] * a["k_x"] + (a["rows_y"] * a["k_y"] if a["d_y"] else 0)),
                                        F * (a["rows_x"] * (a["k_x"] + 192 + (64 if a["res"] else 0)) + (a["rows_y"] * (a["k_y"] + 192) if a["d_y"] else 0))),
    "cmr_wgrad_group_f32": lambda a: (sum(2.0 * r * n * k for r, n, k in a.get("_group", ())), F * sum(r * (n + k) + n * k for r, n, k in a.get("_group", ()))),
    # fused train-mode linear-attention layer (la_fused.hip train instances, la_train.hip)
    "cmr_la_kv_state_train_f32": lambda a: (2.0 * a["B"] * a["S"] * (2 * 4096 + 576), F * (a["B"] * a["S"] * 192 + 2 * 4096)),
    "cmr_la_query_layer_train_f32": lambda a: (2.0 * a["B"] * a["L"] * (4096 + 576 + 4096 + 16384 + 8192), F * (a["B"] * a["L"] * (128 + 448) + 2 * 4096 + 16384 + 8192)),
    "cmr_la_mlp_bwd_f32": lambda a: (2.0 * a["rows"] * (8192 + 16384 + 4096), F * (a["rows"] * 704 + 8192 + 16384 + 4096)),
    "cmr_la_proj_bwd_f32": lambda a: (2.0 * a["_rows"] * 4096, F * a["_rows"] * 64 * 3),
    # dataset-side point work: latency-class; priced on the bytes they must move (cloud once, result once)
    "cmr_fps_f32": lambda a: (0, a["B"] * (16 * a["N"] + 8 * a["npoint"])),
    "cmr_fps_ws_f32": lambda a: (0, a["B"] * (16 * a["N"] + 8 * a["npoint"])),
    "cmr_ball_query_f32": lambda a: (0, a["B"] * (16 * a["N"] + a["S"] * (16 + 8 * a["nsample"]))),
}

# Multiplies the kernel ISSUES on the matrix cores per algorithmic multiply: F(2x2,3x3) Winograd computes 2x2 outputs with 16
# products instead of 36.  `roofline.frac` prices the dominant kernel on issued work (<= 1 by construction); the algorithmic
# figure stays beside it as frac_algorithmic.
ISSUED = {"cmr_conv3x3_wino_nhwc_f32": 16.0 / 36.0, "cmr_conv3x3_wino_stats_nhwc_f32": 16.0 / 36.0, "cmr_conv3x3_wino_bnbwd_nhwc_f32": 16.0 / 36.0,
          "cmr_conv3x3_wgrad_wino_f32": 16.0 / 36.0}


def work(name, args, extra=None):
    """-> (flops, bytes) of one call, or None when the entry point is not modelled."""
    fn = WORK.get(name)
    if fn is None:
        return None
    names = _lib.prototypes()[name][2]
    a = dict(zip(names, args))
    if extra:
        a.update(extra)
    if "_rows" not in a:
        a["_rows"] = a.get("nseg", 0) * max(a.get("fixed_len", 0), 1)
    fl, by = fn(a)
    return float(fl), float(by)


BF16_MFMA_PEAK = 2500e12      # FLOP/s dense, v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md); ridge 312 FLOP/B


def matrix_peak(name):
    """Matrix peak an entry point is priced against: the bf16 peak for the entry points whose products run on the bf16 cores."""
    return BF16_MFMA_PEAK if "bf16" in name else FP32_MFMA_PEAK


def ideal_seconds(flops, nbytes, name=""):
    return max(flops / matrix_peak(name), nbytes / HBM_PEAK)


class CallTimer:
    """HIP-event timing of every C-ABI call on the stream it is issued on (torch's current stream at the call), with the
    algorithmic work of the call.  Use around an EAGER pass (events cannot sit inside a replayed graph)."""

    def __init__(self):
        self.records = []          # (name, e0, e1, flops, bytes)
        self._orig = None

    def __enter__(self):
        self._orig = _lib.call

        def timed(name, *args, allow_unsupported=False, work_extra=None):
            wk = work(name, args, work_extra)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = self._orig(name, *args, allow_unsupported=allow_unsupported)
            e1.record()
            if rc != _lib.UNSUPPORTED:
                self.records.append((name, e0, e1) + (wk if wk is not None else (None, None)))
            return rc
        _lib.call = timed
        return self

    def __exit__(self, *a):
        _lib.call = self._orig

    def table(self):
        """-> list of dicts per entry point, sorted by measured time."""
        agg = {}
        for name, e0, e1, fl, by in self.records:
            ms = e0.elapsed_time(e1)
            d = agg.setdefault(name, dict(name=name, calls=0, ms=0.0, flops=0.0, bytes=0.0, ideal_ms=0.0, issued_flops=0.0,
                                          ideal_issued_ms=0.0, modelled=fl is not None))
            d["calls"] += 1
            d["ms"] += ms
            if fl is not None:
                d["flops"] += fl
                d["bytes"] += by
                d["ideal_ms"] += 1e3 * ideal_seconds(fl, by, name)
                d["issued_flops"] += fl * ISSUED.get(name, 1.0)
                d["ideal_issued_ms"] += 1e3 * ideal_seconds(fl * ISSUED.get(name, 1.0), by, name)
        rows = sorted(agg.values(), key=lambda d: -d["ms"])
        for d in rows:
            d["bound"] = "mfma" if d["bytes"] and d["flops"] / d["bytes"] > matrix_peak(d["name"]) / HBM_PEAK else "hbm"
            d["frac"] = d["ideal_issued_ms"] / d["ms"] if d["ms"] > 0 else 0.0
            d["frac_algorithmic"] = d["ideal_ms"] / d["ms"] if d["ms"] > 0 else 0.0
        return rows
