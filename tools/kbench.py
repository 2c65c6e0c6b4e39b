#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the shapes of BASELINE configs[1] (B=8): HIP-event timing of
single entry points, reported against the fp32 MFMA peak / HBM bandwidth.  Development tool."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, reps=20, warm=3):
    """GPU time per call: the calls are captured into one HIP graph and replayed, so host-side
    launch cost (ctypes + torch, ~10 us per call) does not mask short kernels."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def bench_conv(B, H, W, cin, cout, stride, reps):
    x = torch.randn(B, H, W, cin, device=DEV)
    w = torch.randn(9, cout, cin, device=DEV) / math.sqrt(9 * cin)
    b = torch.randn(cout, device=DEV)
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = torch.randn(B, ho, wo, cout, device=DEV)
    out = torch.empty(B, ho, wo, cout, device=DEV)
    us = timeit(lambda: ops.conv3x3(x, w, b, cout, stride, 0.2, res=res, out=out), reps)
    fl = 2.0 * 9 * cin * cout * B * ho * wo
    by = 4.0 * (x.numel() + 2 * out.numel())
    line = "conv3x3 %4dx%-4d %3d->%-3d s%d : %8.1f us  %6.1f TFLOP/s  (%5.2f TB/s algorithmic)" % (
        H, W, cin, cout, stride, us, fl / us / 1e6, by / us / 1e6)
    if stride == 2 and cin == 64:
        from cmr_agent_amd.models._pack import conv_bf16_frags
        fr = conv_bf16_frags(torch.randn(cout, cin, 3, 3, device=DEV) / math.sqrt(9 * cin))
        if ops.conv3x3_bf16(x, fr, b, cout, 0.2, res=res, stride=2) is not None:
            us3 = timeit(lambda: ops.conv3x3_bf16(x, fr, b, cout, 0.2, res=res, stride=2), reps)
            line += "   | bf16 %8.1f us  %6.1f TFLOP/s  %5.2f TB/s" % (us3, fl / us3 / 1e6, by / us3 / 1e6)
    if stride == 1:
        u = torch.randn(16, cout, cin, device=DEV) / math.sqrt(9 * cin)
        us2 = timeit(lambda: ops.conv3x3(x, w, b, cout, 1, 0.2, res=res, u=u), reps)
        line += "   | winograd %8.1f us  %6.1f algorithmic TFLOP/s" % (us2, fl / us2 / 1e6)
        from cmr_agent_amd.models._pack import conv_bf16_frags
        fr = conv_bf16_frags(torch.randn(cout, cin, 3, 3, device=DEV) / math.sqrt(9 * cin))
        if fr is not None and ops.conv3x3_bf16(x, fr, b, cout, 0.2, res=res) is not None:
            us3 = timeit(lambda: ops.conv3x3_bf16(x, fr, b, cout, 0.2, res=res), reps)
            line += "   | bf16 %8.1f us  %6.1f TFLOP/s  %5.2f TB/s" % (us3, fl / us3 / 1e6, by / us3 / 1e6)
    print(line)


def bench_linear(rows, k1, n_out, k2=0, gather=False, act=0, reps=20):
    x1 = torch.randn(rows, k1, device=DEV)
    w = torch.randn(n_out, k1 + k2, device=DEV) / math.sqrt(k1 + k2)
    b = torch.randn(n_out, device=DEV)
    x2 = idx = None
    if k2:
        m = max(rows // 13, 1)
        x2 = torch.randn(m if gather else rows, k2, device=DEV)
        if gather:
            idx = torch.randint(0, m, (rows,), device=DEV, dtype=torch.int32)
    out = torch.empty(rows, n_out, device=DEV)
    us = timeit(lambda: ops.linear(x1, w, b, x2=x2, idx2=idx, act=act, act_param=0.2, out=out), reps)
    fl = 2.0 * rows * (k1 + k2) * n_out
    by = 4.0 * rows * (k1 + k2 + n_out)
    print("linear %7d x (%4d%s) -> %-4d act%d : %8.1f us  %6.1f TFLOP/s  %5.2f TB/s" % (
        rows, k1, "+%d%s" % (k2, "g" if gather else "") if k2 else "", n_out, act, us, fl / us / 1e6, by / us / 1e6))


def bench_vit(rows_x, rows_y, B, reps):
    from cmr_agent_amd.models._pack import frag_pack
    r = lambda *shape: torch.randn(*shape, device=DEV) * 0.1
    x, y = r(rows_x, 64), r(rows_y, 64)
    g, b = r(64) + 1, r(64)
    wq, wkv, wqkv = frag_pack(r(64, 64)), frag_pack(r(128, 64)), frag_pack(r(192, 64))
    bq, bkv, bqkv = r(64), r(128), r(192)
    wo, w1, w2 = frag_pack(r(64, 64)), frag_pack(r(1024, 64)), frag_pack(r(64, 1024))
    bo, b1, b2 = r(64), r(1024), r(64)
    t1 = timeit(lambda: ops.ln64_linear(x, wqkv, bqkv, g, b, 1e-6), reps)
    t2 = timeit(lambda: ops.ln64_linear(x, wq, bq, g, b, 1e-6, y, wkv, bkv), reps)
    q, kv = ops.ln64_linear(x, wq, bq, g, b, 1e-6, y, wkv, bkv)
    t3 = timeit(lambda: ops.mha(q, kv[:, 0:64], kv[:, 64:128], B, rows_x // B, rows_y // B), reps)
    from cmr_agent_amd import _lib
    old = _lib.use_ab().cmr_set_mha_variant(0)
    t3v = timeit(lambda: ops.mha(q, kv[:, 0:64], kv[:, 64:128], B, rows_x // B, rows_y // B), reps)
    _lib.use_ab().cmr_set_mha_variant(old)
    t4 = timeit(lambda: ops.vit_out_ffn(q, x, wo, bo, (g, b), 1e-6, w1, b1, w2, b2), reps)
    print("vit block rows %d / %d: ln+qkv %.1f us  ln+q,kv %.1f us  mha %.1f us (vector-ALU kernel %.1f us)  out+ffn %.1f us" % (rows_x, rows_y, t1, t2, t3, t3v, t4))
    from cmr_agent_amd.models._pack import frag_pack_bf16 as fb
    wqb, wkvb, wqkvb = fb(r(64, 64)), fb(r(128, 64)), fb(r(192, 64))
    wob, w1b, w2b = fb(r(64, 64)), fb(r(1024, 64), acc_order=True), fb(r(64, 1024), acc_order=True)
    t1 = timeit(lambda: ops.ln64_linear(x, wqkvb, bqkv, g, b, 1e-6), reps)
    t2 = timeit(lambda: ops.ln64_linear(x, wqb, bq, g, b, 1e-6, y, wkvb, bkv), reps)
    t4 = timeit(lambda: ops.vit_out_ffn(q, x, wob, bo, (g, b), 1e-6, w1b, b1, w2b, b2), reps)
    print("   bf16 fragments           : ln+qkv %.1f us  ln+q,kv %.1f us  out+ffn %.1f us" % (t1, t2, t4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="conv,linear")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", default=None, help="A/B: load this build of libcmr_hip.so instead of the in-tree one")
    a = ap.parse_args()
    if a.lib:
        from cmr_agent_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    B = 8
    if "conv" in a.what:
        for (H, W, ci, co, s) in [(352, 1216, 64, 64, 1), (176, 608, 64, 64, 1), (88, 304, 64, 64, 1), (88, 304, 128, 128, 1),
                                  (88, 304, 128, 64, 1), (44, 152, 128, 128, 1), (22, 76, 128, 128, 1), (11, 38, 128, 128, 1),
                                  (352, 1216, 64, 64, 2), (176, 608, 64, 64, 2)]:
            bench_conv(B, H, W, ci, co, s, a.reps)
    if "latency" in a.what:      # dependent-launch floor inside a hipGraph: a chain of tiny kernels
        x = torch.randn(32, 64, device=DEV); g = torch.ones(64, device=DEV); b = torch.zeros(64, device=DEV)
        y = torch.empty_like(x)
        def chain():
            ops.layernorm64(x, g, b, 1e-5, out=y); ops.layernorm64(y, g, b, 1e-5, out=x)
        print("dependent tiny-kernel chain in a graph: %.2f us per kernel" % (timeit(chain, 200) / 2))
        xs = [torch.randn(32, 64, device=DEV) for _ in range(8)]; ys = [torch.empty_like(x) for _ in range(8)]
        def indep():
            for i in range(8): ops.layernorm64(xs[i], g, b, 1e-5, out=ys[i])
        print("independent tiny kernels (same stream) in a graph: %.2f us per kernel" % (timeit(indep, 50) / 8))
    if "cbr" in a.what:
        r = lambda *shape: torch.randn(*shape, device=DEV) * 0.1
        rows, B = 8 * 16384, 8
        for (kx, ch, co, conv, perb) in [(64, 128, 64, True, True), (64, 128, 128, False, True), (8, 8, 64, True, False), (64, 64, 64, False, False), (128, 128, 64, True, False)]:
            x = r(rows, kx)
            w1, w2 = r(ch, kx), r(co, ch)
            b1 = r(B, ch) if perb else r(ch)
            b2 = r(B, co) if perb else r(co)
            wsc = r(co, kx) if conv else None
            t = timeit(lambda: ops.cbr_block(x, w1, b1, w2, b2, wsc, 0.2, rows_per_batch=rows // B, want_colmax=True), a.reps)
            ops.CONV_BF16 = True
            tb = timeit(lambda: ops.cbr_block(x, w1, b1, w2, b2, wsc, 0.2, rows_per_batch=rows // B, want_colmax=True), a.reps)
            ops.CONV_BF16 = False
            fl = 2.0 * rows * (kx * ch + ch * co + (kx * co if conv else 0))
            by = 4.0 * rows * (kx + co)
            print("cbr_block %3d->%3d->%3d sc=%d: %7.1f us (incl. colmax pass)  %5.1f TFLOP/s   | bf16 %7.1f us  %5.1f TFLOP/s  %5.2f TB/s" % (
                kx, ch, co, conv, t, fl / t / 1e6, tb, fl / tb / 1e6, by / tb / 1e6))
    if "heads" in a.what:
        r = lambda *shape: torch.randn(*shape, device=DEV) * 0.1
        x, e3d = r(8 * 418, 128), r(8, 128)
        c24, c26 = (r(128, 128), r(128)), (r(128, 128), r(128))
        heads = [[(r(256, 256), r(256)), (r(256, 256), r(256)), (r(36, 256), r(36))] for _ in range(2)] + [[(r(64, 256), r(64)), (r(64, 64), r(64)), (r(4, 64), r(4))]]
        print("agent_heads: %.1f us" % timeit(lambda: ops.agent_heads(x, 8, 418, c24, c26, e3d, heads, 0.01), a.reps))
    if "vit" in a.what:
        bench_vit(8 * 418, 8 * 256, 8, a.reps)
        bench_vit(8 * 256, 8 * 418, 8, a.reps)
    if "stem" in a.what:
        r = lambda *shape: torch.randn(*shape, device=DEV) * 0.1
        img = torch.rand(8, 3, 352, 1216, device=DEV)
        t = timeit(lambda: ops.stem_block(img, r(3, 3, 3, 3), r(3), r(27, 64), r(3, 64), r(64), 0.2), a.reps)
        by = 8 * 352 * 1216 * (3 + 64) * 4.0
        print("stem_block 8x352x1216 3->64 (stem_a + stem_b): %.1f us  %.2f TB/s algorithmic (%.0f MB)" % (t, by / t / 1e6, by / 1e6))
    if "points" in a.what:
        # BASELINE configs[4] sizes: the point-side ops of the geometric model at 65 536 points (PointNN FPS / ball query /
        # kNN stress), per batch of B clouds.  Bytes = what the op must read / write once.
        for Bp in (1, 8):
            N, S = 65536, 1280
            xyz = (torch.rand(Bp, 3, N, device=DEV) * 80 - 40)
            x4 = ops.planar_to_rows(xyz, 4)
            start = torch.zeros(Bp, dtype=torch.int64, device=DEV)
            t = timeit(lambda: ops.fps(x4, start, Bp, N, S), max(a.reps // 4, 2))
            print("fps            B=%d %6d -> %4d : %9.1f us  = %6.2f us per round (2-barrier floor ~ 2 x 0.5 us); %5.2f GB/s of xyz0 reads if re-streamed "
                  "(the cloud stays in registers / LDS: algorithmic bytes %.2f MB)" % (Bp, N, S, t, t / S, Bp * N * 16.0 * S / t / 1e3, Bp * N * 16 / 1e6))
            idx = ops.fps(x4, start, Bp, N, S)
            nodes4 = ops.gather_rows(x4, (idx + torch.arange(Bp, device=DEV).view(Bp, 1) * N).view(-1).int())
            t = timeit(lambda: ops.ball_query(x4, nodes4, Bp, N, S, 32, 2.0), a.reps)
            by = Bp * (N * 16 + S * 16 + S * 32 * 8)
            print("ball_query     B=%d %4d x %6d k=32 : %9.1f us  %6.3f TB/s algorithmic (%0.2f MB); %5.1f Gpair/s" % (Bp, S, N, t, by / t / 1e6, by / 1e6, Bp * S * N / t / 1e3))
            t = timeit(lambda: ops.knn16(nodes4, Bp, S), a.reps)
            print("knn16          B=%d %4d x %4d        : %9.1f us  %5.1f Gpair/s" % (Bp, S, S, t, Bp * S * S / t / 1e3))
            t = timeit(lambda: ops.nearest(x4, nodes4, Bp, N, S), a.reps)
            by = Bp * (N * (16 + 4 + 8) + S * 16)
            print("nearest        B=%d %6d x %4d      : %9.1f us  %6.3f TB/s algorithmic; %5.1f Gpair/s" % (Bp, N, S, t, by / t / 1e6, Bp * S * N / t / 1e3))
            _, local = ops.nearest(x4, nodes4, Bp, N, S)
            gidx = ops.index_to_global(local, S)
            t = timeit(lambda: ops.csr_build(gidx, Bp, N, S), a.reps)
            by = Bp * (N * 8 + S * 8)
            print("csr_build      B=%d %6d -> %4d seg  : %9.1f us  %6.3f TB/s algorithmic" % (Bp, N, S, t, by / t / 1e6))
            offsets, order = ops.csr_build(gidx, Bp, N, S)
            feat = torch.randn(Bp * N, 64, device=DEV)
            t = timeit(lambda: ops.segment_reduce(feat, order, offsets, Bp * S, "max"), a.reps)
            by = Bp * (N * 260 + S * 256)
            print("segment_reduce B=%d %6d x 64 (max)  : %9.1f us  %6.3f TB/s algorithmic" % (Bp, N, t, by / t / 1e6))
            a_, v_ = torch.randn(Bp * N, 64, device=DEV), torch.randn(Bp * N, 64, device=DEV)
            t = timeit(lambda: ops.segment_softmax(a_, v_, Bp * S, 0.125, order=order, offsets=offsets), a.reps)
            by = Bp * (N * 516 + S * 256)
            print("segment_softmax B=%d %6d x 64       : %9.1f us  %6.3f TB/s algorithmic" % (Bp, N, t, by / t / 1e6))
            sq = timeit(lambda: ops.square_distance(nodes4, x4, Bp, S, N), max(a.reps // 4, 2))
            by = Bp * S * N * 4
            print("square_distance B=%d %4d x %6d      : %9.1f us  %6.3f TB/s (output write)" % (Bp, S, N, sq, by / sq / 1e6))
    if "linear" in a.what:
        N, P4, M, T = 8 * 16384, 8 * 26752, 8 * 1280, 8 * 418
        for args in [(N, 64, 64), (N, 64, 64, 64, True), (N, 64, 128), (N, 64, 128, 64, True), (N, 128, 64), (N, 4, 64), (N, 8, 8),
                     (N, 64, 32), (N, 32, 2), (P4, 64, 64), (P4, 64, 128, 64, False), (P4, 128, 64), (M, 64, 64), (M, 64, 192),
                     (T, 64, 192), (T, 64, 1024), (T, 1024, 64), (T, 4096, 64), (8, 256, 256), (8, 128, 128)]:
            bench_linear(*args, reps=a.reps)


if __name__ == "__main__":
    main()
