"""Round 6, CPU only: is a Winograd-domain fp32 weight gradient accurate enough to build?

    dg = sum over 2x2 output tiles of  G^T [ (A dY A^T) (.) (B^T d B) ] G            (F(2x2,3x3); the transpose of the forward's Y = A^T[(G g G^T)(.)(B^T d B)]A)

= 16 position GEMMs  M[xi][co][ci] = sum_tiles P[xi][tile][co] V[xi][tile][ci]  (K = tiles), then the 4x4 -> 3x3 fold with G.  Compared here, at the
C5 map (352 x 1216 x B, 64 -> 64 channels; K = 107 008 tiles per image), against the direct sum  dg[ky][kx][co][ci] = sum_pixels dY[p][co] d[p + (ky,kx)][ci]:
both in fp32 with the accumulation shape of the HIP kernels (fp32 partial sums over strips of `strip` tiles / 4 x `strip` pixels, strips and images
added in fp32), both against the float64 direct sum.  python tools/wino_wgrad_check.py [B] [strip]"""
import sys
import time

import numpy as np

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
STRIP = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H, W, C = 352, 1216, 64
rng = np.random.default_rng(2023)
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def strips_matmul(a, b, strip, dtype):
    """a [K][M], b [K][N] -> a^T b accumulated strip by strip in `dtype` (fp32: what resident MFMA accumulators + a reduce kernel do)."""
    acc = np.zeros((a.shape[1], b.shape[1]), dtype=dtype)
    for k in range(0, a.shape[0], strip * 64):                      # 64 strips per partial: the reduce kernel's fan-in is not modelled finer
        acc += a[k:k + strip * 64].T @ b[k:k + strip * 64]
    return acc


ref = np.zeros((3, 3, C, C), np.float64)
direct = np.zeros((3, 3, C, C), np.float32)
wino = np.zeros((4, 4, C, C), np.float32)
t0 = time.time()
for b in range(B):
    x = rng.standard_normal((H, W, C)).astype(np.float32)
    d = np.where(x > 0, x, 0.01 * x).astype(np.float32)            # an activated map: positive mean (what the transforms must cope with)
    dy = (rng.standard_normal((H, W, C)) * 1e-3).astype(np.float32)
    dp = np.zeros((H + 2, W + 2, C), np.float32)
    dp[1:-1, 1:-1] = d
    dyr = dy.reshape(-1, C)
    for ky in range(3):
        for kx in range(3):
            sh = np.ascontiguousarray(dp[ky:ky + H, kx:kx + W]).reshape(-1, C)
            ref[ky, kx] += dyr.astype(np.float64).T @ sh.astype(np.float64)
            direct[ky, kx] += strips_matmul(dyr, sh, 4 * STRIP, np.float32)
    # Winograd domain: tiles (ty, tx) -> input 4x4 window at (2ty, 2tx) of the padded map, output-gradient 2x2 block
    th, tw = H // 2, W // 2
    win = np.stack([np.stack([dp[i:i + 2 * th:2, j:j + 2 * tw:2] for j in range(4)], 0) for i in range(4)], 0)      # [4][4][th][tw][C]
    win = win.reshape(4, 4, -1, C)
    blk = np.stack([np.stack([dy[i::2, j::2] for j in range(2)], 0) for i in range(2)], 0).reshape(2, 2, -1, C)       # [2][2][tiles][C]
    Bt32, At32 = Bt.astype(np.float32), At.astype(np.float32)
    V = np.einsum("ai,ijtc,bj->abtc", Bt32, win, Bt32).astype(np.float32)      # B^T d B  (adds / subtracts only)
    P = np.einsum("ia,ijtc,jb->abtc", At32, blk, At32).astype(np.float32)      # A dY A^T (A = At^T: 4x2)
    for a in range(4):
        for c in range(4):
            wino[a, c] += strips_matmul(P[a, c], V[a, c], STRIP, np.float32)
    print("image %d / %d  (%.0f s)" % (b + 1, B, time.time() - t0), flush=True)
G32 = G.astype(np.float32)
dg_w = np.einsum("ak,abnm,bl->klnm", G32, wino, G32).astype(np.float32)          # G^T M G in fp32
dg_w64 = np.einsum("ak,abnm,bl->klnm", G, wino.astype(np.float64), G)
scale = np.abs(ref).max()
e_d = np.abs(direct - ref)
e_w = np.abs(dg_w - ref)
print("map %dx%dx%d, %d -> %d channels, K = %d tiles (%d pixels); strips of %d tiles" % (H, W, B, C, C, B * (H // 2) * (W // 2), B * H * W, STRIP))
print("max |dg| (float64 direct)           %.6e" % scale)
print("direct   fp32: max err %.3e (%.3e of scale), rms %.3e" % (e_d.max(), e_d.max() / scale, np.sqrt((e_d ** 2).mean())))
print("Winograd fp32: max err %.3e (%.3e of scale), rms %.3e" % (e_w.max(), e_w.max() / scale, np.sqrt((e_w ** 2).mean())))
print("Winograd fp32 sums, fold in float64: max err %.3e" % np.abs(dg_w64 - ref).max())
print("ratio of max errors (Winograd / direct) = %.2f ; rms ratio = %.2f   [build if <= 2]" % (e_w.max() / e_d.max(), np.sqrt((e_w ** 2).mean() / (e_d ** 2).mean())))
