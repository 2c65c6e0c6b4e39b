"""cmr_agent_amd.dataset.loader.FrameDataset: the file-reading loader for the reference's on-disk frames (dataset/KittiDataset.py:128-180,
258-423).  CPU tier: three tiny frames are written in the reference's layout and the host half (listing, calibration,
random draws, intrinsics chain) is checked against the reference's arithmetic; GPU tier: the whole sample dict against the conventions of
the reference-generated kitti_frame fixture and the oracle's numpy restatement fed the loader's own draws."""
import os
import random

import numpy as np
import pytest
import torch

import cases as C
import golden_util as G


def _write_dataset(root, seqs=(9,), frames=3, with_image_3=True, n_raw=None, img_hw=None):
    f = C.FRAME
    n_raw = n_raw or f["n_raw"]
    ih, iw = img_hw or (f["img_h"], f["img_w"])
    p2, tr = C.FRAME_P2, C.FRAME_TR
    fmt = lambda name, v: name + ": " + " ".join("%.12e" % x for x in v) + "\n"
    for seq in seqs:
        os.makedirs(os.path.join(root, "calib", "%02d" % seq), exist_ok=True)
        with open(os.path.join(root, "calib", "%02d" % seq, "calib.txt"), "w") as fh:
            p3 = list(p2)
            p3[3] = p2[3] - 386.1448                                   # right colour camera: another baseline term
            fh.write(fmt("P0", p2) + fmt("P1", p2) + fmt("P2", p2) + fmt("P3", p3) + fmt("Tr", tr))
        for cam in ("image_2", "image_3") if with_image_3 else ("image_2",):
            os.makedirs(os.path.join(root, "data_odometry_color_npy", "sequences", "%02d" % seq, cam), exist_ok=True)
        os.makedirs(os.path.join(root, "data_odometry_velodyne_NWU", "sequences", "%02d" % seq, "voxel0.1-SNr0.6"), exist_ok=True)
        rng = np.random.RandomState(100 + seq)
        for i in range(frames):
            raw = C.frame_raw_cloud()[:, :n_raw].copy()
            raw[0] += 0.01 * i
            np.save(os.path.join(root, "data_odometry_velodyne_NWU", "sequences", "%02d" % seq, "voxel0.1-SNr0.6", "%06d.npy" % i), raw)
            for cam in ("image_2", "image_3") if with_image_3 else ("image_2",):
                np.save(os.path.join(root, "data_odometry_color_npy", "sequences", "%02d" % seq, cam, "%06d.npy" % i),
                        rng.randint(0, 256, size=(ih, iw, 3)).astype(np.uint8))


def _config(device="cpu"):
    from cmr_agent_amd.config import KittiConfiguration
    f = C.FRAME
    cfg = KittiConfiguration(cropped_img_H=f["H"], cropped_img_W=f["W"], num_pt=f["num_pt"], device=device)
    cfg.num_node = f["num_node"]
    return cfg


def test_listing_calibration_and_host_draws(tmp_path):
    from cmr_agent_amd.dataset.loader import FrameDataset, read_calib
    root = str(tmp_path)
    _write_dataset(root, seqs=(0, 9), frames=3)
    calib = read_calib(root)
    P_Tr, Kq = C.frame_calib()                                        # the reference helper's arithmetic on the same rows (tests/cases.py)
    assert np.array_equal(np.dot(calib[9]["P2"], calib[9]["Tr"]), P_Tr)
    cfg = _config()
    val = FrameDataset(root, cfg, "val", device="cpu")
    train = FrameDataset(root, cfg, "train", device="cpu")
    assert len(val) == 6 and len(train) == 6                          # 3 frames x (image_2, image_3) of sequence 09 / 00 (KittiDataset.py:172-177)
    assert [f[4] for f in val.frames[:2]] == ["P2", "P3"] and val.frames[0][2] == 9 and train.frames[0][2] == 0
    with pytest.raises(Exception):
        FrameDataset(root, cfg, "bogus", device="cpu")
    random.seed(5)
    np.random.seed(5)
    f = val.read_frame(0)
    F = C.FRAME
    assert f["img"].shape == (F["img_h"], F["img_w"], 3) and f["img"].dtype == np.uint8
    assert f["raw"].shape == (4, F["n_raw"]) and f["raw"].dtype == np.float32
    assert np.array_equal(f["P_Tr"], P_Tr)
    rh, rw = int(round(F["img_h"] * 0.5)), int(round(F["img_w"] * 0.5))
    assert f["crop"] == (int((rw - F["W"]) / 2), int((rh - F["H"]) / 2))          # 'val': the centre crop (:300-302)
    assert np.array_equal(f["K"], Kq) and f["K"].dtype == np.float32            # K at 1/4 scale of the crop = the fixture's chain
    assert f["choice"].shape == (F["num_pt"],) and len(set(f["choice"].tolist())) == F["num_pt"] and f["choice"].max() < F["n_raw"]
    assert f["t"][1] == 0.0 and abs(f["t"][0]) <= 10.0 and f["angles"][0] == 0.0 and abs(f["angles"][1]) <= np.pi
    # a cloud shorter than num_pt is tiled, then topped up without replacement (:186-190)
    small = FrameDataset(root, cfg, "val", device="cpu")
    small.num_pt = 2 * F["n_raw"] + 100
    ch = small.downsample_choice(F["n_raw"])
    assert ch.shape == (small.num_pt,) and np.array_equal(ch[:F["n_raw"]], np.arange(F["n_raw"])) and np.array_equal(ch[F["n_raw"]:2 * F["n_raw"]], np.arange(F["n_raw"]))
    # train mode: random crop inside the half-size image
    g = train.read_frame(1)
    assert 0 <= g["crop"][0] <= rw - F["W"] and 0 <= g["crop"][1] <= rh - F["H"]
    # sequences that are not on disk are skipped, not fatal
    assert all(fr[2] in (0, 9) for fr in train.frames + val.frames)


@pytest.mark.gpu
def test_frame_dataset_sample_dict_on_device(tmp_path):
    from cmr_agent_amd.dataset.loader import FrameDataset, FrameLoader
    from oracle import cmr_oracle as O
    from test_oracle_golden import FRAME_KEYS
    root = str(tmp_path)
    _write_dataset(root, seqs=(9,), frames=3)
    cfg = _config("cuda")
    ds = FrameDataset(root, cfg, "val", device="cuda")
    random.seed(11)
    np.random.seed(11)
    s = ds[2]
    torch.cuda.synchronize()
    d, F = ds.last_draws, C.FRAME
    fx = G.load_case("kitti_frame")
    # the dict contract of KittiDataset.py:400-423 = the keys / shapes / dtypes of the reference-generated fixture (+ img, angles, translation)
    for k in FRAME_KEYS:
        assert k in s, k
        assert tuple(s[k].shape) == tuple(int(x) for x in fx[k]["shape"]), (k, tuple(s[k].shape), tuple(fx[k]["shape"]))
        assert (s[k].dtype == torch.int64) == (fx[k]["sample"].dtype.kind in "iu"), (k, s[k].dtype)
    assert tuple(s["img"].shape) == (3, F["H"], F["W"]) and s["img"].dtype == torch.float32 and 0.0 <= float(s["img"].min()) and float(s["img"].max()) <= 1.0
    assert s["angles"].dtype == torch.float64 and tuple(s["translation"].shape) == (3,)
    # values: the oracle's numpy restatement of __getitem__ fed the SAME draws
    from cmr_agent_amd.dataset.frame import random_transform
    img_folder, pc_folder, seq, i, key = ds.frames[2]
    raw = np.load(os.path.join(pc_folder, "%06d.npy" % i)).astype(np.float32)
    P_Tr, Kq = C.frame_calib() if key == "P2" else (None, None)
    assert key == "P2"
    ref = O.kitti_frame(raw, P_Tr, Kq, random_transform(list(d["t"]), list(d["angles"])), (F["H"] // 4, F["W"] // 4), d["choice"], d["perm"], d["cand"],
                        d["fps_start"], F["num_node"])
    for k in FRAME_KEYS:
        g, r = s[k].cpu().numpy(), np.asarray(ref[k])
        assert g.shape == r.shape, (k, g.shape, r.shape)
        if r.dtype.kind in "iu":
            assert (g == r).all(), (k, int((g != r).sum()))
        else:
            assert np.abs(g.astype(np.float64) - r).max() <= 1e-6 * max(1.0, np.abs(r).max()), (k, np.abs(g - r).max())
    # the image: half-size bilinear resize + centre crop of the stored uint8 frame
    img = np.load(os.path.join(img_folder, "%06d.npy" % i)).astype(np.float64)
    rh, rw = int(round(img.shape[0] * 0.5)), int(round(img.shape[1] * 0.5))
    dx, dy = d["crop"]
    yy, xx = 37, 101                                                  # one output pixel by hand: source centre (2 (y + dy) + 0.5, 2 (x + dx) + 0.5)
    ys, xs = 2 * (yy + dy), 2 * (xx + dx)
    want = img[ys:ys + 2, xs:xs + 2].mean(axis=(0, 1)) if img.shape[0] == 2 * rh and img.shape[1] == 2 * rw else None
    if want is not None:
        assert np.abs(s["img"][:, yy, xx].cpu().numpy() * 255.0 - np.round(want)).max() <= 1.0
    # batches: the reference's DataLoader collate
    loader = FrameLoader(ds, batch_size=2, shuffle=False, drop_last=True)
    batches = list(loader)
    assert len(batches) == 3 and tuple(batches[0]["pc"].shape) == (2, 3, F["num_pt"]) and tuple(batches[0]["img"].shape) == (2, 3, F["H"], F["W"])
    assert batches[0]["pt2node"].dtype == torch.int64 and tuple(batches[0]["node"].shape) == (2, 3, F["num_node"])
    # train mode (:311-312): the colour jitter runs on the device, its draws are recorded, and replaying them on the un-jittered crop gives
    # the sample's image
    from cmr_agent_amd.dataset.loader import color_jitter
    root_t = str(tmp_path / "train")
    os.makedirs(root_t)
    _write_dataset(root_t, seqs=(0,), frames=2)
    dt = FrameDataset(root_t, cfg, "train", device="cuda")
    random.seed(3)
    np.random.seed(3)
    torch.manual_seed(3)
    st = dt[1]
    jit, crop = dt.last_jitter, dt.last_draws["crop"]                  # (image_tensor(jitter=False) below clears last_jitter; read_frame draws anew)
    assert jit is not None and sorted(jit[0]) == [0, 1, 2, 3]
    f = dt.read_frame(1)
    plain = dt.image_tensor(f["img"], f["resized"], crop, jitter=False)
    replay, _ = color_jitter((plain * 255.0).round(), draws=jit)         # (the 0..255 grid the sample's image was jittered on)
    assert torch.equal(st["img"], (replay / 255.0).contiguous()) and not torch.equal(st["img"], plain)
    assert 0.0 <= float(st["img"].min()) and float(st["img"].max()) <= 1.0


def test_color_jitter_against_pil_enhancers():
    """KittiDataset.augment_img (:209-218) hands a PIL image to torchvision's ColorJitter, whose PIL branch is ImageEnhance.Brightness /
    Contrast / Color and an 8-bit HSV hue rotation.  torchvision is not in this image; PIL is: each adjustment of the device restatement
    against PIL's own enhancer on the same uint8 image (+-1 grey level: PIL truncates where the restatement rounds; the hue path goes
    through PIL's 8-bit HSV, so a few levels more), and the composition order / draw order of ColorJitter.get_params."""
    from PIL import Image, ImageEnhance
    from cmr_agent_amd.dataset.loader import color_jitter
    rng = np.random.RandomState(3)
    yy, xx = np.mgrid[0:48, 0:64]
    img = np.stack([(xx * 4) % 256, (yy * 5 + xx) % 256, (255 - xx * 3 - yy) % 256], axis=2).astype(np.uint8)
    img[8:24, 8:40] = rng.randint(0, 256, (16, 32, 3))
    pil = Image.fromarray(img)
    x = torch.from_numpy(img).permute(2, 0, 1).float()

    def mine(order, b=1.0, c=1.0, s=1.0, h=0.0):
        y, d = color_jitter(x, draws=(order, b, c, s, h))
        assert d == (order, b, c, s, h)
        return y.permute(1, 2, 0).numpy()

    assert np.array_equal(mine([0, 1, 2, 3]), img.astype(np.float32))                      # unit factors, zero hue shift: the identity
    for f in (0.8, 1.13, 1.2):
        d = np.abs(mine([0], b=f) - np.asarray(ImageEnhance.Brightness(pil).enhance(f), dtype=np.float32))
        assert d.max() <= 1.0, ("brightness", f, d.max())
        d = np.abs(mine([1], c=f) - np.asarray(ImageEnhance.Contrast(pil).enhance(f), dtype=np.float32))
        assert d.max() <= 1.0, ("contrast", f, d.max())
        d = np.abs(mine([2], s=f) - np.asarray(ImageEnhance.Color(pil).enhance(f), dtype=np.float32))
        assert d.max() <= 2.0 and d.mean() <= 0.6, ("saturation", f, d.max(), d.mean())
    for f in (-0.1, 0.04, 0.1):
        hh, ss, vv = pil.convert("HSV").split()
        nh = np.array(hh, dtype=np.uint8)
        nh = (nh.astype(np.int32) + int(np.uint8(np.int32(f * 255)))).astype(np.uint8)      # uint8 wrap-around, as F_pil.adjust_hue
        ref = np.asarray(Image.merge("HSV", (Image.fromarray(nh, "L"), ss, vv)).convert("RGB"), dtype=np.float32)
        d = np.abs(mine([3], h=f) - ref)
        assert d.mean() <= 2.0 and np.percentile(d, 99) <= 8.0, ("hue", f, d.mean(), d.max())
    # composition: the adjustments apply in the drawn order, each on the previous one's 0..255 result
    two = mine([2, 0], b=0.9, s=1.15)
    step = color_jitter(color_jitter(x, draws=([2], 1.0, 1.0, 1.15, 0.0))[0], draws=([0], 0.9, 1.0, 1.0, 0.0))[0].permute(1, 2, 0).numpy()
    assert np.array_equal(two, step)
    # draws: randperm(4) first, then brightness, contrast, saturation, hue from torch's global generator (ColorJitter.get_params)
    torch.manual_seed(7)
    _, d = color_jitter(x)
    torch.manual_seed(7)
    order = torch.randperm(4).tolist()
    fac = [float(torch.empty(1).uniform_(lo, hi)) for lo, hi in ((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))]
    assert d == (order, *fac)
    assert 0.8 <= d[1] <= 1.2 and -0.1 <= d[4] <= 0.1
