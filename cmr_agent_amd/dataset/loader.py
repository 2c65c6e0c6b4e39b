"""File-reading loader for the reference's on-disk frames (reference dataset/KittiDataset.py:128-180, 258-423).

Layout read (the reference's, KittiDataset.py:156-180, 258-264, 63-99):

    <root>/calib/<seq>/calib.txt                                         P0..P3 and Tr rows of a KITTI odometry sequence
    <root>/<data_color>/sequences/<seq>/image_2|image_3/<i>.npy          uint8 [H, W, 3] images
    <root>/<data_velodyne>/sequences/<seq>/voxel0.1-SNr0.6/<i>.npy       float [4, n] clouds (x, y, z, reflectance), velodyne frame

`FrameDataset(root, config, mode)[index]` does on the HOST what must happen there -- reading the two files, the calibration arithmetic
on 3x3 / 4x4 matrices and the RANDOM DRAWS in the reference's order (down-sampling choice :284, crop offsets :297-302, circle-loss
permutation :340, random pose :349, node candidates :356, FPS start :117) -- and hands the raw cloud to
`dataset.frame.preprocess_frame`, which runs every per-point step as HIP kernels: the frame goes to the device as it lies on disk and
the sample dict (KittiDataset.py:400-423: same keys, shapes and dtypes) comes back as device tensors.

Image side: the half-size bilinear resize (:290-293) and the crop (:296-304) run on the device with torch's half-pixel bilinear filter
(the geometry of cv2.INTER_LINEAR; OpenCV's fixed-point rounding is not reproduced: +-1 grey level).  The colour jitter of training
mode (:209-218, :312: torchvision ColorJitter, brightness / contrast / saturation in 0.8..1.2, hue in +-0.1, the four in a random order)
runs on the device as `color_jitter` below: torchvision's published arithmetic on the 0..255 grid with a rounding after every
adjustment; its draws come from torch's global generator in torchvision's order (randperm(4), then one uniform per factor).  Neither
OpenCV nor torchvision exists in this image, so both image steps are restatements WITHOUT a reference-generated fixture (the per-point
side of the sample dict is pinned by tests/golden/dataset_ops.npz)."""
import os
import random

import numpy as np
import torch

from .frame import camera_matrix, preprocess_frame, random_transform


def _gray(x):
    return 0.2989 * x[0:1] + 0.587 * x[1:2] + 0.114 * x[2:3]


def _shift_hue(x, f):
    """torchvision adjust_hue on a [3, H, W] image in 0..1: RGB -> HSV, h <- (h + f) mod 1, HSV -> RGB."""
    r, g, b = x[0], x[1], x[2]
    maxc, minc = x.max(dim=0)[0], x.min(dim=0)[0]
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    h = torch.remainder(h + f, 1.0)
    v = maxc
    i = torch.floor(h * 6.0)
    fr = h * 6.0 - i
    i = i.to(torch.int64) % 6
    p, q, t = v * (1.0 - s), v * (1.0 - s * fr), v * (1.0 - s * (1.0 - fr))
    pick = lambda c: torch.stack(c, dim=0).gather(0, i.unsqueeze(0))[0]      # noqa: E731
    return torch.stack((pick((v, q, p, p, t, v)), pick((t, v, v, q, p, p)), pick((p, p, t, v, v, q))), dim=0).clamp_(0.0, 1.0)


def color_jitter(img255, brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=(-0.1, 0.1), draws=None):
    """KittiDataset.augment_img (:209-218) on a device image [3, H, W] holding grey levels 0..255 (float): torchvision ColorJitter --
    order = randperm(4), then one uniform factor each for brightness, contrast, saturation, hue (ColorJitter.get_params, torch's global
    generator); brightness / contrast / saturation blend the image with zero / its mean grey level / its grey image
    (f x + (1 - f) other, clamped), hue shifts the HSV angle.  The reference runs on a uint8 PIL image: every adjustment lands on the
    0..255 grid again (rounded here).  draws: (order list, b, c, s, h) to replay a known jitter; -> (image, draws)."""
    if draws is None:
        order = torch.randperm(4).tolist()
        b, c, s, h = (float(torch.empty(1).uniform_(lo, hi)) for lo, hi in (brightness, contrast, saturation, hue))
    else:
        order, b, c, s, h = draws
    x = img255 / 255.0
    grid = lambda y: (y * 255.0).round().clamp_(0.0, 255.0) / 255.0          # noqa: E731
    for k in order:
        if k == 0:
            x = grid(x * b)
        elif k == 1:
            x = grid(c * x + (1.0 - c) * (_gray(x) * 255.0).round().div(255.0).mean())
        elif k == 2:
            x = grid(s * x + (1.0 - s) * _gray(x))
        else:
            x = grid(_shift_hue(x, h))
    return x * 255.0, (order, b, c, s, h)


def read_calib(root):
    """KittiCalibHelper.read_calib_files (KittiDataset.py:63-99) -> {seq: {'P2': 4x4, 'P2_K': 3x3 float32, ..., 'Tr': 4x4}}."""
    out = {}
    base = os.path.join(root, "calib")
    for seq in sorted(os.listdir(base)):
        path = os.path.join(base, seq, "calib.txt")
        if not os.path.isfile(path):
            continue
        d = out.setdefault(int(seq), {})
        with open(path, "r") as f:
            for line in f.readlines():
                if len(line) < 5:
                    continue
                key = line[0:2]
                mat = np.array(line[4:].split(), dtype=np.float64).reshape((3, 4)).astype(np.float32)
                if key == "Tr":
                    P = np.identity(4)
                    P[0:3, :] = mat
                    d[key] = P
                else:
                    K = mat[0:3, 0:3]
                    d[key + "_K"] = K
                    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
                    tz = mat[2, 3]
                    P = np.identity(4)
                    P[0:3, 3] = np.asarray([(mat[0, 3] - cx * tz) / fx, (mat[1, 3] - cy * tz) / fy, tz])
                    d[key] = P
    return out


class FrameDataset:
    """config: the attribute bag of cmr_agent_amd.config (dataset_root is overridden by `root`); mode 'train' | 'val' | 'test'.
    device: where the samples are produced (default: the current HIP device).  Sequences that are not on disk are skipped (the
    reference lists 00-08 for training and 09-10 otherwise and fails on a missing folder)."""

    SEQUENCES = {"train": (0, 1, 2, 3, 4, 5, 6, 7, 8), "val": (9, 10), "test": (9, 10)}
    PC_SUBDIR = "voxel0.1-SNr0.6"

    def __init__(self, root, config, mode, device=None, n_circle=512):
        if mode not in self.SEQUENCES:
            raise Exception("Invalid mode...")                                     # KittiDataset.py:162
        self.root, self.config, self.mode = root, config, mode
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.num_pt, self.num_node = config.num_pt, config.num_node
        self.img_H, self.img_W = config.cropped_img_H, config.cropped_img_W
        self.n_circle = n_circle
        self.amp_t = (config.P_Tx_amplitude, config.P_Ty_amplitude, config.P_Tz_amplitude)
        self.amp_r = (config.P_Rx_amplitude, config.P_Ry_amplitude, config.P_Rz_amplitude)
        self.calib = read_calib(root)
        self.frames = self._list_frames()
        self.last_draws = None
        self.last_jitter = None
        print("%d samples in %s set..." % (len(self.frames), mode))                # :154

    def _list_frames(self):
        cfg, frames = self.config, []
        for seq in self.SEQUENCES[self.mode]:
            img2 = os.path.join(self.root, cfg.data_color, "sequences", "%02d" % seq, "image_2")
            img3 = os.path.join(self.root, cfg.data_color, "sequences", "%02d" % seq, "image_3")
            pcs = os.path.join(self.root, cfg.data_velodyne, "sequences", "%02d" % seq, self.PC_SUBDIR)
            if not (os.path.isdir(img2) and os.path.isdir(pcs)) or seq not in self.calib:
                continue
            num = len(os.listdir(img2))
            if self.mode == "val":
                num = min(num, 100)                                                # :169-170 (100 frames per validation sequence)
            for i in range(num):
                frames.append((img2, pcs, seq, i, "P2"))
                if os.path.isdir(img3):
                    frames.append((img3, pcs, seq, i, "P3"))
        return frames

    def __len__(self):
        return len(self.frames)

    # ------------------------------------------------------------------------------------------------------------- host side
    def downsample_choice(self, n):
        """KittiDataset.py:182-191: num_pt indices into a cloud of n points (without replacement; a short cloud is tiled first)."""
        if n >= self.num_pt:
            return np.random.choice(n, self.num_pt, replace=False)
        fix = np.asarray(range(n))
        while n + fix.shape[0] < self.num_pt:
            fix = np.concatenate((fix, np.asarray(range(n))), axis=0)
        return np.concatenate((fix, np.random.choice(n, self.num_pt - fix.shape[0], replace=False)), axis=0)

    def read_frame(self, index):
        """The host half of __getitem__: files, calibration, every random draw that does not depend on device results.
        -> dict(img uint8 [H, W, 3], raw float32 [>=3, n], P_Tr, K (1/4 scale of the crop), crop, P_random, angles, t, choice, cand, fps_start)."""
        img_folder, pc_folder, seq, i, key = self.frames[index]
        img = np.load(os.path.join(img_folder, "%06d.npy" % i))
        raw = np.ascontiguousarray(np.load(os.path.join(pc_folder, "%06d.npy" % i)), dtype=np.float32)
        P_Tr = np.dot(self.calib[seq][key], self.calib[seq]["Tr"])                  # :273-274
        choice = self.downsample_choice(raw.shape[1])                              # :284
        rh, rw = int(round(img.shape[0] * 0.5)), int(round(img.shape[1] * 0.5))     # :290-293
        if rh < self.img_H or rw < self.img_W:
            raise ValueError("frame %s/%06d: the half-size image %dx%d is smaller than the crop %dx%d" % (img_folder, i, rh, rw, self.img_H, self.img_W))
        if self.mode == "train":
            dx, dy = random.randint(0, rw - self.img_W), random.randint(0, rh - self.img_H)      # :297-299
        else:
            dx, dy = int((rw - self.img_W) / 2), int((rh - self.img_H) / 2)
        K = camera_matrix(self.calib[seq][key + "_K"], 0.5, (dx, dy), 0.25)        # :294, :306, :309
        t = [random.uniform(-a, a) for a in self.amp_t]                             # :241-246
        angles = [random.uniform(-a, a) for a in self.amp_r]
        return dict(img=img, raw=raw, P_Tr=P_Tr, K=K, crop=(dx, dy), resized=(rh, rw), P_random=random_transform(t, angles), angles=np.array(angles),
                    t=np.array(t), choice=choice.astype(np.int64))

    # ----------------------------------------------------------------------------------------------------------- device side
    def image_tensor(self, img, resized, crop, jitter=False):
        """uint8 [H, W, 3] -> float32 [3, img_H, img_W] in 0..1 on the device: half-size bilinear resize, crop, (train mode: colour
        jitter, :311-312), / 255 (:290-304, :401)."""
        x = torch.from_numpy(np.ascontiguousarray(img)).to(self.device).permute(2, 0, 1).unsqueeze(0).float()
        x = torch.nn.functional.interpolate(x, size=resized, mode="bilinear", align_corners=False, antialias=False)
        x = x.round().clamp_(0, 255)
        dx, dy = crop
        x = x[0, :, dy:dy + self.img_H, dx:dx + self.img_W]
        self.last_jitter = None
        if jitter:
            x, self.last_jitter = color_jitter(x)
        return (x / 255.0).contiguous()

    def __getitem__(self, index):
        f = self.read_frame(index)
        dev = self.device
        hw4 = (int(self.img_H * 0.25), int(self.img_W * 0.25))
        draws = dict(choice=f["choice"], crop=f["crop"], t=f["t"], angles=f["angles"])

        def perm(count):                                                           # :340 -- needs the in-picture count: one device round trip
            p = np.random.permutation(count)[0:self.n_circle]
            draws["perm"] = p
            return torch.from_numpy(p.astype(np.int64)).to(dev)

        def nodes():                                                               # :356 candidates, :117 FPS start in {0, 1, 2}: drawn AFTER the
            cand = np.random.choice(self.num_pt, self.num_node * 8, replace=False)  # circle-loss permutation (:340), as in the reference's np.random stream
            fps_start = int(np.random.randint(3))
            draws.update(cand=cand, fps_start=fps_start)
            return torch.from_numpy(cand.astype(np.int64)).to(dev), fps_start

        # the image first: the reference jitters (:312, torch's generator) before it draws the permutation (np.random), the candidates and the pose
        img = self.image_tensor(f["img"], f["resized"], f["crop"], jitter=self.mode == "train")
        out = preprocess_frame(torch.from_numpy(f["raw"]).to(dev), f["P_Tr"], f["K"], f["P_random"], hw4,
                               choice=torch.from_numpy(f["choice"]).to(dev), perm=perm, node_candidates=nodes,
                               num_node=self.num_node, n_circle=self.n_circle)
        out.pop("in_picture_count", None)
        out["img"] = img
        out["angles"] = torch.from_numpy(f["angles"])
        out["translation"] = torch.from_numpy(f["t"])
        self.last_draws = draws
        return out


def collate(samples):
    """default_collate of the reference's DataLoader for the sample dicts: every key stacked along a new batch dimension."""
    return {k: torch.stack([s[k] for s in samples], dim=0) for k in samples[0]}


class FrameLoader:
    """for data in FrameLoader(dataset, batch_size, shuffle, drop_last): the reference's DataLoader(...) over a FrameDataset, in-process
    (the per-point work is on the device: there is nothing for worker processes to do but read files)."""

    def __init__(self, dataset, batch_size, shuffle=False, drop_last=True):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        order = list(range(len(self.dataset)))
        if self.shuffle:
            random.shuffle(order)
        for b in range(len(self)):
            ids = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield collate([self.dataset[i] for i in ids])
