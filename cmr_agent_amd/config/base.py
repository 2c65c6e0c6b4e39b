"""Attribute-bag configurations with the reference's attribute names
(config/KittiConfig.py:12-118, config/NuScenesConfig.py -- the two differ only in
paths, crop width, epochs, val_interval, workers and step_size).

Unlike the reference, construction has no side effects besides building the two
float64 step tables, the image size / point count can be overridden (BASELINE
configs use 352x1216 / 16384 pts, the reference default is 160x512 / 40960), and
H, W are validated: the 8x8 patching followed by the x8 nearest up-sampling needs
the network input to be a multiple of 32 (IMGPCEnDecoder.py:85-88).
"""
import math

import numpy as np
import torch


def default_device():
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


class _Configuration:
    dataset_name = "base"
    _defaults = {}

    def __init__(self, data_root=None, cropped_img_H=None, cropped_img_W=None, num_pt=None, device=None,
                 **overrides):
        d = self._defaults
        # dataset
        self.dataset_root = d["dataset_root"] if data_root is None else data_root
        self.num_pt = 40960 if num_pt is None else int(num_pt)
        self.P_Tx_amplitude, self.P_Ty_amplitude, self.P_Tz_amplitude = 10.0, 0.0, 10.0
        self.P_Rx_amplitude, self.P_Ry_amplitude, self.P_Rz_amplitude = 0.0, math.pi, 0.0
        self.cropped_img_H = 160 if cropped_img_H is None else int(cropped_img_H)
        self.cropped_img_W = d["cropped_img_W"] if cropped_img_W is None else int(cropped_img_W)
        if self.cropped_img_H % 32 or self.cropped_img_W % 32:
            raise ValueError("network input H, W must be multiples of 32 (got %dx%d): 8x8 patches of the 1/4-scale "
                             "map are up-sampled x8 again" % (self.cropped_img_H, self.cropped_img_W))
        # training / testing
        self.seed = 2023
        self.train_batch_size = 8
        self.val_batch_size = 8
        self.val_interval = d["val_interval"]
        self.epoch = d["epoch"]
        self.lr = 0.001
        self.resume = False
        self.checkpoint = None
        self.num_workers = d["num_workers"]
        self.optimizer = "ADAM"
        self.momentum = 0.98
        self.weight_decay = 1e-06
        self.lr_scheduler = "StepLR"
        self.scheduler_gamma = 0.6
        self.step_size = d["step_size"]
        self.logdir = "log/"
        self.ckpt_dir = "checkpoint/"
        # image ViT
        self.image_H = int(self.cropped_img_H * 0.25)
        self.image_W = int(self.cropped_img_W * 0.25)
        self.patch_size = 8
        self.use_resnet_embedding = True
        self.embed_dim = 64
        self.mlp_dim = 1024
        self.embed_dropout = self.mlp_dropout = self.attention_dropout = 0.1
        self.num_sa_layer = 3
        self.num_head = 8
        # point ViT
        self.use_gnn_embedding = False
        self.point_feat_dim = 3
        self.num_node = 1280
        self.num_proxy = 256
        # coarse / fine I2P
        self.num_ca_layer_coarse = 6
        self.sinkhorn_iters = 100
        self.coarse_matching_thres = 0.01
        self.pt_sample_num = 65
        self.fine_dist_theshold = 1
        self.topk_proxy = 3
        self.pixel_positional_embedding = True
        self.fine_loss_weight = 0.5
        self.img_fuse_res_num = 2
        self.node_fuse_res_num = 2
        self.pt_head_res_num = 3
        self.linear_attention_num = 4
        self.LA_head_num = 8
        # agent
        self.is_6_DoF = False
        self.EXPERT_MODE = "steady"
        self.action_num = 10
        dev = default_device() if device is None else torch.device(device)
        deg = np.array([-62.5, -12.5, -2.5, -0.5, -0.1, 0.0, 0.1, 0.5, 2.5, 12.5, 62.5])
        self.r_steps = torch.from_numpy(deg * math.pi / 180).to(dev)            # float64, KittiConfig.py:105-108
        self.t_steps = torch.from_numpy(np.array([-8.1, -2.7, -0.9, -0.3, -0.1, 0.0, 0.1, 0.3, 0.9, 2.7, 8.1])).to(dev)
        self.num_steps = self.r_steps.shape[0]
        self.num_trajectory = 4
        self.GAMMA = 0.99
        self.GAE_LAMBDA = 0.95
        self.alpha = 1.0
        self.CLIP_EPS = 0.2
        self.W_VALUE = 0.3
        self.W_ENTROPY = 1e-3
        for k, v in overrides.items():
            if not hasattr(self, k):
                raise AttributeError("unknown configuration attribute %r" % k)
            setattr(self, k, v)


class KittiConfiguration(_Configuration):
    dataset_name = "kitti"
    _defaults = dict(dataset_root="/home/yao/workspace/I2P/kitti/", cropped_img_W=512, val_interval=500, epoch=64,
                     num_workers=12, step_size=4)

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.data_velodyne = "data_odometry_velodyne_NWU/"
        self.data_color = "data_odometry_color_npy/"


class NuScenesConfiguration(_Configuration):
    dataset_name = "nuscenes"
    _defaults = dict(dataset_root="/home/yao/workspace/I2P/nuscenes2/", cropped_img_W=320, val_interval=1000,
                     epoch=30, num_workers=16, step_size=2)
