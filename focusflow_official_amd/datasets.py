"""Dataset indexes and loaders for (image pair, flow, key-point mask pair) samples (host code).

Same classes, constructor arguments, directory conventions and sample tuple as the reference's
``core/datasets.py`` (FlowDataset :21-124, MpiSintel :126-144, FlyingChairs :147-162, FlyingThings3D :165-194,
KITTI :197-217, HD1K :220-237, OminiFlow :240-257, fetch_dataloader :260-315), so the reference's train /
evaluate scripts can import this module instead.  A sample is

    image1, image2 (3,H,W) float32 in [0,255];  flow (2,H,W);  mask1, mask2 (1,H,W) float32 in [0,255];  valid (H,W)

Each dataset only builds three parallel lists (image pairs, mask pairs, flow files); reading, augmentation and
tensor conversion live in the base class.
"""
import os
import os.path as osp
import random
from glob import glob

import numpy as np
import torch
import torch.utils.data as data
from torch.utils.data import DataLoader, distributed

from . import frame_utils
from .augmentor import FlowAugmentor, SparseFlowAugmentor


def _files(*parts):
    return sorted(glob(osp.join(*parts)))


def _first_channel(m):
    m = np.array(m).astype(np.uint8)
    return m[..., None] if m.ndim == 2 else m[..., :1]


class FlowDataset(data.Dataset):
    def __init__(self, aug_params=None, sparse=False):
        self.sparse = sparse
        self.augmentor = None
        if aug_params is not None:
            self.augmentor = (SparseFlowAugmentor if sparse else FlowAugmentor)(**aug_params)
        self.is_test = False
        self.init_seed = False
        self.flow_list, self.image_list, self.mask_list, self.extra_info = [], [], [], []

    def _seed_worker_once(self):
        if self.init_seed:
            return
        info = torch.utils.data.get_worker_info()
        if info is not None:  # every loader worker gets its own, reproducible random stream
            torch.manual_seed(info.id)
            np.random.seed(info.id)
            random.seed(info.id)
            self.init_seed = True

    def __getitem__(self, index):
        img1, img2 = (np.array(frame_utils.read_gen(f)).astype(np.uint8) for f in self.image_list[index])
        mask1, mask2 = (_first_channel(frame_utils.read_gen(f)) for f in self.mask_list[index])

        if self.is_test:  # no ground truth: images as tensors, masks as read, plus the frame id
            img1, img2 = (torch.from_numpy(i[..., :3].copy()).permute(2, 0, 1).float() for i in (img1, img2))
            return img1, img2, mask1, mask2, self.extra_info[index]

        self._seed_worker_once()
        index = index % len(self.image_list)
        valid = None
        if self.sparse:
            flow, valid = frame_utils.readFlowKITTI(self.flow_list[index])
        else:
            flow = frame_utils.read_gen(self.flow_list[index])
        flow = np.array(flow).astype(np.float32)

        if img1.ndim == 2:  # gray images -> 3 equal channels
            img1, img2 = (np.tile(i[..., None], (1, 1, 3)) for i in (img1, img2))
        else:
            img1, img2 = img1[..., :3], img2[..., :3]

        if self.augmentor is not None:
            if self.sparse:
                img1, img2, flow, valid, mask1, mask2 = self.augmentor(img1, img2, flow, valid, mask1, mask2)
            else:
                img1, img2, flow, mask1, mask2 = self.augmentor(img1, img2, flow, mask1, mask2)

        def chw(a):
            return torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).float()

        img1, img2, flow, mask1, mask2 = chw(img1), chw(img2), chw(flow), chw(mask1), chw(mask2)
        if valid is not None:
            valid = torch.from_numpy(np.ascontiguousarray(valid))
        else:
            valid = (flow[0].abs() < 1000) & (flow[1].abs() < 1000)
        return img1, img2, flow, mask1, mask2, valid.float()

    def __rmul__(self, v):
        self.flow_list = v * self.flow_list
        self.image_list = v * self.image_list
        self.mask_list = v * self.mask_list
        return self

    def __len__(self):
        return len(self.image_list)

    def _add_sequence(self, images, masks, flows=None, info=None):
        """Consecutive frames of one scene: pair (i, i+1) with flow i."""
        for i in range(len(images) - 1):
            self.image_list.append([images[i], images[i + 1]])
            self.mask_list.append([masks[i], masks[i + 1]])
            if info is not None:
                self.extra_info.append((info, i))
        if flows is not None:
            self.flow_list += flows


class MpiSintel(FlowDataset):
    """<root>/<split>/{clean,final,flow}/<scene>/frame_*.png|.flo ; masks <mask_root>/<mask_type>/<split>/<dstype>/<scene>/."""

    def __init__(self, root, mask_root, aug_params=None, split="training", dstype="clean", mask_type="orb"):
        super().__init__(aug_params)
        image_root = osp.join(root, split, dstype)
        self.is_test = split == "testing"
        for scene in os.listdir(image_root):
            self._add_sequence(_files(image_root, scene, "*.png"),
                               _files(mask_root, mask_type, split, dstype, scene, "*.png"),
                               _files(root, split, "flow", scene, "*.flo") if split != "test" else None, info=scene)


class FlyingChairs(FlowDataset):
    """<root>/data/{*.ppm,*.flo} + FlyingChairs_train_val.txt (1 = train, 2 = validation); masks <mask_root>/<mask_type>/*.png."""

    def __init__(self, root, mask_root, aug_params=None, split="training", mask_type="orb"):
        super().__init__(aug_params)
        images, flows = _files(root, "data/*.ppm"), _files(root, "data/*.flo")
        masks = _files(mask_root, mask_type, "*.png")
        assert len(images) == len(masks)
        assert len(images) // 2 == len(flows)
        wanted = {"training": 1, "validation": 2}.get(split)
        labels = np.loadtxt(osp.join(root, "FlyingChairs_train_val.txt"), dtype=np.int32).reshape(-1)
        for i, flow in enumerate(flows):
            if labels[i] == wanted:
                self.flow_list.append(flow)
                self.image_list.append([images[2 * i], images[2 * i + 1]])
                self.mask_list.append([masks[2 * i], masks[2 * i + 1]])


class FlyingThings3D(FlowDataset):
    """TRAIN split, left camera, both time directions (into_past pairs are (i+1, i) with flow i+1)."""

    def __init__(self, root, mask_root, aug_params=None, dstype="frames_cleanpass", mask_type="orb"):
        super().__init__(aug_params)
        cam = "left"
        image_dirs = sorted(osp.join(d, cam) for d in _files(root, dstype, "TRAIN/*/*"))
        mask_dirs = sorted(osp.join(d, cam) for d in _files(mask_root, mask_type, dstype, "TRAIN/*/*"))
        for direction in ("into_future", "into_past"):
            flow_dirs = sorted(osp.join(d, direction, cam) for d in _files(root, "optical_flow/TRAIN/*/*"))
            for idir, fdir, mdir in zip(image_dirs, flow_dirs, mask_dirs):
                images, flows, masks = _files(idir, "*.png"), _files(fdir, "*.pfm"), _files(mdir, "*.png")
                for i in range(len(flows) - 1):
                    a, b = (i, i + 1) if direction == "into_future" else (i + 1, i)
                    self.image_list.append([images[a], images[b]])
                    self.mask_list.append([masks[a], masks[b]])
                    self.flow_list.append(flows[a])


class KITTI(FlowDataset):
    """<root>/<split>/image_2/*_10.png, *_11.png, flow_occ/*_10.png (sparse) ; masks <mask_root>/<mask_type>/<split>/."""

    def __init__(self, root, mask_root, aug_params=None, split="training", mask_type="orb"):
        super().__init__(aug_params, sparse=True)
        self.is_test = split == "testing"
        image_root, mroot = osp.join(root, split), osp.join(mask_root, mask_type, split)
        for img1, img2, m1, m2 in zip(_files(image_root, "image_2/*_10.png"), _files(image_root, "image_2/*_11.png"),
                                      _files(mroot, "*_10.png"), _files(mroot, "*_11.png")):
            self.extra_info.append([img1.split("/")[-1]])
            self.image_list.append([img1, img2])
            self.mask_list.append([m1, m2])
        self.flow_list = _files(image_root, "flow_occ/*_10.png")   # the reference lists it for every split


class HD1K(FlowDataset):
    """Sparse ground truth, no key-point masks shipped (the reference leaves mask_list empty as well)."""

    def __init__(self, root="datasets/HD1k", aug_params=None):
        super().__init__(aug_params, sparse=True)
        seq = 0
        while True:
            flows = _files(root, "hd1k_flow_gt", "flow_occ/%06d_*.png" % seq)
            images = _files(root, "hd1k_input", "image_2/%06d_*.png" % seq)
            if not flows:
                break
            for i in range(len(flows) - 1):
                self.flow_list.append(flows[i])
                self.image_list.append([images[i], images[i + 1]])
            seq += 1


class OminiFlow(FlowDataset):
    """Three synthetic scenes x two takes; the images double as their own masks (datasets.py:240-257)."""

    def __init__(self, root, aug_params=None):
        super().__init__(aug_params)
        for scene in ("CartoonTree", "Forest", "lowPolyModels"):
            for take in (scene, f"{scene}_1"):
                images = _files(root, scene, take, "images/*.png")
                flows = _files(root, scene, take, "ground_truth/*.flo")
                for i in range(len(images) - 1):
                    self.image_list.append([images[i], images[i + 1]])
                    self.mask_list.append([images[i], images[i + 1]])
                    self.flow_list.append(flows[i])


def _training_set(data_root, mask_root, cfg, TRAIN_DS):
    size, mt, stage = cfg.TRAIN.IMAGE_SIZE, cfg.TRAIN.MASK_TYPE, cfg.TRAIN.STAGE

    def aug(lo, hi, flip=True):
        return {"crop_size": size, "min_scale": lo, "max_scale": hi, "do_flip": flip}

    def sintel(dstype, lo=-0.2, hi=0.6):
        return MpiSintel(data_root["sintel"], mask_root["sintel"], dstype=dstype, aug_params=aug(lo, hi), mask_type=mt)

    def things(dstype, lo, hi):
        return FlyingThings3D(data_root["things"], mask_root["things"], dstype=dstype, aug_params=aug(lo, hi), mask_type=mt)

    def kitti():
        return KITTI(data_root["kitti"], mask_root["kitti"], split="training", aug_params=aug(-0.3, 0.5), mask_type=mt)

    if stage == "chairs":
        return FlyingChairs(data_root["chairs"], mask_root["chairs"], aug_params=aug(-0.1, 1.0), split="training", mask_type=mt)
    if stage == "things":
        return things("frames_cleanpass", -0.4, 0.8) + things("frames_finalpass", -0.4, 0.8)
    if stage == "sintel":
        clean, final = sintel("clean"), sintel("final")
        if TRAIN_DS == "C+T+S":
            return 100 * clean + 100 * final + things("frames_cleanpass", -0.2, 0.6)
        if TRAIN_DS == "C+T+S+K":
            return things("frames_cleanpass", -0.2, 0.6) + 100 * clean + 100 * final + 200 * kitti()
        return clean + final
    if stage == "kitti":
        return 100 * sintel("clean") + 100 * sintel("final") + 200 * kitti()
    raise ValueError(f"unknown training stage {stage!r}")


def fetch_dataloader(data_root, mask_root, cfg, rank=-1, world_size=1, TRAIN_DS=None):
    """Training loader for cfg.TRAIN.STAGE; one DistributedSampler shard per rank (datasets.py:260-315)."""
    train_dataset = _training_set(data_root, mask_root, cfg, TRAIN_DS)
    sampler = None if rank == -1 else distributed.DistributedSampler(train_dataset, shuffle=True)
    loader = DataLoader(train_dataset, batch_size=cfg.TRAIN.BATCH_SIZE // world_size, pin_memory=True,
                        shuffle=sampler is None, sampler=sampler, num_workers=cfg.GLOBAL.NUM_WORKERS, drop_last=True)
    print("Training with %d image pairs" % len(train_dataset))
    return loader
