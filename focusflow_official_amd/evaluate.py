"""Validation harness around the FF-RAFT module (host code; reference: ``core/models/ff-raft/evaluate.py``).

``validate_chairs`` (:18-45), ``validate_sintel`` (:48-86) and ``validate_kitti`` (:89-135) keep the reference's
metrics and result keys:

  * ``aepe``  — end-point error averaged over all pixels (KITTI: per image over valid pixels, then over images),
  * ``mepe``  — end-point error averaged over the key-point pixels (mask1 > 0.5) of each batch, then over batches
                (batches without key points are skipped, as the reference's NaN check does),
  * ``f1``    — KITTI outlier rate: epe > 3 px and epe / |gt| > 5 %, in percent over valid pixels.

The model is called exactly as the reference calls it — ``model(image1, image2, mask1, mask2, raft_iters=N,
test_mode=True)`` — so the HIP module and the reference module are interchangeable here.  The metric arithmetic
itself is a handful of reductions per image pair on whatever device the model runs on; it is harness code,
not part of the hot path.  Dataset roots default to the reference's relative paths and can be overridden.
"""
import numpy as np
import torch

from . import ops
import torch.utils.data as data

from . import datasets
from .utils import InputPadder


def _to_device(batch, device):
    return [x.to(device) for x in batch]


def _model_device(model):
    try:
        return next(model.parameters()).device
    except StopIteration:
        return torch.device("cpu")


class _Meter:
    """Accumulates the three reference metrics batch by batch."""

    def __init__(self):
        self.epe_all, self.mepe, self.epe_img, self.outliers = [], [], [], []

    def add(self, flow_pr, flow_gt, mask1, valid_gt=None):
        epe = torch.sum((flow_pr - flow_gt) ** 2, dim=1).sqrt().view(-1)
        key = (mask1 > 0.5).view(-1)
        if valid_gt is None:
            self.epe_all.append(epe.cpu().numpy())
            sel = key
        else:
            val = valid_gt.view(-1) >= 0.5
            mag = torch.sum(flow_gt ** 2, dim=1).sqrt().view(-1)
            out = ((epe > 3.0) & ((epe / mag) > 0.05)).float()
            self.epe_img.append(epe[val].mean().item())
            self.outliers.append(out[val].cpu().numpy())
            sel = key & val
        m = epe[sel].mean()
        if not torch.isnan(m):
            self.mepe.append(m.cpu().numpy())

    def dense(self):
        return float(np.mean(np.concatenate(self.epe_all))), float(np.mean(np.array(self.mepe)))

    def sparse(self):
        return (float(np.mean(np.array(self.epe_img))), float(100 * np.mean(np.concatenate(self.outliers))),
                float(np.mean(np.array(self.mepe))))


@torch.no_grad()
def evaluate_loader(model, loader, iters, pad_mode=None, sparse=False, device=None):
    """Run ``model`` over a loader of (image1, image2, flow_gt, mask1, mask2, valid) batches -> _Meter."""
    device = device or _model_device(model)
    meter = _Meter()
    for batch in loader:
        image1, image2, flow_gt, mask1, mask2, valid_gt = _to_device(batch, device)
        padder = None
        if pad_mode is not None:
            padder = InputPadder(image1.shape) if pad_mode == "sintel" else InputPadder(image1.shape, mode=pad_mode)
            image1, image2, mask1, mask2 = padder.pad(image1, image2, mask1, mask2)
        _, flow_pr = model(image1, image2, mask1, mask2, raft_iters=iters, test_mode=True)
        if padder is not None:
            flow_pr, mask1 = padder.unpad(flow_pr), padder.unpad(mask1)
        meter.add(flow_pr, flow_gt, mask1, valid_gt if sparse else None)
    ops.guard_check(sync=True)        # the always-on range guard looks at a forward when the next one starts: the last one here
    return meter


def _loader(ds, batch_size, workers):
    return data.DataLoader(ds, batch_size=batch_size, pin_memory=False, shuffle=False, num_workers=workers, drop_last=False)


def _count(model):
    print("Parameter Count: %d" % sum(p.numel() for p in model.parameters() if p.requires_grad))


@torch.no_grad()
def validate_chairs(model, mask_type, root="../../../data/FlyingChairs_release",
                    mask_root="../../../data/mask/FlyingChairs_release", workers=8):
    model.eval()
    _count(model)
    ds = datasets.FlyingChairs(root, mask_root, split="validation", mask_type=mask_type)
    aepe, mepe = evaluate_loader(model, _loader(ds, 1, workers), iters=12).dense()
    return {"chairs": aepe, f"chairs-{mask_type}": mepe}


@torch.no_grad()
def validate_sintel(model, mask_type, root="../../../data/Sintel-custom", mask_root="../../../data/mask/Sintel-custom",
                    workers=4):
    model.eval()
    _count(model)
    results = {}
    for dstype in ("clean", "final"):
        ds = datasets.MpiSintel(root, mask_root, dstype=dstype, mask_type=mask_type, split="val")
        aepe, mepe = evaluate_loader(model, _loader(ds, 4, workers), iters=32, pad_mode="sintel").dense()
        results[f"sintel-{dstype}"] = aepe
        results[f"sintel-{dstype}-{mask_type}"] = mepe
    return results


@torch.no_grad()
def validate_kitti(model, mask_type, root="../../../data/KITTI-custom", mask_root="../../../data/mask/KITTI-custom",
                   workers=8):
    model.eval()
    _count(model)
    ds = datasets.KITTI(root, mask_root, split="val", mask_type=mask_type)
    aepe, f1, mepe = evaluate_loader(model, _loader(ds, 1, workers), iters=32, pad_mode="kitti", sparse=True).sparse()
    return {"kitti-epe": aepe, "kitti-f1": f1, f"kitti-{mask_type}": mepe}
