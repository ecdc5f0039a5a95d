"""Training-time augmentation for (image1, image2, flow, mask1, mask2) samples (host code).

Same parameters, probabilities and order of operations as the reference's ``core/utils/augmentor.py``
(FlowAugmentor :15-138, SparseFlowAugmentor :140-279): photometric jitter (asymmetric with p = 0.2 in the dense
case), occlusion "eraser" on image2, random scale / stretch (p = 0.8), flips, random crop; masks follow the
images through every spatial step; sparse flow maps are rescaled by scattering their valid vectors.

The reference leans on OpenCV (``cv2.resize``) and torchvision (``ColorJitter``); neither is in this image, so
the two primitives are written here on numpy: ``resize_linear`` follows cv2.INTER_LINEAR's half-pixel-centre
convention, ``ColorJitter`` follows torchvision's definitions (brightness/contrast/saturation blends, hue as a
rotation of the HSV hue) applied in a random order.  Random streams are numpy's, as in the reference.
"""
import numpy as np


def resize_linear(img, fx, fy):
    """Bilinear resize by factors (fx, fy), cv2.resize(..., None, fx, fy, INTER_LINEAR) convention:
    output size = round(size * f); source coordinate = (dst + 0.5) / f - 0.5, clamped to the image."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    ow, oh = max(1, int(round(w * fx))), max(1, int(round(h * fy)))

    def axis(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(s).astype(np.int64)
        t = s - i0
        lo, hi = np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1)
        t = np.where(i0 < 0, 0.0, t)
        return lo, hi, t

    y0, y1, ty = axis(oh, h)
    x0, x1, tx = axis(ow, w)
    src = img.astype(np.float64)
    if src.ndim == 2:
        ty_, tx_ = ty[:, None], tx[None, :]
    else:
        ty_, tx_ = ty[:, None, None], tx[None, :, None]
    top = src[y0][:, x0] * (1 - tx_) + src[y0][:, x1] * tx_
    bot = src[y1][:, x0] * (1 - tx_) + src[y1][:, x1] * tx_
    out = top * (1 - ty_) + bot * ty_
    if np.issubdtype(img.dtype, np.integer):
        return np.clip(np.rint(out), np.iinfo(img.dtype).min, np.iinfo(img.dtype).max).astype(img.dtype)
    return out.astype(img.dtype)


def _gray(rgb):
    return rgb[..., 0] * 0.299 + rgb[..., 1] * 0.587 + rgb[..., 2] * 0.114


def _rgb_to_hsv(rgb):
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    mx, mn = rgb.max(-1), rgb.min(-1)
    d = mx - mn
    s = np.where(mx > 0, d / np.where(mx > 0, mx, 1), 0.0)
    dd = np.where(d > 0, d, 1)
    hr = ((g - b) / dd) % 6
    hg = (b - r) / dd + 2
    hb = (r - g) / dd + 4
    h = np.where(mx == r, hr, np.where(mx == g, hg, hb))
    h = np.where(d > 0, h / 6.0, 0.0)
    return h, s, mx


def _hsv_to_rgb(h, s, v):
    i = np.floor(h * 6.0)
    f = h * 6.0 - i
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    i = i.astype(np.int64) % 6
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.stack([r, g, b], -1)


class ColorJitter:
    """Brightness / contrast / saturation factors ~ U(1 - a, 1 + a), hue shift ~ U(-h, h) (fraction of the hue
    circle), the four operations applied in a random order — torchvision.transforms.ColorJitter's semantics."""

    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0):
        self.brightness, self.contrast, self.saturation, self.hue = brightness, contrast, saturation, hue

    def __call__(self, img):
        x = np.asarray(img, dtype=np.float64) / 255.0
        fb = np.random.uniform(max(0.0, 1 - self.brightness), 1 + self.brightness)
        fc = np.random.uniform(max(0.0, 1 - self.contrast), 1 + self.contrast)
        fs = np.random.uniform(max(0.0, 1 - self.saturation), 1 + self.saturation)
        fh = np.random.uniform(-self.hue, self.hue)
        for op in np.random.permutation(4):
            if op == 0:
                x = x * fb
            elif op == 1:
                x = (x - _gray(x).mean()) * fc + _gray(x).mean()
            elif op == 2:
                g = _gray(x)[..., None]
                x = (x - g) * fs + g
            else:
                h, s, v = _rgb_to_hsv(np.clip(x, 0, 1))
                x = _hsv_to_rgb((h + fh) % 1.0, s, v)
            x = np.clip(x, 0.0, 1.0)
        return np.rint(x * 255.0).astype(np.uint8)


class _Augmentor:
    sparse = False

    def __init__(self, crop_size, min_scale=-0.2, max_scale=0.5, do_flip=True):
        self.crop_size = crop_size
        self.min_scale, self.max_scale = min_scale, max_scale
        self.spatial_aug_prob, self.stretch_prob, self.max_stretch = 0.8, 0.8, 0.2
        self.do_flip, self.h_flip_prob, self.v_flip_prob = do_flip, 0.5, 0.1
        j = 0.3 if self.sparse else 0.4
        self.photo_aug = ColorJitter(brightness=j, contrast=j, saturation=j, hue=(0.3 if self.sparse else 0.5) / 3.14)
        self.asymmetric_color_aug_prob, self.eraser_aug_prob = 0.2, 0.5

    # -- photometric -------------------------------------------------------------------------------------------
    def color_transform(self, img1, img2):
        if not self.sparse and np.random.rand() < self.asymmetric_color_aug_prob:
            return self.photo_aug(img1), self.photo_aug(img2)
        both = self.photo_aug(np.concatenate([img1, img2], axis=0))
        return both[:img1.shape[0]], both[img1.shape[0]:]

    def eraser_transform(self, img1, img2, bounds=(50, 100)):
        ht, wd = img1.shape[:2]
        if np.random.rand() < self.eraser_aug_prob:
            img2 = img2.copy()
            mean_color = img2.reshape(-1, 3).mean(axis=0)
            for _ in range(np.random.randint(1, 3)):
                x0, y0 = np.random.randint(0, wd), np.random.randint(0, ht)
                dx, dy = np.random.randint(bounds[0], bounds[1]), np.random.randint(bounds[0], bounds[1])
                img2[y0:y0 + dy, x0:x0 + dx, :] = mean_color
        return img1, img2

    # -- spatial -----------------------------------------------------------------------------------------------
    @staticmethod
    def resize_sparse_flow_map(flow, valid, fx=1.0, fy=1.0):
        """Scatter the valid vectors of a sparse flow map to their rounded scaled positions (augmentor.py:187-220)."""
        ht, wd = flow.shape[:2]
        yy0, xx0 = np.nonzero(np.asarray(valid).reshape(ht, wd) >= 1)
        ht1, wd1 = int(round(ht * fy)), int(round(wd * fx))
        xx = np.round(xx0.astype(np.float32) * fx).astype(np.int32)
        yy = np.round(yy0.astype(np.float32) * fy).astype(np.int32)
        keep = (xx > 0) & (xx < wd1) & (yy > 0) & (yy < ht1)
        flow_img = np.zeros([ht1, wd1, 2], dtype=np.float32)
        valid_img = np.zeros([ht1, wd1], dtype=np.int32)
        flow_img[yy[keep], xx[keep]] = flow[yy0[keep], xx0[keep]].astype(np.float32) * [fx, fy]
        valid_img[yy[keep], xx[keep]] = 1
        return flow_img, valid_img

    def spatial_transform(self, img1, img2, flow, valid, mask1, mask2):
        ht, wd = img1.shape[:2]
        slack = 1 if self.sparse else 8
        min_scale = max((self.crop_size[0] + slack) / float(ht), (self.crop_size[1] + slack) / float(wd))
        scale = 2 ** np.random.uniform(self.min_scale, self.max_scale)
        scale_x = scale_y = scale
        if not self.sparse and np.random.rand() < self.stretch_prob:
            scale_x *= 2 ** np.random.uniform(-self.max_stretch, self.max_stretch)
            scale_y *= 2 ** np.random.uniform(-self.max_stretch, self.max_stretch)
        scale_x, scale_y = max(scale_x, min_scale), max(scale_y, min_scale)

        if np.random.rand() < self.spatial_aug_prob:
            img1, img2 = resize_linear(img1, scale_x, scale_y), resize_linear(img2, scale_x, scale_y)
            mask1, mask2 = resize_linear(mask1, scale_x, scale_y), resize_linear(mask2, scale_x, scale_y)
            if self.sparse:
                flow, valid = self.resize_sparse_flow_map(flow, valid, fx=scale_x, fy=scale_y)
            else:
                flow = resize_linear(flow, scale_x, scale_y) * [scale_x, scale_y]

        def flip(ax, sign):
            nonlocal img1, img2, flow, valid, mask1, mask2
            sl = (slice(None), slice(None, None, -1)) if ax == 1 else (slice(None, None, -1),)
            img1, img2, mask1, mask2 = img1[sl], img2[sl], mask1[sl], mask2[sl]
            flow = flow[sl] * sign
            if valid is not None:
                valid = valid[sl]

        if self.do_flip:
            if np.random.rand() < self.h_flip_prob:
                flip(1, [-1.0, 1.0])
            if not self.sparse and np.random.rand() < self.v_flip_prob:
                flip(0, [1.0, -1.0])

        ch, cw = self.crop_size
        y0 = np.random.randint(0, img1.shape[0] - ch) if img1.shape[0] > ch else 0
        x0 = np.random.randint(0, img1.shape[1] - cw) if img1.shape[1] > cw else 0
        win = (slice(y0, y0 + ch), slice(x0, x0 + cw))
        return (img1[win], img2[win], flow[win], None if valid is None else valid[win], mask1[win], mask2[win])

    def _run(self, img1, img2, flow, valid, mask1, mask2):
        img1, img2 = self.color_transform(img1, img2)
        img1, img2 = self.eraser_transform(img1, img2)
        out = self.spatial_transform(img1, img2, flow, valid, mask1, mask2)
        img1, img2, flow, valid, mask1, mask2 = [None if a is None else np.ascontiguousarray(a) for a in out]
        if mask1.ndim == 2:
            mask1 = mask1[:, :, None]
        if mask2.ndim == 2:
            mask2 = mask2[:, :, None]
        return img1, img2, flow, valid, mask1, mask2


class FlowAugmentor(_Augmentor):
    """Dense ground truth: ``aug(img1, img2, flow, mask1, mask2) -> (img1, img2, flow, mask1, mask2)``."""
    sparse = False

    def __call__(self, img1, img2, flow, mask1, mask2):
        img1, img2, flow, _, mask1, mask2 = self._run(img1, img2, flow, None, mask1, mask2)
        return img1, img2, flow, mask1, mask2


class SparseFlowAugmentor(_Augmentor):
    """Sparse ground truth (KITTI): ``aug(img1, img2, flow, valid, mask1, mask2) -> (..., flow, valid, ...)``."""
    sparse = True

    def __init__(self, crop_size, min_scale=-0.2, max_scale=0.5, do_flip=False):
        super().__init__(crop_size, min_scale, max_scale, do_flip)

    def __call__(self, img1, img2, flow, valid, mask1, mask2):
        return self._run(img1, img2, flow, valid, mask1, mask2)
