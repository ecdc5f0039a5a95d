"""Hand-derived known-answer cases for the FF-PWC cost volume (correlation.py:34-102 of the reference, CuPy: cannot run here).

No restatement of the kernel is involved: every expected tensor below is written down from the formula the reference's kernel
text states, so that the two restatements in oracle/ (which could share a misreading) and the HIP kernel are pinned on
channel ORDER, displacement SIGN and BORDER handling independently.

The formula (correlation.py:46-49 centre pixel, :72-73 channel -> displacement, :79-86 product, :97-98 normalisation; the
rearrange kernel :7-31 zero-pads both inputs by 4 pixels):

    top[b, ch, y, x] = 1/C * sum_c one[b, c, y, x] * two_padded[b, c, y + s2p, x + s2o]
    s2o = ch % 9 - 4   (x displacement: the FAST index of ch)        s2p = ch // 9 - 4   (y displacement: the SLOW index)

Case A - one-hot features.  one = a at (c0, y0, x0), two = b at (c0, y1, x1), zero elsewhere, C channels.  The sum has one
non-zero term, at output pixel (y0, x0) and the channel whose displacement is (y1 - y0, x1 - x0):
    ch* = (y1 - y0 + 4) * 9 + (x1 - x0 + 4),   top[ch*, y0, x0] = a b / C,   everything else exactly 0.
Three instances: displacement (+2, -2) -> ch 56; (-4, +4) -> ch 8 (a corner of the 9 x 9 window); (0, 0) -> ch 40.  A fourth
with |dy| = 5 is outside the window: the volume is all zero.

Case B - borders.  one = two = 1 everywhere, C = 2: every product is 1 where the displaced pixel lies inside the image and 0
where it falls into the padding, so top[ch, y, x] = [0 <= y + s2p < H] * [0 <= x + s2o < W]  (C / C = 1).

Case C - sign and axis.  one = 1, two[c, y, x] = x + 100 y (C = 3): top[ch, y, x] = (x + s2o) + 100 (y + s2p) inside the
image, 0 in the padding: ch -> ch + 1 moves +1 in x, ch -> ch + 9 moves +1 in y, ch = 40 reproduces `two`.

    python tests/golden/make_golden_pwc_known_answers.py      ->  tests/golden/pwc_costvolume_known.npz
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def one_hot_case(C, H, W, c0, p0, p1, a, b):
    one = np.zeros((1, C, H, W), np.float32)
    two = np.zeros((1, C, H, W), np.float32)
    one[0, c0, p0[0], p0[1]] = a
    two[0, c0, p1[0], p1[1]] = b
    top = np.zeros((1, 81, H, W), np.float32)
    dy, dx = p1[0] - p0[0], p1[1] - p0[1]
    if abs(dy) <= 4 and abs(dx) <= 4:
        top[0, (dy + 4) * 9 + (dx + 4), p0[0], p0[1]] = np.float32(a) * np.float32(b) / np.float32(C)
    return one, two, top


def main():
    out = {}
    for name, args in {"A_dy+2_dx-2": (4, 6, 7, 1, (2, 3), (4, 1), 3.0, -0.5), "A_corner_dy-4_dx+4": (8, 9, 10, 5, (6, 2), (2, 6), 2.0, 4.0),
                       "A_centre": (4, 5, 5, 0, (2, 2), (2, 2), 1.5, 2.0), "A_outside_dy+5": (4, 8, 6, 2, (1, 3), (6, 3), 1.0, 1.0)}.items():
        one, two, top = one_hot_case(*args)
        out[name + ".one"], out[name + ".two"], out[name + ".top"] = one, two, top
    H, W = 7, 11
    ys, xs = np.arange(H)[:, None], np.arange(W)[None, :]
    topB = np.zeros((1, 81, H, W), np.float32)
    topC = np.zeros((1, 81, H, W), np.float32)
    for ch in range(81):
        s2o, s2p = ch % 9 - 4, ch // 9 - 4
        inside = ((ys + s2p >= 0) & (ys + s2p < H) & (xs + s2o >= 0) & (xs + s2o < W))
        topB[0, ch] = inside.astype(np.float32)
        topC[0, ch] = np.where(inside, (xs + s2o) + 100.0 * (ys + s2p), 0.0).astype(np.float32)
    out["B_ones.one"] = out["B_ones.two"] = np.ones((1, 2, H, W), np.float32)
    out["B_ones.top"] = topB
    out["C_ramp.one"] = np.ones((1, 3, H, W), np.float32)
    out["C_ramp.two"] = np.broadcast_to((xs + 100.0 * ys).astype(np.float32), (1, 3, H, W)).copy()
    out["C_ramp.top"] = topC
    np.savez_compressed(os.path.join(HERE, "pwc_costvolume_known.npz"), **out)
    print("wrote", len(out) // 3, "cases")


if __name__ == "__main__":
    main()
