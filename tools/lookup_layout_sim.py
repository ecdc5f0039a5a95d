"""HBM lines a correlation lookup touches per query, by pyramid layout - a pricing of the structural alternatives to the shipped
tiled layout (csrc/corr_layout.h), without a kernel.  The lookup kernel already runs at the speed a memory-only kernel reaches on
the SAME lines (bench.py roofline.memory_only_kernel), so what is left is the number of lines.

Every query (a pixel of frame 1 at 1/8 resolution) owns a plane of h_l x w_l correlation values per level; one lookup reads the
10 x 10 window around its current target (9 x 9 taps, bilinear: corr.py:29-50).  Algorithmic bytes: 4 levels x 100 x 4 B = 1600 B
read per query.  HBM moves whole 128-byte lines:

  rowmajor     the reference's own layout: a window row is 40 B in a row of w_l floats
  tile 8x4     shipped: 128-byte tiles of 8 x 4 floats of ONE query's plane
  tile 4x8 / 16x2   other tile shapes of one query's plane
  quad 4x2     a line holds 4 x 2 positions of FOUR neighbouring queries (2 x 2): neighbours' windows overlap when the flow is smooth
  oct 2x2      a line holds 2 x 2 positions of EIGHT neighbouring queries (4 x 2)
  q32          a line holds ONE position of 32 neighbouring queries (8 x 4): the limit of that idea

A group of queries fetches the union of its members' windows; the cost per query is lines x 128 B / group size.  The coherence of
neighbouring targets is the unknown: sigma = standard deviation (in 1/8-resolution pixels, level 0) of the difference between the
targets of two neighbouring queries.  0 = rigid translation, 0.1-0.3 = smooth real flow, >= 1 = the noise fields a randomly
initialised network produces (bench.py's synthetic run).

    python tools/lookup_layout_sim.py            (CPU, numpy, a few seconds)
"""
import numpy as np

H0, W0 = 48, 64           # 384 x 512 at 1/8
RNG = np.random.default_rng(0)


def windows(cx, cy, lvl):
    """Integer window [y0, y0 + 10) x [x0, x0 + 10) of a lookup centred at (cx, cy) (level-0 units) on level lvl, clipped to the plane."""
    h, w = H0 >> lvl, W0 >> lvl
    x = cx / (1 << lvl)
    y = cy / (1 << lvl)
    x0 = np.floor(x).astype(int) - 4
    y0 = np.floor(y).astype(int) - 4
    return np.clip(x0, -1000, 1000), np.clip(y0, -1000, 1000), h, w


def lines_single(x0, y0, h, w, tw, th):
    """Lines of one query's own plane (tiles tw x th, 128 B) touched by the window - in-plane part only."""
    xa, xb = np.clip(x0, 0, w), np.clip(x0 + 10, 0, w)
    ya, yb = np.clip(y0, 0, h), np.clip(y0 + 10, 0, h)
    nx = np.where(xb > xa, (xb - 1) // tw - xa // tw + 1, 0)
    ny = np.where(yb > ya, (yb - 1) // th - ya // th + 1, 0)
    return nx * ny


def lines_rowmajor(x0, y0, h, w):
    xa, xb = np.clip(x0, 0, w), np.clip(x0 + 10, 0, w)
    ya, yb = np.clip(y0, 0, h), np.clip(y0 + 10, 0, h)
    n = np.zeros(x0.shape, dtype=np.int64)
    for r in range(10):
        yy = y0 + r
        ok = (yy >= ya) & (yy < yb) & (xb > xa)
        a = (np.clip(yy, 0, h - 1) * w + xa) * 4 // 128
        b = (np.clip(yy, 0, h - 1) * w + xb - 1) * 4 // 128
        n += np.where(ok, b - a + 1, 0)      # (rows of w < 32 floats share lines: an over-count of at most 2x on levels 2-3, which are small)
    return n


def lines_group(x0, y0, h, w, gw, gh, tw, th):
    """Queries in gw x gh groups (of the H0 x W0 query grid) share lines that hold tw x th positions of every member: lines of the UNION
    of the members' windows, per group."""
    qh, qw = x0.shape
    tot = 0
    for gy in range(0, qh, gh):
        for gx in range(0, qw, gw):
            xs, ys = x0[gy:gy + gh, gx:gx + gw].ravel(), y0[gy:gy + gh, gx:gx + gw].ravel()
            seen = set()
            for a, b in zip(xs, ys):
                xa, xb, ya, yb = max(a, 0), min(a + 10, w), max(b, 0), min(b + 10, h)
                if xb <= xa or yb <= ya:
                    continue
                for ty in range(ya // th, (yb - 1) // th + 1):
                    for tx in range(xa // tw, (xb - 1) // tw + 1):
                        seen.add((ty, tx))
            tot += len(seen)
    return tot


def main():
    yy, xx = np.mgrid[0:H0, 0:W0].astype(np.float64)
    print(f"{'sigma':>6} {'layout':>10} " + " ".join(f"{'L' + str(l):>7}" for l in range(4)) + f" {'B/query':>9} {'x algorithmic (1600 B)':>24}")
    for sigma in (0.0, 0.1, 0.3, 1.0, 4.0):
        # a smooth base flow (a few pixels at 1/8 resolution) + white noise whose neighbour DIFFERENCE has the requested deviation
        base_x = 3.0 * np.sin(yy / 17.0) + 2.0 * np.cos(xx / 23.0) + 1.7
        base_y = 2.0 * np.cos(yy / 13.0 + xx / 31.0) - 0.6
        nz = sigma / np.sqrt(2.0)
        cx = xx + base_x + RNG.normal(0, nz, xx.shape) if sigma else xx + base_x
        cy = yy + base_y + RNG.normal(0, nz, yy.shape) if sigma else yy + base_y
        rows = {}
        for lvl in range(4):
            x0, y0, h, w = windows(cx, cy, lvl)
            nq = x0.size
            rows.setdefault("rowmajor", []).append(lines_rowmajor(x0, y0, h, w).sum() / nq)
            rows.setdefault("tile 8x4", []).append(lines_single(x0, y0, h, w, 8, 4).sum() / nq)
            rows.setdefault("tile 4x8", []).append(lines_single(x0, y0, h, w, 4, 8).sum() / nq)
            rows.setdefault("tile 16x2", []).append(lines_single(x0, y0, h, w, 16, 2).sum() / nq)
            rows.setdefault("quad 4x2", []).append(lines_group(x0, y0, h, w, 2, 2, 4, 2) / nq)
            rows.setdefault("oct 2x2", []).append(lines_group(x0, y0, h, w, 4, 2, 2, 2) / nq)
            rows.setdefault("q32", []).append(lines_group(x0, y0, h, w, 8, 4, 1, 1) / nq)
        for name, per in rows.items():
            byt = sum(per) * 128
            print(f"{sigma:6.1f} {name:>10} " + " ".join(f"{v:7.2f}" for v in per) + f" {byt:9.0f} {byt / 1600:24.2f}")
        print()
    print("measured (profiles/r05_lookup_traffic.json, bench.py's synthetic run, tile 8x4): 2304 B read per query")


if __name__ == "__main__":
    main()
