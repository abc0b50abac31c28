"""Deterministic synthetic inputs.  Pure numpy, no reference code:
the golden generator, the tests and bench.py all build their inputs from these formulas so
every party regenerates identical bytes independently."""
import numpy as np


def lonlat(W, H):
    """Texel-centre longitudes/latitudes of the full globe in fp64."""
    lon = (np.arange(W, dtype=np.float64) + 0.5) / W * 2 * np.pi - np.pi
    lat = (np.arange(H, dtype=np.float64) + 0.5) / H * np.pi - np.pi / 2
    return lon, lat


def terrain(W, H):
    """Closed-form height field; land where > 0 (about 40 % land, continents cross the seam and
    touch both poles so the wrap and pole-reflection paths are exercised)."""
    # h = sin(3 lon) cos(2 lat) + 0.5 sin(7 lon + 1) sin(5 lat) - 0.2 in fp64, cast once.  Every factor depends on one
    # coordinate, so the sines are taken per column / per row and the field is assembled in row blocks that stay in cache:
    # the same fp64 products and sums in the same order as on a meshgrid (the committed goldens are built from it), 60x sooner
    # at 16384x8192 (0.4 s instead of 25 s -- bench.py builds this field in every process)
    lon, lat = lonlat(W, H)
    a, c = np.sin(3 * lon), 0.5 * np.sin(7 * lon + 1)
    b, d = np.cos(2 * lat), np.sin(5 * lat)
    out = np.empty((H, W), np.float32)
    R = 64
    t1, t2 = np.empty((R, W)), np.empty((R, W))
    for y in range(0, H, R):
        n = min(R, H - y)
        np.multiply(a[None, :], b[y:y + n, None], out=t1[:n])
        np.multiply(c[None, :], d[y:y + n, None], out=t2[:n])
        np.add(t1[:n], t2[:n], out=t1[:n])
        np.subtract(t1[:n], 0.2, out=t1[:n])
        out[y:y + n] = t1[:n]
    return out


def lcg_uniform(n, seed=12345):
    """n floats uniform in [-0.5, 0.5): s = s*1664525 + 1013904223 (mod 2^32), value (s>>8)/2^24 - 0.5."""
    B = 1 << 16
    a = np.uint64(1664525)
    c = np.uint64(1013904223)
    m = np.uint64(0xFFFFFFFF)
    # per-offset affine maps s -> A_k s + C_k for k = 1..B
    A = np.empty(B, np.uint64)
    C = np.empty(B, np.uint64)
    ak, ck = np.uint64(1), np.uint64(0)
    for k in range(B):
        ak = (ak * a) & m
        ck = (ck * a + c) & m
        A[k], C[k] = ak, ck
    out = np.empty(n, np.float32)
    s = np.uint64(seed)
    for i in range(0, n, B):
        k = min(B, n - i)
        blk = (A[:k] * s + C[:k]) & m
        out[i:i + k] = (blk >> np.uint64(8)).astype(np.float32) / np.float32(1 << 24) - np.float32(0.5)
        s = blk[k - 1]
    return out


def divw(W, H, seed=12345):
    return lcg_uniform(W * H, seed).reshape(H, W)


def velocity(W, H, seed=777, scale=1.0):
    """Smooth-ish RG32F velocity field (interleaved vx,vy), zero on land is NOT enforced here."""
    lon, lat = lonlat(W, H)
    lo, la = np.meshgrid(lon, lat)
    vx = scale * (np.cos(2 * la) * np.sin(lo + 0.3) + 0.3 * np.sin(5 * lo) * np.cos(3 * la))
    vy = scale * (0.7 * np.sin(2 * lo) * np.cos(la) + 0.2 * np.cos(4 * lo + 1) * np.sin(2 * la))
    noise = lcg_uniform(2 * W * H, seed).reshape(H, W, 2) * (0.1 * scale)
    return (np.stack([vx, vy], axis=-1) + noise).astype(np.float32)
