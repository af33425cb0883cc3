"""Seeded synthetic Kinect-v2-like inputs for the fusion path (tests, smoke, bench).

Everything is a pure function of (seed, tick, sensor, pixel) so fixtures can be regenerated anywhere and
hash-checked.  Layouts are the ones KinectServer.CopyLatestFrames packs for the native call
(LiveScanServer/KinectServer.cs:453-498): depth = concatenated little-endian u16 [h][w] per sensor,
colours = concatenated RGB8 [h][w][3], intrinsics 7 floats/sensor {cx,cy,fx,fy,r2,r4,r6},
world transform 12 floats/sensor {t[3], R[3][3] row-major} in the reference's p' = R (p + t) convention
(src/NativeUtils/depthprocessing.cpp:157-160).
"""
import math

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _pixel_keys(seed, tick, sensor, w, h):
    y, x = np.meshgrid(np.arange(h, dtype=np.uint64), np.arange(w, dtype=np.uint64), indexing="ij")
    key = (np.uint64(seed) << np.uint64(52)) ^ (np.uint64(tick) << np.uint64(36)) ^ \
          (np.uint64(sensor) << np.uint64(28)) ^ (y << np.uint64(14)) ^ x
    return splitmix64(key)


def noise_frame(seed, tick, sensor, w=512, h=424):
    """Hash-noise frame: ~10 % invalid (0) pixels, depth 500..4499 mm, random colours.
    Returns (depth u16 [h,w], rgb u8 [h,w,3])."""
    hsh = _pixel_keys(seed, tick, sensor, w, h)
    k = hsh >> np.uint64(11)
    depth = (np.uint64(500) + (k >> np.uint64(4)) % np.uint64(4000)).astype(np.uint16)
    depth[(k % np.uint64(10)) == 0] = 0
    c = splitmix64(hsh)
    rgb = np.stack([(c & np.uint64(0xFF)), ((c >> np.uint64(8)) & np.uint64(0xFF)),
                    ((c >> np.uint64(16)) & np.uint64(0xFF))], axis=-1).astype(np.uint8)
    return depth, rgb


def noise_frames_torch(device, seed, n_ticks, n_sensors, w=512, h=424, sensor0=0, tick0=0):
    """Same generator as noise_frame (ticks tick0.., sensors sensor0..), evaluated with torch int64 ops on `device`.

    torch has no uint16/uint64 arithmetic, so the hash runs in wrapping int64 with logical shifts emulated;
    depth is returned as torch.int16 holding the u16 bit pattern (values < 4500 so they are non-negative),
    rgb as torch.uint8 [n_ticks, n_sensors, h, w, 3]."""
    import torch

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    def c64(v):  # python int -> signed 64-bit constant
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v

    def sm64(x):
        z = x + c64(0x9E3779B97F4A7C15)
        z = (z ^ lsr(z, 30)) * c64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * c64(0x94D049BB133111EB)
        return z ^ lsr(z, 31)

    depth = torch.empty((n_ticks, n_sensors, h, w), dtype=torch.int16, device=device)
    rgb = torch.empty((n_ticks, n_sensors, h, w, 3), dtype=torch.uint8, device=device)
    y = torch.arange(h, dtype=torch.int64, device=device).view(h, 1)
    x = torch.arange(w, dtype=torch.int64, device=device).view(1, w)
    yx = (y << 14) ^ x
    for t in range(n_ticks):
        for s in range(n_sensors):
            key = yx ^ c64((seed << 52) ^ ((t + tick0) << 36) ^ ((s + sensor0) << 28))
            hsh = sm64(key)
            k = lsr(hsh, 11)
            d = 500 + (k >> 4) % 4000
            d = torch.where((k % 10) == 0, torch.zeros_like(d), d)
            depth[t, s] = d.to(torch.int16)
            c = sm64(hsh)
            rgb[t, s, :, :, 0] = (c & 0xFF).to(torch.uint8)
            rgb[t, s, :, :, 1] = ((c >> 8) & 0xFF).to(torch.uint8)
            rgb[t, s, :, :, 2] = ((c >> 16) & 0xFF).to(torch.uint8)
    return depth, rgb


def kinect_intrinsics(w=512, h=424):
    """Kinect-v2-like pinhole parameters scaled with the width; r2,r4,r6 are ignored by the fusion path."""
    f = 365.0 * (w / 512.0)
    return np.array([(w - 1) / 2.0, (h - 1) / 2.0, f, f, 0.09, -0.27, 0.09], dtype=np.float32)


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def ring_pose(sensor, n_sensors, radius=2.0):
    """Sensor on a circle of `radius` around the origin looking at it: R = Ry(2 pi s / N), t = (0,0,-radius)."""
    return rot_y(2.0 * math.pi * sensor / n_sensors), np.array([0.0, 0.0, -radius])


def pack_pose(R, t):
    """-> 12 floats {t[3], R row-major} (KinectServer.cs:479-484)."""
    return np.concatenate([np.asarray(t, dtype=np.float64).ravel(), np.asarray(R, dtype=np.float64).ravel()]).astype(np.float32)


DEFAULT_BOUNDS = np.array([-5, -5, -5, 5, 5, 5], dtype=np.float32)        # KinectSettings.cs:54-60
CROP_BOUNDS = np.array([-1.5, -1.0, -1.5, 1.5, 1.5, 1.5], dtype=np.float32)


def _ray_box(o, d, lo, hi):
    """Slab test; o [3], d [...,3]; returns entry parameter (inf when missed or behind)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (lo - o) * inv
        t1 = (hi - o) * inv
    tmin = np.max(np.minimum(t0, t1), axis=-1)
    tmax = np.min(np.maximum(t0, t1), axis=-1)
    hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0)
    return np.where(hit, tmin, np.inf)


def scene_frame(seed, tick, sensor, n_sensors, w=512, h=424, radius=2.0):
    """Ray-cast (float64) of a fixed scene from ring sensor `sensor`: sphere r=0.5 at the origin, floor y=-0.6
    (disc r=2.5), an off-centre box that breaks the symmetries.  depth = round(1000 z) +-2 mm hash jitter,
    clamped to [500,4500], 2 % dropout.  Returns (depth u16 [h,w], rgb u8 [h,w,3])."""
    intr = kinect_intrinsics(w, h).astype(np.float64)
    cx, cy, fx, fy = intr[:4]
    R, t = ring_pose(sensor, n_sensors, radius)
    y, x = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    dc = np.stack([(x - cx) / fx, (cy - y) / fy, np.ones_like(x)], axis=-1)     # camera ray, z = 1
    o = R @ t                                                                   # camera centre in the world
    d = dc @ R.T                                                                # world ray; p = o + z d
    # sphere |p| = 0.5
    a = np.sum(d * d, axis=-1)
    b = 2.0 * (d @ o)
    c = float(o @ o) - 0.25
    disc = b * b - 4 * a * c
    with np.errstate(invalid="ignore"):
        ts = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
    ts = np.where(ts > 0, ts, np.inf)
    # floor y = -0.6 within radius 2.5
    with np.errstate(divide="ignore", invalid="ignore"):
        tf = (-0.6 - o[1]) / d[..., 1]
        pf = o + tf[..., None] * d                                              # inf * 0 for rays parallel to the floor
        tf = np.where((tf > 0) & (pf[..., 0] ** 2 + pf[..., 2] ** 2 <= 2.5 ** 2), tf, np.inf)
    # box
    tb = _ray_box(o, d, np.array([0.25, -0.6, -0.55]), np.array([0.65, 0.15, -0.15]))
    tb2 = _ray_box(o, d, np.array([-0.8, -0.6, 0.3]), np.array([-0.5, 0.4, 0.5]))
    z = np.minimum(np.minimum(ts, tf), np.minimum(tb, tb2))
    which = np.argmin(np.stack([ts, tf, tb, tb2], axis=-1), axis=-1)

    hsh = _pixel_keys(seed, tick, sensor, w, h)
    jitter = (hsh % np.uint64(5)).astype(np.int64) - 2
    depth = np.where(np.isfinite(z), np.rint(1000.0 * np.where(np.isfinite(z), z, 0)).astype(np.int64) + jitter, 0)
    valid = np.isfinite(z) & (depth >= 500) & (depth <= 4500) & (((hsh >> np.uint64(20)) % np.uint64(50)) != 0)
    depth = np.where(valid, depth, 0).astype(np.uint16)
    base = np.array([[200, 60, 60], [90, 90, 90], [60, 160, 220], [230, 200, 50]], dtype=np.int64)[which]
    tint = ((splitmix64(hsh) >> np.uint64(40)) % np.uint64(32)).astype(np.int64)[..., None]
    rgb = np.clip(base + tint - 16, 0, 255).astype(np.uint8)
    return depth, rgb


class Rig:
    """The six arrays of one native call (KinectServer.cs:453-498) for n sensors."""

    def __init__(self, depths, rgbs, intr, wt, bounds):
        self.n = len(depths)
        self.widths = np.array([d.shape[1] for d in depths], dtype=np.int32)
        self.heights = np.array([d.shape[0] for d in depths], dtype=np.int32)
        self.depth_maps = np.concatenate([np.ascontiguousarray(d, dtype="<u2").ravel() for d in depths]).view(np.uint8) \
            if self.n else np.zeros(0, np.uint8)
        self.depth_colors = np.concatenate([np.ascontiguousarray(c, dtype=np.uint8).ravel() for c in rgbs]) \
            if self.n else np.zeros(0, np.uint8)
        self.intr = np.ascontiguousarray(intr, dtype=np.float32).reshape(-1)
        self.wt = np.ascontiguousarray(wt, dtype=np.float32).reshape(-1)
        self.bounds = np.ascontiguousarray(bounds, dtype=np.float32).reshape(6)
        assert self.intr.size == 7 * self.n and self.wt.size == 12 * self.n


def make_rig(kind, n_sensors, w=512, h=424, seed=1, tick=0, bounds=None, perturb=False):
    """kind = 'noise' | 'scene'.  With perturb=True every sensor k>0 is handed a pose that is off by
    Ry(1 deg) Rx(0.5 deg) and 1 cm (the mis-calibration ICP has to remove)."""
    depths, rgbs, intr, wt = [], [], [], []
    for s in range(n_sensors):
        d, c = (noise_frame(seed, tick, s, w, h) if kind == "noise" else scene_frame(seed, tick, s, n_sensors, w, h))
        depths.append(d)
        rgbs.append(c)
        intr.append(kinect_intrinsics(w, h))
        R, t = ring_pose(s, n_sensors)
        if perturb and s > 0:
            R = R @ rot_y(math.radians(1.0)) @ rot_x(math.radians(0.5))
            t = t + np.array([0.01, -0.004, 0.007])
        wt.append(pack_pose(R, t))
    if bounds is None:
        bounds = CROP_BOUNDS if kind == "scene" else np.array([-1.2, -1.0, -1.5, 1.2, 1.5, 1.5], dtype=np.float32)
    return Rig(depths, rgbs, np.concatenate(intr), np.concatenate(wt), bounds)


def digest(data):
    """sha256 hex digest of an array's bytes (pins large fixtures by hash)."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(data).view(np.uint8).tobytes()).hexdigest()
