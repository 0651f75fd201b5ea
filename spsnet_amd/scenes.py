"""Deterministic synthetic point clouds for parity tests and bench.py.

No dataset ships with the build (no network), so the hot path is exercised on synthetic scenes
whose *shape* follows what the reference feeds it: `collate_batch` + `sample_points`
(pcdet/datasets/processor/data_processor.py:218-253) deliver exactly N shuffled points per
scene, cropped to POINT_CLOUD_RANGE [0,-40,-3,70.4,40,1]
(tools/cfgs/dataset_configs/kitti_dataset.yaml:4), padded by re-drawing existing points when a
scene has fewer than N -- which creates exact duplicates, i.e. exact distance ties in FPS.

  kitti-lidar-v1 : 64-beam spinning-lidar ray model over a ground plane with box-shaped objects
  uniform-v1     : uniform in the crop box (worst case for ball query: no early exits)

Scene s of a batch is generated from numpy.random.default_rng(seed0 + s).
"""
import numpy as np

KITTI_RANGE = np.array([0.0, -40.0, -3.0, 70.4, 40.0, 1.0], dtype=np.float64)
SENSOR_HEIGHT = 1.73


def _in_range(p, rng_box=KITTI_RANGE):
    return ((p[:, 0] >= rng_box[0]) & (p[:, 0] <= rng_box[3]) & (p[:, 1] >= rng_box[1]) & (p[:, 1] <= rng_box[4])
            & (p[:, 2] >= rng_box[2]) & (p[:, 2] <= rng_box[5]))


def _lidar_scene(rng, n, extra_feats=1):
    pts = np.zeros((0, 3))
    # boxes: car-sized (3.9 x 1.6 x 1.56 m) at random poses on the ground
    nbox = 20
    centres = np.stack([rng.uniform(5, 60, nbox), rng.uniform(-30, 30, nbox),
                        np.full(nbox, -SENSOR_HEIGHT + 0.78)], axis=1)
    yaw = rng.uniform(-np.pi, np.pi, nbox)
    half = np.array([1.95, 0.8, 0.78])
    tries = 0
    while pts.shape[0] < n and tries < 8:
        rays = 4 * n
        elev = np.deg2rad(np.linspace(-24.8, 2.0, 64))[rng.integers(0, 64, rays)]
        azim = np.deg2rad(rng.uniform(-45.0, 45.0, rays))
        dirs = np.stack([np.cos(elev) * np.cos(azim), np.cos(elev) * np.sin(azim), np.sin(elev)], axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            r_ground = np.where(elev < 0, SENSOR_HEIGHT / np.tan(-elev), 80.0)
        r = np.minimum(r_ground, 80.0)
        hit = dirs * r[:, None]
        # 30 % of the rays land on an object surface instead of the ground
        on_obj = rng.random(rays) < 0.30
        k = rng.integers(0, nbox, rays)
        local = rng.uniform(-1, 1, (rays, 3)) * half
        face = rng.integers(0, 3, rays)             # clamp one axis to a face -> surface sample
        sign = np.where(rng.random(rays) < 0.5, -1.0, 1.0)
        local[np.arange(rays), face] = sign * half[face]
        c, s_ = np.cos(yaw[k]), np.sin(yaw[k])
        world = np.stack([c * local[:, 0] - s_ * local[:, 1], s_ * local[:, 0] + c * local[:, 1], local[:, 2]], axis=1)
        world += centres[k]
        hit = np.where(on_obj[:, None], world, hit)
        hit += rng.normal(0.0, 0.02, hit.shape)
        hit = hit[_in_range(hit)]
        pts = np.concatenate([pts, hit], axis=0)
        tries += 1
    if pts.shape[0] >= n:
        keep = rng.choice(pts.shape[0], n, replace=False)
    else:  # data_processor.py:236-245: pad by re-drawing existing points (duplicates)
        keep = np.concatenate([np.arange(pts.shape[0]), rng.choice(pts.shape[0], n - pts.shape[0], replace=True)])
    pts = pts[keep]
    return pts


def make_scene(kind, n, seed, dup_fraction=0.0):
    """-> (xyz (n,3) float32, intensity (n,) float32).  `dup_fraction` overwrites that share of the
    points with copies of other points (exact ties), as padded KITTI frames have."""
    rng = np.random.default_rng(seed)
    if kind == "kitti-lidar-v1":
        xyz = _lidar_scene(rng, n)
    elif kind == "uniform-v1":
        xyz = rng.uniform(KITTI_RANGE[:3], KITTI_RANGE[3:], (n, 3))
    else:
        raise ValueError(kind)
    if dup_fraction > 0:
        ndup = int(n * dup_fraction)
        dst = rng.choice(n, ndup, replace=False)
        src = rng.integers(0, n, ndup)
        xyz[dst] = xyz[src]
    rng.shuffle(xyz, axis=0)
    intensity = rng.uniform(0.0, 1.0, n)
    return xyz.astype(np.float32), intensity.astype(np.float32)


def make_batch(kind, batch, n, seed0=0, dup_fraction=0.0):
    """-> xyz (B,n,3) f32, features (B,1,n) f32 (intensity), the layout IASSD_backbone.py:105-122 hands
    to the first SA layer."""
    xs, fs = [], []
    for s in range(batch):
        x, f = make_scene(kind, n, seed0 + s, dup_fraction)
        xs.append(x)
        fs.append(f)
    return np.stack(xs), np.stack(fs)[:, None, :]


def fill_parameters(module, seed):
    """Deterministic, name-keyed values for every parameter and buffer of `module` (a torch.nn.Module): the same
    state for the reference's module and the build's mirror without shipping a state_dict -- they share key names
    and shapes.  Convolution / linear weights ~ N(0, 2 / fan_in), one-dimensional `weight` and `running_var` in
    [0.5, 1.5), `bias` / `running_mean` ~ 0.1 N(0, 1)."""
    import zlib
    import torch
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if not t.dtype.is_floating_point:
                continue
            rng = np.random.default_rng([int(seed), zlib.crc32(name.encode())])
            leaf = name.rsplit('.', 1)[-1]
            if t.dim() >= 2:
                fan_in = int(np.prod(t.shape[1:]))
                v = rng.normal(size=tuple(t.shape)) * np.sqrt(2.0 / max(fan_in, 1))
            elif leaf in ('weight', 'running_var'):
                v = rng.uniform(0.5, 1.5, size=tuple(t.shape))
            else:
                v = rng.normal(size=tuple(t.shape)) * 0.1
            t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return module
