"""Seeded synthetic scenes, local maps and multi-LiDAR scans for the scan-to-map hot path.

This is input generation only (numpy): the shapes follow SURVEY.md 8(d) -- a ground plane plus
axis-aligned "buildings", a surf map on a jittered MAP_SURF_RES grid, and ring-major ray-cast scans with ``scan_start = ring_begin + 5`` /
``scan_end = ring_end - 6`` exactly as ImageSegmenter hands them to ``FeatureExtract::extractCloud``
(reference: estimator/src/imageSegmenter/image_segmenter.hpp:385-387). Extrinsics are the ``body_T_laser``
rows of estimator/config/config_realvehicle_hercules.yaml:56-59; fused clouds carry ``intensity = lidar index``
(estimator/src/utility/visualization.cpp:48).

The corner map is assembled the way the mapper assembles its own (lidar_mapper_keyframe.cpp:254-354): the less-sharp points of
the scans taken from the previous keyframe poses, moved to the map frame and thinned at MAP_CORNER_RES. (Round 1 sampled box
edges instead; the scanner labels mostly range-noise and occlusion points "less sharp", which have no box edge nearby, so ~94 %
of the corner queries were rejected before any fit -- not what a mapper frame looks like.) The keyframe selection below is a
numpy approximation of extractCloud's less-sharp walk -- it only produces INPUT points, no result is ever compared against it.
"""
from __future__ import annotations

import dataclasses
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

# qx qy qz qw px py pz  (config_realvehicle_hercules.yaml, "PS-calib" rows)
HERCULES_BODY_T_LASER = np.array([
    [0, 0, 0, 1, 0, 0, 0],
    [-0.0169, 0.0575, 0.0195, 0.998, 0.5355, 0.0393, -1.131],
    [-0.1118, 0.1894, 0.6845, 0.6951, 0.5116, 0.6440, -0.904],
    [0.0745, 0.1312, -0.7449, 0.6496, 0.4406, -0.628, -1.0295],
], dtype=np.float64)

SENSOR_HEIGHT = 2.2  # m above the ground plane (keeps the lower LiDARs of the rig above z = 0)


def quat_to_rot(q):
    """q = (x, y, z, w) -> 3x3 (same element formulas as Eigen::Quaterniond::toRotationMatrix)."""
    x, y, z, w = q
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def rotvec_to_quat(rv):
    rv = np.asarray(rv, dtype=np.float64)
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.array([0.0, 0.0, 0.0, 1.0])
    ax = rv / th
    s = np.sin(th / 2)
    return np.array([ax[0] * s, ax[1] * s, ax[2] * s, np.cos(th / 2)])


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def pose_to_mat(pose7):
    """[tx ty tz qx qy qz qw] -> 4x4."""
    T = np.eye(4)
    T[:3, :3] = quat_to_rot(pose7[3:7])
    T[:3, 3] = pose7[:3]
    return T


def _empty(cols):
    return np.zeros((0, cols), np.float64)


@dataclasses.dataclass
class Scene:
    L: float                 # ground plane is [-L/2, L/2]^2 at z = 0
    boxes: np.ndarray        # (B, 6): xmin ymin zmin xmax ymax zmax
    seed: int
    # the "hard" family (round 6): what a street has beside planes and boxes
    cyl: np.ndarray = dataclasses.field(default_factory=lambda: _empty(4))     # vertical cylinders -- poles, trunks: cx cy radius height
    sph: np.ndarray = dataclasses.field(default_factory=lambda: _empty(5))     # "vegetation" blobs: cx cy cz radius sigma (extra range noise of a return from it)
    ramps: np.ndarray = dataclasses.field(default_factory=lambda: _empty(7))   # thin tilted slabs met at grazing incidence: x0 y0 x1 y1 (footprint) z0 slope_x slope_y
    family: str = "boxes"


def scene_family(family=None) -> str:
    """"boxes" (ground plane + axis-aligned buildings: every golden fixture and the bench workload) or "hard" (MLOAM_SCENE_FAMILY=hard, or the argument): the same
    boxes plus thin poles and trunks, noisy vegetation blobs, slabs at grazing incidence -- where the line test lambda_2 > 3 lambda_1 (feature_extract.hpp:688-693)
    and the plane gate (:823-840) sit near their thresholds far more often -- and maps with exactly duplicated points and a patch of four-fold density."""
    f = family or os.environ.get("MLOAM_SCENE_FAMILY", "boxes")
    if f not in ("boxes", "hard"):
        raise ValueError(f"unknown scene family {f!r}")
    return f


def _add_hard_objects(scene: "Scene", clear_radius: float) -> None:
    rng = np.random.default_rng(scene.seed + 7777)            # its own generator: the boxes of a seed are the same in both families
    L, B = scene.L, scene.boxes

    def free(x, y, margin):
        if np.hypot(x, y) < clear_radius * 0.4:
            return False
        for b in B:
            if b[0] - margin < x < b[3] + margin and b[1] - margin < y < b[4] + margin:
                return False
        return True
    n_cyl = int(np.clip(L / 4, 16, 90))
    cyl, sph = [], []
    tries = 0
    while len(cyl) < n_cyl and tries < 20000:
        tries += 1
        r_max = min(L / 2 - 4, 45.0)
        rad, ang = rng.uniform(3.0, r_max), rng.uniform(-np.pi, np.pi)
        x, y = rad * np.cos(ang), rad * np.sin(ang)
        if not free(x, y, 0.6):
            continue
        trunk = rng.random() < 0.4
        radius = rng.uniform(0.12, 0.35) if trunk else rng.uniform(0.04, 0.12)
        h = rng.uniform(2.0, 5.0) if trunk else rng.uniform(3.0, 12.0)
        cyl.append([x, y, radius, h])
        if trunk:                                               # a crown on top of the trunk
            sph.append([x, y, h + rng.uniform(0.3, 1.2), rng.uniform(1.0, 2.5), rng.uniform(0.08, 0.25)])
    for _ in range(int(np.clip(L / 8, 8, 40))):                 # bushes
        for _t in range(50):
            rad, ang = rng.uniform(4.0, min(L / 2 - 4, 40.0)), rng.uniform(-np.pi, np.pi)
            x, y = rad * np.cos(ang), rad * np.sin(ang)
            if free(x, y, 1.0):
                r = rng.uniform(0.4, 1.4)
                sph.append([x, y, rng.uniform(0.2, 0.9) * r, r, rng.uniform(0.05, 0.2)])
                break
    ramps = []
    for _ in range(int(np.clip(L / 25, 3, 10))):
        for _t in range(50):
            rad, ang = rng.uniform(6.0, min(L / 2 - 12, 35.0)), rng.uniform(-np.pi, np.pi)
            x, y = rad * np.cos(ang), rad * np.sin(ang)
            sx, sy = rng.uniform(5.0, 18.0, size=2)
            if free(x, y, 0.5 * max(sx, sy) + 1.0):
                slope = np.tan(np.deg2rad(rng.uniform(1.5, 8.0)))
                a = rng.uniform(-np.pi, np.pi)
                ramps.append([x - sx / 2, y - sy / 2, x + sx / 2, y + sy / 2, rng.uniform(0.02, 0.25), slope * np.cos(a), slope * np.sin(a)])
                break
    scene.cyl = np.array(cyl, np.float64).reshape(-1, 4)
    scene.sph = np.array(sph, np.float64).reshape(-1, 5)
    scene.ramps = np.array(ramps, np.float64).reshape(-1, 7)


def make_scene(L: float, n_boxes: int, seed: int = 42, clear_radius: float = 8.0, family=None) -> Scene:
    rng = np.random.default_rng(seed)
    boxes = []
    tries = 0
    while len(boxes) < n_boxes and tries < 200000:
        tries += 1
        sx, sy = rng.uniform(5, 30, size=2)
        h = rng.uniform(3, 20)
        cx, cy = rng.uniform(-L / 2 + 16, L / 2 - 16, size=2)
        b = np.array([cx - sx / 2, cy - sy / 2, 0.0, cx + sx / 2, cy + sy / 2, h])
        # keep the sensor neighbourhood free
        dx = max(b[0] - 0.0, 0.0 - b[3], 0.0)
        dy = max(b[1] - 0.0, 0.0 - b[4], 0.0)
        if np.hypot(dx, dy) < clear_radius:
            continue
        ok = True
        for o in boxes:
            if not (b[3] + 1.0 < o[0] or o[3] + 1.0 < b[0] or b[4] + 1.0 < o[1] or o[4] + 1.0 < b[1]):
                ok = False
                break
        if ok:
            boxes.append(b)
    scene = Scene(L=L, boxes=np.array(boxes).reshape(-1, 6), seed=seed, family=scene_family(family))
    if scene.family == "hard":
        _add_hard_objects(scene, clear_radius)
    return scene


def voxel_mean(points: np.ndarray, leaf: float) -> np.ndarray:
    """One centroid per occupied voxel (all columns averaged), ordered by voxel key. Data preparation only."""
    if len(points) == 0:
        return points.copy()
    ijk = np.floor(points[:, :3].astype(np.float64) / leaf).astype(np.int64)
    ijk -= ijk.min(axis=0)
    dims = ijk.max(axis=0) + 1
    key = ijk[:, 0] + dims[0] * (ijk[:, 1] + dims[1] * ijk[:, 2])
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    out = np.zeros((len(uniq), points.shape[1]), dtype=np.float64)
    for c in range(points.shape[1]):
        out[:, c] = np.bincount(inv, weights=points[:, c].astype(np.float64), minlength=len(uniq)) / cnt
    return out.astype(points.dtype)


def sample_maps(scene: Scene, surf_res: float = 0.4, corner_res: float = 0.2, seed: int = 42,
                noise: float = 0.01, corner_from: str = "keyframes", kf_rings: int = 64, kf_lidars: int = 2, n_keyframes: int = 10):
    """Returns (surf_map (Ns,3) f32, corner_map (Nc,3) f32) in the map frame, voxel-thinned at the map resolutions.
    corner_from = "keyframes": less-sharp points of `n_keyframes` earlier poses x `kf_lidars` LiDARs of `kf_rings` rings (default);
    "edges": the box edges only (round-1 behaviour, kept for comparison runs)."""
    rng = np.random.default_rng(seed + 1000)
    L = scene.L
    surf = []
    # ground
    g = np.arange(-L / 2 + surf_res / 2, L / 2, surf_res)
    gx, gy = np.meshgrid(g, g, indexing="ij")
    gx = gx.ravel() + rng.uniform(-0.1, 0.1, gx.size)
    gy = gy.ravel() + rng.uniform(-0.1, 0.1, gy.size)
    keep = np.ones(gx.size, dtype=bool)
    for b in scene.boxes:
        keep &= ~((gx > b[0]) & (gx < b[3]) & (gy > b[1]) & (gy < b[4]))
    gz = rng.normal(0.0, noise, gx.size)
    surf.append(np.stack([gx[keep], gy[keep], gz[keep]], axis=1))
    corner = []
    for b in scene.boxes:
        x0, y0, _, x1, y1, h = b
        # walls
        for (ax, fixed, lo, hi) in ((0, y0, x0, x1), (0, y1, x0, x1), (1, x0, y0, y1), (1, x1, y0, y1)):
            u = np.arange(lo + surf_res / 2, hi, surf_res)
            z = np.arange(surf_res / 2, h, surf_res)
            uu, zz = np.meshgrid(u, z, indexing="ij")
            uu = uu.ravel() + rng.uniform(-0.1, 0.1, uu.size)
            zz = zz.ravel() + rng.uniform(-0.1, 0.1, zz.size)
            nn = fixed + rng.normal(0.0, noise, uu.size)
            if ax == 0:
                surf.append(np.stack([uu, nn, zz], axis=1))
            else:
                surf.append(np.stack([nn, uu, zz], axis=1))
        # roof
        u = np.arange(x0 + surf_res / 2, x1, surf_res)
        v = np.arange(y0 + surf_res / 2, y1, surf_res)
        uu, vv = np.meshgrid(u, v, indexing="ij")
        uu = uu.ravel() + rng.uniform(-0.1, 0.1, uu.size)
        vv = vv.ravel() + rng.uniform(-0.1, 0.1, vv.size)
        surf.append(np.stack([uu, vv, h + rng.normal(0.0, noise, uu.size)], axis=1))
        # vertical edges
        for (ex, ey) in ((x0, y0), (x0, y1), (x1, y0), (x1, y1)):
            z = np.arange(corner_res / 2, h, corner_res)
            e = np.stack([np.full_like(z, ex), np.full_like(z, ey), z], axis=1)
            corner.append(e + rng.normal(0.0, noise, e.shape))
        # roof edges
        for (p0, p1) in (((x0, y0), (x1, y0)), ((x1, y0), (x1, y1)), ((x1, y1), (x0, y1)), ((x0, y1), (x0, y0))):
            ln = np.hypot(p1[0] - p0[0], p1[1] - p0[1])
            s = np.arange(corner_res / 2, ln, corner_res) / ln
            e = np.stack([p0[0] + s * (p1[0] - p0[0]), p0[1] + s * (p1[1] - p0[1]), np.full_like(s, h)], axis=1)
            corner.append(e + rng.normal(0.0, noise, e.shape))
    if scene.family == "hard":
        hrng = np.random.default_rng(seed + 8888)
        for cx, cy, r, h in scene.cyl:                              # pole / trunk surfaces
            na = max(6, int(np.ceil(2 * np.pi * r / (0.5 * surf_res))))
            a, z = np.meshgrid(np.linspace(0, 2 * np.pi, na, endpoint=False), np.arange(surf_res / 4, h, surf_res / 2), indexing="ij")
            a, z = a.ravel(), z.ravel()
            surf.append(np.stack([cx + r * np.cos(a), cy + r * np.sin(a), z], axis=1) + hrng.normal(0.0, noise, (a.size, 3)))
        for cx, cy, cz, r, sg in scene.sph:                         # vegetation: a noisy shell
            m = max(40, int(4 * np.pi * r * r / (surf_res * surf_res) * 2))
            v = hrng.normal(size=(m, 3)); v /= np.linalg.norm(v, axis=1)[:, None]
            rr = r + hrng.normal(0.0, sg, m)
            pnt = np.array([cx, cy, cz]) + v * rr[:, None]
            surf.append(pnt[pnt[:, 2] > 0.0])
        for x0, y0, x1, y1, z0, sx, sy in scene.ramps:              # thin slabs
            u = np.arange(x0 + surf_res / 2, x1, surf_res); v = np.arange(y0 + surf_res / 2, y1, surf_res)
            uu, vv = np.meshgrid(u, v, indexing="ij")
            uu = uu.ravel() + hrng.uniform(-0.1, 0.1, uu.size); vv = vv.ravel() + hrng.uniform(-0.1, 0.1, vv.size)
            surf.append(np.stack([uu, vv, z0 + sx * (uu - x0) + sy * (vv - y0) + hrng.normal(0.0, noise, uu.size)], axis=1))
    surf = np.concatenate(surf).astype(np.float32)
    corner = np.concatenate(corner).astype(np.float32) if corner else np.zeros((0, 3), np.float32)
    surf = voxel_mean(surf, surf_res)
    if corner_from == "keyframes":
        corner = keyframe_corner_cloud(scene, kf_rings, kf_lidars, n_keyframes, seed=seed)
    corner = voxel_mean(corner, corner_res)
    if scene.family == "hard":
        surf, corner = harden_maps(surf, corner, seed=seed)
    return np.ascontiguousarray(surf), np.ascontiguousarray(corner)


def harden_maps(surf: np.ndarray, corner: np.ndarray, seed: int = 42, dup_fraction: float = 0.03, dense_radius: float = 25.0):
    """What a mapper's local map has and a voxel-thinned synthetic one does not: EXACTLY duplicated points (the same keyframe cloud merged twice: equal distances,
    the k-d tree's order among them decides the fifth neighbour) and a patch of four-fold density (one LiDAR much denser than the others: the quadrant x, y > 0
    within dense_radius of the origin gets three jittered copies of every point). Appended behind the thinned cloud; the order is part of the input."""
    rng = np.random.default_rng(seed + 9999)
    out = []
    for cloud in (surf, corner):
        parts = [cloud]
        if len(cloud):
            n_dup = max(1, int(dup_fraction * len(cloud)))
            pick = rng.choice(len(cloud), size=n_dup, replace=False)
            parts.append(cloud[pick])
            parts.append(cloud[pick[: n_dup // 4]])                 # a quarter of them three times over
            q = cloud[(cloud[:, 0] > 0) & (cloud[:, 1] > 0) & (np.hypot(cloud[:, 0], cloud[:, 1]) < dense_radius)]
            for _ in range(3):
                parts.append((q + rng.normal(0.0, 0.04, q.shape)).astype(cloud.dtype))
        out.append(np.ascontiguousarray(np.concatenate(parts)))
    return out[0], out[1]


def keyframe_poses(n_keyframes: int, spacing: float = 0.35) -> np.ndarray:
    """Body poses of the keyframes behind the current one: a gently weaving track inside the box-free disc around the origin."""
    out = []
    for k in range(1, n_keyframes + 1):
        yaw = np.deg2rad(1.5) * np.sin(0.5 * k)
        out.append([-spacing * k, 0.6 * np.sin(0.35 * k), SENSOR_HEIGHT, 0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
    return np.array(out)


def less_sharp_indices(scan: "Scan", n_pick: int = 20) -> np.ndarray:
    """Approximate `corner_points_less_sharp` of extractCloud (feature_extract.cpp:152-214) for map generation: per ring and sector
    the up-to-20 largest curvatures above 0.1, each pick suppressing its +-5 neighbours (the gap test of cpp:192-213 is left out)."""
    pts = scan.points[:, :3]
    n = len(pts)
    if n < 11:
        return np.zeros(0, np.int64)
    d = np.zeros((n, 3), np.float32)
    for k in range(-5, 6):
        if k:
            d[5:n - 5] += pts[5 + k:n - 5 + k]
    d[5:n - 5] -= 10 * pts[5:n - 5]
    c = (d * d).sum(axis=1)
    rows = []
    for r in range(scan.n_rings):
        s, e = int(scan.scan_start[r]), int(scan.scan_end[r])
        if e - s < 6:
            continue
        for j in range(6):
            sp, ep = s + (e - s) * j // 6, s + (e - s) * (j + 1) // 6 - 1
            if ep >= sp:
                rows.append((sp, ep))
    if not rows:
        return np.zeros(0, np.int64)
    lmax = max(ep - sp + 1 for sp, ep in rows)
    g = len(rows)
    idx = np.full((g, lmax), -1, np.int64)
    for i, (sp, ep) in enumerate(rows):
        idx[i, :ep - sp + 1] = np.arange(sp, ep + 1)
    cm = np.where(idx >= 0, c[np.clip(idx, 0, n - 1)], -np.inf)
    cols, rows_i, picks = np.arange(lmax)[None, :], np.arange(g), []
    for _ in range(n_pick):
        a = cm.argmax(axis=1)
        ok = cm[rows_i, a] > 0.1
        picks.append(idx[rows_i, a][ok])
        cm[(np.abs(cols - a[:, None]) <= 5) & ok[:, None]] = -np.inf
    return np.concatenate(picks)


def keyframe_corner_cloud(scene: Scene, n_rings: int, n_lidars: int, n_keyframes: int, seed: int = 42) -> np.ndarray:
    """Less-sharp points of every keyframe scan in the map frame (ground-truth keyframe poses; un-thinned)."""
    jobs = [(k, i, pose) for k, pose in enumerate(keyframe_poses(n_keyframes)) for i in range(n_lidars)]

    def one(job):
        k, i, pose = job
        scn = simulate_scan(scene, pose, HERCULES_BODY_T_LASER[i], n_rings, seed=seed + 1000 + 10 * k + i)
        T_bl = np.eye(4)
        T_bl[:3, :3] = quat_to_rot(HERCULES_BODY_T_LASER[i][:4])
        T_bl[:3, 3] = HERCULES_BODY_T_LASER[i][4:7]
        return transform_points(scn.points[less_sharp_indices(scn)][:, :3], pose_to_mat(pose) @ T_bl)

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1, 16)) as ex:   # numpy releases the GIL in the ray casts
        parts = list(ex.map(one, jobs))                                                       # results in job order: deterministic
    return np.concatenate(parts).astype(np.float32)


def _raycast(scene: Scene, origin: np.ndarray, dirs: np.ndarray, max_range: float) -> np.ndarray:
    """Nearest hit distance per ray against the ground plane and the boxes (inf when none)."""
    n = dirs.shape[0]
    t_hit = np.full(n, np.inf)
    dz = dirs[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(dz < -1e-9, -origin[2] / dz, np.inf)
        px = origin[0] + tg * dirs[:, 0]
        py = origin[1] + tg * dirs[:, 1]
    okg = np.isfinite(tg) & (np.abs(px) <= scene.L / 2) & (np.abs(py) <= scene.L / 2)
    t_hit = np.where(okg, tg, t_hit)
    # only boxes that can be reached
    B = scene.boxes
    if len(B):
        c = 0.5 * (B[:, :2] + B[:, 3:5])
        r = 0.5 * np.hypot(B[:, 3] - B[:, 0], B[:, 4] - B[:, 1])
        near = np.hypot(c[:, 0] - origin[0], c[:, 1] - origin[1]) - r < max_range
        B = B[near]
    chunk = 32768
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dirs
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        iv = inv[s:e, None, :]                                   # (c,1,3)
        t0 = (B[None, :, 0:3] - origin[None, None, :]) * iv      # (c,B,3)
        t1 = (B[None, :, 3:6] - origin[None, None, :]) * iv
        tmin = np.minimum(t0, t1).max(axis=2)
        tmax = np.maximum(t0, t1).min(axis=2)
        hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)
        tb = np.where(hit, tmin, np.inf).min(axis=1) if B.shape[0] else np.full(e - s, np.inf)
        t_hit[s:e] = np.minimum(t_hit[s:e], tb)
    return t_hit


def _raycast_hard(scene: Scene, origin: np.ndarray, dirs: np.ndarray, t_hit: np.ndarray, max_range: float):
    """The hard family's objects on top of _raycast's result: nearest hit per ray and the extra range noise (sigma) of the object it belongs to."""
    n = dirs.shape[0]
    sigma = np.zeros(n)
    ox, oy, oz = origin
    dx, dy, dz = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        a2 = dx * dx + dy * dy
        for cx, cy, r, h in scene.cyl:
            if np.hypot(cx - ox, cy - oy) - r > max_range:
                continue
            bx, by = ox - cx, oy - cy
            b = bx * dx + by * dy
            c = bx * bx + by * by - r * r
            disc = b * b - a2 * c
            t = (-b - np.sqrt(np.maximum(disc, 0.0))) / a2
            z = oz + t * dz
            ok = (disc > 0) & (t > 0) & (z >= 0.0) & (z <= h) & (t < t_hit)
            t_hit = np.where(ok, t, t_hit); sigma = np.where(ok, 0.0, sigma)
        for cx, cy, cz, r, sg in scene.sph:
            if np.hypot(cx - ox, cy - oy) - r > max_range:
                continue
            bx, by, bz = ox - cx, oy - cy, oz - cz
            b = bx * dx + by * dy + bz * dz
            c = bx * bx + by * by + bz * bz - r * r
            disc = b * b - c
            t = -b - np.sqrt(np.maximum(disc, 0.0))
            ok = (disc > 0) & (t > 0) & (oz + t * dz > 0.0) & (t < t_hit)
            t_hit = np.where(ok, t, t_hit); sigma = np.where(ok, sg, sigma)
        for x0, y0, x1, y1, z0, sx, sy in scene.ramps:
            den = dz - sx * dx - sy * dy
            t = (z0 + sx * (ox - x0) + sy * (oy - y0) - oz) / den
            px, py = ox + t * dx, oy + t * dy
            ok = np.isfinite(t) & (t > 0) & (px > x0) & (px < x1) & (py > y0) & (py < y1) & (t < t_hit)
            t_hit = np.where(ok, t, t_hit); sigma = np.where(ok, 0.0, sigma)
    return t_hit, sigma


@dataclasses.dataclass
class Scan:
    points: np.ndarray       # (n, 4) f32, ring-major, LiDAR frame; column 3 = 0 (the extractor ignores it)
    scan_start: np.ndarray   # (n_rings,) int32
    scan_end: np.ndarray     # (n_rings,) int32
    n_rings: int


def simulate_scan(scene: Scene, pose_w_body: np.ndarray, body_T_laser: np.ndarray, n_rings: int, n_cols: int = 1800,
                  seed: int = 7, range_noise: float = 0.02, max_range: float = 100.0) -> Scan:
    """Ray-cast one LiDAR. pose_w_body / body_T_laser are [tx ty tz qx qy qz qw] / [qx qy qz qw px py pz] rows."""
    rng = np.random.default_rng(seed)
    if n_rings == 16:
        elev = np.deg2rad(np.linspace(-15.0, 15.0, n_rings))
    else:
        elev = np.deg2rad(np.linspace(-24.8, 2.0, n_rings))
    azim = -2.0 * np.pi * np.arange(n_cols) / n_cols
    T_wb = pose_to_mat(pose_w_body)
    T_bl = np.eye(4)
    T_bl[:3, :3] = quat_to_rot(body_T_laser[:4])
    T_bl[:3, 3] = body_T_laser[4:7]
    T_wl = T_wb @ T_bl
    ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
    d_l = np.stack([ce * np.cos(azim)[None, :], ce * np.sin(azim)[None, :], np.broadcast_to(se, (n_rings, n_cols))], axis=2)
    d_l = d_l.reshape(-1, 3)
    # explicit sums, not a BLAS call: a threaded GEMM rounds differently depending on how it splits the rows among its threads (and
    # that depends on what else runs in the process), which would make the generated inputs differ from run to run
    Rw = T_wl[:3, :3]
    d_w = np.stack([(d_l[:, 0] * Rw[r, 0] + d_l[:, 1] * Rw[r, 1]) + d_l[:, 2] * Rw[r, 2] for r in range(3)], axis=1)
    t = _raycast(scene, T_wl[:3, 3], d_w, max_range)
    extra_sigma = None
    if scene.family == "hard":
        t, extra_sigma = _raycast_hard(scene, T_wl[:3, 3], d_w, t, max_range)
    t = t + rng.normal(0.0, range_noise, t.shape)
    if extra_sigma is not None:
        t = t + extra_sigma * np.random.default_rng(seed + 31337).normal(0.0, 1.0, t.shape)     # (its own generator: the clean returns keep the boxes family's noise)
    ok = np.isfinite(t) & (t < max_range) & (t > 0.8)
    pts = (d_l * np.where(ok, t, 0.0)[:, None]).astype(np.float32)
    ok = ok.reshape(n_rings, n_cols)
    pts = pts.reshape(n_rings, n_cols, 3)
    out, starts, ends = [], [], []
    off = 0
    for r in range(n_rings):
        p = pts[r][ok[r]]
        out.append(p)
        starts.append(off + 5)
        ends.append(off + len(p) - 6)
        off += len(p)
    xyz = np.concatenate(out) if out else np.zeros((0, 3), np.float32)
    points = np.zeros((len(xyz), 4), np.float32)
    points[:, :3] = xyz
    return Scan(points=np.ascontiguousarray(points), scan_start=np.array(starts, np.int32),
                scan_end=np.array(ends, np.int32), n_rings=n_rings)


def roughen_scan(scan: "Scan", seed: int = 0, n_sectors: int = 3, near_fraction: float = 0.01, short_rings: int = 2) -> "Scan":
    """A scan with the artefacts a real sensor has and the clean ray-cast does not: whole azimuth SECTORS missing (occlusion by the vehicle, a dirty
    window: 5..40 degrees each), NEAR-RANGE returns well below a metre (self-hits: the point is pulled in along its ray), rings with fewer than twelve points and one
    ring with none at all. Ring-major order and the +5 / -6 insets are kept (a ring too short for them gets start > end, as ImageSegmenter leaves it)."""
    rng = np.random.default_rng(seed)
    starts0, ends0 = scan.scan_start - 5, scan.scan_end + 6          # the rings' own spans
    sectors = [(rng.uniform(-np.pi, np.pi), np.deg2rad(rng.uniform(5.0, 40.0))) for _ in range(n_sectors)]
    short = set(rng.choice(scan.n_rings, size=min(short_rings + 1, scan.n_rings), replace=False).tolist())
    empty = min(short) if short else -1
    out, starts, ends, off = [], [], [], 0
    for r in range(scan.n_rings):
        p = scan.points[starts0[r]:ends0[r]].copy()
        if len(p):
            az = np.arctan2(p[:, 1], p[:, 0])
            keep = np.ones(len(p), bool)
            for a0, w in sectors:
                keep &= np.abs(np.angle(np.exp(1j * (az - a0)))) > 0.5 * w
            p = p[keep]
        if r == empty:
            p = p[:0]
        elif r in short:
            p = p[:int(rng.integers(1, 12))]
        if len(p):
            near = rng.random(len(p)) < near_fraction
            rng_now = np.linalg.norm(p[:, :3], axis=1)
            scale = np.where(near, rng.uniform(0.3, 0.95, len(p)) / np.maximum(rng_now, 1e-6), 1.0).astype(np.float32)
            p[:, :3] *= scale[:, None]
        out.append(p)
        starts.append(off + 5)
        ends.append(off + len(p) - 6)
        off += len(p)
    pts = np.concatenate(out) if out else np.zeros((0, 4), np.float32)
    return Scan(points=np.ascontiguousarray(pts, np.float32), scan_start=np.array(starts, np.int32), scan_end=np.array(ends, np.int32), n_rings=scan.n_rings)


def transform_points(points_xyz: np.ndarray, T: np.ndarray) -> np.ndarray:
    p = points_xyz.astype(np.float64)
    out = np.stack([((p[:, 0] * T[r, 0] + p[:, 1] * T[r, 1]) + p[:, 2] * T[r, 2]) + T[r, 3] for r in range(3)], axis=1)   # no BLAS: see simulate_scan
    return out.astype(np.float32)


def perturbed_pose(gt_pose7: np.ndarray, seed: int = 43, dt: float = 0.2, drot_deg: float = 2.0) -> np.ndarray:
    """Ground truth perturbed by t ~ U[-dt,dt]^3, rotation vector ~ U[-drot,drot]^3."""
    rng = np.random.default_rng(seed)
    t = gt_pose7[:3] + rng.uniform(-dt, dt, 3)
    rv = np.deg2rad(rng.uniform(-drot_deg, drot_deg, 3))
    q = quat_mul(gt_pose7[3:7], rotvec_to_quat(rv))
    q /= np.linalg.norm(q)
    return np.concatenate([t, q])


# target map sizes -> (L, n_boxes) found so that the voxel-thinned surf+corner map is close to the target
SCENE_PRESETS = {
    "50k": dict(L=80.0, n_boxes=4),
    "500k": dict(L=210.0, n_boxes=60),
    "1M": dict(L=300.0, n_boxes=120),
    "2M": dict(L=420.0, n_boxes=250),
    "4M": dict(L=600.0, n_boxes=500),
}


def gt_body_pose() -> np.ndarray:
    return np.array([0.0, 0.0, SENSOR_HEIGHT, 0.0, 0.0, 0.0, 1.0])
