"""Spatial partition of the local map across GPUs (host-side planning; the ownership test itself runs in the match kernel).

The mapper's features cluster around the sensor, so the map is cut into N angular wedges around a vertical axis through
a centre c (the predicted sensor position): every wedge holds a comparable share of the features AND of the map. Rank g
owns the features whose map-frame position lies in wedge g and stages the map points within `halo` of that wedge
(halo >= sqrt(min_match_sq_dis), so every neighbour inside the acceptance radius of an owned feature is local and the
correspondences are identical to the unsharded ones -- SURVEY.md 8(e)).

Wedge g = { s_g(p) >= 0 } n { s_{g+1}(p) < 0 } with s_g(p) = cross(d_g, p - c) = ((a*x + b*y) + 0*z) + d evaluated in f32 in
exactly that order by every rank (mlh_shard_set), so neighbouring ranks take complementary decisions on their shared
plane. For N = 2 the two wedges are the half-planes of a single plane. N = 1: no planes.
"""
from __future__ import annotations

import numpy as np


def wedge_planes(center_xy, n_ranks: int, rank: int, phase: float = 0.1):
    """(lo_plane4 | None, hi_plane4 | None) as float32 arrays for mlh_shard_set."""
    if n_ranks <= 1:
        return None, None
    cx, cy = float(center_xy[0]), float(center_xy[1])

    def plane(g):
        th = phase + 2.0 * np.pi * (g % n_ranks) / n_ranks
        dx, dy = np.cos(th), np.sin(th)
        # cross(d, p - c) = dx*(y - cy) - dy*(x - cx) = (-dy)*x + dx*y + (dy*cx - dx*cy)
        return np.array([-dy, dx, 0.0, dy * cx - dx * cy], dtype=np.float32)

    if n_ranks == 2:
        p = plane(0)
        return (p, None) if rank == 0 else (None, p)
    return plane(rank), plane(rank + 1)


def plane_eval_f32(plane, pts):
    """((a*x + b*y) + c*z) + d in float32, the kernel's evaluation order."""
    p = np.asarray(plane, np.float32)
    x, y, z = (pts[:, i].astype(np.float32) for i in range(3))
    return ((p[0] * x + p[1] * y) + p[2] * z) + p[3]


def owned_mask(pts, lo, hi):
    m = np.ones(len(pts), dtype=bool)
    if lo is not None:
        m &= plane_eval_f32(lo, pts) >= 0
    if hi is not None:
        m &= plane_eval_f32(hi, pts) < 0
    return m


def shard_points_mask(pts, center_xy, n_ranks: int, rank: int, halo: float = 1.1, phase: float = 0.1):
    """Map points a rank must stage: inside its wedge or within `halo` (metres, horizontal distance) of it."""
    if n_ranks <= 1:
        return np.ones(len(pts), dtype=bool)
    lo, hi = wedge_planes(center_xy, n_ranks, rank, phase)
    x = pts[:, 0].astype(np.float64) - float(center_xy[0])
    y = pts[:, 1].astype(np.float64) - float(center_xy[1])

    def sdist(plane):   # signed distance to the boundary LINE (planes are unit-normal)
        return plane[0].astype(np.float64) * pts[:, 0] + plane[1].astype(np.float64) * pts[:, 1] + float(plane[3])

    if n_ranks == 2:
        s = sdist(lo if lo is not None else hi)
        return (s >= -halo) if lo is not None else (s < halo)
    s_lo, s_hi = sdist(lo), sdist(hi)
    inside = (s_lo >= 0) & (s_hi < 0)

    def ray_dist(g):    # distance to boundary ray g (from c outwards)
        th = phase + 2.0 * np.pi * (g % n_ranks) / n_ranks
        dx, dy = np.cos(th), np.sin(th)
        t = np.maximum(x * dx + y * dy, 0.0)
        return np.hypot(x - t * dx, y - t * dy)

    near = (ray_dist(rank) <= halo) | (ray_dist(rank + 1) <= halo)
    return inside | near


def block_owner(n_blocks: int, n_ranks: int):
    """Pose blocks dealt over the ranks (config 4: one block per LiDAR, independent normal equations -- LidarOnlineCalib* factors carry one parameter block each,
    estimator.cpp:1067-1157 -- so no rank needs another rank's sums): block b -> rank b mod min(n_ranks, n_blocks); ranks beyond the block count own nothing."""
    return [b % min(n_ranks, n_blocks) for b in range(n_blocks)]
