"""Synthetic cluster-pair batches for the measurement harness.

numpy only; deterministic per pair: pair k is generated from
numpy.random.default_rng(seed + k), so any rank can materialise its own shard.
"""
import numpy as np

PAD_VALUE = 1e8


def _shell_points(rng, ext, n):
    """n points uniform on 3 visible faces (front, side, top) of a box of extents ext."""
    lx, ly, lz = ext
    areas = np.array([ly * lz, lx * lz, lx * ly])
    face = rng.choice(3, size=n, p=areas / areas.sum())
    u = rng.uniform(-0.5, 0.5, size=(n, 3)) * ext
    u[face == 0, 0] = 0.5 * lx
    u[face == 1, 1] = 0.5 * ly
    u[face == 2, 2] = 0.5 * lz
    return u


def make_pair(k, n_src, n_dst, max_points, seed=0, noise=0.01):
    """One vehicle-like cluster pair -> (src [N,4], dst [N,4], T_true [4,4]) float32."""
    rng = np.random.default_rng(seed + k)
    ext = np.array([rng.uniform(1.5, 5.0), rng.uniform(1.0, 2.2), rng.uniform(1.0, 2.0)])
    centre = np.array([rng.uniform(-40, 40), rng.uniform(-40, 40), rng.uniform(0.0, 1.5)])
    yaw = np.deg2rad(rng.uniform(-3.0, 3.0))
    t = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), rng.uniform(-0.05, 0.05)])
    c, s = np.cos(yaw), np.sin(yaw)
    Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    heading = rng.uniform(-np.pi, np.pi)
    ch, sh = np.cos(heading), np.sin(heading)
    Rh = np.array([[ch, -sh, 0.0], [sh, ch, 0.0], [0.0, 0.0, 1.0]])
    if k % 2 == 0:
        # shared surface samples: the smaller cloud is a subset of the larger one
        shared = _shell_points(rng, ext, max(n_src, n_dst))
        local_src, local_dst = shared[:n_src], shared[:n_dst].copy()
    else:
        local_src = _shell_points(rng, ext, n_src)
        local_dst = _shell_points(rng, ext, n_dst)        # independently resampled surface
    src = local_src @ Rh.T + centre
    dst = (local_dst @ Rh.T) @ Rz.T + centre + t + rng.normal(0.0, noise, size=(n_dst, 3))
    T = np.eye(4)
    T[:3, :3] = Rz
    T[:3, 3] = centre + t - Rz @ centre

    def pad(p):
        out = np.full((max_points, 4), PAD_VALUE, dtype=np.float32)
        out[:, 3] = 0.0
        out[:len(p), 0:3] = p.astype(np.float32)
        out[:len(p), 3] = 1.0
        return out

    return pad(src), pad(dst), T.astype(np.float32)


def make_batch(num_pairs, max_points, seed=0, first=0, ragged=False, n_min=20):
    """[B,N,4] src/dst float32 + ground-truth transforms [B,4,4].

    ragged=False: n_src = n_dst = max_points (BASELINE configs 2 and 4).
    ragged=True : n ~ logUniform(n_min, max_points), padded with (1e8,1e8,1e8,0).
    """
    S = np.empty((num_pairs, max_points, 4), np.float32)
    D = np.empty((num_pairs, max_points, 4), np.float32)
    T = np.empty((num_pairs, 4, 4), np.float32)
    for i in range(num_pairs):
        k = first + i
        if ragged:
            r = np.random.default_rng(10_000_019 + seed + k)
            ns = int(round(np.exp(r.uniform(np.log(n_min), np.log(max_points)))))
            nd = int(round(np.exp(r.uniform(np.log(n_min), np.log(max_points)))))
        else:
            ns = nd = max_points
        S[i], D[i], T[i] = make_pair(k, ns, nd, max_points, seed)
    return S, D, T
