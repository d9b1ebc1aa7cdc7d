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
    ragged=True : n ~ logUniform(n_min, max_points), padded with (1e8,1e8,1e8,0); the two clouds of a pair draw their
                  sizes independently (a 35-point cluster may face a 7000-point one).
    ragged="matched": n_src as above, n_dst = n_src * U(0.8, 1.25) -- the same object seen from two ranges, which is what
                  the candidate pairs of match_pcds are after sanity_check (similar extents, utils_check.py:21-49).
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
            if ragged == "matched":
                nd = int(min(max(round(ns * r.uniform(0.8, 1.25)), n_min), max_points))
        else:
            ns = nd = max_points
        S[i], D[i], T[i] = make_pair(k, ns, nd, max_points, seed)
    return S, D, T


def make_frame_pair(seed=0, n_objects=24, n_min=30, n_max=1500, relabel=0.25, n_background=4000, noise=0.01):
    """A labelled synthetic frame pair in the format of the frame-pair stream (frame_pairs.py).

    n_objects vehicle-like shells on a jittered 14 m grid, object k with n_k ~ logUniform(n_min,
    n_max) source points (the destination frame sees 0.8-1.25x as many of the same surface samples,
    plus sensor noise), each moved by a small
    rigid motion (yaw <= 3 deg, |t_xy| <= 0.8 m).  Source label of object k is k; in the destination
    frame a fraction `relabel` of the objects carries a fresh label (their association has to come
    from stage 2 of match_pcds).  Background: static points labelled ground (-1e8) and noise (-1).
    -> dict(points_src, points_dst, labels_src, labels_dst, pose, gt_flow, T_true [n_objects,4,4])."""
    rng = np.random.default_rng(77_000_003 + seed)
    side = int(np.ceil(np.sqrt(n_objects)))
    ps, pd, ls, ld, gt, Ts = [], [], [], [], [], []
    fresh = 1000
    for k in range(n_objects):
        ns = int(round(np.exp(rng.uniform(np.log(n_min), np.log(n_max)))))
        nd = int(round(ns * rng.uniform(0.8, 1.25)))
        ext = np.array([rng.uniform(1.5, 5.0), rng.uniform(1.0, 2.2), rng.uniform(1.0, 2.0)])
        centre = np.array([(k % side - side / 2) * 14.0 + rng.uniform(-2, 2), (k // side - side / 2) * 14.0 + rng.uniform(-2, 2),
                           rng.uniform(0.5, 1.2)])
        yaw = np.deg2rad(rng.uniform(-3.0, 3.0))
        t = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(-0.03, 0.03)])
        c, s = np.cos(yaw), np.sin(yaw)
        Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        h = rng.uniform(-np.pi, np.pi)
        Rh = np.array([[np.cos(h), -np.sin(h), 0.0], [np.sin(h), np.cos(h), 0.0], [0.0, 0.0, 1.0]])
        shared = _shell_points(rng, ext, max(ns, nd))     # sparse clusters: both frames see the same samples
        src = shared[:ns] @ Rh.T + centre
        dst = (shared[:nd] @ Rh.T) @ Rz.T + centre + t + rng.normal(0.0, noise, size=(nd, 3))
        T = np.eye(4)
        T[:3, :3] = Rz
        T[:3, 3] = centre + t - Rz @ centre
        ps.append(src); pd.append(dst)
        ls.append(np.full(ns, float(k)))
        moved = rng.uniform() < relabel
        ld.append(np.full(nd, float(fresh if moved else k)))
        fresh += 1 if moved else 0
        gt.append(src @ Rz.T + T[:3, 3] - src)
        Ts.append(T)
    span = side * 7.0 + 10.0
    for lab, frac in ((-1e8, 0.8), (-1.0, 0.2)):
        n = int(n_background * frac)
        bg = np.stack([rng.uniform(-span, span, n), rng.uniform(-span, span, n), rng.uniform(-0.2, 0.1, n)], axis=1)
        ps.append(bg); ls.append(np.full(n, lab)); gt.append(np.zeros((n, 3)))
        pd.append(bg + rng.normal(0.0, noise, size=bg.shape)); ld.append(np.full(n, lab))
    perm_s, perm_d = rng.permutation(sum(map(len, ps))), rng.permutation(sum(map(len, pd)))
    f32 = lambda parts, perm: np.concatenate(parts, axis=0).astype(np.float32)[perm]
    return dict(points_src=f32(ps, perm_s), points_dst=f32(pd, perm_d), labels_src=f32(ls, perm_s),
                labels_dst=f32(ld, perm_d), pose=np.eye(4, dtype=np.float32), gt_flow=f32(gt, perm_s),
                T_true=np.stack(Ts).astype(np.float32))


def make_sequence(seed=0, num_frames=3, n_objects=10, n_min=60, n_max=600, speed=1.2, n_background=1500, noise=0.01):
    """A synthetic multi-frame sample in the reference's Waymo / nuScenes format (dataset_pca.py:41-45): the ego
    vehicle drives along x, `n_objects` vehicle-like shells move with constant velocity (up to `speed` m per frame)
    and yaw rate, static background points are labelled ground.  Frame j is given in ITS OWN ego coordinates
    (raw_points), ego_motion_gt[j] maps them into frame 0's, scene_flow is the reference's ground truth
    (dataset_pca.py:67-69: where the point is at time 0, in frame-0 coordinates, minus the raw point).
    -> dict(raw_points [m,3], time_indice [m], ego_motion_gt [F,4,4], nonground [m], scene_flow [m,3])."""
    rng = np.random.default_rng(55_000_007 + seed)
    side = int(np.ceil(np.sqrt(n_objects)))

    def rz(a):
        c, s_ = np.cos(a), np.sin(a)
        return np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])

    objs = []
    for k in range(n_objects):
        n = int(round(np.exp(rng.uniform(np.log(n_min), np.log(n_max)))))
        ext = np.array([rng.uniform(1.5, 5.0), rng.uniform(1.0, 2.2), rng.uniform(1.0, 2.0)])
        centre = np.array([(k % side - side / 2) * 12.0 + rng.uniform(-2, 2), (k // side - side / 2) * 12.0 + rng.uniform(-2, 2),
                           rng.uniform(0.5, 1.2)])
        heading = rng.uniform(-np.pi, np.pi)
        vel = rng.uniform(0.3, 1.0) * speed * np.array([np.cos(heading), np.sin(heading), 0.0]) * (rng.uniform() < 0.7)
        yaw_rate = np.deg2rad(rng.uniform(-1.5, 1.5))
        objs.append((_shell_points(rng, ext, n) @ rz(heading).T, centre, vel, yaw_rate))
    span = side * 6.0 + 10.0
    bg = np.stack([rng.uniform(-span, span, n_background), rng.uniform(-span, span, n_background),
                   rng.uniform(-0.2, 0.1, n_background)], axis=1)
    ego_v = np.array([rng.uniform(0.5, 1.5), rng.uniform(-0.1, 0.1), 0.0])
    ego_yaw = np.deg2rad(rng.uniform(-0.8, 0.8))
    raw, tim, ng, sf, poses = [], [], [], [], []
    for j in range(num_frames):
        P = np.eye(4)                                        # frame-j ego coordinates -> frame-0 coordinates
        P[:3, :3], P[:3, 3] = rz(ego_yaw * j), ego_v * j
        Pinv = np.linalg.inv(P)
        poses.append(P)
        w0s, wjs, flags = [], [], []
        for local, centre, vel, yaw_rate in objs:
            keep = rng.random(len(local)) < 0.9              # every frame sees most of the same surface samples
            w0 = local[keep] + centre                         # world (= frame 0) position at time 0
            wj = (local[keep] @ rz(yaw_rate * j).T) + centre + vel * j + rng.normal(0.0, noise, size=(int(keep.sum()), 3))
            w0s.append(w0); wjs.append(wj); flags.append(np.ones(len(w0), bool))
        w0s.append(bg); wjs.append(bg + rng.normal(0.0, noise, size=bg.shape)); flags.append(np.zeros(len(bg), bool))
        w0, wj = np.concatenate(w0s), np.concatenate(wjs)
        r = wj @ Pinv[:3, :3].T + Pinv[:3, 3]                 # what the sensor of frame j sees
        perm = rng.permutation(len(r))
        raw.append(r[perm]); tim.append(np.full(len(r), j)); ng.append(np.concatenate(flags)[perm]); sf.append((w0 - r)[perm])
    return dict(raw_points=np.concatenate(raw).astype(np.float32), time_indice=np.concatenate(tim).astype(np.int64),
                ego_motion_gt=np.stack(poses).astype(np.float64), nonground=np.concatenate(ng),
                scene_flow=np.concatenate(sf).astype(np.float32))
