"""Frame-pair stream: the unit of work above the registration path.

A frame pair is what the reference's loop hands to `track` (main.py:184-215, demo.py:37-71): two
ego-compensated, ground-free clouds with precomputed cluster labels (ground -1e8, noise -1,
clusters >= 0), the relative ego pose and -- for evaluation -- the ground-truth flow of the source
points.  On disk: one .npz per pair with keys

    points_src [Ns,3]  points_dst [Nd,3]  labels_src [Ns]  labels_dst [Nd]
    pose [4,4] (optional, identity)       gt_flow [Ns,3] (optional)     mask [Ns] (optional)

or the reference's Argoverse/demo keys (dataset_argo.py:34-45: pc1, pc2, pc1_flows_valid_idx,
pc2_flows_valid_idx, gt_flow_0_1) plus labels_src / labels_dst (clustering is precomputed in the BASELINE
configs).  A pair WITHOUT labels is clustered on the GPU when the arguments say how (`cluster="hdbscan"`
-- what the reference's scripts select, --if_hdbscan -- or `cluster="dbscan"`, with min_cluster_size /
num_clusters (/ epsilon): both clouds stacked dst-first and clustered jointly as demo.py:210 /
dataset_argo.py:119 do, utils_cluster.cluster_pcd; optional keys nonground_src /
nonground_dst mark the rows to cluster).

A Waymo / nuScenes sample of the reference (dataset_pca.py:41-45: one npz per SEQUENCE with raw_points,
time_indice, ..., and the ego poses under `ego_motion` -- the reference's <split>_pose files -- or
`ego_motion_gt`) is a multi-gap sample: `load_sequence` turns it into the num_frames - 1 frame pairs
(frame j -> frame 0, j = 1 .. num_frames - 1) the reference's loop registers (dataset_pca.py:164-198,
main.py:189-200): source = frame j moved by its ego pose, destination = frame 0, translation_frame =
2 * max(speed * j, |ego translation of frame j|), flow on the RAW source points with the ego pose composed
in (main.py:230-234, utils_flow.py:23-50).  Ground segmentation is upstream (BASELINE: "precomputed"): a
`nonground` key [m] marks the rows to cluster; without it every point is clustered.

`run_stream` registers every pair of a directory (round-robin over ranks: frame pairs are
independent, main.py:184), and reports ms / frame pair and the reference's accuracy metrics.

    python -m icp_flow_amd.frame_pairs DIR [--max-points 10000] [--speed 1.0] [--repeat 3]
"""
import argparse
import glob
import json
import os
import queue
import threading
import time
from types import SimpleNamespace

import numpy as np
import torch

from . import utils_eval, utils_flow, utils_track

# demo.sh:9-13 / main.sh flags of the registration stage
DEFAULT_ARGS = dict(max_points=10000, min_cluster_size=20, translation_frame=2.0, thres_dist=0.1, thres_box=0.1,
                    thres_rot=0.1, thres_error=0.2, thres_iou=0.2, chunk_size=50, speed=None,
                    cluster=None, epsilon=0.25, num_clusters=200, range_x=None, range_y=None,
                    # not a flag of the reference: candidate batches padded to the longest cluster of the stage instead of
                    # max_points (utils_match._stage_rows; False = the reference's width, same registrations)
                    tight_padding=True,
                    # not a flag of the reference: one call into the library per frame pair (icpflow_track_frame, the host half of
                    # match_pcds in C++, a host thread per frame pair in flight); False = the Python host (generators), same bits
                    native_host=True)


def default_args(**over):
    d = dict(DEFAULT_ARGS)
    d.update(over)
    return SimpleNamespace(**d)


class FramePair:
    def __init__(self, points_src, points_dst, labels_src=None, labels_dst=None, pose=None, gt_flow=None, mask=None,
                 name="", nonground_src=None, nonground_dst=None, gap=1, points_src_raw=None, pose_source="given"):
        self.points_src = np.ascontiguousarray(points_src, dtype=np.float32)[:, 0:3]
        self.points_dst = np.ascontiguousarray(points_dst, dtype=np.float32)[:, 0:3]
        if (labels_src is None) != (labels_dst is None):
            raise ValueError(f"frame pair {name!r}: labels of both clouds or of neither")
        self.labels_src = None if labels_src is None else np.ascontiguousarray(labels_src, dtype=np.float32)
        self.labels_dst = None if labels_dst is None else np.ascontiguousarray(labels_dst, dtype=np.float32)
        self.nonground_src = None if nonground_src is None else np.asarray(nonground_src).astype(bool)
        self.nonground_dst = None if nonground_dst is None else np.asarray(nonground_dst).astype(bool)
        for m, pts in ((self.nonground_src, self.points_src), (self.nonground_dst, self.points_dst)):
            if m is not None and m.shape != (len(pts),):
                raise ValueError(f"frame pair {name!r}: one non-ground flag per point required")
        if self.labels_src is not None and (
                len(self.labels_src) != len(self.points_src) or len(self.labels_dst) != len(self.points_dst)):
            raise ValueError(f"frame pair {name!r}: one label per point required "
                             f"({len(self.labels_src)}/{len(self.points_src)} src, "
                             f"{len(self.labels_dst)}/{len(self.points_dst)} dst)")
        self.pose = np.eye(4, dtype=np.float32) if pose is None else np.asarray(pose, dtype=np.float32).reshape(4, 4)
        # the pose as given (the reference's ego poses are float64 and main.py:200 takes their norm as such)
        self.pose_exact = np.eye(4) if pose is None else np.asarray(pose, dtype=np.float64).reshape(4, 4)
        self.gt_flow = None if gt_flow is None else np.asarray(gt_flow, dtype=np.float32)
        if self.gt_flow is not None and self.gt_flow.shape != self.points_src.shape:
            raise ValueError(f"frame pair {name!r}: gt_flow must be [Ns,3]")
        self.mask = None if mask is None else np.asarray(mask)
        self.name = name
        self.pose_source = pose_source        # where the ego pose came from (load_sequence), reported by run_stream
        # multi-gap samples (Waymo / nuScenes): frame `gap` against frame 0; the flow is reported on the source
        # points BEFORE ego-motion compensation (main.py:230-234), registration runs on the compensated ones
        self.gap = int(gap)
        self.points_src_raw = None if points_src_raw is None else np.ascontiguousarray(points_src_raw, dtype=np.float32)[:, 0:3]
        if self.points_src_raw is not None and self.points_src_raw.shape != self.points_src.shape:
            raise ValueError(f"frame pair {name!r}: points_src_raw must match points_src")


def save_frame_pair(path, fp):
    arrays = dict(points_src=fp.points_src, points_dst=fp.points_dst, pose=fp.pose)
    for k in ("labels_src", "labels_dst", "nonground_src", "nonground_dst"):
        if getattr(fp, k) is not None:
            arrays[k] = getattr(fp, k)
    if fp.gt_flow is not None:
        arrays["gt_flow"] = fp.gt_flow
    if fp.mask is not None:
        arrays["mask"] = fp.mask
    np.savez_compressed(path, **arrays)


def load_frame_pair(path):
    with np.load(path) as z:
        keys = set(z.files)

        def first(*names):
            for n in names:
                if n in keys:
                    return z[n]
            return None

        labels_src, labels_dst = first("labels_src", "label_src"), first("labels_dst", "label_dst")
        ng = dict(nonground_src=first("nonground_src"), nonground_dst=first("nonground_dst"))
        if "points_src" in keys:
            return FramePair(z["points_src"], z["points_dst"], labels_src, labels_dst, first("pose"),
                             first("gt_flow"), first("mask"), name=os.path.basename(path), **ng)
        if "pc1" in keys:                                   # dataset_argo.py:34-53, demo.py:37-51
            v0, v1 = z["pc1_flows_valid_idx"], z["pc2_flows_valid_idx"]
            gt = z["gt_flow_0_1"][v0] if "gt_flow_0_1" in keys else None
            return FramePair(z["pc1"][v0], z["pc2"][v1], labels_src, labels_dst, first("pose"), gt, first("mask"),
                             name=os.path.basename(path), **ng)
        raise ValueError(f"{path}: neither points_src/points_dst nor pc1/pc2 present")


def is_sequence(path):
    with np.load(path) as z:
        return "raw_points" in z.files and "time_indice" in z.files


def _pose_file(path):
    """The reference keeps estimated ego poses next to the split: .../val/x.npz -> .../val_pose/x.npz, key
    `ego_motion` (dataset_pca.py:118-125).  Only a DIRECTORY COMPONENT named train / val / test is a split (the last one
    wins): '/latest/val/x.npz' -> '/latest/val_pose/x.npz', never '/latest_pose/...'."""
    parts = os.path.normpath(path).split(os.sep)
    for k in range(len(parts) - 2, -1, -1):
        if parts[k] in ("train", "val", "test"):
            cand = os.sep.join(parts[:k] + [parts[k] + "_pose"] + parts[k + 1:])
            return cand if os.path.isfile(cand) else None
    return None


POSE_SOURCES = ("auto", "pose_file", "ego_motion", "ego_motion_gt")


def load_sequence(path, args=None, pose_source=None):
    """A multi-frame sample in the reference's Waymo / nuScenes format (dataset_pca.py:41-45) -> the list of
    frame pairs of the reference's loop, gap j = 1 .. num_frames - 1 (dataset_pca.py:164-198).
    Keys: raw_points [m,>=3], time_indice [m] (0 .. F-1), ego poses [F,4,4]; optional nonground [m] (ground segmentation
    is upstream), scene_flow [m,3] (ground truth, dataset_pca.py:67-69), labels [m] (precomputed per-frame cluster
    labels are NOT possible in this format: frames are clustered jointly per gap).  args.range_x / range_y crop
    the scene like dataset_pca.py:61-64.

    Ego poses (`pose_source`, default args.pose_source or "auto"): the reference reads ESTIMATED poses from its
    <split>_pose files (key `ego_motion`, dataset_pca.py:118-125) and estimates them itself (KISS-ICP, out of scope here)
    when the file is missing.  "pose_file": that file, an error without it; "ego_motion": the key of that name in the
    sample itself; "ego_motion_gt": the GROUND-TRUTH poses of the sample -- accuracy measured with them is not the
    reference's protocol; "auto": pose file, else in-file `ego_motion`, else `ego_motion_gt` WITH a warning.  Every
    frame pair records where its pose came from (`FramePair.pose_source`, carried into run_stream's results)."""
    import warnings
    source = pose_source or (getattr(args, "pose_source", None) if args is not None else None) or "auto"
    if source not in POSE_SOURCES:
        raise ValueError(f"pose_source must be one of {POSE_SOURCES} (got {source!r})")
    with np.load(path) as z:
        keys = set(z.files)
        raw = np.asarray(z["raw_points"])[:, 0:3].astype(np.float64)
        t = np.asarray(z["time_indice"]).astype(np.int64)
        side = _pose_file(path) if source in ("auto", "pose_file") else None
        if side is not None:
            with np.load(side) as zp:
                poses = np.asarray(zp["ego_motion"]).astype(np.float64)
            used = "pose_file"
        elif source == "pose_file":
            raise FileNotFoundError(f"{path}: no <split>_pose file next to the split directory (pose_source='pose_file')")
        elif source in ("auto", "ego_motion") and "ego_motion" in keys:
            poses, used = np.asarray(z["ego_motion"]).astype(np.float64), "ego_motion"
        elif source in ("auto", "ego_motion_gt") and "ego_motion_gt" in keys:
            poses, used = np.asarray(z["ego_motion_gt"]).astype(np.float64), "ego_motion_gt"
            if source == "auto":
                warnings.warn(f"{path}: no estimated ego poses (no <split>_pose file, no `ego_motion` key): falling back to "
                              "GROUND-TRUTH poses `ego_motion_gt`; the reference would estimate them (dataset_pca.py:126-130). "
                              "Pass pose_source='ego_motion_gt' to accept this silently.", stacklevel=2)
        else:
            raise KeyError(f"{path}: no ego poses for pose_source={source!r} (keys: {sorted(keys)})")
        nonground = np.asarray(z["nonground"]).astype(bool) if "nonground" in keys else None
        gt = np.asarray(z["scene_flow"])[:, 0:3].astype(np.float32) if "scene_flow" in keys else None
    if raw.shape[0] != t.shape[0] or poses.shape[1:] != (4, 4) or poses.shape[0] != t.max() + 1:
        raise ValueError(f"{path}: raw_points / time_indice / ego poses do not describe one sequence")
    rx, ry = (getattr(args, "range_x", None), getattr(args, "range_y", None)) if args is not None else (None, None)
    if rx is not None and ry is not None:
        keep = (np.abs(raw[:, 0]) < rx) & (np.abs(raw[:, 1]) < ry)                  # dataset_pca.py:61-64
        raw, t = raw[keep], t[keep]
        nonground = None if nonground is None else nonground[keep]
        gt = None if gt is None else gt[keep]
    out = []
    dst = raw[t == 0]
    for j in range(1, poses.shape[0]):
        m = t == j
        src_raw = raw[m]
        hom = np.concatenate([src_raw, np.ones((len(src_raw), 1))], axis=1)
        src_ego = (hom @ poses[j].T)[:, 0:3]                                          # utils_helper.py:89-93
        out.append(FramePair(src_ego, dst, None, None, poses[j], None if gt is None else gt[m], None,
                             name=f"{os.path.basename(path)}#gap{j}",
                             nonground_src=None if nonground is None else nonground[m],
                             nonground_dst=None if nonground is None else nonground[t == 0],
                             gap=j, points_src_raw=src_raw, pose_source=used))
    return out


def load_any(path, args=None, pose_source=None):
    """-> list of FramePair: one for a frame-pair file, num_frames - 1 for a sequence file."""
    return load_sequence(path, args, pose_source) if is_sequence(path) else [load_frame_pair(path)]


def list_frame_pairs(directory):
    return sorted(glob.glob(os.path.join(directory, "**", "*.npz"), recursive=True))


def shard_round_robin(items, rank, world):
    """frame pairs are dealt round-robin to ranks (keeps the association of a frame
    pair on one GPU)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(items[rank::world])


def frame_translation(args, pose, gap=1):
    """main.py:200: translation_frame = 2 * max(speed * gap, |ego translation|); unchanged when the
    caller fixed translation_frame (speed None), as demo.py:205 does."""
    if getattr(args, "speed", None) is None:
        return float(args.translation_frame)
    return float(max(args.speed * gap, float(np.linalg.norm(np.asarray(pose)[0:3, 3])))) * 2.0


def cluster_frame_pair(args, ps, pd, nonground_src=None, nonground_dst=None):
    """Joint clustering of a frame pair as demo.py:210 / dataset_argo.py:112-121 do it: both clouds stacked
    dst-first, one cluster_pcd call, labels split back.  -> (labels_src, labels_dst) float32 device tensors."""
    if getattr(args, "cluster", None) not in ("dbscan", "hdbscan"):
        raise ValueError("frame pair without cluster labels: pass precomputed labels_src / labels_dst or set "
                         "args.cluster = 'hdbscan' | 'dbscan'")
    from . import utils_cluster
    dev = ps.device
    ones = lambda n: torch.ones(n, dtype=torch.bool, device=dev)
    m_src = ones(len(ps)) if nonground_src is None else torch.as_tensor(nonground_src, device=dev).bool()
    m_dst = ones(len(pd)) if nonground_dst is None else torch.as_tensor(nonground_dst, device=dev).bool()
    a = SimpleNamespace(epsilon=float(args.epsilon), min_cluster_size=int(args.min_cluster_size),
                        num_clusters=int(args.num_clusters), if_hdbscan=args.cluster == "hdbscan")
    labels = utils_cluster.cluster_pcd(a, torch.cat([pd, ps], dim=0), torch.cat([m_dst, m_src], dim=0)).float()
    return labels[len(pd):].contiguous(), labels[: len(pd)].contiguous()


_stream_pool = {}


def make_resident(fp, device):
    """Upload the frame pair's arrays once and keep the device tensors with it (`register_frame_pair_native` then takes those:
    inputs resident in HBM, what bench.py's stream workload times).  -> fp"""
    device = torch.device(device)
    keep = {}
    for name in ("points_src", "points_dst", "labels_src", "labels_dst", "points_src_raw"):
        arr = getattr(fp, name)
        if arr is not None:
            keep[name] = torch.from_numpy(arr).to(device)
    fp._resident = (device, keep)
    return fp


def _input(fp, name, device):
    res = getattr(fp, "_resident", None)
    if res is not None and res[0] == torch.device(device) and name in res[1]:
        return res[1][name]
    return _upload(getattr(fp, name), device)


def _upload(arr, device):
    """Host array -> device tensor, from pageable memory.  (Measured on this stack: 41 us for a 63 k-point cloud; the
    same upload through a pinned staging buffer with a non-blocking copy made a stream of frame pairs ten times SLOWER --
    the copies serialise against the kernels of the other streams.)"""
    return torch.from_numpy(arr).to(device)


def register_frame_pair_steps(args, fp, device, gap=None, asynchronous=False):
    """One frame pair through (cluster_pcd when it carries no labels +) track() + flow_estimation_torch() on `device`, as
    a generator that yields at every device -> host hand-over of the association (utils_match.match_pcds_steps) and
    returns dict(pairs [P,10], transformations [P,4,4], flow [Ns,3]) of device tensors."""
    from . import utils_match
    a = SimpleNamespace(**vars(args))
    a.translation_frame = frame_translation(args, fp.pose_exact, fp.gap if gap is None else gap)
    device = torch.device(device)
    ps = _upload(fp.points_src, device)
    pd = _upload(fp.points_dst, device)
    if fp.labels_src is None:
        ls, ld = cluster_frame_pair(args, ps, pd, fp.nonground_src, fp.nonground_dst)
    else:
        ls = _upload(fp.labels_src, device)
        ld = _upload(fp.labels_dst, device)
    pose = torch.from_numpy(fp.pose).to(device)
    # main.py:139 seeds torch's global generator once; a private generator with the same seed gives the same
    # draws (random subsampling of over-long clusters) without touching the caller's global RNG state
    a.generator = torch.Generator()
    a.generator.manual_seed(0)
    # multi-gap sample: flow of the RAW source points, the ego pose composed in (main.py:230-234; T registers the
    # ego-compensated cloud, so a point moves by T * pose); frame-pair files hand over ego-compensated clouds, their `pose`
    # (identity in demo.py:221) is composed as given
    flow_src = ps if fp.points_src_raw is None else _upload(fp.points_src_raw, device)
    # the device-side association enqueues the flow right behind the pair rows, in the same call and ahead of its one
    # read-back (utils_match._match_pcds_device): the frame pair has ONE wait, at its end
    a.flow_request = dict(points=flow_src, labels=ls, pose=pose)
    a.flow_result = None
    a.association_path = None
    pairs, T = yield from utils_match.match_pcds_steps(a, ps, pd, ls, ld, asynchronous)      # utils_track.py:31-35
    if a.association_path == "device" and a.flow_result is not None:
        flow = a.flow_result
    else:
        flow = utils_flow.flow_estimation_torch(a, flow_src, pd, ls, ld, pairs, T, pose)
    return dict(pairs=pairs, transformations=T, flow=flow, translation_frame=a.translation_frame, association=a.association_path or "host")


def _native_host(args):
    """One call into the library per frame pair (icpflow_track_frame) unless switched off: `args.native_host = False`, or
    `args.device_association = False` (the host-side association is the Python host's)."""
    return getattr(args, "native_host", True) and getattr(args, "device_association", None) is not False


def register_frame_pair(args, fp, device, gap=None):
    """One frame pair: through `register_frame_pair_native` (one blocking call into the library) where that serves it,
    otherwise `register_frame_pair_steps` driven to its end (every hand-over blocks).  Same result either way."""
    from . import utils_match
    if _native_host(args) and torch.device(device).type == "cuda":
        out = register_frame_pair_native(args, fp, device, gap)
        if _served(out):
            return out
        return _python_host(args, fp, device, gap, out)
    return utils_match.drive(register_frame_pair_steps(args, fp, device, gap))


def register_in_flight_scheduler(args, fps, device, in_flight=4):
    """Register the frame pairs `fps` (an iterable) with up to `in_flight` of them at once: each on its own HIP stream,
    its device -> host hand-overs as asynchronous copies into pinned memory, and ONE host thread that resumes whichever
    frame pair's transfer has landed -- the host half of one frame pair (candidate lists, reject test, assignment) and
    the gaps between its small launches run under the kernels of the others.  Frame pairs are independent
    (main.py:184-215), every one gets exactly the result of `register_frame_pair` (its own random stream included).
    Yields (index, frame pair, result dict) in completion order; the results' tensors are ready on the device."""
    device = torch.device(device)
    # (the streams live as long as the process: pinned staging buffers and workspaces are kept per stream, and pinned
    # memory is expensive to allocate)
    # (per host thread: two threads registering on one device must not share streams)
    pool = _stream_pool.setdefault((device.type, device.index, threading.get_ident()), [])
    while len(pool) < max(int(in_flight), 1):
        pool.append(torch.cuda.Stream(device))
    streams = pool[: max(int(in_flight), 1)]
    free = list(range(len(streams)))
    running = {}                                   # slot -> [index, fp, generator, pending]
    source = enumerate(fps)
    exhausted = False

    def advance(slot):
        idx, fp, gen, _ = running[slot]
        with torch.cuda.stream(streams[slot]):
            try:
                running[slot][3] = next(gen)
                return None
            except StopIteration as done:
                ev = torch.cuda.Event()
                ev.record()
                del running[slot]
                free.append(slot)
                return idx, fp, done.value, ev

    finished = []
    while True:
        while free and not exhausted:
            try:
                idx, fp = next(source)
            except StopIteration:
                exhausted = True
                break
            slot = free.pop()
            running[slot] = [idx, fp, register_frame_pair_steps(args, fp, device, asynchronous=True), None]
            out = advance(slot)
            if out is not None:
                finished.append(out)
        progressed = False
        for slot in list(running):
            if running[slot][3].ready():
                out = advance(slot)
                progressed = True
                if out is not None:
                    finished.append(out)
        for k in range(len(finished) - 1, -1, -1):
            if finished[k][3].query():
                idx, fp, res, _ = finished.pop(k)
                yield idx, fp, res
                progressed = True
        if not running and not finished and exhausted:
            return
        if not progressed:
            # nothing has landed yet: wait for the oldest hand-over instead of spinning on the queries
            if running:
                next(iter(running.values()))[3].get()
            elif finished:
                finished[0][3].synchronize()


_frame_scratch = {}
# what track_frame_native returns when a stage-2 candidate that cannot be in the superset (an over-long cluster) is needed: the
# device-side association of the Python host would give up on this frame pair as well -- go to the host-side association
NEEDS_HOST_ASSOCIATION = "needs the host-side association"


def _served(out):
    return out is not None and out is not NEEDS_HOST_ASSOCIATION


def _python_host(args, fp, device, gap, out):
    """The generators, for a frame pair the native call did not serve (`out`: what it returned)."""
    from . import utils_match
    if out is NEEDS_HOST_ASSOCIATION:
        args = SimpleNamespace(**vars(args))
        args.device_association = False
    return utils_match.drive(register_frame_pair_steps(args, fp, device, gap))


def register_frame_pair_native(args, fp, device, gap=None):
    """One frame pair through icpflow_track_frame: the whole of match_pcds + flow -- the host half included (cluster tables,
    candidate lists, sanity_check, padded batches and their random subsamples, both association stages on the device) -- in ONE
    blocking call into the library, which releases the interpreter lock.  Bit for bit the result of `register_frame_pair` with
    the device-side association (tests/test_gpu_parity.py::test_native_frame_pair_equals_the_python_host).
    -> the result dict, or None when the call cannot serve this frame pair (options outside the single speculative launch, no
    candidate pair at all, more than 512 clusters): the caller takes `register_frame_pair_steps`."""
    a = SimpleNamespace(**vars(args))
    a.translation_frame = frame_translation(args, fp.pose_exact, fp.gap if gap is None else gap)
    device = torch.device(device)
    ps = _input(fp, "points_src", device)
    pd = _input(fp, "points_dst", device)
    if fp.labels_src is None:
        ls, ld = cluster_frame_pair(args, ps, pd, fp.nonground_src, fp.nonground_dst)
    else:
        ls = _input(fp, "labels_src", device)
        ld = _input(fp, "labels_dst", device)
    pose = torch.from_numpy(fp.pose).to(device)
    flow_src = ps if fp.points_src_raw is None else _input(fp, "points_src_raw", device)
    return track_frame_native(a, ps, pd, ls, ld, pose, flow_src)


_randperm_ok = None    # None: not checked yet in this process


def randperm_restatement_ok():
    """icpflow_track_frame draws the subsamples of over-long clusters itself: a restatement of torch.randperm's CPU stream
    (MT19937, the 32-bit branch of randperm_cpu; csrc/frame.hip) in place of `torch.randperm` under main.py:139's seed
    (utils_helper.py:198-201).  A torch build that draws differently would give another -- valid, but not reference-identical --
    subsample, silently.  So the first native frame pair of a process compares the two on two (seed, n, take) cases, one of
    them mid-stream; on a mismatch: one warning, and every frame pair of the process goes through the Python host
    (`torch.randperm` itself).  Costs ~0.3 ms once."""
    global _randperm_ok
    if _randperm_ok is None:
        import ctypes
        import warnings
        from . import _lib
        ok = True
        for seed, n, take, skip in ((0, 30552, 2048, 0), (12345, 61849, 10000, 700)):
            g = torch.Generator()
            g.manual_seed(seed)
            if skip:
                torch.randperm(skip, generator=g)
            mt = _lib.Mt19937.from_torch(g)
            want = torch.randperm(n, generator=g)[:take].numpy().astype(np.int32)
            got = np.empty(take, np.int32)
            rc = _lib._L.icpflow_selftest_randperm(ctypes.byref(mt), n, take, got.ctypes.data_as(ctypes.c_void_p))
            ok = ok and rc == 0 and np.array_equal(got, want)
        if not ok:
            warnings.warn("icp_flow_amd: this torch build's randperm does not draw like the library's restatement of it "
                          f"(torch {torch.__version__}); frame pairs are registered through the Python host "
                          "(args.native_host=False), which calls torch.randperm itself", RuntimeWarning, stacklevel=3)
        _randperm_ok = ok
    return _randperm_ok


def track_frame_native(a, ps, pd, ls, ld, pose=None, flow_points=None, seed=0, generator=None):
    """icpflow_track_frame on device tensors: `track(a, ps, pd, ls, ld)` (+ `flow_estimation_torch` of `flow_points` under
    `pose` when given) with `a.translation_frame` set; the random subsamples of over-long clusters are torch.randperm's on a
    generator seeded with `seed` -- or on `generator` (a torch CPU generator, or "global" for torch's own: its state goes in,
    and comes back advanced when the call has served the frame pair).
    -> dict(pairs, transformations[, flow]), None (see register_frame_pair_native) or NEEDS_HOST_ASSOCIATION."""
    from . import _lib, utils_match
    import ctypes
    device = ps.device
    max_it, _, stop = utils_match._icp_options(a)
    cur = _lib._current()[-1]
    if not (stop == 0 and 2 <= max_it <= 128 and cur["arith"] == 0 and not (cur["flags"] & _lib.OPT_FLAGS["no_speculative"])):
        return None
    if len(ps) == 0 or len(pd) == 0:
        return None
    if not randperm_restatement_ok():
        return None                     # (the caller's Python host draws with torch.randperm itself)
    _lib.require_gpu(ps, pd, ls, ld)
    ps3, pd3 = ps[:, 0:3].contiguous().float(), pd[:, 0:3].contiguous().float()
    ls, ld = ls.contiguous().float(), ld.contiguous().float()
    rows = torch.empty((1024, 10), dtype=torch.float32, device=device)
    T = torch.empty((1024, 4, 4), dtype=torch.float32, device=device)
    flow = f_pts = f_pose = None
    if flow_points is not None:
        f_pts = ps3 if flow_points is ps else flow_points[:, 0:3].contiguous().float()
        assert len(f_pts) == len(ls)
        f_pose = pose.to(device).contiguous().float()
        flow = torch.empty((len(ps3), 3), dtype=torch.float32, device=device)
    reg, keep_alive = utils_match._registration(a, device)
    f32 = lambda v: float(np.float32(v))   # noqa: E731
    mt = None
    if generator is not None:
        mt = _lib.Mt19937.from_torch(None if isinstance(generator, str) else generator)
    par = _lib.FrameParams(ctypes.sizeof(_lib.FrameParams), int(seed), ctypes.addressof(mt) if mt is not None else None,
                           int(a.max_points), int(a.min_cluster_size),
                           f32(a.translation_frame), f32(a.thres_box), f32(a.thres_iou), f32(a.thres_rot * 90.0), f32(a.thres_error),
                           1 if getattr(a, "tight_padding", True) else 0, int(getattr(a, "device_association_width", 1024)))
    key = (device.index, _lib.stream_handle(device))
    scratch = _frame_scratch.get(key)
    if scratch is None:
        scratch = _frame_scratch[key] = torch.empty((64 << 20,), dtype=torch.uint8, device=device)
    pairs, need = ctypes.c_int32(0), ctypes.c_size_t(0)
    with _lib.options(teams_half_gpu=not getattr(a, "teams_full_gpu", False), no_shared_scans=not getattr(a, "shared_scans", False),
                      no_stage_overlap=getattr(a, "stage_overlap", None) is False):
        opt = _lib.opt()
        for _ in range(5):   # (the scratch is sized in up to three parts -- fixed, stages, exact stage 2 --, each learnt from a refusal)
            rc = _lib._L.icpflow_track_frame(_lib.ptr(ps3), _lib.ptr(ls), len(ps3), _lib.ptr(pd3), _lib.ptr(ld), len(pd3),
                                             ctypes.byref(reg), ctypes.byref(par), _lib.ptr(rows), _lib.ptr(T), ctypes.byref(pairs),
                                             _lib.ptr(f_pts), _lib.ptr(f_pose), _lib.ptr(flow), _lib.ptr(scratch), scratch.numel(),
                                             ctypes.byref(need), _lib.stream(device), opt)
            if rc != -2:
                break
            # (the scratch grows to what this frame pair needs, with some room: the next ones are alike)
            scratch = _frame_scratch[key] = torch.empty((int(need.value * 1.25),), dtype=torch.uint8, device=device)
    if rc != 0:
        msg = _lib._L.icpflow_last_error()
        raise RuntimeError(f"icpflow_track_frame failed (code {rc}): {msg.decode() if msg else ''}")
    P = int(pairs.value)
    if P == -3:
        return NEEDS_HOST_ASSOCIATION
    if P == -2:
        return None
    if P < 0:
        raise RuntimeError("icpflow_hist_icp abandoned the batch: a wait between workgroups timed out -- a team sharing one "
                           "large pair (GPU shared with another process?); retry, or register with _lib.options(no_teams=True)")
    if mt is not None:
        mt.to_torch(None if isinstance(generator, str) else generator)
    out = dict(pairs=rows[:P], transformations=T[:P], translation_frame=a.translation_frame, association="device")
    if flow is not None:
        out["flow"] = flow
    return out


def register_in_flight(args, fps, device, in_flight=4):
    """Up to `in_flight` frame pairs at once: `register_in_flight_native` (a host thread per frame pair in flight, one blocking
    call into the library each) unless `args.native_host` / `args.device_association` is False -- then the generator-based
    scheduler on one host thread (`register_in_flight_scheduler`).  Yields (index, frame pair, result dict) in completion order."""
    if _native_host(args) and torch.device(device).type == "cuda":
        yield from register_in_flight_native(args, fps, device, in_flight)
    else:
        yield from register_in_flight_scheduler(args, fps, device, in_flight)


class _Worker(threading.Thread):
    """A host thread that lives as long as the process, with its own HIP stream: the library keeps state per host thread (side
    streams, pinned staging) and `track_frame_native` keeps its device scratch per stream -- a thread and a stream per CALL of
    register_in_flight_native would allocate all of that again every time (hipMalloc and hipHostMalloc synchronise the device)."""

    def __init__(self, device):
        super().__init__(daemon=True)
        self.device = device
        self.jobs = queue.Queue()
        self.start()

    def run(self):
        torch.cuda.set_device(self.device)
        stream = torch.cuda.Stream(self.device)
        while True:
            job = self.jobs.get()
            with torch.cuda.stream(stream):
                job(stream)


_workers = {}
_workers_lock = threading.Lock()
_ABANDON_S = 120.0     # a consumer that takes no result for this long has gone (register_in_flight_native)


def register_in_flight_native(args, fps, device, in_flight=4):
    """`register_in_flight` with one HOST THREAD per frame pair in flight, each on its own HIP stream, each frame pair one
    blocking `register_frame_pair_native` call (frame pairs that call cannot serve go through `register_frame_pair_steps` on
    the same thread).  The threads spend their time inside the library with the interpreter lock released; the library chains
    team launches per device whatever thread they come from.  Yields (index, frame pair, result dict) in completion order,
    results complete on the device."""
    from . import utils_match
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    source = enumerate(fps)
    lock = threading.Lock()
    n = max(1, int(in_flight))
    # several frame pairs at once keep the GPU busy by themselves: stage 2's initial poses stay behind stage 1 (the overlap of
    # icpflow_track_frame registers stage 2's whole candidate superset -- extra work that shortens one frame pair on an idle GPU
    # and costs throughput on a busy one: 0.74 -> 0.88 ms per demo frame pair with four in flight when tried); same results
    flight_args = args
    if n > 1 and getattr(args, "stage_overlap", None) is None:
        flight_args = SimpleNamespace(**vars(args))
        flight_args.stage_overlap = False
    out = queue.Queue(maxsize=2 * n + n)     # back-pressure: a slow consumer holds at most ~2 results per worker (+ the end marks)
    stop = threading.Event()                 # the consumer has gone (generator closed, or an error): take no further frame pair

    def deliver(item):
        #  the queue is bounded and the workers outlive the call: a consumer that has gone -- the generator closed, or
        # kept somewhere half-consumed -- must not leave a pooled thread blocked in put() for ever (every later call on this
        # device would queue behind it).  Once `stop` is set the item is dropped; a consumer that takes nothing for
        # _ABANDON_S seconds counts as gone.
        waited = 0.0
        while True:
            try:
                out.put(item, timeout=0.2)
                return
            except queue.Full:
                waited += 0.2
                if stop.is_set() or waited >= _ABANDON_S:
                    stop.set()
                    return

    def job(stream):
        try:
            while True:
                with lock:
                    if stop.is_set():
                        return
                    try:
                        idx, fp = next(source)
                    except StopIteration:
                        return
                res = register_frame_pair_native(flight_args, fp, device)
                if not _served(res):
                    res = _python_host(args, fp, device, None, res)
                stream.synchronize()
                deliver((idx, fp, res))
        except BaseException as e:   # noqa: BLE001  (handed to the consumer)
            deliver(e)
        finally:
            deliver(None)

    with _workers_lock:
        pool = _workers.setdefault(device.index, [])
        while len(pool) < n:
            pool.append(_Worker(device))
        mine = pool[:n]
    for w in mine:
        w.jobs.put(job)
    done, error = 0, None
    try:
        while done < n:
            item = out.get()
            if item is None:
                done += 1
            elif isinstance(item, BaseException):
                error = error or item
                stop.set()
            elif error is None:
                # the result tensors were allocated on the worker's stream: tell the caching allocator that the consumer's
                # stream uses them too, so that a block freed here is not handed out again under a kernel still reading it
                cur = torch.cuda.current_stream(device)
                for v in item[2].values():
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(cur)
                yield item
    finally:
        stop.set()                       # (closed early: the workers finish the frame pair they hold and leave)
        waited = 0.0
        while done < n and waited < _ABANDON_S:      # (bounded: a worker stuck inside the library must not hang the consumer too)
            try:
                if out.get(timeout=0.2) is None:
                    done += 1
            except queue.Empty:
                waited += 0.2
    if error is not None:
        raise error


def run_stream(args, paths, device, rank=0, world=1, repeat=1, group=None, register_fn=None, in_flight=1):
    """Register this rank's share of `paths`; -> summary dict (identical on every rank).
    ms / frame pair is the mean wall time per pair, host -> device upload of the clouds included
    (the stream hands over host buffers); frame_pairs_per_s uses the slowest rank's total.
    `register_fn(args, fp, device) -> dict(pairs, transformations, flow)` defaults to the HIP path
    (`register_frame_pair`); the CPU tests of the sharding logic pass the oracle here.
    in_flight > 1 (HIP path only): that many frame pairs at once (`register_in_flight`: one stream and one host thread each, a
    blocking call into the library per frame pair; or the one-thread scheduler with asynchronous hand-overs); ms / frame pair is then the wall time of the whole share over its frame pairs --
    throughput of the stream, not the latency of one pair."""
    import torch.distributed as dist
    device = torch.device(device)
    pipelined = int(in_flight) > 1 and register_fn is None
    register_fn = register_fn or register_frame_pair

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    mine = shard_round_robin(paths, rank, world)
    meter = utils_eval.AverageMeter()
    times, matched = [], 0
    pose_sources = {}

    def account(fp, out):
        nonlocal matched
        pose_sources[fp.pose_source] = pose_sources.get(fp.pose_source, 0) + 1
        matched += int(out["pairs"].shape[0])
        if fp.gt_flow is not None:
            m = utils_eval.compute_epe_test(out["flow"].cpu().numpy(), fp.gt_flow, fp.mask)
            n = int((np.asarray(fp.mask) > 0).sum()) if fp.mask is not None else len(fp.gt_flow)
            meter.update(*m, n)

    every = (fp for path in mine for fp in ([path] if isinstance(path, FramePair) else load_any(path, args)))   # a sequence file yields one pair per gap
    if pipelined:
        sync()
        t0 = time.perf_counter()
        n = 0
        for _, fp, out in register_in_flight(args, every, device, in_flight):
            account(fp, out)
            n += 1
        sync()
        times = [(time.perf_counter() - t0) * 1e3 / max(n, 1)] * n
    else:
        for fp in every:
            if repeat > 1:
                register_fn(args, fp, device)                   # untimed pass: page-in, allocator
            sync()
            t0 = time.perf_counter()
            for _ in range(repeat):
                out = register_fn(args, fp, device)
            sync()
            times.append((time.perf_counter() - t0) / repeat * 1e3)
            account(fp, out)
    # five weighted sums + point count + time + frame pairs + matches: one small all_reduce
    local = [getattr(meter, m + "_sum") for m in utils_eval.METRIC_NAMES] + [meter.num, sum(times), len(times), matched]
    on_gpu = world > 1 and dist.get_backend(group) == "nccl"
    tot = torch.tensor(local, dtype=torch.float64, device=device if on_gpu else "cpu")
    tmax = torch.tensor([sum(times)], dtype=torch.float64, device=tot.device)
    if world > 1:
        dist.all_reduce(tot, group=group)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    tot = tot.tolist()
    n_pts, n_fp = tot[5], int(tot[7])
    out = {"frame_pairs": n_fp, "matched_cluster_pairs": int(tot[8]), "n_gpus": world,
           "ms_per_frame_pair": tot[6] / max(n_fp, 1),
           "frame_pairs_per_s": n_fp / (float(tmax.item()) * 1e-3) if n_fp else 0.0,
           "evaluated_points": int(n_pts),
           "pose_sources": pose_sources}     # (this rank's frame pairs: "ego_motion_gt" = ground-truth ego poses were used)
    if n_pts > 0:
        out.update({m: tot[k] / n_pts for k, m in enumerate(utils_eval.METRIC_NAMES)})
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("directory")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--in-flight", type=int, default=1, help="frame pairs registered at once (a stream and a host thread each)")
    def flag(text):      # (type=bool would read every non-empty string, "False" included, as True)
        if text.lower() in ("1", "true", "yes", "on"):
            return True
        if text.lower() in ("0", "false", "no", "off"):
            return False
        raise argparse.ArgumentTypeError(f"expected true/false, got {text!r}")
    for k, v in DEFAULT_ARGS.items():
        kind = str if k == "cluster" else flag if isinstance(v, bool) else float if isinstance(v, float) or v is None else type(v)
        ap.add_argument("--" + k.replace("_", "-"), type=kind, default=v)
    ns = ap.parse_args(argv)
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    args = SimpleNamespace(**{k: getattr(ns, k) for k in DEFAULT_ARGS})
    args.max_points, args.min_cluster_size, args.chunk_size = int(args.max_points), int(args.min_cluster_size), int(args.chunk_size)
    args.num_clusters = int(args.num_clusters)
    summary = run_stream(args, list_frame_pairs(ns.directory), device, rank, world, ns.repeat, in_flight=ns.in_flight)
    if rank == 0:
        print(json.dumps(summary))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
