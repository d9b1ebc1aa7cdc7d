"""Drop-in for the reference's utils_track.track (utils_track.py:31-35)."""
from .utils_match import match_pcds


def track(args, point_src, point_dst, label_src, label_dst):
    pairs, transformations = match_pcds(args, point_src, point_dst, label_src, label_dst)
    return pairs, transformations
