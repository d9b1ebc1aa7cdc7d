"""Scene-flow accuracy metrics of the reference's evaluation (utils_eval.py:65-182), numpy on the
host like the reference's.  Pinned by tests/golden/g9_epe.npz."""
import numpy as np

METRIC_NAMES = ("epe", "accs", "accr", "outlier", "Routlier")


def compute_epe_test(flow_pred, flow_gt, mask=None):
    """utils_eval.py:137-182 -> (EPE3D, strict accuracy, relaxed accuracy, outliers, R-outliers).

    Per point e = |gt - pred|, r = e / (|gt| + 1e-20):  strict e < 0.05 or r < 0.05;  relaxed
    e < 0.1 or r < 0.1;  outlier e > 0.3 or r > 0.1;  R-outlier e > 0.3 and r > 0.3.  Fractions are
    float32 means, as in the reference."""
    flow_pred, flow_gt = np.asarray(flow_pred), np.asarray(flow_gt)
    assert flow_gt.shape[-1] == 3 and flow_pred.shape[-1] == 3
    if mask is not None:
        keep = np.asarray(mask) > 0
        flow_gt, flow_pred = flow_gt[keep], flow_pred[keep]
    err = np.linalg.norm(flow_gt - flow_pred, axis=-1)
    rel = err / (np.linalg.norm(flow_gt, axis=-1) + 1e-20)

    def frac(cond):
        return cond.astype(np.float32).mean()

    return (err.mean(), frac((err < 0.05) | (rel < 0.05)), frac((err < 0.1) | (rel < 0.1)),
            frac((err > 0.3) | (rel > 0.1)), frac((err > 0.3) & (rel > 0.3)))


def average_meter(errors, nums):
    """utils_eval.py:65-80: point-count weighted mean of per-frame errors."""
    assert len(errors) == len(nums)
    return sum(e * n for e, n in zip(errors, nums)) / sum(nums)


class AverageMeter:
    """utils_eval.py:82-135: running point-weighted averages of the five metrics; keeps the
    per-frame values (`*_data`) like the reference."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.num = 0
        self.num_data = []
        for m in METRIC_NAMES:
            setattr(self, m + "_sum", 0.0)
            setattr(self, m + "_avg", 0.0)
            setattr(self, m + "_data", [])

    def update(self, epe, accs, accr, outlier, Routlier, num):
        self.num += num
        self.num_data.append(num)
        for m, v in zip(METRIC_NAMES, (epe, accs, accr, outlier, Routlier)):
            total = getattr(self, m + "_sum") + v * num
            setattr(self, m + "_sum", total)
            setattr(self, m + "_avg", total / self.num)
            getattr(self, m + "_data").append(v)

    def averages(self):
        return {m: float(getattr(self, m + "_avg")) for m in METRIC_NAMES}
