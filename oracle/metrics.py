"""CPU oracle for the BEV-histogram evaluation -- TEST INFRASTRUCTURE, NOT PRODUCT.

numpy / scipy restatement of metrics/metrics/histogram/{histogram.py:4-18, mmd.py:39-44, dist_helper.py:84-104,128-172,
jsd.py:14-16,90-101}.  Pinned against the reference's own functions by oracle/validate_metrics_against_reference.py
(histogram.py imports cleanly; dist_helper.py under a `pyemd` stub and `np.float = float`, which numpy >= 1.24 removed);
golden vectors in tests/golden/metrics.npz.
"""
import numpy as np


def depth_mask(pc, min_depth=3.0, max_depth=70.0):
    """load_point_cloud_xyz (mmd.py:39-44): keep min_depth < |xyz| < max_depth."""
    pc = np.asarray(pc, np.float32)[:, :3]
    depth = np.linalg.norm(pc, 2, axis=1)
    return pc[np.logical_and(depth > min_depth, depth < max_depth), :]


def point_cloud_to_histogram(field_size, bins, point_cloud):
    """histogram.py:4-18 (returns the counts only)."""
    square = field_size / bins
    half = (bins / 2) * square
    return np.histogramdd(point_cloud[:, 0:2], bins=bins, range=([-half, half], [-half, half]))[0]


def gaussian(x, y, sigma=0.5):
    """dist_helper.py:84-104: for 2-D x, y `np.linalg.norm(x - y, 2)` is the largest singular value."""
    dist = np.linalg.norm(np.asarray(x, np.float64) - np.asarray(y, np.float64), 2)
    return np.exp(-dist * dist / (2 * sigma * sigma))


def spectral_sq(hx, hy):
    """(nx, ny) table of sigma_max(pmf_i - pmf_j)^2."""
    px = [h / np.sum(h) for h in np.asarray(hx, np.float64)]
    py = [h / np.sum(h) for h in np.asarray(hy, np.float64)]
    return np.array([[np.linalg.norm(a - b, 2) ** 2 for b in py] for a in px])


def compute_mmd(samples1, samples2, sigma=0.5):
    """dist_helper.py:156-172 with kernel=gaussian, is_hist=True.  Returns (s1, s2, cross, s1 + s2 - 2 cross)."""
    s1 = [s / np.sum(s) for s in np.asarray(samples1, np.float64)]
    s2 = [s / np.sum(s) for s in np.asarray(samples2, np.float64)]

    def disc(a, b):
        return sum(gaussian(x, y, sigma) for x in a for y in b) / (len(a) * len(b))
    a, b, c = disc(s1, s1), disc(s2, s2), disc(s1, s2)
    return a, b, c, a + b - 2 * c


def jsd(hx, hy):
    """jsd.py:90-101: sum the histograms of each set, normalise, scipy jensenshannon (sqrt of the base-e divergence)."""
    from scipy.spatial.distance import jensenshannon
    p = np.sum(np.asarray(hx, np.float64), axis=0)
    q = np.sum(np.asarray(hy, np.float64), axis=0)
    return float(jensenshannon((p / p.sum()).flatten(), (q / q.sum()).flatten()))
