"""BEV-histogram evaluation on the GPU: the host-side mirror of
metrics/metrics/histogram/{histogram.py, dist_helper.py, mmd.py, jsd.py} -- same function names and argument meaning,
device tensors in, librangeldm_hip.so (rangeldm_amd/csrc/metrics.hip) underneath.  No CPU fallback.

    hists = point_cloud_to_histogram(160, 100, clouds)            # list of (n_i, >= 3) device tensors -> (S, 100, 100)
    jsd   = jsd_2d(model_hists, data_hists)
    mmd   = compute_mmd(data_hists, model_hists)                  # gaussian kernel, sigma = 0.5, is_hist=True
"""
import ctypes as C

import torch

from . import _lib


def _dev_u32(h):
    if not h.is_cuda:
        raise RuntimeError("histograms must live on the GPU (rangeldm_amd has no CPU path)")
    if h.dtype != torch.int32:
        h = h.to(torch.int32)
    return h.contiguous()


def point_cloud_to_histogram(field_size, bins, point_cloud, min_depth=None, max_depth=None):
    """histogram.py:4-18 for one cloud (n, >= 3) or a list of clouds; returns (S, bins, bins) int32 counts on the device.
    min_depth / max_depth: the `load_point_cloud_xyz` mask (mmd.py:39-44: 3 / 70 m for KITTI-360, 2 / 90 m for nuScenes),
    fused into the pass; None keeps every point."""
    clouds = [point_cloud] if torch.is_tensor(point_cloud) else list(point_cloud)
    if not clouds:
        raise ValueError("no point clouds")
    stride = clouds[0].shape[1]
    for c in clouds:
        if not c.is_cuda:
            raise RuntimeError("point clouds must live on the GPU (rangeldm_amd has no CPU path)")
        if c.dim() != 2 or c.shape[1] != stride or stride < 3:
            raise ValueError("every cloud must be (n, k) with the same k >= 3")
    _lib.require_gpu()
    dev = clouds[0].device
    pts = torch.cat([c.detach().float() for c in clouds], 0).contiguous()
    counts = torch.tensor([0] + [c.shape[0] for c in clouds], dtype=torch.int64).cumsum(0).to(torch.int32).to(dev)
    hist = torch.empty((len(clouds), bins, bins), dtype=torch.int32, device=dev)
    lo = -1.0 if min_depth is None else float(min_depth)
    hi = float("inf") if max_depth is None else float(max_depth)
    _lib.check(_lib.lib().rldm_bev_histogram(pts.data_ptr(), counts.data_ptr(), len(clouds), stride, float(field_size),
                                             int(bins), lo, hi, hist.data_ptr(), _lib.stream_ptr(dev)),
               "rldm_bev_histogram")
    return hist


def jsd_2d(hists_p, hists_q):
    """jsd.py:90-101: Jensen-Shannon distance between the summed, normalised histogram sets (S, bins, bins)."""
    p, q = _dev_u32(hists_p), _dev_u32(hists_q)
    out = C.c_double(0.0)
    _lib.check(_lib.lib().rldm_hist_jsd(p.data_ptr(), p.shape[0], q.data_ptr(), q.shape[0], p.shape[1], C.byref(out),
                                        _lib.stream_ptr(p.device)), "rldm_hist_jsd")
    return out.value


def spectral_sq(hists_x, hists_y=None):
    """(nx, ny) fp32 table of `np.linalg.norm(pmf_i - pmf_j, 2) ** 2` (the distance inside dist_helper.gaussian)."""
    x = _dev_u32(hists_x)
    sym = hists_y is None
    y = x if sym else _dev_u32(hists_y)
    lam = torch.empty((x.shape[0], y.shape[0]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().rldm_hist_spectral_sq(x.data_ptr(), x.shape[0], y.data_ptr(), y.shape[0], x.shape[1],
                                                1 if sym else 0, lam.data_ptr(), _lib.stream_ptr(x.device)),
               "rldm_hist_spectral_sq")
    if sym:
        lam = lam + lam.t()
    return lam


def compute_mmd(samples1, samples2, kernel="gaussian", is_hist=True, sigma=0.5, return_terms=False):
    """dist_helper.py:156-172: s1 + s2 - 2 cross with the gaussian kernel on the spectral norm of pmf differences."""
    if kernel != "gaussian" or not is_hist:
        raise NotImplementedError("only compute_mmd(..., gaussian, is_hist=True), the call the reference's metric makes")
    x, y = _dev_u32(samples1), _dev_u32(samples2)
    out = (C.c_double * 4)()
    _lib.check(_lib.lib().rldm_hist_mmd(x.data_ptr(), x.shape[0], y.data_ptr(), y.shape[0], x.shape[1], float(sigma), out,
                                        _lib.stream_ptr(x.device)), "rldm_hist_mmd")
    return tuple(out) if return_terms else out[3]


def load_bin(path, columns=4, device="cuda"):
    """`np.fromfile(file, dtype=np.float32).reshape(-1, columns)` onto the device (mmd.py:40, :73)."""
    import numpy as np
    return torch.from_numpy(np.fromfile(path, dtype=np.float32).reshape(-1, columns)).to(device)


def evaluate_folders(sample_folder, data_files, nuscenes=False, limit=None):
    """calculate_jsd / calculate_mmd (jsd.py:64-101, mmd.py:96-125; *_nus variants :18-62 / :59-94): generated `.bin`
    files of `sample_folder` against the LiDAR sweeps `data_files` (the caller picks and shuffles them the way the
    reference does from its dataset directories).  Returns dict(jsd=..., mmd=...)."""
    import glob
    samples = sorted(glob.glob(f"{sample_folder}/*.bin"))[:limit]
    lo, hi = (2.0, 90.0) if nuscenes else (3.0, 70.0)
    model = point_cloud_to_histogram(160, 100, [load_bin(f, 4) for f in samples], lo, hi)
    data = point_cloud_to_histogram(160, 100, [load_bin(f, 5 if nuscenes else 4) for f in data_files[:len(samples)]], lo, hi)
    return {"jsd": jsd_2d(data, model), "mmd": compute_mmd(data, model)}
