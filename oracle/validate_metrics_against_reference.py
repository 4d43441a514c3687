#!/usr/bin/env python3
"""Pin oracle/metrics.py against the reference's metrics/metrics/histogram/{histogram.py, dist_helper.py} and scipy, and
write tests/golden/metrics.npz (inputs + outputs computed BY THE REFERENCE FUNCTIONS).  Build-container only.

dist_helper.py imports `pyemd` (absent; only its EMD kernels use it) and calls `np.float` (removed in numpy 1.24): a stub
module and `np.float = float` are installed for the import, nothing else is touched.

    python -m oracle.validate_metrics_against_reference [--check]
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/metrics/metrics/histogram"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import metrics as om  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synthetic_clouds(rng, n_clouds, n_pts, spread):
    out = []
    for _ in range(n_clouds):
        r = np.abs(rng.normal(0, spread, n_pts)) + 1.0
        a = rng.uniform(-np.pi, np.pi, n_pts)
        pc = np.stack([r * np.cos(a), r * np.sin(a), rng.normal(-1, 0.5, n_pts), rng.uniform(0, 1, n_pts)], 1).astype(np.float32)
        pc[:4, 0] = [80.0, -80.0, 79.99999, 0.0]           # on the outer edges / just inside (depth mask drops |xyz| >= 70)
        pc[:4, 1] = [0.0, 0.0, 80.0, -80.0]
        pc[4:8, :2] = np.array([[1.6, -1.6], [4.8, 0.0], [3.2, 1.6 * 7], [-30.4, 30.4]], np.float32)   # on inner edges
        pc[8:12, :2] = np.array([[48.0, 3.0], [-48.0, 3.0], [3.0, 49.6], [3.0, -49.6]], np.float32)
        out.append(pc)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    sys.modules.setdefault("pyemd", types.ModuleType("pyemd"))
    if not hasattr(np, "float"):
        np.float = float
    rh = _load("ref_histogram", os.path.join(REF, "histogram.py"))
    rd = _load("ref_dist_helper", os.path.join(REF, "dist_helper.py"))
    rng = np.random.default_rng(20240310)
    fails = []

    def check(name, a, b, tol):
        d = float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
        ok = d <= tol
        print(f"  [{'ok' if ok else 'FAIL'}] {name}: max|diff|={d:.3e} (tol {tol:.1e})")
        if not ok:
            fails.append(name)

    clouds_x = synthetic_clouds(rng, 6, 3000, 18.0)
    clouds_y = synthetic_clouds(rng, 5, 3000, 25.0)
    hx_ref, hy_ref = [], []
    for src, dst in ((clouds_x, hx_ref), (clouds_y, hy_ref)):
        for pc in src:
            xyz = pc[:, :3]
            depth = np.linalg.norm(xyz, 2, axis=1)                      # load_point_cloud_xyz, mmd.py:39-44
            kept = xyz[np.logical_and(depth > 3.0, depth < 70.0), :]
            h = rh.point_cloud_to_histogram(160, 100, kept)[0]
            check("point_cloud_to_histogram", om.point_cloud_to_histogram(160, 100, om.depth_mask(pc)), h, 0)
            dst.append(h)
    # the edge cases without the depth mask (points on +-80 m)
    h_all = rh.point_cloud_to_histogram(160, 100, clouds_x[0][:, :3])[0]
    check("point_cloud_to_histogram (no mask, outer edges)", om.point_cloud_to_histogram(160, 100, clouds_x[0]), h_all, 0)
    mmd_ref = rd.compute_mmd(hx_ref, hy_ref, rd.gaussian, is_hist=True, is_parallel=False)
    mine = om.compute_mmd(hx_ref, hy_ref)
    check("compute_mmd(gaussian, is_hist)", mine[3], mmd_ref, 1e-15)
    k01 = rd.gaussian(hx_ref[0] / hx_ref[0].sum(), hy_ref[1] / hy_ref[1].sum())
    check("gaussian (spectral norm)", om.gaussian(hx_ref[0] / hx_ref[0].sum(), hy_ref[1] / hy_ref[1].sum()), k01, 0)
    from scipy.spatial.distance import jensenshannon
    p, q = np.sum(hx_ref, 0), np.sum(hy_ref, 0)
    jsd_ref = jensenshannon((p / p.sum()).flatten(), (q / q.sum()).flatten())      # jsd_2d, jsd.py:14-16
    check("jsd_2d", om.jsd(hx_ref, hy_ref), jsd_ref, 0)
    print("all checks passed" if not fails else "FAILED: " + ", ".join(fails))
    if fails:
        sys.exit(1)
    if not args.check:
        path = os.path.join(ROOT, "tests", "golden", "metrics.npz")
        np.savez_compressed(path, metrics_clouds_x=np.stack(clouds_x), metrics_clouds_y=np.stack(clouds_y),
                            metrics_hx_ref=np.stack(hx_ref).astype(np.int32), metrics_hy_ref=np.stack(hy_ref).astype(np.int32),
                            metrics_hall_ref=h_all.astype(np.int32),
                            metrics_mmd_ref=np.array([mine[0], mine[1], mine[2], mmd_ref]),
                            metrics_lambda_ref=om.spectral_sq(hx_ref, hy_ref), metrics_jsd_ref=np.array([jsd_ref]))
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
