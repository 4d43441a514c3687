#!/usr/bin/env python3
"""Pin oracle/lidar.py against the reference's own `point_cloud_to_range_image` classes and write
tests/golden/lidar.npz (inputs + outputs computed BY THE REFERENCE CLASSES).

Build-container only (needs /root/reference, read-only).  ldm/dataset.py imports pytorch_lightning for one base class
(`pl.LightningDataModule`, dataset.py:7,382): a stub module with that attribute is registered, then dataset.py,
kitti360_range_image.py and nuscenes_range_image.py are imported by path (they `from dataset import ...`).

    python -m oracle.validate_lidar_against_reference            # check + write the golden file
    python -m oracle.validate_lidar_against_reference --check
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle.lidar import LidarOracle  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")
        pl.LightningDataModule = type("LightningDataModule", (), {})
        sys.modules["pytorch_lightning"] = pl
    ds = _load("dataset", os.path.join(REF, "ldm", "dataset.py"))
    kitti = _load("kitti360_range_image", os.path.join(REF, "ldm", "kitti360_range_image.py"))
    nusc = _load("nuscenes_range_image", os.path.join(REF, "ldm", "nuscenes_range_image.py"))
    return ds, kitti, nusc


FAILS = []


def check(name, a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = float(np.abs(a - b).max()) if a.size else 0.0
    ok = a.shape == b.shape and d <= tol
    print(f"  [{'ok' if ok else 'FAIL'}] {name}: max|diff|={d:.3e} (tol {tol:.1e})")
    if not ok:
        FAILS.append(name)


def synthetic_range_images(rng, B, W, H, mode):
    """Normalised range images the way the sampler emits them (ch0 range code, ch1 remission in [0, 1])."""
    metres = rng.uniform(1.0, 60.0, (B, W, H)).astype(np.float32)
    metres[rng.uniform(size=(B, W, H)) < 0.05] = -3.0         # a few negative ranges: the r < 0 -> 100 branch
    if mode == "log":
        code = np.log2(np.maximum(metres, 0) + 1) / 6
    elif mode == "inverse":
        code = 1 / np.maximum(metres, 0.5)
    else:
        code = (metres - 20.0) / 40.0
    return np.stack([code.astype(np.float32), rng.uniform(0, 1, (B, W, H)).astype(np.float32)], 1)


def synthetic_sweep(rng, to_range, n, width, jitter=0.25, ring_column=False):
    """Returns aimed at pixel centres (+- jitter px) so ulp-level atan2 differences cannot move a return across a pixel
    boundary; several returns per pixel at different ranges exercise nearest-wins."""
    H = to_range.H
    rows = rng.integers(0, H, n)
    cols = rng.integers(0, width, n)
    colf = cols + rng.uniform(-jitter, jitter, n)
    azi = (width - 1.0 + 0.5 - colf) / width * 2 * np.pi - np.pi
    r = rng.uniform(3.0, 95.0, n)
    incl = to_range.incl[rows].astype(np.float64)
    z = to_range.height[rows] - r * np.sin(incl)
    xy = r * np.cos(incl)
    pc = np.stack([xy * np.cos(azi), xy * np.sin(azi), z, rng.uniform(0, 1, n)], 1).astype(np.float32)
    if ring_column:
        pc = np.concatenate([pc, (H - 1 - rows)[:, None].astype(np.float32)], 1)
    return pc, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    ds, kitti, nusc = import_reference()
    rng = np.random.default_rng(20240310)
    gold = {}

    print("== f1: to_pc_torch / to_voxel (ldm/dataset.py:228-294)")
    for tag, cls, mode, W, grid in (("kitti", kitti.point_cloud_to_range_image_KITTI, "linear", 48, [1, 64, 64]),
                                    ("nusc", nusc.point_cloud_to_range_image_nuScenes, "linear", 64, [1, 48, 40]),
                                    ("kittilog", kitti.point_cloud_to_range_image_KITTI, "log", 32, [1, 64, 64]),
                                    ("kittiinv", kitti.point_cloud_to_range_image_KITTI, "inverse", 32, [1, 64, 64]),
                                    ("vol3d", nusc.point_cloud_to_range_image_nuScenes, "linear", 32, [4, 24, 24])):
        ref = cls(width=1024, grid_sizes=grid, log=mode == "log", inverse=mode == "inverse")
        mine = LidarOracle(ref.incl, ref.height, width=1024, grid_sizes=grid, log=mode == "log", inverse=mode == "inverse")
        img = synthetic_range_images(rng, 2, W, ref.H, mode)
        pc_ref = ref.to_pc_torch(torch.from_numpy(img.copy())).numpy()
        check(f"{tag}: to_pc_torch", mine.to_pc(img), pc_ref, 2e-5)
        vox_ref = ref.to_voxel(torch.from_numpy(img.copy())).numpy()
        # the splat alone on the reference's own cloud (fp64 accumulation here, sequential fp32 there) ...
        check(f"{tag}: _splat_points_to_volumes", mine.to_voxel(img, pc=pc_ref), vox_ref, 2e-5)
        # ... and end to end: a 1e-5 shift of a point moves a vote weight by ~1e-5, which `feature / clamp(density, 1e-4)`
        # amplifies by up to 1e4 in nearly empty cells
        check(f"{tag}: to_voxel", mine.to_voxel(img), vox_ref, 2e-3)
        gold[f"lidar_{tag}_img"], gold[f"lidar_{tag}_pc_ref"], gold[f"lidar_{tag}_vox_ref"] = img, pc_ref, vox_ref
        gold[f"lidar_{tag}_grid"] = np.array(grid)

    print("== the per-image tail of ldm/inference.py:171-183")
    pc = gold["lidar_kitti_pc_ref"][0]
    depth = np.linalg.norm(pc[:, :3], 2, axis=1)
    kept = pc[depth < 40.0]                                         # the driver's expression with a threshold that bites
    check("filter_points", LidarOracle.filter_points(pc, 40.0), kept, 0.0)
    gold["lidar_filter_ref"] = kept
    img = torch.from_numpy(gold["lidar_kitti_img"])
    png = (img[0].permute(2, 1, 0).numpy().clip(0, 1) * 255.).astype(np.uint8)[:, :, 0]
    check("render_u8", LidarOracle.render_u8(img[0].numpy()), png, 0.0)
    gold["lidar_png_ref"] = png

    print("== f3: __call__ / process_miss_value / normalize (ldm/dataset.py:159-226)")
    for tag, cls, mode, width, n in (("kitti", kitti.point_cloud_to_range_image_KITTI, "linear", 128, 6000),
                                     ("nusc", nusc.point_cloud_to_range_image_nuScenes, "linear", 96, 2500),
                                     ("kittilog", kitti.point_cloud_to_range_image_KITTI, "log", 64, 3000),
                                     ("kittiinv", kitti.point_cloud_to_range_image_KITTI, "inverse", 64, 3000)):
        ref = cls(width=width, log=mode == "log", inverse=mode == "inverse")
        mine = LidarOracle(ref.incl, ref.height, width=width, log=mode == "log", inverse=mode == "inverse")
        is_nusc = tag == "nusc"
        sweep, rows_true = synthetic_sweep(rng, ref, n, width, ring_column=is_nusc)
        if is_nusc:
            sweep[:40, :3] *= 0.01                                 # some returns inside the 2 m exclusion radius
        rows_ref = ref.get_row_inds(sweep if not is_nusc else sweep[np.linalg.norm(sweep[:, :3], 2, axis=1) > 2.0])
        raw_ref = ref(sweep.copy())                                # the reference shifts pc[:, 2] in place: pass a copy
        if is_nusc:
            keep = np.linalg.norm(sweep[:, :3], 2, axis=1) > 2.0
            raw = mine.project(sweep[keep], 31 - sweep[keep][:, 4].astype(np.int32))
        else:
            rows = mine.row_inds_nearest_beam(sweep)
            check(f"{tag}: get_row_inds", rows, rows_ref, 0)
            raw = mine.project(sweep, rows)
        check(f"{tag}: __call__", raw, raw_ref, 0.0)
        filled_ref, mask_ref, car_ref = ref.process_miss_value(raw_ref.copy())
        filled, mask, car = mine.process_miss_value(raw)
        check(f"{tag}: process_miss_value image", filled, filled_ref, 0.0)
        check(f"{tag}: process_miss_value mask", mask, mask_ref, 0)
        check(f"{tag}: car_window_mask", car, car_ref, 0)
        jpg_ref = torch.from_numpy(ref.normalize(filled_ref.copy())).permute(2, 1, 0).numpy()
        check(f"{tag}: normalize", np.transpose(mine.normalize(filled), (2, 1, 0)), jpg_ref, 0.0)
        gold[f"lidar_proj_{tag}_sweep"] = sweep
        gold[f"lidar_proj_{tag}_jpg_ref"] = jpg_ref
        gold[f"lidar_proj_{tag}_mask_ref"] = mask_ref.T
        gold[f"lidar_proj_{tag}_car_ref"] = car_ref.T
        gold[f"lidar_proj_{tag}_width"] = np.array([width])

    print("== conditions of the conditional path: RangeDataset.__getitem__ (ldm/dataset.py:340-362)")
    import tempfile
    from rangeldm_amd.conditional import downsample_range_image, inpainting_inputs
    cgold = {}
    with tempfile.TemporaryDirectory() as td:
        jpg = torch.from_numpy(rng.standard_normal((2, 64, 16)).astype(np.float32))
        pth = os.path.join(td, "item.pth")
        torch.save({"jpg": jpg, "mask": torch.ones(64, 16), "car_window_mask": torch.ones(64, 16)}, pth)
        for tag, kw in (("up4", dict(downsample=4)), ("up24", dict(downsample=[2, 4])), ("inp", dict(inpainting=0.0625)),
                        ("inpwrap", dict(inpainting=1.25))):
            item = ds.RangeDataset(**kw)
            item.file_paths = ["item"]
            item.get_pth_path = lambda p_: pth                       # the cached-.pth branch of __getitem__ (:321-322)
            ret = item[0]
            if "downsample" in kw:
                check(f"RangeDataset down {kw['downsample']}", downsample_range_image(jpg, kw["downsample"]).numpy(), ret["down"].numpy(), 0.0)
                cgold[f"condds_{tag}_down_ref"] = ret["down"].numpy()
            else:
                m, mi = inpainting_inputs(jpg, kw["inpainting"])
                check(f"RangeDataset inpainting_mask {kw['inpainting']}", m.numpy(), ret["inpainting_mask"].numpy(), 0.0)
                check(f"RangeDataset masked_image {kw['inpainting']}", mi.numpy(), ret["masked_image"].numpy(), 0.0)
                cgold[f"condds_{tag}_mask_ref"] = ret["inpainting_mask"].numpy()
                cgold[f"condds_{tag}_masked_ref"] = ret["masked_image"].numpy()
        cgold["condds_jpg"] = jpg.numpy()

    print(f"\n{'all checks passed' if not FAILS else 'FAILED: ' + ', '.join(FAILS)}")
    if FAILS:
        sys.exit(1)
    if not args.check:
        cpath = os.path.join(GOLD, "condds.npz")
        np.savez_compressed(cpath, **{k: np.ascontiguousarray(v) for k, v in cgold.items()})
        print(f"wrote {cpath} ({os.path.getsize(cpath) / 1024:.0f} KiB)")
        path = os.path.join(GOLD, "lidar.npz")
        np.savez_compressed(path, **{k: np.ascontiguousarray(v) for k, v in gold.items()})
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
