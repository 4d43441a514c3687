"""Range image <-> point cloud.

CPU: oracle/lidar.py reproduces tests/golden/lidar.npz, the outputs of the REFERENCE's own
`point_cloud_to_range_image` classes captured by oracle/validate_lidar_against_reference.py.
GPU: rangeldm_amd.range_image (HIP kernels behind the C ABI) against the same golden vectors, against the oracle on
seeded inputs, and -- at the full 64 x 1024 / 1024 x 1024 sizes -- through size-independent properties.

Tolerances.  Integer / byte / index results (projection, masks, 8-bit rendering, the ordered depth filter) are
bit-exact.  xyz: 2e-5 absolute on coordinates of up to 100 m (cos / sin of the beam and azimuth angles come from
different libm implementations: <= 1 ulp each).  BEV volume: density 2e-5; features 2e-3 because
`feature / clamp(density, 1e-4)` amplifies a 1e-5 vote-weight change by up to 1e4 in nearly empty cells.
"""
import numpy as np
import pytest
import torch

from oracle.lidar import LidarOracle
from rangeldm_amd import range_image as RI

F1_CASES = [("kitti", "KITTI", {}), ("nusc", "nuScenes", {}), ("kittilog", "KITTI", {"log": True}),
            ("kittiinv", "KITTI", {"inverse": True}), ("vol3d", "nuScenes", {})]
F3_CASES = [("kitti", "KITTI", {}), ("nusc", "nuScenes", {}), ("kittilog", "KITTI", {"log": True}),
            ("kittiinv", "KITTI", {"inverse": True})]
CLS = {"KITTI": RI.point_cloud_to_range_image_KITTI, "nuScenes": RI.point_cloud_to_range_image_nuScenes}


def oracle_for(sensor, **kw):
    t = CLS[sensor](**kw)                            # tables only: the device handle is created on first use
    return LidarOracle(t.incl, t.height, width=t.width, grid_sizes=t.grid_sizes, log=t.log, inverse=t.inverse)


# ---- CPU: the oracle against the reference's outputs -------------------------------------------------------------
@pytest.mark.parametrize("tag,sensor,kw", F1_CASES)
def test_oracle_to_pc_and_voxel_match_reference(golden, tag, sensor, kw):
    g = golden("lidar")
    o = oracle_for(sensor, grid_sizes=list(g[f"lidar_{tag}_grid"]), **kw)
    img = g[f"lidar_{tag}_img"]
    assert np.abs(o.to_pc(img) - g[f"lidar_{tag}_pc_ref"]).max() < 2e-5
    assert np.abs(o.to_voxel(img, pc=g[f"lidar_{tag}_pc_ref"]) - g[f"lidar_{tag}_vox_ref"]).max() < 2e-5
    assert np.abs(o.to_voxel(img) - g[f"lidar_{tag}_vox_ref"]).max() < 2e-3


def test_oracle_driver_tail_matches_reference(golden):
    g = golden("lidar")
    assert np.array_equal(LidarOracle.filter_points(g["lidar_kitti_pc_ref"][0], 40.0), g["lidar_filter_ref"])
    assert np.array_equal(LidarOracle.render_u8(g["lidar_kitti_img"][0]), g["lidar_png_ref"])


@pytest.mark.parametrize("tag,sensor,kw", F3_CASES)
def test_oracle_projection_matches_reference(golden, tag, sensor, kw):
    g = golden("lidar")
    o = oracle_for(sensor, width=int(g[f"lidar_proj_{tag}_width"][0]), **kw)
    sweep = g[f"lidar_proj_{tag}_sweep"]
    if sensor == "nuScenes":
        keep = np.linalg.norm(sweep[:, :3], 2, axis=1) > 2.0
        raw = o.project(sweep[keep], 31 - sweep[keep][:, 4].astype(np.int32))
    else:
        raw = o.project(sweep, o.row_inds_nearest_beam(sweep))
    filled, mask, car = o.process_miss_value(raw)
    assert np.array_equal(np.transpose(o.normalize(filled), (2, 1, 0)), g[f"lidar_proj_{tag}_jpg_ref"])
    assert np.array_equal(mask.T, g[f"lidar_proj_{tag}_mask_ref"])
    assert np.array_equal(car.T, g[f"lidar_proj_{tag}_car_ref"])


def test_range_image_classes_need_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    t = RI.point_cloud_to_range_image_KITTI()
    assert t.H == 64 and t.incl.dtype == np.float32 and t.grid_sizes == [1, 1024, 1024]
    with pytest.raises(RuntimeError, match="GPU"):
        t.to_pc_torch(torch.zeros(1, 2, 8, 64))
    with pytest.raises(NotImplementedError):
        RI.point_cloud_to_range_image().get_row_inds(None)


# ---- GPU: HIP kernels against the reference's outputs ------------------------------------------------------------
def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,sensor,kw", F1_CASES)
def test_hip_to_pc_and_voxel_match_reference(golden, tag, sensor, kw):
    g = golden("lidar")
    t = CLS[sensor](grid_sizes=list(g[f"lidar_{tag}_grid"]), **kw)
    img = dev(g[f"lidar_{tag}_img"])
    before = img.clone()
    pc = t.to_pc_torch(img)
    assert torch.equal(img, before)                                  # inputs are never mutated
    assert pc.shape == g[f"lidar_{tag}_pc_ref"].shape
    assert np.abs(pc.cpu().numpy() - g[f"lidar_{tag}_pc_ref"]).max() < 2e-5
    vox = t.to_voxel(img).cpu().numpy()
    ref = g[f"lidar_{tag}_vox_ref"]
    D = ref.shape[1] // 2
    assert vox.shape == ref.shape
    assert np.abs(vox[:, :D] - ref[:, :D]).max() < 2e-5              # density planes
    assert np.abs(vox[:, D:] - ref[:, D:]).max() < 2e-3              # density-normalised remission
    # one-channel images give xyz only (ldm/dataset.py:273-276)
    pc3 = t.to_pc_torch(img[:, :1])
    assert pc3.shape[2] == 3 and torch.equal(pc3, pc[:, :, :3])
    with pytest.raises(RuntimeError, match="remission"):
        t.to_voxel(img[:, :1])


@pytest.mark.gpu
def test_hip_driver_tail_matches_reference(golden):
    g = golden("lidar")
    t = RI.point_cloud_to_range_image_KITTI()
    pc = dev(g["lidar_kitti_pc_ref"])
    out, counts = t.filter_points(pc, 40.0)
    n = int(counts[0])
    assert n == len(g["lidar_filter_ref"])
    assert np.array_equal(out[0, :n].cpu().numpy(), g["lidar_filter_ref"])          # same rows, same order, same bits
    ref1 = LidarOracle.filter_points(g["lidar_kitti_pc_ref"][1], 40.0)
    assert int(counts[1]) == len(ref1) and np.array_equal(out[1, :len(ref1)].cpu().numpy(), ref1)
    xyz, c3 = t.filter_points(pc[:, :, :3].contiguous(), 40.0)                         # 3-column clouds
    assert torch.equal(c3, counts) and torch.equal(xyz[0, :n], out[0, :n, :3])
    png = RI.render_u8(dev(g["lidar_kitti_img"]))
    assert png.dtype == torch.uint8 and np.array_equal(png[0].cpu().numpy(), g["lidar_png_ref"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,sensor,kw", F3_CASES)
def test_hip_projection_matches_reference(golden, tag, sensor, kw):
    g = golden("lidar")
    t = CLS[sensor](width=int(g[f"lidar_proj_{tag}_width"][0]), **kw)
    sweep = dev(g[f"lidar_proj_{tag}_sweep"])
    before = sweep.clone()
    out = t.project(sweep)
    assert torch.equal(sweep, before)
    jpg, ref = out["jpg"].cpu().numpy(), g[f"lidar_proj_{tag}_jpg_ref"]
    assert np.array_equal(out["mask"].cpu().numpy(), g[f"lidar_proj_{tag}_mask_ref"])
    assert np.array_equal(out["car_window_mask"].cpu().numpy(), g[f"lidar_proj_{tag}_car_ref"])
    assert np.array_equal(jpg[1], ref[1])                                             # remission: copied bits
    if kw.get("log"):
        assert np.abs(jpg[0] - ref[0]).max() < 1e-6                                  # log2f: device libm vs numpy
    else:
        assert np.array_equal(jpg[0], ref[0])


@pytest.mark.gpu
def test_hip_projection_edge_cases():
    t = RI.point_cloud_to_range_image_KITTI(width=64)
    empty = t.project(torch.zeros((0, 4), device="cuda"))                             # empty sweep: everything filled
    assert torch.all(empty["jpg"][0] == (100.0 - 20.0) / 40.0) and torch.all(empty["jpg"][1] == 0)
    assert not empty["mask"].any() and not empty["car_window_mask"].any()
    # two returns in one pixel: the nearer one survives whatever their order (ldm/dataset.py:173-185)
    o = oracle_for("KITTI", width=64)
    r = np.array([30.0, 12.0, 50.0], np.float32)
    inc = float(t.incl[10])
    pts = np.stack([r * np.cos(inc), np.zeros(3, np.float32), t.height[10] - r * np.sin(inc),
                    np.array([0.1, 0.2, 0.3], np.float32)], 1).astype(np.float32)
    for perm in ([0, 1, 2], [2, 1, 0], [1, 0, 2]):
        got = t.project(dev(pts[perm]))["jpg"].cpu().numpy()
        raw = o.project(pts[perm], o.row_inds_nearest_beam(pts[perm]))
        want = np.transpose(o.normalize(o.process_miss_value(raw)[0]), (2, 1, 0))
        assert np.array_equal(got, want)
        w, h = np.argwhere(got[1] == np.float32(0.2))[0]
        assert h == 10 and abs(got[0, w, h] * 40 + 20 - 12.0) < 1e-3
    # exact range ties: the later return wins (stable farthest-first order)
    tie = np.repeat(pts[:1], 2, 0)
    tie[1, 3] = 0.9
    assert (t.project(dev(tie))["jpg"][1] == 0.9).sum() >= 1


@pytest.mark.gpu
def test_hip_random_sweep_against_oracle():
    """Unstructured returns (no pixel-centre aiming): an ulp of atan2 may move a return across a pixel boundary, so a
    handful of pixels may differ; everything else is exact."""
    rng = np.random.default_rng(7)
    n = 120000
    r = rng.uniform(2.5, 80.0, n)
    azi = rng.uniform(-np.pi, np.pi, n)
    inc = rng.uniform(-0.43, 0.03, n)
    pts = np.stack([r * np.cos(inc) * np.cos(azi), r * np.cos(inc) * np.sin(azi), 0.18 - r * np.sin(inc),
                    rng.uniform(0, 1, n)], 1).astype(np.float32)
    t = RI.point_cloud_to_range_image_KITTI()
    o = oracle_for("KITTI")
    got = t.project(dev(pts))
    raw = o.project(pts, o.row_inds_nearest_beam(pts))
    filled, mask, car = o.process_miss_value(raw)
    want = np.transpose(o.normalize(filled), (2, 1, 0))
    differing = int((got["jpg"].cpu().numpy() != want).any(0).sum())
    assert differing <= 64 * 1024 * 1e-3, differing
    assert int((got["mask"].cpu().numpy() != mask.T).sum()) <= 64


@pytest.mark.gpu
def test_hip_full_size_properties():
    """BASELINE config-2 output size (16, 2, 1024, 64), loader-default BEV grid 1024 x 1024."""
    rng = np.random.default_rng(11)
    B, W, H = 16, 1024, 64
    metres = rng.uniform(1.0, 120.0, (B, W, H)).astype(np.float32)
    img = np.stack([(metres - 20) / 40, rng.uniform(0, 1, (B, W, H)).astype(np.float32)], 1)
    t = RI.point_cloud_to_range_image_KITTI()
    x = dev(img)
    pc = t.to_pc_torch(x)
    assert pc.shape == (B, W * H, 4) and torch.isfinite(pc).all()
    # |xyz - (0, 0, height[h])| is the decoded range
    hgt = torch.from_numpy(t.height).cuda().repeat(W)[None, :]
    rr = torch.sqrt(pc[..., 0] ** 2 + pc[..., 1] ** 2 + (pc[..., 2] - hgt) ** 2)
    assert (rr - dev(metres).reshape(B, -1)).abs().max() < 2e-4
    # ordered filter: count and content against a torch mask on the device
    out, counts = t.filter_points(pc, 90.0)
    keep = torch.linalg.vector_norm(pc[..., :3], dim=2) < 90.0
    assert torch.equal(counts.long(), keep.sum(1))
    for b in (0, B - 1):
        assert torch.equal(out[b, :int(counts[b])], pc[b][keep[b]])
    # BEV: the votes of a point sum to 1 wherever all its cells are inside; total density is conserved
    vox = t.to_voxel(x)
    assert vox.shape == (B, 2, 1024, 1024) and torch.isfinite(vox).all()
    inside = ((pc[..., 0].abs() < 25.5) & (pc[..., 1].abs() < 25.5)).sum(1).double()
    total = torch.expm1(vox[:, 0].double()).sum((1, 2))
    assert ((total - inside).abs() / inside.clamp(min=1)).max() < 2e-2      # border cells lose part of their votes
    feat = vox[:, 1]
    assert feat.min() >= 0 and feat.max() <= 1.0 + 1e-4                    # convex combinations of remission in [0, 1]
    # images are independent: the same image in two batch slots gives the same volume (up to the atomics' summation order)
    twice = t.to_voxel(torch.cat([x[:1], x[5:6], x[:1]], 0))
    assert (twice[0] - twice[2]).abs().max() < 1e-4 and (twice[0] - vox[0]).abs().max() < 1e-4
    png = RI.render_u8(vox)
    assert png.shape == (B, 1024, 1024)
    assert torch.equal(png[3], (vox[3].permute(2, 1, 0).clip(0, 1) * 255.).to(torch.uint8)[:, :, 0])
    # round trip: project the decoded cloud back -> the same range image wherever a return landed alone
    back = t.project(pc[0])
    near = dev(metres[0]) < 99.0                                           # farther returns clamp to range_fill_value
    same = (back["jpg"][0] - x[0, 0]).abs() < 1e-4
    assert same[near].float().mean() > 0.999
    assert torch.equal(back["jpg"][1][near], x[0, 1][near])


def test_png_fallback_encoder_round_trips(tmp_path):
    """rangeldm_amd.inference.save_png without PIL writes a valid 8-bit grayscale PNG (zlib stream checked by hand)."""
    import builtins
    import struct
    import zlib
    from rangeldm_amd.inference import save_png
    px = (np.arange(40 * 24) % 251).astype(np.uint8).reshape(24, 40)
    real = builtins.__import__

    def no_pil(name, *a, **k):
        if name == "PIL":
            raise ImportError(name)
        return real(name, *a, **k)
    builtins.__import__ = no_pil
    try:
        save_png(px, str(tmp_path / "t.png"))
    finally:
        builtins.__import__ = real
    blob = (tmp_path / "t.png").read_bytes()
    assert blob[:8] == b"\x89PNG\r\n\x1a\n"
    w, h, depth, ctype = struct.unpack(">IIBB", blob[16:26])
    assert (w, h, depth, ctype) == (40, 24, 8, 0)
    idat = blob.index(b"IDAT")
    n = struct.unpack(">I", blob[idat - 4:idat])[0]
    raw = zlib.decompress(blob[idat + 4:idat + 4 + n])
    rows = np.frombuffer(raw, np.uint8).reshape(24, 41)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:], px)


@pytest.mark.gpu
def test_inference_driver_writes_reference_outputs(tmp_path):
    """rangeldm_amd.inference end to end (ldm/inference.py:154-183): file set, index arithmetic and file contents."""
    from rangeldm_amd import inference
    out = tmp_path / "generated"
    inference.main(["--cfg", "RangeLDM", "--samples", "3", "--batch_size", "2", "--out", str(out), "--save-npy"])
    names = sorted(p.name for p in out.iterdir())
    assert names == sorted([f"{i}{s}" for i in range(3) for s in (".bin", ".png", "_range.png", ".npy")])
    o = oracle_for("KITTI")
    for i in range(3):
        img = np.load(out / f"{i}.npy")
        assert img.shape == (2, 1024, 64)
        want = LidarOracle.filter_points(o.to_pc(img[None])[0], 90.0)
        got = np.fromfile(out / f"{i}.bin", np.float32).reshape(-1, 4)
        assert abs(len(got) - len(want)) <= 2                               # a return within an ulp of 90 m may flip
        if len(got) == len(want):
            assert np.abs(got - want).max() < 2e-4
        try:
            from PIL import Image
        except ImportError:
            continue
        assert np.array_equal(np.array(Image.open(out / f"{i}_range.png")), LidarOracle.render_u8(img))
        bev = np.array(Image.open(out / f"{i}.png"))
        assert bev.shape == (1024, 1024)


@pytest.mark.gpu
def test_conditional_inference_driver_writes_reference_layout(tmp_path):
    """python -m rangeldm_amd.inference_conditional (ldm/inference_conditional.py counterpart, BASELINE config 4 shapes, batch 2):
    `<j>_seed_<seed>.bin / .png` in densification_result/, and target / input pictures for seed 0."""
    import os
    from rangeldm_amd import inference_conditional as IC
    out = str(tmp_path / "generated")
    IC.main(["--cfg", "upsample", "--batch_size", "2", "--samples", "2", "--out", out])
    for d in ("densification_result", "densification_target", "densification_input"):
        for j in range(2):
            for ext in ("bin", "png"):
                assert os.path.getsize(os.path.join(out, d, f"{j}_seed_0.{ext}")) > 0, (d, j, ext)
    assert os.path.exists(os.path.join(out, "densification_result", "1_seed_1.bin"))      # samples // B // world + 1 iterations
    pts = np.fromfile(os.path.join(out, "densification_target", "0_seed_0.bin"), dtype=np.float32).reshape(-1, 4)
    assert len(pts) > 1000 and float(np.linalg.norm(pts[:, :3], axis=1).max()) < 70.0      # KITTI-360 range limit of the driver
    sparse = np.fromfile(os.path.join(out, "densification_input", "0_seed_0.bin"), dtype=np.float32).reshape(-1, 4)
    assert 0 < len(sparse) < len(pts)                                                       # a quarter of the beams carry returns
