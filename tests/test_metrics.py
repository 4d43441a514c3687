"""BEV-histogram JSD / MMD.

CPU: oracle/metrics.py reproduces tests/golden/metrics.npz -- outputs of the REFERENCE's point_cloud_to_histogram,
compute_mmd(gaussian) and scipy's jensenshannon captured by oracle/validate_metrics_against_reference.py.
GPU: rangeldm_amd.metrics against the same vectors.  Histogram counts: bit-exact.  Spectral norm squared: 1e-4 relative
(fp32 products, power 2^12 + Rayleigh quotient).  JSD: 1e-12 (fp64 on both sides).  MMD: 2e-4 relative -- it is linear in
the squared distances, and the (1 - k) means are accumulated in fp64 so nothing cancels against 1.
"""
import numpy as np
import pytest
import torch

from oracle import metrics as om


def test_oracle_matches_reference(golden):
    g = golden("metrics")
    for cl, ref in ((g["metrics_clouds_x"], g["metrics_hx_ref"]), (g["metrics_clouds_y"], g["metrics_hy_ref"])):
        for pc, h in zip(cl, ref):
            assert np.array_equal(om.point_cloud_to_histogram(160, 100, om.depth_mask(pc)), h)
    assert np.array_equal(om.point_cloud_to_histogram(160, 100, g["metrics_clouds_x"][0]), g["metrics_hall_ref"])
    mmd = om.compute_mmd(g["metrics_hx_ref"], g["metrics_hy_ref"])
    assert np.allclose(mmd, g["metrics_mmd_ref"], rtol=0, atol=1e-15)
    assert abs(om.jsd(g["metrics_hx_ref"], g["metrics_hy_ref"]) - g["metrics_jsd_ref"][0]) < 1e-15
    # the distance inside the reference's `gaussian` is the SPECTRAL norm of the 2-D difference, not the Frobenius norm
    a = g["metrics_hx_ref"][0] / g["metrics_hx_ref"][0].sum()
    b = g["metrics_hy_ref"][0] / g["metrics_hy_ref"][0].sum()
    lam = g["metrics_lambda_ref"][0, 0]
    assert abs(lam - np.linalg.svd(a - b, compute_uv=False)[0] ** 2) < 1e-18
    assert lam < 0.9 * np.sum((a - b) ** 2)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_hip_histogram_is_exact(golden):
    from rangeldm_amd import metrics as M
    g = golden("metrics")
    hx = M.point_cloud_to_histogram(160, 100, [dev(c) for c in g["metrics_clouds_x"]], 3.0, 70.0)
    hy = M.point_cloud_to_histogram(160, 100, [dev(c) for c in g["metrics_clouds_y"]], 3.0, 70.0)
    assert hx.dtype == torch.int32 and np.array_equal(hx.cpu().numpy(), g["metrics_hx_ref"])
    assert np.array_equal(hy.cpu().numpy(), g["metrics_hy_ref"])
    # no depth mask: points exactly on +-80 m land in the outer bins like np.histogramdd puts them
    h_all = M.point_cloud_to_histogram(160, 100, dev(g["metrics_clouds_x"][0]))
    assert np.array_equal(h_all[0].cpu().numpy(), g["metrics_hall_ref"])
    # ragged batch incl. an empty cloud and xyz-only clouds
    rag = [dev(g["metrics_clouds_x"][1][:, :3]), torch.zeros((0, 3), device="cuda"), dev(g["metrics_clouds_x"][2][:777, :3])]
    hr = M.point_cloud_to_histogram(160, 100, rag, 3.0, 70.0).cpu().numpy()
    assert np.array_equal(hr[0], g["metrics_hx_ref"][1]) and hr[1].sum() == 0
    assert np.array_equal(hr[2], om.point_cloud_to_histogram(160, 100, om.depth_mask(g["metrics_clouds_x"][2][:777])))
    # every integer edge of the grid, both coordinates (searchsorted side='right' semantics)
    e = np.linspace(-80, 80, 101)
    grid = np.stack([np.repeat(e, 101), np.tile(e, 101), np.zeros(101 * 101)], 1).astype(np.float32)
    assert np.array_equal(M.point_cloud_to_histogram(160, 100, dev(grid))[0].cpu().numpy(),
                          om.point_cloud_to_histogram(160, 100, grid))


@pytest.mark.gpu
def test_hip_jsd_and_mmd_match_reference(golden):
    from rangeldm_amd import metrics as M
    g = golden("metrics")
    hx, hy = dev(g["metrics_hx_ref"]), dev(g["metrics_hy_ref"])
    assert abs(M.jsd_2d(hx, hy) - g["metrics_jsd_ref"][0]) < 1e-12
    assert M.jsd_2d(hx, hx) == 0.0
    lam = M.spectral_sq(hx, hy).cpu().numpy()
    ref = g["metrics_lambda_ref"]
    assert np.abs(lam / ref - 1).max() < 1e-4
    sym = M.spectral_sq(hx).cpu().numpy()
    ref_sym = om.spectral_sq(g["metrics_hx_ref"], g["metrics_hx_ref"])
    assert np.array_equal(sym, sym.T) and (np.diag(sym) == 0).all()
    off = ~np.eye(len(sym), dtype=bool)
    assert np.abs(sym[off] / ref_sym[off] - 1).max() < 1e-4
    s1, s2, cross, mmd = M.compute_mmd(hx, hy, return_terms=True)
    r = g["metrics_mmd_ref"]
    assert abs(s1 - r[0]) < 1e-8 and abs(s2 - r[1]) < 1e-8 and abs(cross - r[2]) < 1e-8
    assert abs(mmd / r[3] - 1) < 2e-4
    assert abs(M.compute_mmd(hx, hx)) < 1e-12                      # identical sets
    with pytest.raises(NotImplementedError):
        M.compute_mmd(hx, hy, kernel="gaussian_emd")


@pytest.mark.gpu
def test_hip_spectral_norm_hard_cases():
    """Near-degenerate leading singular values and rank-1 differences (where a plain power iteration stalls)."""
    from rangeldm_amd import metrics as M
    rng = np.random.default_rng(5)
    n = 100
    base = rng.integers(50, 60, (n, n))
    cases = [base.copy() for _ in range(4)]
    cases[1][:50, :50] += 40                      # block structure: two comparable singular values
    cases[1][50:, 50:] += 39
    cases[2][7, :] += 500                         # rank-1 bump
    cases[3] = rng.integers(0, 3, (n, n)) * rng.integers(0, 2, (n, n))      # sparse, many empty bins
    cases[3][0, 0] += 1
    h = np.stack(cases).astype(np.int32)
    lam = M.spectral_sq(dev(h)).cpu().numpy()
    ref = om.spectral_sq(h, h)
    off = ~np.eye(4, dtype=bool)
    assert np.abs(lam[off] / ref[off] - 1).max() < 2e-4
