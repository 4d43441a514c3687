"""CPU oracle for the range-image <-> point-cloud rows -- TEST INFRASTRUCTURE, NOT PRODUCT.

numpy restatement of `point_cloud_to_range_image` (ldm/dataset.py:135-294), `_splat_points_to_volumes`
(ldm/dataset.py:13-132) and the per-image post-processing of the sampling driver (ldm/inference.py:171-183).
Pinned against the reference's own classes (imported under a pytorch_lightning stub) by
oracle/validate_lidar_against_reference.py; golden vectors in tests/golden/lidar.npz.
"""
import numpy as np

F = np.float32


class LidarOracle:
    """ldm/dataset.py:136-154 state: per-beam inclination / height tables, (mean, std) = (20, 40), fill (100, 0)."""

    def __init__(self, incl, height, width=1024, grid_sizes=(1, 1024, 1024), pc_range=(-25.6, -25.6, -3., 25.6, 25.6, 1.),
                 log=False, inverse=False, normalize_volume_densities=True):
        self.incl = np.asarray(incl, F)
        self.height = np.asarray(height, F)
        self.H = len(self.incl)
        self.width = width
        self.grid_sizes = tuple(grid_sizes)
        self.pc_range = np.asarray(pc_range, F)
        self.log, self.inverse = log, inverse
        self.normalize_volume_densities = normalize_volume_densities
        self.mean, self.std = F(20.), F(40.)
        self.range_fill_value = np.array([100, 0])

    # ---- f1: range image -> points (ldm/dataset.py:228-278) ---------------------------------------------------
    def to_pc(self, img):
        img = np.asarray(img, F)
        B, C, W, H = img.shape
        if self.log:
            r = np.exp2(img[:, 0] * F(6)) - F(1)
        elif self.inverse:
            r = F(1) / np.maximum(img[:, 0], F(0.0001))
        else:
            r = img[:, 0] * self.std + self.mean
        r = np.where(r < 0, F(self.range_fill_value[0]), r).astype(F)
        z = (self.height[None, None, :] - r * np.sin(self.incl)[None, None, :]).astype(F)
        xy = (r * np.cos(self.incl)[None, None, :]).astype(F)
        azi = self.azimuth(W)
        x = (xy * np.cos(azi)[None, :, None]).astype(F)
        y = (xy * np.sin(azi)[None, :, None]).astype(F)
        cols = [x.reshape(B, -1), y.reshape(B, -1), z.reshape(B, -1)]
        if C > 1:
            cols.append(img[:, 1].reshape(B, -1))
        return np.stack(cols, axis=2)

    @staticmethod
    def azimuth(W):
        """ldm/dataset.py:266-267, evaluated in fp32 like the torch expression."""
        a = (F(W) - F(0.5) - np.arange(W).astype(F)) / F(W)
        return (a * F(2.) * F(np.pi) - F(np.pi)).astype(F)

    # ---- f1: points -> BEV volume (ldm/dataset.py:280-294 + 13-132) -------------------------------------------
    def to_voxel(self, img, pc=None):
        """pc: optionally the (B, N, 4) cloud to splat (lets the splat be pinned separately from to_pc's cos / sin)."""
        pc = self.to_pc(img) if pc is None else np.asarray(pc, F)
        B = pc.shape[0]
        D, GH, GW = self.grid_sizes
        lo, hi = self.pc_range[:3], self.pc_range[3:]
        p = (pc[:, :, :3] - (hi + lo) / F(2)) / ((hi - lo) / F(2))
        feat = pc[:, :, 3].astype(np.float64)
        g_xyz = np.array([GW, GH, D], F)
        pi = ((p + F(1)) * F(0.5)) * (g_xyz - F(1))
        base = np.floor(pi)
        rem = (pi - base).astype(F)
        base = base.astype(np.int64)
        nvox = D * GH * GW
        dens = np.zeros((B, nvox), np.float64)
        vol = np.zeros((B, nvox), np.float64)
        for dx in (0, 1):
            X = base[..., 0] + dx
            wx = (1 - dx) + (2 * dx - 1) * rem[..., 0]
            for dy in (0, 1):
                Y = base[..., 1] + dy
                wy = (1 - dy) + (2 * dy - 1) * rem[..., 1]
                for dz in (0, 1):
                    Z = base[..., 2] + dz
                    wz = (1 - dz) + (2 * dz - 1) * rem[..., 2]
                    w = (wx * wy * wz).astype(F)
                    ok = (X >= 0) & (X < GW) & (Y >= 0) & (Y < GH) & (Z >= 0) & (Z < D)
                    idx = (Z * GH + Y) * GW + X
                    for b in range(B):
                        m = ok[b]
                        np.add.at(dens[b], idx[b][m], w[b][m].astype(np.float64))
                        np.add.at(vol[b], idx[b][m], w[b][m].astype(np.float64) * feat[b][m])
        vol = vol / np.maximum(dens, 1e-4)
        if self.normalize_volume_densities:
            dens = np.log(dens + 1)
        return np.concatenate([dens.reshape(B, D, GH, GW), vol.reshape(B, D, GH, GW)], axis=1).astype(F)

    # ---- the per-image tail of ldm/inference.py:171-183 -------------------------------------------------------
    @staticmethod
    def filter_points(pc, max_depth=90.0):
        """`pc[np.linalg.norm(pc[:, :3], 2, axis=1) < 90.0]` (ldm/inference.py:177-179): what lands in <idx>.bin."""
        depth = np.linalg.norm(pc[:, :3], 2, axis=1)
        return pc[depth < max_depth]

    @staticmethod
    def render_u8(chw, channel=0):
        """`(x.permute(2, 1, 0).clip(0, 1) * 255).astype(uint8)[:, :, c]` (ldm/inference.py:180-183): (H, W) bytes."""
        a = np.transpose(np.asarray(chw, F), (2, 1, 0))
        return (np.clip(a, 0, 1) * F(255.)).astype(np.uint8)[:, :, channel]

    # ---- f3: point cloud -> range image (ldm/dataset.py:159-226) ----------------------------------------------
    def row_inds_nearest_beam(self, pc):
        """ldm/kitti360_range_image.py:51-61: beam whose inclination is closest to the point's elevation."""
        xy = np.linalg.norm(pc[:, :2], ord=2, axis=1)
        err = np.stack([np.abs(self.incl[i] - np.arctan2(self.height[i] - pc[:, 2], xy)) for i in range(self.H)], -1)
        return np.argmin(err, axis=-1)

    def project(self, pc, row_inds):
        """ldm/dataset.py:159-187 (`__call__`).  pc: (N, >=4) float32; returns (H, width, 2) with -1 where empty."""
        pc = np.array(pc, F, copy=True)
        W = self.width
        azi = np.arctan2(pc[:, 1], pc[:, 0])
        col = W - 1.0 + 0.5 - (azi + np.pi) / (2.0 * np.pi) * W
        col = np.round(col).astype(np.int32)
        col[col == W] = W - 1
        col[col < 0] = 0
        out = np.full((self.H, W, 2), -1, dtype=F)
        pc[:, 2] -= self.height[row_inds]
        rng = np.linalg.norm(pc[:, :3], axis=1, ord=2)
        rng[rng > self.range_fill_value[0]] = self.range_fill_value[0]
        order = np.argsort(-rng, kind="stable")
        if self.log:
            val = np.log2(rng[order] + 1) / 6
        elif self.inverse:
            val = 1 / rng[order]
        else:
            val = rng[order]
        out[row_inds[order], col[order], :] = np.concatenate([val[:, None], pc[order, 3:4]], axis=1)
        return out

    def process_miss_value(self, ri):
        """ldm/dataset.py:195-221."""
        ri = np.array(ri, F, copy=True)
        H, W, _ = ri.shape
        mask = ri[..., 0] > 0
        miss = ri[:, :, 0] == -1
        sh = np.roll(ri, -1, axis=1)
        ri[miss, :] = sh[miss, :]
        msh = np.roll(mask, -1, axis=1)
        mask = np.where(miss, msh, mask)
        still = ri[:, :, 0] == -1
        r0 = ri[:, :, 0]
        down2, top2 = np.roll(r0, 2, axis=0), np.roll(r0, -2, axis=0)
        right2, left2 = np.roll(r0, 2, axis=1), np.roll(r0, -2, axis=1)
        car = still & ((down2 != -1) | (top2 != -1) | (right2 != -1) | (left2 != -1))
        if self.log:
            ri[still, :] = (np.log2(self.range_fill_value + 1) / 6).astype(F)
        elif self.inverse:
            ri[still, :] = np.array([1 / self.range_fill_value[0], self.range_fill_value[1]], F)
        else:
            ri[still, :] = self.range_fill_value.astype(F)
        return ri, mask, car

    def normalize(self, ri):
        ri = np.array(ri, F, copy=True)
        if not self.log and not self.inverse:
            ri[..., 0] = (ri[..., 0] - self.mean) / self.std
        return ri
