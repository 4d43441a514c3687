"""Range image <-> point cloud on the GPU: the host-side mirror of the reference's `point_cloud_to_range_image`
(ldm/dataset.py:135-294) and its sensor subclasses (ldm/kitti360_range_image.py:14-61,
ldm/nuscenes_range_image.py:16-46).  Same class / method / argument names; every method hands device tensors to
librangeldm_hip.so (rangeldm_amd/csrc/lidar.hip) and returns freshly allocated device tensors.  No CPU fallback.

Differences a caller can observe (both deliberate):
  * `to_pc_torch` does not need a mutable input (the reference rewrites `r_true` in a temporary, too);
  * `__call__` does not shift the caller's `pc[:, 2]` in place (ldm/dataset.py:168 does).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


class point_cloud_to_range_image:
    def __init__(self, width=1024, grid_sizes=[1, 1024, 1024], pc_range=[-25.6, -25.6, -3., 25.6, 25.6, 1.], log=False,
                 normalize_volume_densities=True, inverse=False) -> None:
        self.range_fill_value = np.array([100, 0])
        self.width = width
        self.grid_sizes = list(grid_sizes)
        self.pc_range = list(pc_range)
        self.log = log
        self.normalize_volume_densities = normalize_volume_densities
        self.inverse = inverse
        self.mean = 20.
        self.std = 40.
        self.min_depth = 0.0            # nuScenes drops returns closer than 2 m before projecting
        if not hasattr(self, "incl"):
            self.height = self.zenith = self.incl = None
            self.H = None
        self._h = None

    # ---- handle ---------------------------------------------------------------------------------------------
    def _handle(self):
        if self._h is None:
            if self.incl is None:
                raise NotImplementedError("sensor tables (incl / height) are defined by the subclasses")
            _lib.require_gpu()
            cfg = _lib.LidarConfigC()
            cfg.beams, cfg.width = int(self.H), int(self.width)
            cfg.mode = 1 if self.log else (2 if self.inverse else 0)
            cfg.mean, cfg.std = float(self.mean), float(self.std)
            cfg.range_fill, cfg.intensity_fill = float(self.range_fill_value[0]), float(self.range_fill_value[1])
            for i in range(3):
                cfg.grid[i] = int(self.grid_sizes[i])
            for i in range(6):
                cfg.pc_range[i] = float(self.pc_range[i])
            cfg.normalize_volume_densities = int(bool(self.normalize_volume_densities))
            incl = np.ascontiguousarray(self.incl, np.float32)
            height = np.ascontiguousarray(self.height, np.float32)
            h = C.c_void_p()
            _lib.check(_lib.lib().rldm_lidar_create(C.byref(cfg), incl.ctypes.data_as(C.POINTER(C.c_float)),
                                                    height.ctypes.data_as(C.POINTER(C.c_float)), C.byref(h)),
                       "rldm_lidar_create")
            self._h = h
        return self._h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None:
            try:
                _lib.lib().rldm_lidar_destroy(h)
            except Exception:
                pass

    def _images(self, range_images):
        if range_images.dim() != 4 or range_images.shape[3] != self.H:
            raise ValueError(f"range_images must be (B, C, W, {self.H}), got {tuple(range_images.shape)}")
        if not range_images.is_cuda:
            raise RuntimeError("range_images must live on the GPU (rangeldm_amd has no CPU path)")
        return range_images.detach().float().contiguous()

    # ---- f1: after the sampler ------------------------------------------------------------------------------
    def to_pc_torch(self, range_images):
        """range_images: B x C x W x H -> point_cloud: B x N x (4 if C > 1 else 3)   (ldm/dataset.py:228-278)"""
        x = self._images(range_images)
        B, Cc, W, H = x.shape
        out = torch.empty((B, W * H, 4 if Cc > 1 else 3), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().rldm_lidar_to_points(self._handle(), x.data_ptr(), B, Cc, W, out.data_ptr(),
                                                   _lib.stream_ptr(x.device)), "rldm_lidar_to_points")
        return out

    def to_voxel(self, range_images):
        """-> (B, 2 * D, H, W) BEV volume: vote density and density-normalised remission (ldm/dataset.py:280-294)"""
        x = self._images(range_images)
        B, Cc, W, H = x.shape
        D, GH, GW = self.grid_sizes
        out = torch.empty((B, 2 * D, GH, GW), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().rldm_lidar_to_voxel(self._handle(), x.data_ptr(), B, Cc, W, out.data_ptr(),
                                                  _lib.stream_ptr(x.device)), "rldm_lidar_to_voxel")
        return out

    def filter_points(self, point_cloud, max_depth=90.0):
        """Per image `pc[np.linalg.norm(pc[:, :3], 2, axis=1) < max_depth]`, order kept (ldm/inference.py:177-179).
        Returns (points (B, N, cols) with the kept rows packed to the front, counts (B,) int32), both on device."""
        pc = point_cloud.detach().float().contiguous()
        B, N, cols = pc.shape
        out = torch.empty_like(pc)
        counts = torch.empty((B,), dtype=torch.int32, device=pc.device)
        _lib.check(_lib.lib().rldm_lidar_filter_points(self._handle(), pc.data_ptr(), B, N, cols, float(max_depth),
                                                       out.data_ptr(), counts.data_ptr(), _lib.stream_ptr(pc.device)),
                   "rldm_lidar_filter_points")
        return out, counts

    # ---- f3: in front of the training path -------------------------------------------------------------------
    def get_row_inds(self, pc):
        raise NotImplementedError

    def project(self, pc):
        """`__call__` + `process_miss_value` + `normalize` + the permutes of RangeDataset.__getitem__
        (ldm/dataset.py:159-226, 320-333) in one pass: pc (N, >= 4) device fp32 ->
        dict(jpg=(2, W, H) fp32, mask=(W, H) bool, car_window_mask=(W, H) bool) on the device."""
        if not pc.is_cuda:
            raise RuntimeError("pc must live on the GPU (rangeldm_amd has no CPU path)")
        pc = pc.detach().float().contiguous()
        n, stride = pc.shape
        rows = self.get_row_inds(pc)
        if rows is not None:
            rows = rows.to(torch.int32).contiguous()
        W, H = self.width, self.H
        img = torch.empty((2, W, H), dtype=torch.float32, device=pc.device)
        mask = torch.empty((W, H), dtype=torch.uint8, device=pc.device)
        car = torch.empty((W, H), dtype=torch.uint8, device=pc.device)
        _lib.check(_lib.lib().rldm_lidar_project(self._handle(), pc.data_ptr(), n, stride,
                                                 rows.data_ptr() if rows is not None else None, float(self.min_depth),
                                                 img.data_ptr(), mask.data_ptr(), car.data_ptr(),
                                                 _lib.stream_ptr(pc.device)), "rldm_lidar_project")
        return {"jpg": img, "mask": mask.bool(), "car_window_mask": car.bool()}


def render_u8(images, channel=0):
    """`(images[j].permute(2, 1, 0).clip(0, 1) * 255).astype(uint8)[:, :, channel]` for every j
    (ldm/inference.py:180-183): (B, C, W, H) device fp32 -> (B, H, W) device uint8, the pixels of the 8-bit PNGs."""
    if not images.is_cuda:
        raise RuntimeError("images must live on the GPU (rangeldm_amd has no CPU path)")
    x = images.detach().float().contiguous()
    B, Cc, W, H = x.shape
    out = torch.empty((B, H, W), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().rldm_render_u8(x.data_ptr(), B, Cc, W, H, int(channel), out.data_ptr(),
                                         _lib.stream_ptr(x.device)), "rldm_render_u8")
    return out


# Per-beam calibration of the two sensors (data, ldm/kitti360_range_image.py:19-48 / ldm/nuscenes_range_image.py:20-35):
# mounting height offset [m] and zenith angle [rad] of every laser, top beam first.
_KITTI_HEIGHT = (
    0.20966667, 0.2092, 0.2078, 0.2078, 0.2078, 0.20733333, 0.20593333, 0.20546667, 0.20593333, 0.20546667, 0.20453333,
    0.205, 0.2036, 0.20406667, 0.2036, 0.20313333, 0.20266667, 0.20266667, 0.20173333, 0.2008, 0.2008, 0.2008, 0.20033333,
    0.1994, 0.20033333, 0.19986667, 0.1994, 0.1994, 0.19893333, 0.19846667, 0.19846667, 0.19846667, 0.12566667, 0.1252,
    0.1252, 0.12473333, 0.12473333, 0.1238, 0.12333333, 0.1238, 0.12286667, 0.1224, 0.12286667, 0.12146667, 0.12146667,
    0.121, 0.12053333, 0.12053333, 0.12053333, 0.12006667, 0.12006667, 0.1196, 0.11913333, 0.11866667, 0.1182, 0.1182,
    0.1182, 0.11773333, 0.11726667, 0.11726667, 0.1168, 0.11633333, 0.11633333, 0.1154)
_KITTI_ZENITH = (
    0.03373091, 0.02740409, 0.02276443, 0.01517224, 0.01004049, 0.00308099, -0.00155868, -0.00788549, -0.01407172,
    -0.02103122, -0.02609267, -0.032068, -0.03853542, -0.04451074, -0.05020488, -0.0565317, -0.06180405, -0.06876355,
    -0.07361411, -0.08008152, -0.08577566, -0.09168069, -0.09793721, -0.10398284, -0.11052055, -0.11656618, -0.12219002,
    -0.12725147, -0.13407038, -0.14067839, -0.14510716, -0.15213696, -0.1575499, -0.16711043, -0.17568678, -0.18278688,
    -0.19129293, -0.20247031, -0.21146846, -0.21934183, -0.22763699, -0.23536977, -0.24528179, -0.25477201, -0.26510582,
    -0.27326038, -0.28232882, -0.28893683, -0.30004392, -0.30953414, -0.31993824, -0.32816311, -0.33723155, -0.34447224,
    -0.352908, -0.36282001, -0.37216965, -0.38292524, -0.39164219, -0.39895318, -0.40703745, -0.41835542, -0.42777535,
    -0.43621111)
_NUSC_HEIGHT = (
    -0.00216031, -0.00098729, -0.00020528, 0.00174976, 0.0044868, -0.00294233, -0.00059629, -0.00020528, 0.00174976,
    -0.00294233, -0.0013783, 0.00018573, 0.00253177, -0.00098729, 0.00018573, 0.00096774, -0.00411535, -0.0013783,
    0.00018573, 0.00018573, -0.00294233, -0.0013783, -0.00098729, -0.00020528, 0.00018573, 0.00018573, 0.00018573,
    -0.00020528, 0.00018573, 0.00018573, 0.00018573, 0.00018573)
_NUSC_ZENITH = (
    1.86705767e-01, 1.63245357e-01, 1.39784946e-01, 1.16324536e-01, 9.28641251e-02, 7.01857283e-02, 4.67253177e-02,
    2.32649071e-02, -1.95503421e-04, -2.28739003e-02, -4.63343109e-02, -6.97947214e-02, -9.32551320e-02, -1.15933529e-01,
    -1.39393939e-01, -1.62854350e-01, -1.85532747e-01, -2.08993157e-01, -2.32453568e-01, -2.55913978e-01, -2.78592375e-01,
    -3.02052786e-01, -3.25513196e-01, -3.48973607e-01, -3.72434018e-01, -3.95894428e-01, -4.19354839e-01, -4.42033236e-01,
    -4.65493646e-01, -4.88954057e-01, -5.12414467e-01, -5.35874878e-01)


class point_cloud_to_range_image_KITTI(point_cloud_to_range_image):
    """64-beam KITTI-360 sensor; beam of a return = closest inclination (ldm/kitti360_range_image.py:51-61), found
    inside the projection kernel."""

    def __init__(self, **kwargs) -> None:
        self.height = np.array(_KITTI_HEIGHT, dtype=np.float32)
        self.zenith = np.array(_KITTI_ZENITH, dtype=np.float32)
        self.incl = -self.zenith
        self.H = 64
        super().__init__(**kwargs)

    def get_row_inds(self, pc):
        return None                     # nearest-inclination search runs on the device


class point_cloud_to_range_image_nuScenes(point_cloud_to_range_image):
    """32-beam nuScenes sensor; the sweep carries the ring index in column 4 (ldm/nuscenes_range_image.py:43-45) and
    returns closer than 2 m are dropped (:37-41)."""

    def __init__(self, **kwargs) -> None:
        self.height = np.array(_NUSC_HEIGHT, dtype=np.float32)
        self.zenith = np.array(_NUSC_ZENITH, dtype=np.float32)
        self.incl = -self.zenith
        self.H = 32
        super().__init__(**kwargs)
        self.min_depth = 2.0

    def get_row_inds(self, pc):
        return 31 - pc[:, 4].to(torch.int32)
