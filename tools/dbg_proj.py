import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from rangeldm_amd import range_image as RI
from oracle.lidar import LidarOracle
g = dict(np.load('/root/repo/tests/golden/lidar.npz'))
t = RI.point_cloud_to_range_image_KITTI(width=128)
sweep = g['lidar_proj_kitti_sweep']
out = t.project(torch.from_numpy(sweep).cuda())
jpg = out['jpg'].cpu().numpy(); ref = g['lidar_proj_kitti_jpg_ref']
d = jpg[0] != ref[0]
print('mismatch', d.sum(), 'of', d.size)
a = jpg[0][d][:10]; b = ref[0][d][:10]
print(a, b, (a.view(np.int32) - b.view(np.int32)))
o = LidarOracle(t.incl, t.height, width=128)
rows = o.row_inds_nearest_beam(sweep)
# device-side check of sqrt/div via torch
x = torch.from_numpy(sweep).cuda()
zz = x[:,2] - torch.from_numpy(t.height[rows]).cuda()
r_t = torch.sqrt(x[:,0]*x[:,0] + x[:,1]*x[:,1] + zz*zz).cpu().numpy()
pc = sweep.copy(); pc[:,2] -= t.height[rows]
r_n = np.linalg.norm(pc[:,:3], axis=1, ord=2)
print('torch-gpu vs numpy range mismatches', (r_t != r_n).sum())
s = (pc[:,:3]*pc[:,:3])
r_m = np.sqrt((s[:,0]+s[:,1])+s[:,2]); print('manual order vs norm', (r_m != r_n).sum())
r_m2 = np.sqrt(s[:,0]+(s[:,1]+s[:,2])); print('alt order vs norm', (r_m2 != r_n).sum())
