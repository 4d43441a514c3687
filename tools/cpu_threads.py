import time, torch, sys, os
sys.path.insert(0, os.getcwd())
from rangeldm_amd.config import PRESETS
from rangeldm_amd.params import unet_param_shapes
from rangeldm_amd.synth import synth_state_dict
from oracle.unet import OracleUNet
p = PRESETS["RangeLDM"]
sd = synth_state_dict(unet_param_shapes(p["unet"]))
ou = OracleUNet(p["unet"], sd)
x = torch.randn(16, 5, 256, 16)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    ou(x[:1], 3)
    t0 = time.perf_counter(); ou(x, 3); print(th, "threads: B=16 unet fwd", round(time.perf_counter() - t0, 2), "s", flush=True)
