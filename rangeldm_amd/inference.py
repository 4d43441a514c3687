"""Unconditional sampling driver -- the MI355X counterpart of the reference's `ldm/inference.py:154-183`
(`accelerate launch ldm/inference.py --cfg configs/RangeLDM.yaml`): one process per GPU, every rank samples
`eval_batch_size` images per outer iteration, image index = (rank + nproc * i) * B + j, truncation at `--samples`,
one file per image.

    python -m rangeldm_amd.inference --cfg RangeLDM --samples 64 --out outputs/RangeLDM/generated
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m rangeldm_amd.inference --cfg RangeLDM ...

`--cfg` is a preset name (rangeldm_amd.config.PRESETS) or a reference-style yaml whose `model_config` holds the
UNet2DModel kwargs (ldm/configs/*.yaml).  Weights: a directory laid out like the reference's output_dir
(`unet/diffusion_pytorch_model.safetensors`, `vae/diffusion_pytorch_model.safetensors`, diffusers keys) or, when
none is given, the deterministic synthetic state dict (no checkpoints exist offline).  Output per image, as
ldm/inference.py:171-183 writes it: `<idx>.bin` (float32 [N, 4] x, y, z, remission of the returns closer than 90 m),
`<idx>.png` (8-bit BEV density) and `<idx>_range.png` (8-bit range channel) -- the point cloud, the BEV volume, the depth
filter and the 8-bit rendering all run on the GPU (rangeldm_amd.range_image); only finished bytes cross PCIe.
`--save-npy` adds `<idx>.npy`, the raw (2, W, H) fp32 range image.
"""
import argparse
import os

import numpy as np
import torch

from . import distributed as D
from .config import PRESETS, UNetConfig, VAEConfig


def plan_iterations(samples, batch_size, world):
    """Outer-loop length of ldm/inference.py:159 (`samples // B // nproc + 1`: up to one wasted batch per rank)."""
    return samples // batch_size // world + 1


def image_indices(iteration, batch_size, rank, world, samples):
    """(j, global index) pairs rank `rank` keeps in iteration `iteration` -- ldm/inference.py:174-176."""
    out = []
    for j in range(batch_size):
        idx = (rank + world * iteration) * batch_size + j
        if idx >= samples:
            break
        out.append((j, idx))
    return out


def load_config(cfg):
    if cfg in PRESETS:
        return dict(PRESETS[cfg])
    import yaml
    with open(cfg) as f:
        y = yaml.safe_load(f)
    mc = dict(y["model_config"])
    mc["sample_size"] = tuple(mc["sample_size"])
    # ldm/inference.py:99-118: `all_circonv` swaps every conv / downsampler for the circular ones; `sub_circonv` only part of
    # the net; neither: a plain zero-padded UNet.  Only the first is built -- refuse the others loudly.
    if y.get("sub_circonv", False) and not y.get("all_circonv", False):
        raise NotImplementedError("sub_circonv (ldm/inference.py:105-118) is not supported; use all_circonv")
    if not y.get("all_circonv", False):
        raise NotImplementedError("configs without all_circonv (zero-padded convolutions) are not supported")
    unet = UNetConfig(**{k: v for k, v in mc.items() if k in UNetConfig.__dataclass_fields__})
    vae = None
    if y.get("with_vae", False):
        vae = vae_config_for(y, unet, os.path.dirname(os.path.abspath(cfg)))
    return dict(unet=unet, vae=vae, pos_encoding=bool(y.get("pos_encoding", False)), cond_channels=0,
                steps=int(y.get("ddpm_num_inference_steps", 50)), ddim=bool(y.get("ddim", False)),
                batch=int(y.get("eval_batch_size", 16)))


def vae_config_for(y, unet, base_dir):
    """VAE geometry of a reference yaml: `vae_config` names the sgm yaml the VAE was trained with
    (ldm/configs/RangeLDM.yaml: ../vae/configs/kitti360.yaml; ldm/train_unconditional.py:250-271 reads it).  When that file
    is reachable its ddconfig is used; otherwise the shipped default (ch 64, ch_mult [1, 2, 4], 4x) -- a `--weights`
    directory always overrides both with the checkpoint's own config.json."""
    from .checkpoint import vae_config_from_sgm_yaml
    path = y.get("vae_config")
    if path:
        path = path if os.path.isabs(path) else os.path.normpath(os.path.join(base_dir, path))
        if os.path.isfile(path):
            import yaml
            with open(path) as f:
                v = vae_config_from_sgm_yaml(yaml.safe_load(f))
            f_ = v.downscale
            v.sample_size = (unet.sample_size[0] * f_, unet.sample_size[1] * f_)
            return v
    f_ = VAEConfig().downscale                      # vae/configs/kitti360.yaml:35-41: ch_mult [1,2,4] -> 4x
    return VAEConfig(sample_size=(unet.sample_size[0] * f_, unet.sample_size[1] * f_))


def device_step_noise(seed, global_indices, steps, lat_shape, device):
    """[steps][B, C, W, H] ancestral noise, drawn ON THE DEVICE from one generator per GLOBAL sample index (seed and index
    only: independent of rank, world size and batch composition).  The reference draws it from the unseeded global device RNG
    (ldm/pipelines.py:362 passes no generator), which is not reproducible across runs or GPU counts."""
    out = torch.empty((steps, len(global_indices), *lat_shape), device=device, dtype=torch.float32)
    for b, gidx in enumerate(global_indices):
        g = torch.Generator(device=device)
        g.manual_seed((int(seed) * 1000003 + int(gidx)) & 0x7fffffffffffffff)
        out[:, b] = torch.randn((steps, *lat_shape), generator=g, device=device, dtype=torch.float32)
    return out


def save_png(pixels_u8, path):
    """(rows, cols) uint8 -> 8-bit grayscale PNG (ldm/inference.py:180-183 uses PIL; without it a minimal encoder)."""
    try:
        from PIL import Image
        Image.fromarray(pixels_u8, mode="L").save(path)
        return
    except ImportError:
        pass
    import struct
    import zlib

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)

    h, w = pixels_u8.shape
    raw = b"".join(b"\x00" + pixels_u8[r].tobytes() for r in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def sensor_for(beams, **kw):
    """ldm/inference.py:60-70 picks the projection class from the config; here: by the number of beams."""
    from .range_image import point_cloud_to_range_image_KITTI, point_cloud_to_range_image_nuScenes
    if beams == 64:
        return point_cloud_to_range_image_KITTI(**kw)
    if beams == 32:
        return point_cloud_to_range_image_nuScenes(**kw)
    raise ValueError(f"no sensor table for {beams} beams (64: KITTI-360, 32: nuScenes)")


def postprocess(to_range, image, max_depth=90.0):
    """The per-batch tail of ldm/inference.py:171-183 on the device.  image: (B, 2, W, H) fp32 cuda.
    Returns host arrays: points (B, N, 4), counts (B,), bev_u8 (B, gx, gy), range_u8 (B, H, W)."""
    from .range_image import render_u8
    pc_all = to_range.to_pc_torch(image)
    bev_out_all = to_range.to_voxel(image)
    kept, counts = to_range.filter_points(pc_all, max_depth)
    bev_u8 = render_u8(bev_out_all)
    range_u8 = render_u8(image)
    return kept.cpu().numpy(), counts.cpu().numpy(), bev_u8.cpu().numpy(), range_u8.cpu().numpy()


def main(argv=None):
    ap = argparse.ArgumentParser(description="RangeLDM sampler on MI355X (ldm/inference.py counterpart)")
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--weights", default=None, help="reference-style output_dir with unet/ and vae/ safetensors")
    ap.add_argument("--seed", type=int, default=20240310)
    ap.add_argument("--ema", action="store_true", help="read <weights>/unet_ema instead of <weights>/unet")
    ap.add_argument("--save-npy", action="store_true", help="also write the raw (2, W, H) fp32 range image")
    a = ap.parse_args(argv)

    from .params import unet_param_shapes, vae_param_shapes
    from .pipelines import DDIMPipelineRange, DDPMPipelineRange, LDMPipelineRange
    from .schedulers import DDIMSchedulerHIP, DDPMSchedulerHIP
    from .synth import latent_noise, synth_state_dict
    from .unet import UNet2DModelHIP
    from .vae import AutoencoderKLHIP

    cfg = load_config(a.cfg)
    B = a.batch_size or cfg.get("batch", 16)
    steps = cfg.get("steps", 50)
    rank, world, local = D.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    out_dir = a.out or os.path.join("outputs", os.path.splitext(os.path.basename(a.cfg))[0], "generated")
    os.makedirs(out_dir, exist_ok=True)

    sched_cfg = None
    if a.weights:
        # ldm/inference.py:46-52,84-127: configs and weights come from the training run's output_dir
        from .checkpoint import load_output_dir
        ck = load_output_dir(a.weights, with_vae=cfg["vae"] is not None, ema=a.ema)
        cfg["unet"], usd, sched_cfg = ck["unet_config"], ck["unet"], ck["scheduler_config"]
        if cfg["vae"] is not None:
            cfg["vae"], vsd = ck["vae_config"], ck["vae"]
    else:
        usd = synth_state_dict(unet_param_shapes(cfg["unet"]), seed=a.seed, prefix="")
        if cfg["vae"] is not None:
            vsd = synth_state_dict(vae_param_shapes(cfg["vae"]), seed=a.seed, prefix="vae.")
    unet = UNet2DModelHIP(cfg["unet"])
    unet.load_state_dict(usd)
    if cfg["vae"] is not None:
        vae = AutoencoderKLHIP(cfg["vae"])
        vae.load_state_dict(vsd)
        # ldm/inference.py:131-136: the LDM branch keeps the DDPM scheduler (strided ancestral sampling)
        pipe = LDMPipelineRange(vae=vae, unet=unet, scheduler=DDPMSchedulerHIP(sched_cfg),
                                pos_encoding=cfg["pos_encoding"])
    elif cfg.get("ddim", True):
        pipe = DDIMPipelineRange(unet=unet, scheduler=DDIMSchedulerHIP(sched_cfg), pos_encoding=cfg["pos_encoding"])
    else:
        # ldm/inference.py:141-145: ddim False -> DDPMPipelineRange (ancestral sampling in pixel space; it takes no
        # pos_encoding argument in the reference either: ldm/pipelines.py:27-31)
        pipe = DDPMPipelineRange(unet=unet, scheduler=DDPMSchedulerHIP(sched_cfg))
    ddpm = isinstance(pipe.scheduler, DDPMSchedulerHIP)
    lat_ch = cfg["unet"].in_channels if isinstance(pipe, DDPMPipelineRange) else cfg["unet"].out_channels
    lat_shape = (lat_ch, *cfg["unet"].sample_size)
    to_range = None

    for i in range(plan_iterations(a.samples, B, world)):
        keep = image_indices(i, B, rank, world, a.samples)
        if not keep:
            continue
        # x_T AND the ancestral step noise are functions of the GLOBAL image index: any GPU count produces the same images
        idx = D.global_sample_indices(i, B, rank, world)
        x_T = torch.from_numpy(np.stack([latent_noise(a.seed, j, lat_shape) for j in idx])).to(dev)
        kw = {}
        if ddpm:
            kw["step_noise"] = device_step_noise(a.seed, idx, steps, lat_shape, dev)
        image = pipe(batch_size=B, num_inference_steps=steps, output_type="torch", latents=x_T, **kw)
        if to_range is None:
            to_range = sensor_for(image.shape[3])
        points, counts, bev_u8, range_u8 = postprocess(to_range, image)
        host = image.float().cpu().numpy() if a.save_npy else None
        for j, gidx in keep:
            points[j, :counts[j]].tofile(os.path.join(out_dir, f"{gidx}.bin"))          # ldm/inference.py:177-179
            save_png(bev_u8[j], os.path.join(out_dir, f"{gidx}.png"))                   # :180-181
            save_png(range_u8[j], os.path.join(out_dir, f"{gidx}_range.png"))           # :182-183
            if host is not None:
                np.save(os.path.join(out_dir, f"{gidx}.npy"), host[j])
    D.barrier()
    if rank == 0:
        print(f"wrote {a.samples} range images to {out_dir}")


if __name__ == "__main__":
    main()
