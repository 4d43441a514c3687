"""Conditional sampling driver -- the MI355X counterpart of the reference's `ldm/inference_conditional.py` (BASELINE config 4:
`--cfg configs/upsample.yaml`, 16 -> 64 beams; `configs/inpainting.yaml`): LDMUpscalePipelineRange over a fixed batch of
range images, one pass per seed, `densification_{result,target,input}/` or `inpainting_{...}/` next to each other.

    python -m rangeldm_amd.inference_conditional --cfg upsample --samples 32 --out outputs/upsample/generated
    python -m rangeldm_amd.inference_conditional --cfg /path/to/inpainting.yaml --input /data/range_images --weights outputs/inpainting

What is the reference's and what is not:
  * the loop (`ldm/inference_conditional.py:158-210`): ONE batch is taken from the test loader before the loop and re-sampled
    with seed = rank + nproc * i in iteration i; outputs `<j>_seed_<seed>.bin / .png` in the result directory, and for seed 0
    the ground truth and the sparse / masked input in the target / input directories; points farther than 70 m (KITTI-360;
    90 m nuScenes) are dropped; the BEV PNG is the 8-bit density plane.
  * the batch: `--input DIR` reads `*.npy` range images of shape (2, W, H) in the reference's normalisation
    (`point_cloud_to_range_image.normalize`); without it a deterministic synthetic batch stands in (no dataset is reachable
    offline).  `down` / `masked_image` / `inpainting_mask` are derived as `ldm/dataset.py:340-362` does (rangeldm_amd.conditional).
  * the networks: `--weights` = the training run's output_dir (`unet/ vae/ scheduler/`), else synthetic weights.
"""
import argparse
import glob
import os

import numpy as np
import torch

from . import distributed as D
from .conditional import downsample_range_image, inpainting_inputs, sparse_input_image
from .config import PRESETS, UNetConfig, VAEConfig
from .inference import postprocess, save_png, sensor_for, vae_config_for


def load_conditional_config(cfg):
    """Preset name ("upsample", "inpainting") or a reference yaml.  The yaml's `upsample` (rate) / `inpainting` (masked
    fraction) select the task; `model_config: null` means the default conditional UNet of ldm/train_conditional.py:232-251
    (latent channels + 8 for up-sampling, + 5 for in-painting)."""
    if cfg in ("upsample", "inpainting"):
        p = dict(PRESETS[cfg])
        p.update(task=cfg, rate=4, fraction=0.0625, steps=50, batch=16, range_limit=70.0)
        return p
    import yaml
    with open(cfg) as f:
        y = yaml.safe_load(f)
    if not y.get("all_circonv", False):
        raise NotImplementedError("only all_circonv configs are supported (ldm/inference_conditional.py:99-114)")
    if not y.get("with_vae", True):
        raise NotImplementedError("the conditional pipelines are latent-space (with_vae)")
    up, inp = y.get("upsample"), y.get("inpainting")
    if bool(up) == bool(inp):
        raise ValueError("exactly one of `upsample` (rate) and `inpainting` (masked fraction) must be set")
    res = tuple(y.get("resolution", (1024, 64)))
    mc = y.get("model_config")
    if mc:
        mc = dict(mc)
        mc["sample_size"] = tuple(mc["sample_size"])
        unet = UNetConfig(**{k: v for k, v in mc.items() if k in UNetConfig.__dataclass_fields__})
    else:
        kw = dict(sample_size=(res[0] // 4, res[1] // 4), in_channels=4 + (8 if up else 5), out_channels=4)
        if y.get("block_out_channels"):
            kw["block_out_channels"] = tuple(y["block_out_channels"])
        unet = UNetConfig(**kw)
    vae = vae_config_for(y, unet, os.path.dirname(os.path.abspath(cfg)))
    return dict(unet=unet, vae=vae, pos_encoding=False, cond_channels=unet.in_channels - unet.out_channels,
                task="upsample" if up else "inpainting", rate=int(up) if up else 0, fraction=float(inp) if inp else 0.0,
                steps=int(y.get("ddpm_num_inference_steps", 50)), batch=int(y.get("eval_batch_size", 16)),
                range_limit=90.0 if y.get("nuscenes") else 70.0)


def load_batch(input_dir, B, shape, seed):
    """(B, 2, W, H) range images: the first B `*.npy` files of `input_dir` (the reference takes one batch of its test loader), or
    a synthetic batch: channel 0 a smooth normalised range profile in [-0.5, 2], channel 1 a remission in [0, 1]."""
    if input_dir:
        files = sorted(glob.glob(os.path.join(input_dir, "*.npy")))[:B]
        if len(files) < B:
            raise ValueError(f"{input_dir}: {len(files)} range images, batch size {B}")
        x = np.stack([np.load(f).astype(np.float32) for f in files])
        if tuple(x.shape[1:]) != tuple(shape):
            raise ValueError(f"range images of shape {x.shape[1:]}, expected {shape}")
        return torch.from_numpy(x)
    from .synth import uniform
    C, W, H = shape
    w = torch.arange(W, dtype=torch.float32)[None, None, :, None] / W
    h = torch.arange(H, dtype=torch.float32)[None, None, None, :] / H
    ph = torch.from_numpy(uniform(seed, "cond/phase", (B, 1, 1, 1)))
    rng = 0.6 + 0.8 * torch.sin(2 * np.pi * (w + ph)) * (0.5 + 0.5 * h) + 0.3 * torch.cos(6 * np.pi * w)
    rem = 0.5 + 0.5 * torch.from_numpy(uniform(seed, "cond/remission", (B, 1, W, H)))
    return torch.cat([rng.expand(B, 1, W, H).clamp(-0.5, 2.0), rem], 1).contiguous()


def main(argv=None):
    ap = argparse.ArgumentParser(description="RangeLDM conditional sampler on MI355X (ldm/inference_conditional.py counterpart)")
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--input", default=None, help="directory of (2, W, H) .npy range images (default: synthetic batch)")
    ap.add_argument("--weights", default=None, help="reference-style output_dir with unet/ and vae/ safetensors")
    ap.add_argument("--seed", type=int, default=20240310)
    ap.add_argument("--ema", action="store_true")
    a = ap.parse_args(argv)

    from .encoders import SparseRangeImageEncoder2
    from .params import unet_param_shapes, vae_param_shapes
    from .pipelines import LDMUpscalePipelineRange
    from .schedulers import DDPMSchedulerHIP
    from .synth import synth_state_dict
    from .unet import UNet2DModelHIP
    from .vae import AutoencoderKLHIP

    cfg = load_conditional_config(a.cfg)
    B = a.batch_size or cfg["batch"]
    steps = cfg["steps"]
    rank, world, local = D.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    name = os.path.splitext(os.path.basename(a.cfg))[0]
    out = a.out or os.path.join("outputs", name, "generated")
    stem = "densification" if cfg["task"] == "upsample" else "inpainting"
    result_path, target_path, input_path = (os.path.join(out, f"{stem}_{k}") for k in ("result", "target", "input"))
    for d in (result_path, target_path, input_path):
        os.makedirs(d, exist_ok=True)

    sched_cfg = None
    if a.weights:
        from .checkpoint import load_output_dir
        ck = load_output_dir(a.weights, with_vae=True, ema=a.ema)
        cfg["unet"], usd, sched_cfg, cfg["vae"], vsd = ck["unet_config"], ck["unet"], ck["scheduler_config"], ck["vae_config"], ck["vae"]
    else:
        usd = synth_state_dict(unet_param_shapes(cfg["unet"]), seed=a.seed, prefix="")
        vsd = synth_state_dict(vae_param_shapes(cfg["vae"]), seed=a.seed, prefix="vae.")
    unet = UNet2DModelHIP(cfg["unet"])
    unet.load_state_dict(usd)
    vae = AutoencoderKLHIP(cfg["vae"])
    vae.load_state_dict(vsd)
    # ldm/inference_conditional.py:121-134: DDPM scheduler (strided ancestral sampling), SparseRangeImageEncoder2 for up-sampling
    pipe = LDMUpscalePipelineRange(unet=unet, scheduler=DDPMSchedulerHIP(sched_cfg), vae=vae)
    condition_encoder = SparseRangeImageEncoder2() if cfg["task"] == "upsample" else None

    f = cfg["vae"].downscale
    img_shape = (cfg["vae"].in_channels, cfg["unet"].sample_size[0] * f, cfg["unet"].sample_size[1] * f)
    jpg = load_batch(a.input, B, img_shape, a.seed).to(dev)
    batch = {"jpg": jpg}
    if cfg["task"] == "upsample":
        batch["down"] = downsample_range_image(jpg, cfg["rate"]).contiguous()
    else:
        batch["inpainting_mask"], batch["masked_image"] = inpainting_inputs(jpg, cfg["fraction"])
    to_range = sensor_for(jpg.shape[3])
    lim = cfg["range_limit"]

    def write(dirname, image, seed):
        points, counts, bev_u8, _ = postprocess(to_range, image, max_depth=lim)
        for j in range(image.shape[0]):
            points[j, :counts[j]].tofile(os.path.join(dirname, f"{j}_seed_{seed}.bin"))     # ldm/inference_conditional.py:190-193
            save_png(bev_u8[j], os.path.join(dirname, f"{j}_seed_{seed}.png"))              # :194-195

    n_iter = a.samples // B // world + 1                   # ldm/inference_conditional.py:158
    for i in range(n_iter):
        seed = rank + world * i
        generator = torch.Generator().manual_seed(seed)    # :159 (a CPU generator: x_T is drawn on the host, as there)
        images = pipe(image=batch["down"] if cfg["task"] == "upsample" else batch["masked_image"],
                      mask=None if cfg["task"] == "upsample" else batch["inpainting_mask"],
                      condition_encoder=condition_encoder, generator=generator, batch_size=B,
                      num_inference_steps=steps, output_type="torch")
        write(result_path, images, seed)
        if seed == 0:                                      # :196-210
            write(target_path, jpg, seed)
            shown = (sparse_input_image(jpg, batch["down"], cfg["rate"]) if cfg["task"] == "upsample"
                     else batch["masked_image"])
            write(input_path, shown.contiguous(), seed)
    D.barrier()
    if rank == 0:
        print(f"wrote {n_iter * B} {stem} results to {result_path}")


if __name__ == "__main__":
    main()
