#!/usr/bin/env python3
"""Goldens at the HEADLINE batch (BASELINE config 2: `eval_batch_size: 16`, ldm/configs/RangeLDM.yaml:30) and a full-width
run of BASELINE config 4's pipeline, both computed by REFERENCE code -- test infrastructure, build container only.

Writes tests/golden/b16.npz and tests/golden/upfull.npz:
  * b16_eps_t{480,37}_f16   eps of the reference-composed full-width `Model` (vae/sgm/modules/diffusionmodules/model.py:521-704
                            after the reference's surgery, see validate_unet_against_reference.py) on a (16, 5, 256, 16) batch
  * b16_ddim3_latent_f16    x_0 (before the /0.18215) of a 3-step DDIM run of the reference's own `LDMPipelineRange.__call__`
                            loop (ldm/pipelines.py:353-362) at batch 16 with pos-encoding, driving that `Model`
  * b16_up_eps_t700_f16, b16_nusc4_eps_t250, b16_nusc32_eps_t610_f16   the 12-channel (config 4) Model at batch 16 and the nuScenes-shape
                            (256 x 8) Model at the 4 images per GPU config 3 runs at on 8 GPUs and at its whole batch of 32, one forward each
  * upfull_latent, upfull_image_f16   10 strided-DDPM steps (injected noise) of `LDMUpscalePipelineRange.__call__`
                            (ldm/pipelines.py:414-519, loop :466-507) at batch 2 on the full-width 12-channel UNet with the
                            reference's `SparseRangeImageEncoder2` (ldm/encoders.py:90-95) and the sgm Decoder
Inputs are regenerated from rangeldm_amd.synth on both sides (seeds below) and are not stored.  The oracle is checked against
the same outputs here, so the CPU suite can re-check it from the vectors alone.

    python -m oracle.validate_batch16_against_reference [--check] [--long]        (~3 min on 8 cores; --long: + ~25 min)
  * b16long_ddim50_latent_f16 (--long; tests/golden/b16long.npz)   x_0 of the HEADLINE workload: the same loop for all 50 DDIM steps at batch 16
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import validate_against_reference as V  # noqa: E402
from oracle import validate_unet_against_reference as VU  # noqa: E402
from oracle import unet as o_unet, vae as o_vae, schedulers as o_sched, pipelines as o_pipe  # noqa: E402
from rangeldm_amd.config import UNetConfig, VAEConfig  # noqa: E402
from rangeldm_amd.params import vae_param_shapes  # noqa: E402
from rangeldm_amd.synth import synth_state_dict, normal  # noqa: E402

T = VU.T
check = V.check

B16_X = (61, "b16/x", (16, 5, 256, 16))
B16_XT = (62, "b16/x_T", (16, 4, 256, 16))
UP_COND = (63, "upfull/cond", (2, 2, 1024, 16))
UP_XT = (64, "upfull/x_T", (2, 4, 256, 16))
UP_STEPS = 10
UP16_X = (66, "b16/up_x", (16, 12, 256, 16))
NUSC4_X = (67, "b16/nusc_x", (4, 5, 256, 8))
NUSC32_X = (68, "b16/nusc32_x", (32, 5, 256, 8))


def up_step_noise(i):
    return normal(65, f"upfull/z/{i}", (2, 4, 256, 16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--long", action="store_true", help="also the 50-step batch-16 run (tests/golden/b16long.npz, ~25 min)")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gold = {}
    sgm = V.import_sgm_model()
    lu = V.import_ldm_utils()
    lp = V.import_ldm_pipelines()
    att = sys.modules["sgm.modules.attention"]
    le = V._load("ref_encoders", os.path.join(V.REF, "ldm", "encoders.py"))
    dist = V._load("ref_distributions", os.path.join(V.REF, "vae", "sgm", "modules", "distributions", "distributions.py"))

    print("== batch 16, full-width RangeLDM UNet: forward of the reference-composed Model at two timesteps")
    full_sgm = VU.sgm_sinusoid({})
    fsd = VU.synth_unet_sd(full_sgm, "ref/full.")
    fm = VU.build_reference_unet(sgm, lu, att, full_sgm)
    VU.load_ref_unet(fm, fsd, 4)
    fsd_t = {k: T(v) for k, v in fsd.items()}
    x = T(normal(*B16_X))
    for t in (480, 37):
        t0 = time.time()
        with torch.no_grad():
            ref = fm(x, torch.full((16,), t))
        mine = o_unet.unet_forward(fsd_t, full_sgm, x, t)
        check(f"Model(RangeLDM full width) B=16 t={t} ({time.time() - t0:.0f} s)", mine, ref, 2e-5 * float(ref.abs().max()))
        assert float(ref.abs().max()) < 6e4
        gold[f"b16_eps_t{t}_f16"] = ref.numpy().astype(np.float16)

    print("== batch 16, 3 DDIM steps of the reference's LDMPipelineRange loop (latent only: no decode)")
    zrec = []

    class LatentOnlyVAE:
        config = VAEConfig()

        class _D:
            def __init__(self, s):
                self.sample = s

        def decode(self, z):
            zrec.append(z.clone())
            return self._D(z)

    x_T = T(normal(*B16_XT))
    lp.randn_tensor = lambda shape, generator=None, device=None, dtype=None, layout=None: x_T.clone()
    pipe = lp.LDMPipelineRange(vae=LatentOnlyVAE(), unet=VU.RefUNet(fm, full_sgm), scheduler=o_sched.OracleDDIMScheduler(),
                               pos_encoding=True)
    t0 = time.time()
    pipe(batch_size=16, generator=None, num_inference_steps=3, output_type="torch")
    lat_ref = zrec[-1] * VAEConfig().scaling_factor
    print(f"  (reference loop {time.time() - t0:.0f} s, |x_0| max {float(lat_ref.abs().max()):.1f})")
    ofull = o_unet.OracleUNet(full_sgm, fsd)
    mine = o_pipe.ldm_pipeline(None, ofull, o_sched.OracleDDIMScheduler(), x_T, 3, pos_encoding=True, decode=False)
    check("LDMPipelineRange 3 DDIM steps B=16, final latent", mine, lat_ref, 2e-4 * float(lat_ref.abs().max()))
    assert float(lat_ref.abs().max()) < 6e4
    gold["b16_ddim3_latent_f16"] = lat_ref.numpy().astype(np.float16)
    if args.long:
        # the headline workload itself: 50 DDIM steps at batch 16 through the reference's loop (~12 min of reference + ~12 min of oracle)
        print("== batch 16, the full 50 DDIM steps of the reference's LDMPipelineRange loop (latent only)")
        t0 = time.time()
        pipe(batch_size=16, generator=None, num_inference_steps=50, output_type="torch")
        lat50 = zrec[-1] * VAEConfig().scaling_factor
        print(f"  (reference loop {time.time() - t0:.0f} s, |x_0| max {float(lat50.abs().max()):.1f})")
        t0 = time.time()
        mine = o_pipe.ldm_pipeline(None, ofull, o_sched.OracleDDIMScheduler(), x_T, 50, pos_encoding=True, decode=False)
        check(f"LDMPipelineRange 50 DDIM steps B=16, final latent ({time.time() - t0:.0f} s)", mine, lat50,
              2e-4 * float(lat50.abs().max()))
        assert float(lat50.abs().max()) < 6e4
        gold["b16long_ddim50_latent_f16"] = lat50.numpy().astype(np.float16)
    del fm, pipe

    print("== config 4 at full width: LDMUpscalePipelineRange, 10 strided-DDPM steps, batch 2, 12-channel UNet")
    ucfg = VU.sgm_sinusoid(dict(in_channels=12))
    usd = VU.synth_unet_sd(ucfg, "ref/up.")
    um = VU.build_reference_unet(sgm, lu, att, ucfg)
    VU.load_ref_unet(um, usd, 4)
    vcfg = VAEConfig()
    vsd = synth_state_dict(vae_param_shapes(vcfg), prefix="vae.")
    enc, dec = VU.build_reference_vae(sgm, vcfg, vsd)
    zrec = []
    rv = VU.RefVAE(enc, dec, dist, vcfg, record=zrec)
    cond_img = T(normal(*UP_COND))
    x_T = T(normal(*UP_XT))
    step_z = [T(up_step_noise(i)) for i in range(UP_STEPS)]
    lp.randn_tensor = lambda shape, generator=None, device=None, dtype=None, layout=None: x_T.clone()

    class InjectedDDPM(o_sched.OracleDDPMScheduler):
        def step(self, eps, t, x, **kw):
            i = int((self.timesteps == int(t)).nonzero()[0])
            return super().step(eps, t, x, noise=step_z[i] if int(t) > 0 else None)

    pipe = lp.LDMUpscalePipelineRange(vae=rv, unet=VU.RefUNet(um, ucfg), scheduler=InjectedDDPM())
    t0 = time.time()
    ref_img = pipe(image=cond_img, condition_encoder=le.SparseRangeImageEncoder2(), batch_size=2, generator=None,
                   num_inference_steps=UP_STEPS, output_type="torch")
    lat_ref = zrec[-1] * vcfg.scaling_factor
    print(f"  (reference loop {time.time() - t0:.0f} s, |x_0| max {float(lat_ref.abs().max()):.1f}, image |max| "
          f"{float(ref_img.abs().max()):.2f})")
    ovae = o_vae.OracleVAE(vcfg, vsd)
    oun = o_unet.OracleUNet(ucfg, usd)
    zs = step_z[:UP_STEPS - 1] + [None]
    mine = o_pipe.ldm_pipeline(ovae, oun, o_sched.OracleDDPMScheduler(), x_T, UP_STEPS, pos_encoding=False, step_noise=zs,
                               cond=o_pipe.sparse_range_image_encoder2(cond_img))
    check("LDMUpscalePipelineRange full width, 10 steps, image", mine, ref_img, 2e-3 * float(ref_img.abs().max()))
    gold["upfull_latent"] = lat_ref.numpy()
    gold["upfull_image_f16"] = ref_img.numpy().astype(np.float16)

    print("== the other configurations at the batch they run at: config 4 (12-channel UNet) at 16, config 3 (nuScenes 256x8) at 4 per GPU")
    xu = T(normal(*UP16_X))
    with torch.no_grad():
        ref = um(xu, torch.full((16,), 700))
    mine = o_unet.unet_forward({k: T(v) for k, v in usd.items()}, ucfg, xu, 700)
    check("Model(upsample full width, 12 ch) B=16", mine, ref, 2e-5 * float(ref.abs().max()))
    gold["b16_up_eps_t700_f16"] = ref.numpy().astype(np.float16)
    del um
    ncfg = VU.sgm_sinusoid(dict(sample_size=(256, 8)))
    nsd = VU.synth_unet_sd(ncfg, "ref/nusc.")
    nm = VU.build_reference_unet(sgm, lu, att, ncfg)
    VU.load_ref_unet(nm, nsd, 4)
    xn = T(normal(*NUSC4_X))
    with torch.no_grad():
        ref = nm(xn, torch.full((4,), 250))
    mine = o_unet.unet_forward({k: T(v) for k, v in nsd.items()}, ncfg, xn, 250)
    check("Model(nuScenes 256x8 full width) B=4", mine, ref, 2e-5 * float(ref.abs().max()))
    gold["b16_nusc4_eps_t250"] = ref.numpy()
    # ... and config 3 as ONE GPU runs it (`eval_batch_size: 32`): the batch at which the nuScenes network gets its persistent launches
    xn32 = T(normal(*NUSC32_X))
    with torch.no_grad():
        ref = nm(xn32, torch.full((32,), 610))
    mine = o_unet.unet_forward({k: T(v) for k, v in nsd.items()}, ncfg, xn32, 610)
    check("Model(nuScenes 256x8 full width) B=32", mine, ref, 2e-5 * float(ref.abs().max()))
    assert float(ref.abs().max()) < 6e4
    gold["b16_nusc32_eps_t610_f16"] = ref.numpy().astype(np.float16)

    bad = [c for c in V.CHECKS if not c[3]]
    print(f"\n{len(V.CHECKS) - len(bad)}/{len(V.CHECKS)} checks passed")
    if bad:
        sys.exit(1)
    if not args.check:
        groups = {}
        for k, v in gold.items():
            groups.setdefault(k.split("_", 1)[0], {})[k] = np.ascontiguousarray(v)
        for gname, d in groups.items():
            path = os.path.join(V.GOLD, f"{gname}.npz")
            np.savez_compressed(path, **d)
            print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
