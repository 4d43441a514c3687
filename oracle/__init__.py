"""CPU oracle for the RangeLDM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain PyTorch fp32 restatement of the arithmetic the reference executes on the denoising path
(UNet2DModel forward, DDPM/DDIM scheduler steps, AutoencoderKL encode/decode, the four pipeline loops).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package; nothing under
`rangeldm_amd/` does, and the product path fails loudly when the HIP library is missing.

Pinning status (see oracle/validate_against_reference.py, DESIGN.md section "Oracle"):
  * VAE Encoder/Decoder, ResnetBlock (with temb), single-head AttnBlock, circular Conv2d, Downsample2D, Upsample,
    DiagonalGaussian sampling, SparseRangeImageEncoder2 and the pipeline loops are checked against the reference's own
    importable modules (/root/reference/vae/sgm/modules/diffusionmodules/model.py, ldm/utils.py, ldm/encoders.py,
    ldm/pipelines.py under diffusers stubs) in this container; golden vectors produced by that run are committed
    under tests/golden/.
  * UNet2DModel / DDPMScheduler / DDIMScheduler live in third-party `diffusers` (un-vendored, un-pinned:
    `check_min_version("0.21.0.dev0")`, ldm/train_unconditional.py:49; API window 0.21-0.26), which is absent from
    /root/reference and not installed.  The reference holds no tests or golden vectors for them, so for these pieces
    parity is UNPINNED beyond the cross-checks of SURVEY.md A.6 (parameter count == README's 115 MB, block-level
    equality with the sgm analogues, closed-form scheduler known answers B.4).
"""
