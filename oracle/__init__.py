"""CPU oracle for the RangeLDM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain PyTorch fp32 restatement of the arithmetic the reference executes on the denoising path
(UNet2DModel forward, DDPM/DDIM scheduler steps, AutoencoderKL encode/decode, the four pipeline loops).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package; nothing under
`rangeldm_amd/` does, and the product path fails loudly when the HIP library is missing.

Pinning status (oracle/validate_against_reference.py, oracle/validate_unet_against_reference.py):
  * VAE Encoder/Decoder, ResnetBlock (with temb), single-head AttnBlock, circular Conv2d, Downsample2D, Upsample,
    DiagonalGaussian sampling, SparseRangeImageEncoder2 and the pipeline loops are checked against the reference's own
    importable modules (/root/reference/vae/sgm/modules/diffusionmodules/model.py, ldm/utils.py, ldm/encoders.py,
    ldm/pipelines.py under diffusers stubs) in the build container.
  * The multi-head (d = 8) attention block is checked against the reference's CrossAttention (vae/sgm/modules/attention.py:
    194-284), the sinusoidal time embedding against get_timestep_embedding (model.py:28-46), and the whole UNet2DModel wiring
    (skip concat, temb, up / down sampling, attention placement) against the reference's skip-concat UNet `Model`
    (model.py:521-704) after the reference's own surgery -- small, full-width RangeLDM, upsample and full-size RangeDM
    configurations -- plus 50-step full-width samplers driven by the reference's LDMPipelineRange loop.
  * Golden vectors produced by those runs are committed under tests/golden/ (weights are regenerated from
    rangeldm_amd.synth on both sides).
  * Still restated only: the closed forms of diffusers' DDPMScheduler / DDIMScheduler (third-party, un-vendored, un-pinned:
    `check_min_version("0.21.0.dev0")`, ldm/train_unconditional.py:49; the reference holds no test or vector for them) --
    anchored on the known answers of SURVEY.md B.4.
"""
