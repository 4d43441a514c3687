"""CPU: the C-ABI library loads and exports exactly what include/rangeldm_hip.h declares; host-side mirrors behave
like the reference objects (no GPU compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "rangeldm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rldm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rangeldm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rangeldm_hip.h but not exported"
    # and the python binding table covers the header, so a new entry point cannot be forgotten in the shim
    assert sorted(_lib.PROTOTYPES) == syms
    _lib.lib()


def test_struct_layouts_match_header():
    from rangeldm_amd import _lib
    assert ctypes.sizeof(_lib.UNetConfigC) == 4 * (6 + 3 * 8 + 6)
    assert ctypes.sizeof(_lib.VAEConfigC) == 4 * (4 + 8 + 6)
    assert ctypes.sizeof(_lib.ConvDescC) == 4 * 13
    assert ctypes.sizeof(_lib.TrainConvDescC) == 4 * 8 and ctypes.sizeof(_lib.AdamWConfigC) == 4 * 8
    assert ctypes.sizeof(_lib.PackDescC) == 48 and _lib.PackDescC.N.offset == 32
    assert ctypes.sizeof(_lib.TrainFuseC) == 152 and _lib.TrainFuseC.cs_out.offset == 64 and _lib.TrainFuseC.gs_out.offset == 144
    assert ctypes.sizeof(_lib.LidarConfigC) == 4 * (7 + 3 + 6 + 1)
    assert _lib.SamplerConfigC.coef.offset == 24 and _lib.SamplerConfigC.timesteps.offset == 32
    assert _lib.SamplerConfigC.plan_flags.offset == 40 and ctypes.sizeof(_lib.SamplerConfigC) == 48
    assert _lib.SamplerConfigC.prediction_type.offset == 44


def test_product_has_no_cpu_fallback():
    """The product path must fail loudly without a GPU and must never import the oracle."""
    import subprocess
    import sys
    code = ("import sys; import rangeldm_amd, rangeldm_amd.unet, rangeldm_amd.vae, rangeldm_amd.pipelines, "
            "rangeldm_amd.schedulers, rangeldm_amd.distributed; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'product imports oracle'")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    if not torch.cuda.is_available():
        from rangeldm_amd.unet import UNet2DModelHIP
        from rangeldm_amd.config import UNetConfig
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            UNet2DModelHIP(UNetConfig())


def test_scheduler_host_side_matches_oracle():
    from rangeldm_amd.schedulers import DDIMSchedulerHIP, DDPMSchedulerHIP
    from oracle.schedulers import OracleDDIMScheduler, OracleDDPMScheduler
    for n in (50, 10, 1000, 7):
        s, so = DDIMSchedulerHIP(), OracleDDIMScheduler()
        s.set_timesteps(n)
        so.set_timesteps(n)
        assert s.timesteps.tolist() == so.timesteps.tolist()
    p, po = DDPMSchedulerHIP(), OracleDDPMScheduler()
    p.set_timesteps(50)
    po.set_timesteps(50)
    for t in (980, 500, 20, 0):
        c = p.coefficients(t)
        ref = po.coefficients(t)
        assert abs(c[2] - ref[0]) < 1e-7 and abs(c[3] - ref[1]) < 1e-7 and abs(c[4] - ref[2]) < 1e-7
    assert p.coefficients(0)[4] == 0.0                         # no noise at the last step (t == 0)
    assert abs(p.coefficients(980)[4] - 0.5697414) < 1e-6
    assert torch.equal(p.alphas_cumprod, po.alphas_cumprod)
    # DDIMPipelineRange converts whatever scheduler it is given (ldm/pipelines.py:135-139)
    d = DDIMSchedulerHIP.from_config(p.config)
    assert d.config.clip_sample is False and d.config.num_train_timesteps == 1000
    assert s.init_noise_sigma == 1.0 and s.scale_model_input("x", 3) == "x"
    import inspect
    assert "eta" in inspect.signature(s.step).parameters and "eta" not in inspect.signature(p.step).parameters


@pytest.mark.parametrize("ptype", ["epsilon", "v_prediction", "sample"])
def test_oracle_scheduler_prediction_types_known_answers(ptype):
    """The oracle's `prediction_type` branches (restated from diffusers' published forms; the reference trains epsilon or
    v_prediction, ldm/train_unconditional.py:505-510) against the closed forms evaluated in float64, the velocity target, the
    min-SNR `+ 1` of :532-534, and the host shim's acceptance of the three types."""
    import numpy as np
    from rangeldm_amd.config import SchedulerConfig
    from rangeldm_amd.training import snr_weights
    from oracle.schedulers import OracleDDIMScheduler, OracleDDPMScheduler
    cfg = SchedulerConfig(prediction_type=ptype)
    g = torch.Generator().manual_seed(11)
    x, o = torch.randn(2, 3, 4, 4, generator=g), torch.randn(2, 3, 4, 4, generator=g)
    ac = np.cumprod(1.0 - np.linspace(1e-4, 0.02, 1000, dtype=np.float32).astype(np.float64))
    for t, tp in ((980, 960), (20, 0)):
        a_t, a_p = ac[t], ac[tp]
        xd, od = x.double().numpy(), o.double().numpy()
        if ptype == "epsilon":
            x0, pe = (xd - np.sqrt(1 - a_t) * od) / np.sqrt(a_t), od
        elif ptype == "v_prediction":
            x0, pe = np.sqrt(a_t) * xd - np.sqrt(1 - a_t) * od, np.sqrt(a_t) * od + np.sqrt(1 - a_t) * xd
        else:
            x0, pe = od, (xd - np.sqrt(a_t) * od) / np.sqrt(1 - a_t)
        d = OracleDDIMScheduler(cfg)
        d.set_timesteps(50)
        got = d.step(o, t, x).prev_sample.double().numpy()
        want = np.sqrt(a_p) * x0 + np.sqrt(1 - a_p) * pe
        assert np.abs(got - want).max() < 3e-5 * (1 + np.abs(want).max())
        p = OracleDDPMScheduler(cfg)
        p.set_timesteps(50)
        gotp = p.step(o, t, x, noise=torch.zeros_like(x)).prev_sample.double().numpy()
        cur_a = a_t / a_p
        wantp = np.sqrt(a_p) * (1 - cur_a) / (1 - a_t) * x0 + np.sqrt(cur_a) * (1 - a_p) / (1 - a_t) * xd
        assert np.abs(gotp - wantp).max() < 3e-5 * (1 + np.abs(wantp).max())
    ts = torch.tensor([3, 977])
    v = OracleDDPMScheduler(cfg).get_velocity(x, o, ts).double().numpy()
    sa, sb = np.sqrt(ac[[3, 977]]).reshape(2, 1, 1, 1), np.sqrt(1 - ac[[3, 977]]).reshape(2, 1, 1, 1)
    assert np.abs(v - (sa * o.double().numpy() - sb * x.double().numpy())).max() < 3e-6
    acf = torch.from_numpy(ac).float()
    w = snr_weights(acf, ts, 5.0, v_prediction=True).double().numpy()
    snr = ac[[3, 977]] / (1 - ac[[3, 977]]) + 1.0
    assert np.abs(w - np.minimum(snr, 5.0) / snr).max() < 1e-6
    with pytest.raises(ValueError, match="must be one of"):
        from rangeldm_amd.schedulers import DDPMSchedulerHIP
        DDPMSchedulerHIP(SchedulerConfig(prediction_type="velocity"))


def test_condition_encoder_matches_oracle():
    from rangeldm_amd.encoders import SparseRangeImageEncoder2
    from oracle.pipelines import sparse_range_image_encoder2
    x = torch.randn(2, 2, 64, 4)
    assert torch.equal(SparseRangeImageEncoder2()(x), sparse_range_image_encoder2(x))


def test_synth_is_deterministic_and_index_addressed():
    from rangeldm_amd.synth import latent_noise, synth_state_dict
    a = latent_noise(1, 5, (4, 8, 2))
    assert np.array_equal(a, latent_noise(1, 5, (4, 8, 2))) and not np.array_equal(a, latent_noise(1, 6, (4, 8, 2)))
    sd = synth_state_dict({"a.conv1.weight": (4, 3, 3, 3), "a.norm1.weight": (4,), "a.conv1.bias": (4,)})
    assert abs(float(sd["a.norm1.weight"].mean()) - 1) < 0.2 and np.abs(sd["a.conv1.bias"]).max() <= 0.02


def test_randn_tensor_semantics():
    from rangeldm_amd.schedulers import randn_tensor
    g = torch.Generator().manual_seed(3)
    a = randn_tensor((2, 3), generator=g, device="cpu")
    g = torch.Generator().manual_seed(3)
    assert torch.equal(a, torch.randn((2, 3), generator=g))
    gs = [torch.Generator().manual_seed(i) for i in range(2)]
    b = randn_tensor((2, 3), generator=gs, device="cpu")
    assert torch.equal(b[1:], torch.randn((1, 3), generator=torch.Generator().manual_seed(1)))


def test_inference_driver_index_arithmetic():
    """rangeldm_amd.inference reproduces ldm/inference.py:159,174-176: every image index below `samples` is written by
    exactly one (rank, iteration, j), for any GPU count."""
    from rangeldm_amd.inference import image_indices, plan_iterations
    for samples, B, world in [(1000, 16, 1), (1000, 16, 8), (37, 4, 3), (16, 16, 2), (5, 8, 4)]:
        seen = []
        for rank in range(world):
            for i in range(plan_iterations(samples, B, world)):
                seen += [g for _, g in image_indices(i, B, rank, world, samples)]
        assert sorted(seen) == list(range(samples)), (samples, B, world)


def test_inference_driver_reads_reference_yaml(tmp_path):
    from rangeldm_amd.inference import load_config
    y = tmp_path / "RangeLDM.yaml"
    y.write_text("eval_batch_size: 16\nddpm_num_inference_steps: 50\nwith_vae: True\npos_encoding: True\nall_circonv: True\n"
                 "ddim: True\nvae_config: ../vae/configs/kitti360.yaml\n"
                 "model_config:\n  sample_size: [256, 16]\n  in_channels: 5\n  out_channels: 4\n  layers_per_block: 2\n"
                 "  block_out_channels: [128, 128, 256, 256]\n"
                 "  down_block_types: [DownBlock2D, AttnDownBlock2D, AttnDownBlock2D, AttnDownBlock2D]\n"
                 "  up_block_types: [AttnUpBlock2D, AttnUpBlock2D, AttnUpBlock2D, UpBlock2D]\n")
    c = load_config(str(y))
    assert c["unet"].sample_size == (256, 16) and c["unet"].in_channels == 5 and c["vae"].sample_size == (1024, 64)
    assert c["steps"] == 50 and c["batch"] == 16 and c["pos_encoding"] is True and c["ddim"] is True
    # ldm/inference.py:99-118: only the all_circonv surgery is built; anything else must refuse, not run with other padding
    import pytest
    txt = y.read_text()
    (tmp_path / "plain.yaml").write_text(txt.replace("all_circonv: True\n", ""))
    with pytest.raises(NotImplementedError):
        load_config(str(tmp_path / "plain.yaml"))
    (tmp_path / "sub.yaml").write_text(txt.replace("all_circonv: True\n", "sub_circonv: True\n"))
    with pytest.raises(NotImplementedError):
        load_config(str(tmp_path / "sub.yaml"))
    # the VAE geometry comes from the sgm yaml `vae_config` points at when it is reachable (here: an 8x VAE)
    vdir = tmp_path / "vae"
    vdir.mkdir()
    (vdir / "v.yaml").write_text("model:\n  params:\n    ddconfig:\n      attn_type: none\n      double_z: true\n      z_channels: 4\n"
                                 "      in_channels: 2\n      out_ch: 2\n      ch: 32\n      ch_mult: [1, 2, 4, 4]\n      num_res_blocks: 2\n")
    (tmp_path / "v8.yaml").write_text(txt.replace("../vae/configs/kitti360.yaml", "vae/v.yaml"))
    c8 = load_config(str(tmp_path / "v8.yaml"))
    assert c8["vae"].ch == 32 and c8["vae"].downscale == 8 and c8["vae"].sample_size == (2048, 128)


def test_conditional_inputs_match_reference_dataset(golden):
    """`down` / `inpainting_mask` / `masked_image` as RangeDataset.__getitem__ builds them (ldm/dataset.py:340-362; vectors from
    the reference class, oracle/validate_lidar_against_reference.py) and the sparse-input picture of the conditional driver."""
    import numpy as np
    import torch
    from rangeldm_amd.conditional import downsample_range_image, inpainting_inputs, sparse_input_image, encode_condition
    g = golden("condds")
    jpg = torch.from_numpy(g["condds_jpg"])
    assert np.array_equal(downsample_range_image(jpg, 4).numpy(), g["condds_up4_down_ref"])
    assert np.array_equal(downsample_range_image(jpg, [2, 4]).numpy(), g["condds_up24_down_ref"])
    for tag, frac in (("inp", 0.0625), ("inpwrap", 1.25)):
        m, mi = inpainting_inputs(jpg, frac)
        assert np.array_equal(m.numpy(), g[f"condds_{tag}_mask_ref"]) and np.array_equal(mi.numpy(), g[f"condds_{tag}_masked_ref"])
    # batched form + the -1-filled picture of ldm/inference_conditional.py:176-182
    xb = torch.stack([jpg, jpg * 2])
    down = downsample_range_image(xb, 4)
    assert down.shape == (2, 2, 64, 4) and torch.equal(down[1], downsample_range_image(jpg * 2, 4))
    shown = sparse_input_image(xb, down, 4)
    assert torch.equal(shown[..., 2::4], down) and float(shown[..., 0::4].max()) == -1.0
    # the up-sampling condition is the folded low-resolution image (ldm/train_conditional.py:419-421)
    cond = encode_condition({"down": down})
    assert cond.shape == (2, 8, 16, 4)


def test_conditional_driver_reads_reference_yaml(tmp_path):
    from rangeldm_amd.inference_conditional import load_conditional_config
    import pytest
    y = tmp_path / "upsample.yaml"
    y.write_text("ddim: True\nddpm_num_inference_steps: 50\nwith_vae: True\nvae_config: ../vae/configs/kitti360.yaml\n"
                 "block_out_channels: [128, 128, 256, 256]\ninpainting: null\nupsample: 4\nall_circonv: True\nmodel_config: null\n"
                 "resolution: [1024, 64]\neval_batch_size: 16\n")
    c = load_conditional_config(str(y))
    assert c["task"] == "upsample" and c["rate"] == 4 and c["unet"].in_channels == 12 and c["unet"].sample_size == (256, 16)
    assert c["cond_channels"] == 8 and c["vae"].sample_size == (1024, 64) and c["range_limit"] == 70.0
    (tmp_path / "inp.yaml").write_text(y.read_text().replace("inpainting: null", "inpainting: 0.0625").replace("upsample: 4", "upsample: null"))
    c = load_conditional_config(str(tmp_path / "inp.yaml"))
    assert c["task"] == "inpainting" and c["unet"].in_channels == 9 and abs(c["fraction"] - 0.0625) < 1e-9
    (tmp_path / "both.yaml").write_text(y.read_text().replace("inpainting: null", "inpainting: 0.1"))
    with pytest.raises(ValueError):
        load_conditional_config(str(tmp_path / "both.yaml"))
    assert load_conditional_config("upsample")["unet"].in_channels == 12
