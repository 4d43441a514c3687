#!/usr/bin/env python3
"""bench.py -- range-images/sec for RangeLDM KITTI-360 64x1024, 50-step DDIM (eta=0) + VAE decode at batch 16 per GPU
(BASELINE.json metric, configs[1]).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: x_T (already resident in HBM) -> 50 x [UNet forward + DDIM step]
-> VAE decode -> (N>1: RCCL all-gather of the finished range images).  Weak scaling (default): every GPU samples its own
batch of 16 (whole samples are sharded; no collective on the data path except the final all-gather).
`--scaling strong --batch 32 --preset nuscenes` is BASELINE config 3: ONE global batch of 32 split over the ranks
(4 images per GPU at N = 8), same all-gather; `value` is then global-batch images / time.  Weights are synthetic
(rangeldm_amd.synth), the architecture and sizes are the reference's.

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  roofline     -- dominant kernel: algorithmic FLOPs of its launches / HIP-event time of those launches, measured in this
                  process right after the timed region by an instrumented eager pass on the sampler's stream
  roofline_worst -- the kernel maximising (share of kernel time) x (1 - fraction of its roofline), with its counter traffic ratio
  cpu_baseline -- the CPU oracle (torch fp32 restatement of the reference path; kind "port") timed on this host's
                  cores on a bounded sample of the same workload (2 warm-ups, median of 3; host_cores / threads / cpu_model stated)
  cpu_baseline_c1 -- BASELINE config 1 in full on the host: RangeDM, 10-step DDIM, batch 1 (the GPU leg is in other_configs)
  fell_back    -- whether any persistent launch failed its self-check during the run (plans rebuilt as one launch per layer)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA bf16, MI355X_MICROARCH.md (measured 2495)
PEAK_HBM_GBS = 8000.0


def build_models(preset, seed):
    from rangeldm_amd.config import PRESETS
    from rangeldm_amd.params import unet_param_shapes, vae_param_shapes
    from rangeldm_amd.synth import synth_state_dict
    from rangeldm_amd.unet import UNet2DModelHIP
    from rangeldm_amd.vae import AutoencoderKLHIP
    p = PRESETS[preset]
    usd = synth_state_dict(unet_param_shapes(p["unet"]), seed=seed)
    unet = UNet2DModelHIP(p["unet"])
    unet.load_state_dict(usd)
    vae, vsd = None, None
    if p["vae"] is not None:
        vsd = synth_state_dict(vae_param_shapes(p["vae"]), seed=seed, prefix="vae.")
        vae = AutoencoderKLHIP(p["vae"])
        vae.load_state_dict(vsd)
    return p, unet, vae, usd, vsd


def host_info():
    """CPU model, physical cores (unique (package, core id) pairs of /proc/cpuinfo) and logical CPUs of this host."""
    model, pairs, phys, core, logical = "unknown", set(), None, None, 0
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                pairs.add((phys, v))
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    return {"cpu_model": model, "host_cores": len(pairs) or logical, "logical_cpus": logical}


def _median_timed(fn, warm=2, runs=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def cpu_baseline(p, usd, vsd, batch, steps):
    """The headline workload on the host cores (BASELINE.md section 3 protocol: fp32, 2 warm-ups, median of 3): the oracle's UNet
    forward at the full batch and its VAE decode of `ndec` images, extrapolated to `steps` steps + a decode of the batch.
    Thread count: torch's fp32 conv path scales to ~16-32 threads on the GPU box's EPYC host and gets slower beyond (probe,
    tools/cpu_threads.py: 8 thr 1.68 s, 16 thr 1.24 s, 32 thr 1.46 s, 128 thr 9.0 s per UNet forward at batch 16), so the
    baseline times one forward at 16 and at 32 threads IN THIS RUN and uses the faster count (`thread_probe_unet_forward_s`) --
    `cores` is the number of threads used, `host_cores` what the host has."""
    from oracle.unet import OracleUNet
    from oracle.vae import OracleVAE
    from rangeldm_amd.synth import normal
    hi = host_info()
    ucfg, vcfg = p["unet"], p["vae"]
    ou = OracleUNet(ucfg, usd)
    x = torch.from_numpy(normal(1, "cpu/x", (batch, ucfg.in_channels, *ucfg.sample_size)))
    ts = iter([480, 460, 440, 420, 400, 380, 360, 340, 320])
    # the host's best: one warm + one timed forward at 16 and at 32 threads, the faster count runs the protocol (its two probe
    # forwards are the protocol's two warm-ups)
    probe = {}
    for th in sorted({max(1, min(16, hi["host_cores"])), max(1, min(32, hi["host_cores"]))}):
        torch.set_num_threads(th)
        ou(x, next(ts))
        t0 = time.perf_counter()
        ou(x, next(ts))
        probe[th] = time.perf_counter() - t0
    threads = min(probe, key=lambda k: probe[k])
    torch.set_num_threads(threads)
    t_unet, unet_runs = _median_timed(lambda: ou(x, next(ts)), warm=0 if len(probe) > 1 and threads == max(probe) else 1)
    t_dec, ndec, dec_runs = 0.0, 2, []
    if vcfg is not None:
        ov = OracleVAE(vcfg, vsd)
        z = torch.from_numpy(normal(1, "cpu/z", (ndec, vcfg.z_channels, *ucfg.sample_size)))
        t_dec, dec_runs = _median_timed(lambda: ov.decode(z))
    per_batch = steps * t_unet + (batch / ndec) * t_dec
    return dict(hi, **{
        "value": batch / per_batch, "unit": "range-images/sec", "cores": threads, "threads": threads, "kind": "port",
        "sample": f"oracle (torch {torch.__version__} fp32 CPU, {threads} threads on {hi['host_cores']} physical cores of "
                  f"{hi['cpu_model']}): UNet forward at batch {batch}, 2 warm-ups + median of 3 ({t_unet:.2f} s) x {steps} steps + "
                  f"VAE decode of {ndec} of {batch} images, 2 warm-ups + median of 3 ({t_dec:.2f} s) x {batch // ndec}, extrapolated",
        "unet_forward_s": [round(t, 3) for t in unet_runs], "decode_s": [round(t, 3) for t in dec_runs],
        "thread_probe_unet_forward_s": {str(k): round(v, 3) for k, v in probe.items()},
        "seconds_per_batch_extrapolated": per_batch})


def cpu_baseline_c1(seed, budget_s=30.0, threads=None):
    """BASELINE config 1 AS BASELINE DEFINES IT: RangeDM (ldm/configs/RangeDM.yaml, pixel space 3 -> 2 channels at 1024 x 64),
    10-step DDIM, batch 1, on the host -- the oracle's restatement of the reference loop (ldm/pipelines.py:224-248) IN FULL, no
    extrapolation.  2 warm-up forwards, then the whole 10-step loop `runs` times (3, or as many as fit the time budget -- stated
    in `runs`); value = 1 / median."""
    from oracle.unet import OracleUNet
    from oracle.schedulers import OracleDDIMScheduler
    from oracle.pipelines import ddim_pipeline
    from rangeldm_amd.config import PRESETS
    from rangeldm_amd.params import unet_param_shapes
    from rangeldm_amd.synth import synth_state_dict, latent_noise
    hi = host_info()
    threads = threads or max(1, min(32, hi["host_cores"]))       # (the count cpu_baseline's probe found faster, when it ran first)
    torch.set_num_threads(threads)
    p = PRESETS["RangeDM"]
    ucfg = p["unet"]
    ou = OracleUNet(ucfg, synth_state_dict(unet_param_shapes(ucfg), seed=seed))
    x_T = torch.from_numpy(latent_noise(seed, 0, (ucfg.out_channels, *ucfg.sample_size)))[None]
    warm = torch.cat([x_T, torch.zeros(1, ucfg.in_channels - ucfg.out_channels, *ucfg.sample_size)], 1)
    for t in (900, 800):
        ou(warm, t)
    runs = []
    while True:
        t0 = time.perf_counter()
        img = ddim_pipeline(ou, OracleDDIMScheduler(), x_T, 10, eta=0.0, pos_encoding=p["pos_encoding"])
        runs.append(time.perf_counter() - t0)
        if len(runs) >= 3 or runs[0] * (len(runs) + 1) > budget_s:      # 3 runs, or as many as fit the budget
            break
    assert torch.isfinite(img).all()
    med = float(np.median(runs))
    return dict(hi, **{
        "value": 1.0 / med, "unit": "range-images/sec", "cores": threads, "threads": threads, "kind": "port",
        "workload": "config 1 in full: RangeDM 64x1024 pixel space, 10-step DDIM, batch 1, fp32 on the host",
        "runs": len(runs), "seconds_per_image": [round(t, 3) for t in runs],
        "sample": f"oracle (torch {torch.__version__} fp32 CPU, {threads} threads): the whole 10-step loop, 2 warm-up forwards + "
                  f"median of {len(runs)} full runs ({med:.2f} s per image), nothing extrapolated"})


def in_graph_launch_us(preset, seed, batch, kernel, dev):
    """The dominant kernel's duration INSIDE the sampler's captured step graph (the production regime): a second sampler built
    with rldm_debug_set_flags(8192) has a one-thread kernel write the 100 MHz real-time counter between consecutive launches
    (tools/graph_trace.py).  The HIP-event figure of `roofline` times launches enqueued one by one, which spaces them out and
    lengthens each; this one is what the launch costs back to back.  -> (average us of `kernel`, us of the whole UNet forward)"""
    import ctypes as C
    from rangeldm_amd import _lib
    from rangeldm_amd.pipelines import LDMPipelineRange
    from rangeldm_amd.schedulers import DDIMSchedulerHIP
    from rangeldm_amd.synth import latent_noise
    base_flags = int(os.environ.get("RLDM_DBG_FLAGS", "0"))
    _lib.lib().rldm_debug_set_flags(8192 | base_flags)
    try:
        p, unet, vae, _, _ = build_models(preset, seed)
        pipe = LDMPipelineRange(vae=vae, unet=unet, scheduler=DDIMSchedulerHIP(), pos_encoding=p["pos_encoding"])
        shape = (p["unet"].out_channels, *p["unet"].sample_size)
        x = torch.from_numpy(np.stack([latent_noise(1, j, shape) for j in range(batch)])).to(dev)
        for _ in range(2):
            pipe(batch_size=batch, num_inference_steps=10, latents=x, output_type="torch")
        torch.cuda.synchronize()
        stamps = (C.c_ulonglong * 4096)()
        names = C.create_string_buffer(1 << 16)
        n = _lib.lib().rldm_debug_graph_trace(stamps, 4096, names, len(names))
        if n <= 0:
            return None
        nm = names.value.decode().split("\n")[:n]
        t = np.array([stamps[i] for i in range(n + 1)], dtype=np.int64)
        d = (t[1:] - t[:-1]) * 0.01                     # us (100 MHz)
        mine = [v for k, v in zip(nm, d) if k == kernel]
        return (float(np.mean(mine)) if mine else None), float(d.sum()), n
    finally:
        _lib.lib().rldm_debug_set_flags(base_flags)


NOMINAL_CLOCK_MHZ = 2400.0      # the engine clock the 2.5 PFLOP/s dense bf16 peak is quoted at (256 CUs x 4 SIMDs x 1024 FLOP/clk)


def box_calibration(dev):
    """What THIS box does, measured in this process behind the timed region (rldm_calibrate): a pure-MFMA loop on every SIMD
    (TFLOP/s + the shader clock under it) and a 1 GiB device copy (GB/s read + written).  The boxes of a pool differ by several
    per cent and the clock sags under matrix load: `value` is to be read against these."""
    from rangeldm_amd import _lib
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().rldm_calibrate(C.byref(a), C.byref(b), C.byref(c), _lib.stream_ptr(dev)), "rldm_calibrate")
    return {"mfma_tflops": round(a.value, 1), "mfma_clock_mhz": round(b.value, 1), "copy_gbs": round(c.value, 1),
            "mfma_frac_of_peak": round(a.value / PEAK_BF16_TFLOPS, 4),
            "note": "pure v_mfma_f32_32x32x16_bf16 loop, one wave per SIMD, ~20 ms (HIP events; clock = s_memtime / s_memrealtime "
                    "inside the kernel); copy = 1 GiB device to device, bytes read + written"}


def workload_clock_mhz(fn, dev):
    """Mean shader clock while `fn()` (work enqueued on the current stream) runs: clock stamps (one workgroup per XCD: XCC id,
    s_memtime, s_memrealtime) in front of and behind it, matched by XCC id -- (d s_memtime / d s_memrealtime) x 100 MHz."""
    from rangeldm_amd import _lib
    nb = 16
    s0 = torch.zeros(4 * nb, dtype=torch.int64, device=dev)
    s1 = torch.zeros(4 * nb, dtype=torch.int64, device=dev)
    st = _lib.stream_ptr(dev)
    _lib.check(_lib.lib().rldm_calib_clock_stamp(C.c_void_p(s0.data_ptr()), st), "rldm_calib_clock_stamp")
    fn()
    _lib.check(_lib.lib().rldm_calib_clock_stamp(C.c_void_p(s1.data_ptr()), st), "rldm_calib_clock_stamp")
    torch.cuda.synchronize()
    a, b = s0.cpu().numpy().reshape(nb, 4), s1.cpu().numpy().reshape(nb, 4)
    first = {}
    for r in a:
        if r[3] == 1:
            first.setdefault(int(r[0]), r)
    clocks = []
    for r in b:
        f = first.get(int(r[0]))
        if r[3] == 1 and f is not None and r[2] > f[2]:
            clocks.append(float(r[1] - f[1]) / float(r[2] - f[2]) * 100.0)
    return (round(float(np.median(clocks)), 1), len(set(int(r[0]) for r in b))) if clocks else (None, 0)


def roofline(pipe, sampler_handle, x_T, steps):
    from rangeldm_amd import _lib
    buf = C.create_string_buffer(1 << 16)
    _lib.check(_lib.lib().rldm_sampler_profile(sampler_handle, C.c_void_p(x_T.data_ptr()), buf, len(buf)),
               "rldm_sampler_profile")
    prof = json.loads(buf.value.decode())
    lanes = prof.get("lanes", 1)       # the sampler runs `lanes` concurrent chains of batch/lanes samples; lane 0 is profiled
    tot = {}
    for part, mult in (("unet_step", steps * lanes), ("vae_decode", lanes)):
        for k, v in prof.get(part, {}).items():
            t = tot.setdefault(k, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            for f in t:
                t[f] += v[f] * mult
    all_ms = sum(v["ms"] for v in tot.values())
    # dominant kernel = the measured leader of this run's per-launch HIP-event times (no name is pinned: conv_stream<256,128>
    # and the fused attention trade places between runs, and whichever leads is reported; every kernel's own figures are in
    # `kernels`)
    # (persistent launches of one kernel differ only in their phase count -- "trunk_kernel<conv_stream 256x128, 4 phases>" and
    #  "..., 7 phases>" are the same kernel function: they compete for "dominant" together, and the longest of them is reported)
    fam = lambda k: k.split(",")[0] + ">" if k.startswith("trunk_kernel<") else k
    fam_ms = {}
    for k, v in tot.items():
        fam_ms[fam(k)] = fam_ms.get(fam(k), 0.0) + v["ms"]
    dom_fam = max(fam_ms, key=lambda k: fam_ms[k])
    dom = max((k for k in tot if fam(k) == dom_fam), key=lambda k: tot[k]["ms"])
    d = tot[dom]
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    kernels = {k: {"share": round(v["ms"] / all_ms, 4), "launches_per_batch": v["launches"],
                   "avg_us": round(v["ms"] * 1e3 / v["launches"], 2),
                   "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                   "alg_gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
               for k, v in sorted(tot.items(), key=lambda kv: -kv[1]["ms"])}
    # HBM bytes per launch from the rocprofv3 --pmc passes (tools/collect_profiles.sh), newest round first
    import glob
    import re
    def tkey(f):                                    # round5_traffic.json > round5_v2_traffic.json > round4_traffic.json > ...
        m = re.search(r"round(\d+)(?:_v(\d+))?_traffic", os.path.basename(f))
        return (-int(m.group(1)), -(int(m.group(2)) if m.group(2) else 1 << 20)) if m else (0, 0)
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_traffic.json")), key=tkey)

    def counter_traffic(kname):
        """(bytes per launch, file) of `kname` from the newest traffic file that knows it."""
        for tpath in tfiles:
            try:
                tj = json.load(open(tpath))
                t = tj.get(kname)
                if t is None and kname.startswith("trunk_kernel<"):
                    # persistent launches: the counters know the kernel VARIANT only ("trunk_kernel<conv_stream 256x128>": bytes per
                    # launch averaged over its launches of 4 and 7 phases); this launch's share = its phases / the variant's mean
                    variant = kname.split(",")[0] + ">"
                    phases = lambda k: int(k.split(",")[1].split()[0])
                    same = {k: v for k, v in tot.items() if k.startswith(kname.split(",")[0] + ",")}
                    mean_ph = sum(phases(k) * v["launches"] for k, v in same.items()) / max(1, sum(v["launches"] for v in same.values()))
                    if tj.get(variant) is not None and mean_ph > 0:
                        t = int(tj[variant] * phases(kname) / mean_ph)
                if t is not None:
                    return t, os.path.basename(tpath)
            except (OSError, ValueError, IndexError):
                pass
        return None, None

    traffic, tname = counter_traffic(dom)
    traffic_src = None if traffic is None else (
        f"profiles/{tname}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/collect_profiles.sh), "
        f"not re-measured in this run")
    step_launches = sum(v["launches"] for v in prof.get("unet_step", {}).values())      # what the captured step really launches
    rl = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
          "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
          "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
          "alg_flops_per_launch": d["flops"] / d["launches"], "alg_bytes_per_launch": d["bytes"] / d["launches"],
          "launches_per_batch": d["launches"], "eager_kernel_ms_per_batch": round(all_ms, 3),
          "concurrent_chains": lanes, "chain_batch": prof.get("lane_batch"), "step_launches": step_launches}
    # the kernel that costs the step most: share of the kernel time x distance from ITS roofline (MFMA for the contractions, HBM for
    # the elementwise launches -- whichever fraction is larger)
    def frac_of(v):
        sec = v["ms"] * 1e-3
        return max(v["flops"] / sec / 1e12 / PEAK_BF16_TFLOPS, v["bytes"] / sec / 1e9 / PEAK_HBM_GBS)
    worst = max(tot, key=lambda k: tot[k]["ms"] / all_ms * (1.0 - min(1.0, frac_of(tot[k]))))
    w = tot[worst]
    wsec = w["ms"] * 1e-3
    w_mfma = w["flops"] / wsec / 1e12 / PEAK_BF16_TFLOPS >= w["bytes"] / wsec / 1e9 / PEAK_HBM_GBS
    w_traffic, w_tname = counter_traffic(worst)
    w_alg = w["bytes"] / w["launches"]
    rl_worst = {"kernel": worst, "share": round(w["ms"] / all_ms, 4), "bound": "mfma" if w_mfma else "hbm",
                "achieved": round(w["flops"] / wsec / 1e12 if w_mfma else w["bytes"] / wsec / 1e9, 2),
                "peak": PEAK_BF16_TFLOPS if w_mfma else PEAK_HBM_GBS, "unit": "TFLOP/s" if w_mfma else "GB/s",
                "frac": round(frac_of(w), 4), "score_share_x_distance": round(w["ms"] / all_ms * (1.0 - min(1.0, frac_of(w))), 4),
                "avg_launch_us": round(w["ms"] * 1e3 / w["launches"], 2), "launches_per_batch": w["launches"],
                "alg_flops_per_launch": w["flops"] / w["launches"], "alg_bytes_per_launch": w_alg,
                "traffic": w_traffic, "traffic_ratio": None if not w_traffic or not w_alg else round(w_traffic / w_alg, 2),
                "traffic_source": None if w_traffic is None else f"profiles/{w_tname}"}
    rl["fell_back"] = bool(prof.get("fell_back", False))
    rl["plan_flags"] = prof.get("plan_flags")
    return rl, kernels, rl_worst


def other_configs(seed, dev):
    """The BASELINE configs the headline line does not cover, each as a short leg on ONE GPU AFTER (outside) the headline's timed
    region: (3) nuScenes at its whole batch of 32 and at its per-GPU share on 8 GPUs (4 images), (4) conditional up-sampling at
    batch 16, (1) pixel-space RangeDM at batch 1 / 10 steps, (5) the training step at 8 samples.  Same pipelines, synthetic weights,
    inputs resident in HBM; every leg: build + one warm call (graph capture), then `iters` timed calls between synchronisations.
    frac_of_peak = algorithmic FLOPs / time / the dense bf16 MFMA peak."""
    from rangeldm_amd.pipelines import LDMPipelineRange, DDIMPipelineRange, LDMUpscalePipelineRange
    from rangeldm_amd.schedulers import DDIMSchedulerHIP
    from rangeldm_amd.synth import latent_noise, sparse_range_condition
    legs = []

    def sample_leg(name, preset, B, S, iters):
        t_leg = time.perf_counter()
        p, unet, vae, _, _ = build_models(preset, seed)
        sched = DDIMSchedulerHIP()
        cond_enc, conds = None, None
        lat_shape = (p["unet"].out_channels, *p["unet"].sample_size)
        if p["cond_channels"] == 8:
            from rangeldm_amd.encoders import SparseRangeImageEncoder2
            pipe = LDMUpscalePipelineRange(vae=vae, unet=unet, scheduler=sched)
            cond_enc = SparseRangeImageEncoder2()
            W4, H4 = p["unet"].sample_size
            conds = torch.from_numpy(np.stack([sparse_range_condition(seed, j, (2, 4 * W4, H4)) for j in range(B)])).to(dev)
        elif vae is not None:
            pipe = LDMPipelineRange(vae=vae, unet=unet, scheduler=sched, pos_encoding=p["pos_encoding"])
        else:
            pipe = DDIMPipelineRange(unet=unet, scheduler=sched, pos_encoding=p["pos_encoding"])
        x = torch.from_numpy(np.stack([latent_noise(seed, j, lat_shape) for j in range(B)])).to(dev)
        kw = dict(batch_size=B, num_inference_steps=S, latents=x, output_type="torch", check=False)
        if conds is not None:
            kw.update(image=conds, condition_encoder=cond_enc)
        out = pipe(**kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = pipe(**kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        pipe._fused.status_all()
        assert torch.isfinite(out).all()
        gflop = (S * unet.flops(B) + (vae.decode_flops(B, *lat_shape[1:]) if vae else 0.0)) / B / 1e9
        legs.append({"workload": name, "value": round(B / dt, 2), "unit": "range-images/sec", "ms": round(dt * 1e3, 3),
                     "frac_of_peak": round(B / dt * gflop / 1e3 / PEAK_BF16_TFLOPS, 4), "leg_seconds": round(time.perf_counter() - t_leg, 2)})
        del pipe, unet, vae

    def train_leg(iters):
        from rangeldm_amd.config import PRESETS
        from rangeldm_amd.params import unet_param_shapes, vae_param_shapes
        from rangeldm_amd.schedulers import DDPMSchedulerHIP
        from rangeldm_amd.synth import synth_state_dict, normal
        from rangeldm_amd.training import UNetTrainer, training_step
        from rangeldm_amd.vae import AutoencoderKLHIP
        t_leg = time.perf_counter()
        p = PRESETS["RangeLDM"]
        tr = UNetTrainer(p["unet"], synth_state_dict(unet_param_shapes(p["unet"]), seed=seed), device=dev)
        vae = AutoencoderKLHIP(p["vae"])
        vae.load_state_dict(synth_state_dict(vae_param_shapes(p["vae"]), seed=seed, prefix="vae."))
        sched = DDPMSchedulerHIP()
        B = 8
        gen = torch.Generator().manual_seed(seed)
        imgs = [torch.from_numpy(normal(seed, f"train/0/{i}", (B, 2, 1024, 64))).mul_(0.5).to(dev) for i in range(2 + iters)]
        for i in range(2):                               # (step 1 runs eagerly and sizes the scratch, step 2 captures the graphs)
            training_step(tr, vae, sched, imgs[i], generator=gen, pos_encoding=True, graphed=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2, 2 + iters):
            loss = training_step(tr, vae, sched, imgs[i], generator=gen, pos_encoding=True, graphed=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        gflop = 3 * 34.071 + 75.73
        assert np.isfinite(float(loss))
        legs.append({"workload": "config 5 per-GPU share: train_unconditional step (VAE encode + UNet fwd + bwd + AdamW + EMA), batch 8",
                     "value": round(B / dt, 1), "unit": "samples/sec", "ms": round(dt * 1e3, 3),
                     "frac_of_peak": round(B / dt * gflop / 1e3 / PEAK_BF16_TFLOPS, 4), "leg_seconds": round(time.perf_counter() - t_leg, 2)})

    for fn, a in ((sample_leg, ("config 3 on one GPU: nuScenes 32x1024, 50-step DDIM, batch 32", "nuscenes", 32, 50, 5)),
                  (sample_leg, ("config 3 per-GPU share at 8 GPUs: nuScenes, 50-step DDIM, batch 4", "nuscenes", 4, 50, 8)),
                  (sample_leg, ("config 4: conditional up-sampling 16 -> 64 beams, 50-step DDIM, batch 16", "upsample", 16, 50, 5)),
                  (sample_leg, ("config 1: RangeDM 64x1024 pixel space, 10-step DDIM, batch 1", "RangeDM", 1, 10, 10)),
                  (train_leg, (10,))):
        try:
            fn(*a)
        except Exception as e:                           # a leg that fails is reported, not hidden; the headline line still prints
            legs.append({"workload": str(a[0]), "error": repr(e)[:300]})
        torch.cuda.empty_cache()
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step (weak) / global batch (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch images on EVERY GPU; strong: ONE global batch of --batch images sharded over the GPUs "
                         "(BASELINE config 3: --preset nuscenes --batch 32 --scaling strong)")
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--sampler", choices=["ddim", "ddpm"], default="ddim")
    ap.add_argument("--preset", default="RangeLDM")
    ap.add_argument("--seed", type=int, default=20240310)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the secondary three-requests-in-flight measurement")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` block (BASELINE configs 1, 3, 4, 5 as short legs after the headline measurement)")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE config 5 instead of the headline: the data-parallel training step (tools/bench_train.py, same "
                         "--gpus / --steps / --warmup contract, metric training samples/sec)")
    args, rest = ap.parse_known_args()
    if args.train:
        sys.argv = [os.path.join(ROOT, "tools", "bench_train.py"), "--gpus", str(args.gpus), "--steps", str(args.steps),
                    "--warmup", str(args.warmup)] + rest
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_train
        return bench_train.main()
    if rest:
        ap.error("unrecognized arguments: " + " ".join(rest))

    from rangeldm_amd import distributed as D
    from rangeldm_amd.pipelines import LDMPipelineRange, DDIMPipelineRange, LDMUpscalePipelineRange
    from rangeldm_amd.schedulers import DDIMSchedulerHIP, DDPMSchedulerHIP
    from rangeldm_amd.synth import latent_noise, step_noise, sparse_range_condition

    rank, world, local = D.init_from_env("nccl")
    assert world == args.gpus or world == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    # With N > 1 ranks the finished images travel through the C-ABI RCCL communicator (rldm_comm_*).  If it cannot be bound or
    # bootstrapped on ANY rank, all ranks learn it in the bootstrap's agreement rounds and all of them use torch.distributed's
    # all-gather instead (backend "nccl" = the same RCCL) -- never silently: the reason is printed and the line's `comm` block says
    # `"collective": "torch"` + `"cabi_unavailable": "<reason>"`.  The first multi-GPU run of a round happens on the driver's node,
    # unobserved: a measured, labelled record beats an aborted one.  RLDM_REQUIRE_CABI=1 turns the fall-back into an error.
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    p, unet, vae, usd, vsd = build_models(args.preset, args.seed)
    sched = DDIMSchedulerHIP() if args.sampler == "ddim" else DDPMSchedulerHIP()
    cond_enc = None
    if p["cond_channels"] == 8:                          # BASELINE config 4: up-sampling 16 -> 64 beams (ldm/configs/upsample.yaml)
        from rangeldm_amd.encoders import SparseRangeImageEncoder2
        pipe = LDMUpscalePipelineRange(vae=vae, unet=unet, scheduler=sched)
        cond_enc = SparseRangeImageEncoder2()
    elif p["cond_channels"]:
        ap.error(f"--preset {args.preset}: no synthetic condition for this preset")
    elif vae is not None:
        pipe = LDMPipelineRange(vae=vae, unet=unet, scheduler=sched, pos_encoding=p["pos_encoding"])
    else:
        pipe = DDIMPipelineRange(unet=unet, scheduler=sched, pos_encoding=p["pos_encoding"])
    S = args.inference_steps
    strong = args.scaling == "strong"
    if strong:
        if args.batch % world:
            ap.error(f"--scaling strong needs --batch ({args.batch}) divisible by the number of GPUs ({world})")
        lo, hi = D.shard_range(args.batch, rank, world)      # this rank's contiguous slice of every global batch
        B = hi - lo
    else:
        B = args.batch
    lat_shape = (p["unet"].out_channels, *p["unet"].sample_size)
    n_iter = args.warmup + args.steps

    # inputs resident in HBM before the timed region; x_T is a function of the GLOBAL sample index
    xs = []
    for i in range(n_iter):
        idx = ([i * args.batch + j for j in range(lo, hi)] if strong else D.global_sample_indices(i, B, rank, world))
        xs.append(torch.from_numpy(np.stack([latent_noise(args.seed, j, lat_shape) for j in idx])).to(dev))
    zs = None
    if args.sampler == "ddpm":
        first = lo if strong else 0
        zs = torch.from_numpy(np.stack([np.stack([step_noise(args.seed, first + j, s, lat_shape) for j in range(B)])
                                        for s in range(S)])).to(dev)

    conds = None
    if cond_enc is not None:                             # the low-resolution images, resident in HBM like x_T
        W4, H4 = p["unet"].sample_size
        conds = [torch.from_numpy(np.stack([sparse_range_condition(args.seed, (i * B + j) if not strong else j,
                                                                    (2, 4 * W4, H4)) for j in range(B)])).to(dev)
                 for i in range(n_iter)]

    def one_step(i):
        # check=False: the timed loop stays asynchronous; the (sticky) self-check word of the persistent launches is read once,
        # behind the loop's final synchronisation (pipe._fused.status_all() below)
        kw = dict(batch_size=B, num_inference_steps=S, latents=xs[i], output_type="torch", check=False)
        if zs is not None:
            kw["step_noise"] = zs
        if conds is not None:
            kw.update(image=conds[i], condition_encoder=cond_enc)
        img = pipe(**kw)
        return D.all_gather_images(img)

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_iter):
        out = one_step(i)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, dev)
    pipe._fused.status_all()                             # raises if any call of the loop tripped the persistent launches' self-check
    assert torch.isfinite(out).all()
    if rank == 0 and os.environ.get("RLDM_BENCH_DUMP"):  # (tests: the gathered images of the last step, for the N-rank == 1-rank check)
        torch.save(out.detach().cpu(), os.environ["RLDM_BENCH_DUMP"])
    # the exchange step on its own (it is inside the timed region too): one all-gather of a finished batch, HIP events on the
    # stream it is issued on; and the proof that `world` ranks met -- every rank's id through the same collective path
    comm = D.comm_info(dev)
    gather_ms = 0.0
    if world > 1:
        img_l = out[rank * B:(rank + 1) * B].contiguous()
        D.all_gather_images(img_l)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        D.barrier()
        e0.record()
        for _ in range(5):
            D.all_gather_images(img_l)
        e1.record()
        torch.cuda.synchronize()
        gather_ms = D.max_over_ranks(e0.elapsed_time(e1) / 5, dev)
    assert sorted(comm["ranks_seen"]) == list(range(world)), comm

    if rank == 0:
        total_images = world * B * args.steps
        res = {
            "metric": ("range-images/sec, KITTI-360 64x1024 50-step DDIM @ batch16, 1/2/4/8 GPU" if args.preset == "RangeLDM"
                       else f"range-images/sec, {args.preset} {S}-step {args.sampler.upper()}"),
            "value": total_images / dt, "unit": "range-images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.preset} (latent {lat_shape[0]}x{lat_shape[1]}x{lat_shape[2]}), "
                                   f"{S}-step {args.sampler.upper()}" + (" + VAE decode (4x)" if vae is not None else "") +
                                   (f", global batch {args.batch} sharded {B} per GPU" if strong else f", batch {B} per GPU") +
                                   ", synthetic weights, x_T resident in HBM",
                       "global_batch": B * world, "batch_per_gpu": B, "inference_steps": S, "sampler": args.sampler,
                       "parallelism": f"sample-sharded x{world}, RCCL all-gather of finished images"},
            # N ranks really met: world / ranks_seen from a collective over the images' own path, the RCCL communicator's view
            "comm": dict(comm, allgather_ms=round(gather_ms, 4), allgather_bytes_per_rank=int(B * out[0].numel() * 4)),
        }
        if True:                                        # (per-GPU figures, measured on rank 0's GPU for every N)
            h = pipe._fused.get(unet, vae, sched, B, S, 0 if args.sampler == "ddim" else 1, p["pos_encoding"], p["cond_channels"])
            rl, kernels, rl_worst = roofline(pipe, h, xs[0], S)
            gflop_per_image = (S * unet.flops(B) + (vae.decode_flops(B, *lat_shape[1:]) if vae else 0.0)) / B / 1e9
            res["roofline"] = rl
            res["roofline_worst"] = rl_worst
            # did any persistent launch fail its self-check (the sampler then rebuilt itself as one launch per layer: correct images,
            # 4-5 % slower)?  `plan_flags` = the routing bits in effect at the end of the timed loop, 0 = the defaults
            res["fell_back"] = rl.pop("fell_back")
            res["plan_flags"] = rl.pop("plan_flags")
            res["kernels"] = kernels
            res["gflop_per_image"] = round(gflop_per_image, 1)
            res["end_to_end_tflops"] = round(res["value"] * gflop_per_image / 1e3, 1)
            res["end_to_end_frac_of_mfma_peak"] = round(res["value"] / world * gflop_per_image / 1e3 / PEAK_BF16_TFLOPS, 4)
            res["unet_launches_per_step"] = rl.pop("step_launches") or unet.num_launches(B)
            # ---- self-calibration (outside the timed region; `value` untouched) ------------------------------------------------
            # graph_over_eager: the batch's wall time (captured graphs, launches back to back) over the sum of its kernels' HIP-event
            # times enqueued one by one -- < 1: the graph hides launch gaps; it moves with the box (0.87 - 0.92 seen)
            res["graph_over_eager"] = round(res["ms_per_step"] / rl["eager_kernel_ms_per_batch"], 4)
            try:
                kwc = dict(batch_size=B, num_inference_steps=S, latents=xs[0], output_type="torch", check=False)
                if zs is not None:
                    kwc["step_noise"] = zs
                if conds is not None:
                    kwc.update(image=conds[0], condition_encoder=cond_enc)
                pipe(**kwc)
                wclk, nxcd = workload_clock_mhz(lambda: [pipe(**kwc) for _ in range(3)], dev)
                cal = box_calibration(dev)
                cal["workload_clock_mhz"] = wclk
                cal["workload_clock_xcds_seen"] = nxcd
                cal["nominal_clock_mhz"] = NOMINAL_CLOCK_MHZ
                res["calibration"] = cal
                if wclk:
                    # the dominant launch against the peak AT THE CLOCK THE CHIP RAN THE WORKLOAD AT (peak x clock / nominal)
                    rl["frac_clock_adjusted"] = round(rl["achieved"] / (PEAK_BF16_TFLOPS * wclk / NOMINAL_CLOCK_MHZ), 4)
                    rl["workload_clock_mhz"] = wclk
                # `value` per calibration TFLOP/s: the figure that should agree between boxes running the same build
                res["value_per_calibration_tflops"] = round(res["value"] / world / cal["mfma_tflops"], 5) if cal["mfma_tflops"] else None
                # ... and per GHz of the clock the chip held under THIS workload (agrees within ~1 % between boxes)
                res["value_per_workload_ghz"] = round(res["value"] / world / (wclk / 1000.0), 2) if wclk else None
            except Exception as e:                      # noqa: BLE001  (secondary figures: never fail the line)
                res["calibration"] = {"error": str(e)[:200]}
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(p, usd, vsd, B, S)
                if args.preset == "RangeLDM" and not strong:
                    res["cpu_baseline_c1"] = cpu_baseline_c1(args.seed, threads=res["cpu_baseline"]["threads"])
            if world == 1 and not args.no_pipelined and args.preset == "RangeLDM" and B == 16 and zs is None and conds is None:
                # NOT `value`: the same kernels with THREE batch-16 requests in flight (three chains of 16 on separate HIP streams,
                # one pipeline call of 48 samples) -- what a throughput driver that does not wait for batch i before it starts
                # batch i + 1 gets out of the chip.  `value` above stays the one-request-at-a-time figure (ldm/inference.py:159-183).
                nreq = 3
                xp = [torch.cat([xs[(i * nreq + j) % n_iter] for j in range(nreq)]) for i in range(3)]
                runp = lambda i: pipe(batch_size=nreq * B, num_inference_steps=S, latents=xp[i], output_type="torch")
                runp(0)
                torch.cuda.synchronize()
                tp0 = time.perf_counter()
                for i in (1, 2):
                    outp = runp(i)
                torch.cuda.synchronize()
                dtp = time.perf_counter() - tp0
                res["pipelined"] = {"requests_in_flight": nreq, "request_batch": B, "value": round(2 * nreq * B / dtp, 2),
                                    "unit": "range-images/sec", "ms_per_request": round(dtp / (2 * nreq) * 1e3, 3),
                                    "note": "secondary figure; `value` is one request at a time"}
                assert torch.isfinite(outp).all()
            if (world == 1 and not args.no_other_configs and args.preset == "RangeLDM" and B == 16 and zs is None and conds is None
                    and not strong):
                del pipe
                torch.cuda.empty_cache()
                t_oc = time.perf_counter()
                res["other_configs"] = other_configs(args.seed, dev)
                res["other_configs_seconds"] = round(time.perf_counter() - t_oc, 1)
                try:                                    # (after everything timed; its stamp launches must not touch `value`)
                    ig = in_graph_launch_us(args.preset, args.seed, B, rl["kernel"], dev)
                    if ig and ig[0]:
                        ach = rl["alg_flops_per_launch"] / (ig[0] * 1e-6) / 1e12
                        rl["in_graph"] = {"avg_launch_us": round(ig[0], 2), "achieved": round(ach, 1), "frac": round(ach / rl["peak"], 4),
                                          "unet_forward_us": round(ig[1], 1), "launches": ig[2],
                                          "note": "the same kernel timed inside the sampler's captured step graph (device real-time "
                                                  "counter between launches, tools/graph_trace.py); `achieved` / `frac` above are the "
                                                  "HIP-event figures of launches enqueued one by one"}
                except Exception as e:                  # noqa: BLE001  (a secondary figure: never fail the line)
                    rl["in_graph"] = {"error": str(e)[:200]}
        print(json.dumps(res), flush=True)
    D.barrier()
    D.close()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
