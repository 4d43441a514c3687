"""Deterministic synthetic weights and inputs (no dataset / checkpoint is reachable offline).

Every tensor is drawn from a counter-based Philox stream keyed by (seed, crc32(name)), so the CPU oracle, the HIP
library and every rank of a multi-GPU run regenerate bit-identical fp32 values without shipping 120 MB of weights.

  conv / linear weight : U(-1,1) * sqrt(3 / fan_in) * gain      (unit-variance-preserving; gain damps residual growth)
  bias                 : U(-1,1) * 0.02
  GroupNorm weight     : 1 + 0.1 * U(-1,1) ; GroupNorm bias : 0.1 * U(-1,1)
"""
import zlib
import numpy as np

DEFAULT_SEED = 20240310


def _stream(seed: int, name: str):
    return np.random.Generator(np.random.Philox(key=[np.uint64(seed), np.uint64(zlib.crc32(name.encode()))]))


def uniform(seed: int, name: str, shape) -> np.ndarray:
    return (_stream(seed, name).random(size=tuple(shape), dtype=np.float64) * 2.0 - 1.0).astype(np.float32)


def normal(seed: int, name: str, shape) -> np.ndarray:
    return _stream(seed, name).standard_normal(size=tuple(shape), dtype=np.float64).astype(np.float32)


def synth_state_dict(shapes, seed: int = DEFAULT_SEED, prefix: str = "") -> dict:
    """name -> float32 ndarray for every entry of `shapes` (from rangeldm_amd.params)."""
    out = {}
    for name, shape in shapes.items():
        u = uniform(seed, prefix + name, shape)
        leaf = name.rsplit(".", 1)[-1]
        is_norm = ".norm" in name or "group_norm" in name or "conv_norm_out" in name
        if is_norm:
            w = 1.0 + 0.1 * u if leaf == "weight" else 0.1 * u
        elif leaf == "bias":
            w = 0.02 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 0.5 if (".conv2." in name or "to_out" in name) else 1.0
            w = u * np.float32(np.sqrt(3.0 / fan_in) * gain)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def latent_noise(seed: int, sample_index: int, shape) -> np.ndarray:
    """x_T for one global sample index (shape excludes batch) -- identical for any rank/GPU count."""
    return normal(seed, f"x_T/{sample_index}", shape)


def step_noise(seed: int, sample_index: int, step: int, shape) -> np.ndarray:
    """Injected ancestral-DDPM noise for (sample, step)."""
    return normal(seed, f"z/{sample_index}/{step}", shape)


def sparse_range_condition(seed: int, sample_index: int, shape=(2, 1024, 16)) -> np.ndarray:
    """Config-4 condition: ch0 ~ U(-0.5, 2) normalised range, ch1 ~ U(0, 1) intensity."""
    u = uniform(seed, f"cond/{sample_index}", shape)
    out = np.empty(shape, np.float32)
    out[0] = (u[0] * 0.5 + 0.5) * 2.5 - 0.5
    out[1] = u[1] * 0.5 + 0.5
    return out
