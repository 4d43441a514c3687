"""rangeldm_amd -- MI355X-native (gfx950) implementation of the RangeLDM denoising hot path.

Host-side mirror of the reference's duck-typed surface over a C-ABI HIP library
(include/rangeldm_hip.h, rangeldm_amd/csrc).  Importing this package never imports `oracle`.
"""
from .config import UNetConfig, VAEConfig, SchedulerConfig, PRESETS  # noqa: F401

__version__ = "0.1.0"
