"""flux_generator_amd — MI355X (gfx950) native denoise/decode hot path behind the reference's
``flux`` package interface (FluxPipeline / Flux / FluxSampler / AutoEncoder).

The compute path is libfluxhip.so (hand-written HIP, C ABI in include/fluxhip.h); this package is
the Python host side that mirrors the reference's operator interface.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
