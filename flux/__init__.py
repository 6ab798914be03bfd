"""Drop-in alias: `from flux import FluxPipeline` resolves to the MI355X implementation
(flux_generator_amd.flux), so the reference's callers (txt2image.py, flux_app.py) keep their imports."""
from flux_generator_amd.flux import *  # noqa: F401,F403
from flux_generator_amd.flux import (AutoEncoder, AutoEncoderParams, Flux, FluxParams, FluxPipeline, FluxSampler, configs,
                                     load_ae, load_flow_model)
from flux_generator_amd.flux.utils import load_clip, load_clip_tokenizer, load_t5, load_t5_tokenizer  # noqa: F401
