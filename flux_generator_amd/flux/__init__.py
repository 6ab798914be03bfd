"""Drop-in for the reference's ``flux`` package (flux/__init__.py): same public names."""
from .autoencoder import AutoEncoder, AutoEncoderParams
from .flux import FluxPipeline
from .model import Flux, FluxParams
from .sampler import FluxSampler
from .utils import configs, load_ae, load_flow_model

__all__ = ["FluxPipeline", "Flux", "FluxParams", "FluxSampler", "AutoEncoder", "AutoEncoderParams", "configs",
           "load_ae", "load_flow_model"]
