"""Drop-in alias: `from stable_diffusion import StableDiffusion, StableDiffusionXL` resolves to the MI355X
implementation (flux_generator_amd.stable_diffusion)."""
from flux_generator_amd.stable_diffusion import StableDiffusion, StableDiffusionXL  # noqa: F401
