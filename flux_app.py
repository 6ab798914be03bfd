#!/usr/bin/env python3
"""A1111 / Open-WebUI compatible image API over the MI355X pipelines — the HTTP surface of the
reference's flux_app.py (routes and JSON shapes of :47-62,:90-321; server flags of :786-793).
The Gradio UI and the MusicGen tab are out of scope; the Darwin/arm64 gate is replaced by a ROCm check."""
import argparse
import base64
import io
import socket
import sys
import threading
from typing import List, Optional, Tuple

from fastapi import FastAPI, HTTPException
from fastapi.middleware.cors import CORSMiddleware
from pydantic import BaseModel

_MODELS = (("flux-schnell", "Flux Schnell (Fast)", "flux-schnell.safetensors"),
           ("flux-dev", "Flux Dev (High Quality)", "flux-dev.safetensors"),
           ("stabilityai/stable-diffusion-2-1-base", "SD 2.1 Base (High Quality)", "sd-2-1-base.safetensors"),
           ("stabilityai/sdxl-turbo", "SDXL Turbo (Fast)", "sdxl-turbo.safetensors"))


class SDAPIRequest(BaseModel):
    prompt: str
    negative_prompt: Optional[str] = None
    width: int = 512
    height: int = 512
    steps: Optional[int] = None
    cfg_scale: float = 4.0
    batch_size: int = 1
    n_iter: int = 1
    seed: int = -1
    model: str = "schnell"


class SDAPIResponse(BaseModel):
    images: List[str]
    parameters: dict
    info: str


class FluxAPI:
    """One pipeline cache shared by the HTTP routes and direct callers."""

    def __init__(self):
        # libfluxhip is single-stream per process (one split-K workspace, include/fluxhip.h): requests are serialised
        self._lock = threading.Lock()
        self.pipeline = None
        self.sd_pipeline = None
        self.current_model = None

    def init_pipeline(self, model: str):
        if model.startswith("stabilityai/"):
            if self.sd_pipeline is None or self.current_model != model:
                from stable_diffusion import StableDiffusion, StableDiffusionXL
                self.sd_pipeline = (StableDiffusionXL("stabilityai/sdxl-turbo", float16=True) if "sdxl-turbo" in model
                                    else StableDiffusion("stabilityai/stable-diffusion-2-1-base", float16=True))
                self.current_model = model
            return self.sd_pipeline
        name = model if model.startswith("flux-") else f"flux-{model}"
        if self.pipeline is None or self.current_model != name:
            from flux import FluxPipeline
            self.pipeline = FluxPipeline(name)
            self.current_model = name
        return self.pipeline

    DECODE_BATCH = 4

    def generate_images(self, prompt: str, model: str = "schnell", width: int = 512, height: int = 512,
                        steps: Optional[int] = None, guidance: float = 4.0, seed: Optional[int] = None,
                        batch_size: int = 1, n_iter: int = 1, return_pil: bool = False):
        with self._lock:
            return self._generate_images(prompt, model, width, height, steps, guidance, seed, batch_size, n_iter, return_pil)

    def _generate_images(self, prompt, model, width, height, steps, guidance, seed, batch_size, n_iter, return_pil):
        import numpy as np
        import torch
        from PIL import Image
        pipe = self.init_pipeline(model)
        n = batch_size * n_iter
        latent_size = (height // 8, width // 8)
        sd = model.startswith("stabilityai/")
        if sd:
            steps = steps or (2 if "sdxl-turbo" in model else 50)
            guidance = guidance or (0.0 if "sdxl-turbo" in model else 7.5)
            latents = pipe.generate_latents(prompt, n_images=n, cfg_weight=guidance, num_steps=steps, seed=seed)
        else:
            steps = steps or (50 if model == "flux-dev" else 2)
            latents = pipe.generate_latents(prompt, n_images=n, num_steps=steps, latent_size=latent_size,
                                            guidance=guidance, seed=seed)
            next(latents)                       # conditioning tuple
        x_t = None
        for x_t in latents:
            pass
        out = []
        # decode in batches (the reference decodes one image at a time, flux_app.py:186-192): up to DECODE_BATCH latents per
        # VAE pass, one device->host copy per batch; float -> uint8 by truncation like the reference
        arrs = []
        for i in range(0, n, self.DECODE_BATCH):
            chunk = x_t[i:i + self.DECODE_BATCH]
            img = pipe.decode(chunk) if sd else pipe.decode(chunk, latent_size)
            arrs.extend((img * 255).to(torch.uint8).cpu().numpy())
        for arr in arrs:
            pil = Image.fromarray(np.asarray(arr))
            if return_pil:
                out.append(pil)
            else:
                buf = io.BytesIO()
                pil.save(buf, format="PNG")
                out.append(base64.b64encode(buf.getvalue()).decode())
        return out

    async def txt2img(self, request: SDAPIRequest) -> SDAPIResponse:
        try:
            images = self.generate_images(prompt=request.prompt, model=request.model, width=request.width,
                                          height=request.height, steps=request.steps, guidance=request.cfg_scale,
                                          seed=request.seed if request.seed >= 0 else None,
                                          batch_size=request.batch_size, n_iter=request.n_iter, return_pil=False)
        except Exception as e:  # the reference maps every failure to HTTP 500 + str(e)
            raise HTTPException(status_code=500, detail=str(e))
        params = request.model_dump() if hasattr(request, "model_dump") else request.dict()
        keep = ("prompt", "negative_prompt", "width", "height", "steps", "cfg_scale", "seed", "model")
        return SDAPIResponse(images=images, parameters={k: params[k] for k in keep},
                             info=f"Generated with Flux {request.model} model")

    def list_models(self):
        return [dict(title=t, name=n, model_name=t, hash=None, sha256=None, filename=f, config=None) for t, n, f in _MODELS]

    def get_options(self):
        order = (0, 2, 1, 3)
        # "sd_backend" is the reference's literal (Open-WebUI reads it); "sd_device" is an addition: this server process drives
        # ONE GPU (libfluxhip binds one device per process); batches over several GPUs go through `torchrun txt2image.py`
        return {"sd_model_checkpoint": "stabilityai/stable-diffusion-2-1-base", "sd_backend": "Flux MLX",
                "sd_device": "1 x MI355X per server process",
                "sd_model_list": [dict(title=_MODELS[i][1], name=_MODELS[i][0], model_name=_MODELS[i][0]) for i in order]}

    def set_options(self, options: dict):
        return {"success": True}

    def get_progress(self):
        return {"progress": 0, "eta_relative": 0,
                "state": {"skipped": False, "interrupted": False, "job": "", "job_count": 0, "job_timestamp": ""},
                "current_image": None, "textinfo": "Idle"}


api = FluxAPI()


def create_api(app: FastAPI) -> FluxAPI:
    @app.post("/sdapi/v1/txt2img")
    async def txt2img(request: SDAPIRequest):
        return await api.txt2img(request)

    @app.get("/sdapi/v1/sd-models")
    async def list_models():
        return api.list_models()

    @app.get("/sdapi/v1/options")
    async def get_options():
        return api.get_options()

    @app.post("/sdapi/v1/options")
    async def set_options(options: dict):
        return api.set_options(options)

    @app.get("/sdapi/v1/progress")
    async def get_progress():
        return api.get_progress()

    return api


def _with_cors(app: FastAPI) -> FastAPI:
    app.add_middleware(CORSMiddleware, allow_origins=["*"], allow_credentials=True, allow_methods=["*"],
                       allow_headers=["*"])
    return app


def get_app() -> FastAPI:
    app = _with_cors(FastAPI())
    create_api(app)
    return app


app = get_app()


def check_system_compatibility() -> bool:
    """The reference gates on Darwin/arm64 (flux_app.py:323-331); here the gate is a HIP device."""
    import torch
    if not torch.cuda.is_available() or getattr(torch.version, "hip", None) is None:
        raise SystemError("This application requires an AMD Instinct GPU with ROCm (MI355X / gfx950)")
    return True


def to_latent_size(size: Tuple[int, int]) -> Tuple[int, int]:
    h, w = (((s + 15) // 16) * 16 for s in size)
    if (h, w) != tuple(size):
        print(f"Warning: The image dimensions need to be divisible by 16px. Changing size to {h}x{w}.")
    return (h // 8, w // 8)


def check_port_available(host: str, port: int) -> bool:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        try:
            s.bind((host, port))
            return True
        except OSError:
            return False


def find_available_port(host: str, start_port: int, max_attempts: int = 10) -> int:
    for port in range(start_port, start_port + max_attempts):
        if check_port_available(host, port):
            return port
    raise RuntimeError(f"Could not find an available port in range {start_port}-{start_port + max_attempts - 1}")


def generate_images(prompt: str, model: str = "schnell", width: int = 512, height: int = 512, steps: Optional[int] = None,
                    guidance: float = 4.0, seed: Optional[int] = None, batch_size: int = 1, n_iter: int = 1,
                    return_pil: bool = False):
    return FluxAPI().generate_images(prompt=prompt, model=model, width=width, height=height, steps=steps,
                                     guidance=guidance, seed=seed, batch_size=batch_size, n_iter=n_iter,
                                     return_pil=return_pil)


def main():
    try:
        check_system_compatibility()
        parser = argparse.ArgumentParser(description="FLUX Image Generator (MI355X)")
        parser.add_argument("--port", type=int, default=7860, help="Port to run the server on")
        parser.add_mutually_exclusive_group().add_argument("--listen-all", action="store_true",
                                                           help="Listen on all network interfaces (0.0.0.0)")
        args = parser.parse_args()
        host = "0.0.0.0" if args.listen_all else "127.0.0.1"
        port = args.port if check_port_available(host, args.port) else find_available_port(host, args.port)
        if port != args.port:
            print(f"Warning: Port {args.port} is in use, using port {port} instead")
        print(f"Starting Flux server on {host}:{port}")
        import uvicorn
        uvicorn.Server(uvicorn.Config(get_app(), host=host, port=port, log_level="info")).run()
    except SystemError as e:
        print(f"Error: {e}")
        sys.exit(1)


if __name__ == "__main__":
    main()
