#!/usr/bin/env python3
"""A1111 / Open-WebUI compatible image API over the MI355X pipelines — the HTTP surface of the
reference's flux_app.py (routes and JSON shapes of :47-62,:90-321; server flags of :786-793).
The Gradio UI and the MusicGen tab are out of scope; the Darwin/arm64 gate is replaced by a ROCm check.

More than one GPU:  `torchrun --nproc-per-node N --master-addr 127.0.0.1 flux_app.py [--port P]`.  Rank 0 serves HTTP; ranks
1..N-1 sit in `worker_loop`.  A request's `batch_size * n_iter` images (the reference generates them as one batch,
flux_app.py:123-204) are sharded by image over the N ranks: rank 0 broadcasts the request, every rank builds / reuses the same
pipeline, the pipelines' own sharded `generate_latents` (rank-0 text conditioning broadcast over RCCL, same-seed prior slice,
no collective inside a step) and `gather_images` (uint8 to rank 0) do the rest - the path `torchrun txt2image.py` takes."""
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


class RequestFailedOnAllRanks(RuntimeError):
    """A sharded request failed at a point every rank of the job knows about (agreed through `_all_ok`, or carried in a
    broadcast header): the ranks are still in step, the server keeps serving."""


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
            if _dist_world()[1] > 1:
                # rank 0 of a torchrun job: hand the request to the worker ranks (they block in worker_loop), then run it too
                req = dict(prompt=prompt, model=model, width=width, height=height, steps=steps, guidance=guidance, seed=seed,
                           batch_size=batch_size, n_iter=n_iter)
                _bcast_obj(req)
                return self._generate_sharded(return_pil=return_pil, **req)
            return self._generate_images(prompt, model, width, height, steps, guidance, seed, batch_size, n_iter, return_pil)

    def _generate_sharded(self, prompt, model, width, height, steps, guidance, seed, batch_size, n_iter, return_pil=False):
        """One request on every rank of the job (rank 0: called from the route; ranks > 0: from worker_loop).  Returns the
        encoded images on rank 0, None elsewhere.  A rank that cannot build the pipeline (missing checkpoint ...) must not
        leave the others waiting inside a collective: the outcome of init_pipeline is agreed on first."""
        import torch
        from flux_generator_amd.parallel import CollectiveStepFailed
        rank, world = _dist_world()
        if self._desync:
            raise RuntimeError("the job's ranks are out of step after an earlier failure inside a collective; restart the job")
        err = None
        try:
            pipe = self.init_pipeline(model)
        except Exception as e:                   # noqa: BLE001 - reported to every rank below
            err, pipe = e, None
        if not _all_ok(err is None):
            raise RequestFailedOnAllRanks(str(err) if err is not None else "another rank of the job could not build the pipeline")
        # Fail-stop for the rest of the request.  The ranks meet in collectives at three places: the conditioning broadcast inside
        # generate_latents (a failure of the rank-0-only text towers travels in the broadcast header and is raised on every rank,
        # parallel.shard_generation_inputs), nowhere inside the denoise steps or the decode, and the uint8 gather.  A rank-local
        # failure between two of them (an OOM in a decode ...) would strand the others in the next one, so the outcome of every
        # rank-local stretch is AGREED ON (`_all_ok`) before the next collective is entered; an exception that is not an agreed
        # one - raised by a collective itself - leaves the ranks out of step, and then the job stops (`_fatal`).
        try:
            out = self._sharded_body(pipe, prompt, model, width, height, steps, guidance, seed, batch_size, n_iter)
        except (RequestFailedOnAllRanks, CollectiveStepFailed):
            raise
        except Exception as e:                   # noqa: BLE001
            if getattr(e, "_agreed", False):     # raised at the same point on every rank (see _sharded_body)
                raise RequestFailedOnAllRanks(str(e)) from e
            self._fatal(e)
            raise
        if out is None:
            return None
        return self._encode(out, return_pil)

    _desync = False

    def _fatal(self, e) -> None:
        """An exception out of a collective (timeout, peer gone): the ranks no longer agree on where they are.  Refuse further
        sharded requests and end this process shortly (the HTTP 500 of the current request still goes out on rank 0); torchrun
        tears the other ranks down with it."""
        import os
        self._desync = True
        print(f"[flux_app rank {_dist_world()[0]}] fatal: {type(e).__name__}: {e} - leaving the job", file=sys.stderr, flush=True)
        if _dist_world()[1] > 1 and os.environ.get("FLUX_APP_NO_EXIT") != "1":
            threading.Timer(1.0, lambda: os._exit(70)).start()

    def _sharded_body(self, pipe, prompt, model, width, height, steps, guidance, seed, batch_size, n_iter):
        """The request on this rank; returns the list of uint8 images on rank 0, None elsewhere."""
        import torch
        n = batch_size * n_iter
        latent_size = (height // 8, width // 8)
        sd = model.startswith("stabilityai/")
        if sd:
            steps = steps or (2 if "sdxl-turbo" in model else 50)
            guidance = guidance or (0.0 if "sdxl-turbo" in model else 7.5)
            latents = pipe.generate_latents(prompt, n_images=n, cfg_weight=guidance, num_steps=steps, seed=seed)
            out_hw = (512, 512)                  # (the reference does not pass latent_size on this route: always 64 x 64 latents)
        else:
            steps = steps or (50 if model == "flux-dev" else 2)
            latents = pipe.generate_latents(prompt, n_images=n, num_steps=steps, latent_size=latent_size,
                                            guidance=guidance, seed=seed)
            next(latents)                        # conditioning tuple (computed on rank 0, broadcast)
            out_hw = (latent_size[0] * 8, latent_size[1] * 8)
        err, local = None, None
        try:                                     # rank-local stretch: denoise steps + decode (no collective inside)
            x_t = None
            for x_t in latents:                  # this rank's rows of the batch (pipe.shard); possibly none when n < world
                pass
            imgs = [pipe.decode(x_t[i:i + self.DECODE_BATCH]) if sd else pipe.decode(x_t[i:i + self.DECODE_BATCH], latent_size)
                    for i in range(0, len(x_t), self.DECODE_BATCH)]
            local = torch.cat(imgs, dim=0) if imgs else torch.empty(0, *out_hw, 3, device=x_t.device)
            if local.is_cuda:
                torch.cuda.synchronize(local.device)     # an asynchronous fault of this stretch surfaces here, not inside the gather
        except Exception as e:                   # noqa: BLE001 - agreed on below
            err = e
        if not _all_ok(err is None):             # agreement point in front of the gather
            e = err if err is not None else RuntimeError("another rank of the job failed in its denoise / decode")
            e._agreed = True
            raise e
        allimg = pipe.gather_images(local, n)    # uint8 [n, H, W, 3] on rank 0 (batch order), None elsewhere
        if allimg is None:
            return None
        return list(allimg.cpu().numpy())

    @staticmethod
    def _encode(arrs, return_pil):
        import numpy as np
        from PIL import Image
        out = []
        for arr in arrs:
            pil = Image.fromarray(np.asarray(arr))
            if return_pil:
                out.append(pil)
            else:
                buf = io.BytesIO()
                pil.save(buf, format="PNG")
                out.append(base64.b64encode(buf.getvalue()).decode())
        return out

    def _generate_images(self, prompt, model, width, height, steps, guidance, seed, batch_size, n_iter, return_pil):
        import torch
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
        # decode in batches (the reference decodes one image at a time, flux_app.py:186-192): up to DECODE_BATCH latents per
        # VAE pass, one device->host copy per batch; float -> uint8 by truncation like the reference
        arrs = []
        for i in range(0, n, self.DECODE_BATCH):
            chunk = x_t[i:i + self.DECODE_BATCH]
            img = pipe.decode(chunk) if sd else pipe.decode(chunk, latent_size)
            arrs.extend((img * 255).to(torch.uint8).cpu().numpy())
        return self._encode(arrs, return_pil)

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
        # "sd_backend" is the reference's literal (Open-WebUI reads it); "sd_device" is an addition: one process drives ONE GPU
        # (libfluxhip binds one device per process); under `torchrun flux_app.py` the job's ranks share every request
        return {"sd_model_checkpoint": "stabilityai/stable-diffusion-2-1-base", "sd_backend": "Flux MLX",
                "sd_device": f"{_dist_world()[1]} x MI355X (one process per GPU; a request's images are sharded over them)",
                "sd_model_list": [dict(title=_MODELS[i][1], name=_MODELS[i][0], model_name=_MODELS[i][0]) for i in order]}

    def set_options(self, options: dict):
        return {"success": True}

    def get_progress(self):
        return {"progress": 0, "eta_relative": 0,
                "state": {"skipped": False, "interrupted": False, "job": "", "job_count": 0, "job_timestamp": ""},
                "current_image": None, "textinfo": "Idle"}


def _dist_world():
    """(rank, world size) of the torch.distributed job this server belongs to; (0, 1) for a plain `python flux_app.py`."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:            # pragma: no cover
        pass
    return 0, 1


def _bcast_obj(obj, src: int = 0):
    """Rank `src` passes the object, the other ranks pass None and receive it (a request dict; None = shut down)."""
    import torch.distributed as dist
    box = [obj]
    if dist.get_backend() == "nccl":       # RCCL moves device memory: the pickled bytes travel through the rank's GPU
        import torch
        dist.broadcast_object_list(box, src=src, device=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.broadcast_object_list(box, src=src)
    return box[0]


def _all_ok(ok: bool) -> bool:
    """True iff every rank reports ok (one tiny all-reduce; keeps a failing rank from desynchronising the job)."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(t)
    return int(t.item()) == 0


def worker_loop(api_: "FluxAPI") -> int:
    """Ranks 1..N-1 of `torchrun flux_app.py`: wait for rank 0 to broadcast a request, run this rank's share of it, repeat
    until rank 0 broadcasts None (server shutdown).  Returns the number of requests served."""
    served = 0
    while True:
        req = _bcast_obj(None)
        if req is None:
            return served
        from flux_generator_amd.parallel import CollectiveStepFailed
        try:
            api_._generate_sharded(**req)
        except (RequestFailedOnAllRanks, CollectiveStepFailed) as e:
            # failed at an agreed point: rank 0 reports it to the client (HTTP 500), every rank is back at the top of its loop
            print(f"[flux_app worker rank {_dist_world()[0]}] request failed: {e}", file=sys.stderr)
        # anything else left a collective half-done: _generate_sharded has called _fatal, the exception ends this rank (fail-stop)
        served += 1


def shutdown_workers() -> None:
    rank, world = _dist_world()
    if rank == 0 and world > 1:
        _bcast_obj(None)


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
        import os
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1:                  # torchrun: one process per GPU, rank 0 serves, the others work
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
            torch.cuda.set_device(device)
            if not dist.is_initialized():
                import datetime
                # explicit collective timeout: a rank that died between two agreement points must not hang its peers for
                # the backend's default (10-30 min); FLUX_APP_COLLECTIVE_TIMEOUT_S overrides
                dist.init_process_group("nccl", device_id=torch.device(device),
                                        timeout=datetime.timedelta(seconds=int(os.environ.get("FLUX_APP_COLLECTIVE_TIMEOUT_S", "300"))))
            if dist.get_rank() != 0:
                n = worker_loop(api)
                print(f"[flux_app worker rank {dist.get_rank()}] served {n} requests, exiting")
                dist.destroy_process_group()
                return
        host = "0.0.0.0" if args.listen_all else "127.0.0.1"
        port = args.port if check_port_available(host, args.port) else find_available_port(host, args.port)
        if port != args.port:
            print(f"Warning: Port {args.port} is in use, using port {port} instead")
        print(f"Starting Flux server on {host}:{port}")
        import uvicorn
        try:
            uvicorn.Server(uvicorn.Config(get_app(), host=host, port=port, log_level="info")).run()
        finally:
            shutdown_workers()       # (no-op for a single process) the worker ranks leave their loop
    except SystemError as e:
        print(f"Error: {e}")
        sys.exit(1)


if __name__ == "__main__":
    main()
