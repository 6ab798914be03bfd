#!/usr/bin/env python3
"""Flux text-to-image CLI on MI355X — same flags and outputs as the reference's txt2image.py
(positional prompt; --model --n-images --image-size --steps --guidance --n-rows --decoding-batch-size
--quantize/-q --preload-models --output --save-raw --seed --verbose/-v --adapter --fuse-adapter
--no-t5-padding; reference txt2image.py:43-65).  Runs on the HIP device only: there is no CPU path."""
import argparse
import os
import sys

import torch
from PIL import Image

from flux import FluxPipeline


def to_latent_size(image_size):
    """Round each side UP to a multiple of 16 px, return the latent size (side / 8)."""
    h, w = (((s + 15) // 16) * 16 for s in image_size)
    if (h, w) != tuple(image_size):
        print(f"Warning: The image dimensions need to be divisible by 16px. Changing size to {h}x{w}.")
    return (h // 8, w // 8)


def build_parser():
    p = argparse.ArgumentParser(description="Generate images from a textual prompt using Flux on MI355X")
    p.add_argument("prompt")
    p.add_argument("--model", choices=["schnell", "dev"], default="schnell")
    p.add_argument("--n-images", type=int, default=4)
    p.add_argument("--image-size", type=lambda x: tuple(map(int, x.split("x"))), default=(512, 512))
    p.add_argument("--steps", type=int, help="Number of steps (min: 1, default: 2 for schnell, 50 for dev)")
    p.add_argument("--guidance", type=float, default=4.0)
    p.add_argument("--n-rows", type=int, default=1)
    p.add_argument("--decoding-batch-size", type=int, default=1)
    p.add_argument("--quantize", "-q", action="store_true",
                   help="fp8 (e4m3) weights + fp8 activations (per-token scales out of LayerNorm, block scales out of attention / GELU) on the fp8 matrix cores for the Linears of the flow "
                        "transformer's blocks (99.6%% of its FLOPs), of the T5 encoder and of CLIP (the layers the reference's "
                        "in_dim %% 512 predicate selects)")
    p.add_argument("--preload-models", action="store_true")
    p.add_argument("--output", default="out.png")
    p.add_argument("--save-raw", action="store_true")
    p.add_argument("--seed", type=int)
    p.add_argument("--verbose", "-v", action="store_true")
    p.add_argument("--adapter")
    p.add_argument("--fuse-adapter", action="store_true",
                   help="fold the adapter into the weights at load time (W + scale*B^T A^T, one bf16 rounding); without it the "
                        "low-rank branches stay separate, like the reference's LoRALinear layers")
    p.add_argument("--no-t5-padding", dest="t5_padding", action="store_false")
    return p


def peak_gb(dev):
    v = torch.cuda.max_memory_allocated(dev) / 1024 ** 3
    torch.cuda.reset_peak_memory_stats(dev)
    return v


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.steps is not None and args.steps < 1:
        parser.error("Number of steps must be at least 1")
    args.steps = args.steps or (50 if args.model == "dev" else 2)
    if not torch.cuda.is_available():
        sys.exit("txt2image.py needs an MI355X (HIP) device: the denoise/decode path has no CPU fallback")

    # `torchrun --nproc-per-node N txt2image.py ...`: one process per GPU, the --n-images batch is sharded by image
    # (rank 0 encodes the prompt and broadcasts txt / vec over RCCL, rank 0 saves the gathered images)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = "cuda"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
        torch.cuda.set_device(device)
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device(device))
    flux = FluxPipeline("flux-" + args.model, t5_padding=args.t5_padding, device=device)
    dev = flux.device
    if args.adapter:
        n = flux.load_adapter(args.adapter, fuse=args.fuse_adapter)
        how = "folded into the weights"
        if not args.fuse_adapter:
            al = getattr(flux, "adapter_layers", None) or dict(branches=n, folded=0)
            how = (f"{al['branches']} separate low-rank branches, like the unfused LoRALinear"
                   + (f"; {al['folded']} modulation Linears folded into the modulation table" if al["folded"] else ""))
            if args.quantize:
                how += "; --quantize folds the branches before quantising (the fp8 plan carries no separate branch)"
        print(f"Applied LoRA adapter {args.adapter} to {n} layers ({how})", file=sys.stderr)
    if args.quantize:
        # the reference's nn.quantize (txt2image.py:79-82) re-designed for CDNA4: e4m3 weights (per output channel) and
        # e4m3 activations (per token / block-scaled) of the transformer blocks' Linears on the fp8 matrix cores
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # (the folding of unfused branches is announced above)
            flux.flow.enable_fp8()
        flux.quantize_text()
        print("--quantize: fp8 e4m3 Linears in the flow transformer's blocks, the T5 encoder (all but the value projection) and "
              "CLIP (second MLP Linear: the reference's in_dim % 512 predicate); the VAE keeps float32", file=sys.stderr)
    if args.preload_models:
        flux.ensure_models_are_loaded()

    latent_size = to_latent_size(args.image_size)
    latents = flux.generate_latents(args.prompt, n_images=args.n_images, num_steps=args.steps, latent_size=latent_size,
                                    guidance=args.guidance, seed=args.seed)
    next(latents)                                   # conditioning
    torch.cuda.synchronize(dev)
    mem_text = peak_gb(dev)
    x_t = None
    for i, x_t in enumerate(latents):
        torch.cuda.synchronize(dev)
        print(f"step {i + 1}/{args.steps}", file=sys.stderr)
    mem_gen = peak_gb(dev)

    decoded = [flux.decode(x_t[i:i + args.decoding_batch_size], latent_size)
               for i in range(0, len(x_t), args.decoding_batch_size)]
    torch.cuda.synchronize(dev)
    mem_dec = peak_gb(dev)
    x = (torch.cat(decoded, dim=0) if decoded else
         torch.empty(0, latent_size[0] * 8, latent_size[1] * 8, 3, device=dev))
    x = flux.gather_images(x, args.n_images)        # uint8 (truncation, like the reference); all images on rank 0
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if x is None:                                   # ranks > 0 are done
        return

    if args.save_raw:
        *name, suffix = args.output.split(".")
        stem = ".".join(name)
        arr = x.cpu().numpy()
        for i in range(len(arr)):
            Image.fromarray(arr[i]).save(".".join([stem, str(i), suffix]))
    else:
        x = torch.nn.functional.pad(x, (0, 0, 4, 4, 4, 4))      # 4 px border around every image
        B, H, W, C = x.shape
        rows = args.n_rows
        x = x.reshape(rows, B // rows, H, W, C).permute(0, 2, 1, 3, 4).reshape(rows * H, B // rows * W, C)
        Image.fromarray(x.cpu().numpy()).save(args.output)

    if args.verbose:
        print(f"Peak memory used for the text:       {mem_text:.3f}GB")
        print(f"Peak memory used for the generation: {mem_gen:.3f}GB")
        print(f"Peak memory used for the decoding:   {mem_dec:.3f}GB")
        print(f"Peak memory used overall:            {max(mem_text, mem_gen, mem_dec):.3f}GB")


if __name__ == "__main__":
    main()
