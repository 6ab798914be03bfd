"""Worker of tests/test_configs_gpu.py::test_two_ranks_share_one_gpu_end_to_end: one rank of a REAL world-size-2 job whose two
processes share one GPU (gloo carries the collectives — RCCL refuses two ranks on one device —, everything else is the
production path: FluxPipeline.generate_images under a process group, libfluxhip kernels, gather_images).

usage: python tests/dist_gpu_worker.py <rank> <world> <port> <result.pt>"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def apply_tiny_zoo():
    """The small Flux / AE / T5 / CLIP zoo of test_configs_gpu._tiny_flux_zoo, patched process-wide."""
    from flux_generator_amd.flux import utils
    from flux_generator_amd.flux.autoencoder import AutoEncoderParams
    from flux_generator_amd.flux.model import FluxParams
    small = FluxParams(in_channels=64, vec_in_dim=128, context_in_dim=256, hidden_size=256, mlp_ratio=4.0, num_heads=2,
                       depth=2, depth_single_blocks=2, axes_dim=[16, 56, 56], theta=10_000, qkv_bias=True,
                       guidance_embed=False)
    ae = AutoEncoderParams(resolution=256, in_channels=3, ch=128, out_ch=3, ch_mult=[1, 2, 2, 2], num_res_blocks=1,
                           z_channels=16, scale_factor=0.3611, shift_factor=0.1159)
    utils.configs["flux-schnell"] = utils.ModelSpec(params=small, ae_params=ae, ckpt_path=None, ae_path=None, repo_id=None,
                                                    repo_flow=None, repo_ae=None)
    utils.CLIP_L = dict(num_layers=2, model_dims=128, num_heads=2, max_length=77, vocab_size=49408, hidden_act="quick_gelu")
    utils.T5_XXL = dict(vocab_size=32128, num_layers=2, num_heads=4, relative_attention_num_buckets=32, d_kv=64, d_model=256,
                        feed_forward_proj="gated-gelu", tie_word_embeddings=False, d_ff=512)
    os.environ.pop("FLUX_TEXT_DIR", None)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    warnings.simplefilter("ignore")
    apply_tiny_zoo()
    from flux_generator_amd import parallel
    from flux_generator_amd.flux.flux import FluxPipeline
    kw = dict(n_images=5, num_steps=2, latent_size=(16, 16), seed=5, progress=False, reload_text_encoders=False)
    pipe = FluxPipeline("flux-schnell", device="cuda:0")
    want = None
    if rank == 0:                                    # the plain single-process result of the same job, first
        want = pipe.gather_images(pipe.generate_images("two cats", **kw), 5).cpu()
        pipe._t5 = pipe._clip = None                 # (rank 0 rebuilds them on demand below)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    assert parallel.active() and parallel.world() == (rank, world)
    imgs = pipe.generate_images("two cats", **kw)     # 5 images over 2 ranks: 3 + 2
    lo, hi = parallel.shard_range(5, rank, world)
    ok = pipe.shard == (lo, hi) and imgs.shape[0] == hi - lo
    got = pipe.gather_images(imgs, 5)
    if rank == 0:
        ok = ok and got is not None and got.dtype == torch.uint8 and bool(torch.equal(got.cpu(), want))
        ok = ok and float(got.float().std()) > 1.0 and pipe._t5 is not None
    else:
        ok = ok and got is None and pipe._t5 is None and pipe._clip is None      # the text towers were never built here
    dist.barrier()
    torch.save({"rank": rank, "ok": bool(ok), "shard": pipe.shard}, out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
