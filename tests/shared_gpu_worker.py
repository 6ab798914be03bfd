"""Worker of tests/test_configs_gpu.py::test_two_default_mode_processes_share_one_gpu: one of several INDEPENDENT processes
that run the full-size Flux-schnell forward (19 + 38 blocks, C2 shape: L = 1024, S = 256, batch 1; seed-0 random init) on the
same GPU at the same time, with the library's DEFAULT split-K hand-off (reduce-scatter).  Neither process owns the GPU, so
the 240-block split-K grids of the two are not resident as a whole: the hand-off's bounded poll + orphan completion has to
carry them (include/fluxhip.h, fluxhip_gemm_set_splitk_mode).

`matmul` as the tag starts a co-tenant of plain torch.matmul kernels instead (it runs until <sync_dir>/done exists).

usage: python tests/shared_gpu_worker.py <tag> <n_forwards> <sync_dir> <n_procs> <result.pt>"""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def matmul_cotenant(sync_dir, n_procs):
    """A co-tenant that is NOT this library: dependent torch.matmul / gelu kernels (hipBLASLt) until the workers are done."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    Ws = [(torch.randn(4096, 4096, generator=g) / 64).to(torch.bfloat16).to(dev) for _ in range(8)]
    x0 = torch.randn(2048, 4096, generator=g).to(torch.bfloat16).to(dev)
    open(os.path.join(sync_dir, "ready_matmul"), "w").close()
    t0 = time.time()
    while not os.path.exists(os.path.join(sync_dir, "done")) and time.time() - t0 < 600:
        x = x0
        for W in Ws * 6:
            x = torch.nn.functional.gelu(x @ W) + 0.5
        torch.cuda.synchronize()


def main():
    if sys.argv[1] == "matmul":
        return matmul_cotenant(sys.argv[3], int(sys.argv[4]))
    tag, n_fwd, sync_dir, n_procs, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    import torch
    warnings.simplefilter("ignore")
    assert os.environ.get("FLUXHIP_SPLITK") is None
    from flux_generator_amd import _lib
    from flux_generator_amd.flux.model import Flux
    from flux_generator_amd.flux.utils import configs
    dev = torch.device("cuda:0")
    BF = torch.bfloat16
    P = configs["flux-schnell"].params
    model = Flux(P, device=dev).init_random(0)
    g = torch.Generator().manual_seed(3)
    B, S, L = 1, 256, 1024
    img = torch.randn(B, L, 64, generator=g).to(BF).to(dev)
    txt = (torch.randn(B, S, P.context_in_dim, generator=g) * 0.5).to(BF).to(dev)
    vec = torch.randn(B, P.vec_in_dim, generator=g).to(BF).to(dev)
    ii, jj = torch.meshgrid(torch.arange(32, dtype=torch.int32), torch.arange(32, dtype=torch.int32), indexing="ij")
    img_ids = torch.stack([torch.zeros_like(ii), ii, jj], dim=-1).reshape(1, L, 3).to(dev)
    txt_ids = torch.zeros(B, S, 3, dtype=torch.int32, device=dev)
    t = torch.full((B,), 0.5, dtype=BF, device=dev)
    model(img, img_ids, txt, txt_ids, t, vec)                 # warm-up: workspaces, kernel attributes
    torch.cuda.synchronize()
    # start together: every process drops a file and waits for the others'
    open(os.path.join(sync_dir, f"ready_{tag}"), "w").close()
    t0 = time.time()
    while len([f for f in os.listdir(sync_dir) if f.startswith("ready_")]) < n_procs:
        if time.time() - t0 > 600:
            raise RuntimeError("peers never became ready")
        time.sleep(0.01)
    lib = _lib.load()
    n0 = int(lib.fluxhip_gemm_rs_launches())
    t0 = time.perf_counter()
    preds = []
    x = img
    for i in range(n_fwd):                                    # a dependent chain: pred feeds the next input
        pred = model(x, img_ids, txt, txt_ids, t, vec)
        x = (img + 0.25 * pred).to(BF)
        preds.append(pred)
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    torch.save({"tag": tag, "last": preds[-1].cpu(), "first": preds[0].cpu(), "seconds": secs,
                "rs_launches": int(lib.fluxhip_gemm_rs_launches()) - n0,
                "finite": bool(torch.isfinite(preds[-1]).all())}, out)


if __name__ == "__main__":
    main()
