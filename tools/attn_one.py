"""One Flux-shaped attention workload (B=1, 24 heads, T=1280), 6 launches: the target of `rocprofv3 --pmc ...` passes."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from flux_generator_amd import ops
B, H, T = 1, 24, 1280
q = torch.randn(B, H, T, 128, device="cuda").to(torch.bfloat16)
k = torch.randn(B, H, T, 128, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, H, 128, T, device="cuda").to(torch.bfloat16)
o = torch.empty(B, T, H * 128, dtype=torch.bfloat16, device="cuda")
for _ in range(6): ops.attention_d128(q, k, vt, o, H * 128, B, H, T, T, 128 ** -0.5)
torch.cuda.synchronize()
