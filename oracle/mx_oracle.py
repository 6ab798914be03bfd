"""CPU restatement of the block-scaled ("MX") fp8 format of the fp8 path — TEST INFRASTRUCTURE ONLY (tests/, smoke() and bench.py's
cpu_baseline may import it; the product path never does).

The reference's `--quantize` (txt2image.py:26-28,79-82) calls mlx.nn.quantize, a group-wise affine weight quantiser; this build's
fp8 path replaces it with OCP e4m3 operands on the CDNA4 block-scaled MFMA.  The format restated here is this
build's own (include/fluxhip.h, fluxhip_fp8_mx), after the OCP Microscaling spec (MXFP8: e4m3 elements, one E8M0 scale per 32):
    scale of a block = 2^e, e the smallest integer with max|v| / 2^e <= 448;  byte = e + 127 clamped to [1, 253]
    elements         = round-to-nearest-even(v / 2^e) in e4m3fn (torch.float8_e4m3fn)
Parity unpinned by the reference (no such arithmetic exists there); pinned by its own known-answer tests in tests/test_oracle_mx.py.
"""
import numpy as np
import torch

E4M3_MAX = 448.0


def mx_quantize(x: torch.Tensor):
    """x float32 [rows, K] (K % 32 == 0) -> (q uint8 [rows, K], e8 uint8 [rows, K // 32])."""
    x = x.detach().to(torch.float32).cpu()
    rows, K = x.shape
    assert K % 32 == 0
    blk = x.reshape(rows, K // 32, 32)
    am = blk.abs().amax(dim=-1).numpy().astype(np.float32)
    bits = am.view(np.uint32).astype(np.int64)
    e8 = (bits >> 23) - 8 + ((bits & 0x7FFFFF) > 0x600000)          # 448 = 1.75 * 2^8
    e8 = np.clip(e8, 1, 253)
    mul = np.ldexp(np.float32(1.0), (127 - e8).astype(np.int32)).astype(np.float32)
    q = (blk * torch.from_numpy(mul)[..., None]).to(torch.float8_e4m3fn).view(torch.uint8).reshape(rows, K)
    return q, torch.from_numpy(e8.astype(np.uint8))


def mx_dequantize(q: torch.Tensor, e8: torch.Tensor) -> torch.Tensor:
    """(q uint8 [rows, K], e8 uint8 [rows, K // 32]) -> float64 [rows, K]."""
    rows, K = q.shape
    v = q.cpu().view(torch.float8_e4m3fn).to(torch.float64).reshape(rows, K // 32, 32)
    sc = torch.from_numpy(np.ldexp(1.0, e8.cpu().numpy().astype(np.int32) - 127))
    return (v * sc[..., None]).reshape(rows, K)


def mx_tile_index(row: np.ndarray, kb: np.ndarray, kstride: int) -> np.ndarray:
    """Byte offset of the scale of (scale-buffer row, 32-column block kb) in the tiled layout the GEMM's K loop reads:
    one dword per (K-step of 128, 64-row group, K block in the step, row mod 16), byte = which 16-row fragment of the group."""
    row = np.asarray(row, dtype=np.int64)
    kb = np.asarray(kb, dtype=np.int64)
    return ((((kb >> 2) * kstride + (row >> 6) * 64 + (kb & 3) * 16 + (row & 15)) << 2) + ((row >> 4) & 3))


def mx_untile(mx: torch.Tensor, rows: int, K: int, row0: int = 0, kstride: int = None) -> torch.Tensor:
    """Tiled scale bytes -> e8 [rows, K // 32] (row r of the result = scale-buffer row row0 + r)."""
    kstride = kstride or (row0 + rows)
    r = np.arange(rows)[:, None] + row0
    kb = np.arange(K // 32)[None, :]
    return torch.from_numpy(mx.cpu().numpy()[mx_tile_index(r, kb, kstride)])


def mx_tile(e8: torch.Tensor, kstride: int = None, row0: int = 0) -> torch.Tensor:
    """e8 [rows, K // 32] -> tiled scale bytes (uint8 [(K / 128) * kstride * 4], untouched bytes = 127)."""
    rows, nb = e8.shape
    kstride = kstride or (row0 + rows)
    out = np.full(((nb + 3) // 4) * kstride * 4, 127, dtype=np.uint8)
    r = np.arange(rows)[:, None] + row0
    kb = np.arange(nb)[None, :]
    out[mx_tile_index(r, kb, kstride)] = e8.cpu().numpy()
    return torch.from_numpy(out)
