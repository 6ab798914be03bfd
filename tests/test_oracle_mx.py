"""Known-answer tests of the block-scaled fp8 restatement (oracle/mx_oracle.py): scale choice at the e4m3 boundary, round trip
within the e4m3 step, the tiled scale layout is a bijection with the dword structure the GEMM's lanes read."""
import numpy as np
import torch

from oracle.mx_oracle import mx_dequantize, mx_quantize, mx_tile, mx_tile_index, mx_untile


def test_scale_choice_known_answers():
    x = torch.zeros(6, 32)
    x[0, 3] = 448.0          # exactly the e4m3 maximum: scale 2^0
    x[1, 3] = 449.0          # just above: scale 2^1
    x[2, 5] = -0.875         # 1.75 * 2^-1 -> 448 * 2^-9
    x[3, 7] = 1.0            # 1.0 = 256 * 2^-8 -> scale 2^-8
    x[4, 0] = 3.0e-3
    q, e8 = mx_quantize(x)
    assert e8[:4, 0].tolist() == [127, 128, 127 - 9, 127 - 8]
    assert e8[5, 0] == 1 and int(q[5].sum()) == 0                    # all-zero block
    d = mx_dequantize(q, e8)
    assert d[0, 3] == 448.0 and d[1, 3] == 448.0                     # 449 / 2 = 224.5 rounds to 224 (step 16 in [128, 256))
    assert d[2, 5] == -0.875 and d[3, 7] == 1.0
    # nothing saturates: the largest element of every block lands in (224, 448]
    g = torch.Generator().manual_seed(0)
    y = torch.randn(64, 256, generator=g) * torch.logspace(-6, 6, 64)[:, None]
    q, e8 = mx_quantize(y)
    top = q.view(torch.float8_e4m3fn).float().abs().reshape(64, 8, 32).amax(-1)
    assert float(top.min()) > 224.0 - 16.0 and float(top.max()) <= 448.0
    err = (mx_dequantize(q, e8) - y.double()).abs().reshape(64, 8, 32)
    bound = y.abs().reshape(64, 8, 32).amax(-1, keepdim=True).double() * 2.0 ** -4      # half a step of the top binade
    assert bool((err <= bound).all())


def test_tiled_scale_layout_is_a_bijection():
    rows, K, ks = 192, 512, 256
    r = np.arange(rows)[:, None] + 64
    kb = np.arange(K // 32)[None, :]
    idx = mx_tile_index(r, kb, ks)
    assert len(np.unique(idx)) == idx.size and idx.max() < (K // 128) * ks * 4
    # the four fragments (rows r, r+16, r+32, r+48 of a 64-row group) of one (K block, row mod 16) share a dword
    assert (idx[0:16] >> 2 == idx[16:32] >> 2).all() and (idx[16:32] & 3 == 1).all()
    # lane (q4 * 16 + r16) of a wave reads consecutive dwords
    base = idx[0, 0] >> 2
    lanes = np.array([[mx_tile_index(64 + r16, q4, ks) >> 2 for r16 in range(16)] for q4 in range(4)]).reshape(-1)
    assert (lanes == base + np.arange(64)).all()
    e8 = torch.randint(1, 254, (rows, K // 32), dtype=torch.uint8)
    assert torch.equal(mx_untile(mx_tile(e8, ks, 64), rows, K, 64, ks), e8)
