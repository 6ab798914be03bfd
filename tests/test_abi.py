"""The C-ABI library loads and exports every symbol include/fluxhip.h declares (no compute calls:
this runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fluxhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fluxhip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from flux_generator_amd import _lib
    assert _lib.LIB_PATH.exists(), "build with `python -c 'import __graft_entry__ as g; g.build()'`"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fluxhip.h but not exported"
    assert set(_lib.SIGNATURES) == set(names), "ctypes binding table and header disagree"


def test_loader_checks_abi_and_arch():
    from flux_generator_amd import _lib
    lib = _lib.load()
    assert lib.fluxhip_abi_version() == 9 and lib.fluxhip_arch() == b"gfx950"


def test_gemm_desc_layout_matches_header():
    """sizeof/offsets of the ctypes mirror = the C struct (LP64): 2 x 104-byte groups + 80 bytes."""
    from flux_generator_amd._lib import GemmDesc, GemmGroup
    assert ctypes.sizeof(GemmGroup) == 104 and GemmGroup.M.offset == 80 and GemmGroup.add.offset == 88
    assert ctypes.sizeof(GemmDesc) == 288 and GemmDesc.C2.offset == 248 and GemmDesc.alpha.offset == 272
    assert GemmDesc.ld_add.offset == 280
    from flux_generator_amd._lib import GemmX3Desc      # fluxhip_gemm_x3_desc: 5 pointers, 7 int64, 10 int32, float, pad
    from flux_generator_amd._lib import Fp8Mx
    assert ctypes.sizeof(Fp8Mx) == 96 and Fp8Mx.c8.offset == 32 and Fp8Mx.ldc8.offset == 56 and Fp8Mx.c_mx.offset == 64
    assert ctypes.sizeof(GemmX3Desc) == 144 and GemmX3Desc.a_lo.offset == 40 and GemmX3Desc.M.offset == 96
    assert GemmX3Desc.alpha.offset == 136


def test_tile_picker_is_host_only():
    """fluxhip_gemm_tile_cfg does no device work, so it can be exercised on CPU: Flux shapes at
    T = 1280 pick the tiles the sweep in profiles/ found best."""
    from flux_generator_amd import _lib, ops
    lib = _lib.load()

    def cfg(groups, N, K):
        d = ops.make_gemm_desc([dict(A=1, W=1, C=1, M=m) for m in groups], 1, N, K, K, N)
        return lib.fluxhip_gemm_tile_cfg(ctypes.byref(d))

    bm, bn, th = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    for groups, N, K, want in (([1280], 21504, 3072, (256, 224)), ([256, 1024], 9216, 3072, (256, 192)),
                               ([256, 1024], 12288, 3072, (256, 256)), ([1280], 3072, 15360, (128, 128))):
        c = cfg(groups, N, K)                   # tile cfg | split-K factor << 8
        assert lib.fluxhip_gemm_tile_shape(c, bm, bn, th) == 0
        import torch
        if torch.cuda.is_available():           # with a GPU the loader attaches the split-K workspace and the N = 3072 shape splits
            continue
        assert (bm.value, bn.value) == want, (groups, N, K, c)
        assert c >> 8 == 1, "no split-K without a workspace (none is attached on a host without a GPU)"


def test_product_path_has_no_oracle_or_cpu_fallback():
    """The package never imports the oracle, and its ops refuse CPU tensors."""
    import subprocess, sys
    pk = os.path.join(ROOT, "flux_generator_amd")
    out = subprocess.run(["grep", "-rIl", "--include=*.py", "-E", r"^\s*(from|import)\s+oracle", pk],
                         capture_output=True, text=True).stdout.strip()
    assert out == "", f"product code imports the oracle: {out}"
    import pytest, torch
    from flux_generator_amd import ops
    with pytest.raises(ops.FluxHipError):
        ops.linear(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))
    from flux_generator_amd.flux.model import Flux, FluxParams
    with pytest.raises(ops.FluxHipError):
        Flux(FluxParams(64, 64, 128, 256, 4.0, 2, 1, 1, [16, 56, 56], 10000, True, False), device="cpu")


def test_integration_stub_matches_the_header():
    """INTEGRATION.md's ctypes stub is executed as written (the library path comes from FLUXHIP_LIB, as the stub says) and
    every struct it declares must have the size and the field offsets of the loader's mirror of include/fluxhip.h; the
    ABI version its prose quotes must be the header's.  A maintainer pasting a stale stub would hand under-sized structs
    to fluxhip_gemm_bf16 (round-4 review: the doc had drifted two ABI versions behind)."""
    from flux_generator_amd import _lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# fluxhip_binding\.py.*?)```", doc, flags=re.S)
    assert m, "INTEGRATION.md lost its binding stub"
    hdr = open(os.path.join(ROOT, "include", "fluxhip.h")).read()
    abi = int(re.search(r"#define\s+FLUXHIP_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert re.search(rf"`FLUXHIP_ABI_VERSION` is {abi}\b", doc), "INTEGRATION.md quotes another ABI version than the header"
    old = os.environ.get("FLUXHIP_LIB")
    os.environ["FLUXHIP_LIB"] = str(_lib.LIB_PATH)
    ns = {}
    try:
        exec(compile(m.group(1), "INTEGRATION.md:stub", "exec"), ns)      # loads the library, asserts its ABI, declares the structs
    finally:
        if old is None:
            os.environ.pop("FLUXHIP_LIB", None)
        else:
            os.environ["FLUXHIP_LIB"] = old
    for name in ("GemmGroup", "GemmDesc"):
        doc_t, lib_t = ns[name], getattr(_lib, name)
        assert ctypes.sizeof(doc_t) == ctypes.sizeof(lib_t), name
        assert [f[0] for f in doc_t._fields_] == [f[0] for f in lib_t._fields_], name
        for f in doc_t._fields_:
            assert getattr(doc_t, f[0]).offset == getattr(lib_t, f[0]).offset, (name, f[0])
            assert getattr(doc_t, f[0]).size == getattr(lib_t, f[0]).size, (name, f[0])
    # the argtypes the stub sets are the loader's
    for fn in ("fluxhip_gemm_bf16", "fluxhip_attention_d128_bf16"):
        got = getattr(ns["_lib"], fn).argtypes
        want = _lib.SIGNATURES[fn][1]
        assert len(got) == len(want), fn
        for a, b in zip(got, want):
            assert ctypes.sizeof(a) == ctypes.sizeof(b), fn
    # the header's structs themselves, compiled: sizeof from the C compiler = the ctypes mirror
    import shutil, subprocess, tempfile
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc:
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "sz.c")
            open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "fluxhip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                                 'sizeof(fluxhip_gemm_group),sizeof(fluxhip_gemm_desc),offsetof(fluxhip_gemm_group,add),'
                                 'offsetof(fluxhip_gemm_desc,ld_add),sizeof(fluxhip_fp8_mx));return 0;}\n')
            exe = os.path.join(td, "sz")
            subprocess.check_call([cc, "-I", os.path.join(ROOT, "include"), src, "-o", exe])
            out = subprocess.check_output([exe], text=True).split()
        assert [int(v) for v in out] == [ctypes.sizeof(_lib.GemmGroup), ctypes.sizeof(_lib.GemmDesc), _lib.GemmGroup.add.offset,
                                         _lib.GemmDesc.ld_add.offset, ctypes.sizeof(_lib.Fp8Mx)]
