"""CPU: the C-ABI library loads and exports every symbol include/grappa_hip.h declares (no compute calls: no GPU here),
and the ctypes binding table covers exactly that set."""
import ctypes
import os
import re

from grappa_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "grappa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(grappa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/grappa_hip.h but not exported by libgrappa_hip.so"
    assert sorted(_lib.SIGNATURES.keys()) == names


def test_host_only_entry_points():
    lib = _lib.load()
    assert lib.grappa_abi_version() == _lib.ABI_VERSION == 11
    assert lib.grappa_build_arch() == b"gfx950"
    # pure host helpers: workspace queries and the dropout hash
    assert lib.grappa_gemm_f32_workspace_bytes(512, 512, 100000) > 0
    assert lib.grappa_gemm_f32_workspace_bytes(65536, 2048, 512) == 65536 * 64 * 4      # 8192 tiles = 32 per CU: no split, no tail -- only the row-maxima segments (out_amax)
    assert lib.grappa_layernorm_bwd_workspace_bytes(1000, 512) >= 250 * 2 * 512 * 4
    from oracle.ops_ref import dropout_keep
    import torch
    idx = torch.arange(0, 2000, dtype=torch.int64)
    for seed in (0, 12345, 2 ** 63 - 1):
        want = dropout_keep(seed, idx, 0.3)
        got = torch.tensor([lib.grappa_dropout_keep(seed, int(i), 0.3) for i in idx], dtype=torch.bool)
        assert torch.equal(got, want)


def test_product_has_no_cpu_fallback():
    """without a GPU the product backend must refuse to start (no silent eager/PyTorch path)."""
    import pytest
    import torch
    from grappa_amd import backend
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    old = backend._BACKEND
    backend.set_backend(None)
    try:
        with pytest.raises(RuntimeError):
            backend.get_backend()
    finally:
        backend.set_backend(old)
    # and the product package never imports the oracle
    import subprocess, sys
    code = ("import sys, grappa_amd, grappa_amd.ops, grappa_amd.datasets, grappa_amd.optim, grappa_amd.dist, grappa_amd.trainer, grappa_amd.pdb, "
            "grappa_amd.moldata, grappa_amd.dataloader; assert not any(m.startswith('oracle') for m in sys.modules)")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_device_code_has_no_packed_fp32_instructions():
    """csrc/Makefile compiles the kernels without v_pk_{fma,mul,add}_f32 (2 % faster, and the instruction the
    multi-queue deviation needs): every gfx950 code object inside the shared library is disassembled and searched"""
    import re
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    data = open(_lib.LIB_PATH, "rb").read()
    objects = packed = 0
    for m in re.finditer(b"\x7fELF", data):
        blob = data[m.start():]
        if m.start() == 0 or blob[18:20] != b"\xe0\x00":          # e_machine 224 = EM_AMDGPU
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            out = subprocess.run([objdump, "-d", f.name], capture_output=True, text=True).stdout
        objects += 1
        packed += len(re.findall(r"v_pk_(?:fma|mul|add)_f32", out))
    assert objects >= 10, objects
    assert packed == 0, packed


def test_kernels_keep_the_register_budgets_their_occupancy_needs():
    """two workgroups per CU are part of these kernels' design: the one-plane bf16 product (512 threads: <= 128
    registers), the pair-format products (256 threads: <= 256) -- and none of the product kernels spills.  Read from the code objects'
    metadata (a shared epilogue change that costs ten registers otherwise shows up as a 15 % slower bf16 configuration, as in round 3)."""
    import re
    import subprocess
    import tempfile
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not found")
    data = open(_lib.LIB_PATH, "rb").read()
    kernels = {}
    for m in re.finditer(b"\x7fELF", data):
        blob = data[m.start():]
        if m.start() == 0 or blob[18:20] != b"\xe0\x00":
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            out = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in out.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
            sp = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            if name and vg:
                kernels[name.group(1)] = (int(vg.group(1)), int(sp.group(1)) if sp else 0)
    planes1 = {k: v for k, v in kernels.items() if "gemm_planes_kernelILi1E" in k}
    pairs = {k: v for k, v in kernels.items() if "gemm_pairs_kernel" in k or "gemm_wpairs_kernel" in k}
    # the kernels of the default arithmetic (mode 103 = two fp16 pieces), of the bf16 storage configuration and of the pair format
    hot = {k: v for k, v in kernels.items() if re.search(r"gemm_bf16x_(grouped_)?kernelILi512ELi103E", k)}
    hot.update(planes1)
    hot.update(pairs)
    assert len(planes1) == 2 and len(pairs) >= 3 and len(hot) >= 19, (len(planes1), len(pairs), len(hot))
    assert all(v[0] <= 128 for v in planes1.values()), planes1
    assert all(v[0] <= 256 for v in pairs.values()), pairs
    assert all(v[1] == 0 for v in hot.values()), {k: v for k, v in hot.items() if v[1]}
