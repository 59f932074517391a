"""The C-ABI shared library loads without a GPU and exports exactly the entry points include/upnerf_hip.h declares
(no compute calls here).  Also: the binding refuses to work without the library (no fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "upnerf_hip.h")).read()
    # entry points of the diagnostic build (-DUPNERF_STAMPS, libupnerf_hip_stamps.so) are not part of the shipped library
    src = re.sub(r"#ifdef UPNERF_STAMPS.*?#endif", "", src, flags=re.S)
    return sorted(set(re.findall(r"^int\s+(upnerf_\w+)\s*\(", src, flags=re.M)))


def test_header_symbols_are_exported_and_bound():
    from upnerf_amd import _lib
    names = declared()
    assert len(names) >= 15
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), f"{n} declared in include/upnerf_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "ctypes binding and header disagree"
    assert _lib.lib.upnerf_abi_version() == _lib.ABI_VERSION == 10


def test_struct_sizes_match_the_c_layout():
    """ctypes mirrors of the argument structs must have the size the C compiler gives them."""
    import subprocess
    import tempfile
    from upnerf_amd import _lib
    prog = r'''
    #include <stdio.h>
    #include "upnerf_hip.h"
    int main(){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(upnerf_layout), sizeof(upnerf_field_fwd_args),
      sizeof(upnerf_composite_fwd_args), sizeof(upnerf_composite_bwd_args), sizeof(upnerf_field_bwd_args),
      sizeof(upnerf_loss_args), sizeof(upnerf_loss_grads), sizeof(upnerf_frag_desc), sizeof(upnerf_frag16_desc),
      sizeof(upnerf_gather_rays_args), sizeof(upnerf_pack_desc), sizeof(upnerf_wgrad_group), sizeof(upnerf_embed_group),
      sizeof(upnerf_embed_rows_group), sizeof(upnerf_rng), sizeof(upnerf_add_pair)); return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    mine = [ctypes.sizeof(t) for t in (_lib.Layout, _lib.FieldFwdArgs, _lib.CompositeFwdArgs, _lib.CompositeBwdArgs,
                                       _lib.FieldBwdArgs, _lib.LossArgs, _lib.LossGrads, _lib.FragDesc, _lib.Frag16Desc,
                                       _lib.GatherRaysArgs, _lib.PackDesc, _lib.WgradGroup, _lib.EmbedGroup,
                                       _lib.EmbedRowsGroup, _lib.Rng, _lib.AddPair)]
    assert sizes == mine


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import upnerf_amd._lib as L
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        L._load()


def test_argument_errors_are_reported_not_crashed():
    from upnerf_amd import _lib
    assert _lib.lib.upnerf_sort_rows(0, 4, None, None) == -1
    assert _lib.lib.upnerf_linear(4, 4, 7, None, 0, None, 0, None, None, 0, 0, None) == -1
    L = _lib.Layout()
    L.W, L.D, L.skip = 100, 8, 4
    assert _lib.lib.upnerf_field_fwd(ctypes.byref(L), ctypes.byref(_lib.FieldFwdArgs()), None) == -2
    # the register-resident kernels write whole 256-sample tiles: a caller that states how many rows its tensors have is refused
    # when that is less than ceil(M / 256) * 256 (r4 ADVICE) -- before anything is launched
    L.W, L.D, L.skip = 256, 8, 4
    one = ctypes.cast(ctypes.c_void_p(16), ctypes.c_void_p)  # (non-null, never dereferenced on the host)
    fa = _lib.FieldFwdArgs(R=10, S=64, planes=1, tile_rows=256, rays_o=one, rays_d=one, z=one, P=one, P16=one, wexp=one, x0=one,
                           sigma_s=one, wnorm=one, rows_capacity=640)
    assert _lib.lib.upnerf_field_fwd_f16x3(ctypes.byref(L), ctypes.byref(fa), None) == -1  # 640 rows < 768


def test_host_wrappers_refuse_or_fall_back_cleanly_without_a_gpu():
    """Host-side guards of the newer entry points: the sampler has no CPU path (raises), the embedding helper hands CPU
    tables to the module itself, the optimiser factory keeps torch.optim.Adam for CPU parameters."""
    import torch
    from upnerf_amd.ops import embed_rows
    from upnerf_amd.optim import FlatAdam, get_optimizer
    from upnerf_amd.ray_sampler import GpuRaySampler
    with pytest.raises(RuntimeError):
        GpuRaySampler(torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(1, 3, 4), device="cpu")
    emb = torch.nn.Embedding(5, 3)
    idx = torch.tensor([4, 0, 4])
    rows = embed_rows(emb, idx)
    assert torch.equal(rows, emb(idx))
    rows.sum().backward()
    assert emb.weight.grad is not None and float(emb.weight.grad[4].sum()) == 6.0
    lin = torch.nn.Linear(3, 2)
    opt = get_optimizer("adam", 1e-3, [lin])
    assert isinstance(opt, torch.optim.Adam) and not isinstance(opt, FlatAdam)
    with pytest.raises(ValueError):
        FlatAdam(lin.parameters())


def test_no_inline_asm_reads_an_mfma_result_inside_the_hazard_window(tmp_path):
    """hipcc does not pad an MFMA-result -> VALU-read hazard when the reader is an inline-asm statement (stale
    lanes in a few rows per 65 536, different rows every run).  The rule of this code base -- inline asm (v_fma_mix_f32 in
    resid16 / mix16) only reads values that went through a compiler-visible vector instruction first -- is checked on the
    generated ISA: no v_fma_mix_f32 may read a register that a v_mfma wrote within the previous 12 instructions."""
    import re
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "upnerf_amd", "csrc")
    bad = []
    for src in ("field16.hip",):
        out = tmp_path / (src + ".s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                            "--cuda-device-only", os.path.join(csrc, src), "-o", str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        window = []  # (first, last) destination registers of recent v_mfma, one entry per instruction slot
        n_mix = 0
        for line in open(out):
            t = line.strip()
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            op = t.split()[0]
            m = re.match(r"v_mfma\S*\s+v\[(\d+):(\d+)\]", t)
            if op == "s_nop":  # an s_nop N stands for N + 1 wait states
                window = (window + [None] * (int(t.split()[1]) + 1))[-12:]
                continue
            if op.startswith("v_fma_mix_f32"):
                n_mix += 1
                srcs = [int(x) for x in re.findall(r"v(\d+)", t.split(",", 1)[1])]
                for w in window:
                    if w and any(w[0] <= s <= w[1] for s in srcs):
                        bad.append((src, t))
            window = (window + [(int(m.group(1)), int(m.group(2))) if m else None])[-12:]
        assert n_mix > 100, (src, n_mix)  # the check looked at something
    assert not bad, bad[:5]
