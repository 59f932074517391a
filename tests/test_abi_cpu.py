"""The C-ABI shared library loads without a GPU and exports exactly the entry points include/upnerf_hip.h declares
(no compute calls here).  Also: the binding refuses to work without the library (no fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "upnerf_hip.h")).read()
    return sorted(set(re.findall(r"^int\s+(upnerf_\w+)\s*\(", src, flags=re.M)))


def test_header_symbols_are_exported_and_bound():
    from upnerf_amd import _lib
    names = declared()
    assert len(names) >= 15
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), f"{n} declared in include/upnerf_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "ctypes binding and header disagree"
    assert _lib.lib.upnerf_abi_version() == 1


def test_struct_sizes_match_the_c_layout():
    """ctypes mirrors of the argument structs must have the size the C compiler gives them."""
    import subprocess
    import tempfile
    from upnerf_amd import _lib
    prog = r'''
    #include <stdio.h>
    #include "upnerf_hip.h"
    int main(){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(upnerf_layout), sizeof(upnerf_field_fwd_args),
      sizeof(upnerf_composite_fwd_args), sizeof(upnerf_composite_bwd_args), sizeof(upnerf_field_bwd_args),
      sizeof(upnerf_loss_args), sizeof(upnerf_loss_grads), sizeof(upnerf_frag_desc), sizeof(upnerf_frag16_desc),
      sizeof(upnerf_gather_rays_args), sizeof(upnerf_pack_desc)); return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    mine = [ctypes.sizeof(t) for t in (_lib.Layout, _lib.FieldFwdArgs, _lib.CompositeFwdArgs, _lib.CompositeBwdArgs,
                                       _lib.FieldBwdArgs, _lib.LossArgs, _lib.LossGrads, _lib.FragDesc, _lib.Frag16Desc,
                                       _lib.GatherRaysArgs, _lib.PackDesc)]
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
