"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol
include/hallucidet_hip.h declares; the ctypes table covers the header; argument validation returns status
codes (never throws across the C boundary)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hallucidet_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hd_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from hallucidet_amd import _abi
    return _abi.load()


def test_header_symbols_are_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, "declared in include/hallucidet_hip.h but not exported: %s" % missing


def test_ctypes_table_matches_header():
    from hallucidet_amd import _abi
    syms = set(declared_symbols())
    table = set(_abi.PROTOTYPES)
    assert syms == table, "header-only: %s ; table-only: %s" % (sorted(syms - table), sorted(table - syms))


def test_identity(lib):
    assert lib.hd_abi_version() == 5
    assert lib.hd_arch() == b"gfx950"


def test_struct_layout_matches_c(tmp_path):
    """sizeof / offsetof of the ctypes mirrors equal what the C compiler lays out for include/hallucidet_hip.h (gcc, LP64)."""
    import os
    import subprocess
    from hallucidet_amd._abi import ConvArgs, WgradArgs
    fields = {"hd_conv_args": (ConvArgs, ["x", "y", "stats", "N", "out_mode", "in_scale", "in_shift", "in_relu", "out_pool2", "bs_y", "bs_z", "bs_mean", "bs_invstd", "bs_gamma", "bs_beta", "bs_relu", "y2"]),
              "hd_wgrad_args": (WgradArgs, ["x", "slab", "N", "nsplit", "in_scale", "in_shift", "in_relu", "dw_oihw", "dw_scale"])}
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "hallucidet_hip.h"\nint main(void) {\n'
    for st, (_, fs) in fields.items():
        src += '  printf("%s %%zu\\n", sizeof(%s));\n' % (st, st)
        for f in fs:
            src += '  printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (st, f, st, f)
    src += "  return 0;\n}\n"
    c = tmp_path / "layout.c"
    c.write_text(src)
    exe = str(tmp_path / "layout")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-I", inc, "-o", exe, str(c)], check=True)
    got = dict(l.split() for l in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines())
    for st, (cls, fs) in fields.items():
        assert int(got[st]) == ctypes.sizeof(cls), st
        for f in fs:
            assert int(got["%s.%s" % (st, f)]) == getattr(cls, f).offset, (st, f)
    assert ctypes.sizeof(ConvArgs) == 8 * 8 + 18 * 4 + 2 * 8 + 2 * 4 + 6 * 8 + 2 * 4 + 8 and ConvArgs.N.offset == 64


def test_bad_arguments_return_status_codes(lib):
    from hallucidet_amd._abi import ConvArgs
    a = ConvArgs()
    assert lib.hd_conv2d(ctypes.byref(a), None) == -1
    assert b"null" in lib.hd_last_error()
    assert lib.hd_bn_apply(None, None, None, None, None, 0, 8, 1, None) == -1
    assert lib.hd_adam_step(None, None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, None) == -1
    assert lib.hd_colsum(None, 0, 0, None, None, None) == -1
    assert lib.hd_rowsum(None, 0, 0, None, 0, None) == -1


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    from hallucidet_amd import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_abi.HipLibraryMissing):
        _abi.load()


def test_ops_refuse_cpu_tensors():
    import torch
    from hallucidet_amd import ops
    x = torch.zeros(1, 4, 4, 8, dtype=torch.float16)
    w = torch.zeros(8, 72, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.conv2d(x, w, 3, 3, pad=1)
