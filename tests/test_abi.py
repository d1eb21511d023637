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
    from hallucidet_amd import _abi
    assert lib.hd_abi_version() == _abi.ABI_VERSION == 6
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


def _gcc_sizeofs(tmp_path, names):
    import subprocess
    src = '#include <stdio.h>\n#include "hallucidet_hip.h"\nint main(void) {\n'
    for st in names:
        src += '  printf("%s %%zu\\n", sizeof(%s));\n' % (st, st)
    src += "  return 0;\n}\n"
    c = tmp_path / "sizes.c"
    c.write_text(src)
    exe = str(tmp_path / "sizes")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, str(c)], check=True)
    return {k: int(v) for k, v in (l.split() for l in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines())}


def test_integration_md_structs_match_the_header(tmp_path):
    """INTEGRATION.md is the document a maintainer pastes from: every ```python block in it that defines a ctypes Structure is
    executed (the CDLL / torch lines stubbed out) and each Structure's sizeof must equal what gcc lays out for the struct the
    snippet says it mirrors.  A header change that is not carried into the document fails here (round 5's review: the pasted
    hd_conv_args was 136 bytes, the header's 224)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "C.Structure" in b]
    assert blocks, "INTEGRATION.md no longer shows the ctypes binding"
    checked = 0
    for b in blocks:
        mirrors = dict(re.findall(r"class\s+(\w+)\(C\.Structure\):\s*#\s*mirrors struct (\w+)", b))
        assert mirrors, "a Structure in INTEGRATION.md does not say which struct it mirrors"
        ns = {}
        # keep only the import and the class statements: nothing that opens the library or touches a device
        code = "import ctypes as C\n"
        for m in re.finditer(r"^class\s+\w+\(C\.Structure\):.*?(?=^\S)", b, flags=re.S | re.M):
            code += m.group(0)
        exec(compile(code, "INTEGRATION.md", "exec"), ns)
        sizes = _gcc_sizeofs(tmp_path, sorted(set(mirrors.values())))
        for cls, st in mirrors.items():
            assert ctypes.sizeof(ns[cls]) == sizes[st], "INTEGRATION.md's %s is %d bytes, %s is %d" % (cls, ctypes.sizeof(ns[cls]), st, sizes[st])
            checked += 1
        m = re.search(r"hd_abi_version\(\)\s*==\s*(\d+)", b)
        if m:
            from hallucidet_amd import _abi
            assert int(m.group(1)) == _abi.ABI_VERSION
    assert checked >= 1
    from hallucidet_amd._abi import ConvArgs
    assert [f[0] for f in ns["ConvArgs"]._fields_] == [f[0] for f in ConvArgs._fields_]


def test_load_refuses_a_library_of_another_abi_version(monkeypatch, lib):
    from hallucidet_amd import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "ABI_VERSION", _abi.ABI_VERSION + 1)
    with pytest.raises(_abi.HipLibraryMissing, match="rebuild"):
        _abi.load()


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
