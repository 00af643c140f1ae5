"""CPU-side checks of the C-ABI library: it loads and exports every symbol the header
declares.  No compute call is made here (there is no GPU in the CI container)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "vapor_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vapor_[a-z_0-9]+)\s*\(", src)))


def test_header_functions_are_exported():
    from vapor_amd import build, _lib
    so = build.build()
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libvapor_hip.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names


def test_abi_version_and_error_string():
    from vapor_amd import _lib
    lib = _lib.load()
    assert lib.vapor_abi_version() == 1
    assert isinstance(lib.vapor_last_error(), bytes)


def test_pair_struct_layout_matches_header():
    from vapor_amd import _lib
    assert _lib.PAIR_DTYPE.itemsize == 20
    assert [_lib.PAIR_DTYPE.fields[k][1] for k in ("seq1", "seq2", "off2", "k", "flags")] == [0, 4, 8, 12, 16]
