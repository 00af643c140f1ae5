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
    assert lib.vapor_abi_version() == _lib.ABI_VERSION == 3
    assert isinstance(lib.vapor_last_error(), bytes)


def test_product_build_carries_no_developer_switch():
    """timing stamps, A/B variants and tuning constants exist only behind -DVAPOR_DEV_BUILD, which
    reports itself; the library the package loads reports none, and a developer switch without the guard does not compile."""
    import subprocess
    from vapor_amd import _lib, build
    lib = _lib.load()
    assert lib.vapor_build_flags() == b""
    src = open(os.path.join(ROOT, "vapor_amd", "csrc", "vapor_kernels.h")).read() + open(os.path.join(ROOT, "vapor_amd", "csrc", "vapor_hip.hip")).read()
    for gone in ("VAPOR_ABL_", "VAPOR_AB_OLD", "VAPOR_AB_DYN_LDS", "JoinHalf"):
        assert gone not in src, gone
    # the guard itself: preprocessing the kernels with a developer switch and without VAPOR_DEV_BUILD fails
    cmd = [build.hipcc(), "--offload-arch=gfx950", "-std=c++17", "-E", "-DVAPOR_PHASE_TIMING", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "vapor_amd", "csrc"), "-x", "hip", "--cuda-host-only",
           os.path.join(ROOT, "vapor_amd", "csrc", "vapor_kernels.h"), "-o", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode != 0 and "developer switch without -DVAPOR_DEV_BUILD" in r.stderr, r.stderr[-500:]


def test_pair_struct_layout_matches_header():
    from vapor_amd import _lib
    assert _lib.PAIR_DTYPE.itemsize == 20
    assert [_lib.PAIR_DTYPE.fields[k][1] for k in ("seq1", "seq2", "off2", "k", "flags")] == [0, 4, 8, 12, 16]


def test_the_cpu_twin_says_what_it_is_and_the_loader_refuses_it(monkeypatch):
    """the CPU twin of the C ABI (oracle/, test infrastructure) reports the build flag "cpu-twin"; the
    product loader refuses it - also when VAPOR_HIP_LIB names it - unless VAPOR_ALLOW_TWIN=1 is set beside it, which only
    tests/at_size_check.py does."""
    import pytest
    from oracle import oracle as orc
    from vapor_amd import _lib
    twin = orc.build_twin()
    raw = ctypes.CDLL(twin)
    raw.vapor_build_flags.restype = ctypes.c_char_p
    assert raw.vapor_build_flags() == b"cpu-twin"
    monkeypatch.delenv("VAPOR_HIP_LIB", raising=False)
    monkeypatch.delenv("VAPOR_ALLOW_TWIN", raising=False)
    with pytest.raises(RuntimeError, match="CPU twin"):
        _lib.checked(ctypes.CDLL(twin), twin)
    monkeypatch.setenv("VAPOR_HIP_LIB", twin)
    with pytest.raises(RuntimeError, match="CPU twin"):
        _lib.checked(ctypes.CDLL(twin), twin)
    monkeypatch.setenv("VAPOR_ALLOW_TWIN", "1")
    assert _lib.checked(ctypes.CDLL(twin), twin).vapor_abi_version() == _lib.ABI_VERSION


def test_a_stale_library_gets_the_rebuild_message(tmp_path):
    """a library without the newer symbols (or another ABI version) fails with the rebuild message, not with an
    'undefined symbol' from the binding."""
    import subprocess
    import pytest
    from vapor_amd import _lib
    src = tmp_path / "stale.c"
    src.write_text('int vapor_abi_version(void) { return 1; }\nconst char* vapor_build_flags(void) { return ""; }\n')
    so = str(tmp_path / "libstale.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", so, str(src)])
    with pytest.raises(RuntimeError, match="rebuild with"):
        _lib.checked(ctypes.CDLL(so), so)
    src.write_text('int vapor_abi_version(void) { return %d; }\nconst char* vapor_build_flags(void) { return ""; }\n' % _lib.ABI_VERSION)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", so + "2", str(src)])
    with pytest.raises(RuntimeError, match="rebuild with"):          # right version, symbols missing
        _lib.checked(ctypes.CDLL(so + "2"), so + "2")


def test_the_library_is_tied_to_its_sources_by_content(tmp_path, monkeypatch):
    """the binary carries the sha of the sources it was built from (vapor_source_id); build() rebuilds on a
    mismatch instead of on mtime, the loader refuses a library built from other sources, bench.py's kernel_source_id is the
    loaded library's."""
    import subprocess
    import pytest
    from vapor_amd import _lib, build
    so = build.build()
    sid = build.source_id()
    assert re.fullmatch(r"[0-9a-f]{16}:[0-9a-f]{16}", sid) and sid.split(":")[0] == build.kernel_source_id()
    assert build.embedded_source_id(so) == sid                            # read from the file, no dlopen
    assert _lib.load().vapor_source_id().decode() == sid                  # and what the loaded library says
    import bench
    assert bench.kernel_source_id() == sid.split(":")[0]
    # a library with the right ABI version and every symbol, built from "other sources", is refused by content
    names = [n for n in _declared() if n not in ("vapor_abi_version", "vapor_build_flags", "vapor_source_id")]
    src = tmp_path / "other.c"
    src.write_text('int vapor_abi_version(void) { return %d; }\nconst char* vapor_build_flags(void) { return ""; }\n'
                   'const char* vapor_source_id(void) { return "0123456789abcdef:0123456789abcdef"; }\n' % _lib.ABI_VERSION
                   + "".join("int %s(void) { return -1; }\n" % n for n in names))
    other = str(tmp_path / "libother.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", other, str(src)])
    monkeypatch.delenv("VAPOR_HIP_LIB", raising=False)
    with pytest.raises(RuntimeError, match="built from sources 0123456789abcdef"):
        _lib.checked(ctypes.CDLL(other), other)
    # an untouched source with a fresh mtime does not trigger a rebuild; the decision is the content's
    before = os.path.getmtime(so)
    os.utime(os.path.join(ROOT, "vapor_amd", "csrc", "vapor_kernels.h"))
    assert build.build() == so and os.path.getmtime(so) == before
