"""In-tree build of libvapor_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "libvapor_hip.so")
SOURCES = [os.path.join(HERE, "csrc", "vapor_hip.hip"), os.path.join(HERE, "csrc", "vapor_bam.cpp")]
DEPS = SOURCES + [os.path.join(HERE, "csrc", "vapor_kernels.h"), os.path.join(HERE, "csrc", "vapor_inflate.h"),
                  os.path.join(HERE, "csrc", "vapor_bamdev.h"),
                  os.path.join(ROOT, "include", "vapor_hip.h")]


# The kernels issue their wave-level atomics from one lane already (`if (lane == 0) atomicAdd(...)`); LLVM's atomic
# optimizer wraps each of them in another mbcnt / compare / exec-mask sequence.  Off: clean_kernel 0.0837 -> 0.0826 ms.
# The kernels are vector-issue-bound with their occupancy set by LDS, not by registers: the scheduler strategy that orders for
# instruction-level parallelism instead of for the fewest registers takes a cfg2 pass from 0.1258 to 0.1246 ms (two plans in
# flight; cfg3 the same within 0.2 %, checksums equal; profiles/r05_sched_strategy.txt).
EXTRA_FLAGS = ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


KERNEL_FILES = [os.path.join(HERE, "csrc", "vapor_kernels.h"), os.path.join(HERE, "csrc", "vapor_bamdev.h"), os.path.join(HERE, "csrc", "vapor_hip.hip"),
                os.path.abspath(__file__)]
_ID_RE = re.compile(rb"VAPOR_SOURCE_ID=([0-9a-f]{16}:[0-9a-f]{16})")


def _sha16(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def kernel_source_id() -> str:
    """sha256 (16 hex digits) over the device code, the host code that launches it and this file (the compiler flags): what a
    committed counter summary must have been measured on.  The host-only helpers (BAM reader, inflate) and the header's
    prose are not part of it: they cannot move a kernel's counters."""
    return _sha16(KERNEL_FILES)


def source_id() -> str:
    """`<kernel_source_id>:<sha over every file the library is built from>` - compiled into the library (vapor_source_id())."""
    return kernel_source_id() + ":" + _sha16(sorted(set(DEPS + [os.path.abspath(__file__)])))


def embedded_source_id(so: str = SO):
    """The id a built library carries, read from the file itself (no dlopen); None for a library built without one."""
    try:
        with open(so, "rb") as f:
            m = _ID_RE.search(f.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def build(force: bool = False, verbose: bool = False) -> str:
    # (the binary is tied to its sources by content, not by mtime: a library travels to the GPU box in the working tree, and
    # one that is newer than sources it was not built from would be quoted against fresh counters silently)
    sid = source_id()
    if not force and os.path.exists(SO) and embedded_source_id() == sid:
        return SO
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", '-DVAPOR_SOURCE_ID="%s"' % sid] + EXTRA_FLAGS + [
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
           "-Wall", "-Wno-unused-function", "-o", SO] + SOURCES + ["-lz"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
