"""In-tree build of libvapor_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "libvapor_hip.so")
SOURCES = [os.path.join(HERE, "csrc", "vapor_hip.hip"), os.path.join(HERE, "csrc", "vapor_bam.cpp")]
DEPS = SOURCES + [os.path.join(HERE, "csrc", "vapor_kernels.h"), os.path.join(HERE, "csrc", "vapor_inflate.h"),
                  os.path.join(ROOT, "include", "vapor_hip.h")]


# The kernels issue their wave-level atomics from one lane already (`if (lane == 0) atomicAdd(...)`); LLVM's atomic
# optimizer wraps each of them in another mbcnt / compare / exec-mask sequence.  Off: clean_kernel 0.0837 -> 0.0826 ms.
EXTRA_FLAGS = ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in DEPS):
        return SO
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared"] + EXTRA_FLAGS + [
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
           "-Wall", "-Wno-unused-function", "-o", SO] + SOURCES + ["-lz"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
