"""The DEFLATE decoder of bgzf_inflate_kernel (vapor_amd/csrc/vapor_bamdev.h) compiled for the host - one "lane" doing the
wavefront's loops in order - against zlib (tools/bamdev_emu.cpp: streams of every level and strategy, stored / fixed / dynamic
blocks, several blocks in a stream, sizes 0 .. 65 536, the CRC-32 by slices, damaged and truncated streams), under the address
and undefined-behaviour sanitizers.  What this cannot see - the wavefront's memory ordering, the kernels around the decoder - is
what tests/test_gpu_bamdev.py checks on the GPU."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.parametrize("llb", ["9", "10"])
def test_decoder_core_against_zlib_under_sanitizers(llb, tmp_path):
    exe = str(tmp_path / "bamdev_emu")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-DVBD_EMU", "-DVBD_LLB=" + llb,
                           "-I" + os.path.join(ROOT, "vapor_amd", "csrc"), os.path.join(ROOT, "tools", "bamdev_emu.cpp"), "-lz", "-o", exe])
    r = subprocess.run([exe, "6"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    assert "streams equal zlib's bytes and CRC-32 (0 refused for table size)" in r.stdout
