"""The N > 1 path on CPU: two gloo ranks shard the loci of a `vapor bed` run, all-gather the
per-locus records, and rank 0 writes the same table a single process (and the reference) writes."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_bed_run_matches_reference(tmp_path, world, oracle):
    case = [c for c in load_golden("locus_bed.json.gz")["cases"] if c["name"] == "bed_hom_alt"][0]
    bed = tmp_path / "in.bed"
    bed.write_text(case["bed"])
    out = tmp_path / "out.vapor"
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), case["name"],
                                       str(bed), str(out), str(tmp_path / "figs")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
        assert p.returncode == 0, o
    assert out.read_text() == case["vapor_text"]
    # every rank did device work on its own share only
    assert all("plans" in l for l in logs)
