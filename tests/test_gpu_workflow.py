"""The workflow layer with real ranks on a real GPU: `python -m vapor_amd.workflow --gpus 1
--ranks-per-gpu 2` starts two fresh processes that share the GPU (gloo between them), each scoring its share of
the loci from FASTA/.fai and BAM/.bai files through the in-process readers; rank 0 writes the table, the launcher
the sorted, block-gzipped table and its tabix index (what wdl/TasksBenchmark.wdl:286-301 produces)."""
import gzip
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_gpu_write_the_single_process_table(tmp_path):
    from vapor_amd import cli, pipeline, seqio, synth, workflow
    w = synth.make_world(seed=31, n_loci=18, svtypes=("DEL", "DEL", "INV", "INS"), span_range=(150, 1200), read_len=3600, n_reads=9)
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    fa, bam = synth.write_world_files(w, str(tmp_path))
    bed = tmp_path / "in.bed"
    bed.write_text(synth.bed_text(w))
    # single process, same files
    pipeline.set_engine(None)
    seqio.set_backend(seqio.InProcessBam())
    try:
        one = tmp_path / "one.vapor"
        assert cli.main(["bed", "--sv-input", str(bed), "--reference", fa, "--pacbio-input", bam,
                         "--output-path", str(tmp_path / "figs1"), "--output-file", str(one), "--no-figures"]) == 0
    finally:
        seqio.set_backend(None)
    rows = one.read_text().splitlines()
    assert len(rows) == 19 and sum("NA" not in r for r in rows[1:]) >= 6
    # two ranks sharing the GPU, fresh processes
    two = tmp_path / "two.vapor"
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_PORT="29533",
               OMP_NUM_THREADS="1")
    env.pop("VAPOR_BAM_BACKEND", None)
    p = subprocess.run([sys.executable, "-m", "vapor_amd.workflow", "--gpus", "1", "--ranks-per-gpu", "2", "--prefix",
                        str(tmp_path / "two"), "bed", "--sv-input", str(bed), "--reference", fa, "--pacbio-input", bam,
                        "--output-path", str(tmp_path / "figs2"), "--output-file", str(two), "--no-figures", "--chunk", "5"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert two.read_text() == one.read_text()
    gz = str(tmp_path / "two.bed.gz")
    body = gzip.open(gz, "rt").read().splitlines()
    assert body == workflow.sort_rows(rows[1:])
    l = w.loci[3]
    hit = workflow.tabix_query(gz, l.chrom, l.start, l.end)
    assert len(hit) == 1 and hit[0].split("\t")[0] == l.chrom
