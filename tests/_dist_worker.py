"""Worker for tests/test_dist_gloo.py: one rank of a world_size-N `vapor bed` run on CPU
(gloo, or the launcher's own exchange through files), device work answered by the oracle-backed fake engine (test infrastructure)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from conftest import load_golden  # noqa: E402
from fake_engine import FakeEngine  # noqa: E402


def main():
    case_name, bed, out, figs = sys.argv[1:5]
    from oracle import oracle as orc
    from vapor_amd import cli, dist, pipeline, seqio, synth
    case = [c for c in load_golden("locus_bed.json.gz")["cases"] if c["name"] == case_name][0]
    pipeline.set_engine(FakeEngine(orc))
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
    dist.init_from_env(os.environ.get("VAPOR_TEST_BACKEND", "gloo"))
    rc = cli.main(["bed", "--sv-input", bed, "--reference", "ref.fa", "--pacbio-input", "x.bam",
                   "--output-path", figs, "--output-file", out, "--no-figures", "--chunk", "3"])
    n_plans = len(pipeline.get_engine().batches)
    print("rank %s plans %d" % (os.environ.get("RANK"), n_plans))
    sys.exit(rc)


if __name__ == "__main__":
    main()
