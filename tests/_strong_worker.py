"""Worker for tests/test_dist_gloo.py::test_bench_strong_record_two_ranks: one rank of bench.py's strong-scaling leg
(bench.strong_record: BASELINE configs[3]'s call set sharded over the ranks by cli.job_cost, one gather of the scores, the table's
sha against one rank's) on CPU - gloo between the ranks, device work answered by the oracle-backed fake engine (test
infrastructure), the world shrunk to reads the oracle scores in seconds."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from fake_engine import FakeEngine  # noqa: E402


def main():
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    from vapor_amd import workload as wl
    import bench
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl.AT_SIZE["cfg4"] = dict(mode="vcf", loci=24, base=24, n_reads=6, read_len=2600, seed=404)      # (the shape, not the size)
    os.environ["VAPOR_HOST_PROCS"] = "0"
    rec = bench.strong_record(FakeEngine(orc), dist, "gloo", world, rank, int(sys.argv[1]), int(sys.argv[2]), dist.barrier, torch)
    if rank == 0:
        print("STRONG " + json.dumps(rec))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
