#!/usr/bin/env python3
"""bench.py - throughput of the recurrence-plot scoring hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic loci whose packed
sequences and pair descriptors are already resident in HBM:
    join kernel(s) -> clean kernel -> finish kernel (float64: per-read scores, VaPoR_QS / VaPoR_GS /
    VaPoR_GT / VaPoR_GQ per locus) -> per-locus records to the host (N = 1) or RCCL all-gather (N > 1).
Weak scaling: every rank processes its own batch of the same shape.

Prints ONE JSON line on rank 0 (see the keys below).  `roofline` is measured live with HIP
events on the library's own stream (vapor_plan_timings); `cpu_baseline` times the CPU oracle
(oracle/, a restatement: kind "port") on a bounded sample of the same pairs, 1 thread.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--reads-per-task", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--streams", type=int, default=0,
                    help="plans in flight, steps alternate between them (0 = two when a join launch is one workgroup per "
                         "CU, else one)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    # VAPOR_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks on one GPU (RCCL wants a GPU per rank)
    backend = os.environ.get("VAPOR_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # one explicit stream for everything timed: the library's kernels and the all-gather are enqueued on it in
    # program order (the legacy default stream cannot be handed to the library)
    work_stream = torch.cuda.Stream()
    torch.cuda.set_stream(work_stream)

    from vapor_amd import workload as wl
    from vapor_amd.engine import Engine

    spec = wl.WORKLOADS[args.workload]
    w = wl.make_workload(args.workload, seed=1000 + rank, **spec)

    # `--streams` lanes, each with its own context, packed sequences, plan and HIP stream; step i runs on lane
    # i % streams.  The join owns every CU's LDS while it runs, so inside one stream the clean kernel only starts
    # when the slowest join workgroup is done; with two lanes the clean and finish kernels of one step fill the CUs
    # the next step's join has not reached yet or has already left.  Every step is still one complete pass of the
    # hot path over the batch, and all K steps finish inside the timed region.
    class Lane:
        pass
    lanes = []
    upload_s = 0.0
    # A batch whose join is a single wave of workgroups (one per CU) leaves CUs idle at its tail, which a second lane
    # fills; a join of many waves of workgroups keeps the chip busy by itself, and a second lane would only stretch
    # every kernel's event interval.
    n_lanes = args.streams if args.streams > 0 else (2 if len(w.pairs) <= 64 * 256 else 1)
    for li in range(n_lanes):
        ln = Lane()
        ln.stream = work_stream if li == 0 else torch.cuda.Stream()
        ln.eng = Engine(local)
        if args.reads_per_task:
            ln.eng.set_param("reads_per_task", args.reads_per_task)
        t0 = time.perf_counter()
        ln.ss = ln.eng.seqset(w.seqs)
        if li == 0:
            upload_s = time.perf_counter() - t0
        ln.plan = ln.eng.plan(ln.ss, w.pairs)
        ln.plan.set_reads(wl.read_table(w), w.n_loci)
        ln.rec_dev = torch.empty((w.n_loci, 8), dtype=torch.float64, device="cuda")
        # the library enqueues on the lane's stream, so its kernels and the all-gather order themselves
        ln.eng.set_stream(ln.stream.cuda_stream)
        lanes.append(ln)
    plan = lanes[0].plan

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gathered = [None]

    def gather(ln):
        if dist is not None:
            out = torch.empty((world * ln.rec_dev.shape[0], ln.rec_dev.shape[1]), dtype=ln.rec_dev.dtype, device=ln.rec_dev.device)
            dist.all_gather_into_tensor(out, ln.rec_dev)
            gathered[0] = out

    def step(i):
        # join -> clean -> finish enqueued without a host round trip: records to rec_dev and (pinned, async) to the host
        ln = lanes[i % len(lanes)]
        with torch.cuda.stream(ln.stream):
            ln.plan.run_loci_async(device_out=ln.rec_dev.data_ptr())
            gather(ln)

    # one blocking run per lane sizes the record slots (reruns pairs that overflow their first guess); its kernel
    # times are those of the kernels running alone
    for ln in lanes:
        with torch.cuda.stream(ln.stream):
            ln.plan.run_loci(device_out=ln.rec_dev.data_ptr(), want_host=False)
    alone = plan.timings()
    for i in range(args.warmup):
        step(i)
    for ln in lanes:
        ln.plan.sync(want_host=False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    join_ms = clean_ms = dev_ms = fin_ms = 0.0
    for li, ln in enumerate(lanes):
        rec = ln.plan.sync()            # waits (nothing left), averages the lane's per-step HIP events
        n_l = len(range(li, args.steps, len(lanes)))
        tm = ln.plan.timings()
        join_ms += tm["join_ms"] * n_l; clean_ms += tm["clean_ms"] * n_l
        dev_ms += tm["total_ms"] * n_l; fin_ms += tm["finish_ms"] * n_l
        if dist is None and li == (args.steps - 1) % len(lanes):
            gathered[0] = rec.copy()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # untimed: per-pair statistics once, for the algorithmic byte count and the oracle cross-check; and the
    # host-side finish on the same statistics must agree with what the device produced
    st = plan.run().copy()
    host_rec = wl.finish_workload(w, st)
    dev_rec = plan.run_loci().copy()
    ok = ~np.isnan(host_rec[:, 0])
    assert np.array_equal(np.isnan(dev_rec[:, 0]), ~ok) and np.array_equal(dev_rec[ok, 2], host_rec[ok, 2]) \
        and np.allclose(dev_rec[ok, :2], host_rec[ok, :2], rtol=0, atol=1e-9), "device / host finish mismatch"
    alg_bytes, cells = plan.algorithmic()
    n_pairs = len(w.pairs)
    steps = max(args.steps, 1)
    join_avg, clean_avg = join_ms / steps, clean_ms / steps
    launches = plan.timings()["join_launches"]
    ms_per_step = elapsed / steps * 1e3
    loci_s = w.n_loci * world * steps / elapsed
    cells_s = cells * world * steps / elapsed
    # the dominant kernel is the one that takes longest when it runs alone (with two lanes the other kernels' event
    # intervals include waiting for CUs the join holds); its duration is the live average over the timed steps
    dom = "join_kernel" if alone["join_ms"] >= alone["clean_ms"] else "clean_kernel"
    dom_ms = join_avg if dom == "join_kernel" else clean_avg
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0

    # HBM-side bytes of the dominant kernel per launch, from the committed PMC passes of this same workload
    # (the byte counts depend on the batch only, not on the run)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_%s_traffic.json" % args.workload)
    if os.path.exists(tpath):
        try:
            traffic = int(json.load(open(tpath))[dom]["bytes"])
        except Exception:
            traffic = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:       # reported at N = 1 only
        from oracle import oracle as orc
        orc.build()
        done = 0
        c_cells = 0
        t0 = time.perf_counter()
        while done < n_pairs and time.perf_counter() - t0 < args.cpu_seconds:
            p = w.pairs[done]
            exp = orc.pair_stats(int(p["k"]), w.seqs[p["seq1"]], w.seqs[p["seq2"]][int(p["off2"]):])
            if not int(p["flags"]) & 1:
                exp[3] = exp[4] = 0
            if not int(p["flags"]) & 2:
                exp[5] = exp[6] = exp[9] = 0
            assert exp[:10].tolist() == st[done, :10].tolist(), "GPU/oracle mismatch on pair %d" % done
            c_cells += len(w.seqs[p["seq1"]]) * (len(w.seqs[p["seq2"]]) - int(p["off2"]))
            done += 1
        ct = time.perf_counter() - t0
        pairs_per_locus = n_pairs / w.n_loci
        cpu = {"value": round(done / pairs_per_locus / ct, 4), "unit": "loci/s", "cores": 1, "kind": "port",
               "sample": "first %d of %d (read, allele) dot plots of the same batch (fill + C1/C2 clean + counts, "
                         "oracle/vapor_oracle.c, gcc -O2, 1 thread, %.1f s); each checked equal to the GPU record"
                         % (done, n_pairs, ct),
               "cells_per_s": round(c_cells / ct, 1)}

    if rank == 0:
        out = {
            "metric": "SV loci validated/sec",
            "value": round(loci_s, 3),
            "unit": "loci/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "%s: %d loci/GPU x %d reads (%d bp) x 2 allele windows (%d bp), k=10, types %s"
                                   % (args.workload, w.n_loci, spec["reads_per_locus"], spec["read_len"],
                                      spec["allele_len"], "/".join(sorted(set(w.svtypes)))),
                       "pairs_per_step_per_gpu": n_pairs, "parallelism": "loci sharded over %d GPU(s)" % world},
            "cells_per_s": round(cells_s, 1),
            "hits_per_step": int(st[:, 0].sum()),
            "loci_with_scores": int(np.isfinite(gathered[0].reshape(-1, 8)[:, 0].cpu().numpy() if hasattr(gathered[0], "cpu")
                                                else gathered[0][:, 0]).sum()),
            "kernel_ms": {"join": round(join_avg, 4), "clean": round(clean_avg, 4), "finish": round(fin_ms / steps, 4),
                          "device_total": round(dev_ms / steps, 4), "join_launches": launches, "streams": len(lanes),
                          "alone": {"join": round(alone["join_ms"], 4), "clean": round(alone["clean_ms"], 4)}},
            "upload_pack_s": round(upload_s, 4),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(alg_bytes)},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
