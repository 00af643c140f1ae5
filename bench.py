#!/usr/bin/env python3
"""bench.py - throughput of the recurrence-plot scoring hot path on MI355X.

One PASS = the hot path over one batch of synthetic loci (BASELINE.json configs[1], "cfg2": 100 DEL/TANDUP loci x 20
reads of 10 kb x ref and alt windows of 20 kb) whose packed sequences and pair descriptors are already resident in HBM:
    join kernel(s) -> clean kernel -> finish kernel (float64: per-read scores, VaPoR_QS / VaPoR_GS / VaPoR_GT /
    VaPoR_GQ per locus) -> per-locus records to the host (N = 1) or into the step's device buffer, which is all-gathered
    over RCCL once per step (N > 1: every pass's records, in one collective per step).
One STEP = `passes_per_step` passes back to back; the number is sized by a probe before the timed region so that the
K timed steps last about a second or more (a single pass takes 0.3 ms; K is the driver's).  Two plans over the same
batch are kept in flight and alternate (the library runs each plan on one of its two streams), so the clean and finish
kernels of one pass fill the CUs the next pass's join has not reached or has already left.
Weak scaling: every rank processes its own batch of the same shape.

Prints ONE JSON line on rank 0.  `value` is the resident-kernel rate (inputs in HBM, window size fixed at 10);
`inclusive` next to it is the rate when every batch also pays upload + packing, planning and the self dot plots of
window_size_refine (SF:2030-2046), `sub` repeats the resident measurement on BASELINE configs[2] ("cfg3").
`roofline` is measured live with HIP events on the library's streams (vapor_plan_timings); `cpu_baseline` times the
CPU oracle (oracle/, a restatement: kind "port") on a bounded sample of the same pairs, 1 thread.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def _load_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except Exception:
            return None
    return None


def kernel_source_id() -> str:
    """The id of the kernel sources in the tree (vapor_amd/build.py) - and the LOADED library must carry it: what a committed
    counter summary must have been measured on."""
    from vapor_amd import _lib, build
    sid = build.kernel_source_id()
    have = _lib.load().vapor_source_id().decode()
    assert have.split(":")[0] == sid, "the loaded library was built from other kernel sources (%s) than the tree holds (%s)" % (have, sid)
    return sid


PROFILE_ROUNDS = ("r05", "r04", "r03")       # newest first: the committed counter summaries bench.py may quote


def _profile_for(kind, workload, src_id):
    """The newest committed summary profiles/<round>_<workload>_<kind>.json (traffic / util / kernel_stats) and whether it was
    measured on THIS kernel source (only then are its numbers quoted; its provenance is printed either way)."""
    for rnd in PROFILE_ROUNDS:
        j = _load_json("%s_%s_%s.json" % (rnd, workload, kind))
        if j:
            prov = {"file": "profiles/%s_%s_%s.json" % (rnd, workload, kind), "measured_on_source": j.get("source_id"), "this_source": src_id,
                    "live": False}
            return (j if j.get("source_id") == src_id else None), prov
    return None, None


def roofline_record(workload, alg_bytes, n_rec, ms_per_pass, live, alone, src_id):
    """SURVEY 8d's record for one workload.  `live`: HIP-event intervals (ms) of join (+ remap_kernel where the plan runs it), clean
    and finish averaged over the timed passes (two plans in flight: an interval includes waiting for CUs the other plan holds);
    `alone`: the same kernels by themselves on an idle device.  The dominant kernel is the one with the largest share of GPU time
    in the committed rocprofv3 --stats summary of the same command when that was measured on this source, else the one with the
    largest live interval.  achieved / frac: algorithmic bytes per launch over that kernel's live interval, against the HBM roof the
    task names; frac_rocprof the same over the rocprof average duration; frac_whole_pass over the pass.  The kernels are
    vector-issue-bound: valu_frac = quad-cycles the vector ALUs were occupied / quad-cycles available while
    a CU was busy, (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) / SQ_BUSY_CU_CYCLES from the committed --pmc passes (two simple
    instructions of different waves share a quad-cycle on gfx950: tools/micro/valu_ops.hip, profiles/r02_valu_ops.txt)."""
    traffic, t_prov = _profile_for("traffic", workload, src_id)
    util, u_prov = _profile_for("util", workload, src_id)
    stats, s_prov = _profile_for("kernel_stats", workload, src_id)
    names = {"join_kernel": "join", "clean_kernel": "clean"}
    if stats:
        share = {k: stats["kernels"].get(k, {}).get("percentage", 0.0) for k in names}
        dom = max(share, key=share.get)
        dom_by = "largest share of GPU time in " + s_prov["file"]
    else:
        dom = "join_kernel" if live["join"] >= live["clean"] else "clean_kernel"
        dom_by = "largest live HIP-event interval (no rocprofv3 summary of this source committed)"
    key = names[dom]
    dom_ms = live[key]
    gbs = lambda ms: alg_bytes / (ms * 1e-3) / 1e9 if ms and ms > 0 else None
    frac = lambda ms: round(gbs(ms) / HBM_PEAK_GBS, 5) if ms and ms > 0 else None
    rp_us = stats["kernels"].get(dom, {}).get("avg_us") if stats else None
    out = {"bound": ("valu_issue" if util and dom in util else "hbm"), "priced_against": "hbm", "kernel": dom, "kernel_chosen_by": dom_by,
           "achieved": round(gbs(dom_ms), 2) if gbs(dom_ms) else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac(dom_ms),
           "algorithmic_bytes_per_launch": int(alg_bytes), "record_bytes_per_launch": 8 * int(n_rec),
           "duration_us": {"live_interval": round(dom_ms * 1e3, 2), "alone": round(alone[key] * 1e3, 2),
                           "rocprof_avg": round(rp_us, 2) if rp_us else None},
           "frac_alone": frac(alone[key]), "frac_rocprof": frac(rp_us * 1e-3) if rp_us else None,
           "frac_whole_pass": frac(ms_per_pass),
           "kernel_shares_rocprof": ({k: stats["kernels"][k] for k in stats["kernels"]} if stats else None), "kernel_stats_source": s_prov}
    # fabric-side bytes (PMC): the dominant kernel's, and every kernel of a pass summed - one basis for numerator and denominator
    tr = None
    if traffic:
        tr = int(traffic[dom]["bytes"]) if dom in traffic else None
        allk = {k: int(v["bytes"]) for k, v in traffic.items() if isinstance(v, dict) and "bytes" in v}
        out["traffic_all_kernels"] = {"bytes": int(sum(allk.values())), "by_kernel": allk,
                                      "over_algorithmic": round(sum(allk.values()) / alg_bytes, 4),
                                      "gbs_over_pass": round(sum(allk.values()) / (ms_per_pass * 1e-3) / 1e9, 2),
                                      "frac_of_hbm_over_pass": round(sum(allk.values()) / (ms_per_pass * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    out["traffic"] = tr
    out["traffic_source"] = t_prov
    out["traffic_gbs"] = round(tr / (dom_ms * 1e-3) / 1e9, 2) if tr and dom_ms > 0 else None
    out["traffic_frac"] = round(tr / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if tr and dom_ms > 0 else None
    # the roof the kernels really sit under
    if util and dom in util:
        u = util[dom]
        out["valu_frac"] = u.get("valu_busy")
        quad = 0.0
        clock = None
        for k, v in util.items():
            if isinstance(v, dict) and "valu_wave_instructions" in v:
                quad += v["valu_wave_instructions"] * (1.0 - v.get("valu_instructions_issued_in_pairs", 0.0) / 2.0)
                if v.get("cu_busy_cycles_per_cu") and v.get("us_under_pmc"):
                    clock = max(clock or 0.0, v["cu_busy_cycles_per_cu"] / (v["us_under_pmc"] * 1e-6))
        if clock:
            # every kernel of a pass: quad-cycles of vector work over what 1 024 SIMDs offer in the pass's time at the clock the
            # counters show (cu busy cycles / kernel time)
            out["valu_frac_whole_pass"] = round(quad / (1024.0 * ms_per_pass * 1e-3 * clock / 4.0), 4)
            out["clock_ghz_under_counters"] = round(clock / 1e9, 3)
    out["binds"] = util[dom] if util and dom in util else None
    out["binds_source"] = u_prov
    return out




class Resident:
    """A batch resident in HBM with `n_plans` plans over it; pass i runs on plan i % n_plans."""

    def __init__(self, eng, w, wl, n_plans, torch, dist, world, coll_stream):
        self.w, self.torch, self.dist, self.world, self.coll = w, torch, dist, world, coll_stream
        t0 = time.perf_counter()
        self.ss = w.upload(eng)
        self.upload_s = time.perf_counter() - t0
        self.plans, self.rec = [], []
        for _ in range(n_plans):
            p = eng.plan(self.ss, w.pairs)
            p.set_reads(wl.read_table(w), w.n_loci)
            self.plans.append(p)
            self.rec.append(torch.empty((w.n_loci, 8), dtype=torch.float64, device="cuda"))
        self.gathered = None
        self.stepbuf = self.gbuf = None
        self.i = 0
        # one blocking run per plan sizes the record slots (reruns pairs that overflow their first guess); its
        # kernel times are those of the kernels running alone
        for p, r in zip(self.plans, self.rec):
            p.run_loci(device_out=r.data_ptr(), want_host=False)
        self.alone = self.plans[0].timings()

    def set_step(self, inner):
        """N > 1: the per-locus records of a whole step (`inner` passes) go to one device buffer and are all-gathered ONCE per
        step - fewer, larger collectives: a pass is 0.18 ms, an all-gather call costs the host half of that whatever it
        carries (measured with one rank: 0.178 -> 0.265 ms per pass with a gather behind every pass), and a real run gathers
        its scores once.  Every pass's records are still gathered."""
        if self.dist is None:
            return
        n = self.w.n_loci
        self.stepbuf = self.torch.empty((inner, n, 8), dtype=self.torch.float64, device="cuda")
        self.gbuf = self.torch.empty((self.world * inner, n, 8), dtype=self.torch.float64, device="cuda")

    def one_pass(self, slot=0):
        k = self.i % len(self.plans)
        self.i += 1
        p = self.plans[k]
        out = self.rec[k] if self.stepbuf is None else self.stepbuf[slot]
        p.run_loci_async(device_out=out.data_ptr())     # join -> clean -> finish enqueued, no host round trip

    def begin_step(self):
        if self.stepbuf is not None:
            # the passes of this step overwrite the buffer the last step's all-gather reads: they wait for it on the device
            for p in self.plans:
                p.after(self.coll.cuda_stream)

    def end_step(self):
        if self.stepbuf is not None:
            # the collective runs on torch's stream, which waits (on the device) for the last pass of every plan
            for p in self.plans:
                p.then(self.coll.cuda_stream)
            with self.torch.cuda.stream(self.coll):
                self.dist.all_gather_into_tensor(self.gbuf, self.stepbuf)
            self.gathered = self.gbuf

    def drain(self, want_host=False):
        """Waits for every plan; returns (passes folded per plan, timings per plan, last records of plan 0)."""
        out = []
        for p in self.plans:
            rec = p.sync(want_host=want_host)
            out.append((p.timings(), rec.copy() if want_host else None))
        return out

    def close(self):
        for p in self.plans:
            p.close()
        self.ss.close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--passes-per-step", type=int, default=0, help="0: sized by a probe so that the timed region lasts >= --min-seconds")
    # (>= 6 s: the driver samples the GPU's busy state every 5 s - a 1.2 s timed region, as in rounds 1-3, fell between samples)
    ap.add_argument("--min-seconds", type=float, default=6.5)
    ap.add_argument("--plans", type=int, default=0, help="plans in flight (0 = two)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the inclusive rate and the cfg3 sub-record")
    ap.add_argument("--sub", default="cfg3", help="workload of the sub-record ('' = none)")
    ap.add_argument("--param", action="append", default=[], help="engine parameter name=value (development: A/B of a kernel route)")
    ap.add_argument("--strong-loci", type=int, default=10000, help="N > 1: records of the cfg4 VCF sharded over the ranks (0 = skip)")
    ap.add_argument("--strong-base", type=int, default=300, help="distinct loci of that world (tiled up to --strong-loci)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: this process has not touched the GPU (torch is not even imported yet);
        # it starts N fresh rank processes and relays their output and status
        raise SystemExit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: start one rank per GPU (or plain `python bench.py --gpus N`, "
                         "which starts them itself)" % (args.gpus, world))

    import torch
    n_dev = torch.cuda.device_count()           # (counting devices does not initialise the GPU)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    # RCCL wants a GPU per rank.  With fewer GPUs than ranks (a rehearsal of the N > 1 path on one card) the ranks share
    # the GPUs and the all-gather goes through gloo; VAPOR_BENCH_BACKEND forces either.
    backend = os.environ.get("VAPOR_BENCH_BACKEND", "nccl" if world <= n_dev else "gloo")
    local = local % n_dev
    torch.cuda.set_device(local)
    dist = None
    # (VAPOR_BENCH_FORCE_DIST=1: a process group even for one rank - RCCL with a world of one exercises the collective's
    # stream ordering against the library's streams on a one-GPU box)
    if world > 1 or os.environ.get("VAPOR_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl" and os.environ.get("VAPOR_BENCH_NCCL_EAGER") == "1":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        elif backend == "nccl":
            # (no device_id: with it torch creates the RCCL communicator - and its streams - at once, before the library makes
            # its own; HIP runs a process's streams on at most GPU_MAX_HW_QUEUES (4) hardware queues, the library's two plans
            # then share one and stop overlapping: 0.139 -> 0.235 ms per pass with not a single collective in the timed region,
            # nothing with --plans 1, nothing with GPU_MAX_HW_QUEUES=8 (profiles/r04_eager_communicator.txt).  Created at the
            # first collective it costs nothing.  VAPOR_BENCH_NCCL_EAGER=1 reproduces it.)
            dist.init_process_group("nccl", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll = torch.cuda.Stream()          # the collectives' stream; the library's kernels run on its own two streams

    from vapor_amd import workload as wl
    from vapor_amd.engine import Engine

    eng = Engine(local)
    for kv in args.param:
        eng.set_param(kv.split("=")[0], int(kv.split("=")[1]))
    spec = wl.WORKLOADS[args.workload]
    w = wl.make_workload(args.workload, seed=1000 + rank, **spec)
    # Two plans in flight: one plan's join runs beside the other's clean, and the tail of either kernel is filled by the other
    # plan's pass.  Measured on both shapes: cfg2 (4 000 pairs a pass) 0.193 -> 0.138 ms a pass, cfg3 (80 000 pairs, which
    # this line ran with one plan until the end of round 4 on the assumption that a long join keeps the chip busy by itself)
    # 3.68 -> 3.29 ms, +12 %; three or four plans lose on cfg2 (profiles/r04_remap_experiments.txt).
    n_plans = args.plans if args.plans > 0 else 2
    res = Resident(eng, w, wl, n_plans, torch, dist, world, coll)

    def barrier():
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # probe: how many passes make a step long enough
    for _ in range(2 * n_plans):
        res.one_pass()
    res.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_probe = 40
    for _ in range(n_probe):
        res.one_pass()
    torch.cuda.synchronize()
    t_pass = (time.perf_counter() - t0) / n_probe
    res.drain()
    inner = args.passes_per_step if args.passes_per_step > 0 else max(1, math.ceil(args.min_seconds / (max(args.steps, 1) * t_pass)))
    if dist is not None:
        t = torch.tensor([inner], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)     # every rank runs the same number of passes
        inner = int(t.item())

    res.set_step(inner)

    def step():
        res.begin_step()
        for j in range(inner):
            res.one_pass(j)
        res.end_step()

    for _ in range(args.warmup):
        step()
    res.drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    tm = res.drain(want_host=True)
    n_passes = args.steps * inner
    # every plan ran n_passes / n_plans passes (+-1); its timings are averages over them
    wts = [len(range(k, n_passes, n_plans)) for k in range(n_plans)]
    avg = {key: sum(t[0][key] * wt for t, wt in zip(tm, wts)) / max(sum(wts), 1) for key in ("join_ms", "clean_ms", "total_ms", "finish_ms")}
    for key in ("clean_workgroups_per_cu", "remap_in_clean"):      # (the plan's geometry, not a time)
        avg[key] = tm[0][0].get(key)
    last_rec = tm[(n_passes - 1) % n_plans][1]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if res.gathered is not None:
            # the last pass of every rank: slot inner - 1 of its share of the gathered step
            last_rec = res.gathered.view(world, inner, w.n_loci, 8)[:, inner - 1].cpu().numpy()

    # untimed: per-pair statistics once, for the algorithmic byte count and the oracle cross-check; and the
    # host-side finish on the same statistics must agree with what the device produced
    plan = res.plans[0]
    st = plan.run().copy()
    host_rec = wl.finish_workload(w, st)
    dev_rec = plan.run_loci().copy()
    ok = ~np.isnan(host_rec[:, 0])
    assert np.array_equal(np.isnan(dev_rec[:, 0]), ~ok) and np.array_equal(dev_rec[ok, 2], host_rec[ok, 2]) \
        and np.allclose(dev_rec[ok, :2], host_rec[ok, :2], rtol=0, atol=1e-9), "device / host finish mismatch"
    alg_bytes, cells = plan.algorithmic()
    n_rec = int(plan.record_counts().sum())
    n_pairs = len(w.pairs)
    launches = plan.timings()["join_launches"]
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    loci_s = w.n_loci * world * n_passes / elapsed
    cells_s = cells * world * n_passes / elapsed
    src_id = kernel_source_id()
    roof = roofline_record(args.workload, alg_bytes, n_rec, elapsed / n_passes * 1e3,
                           {"join": avg["join_ms"], "clean": avg["clean_ms"], "finish": avg["finish_ms"]},
                           {"join": res.alone["join_ms"], "clean": res.alone["clean_ms"]}, src_id)

    extras = {}
    cpu = None
    if rank == 0 and world == 1:
        if not args.no_extras:
            extras["inclusive"] = inclusive_rate(eng, w, wl)
            extras["pipeline"] = pipeline_rate()
            extras["pipeline_simulate_spans"] = pipeline_rate(span_dist="simulate")
            extras["files_path"] = files_rate()
            if args.sub and args.sub != args.workload:
                extras["sub"] = sub_record(eng, wl, args.sub, torch, cpu_seconds=0.0 if args.no_cpu else min(args.cpu_seconds, 10.0))
        if not args.no_cpu:
            cpu = cpu_baseline(w, st, n_pairs, args.cpu_seconds)
    strong = None
    # (VAPOR_BENCH_FORCE_STRONG=1 with VAPOR_BENCH_FORCE_DIST=1: the same leg with a world of one - on a one-GPU box the only way
    # to take the product's gather through RCCL: vapor_amd.dist on CUDA tensors, all_gather_object under nccl)
    if world > 1 or (dist is not None and os.environ.get("VAPOR_BENCH_FORCE_STRONG") == "1"):
        # N > 1: the CPU baseline on rank 0 after the timed region (the other ranks wait at the next barrier), and the
        # fixed-size workload north_star's multi-GPU configurations describe: configs[3]'s 10 000-record VCF sharded over
        # the ranks by estimated cost, scored through the product drivers, one gather of the scores (strong scaling)
        if rank == 0 and not args.no_cpu:
            cpu = cpu_baseline(w, st, n_pairs, args.cpu_seconds)
        barrier()
        if args.strong_loci > 0:
            strong = strong_record(eng, dist, backend, world, rank, args.strong_loci, args.strong_base, barrier, torch)

    if rank == 0:
        out = {
            "metric": "SV loci validated/sec",
            "value": round(loci_s, 3),
            "unit": "loci/s",
            "n_gpus": args.gpus,
            "ranks_seen": dist.get_world_size() if dist is not None else 1,
            "backend": (backend if dist is not None else None),
            "gpus_visible": n_dev,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "%s: %d loci/GPU x %d reads (%d bp) x 2 allele windows (%d bp), k=10, types %s; ONE batch packed in HBM before the "
                                   "timed region and re-scored every pass (its 22 MB of planes and records stay in the 256 MB Infinity Cache: the "
                                   "kernels are vector-issue-bound, see roofline.valu_frac)"
                                   % (args.workload, w.n_loci, spec["reads_per_locus"], spec["read_len"],
                                      spec["allele_len"], "/".join(sorted(set(w.svtypes)))),
                       "pairs_per_pass_per_gpu": n_pairs, "passes_per_step": inner, "plans_in_flight": n_plans,
                       # (N > 1: the per-locus records of all passes of a step travel in ONE all-gather per step since round 3;
                       # rounds 1-2 gathered behind every pass - their SCALE / BENCH multi-GPU lines are not comparable with these)
                       "gathers_per_step": (1 if dist is not None else 0),
                       "parallelism": "loci sharded over %d GPU(s)" % world},
            "timed_region_s": round(elapsed, 4),
            "ms_per_pass": round(elapsed / n_passes * 1e3, 5),
            "cells_per_s": round(cells_s, 1),
            "hits_per_pass": int(st[:, 0].sum()),
            "records_per_pass": n_rec,
            "loci_with_scores": int(np.isfinite(np.asarray(last_rec).reshape(-1, 8)[:, 0]).sum()),
            "kernel_ms": {"join": round(avg["join_ms"], 4), "clean": round(avg["clean_ms"], 4), "finish": round(avg["finish_ms"], 4),
                          "device_total": round(avg["total_ms"], 4), "join_launches": launches,
                          "clean_workgroups_per_cu": avg.get("clean_workgroups_per_cu"), "remap_in_clean": avg.get("remap_in_clean"),
                          "alone": {"join": round(res.alone["join_ms"], 4), "clean": round(res.alone["clean_ms"], 4)},
                          "note": "a read is joined once against a window and the alleles derived from it (shared joins); the pairs' records are cut out "
                                  "of the shared dot plots by the clean workgroups themselves (this workload: inside `clean`) or, for plans of many "
                                  "rounds of clean workgroups, by remap_kernel (inside `join`); join / clean / finish are HIP-event INTERVALS inside the two-plan overlap (the two plans' kernels "
                                  "share the CUs and their intervals overlap: their sum over the passes exceeds the step), not serial "
                                  "costs; `alone` is each kernel by itself on an otherwise idle device"},
            "upload_pack_s": round(res.upload_s, 4),
            # (roofline_record: the kernel with the largest share of GPU time, its HBM fraction on the algorithmic bytes, the
            # fabric bytes the counters saw, and the fraction of the vector-issue roof the kernels really sit under)
            "roofline": roof,
            "kernel_source_id": src_id,
            "cpu_baseline": cpu,
        }
        out["gpu_max_hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "unset (HIP default: 4)")
        if strong is not None:
            # N > 1: the fixed call set sharded over the ranks (north_star's multi-GPU configuration) leads the line, right behind
            # the headline fields; `value` stays the resident rate so that the driver's per-N values compare like with like
            lead = {"strong_value": strong.get("value"), "strong_unit": "loci/s (configs[3] call set, one `vapor vcf` run over all ranks)",
                    "strong_speedup_over_one_rank": strong.get("speedup_over_one_rank"), "strong_tables_equal": strong.get("tables_equal"),
                    "strong": strong}
            items = list(out.items())
            cut = [k for k, _v in items].index("ranks_seen") + 1
            out = dict(items[:cut] + list(lead.items()) + items[cut:])
        out.update(extras)
        if "pipeline" in extras and "value" in extras["pipeline"]:
            out["pipeline_value"] = extras["pipeline"]["value"]    # the drivers end to end, one process (host-bound)
        if "files_path" in extras and "value" in extras["files_path"]:
            out["files_value"] = extras["files_path"]["value"]     # ... and from FASTA/BAM files, reads extracted on the device
        if "inclusive" in extras:
            # the same batch when every pass also pays upload + packing, the window self plots, planning and a blocking pass
            out["inclusive_value"] = extras["inclusive"]["value"]
        sub = extras.get("sub")
        if sub and sub.get("cpu_baseline") and sub["cpu_baseline"].get("reference"):
            # north_star's target is stated on 1 000 loci x 40x (configs[2] = cfg3): GPU over the reference's Cython path
            ref = sub["cpu_baseline"]["reference"]
            est = ref["estimated_reference_cython_loci_per_s_here"]
            out["target_200x"] = {"config": "cfg3", "gpu_loci_per_s": sub["value"], "cpu_port_loci_per_s": sub["cpu_baseline"]["value"],
                                  "port_over_reference_cython": ref["port_over_reference_cython"],
                                  "estimated_reference_cython_loci_per_s": est,
                                  "gpu_over_reference": round(sub["value"] / est, 1) if est > 0 else None,
                                  "gpu_over_port": round(sub["value"] / sub["cpu_baseline"]["value"], 1),
                                  "met": bool(est > 0 and sub["value"] / est >= 200.0)}
        print(json.dumps(out))
    res.close()
    if dist is not None:
        dist.destroy_process_group()


def spawn_ranks(n: int) -> int:
    """Starts `n` ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and
    returns their exit status.  Called before anything in this process has touched the GPU; the ranks are children, never
    an exec of this process."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL between processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def strong_record(eng, dist, backend, world, rank, n_loci, base, barrier, torch):
    """Strong scaling on the workload BASELINE.json configs[3] names: ONE `vapor vcf` call set of `n_loci` records (DEL / INV /
    INS and a third complex records; 40 reads of 15 kb per locus; a tiled in-memory world every rank builds from the same
    seed) sharded over the ranks by estimated cost (cli.job_cost, greedy LPT), each rank scoring its share through the
    reference-named drivers on its own GPU, one all-gather of the score vectors (vapor_amd.dist over this process group:
    RCCL when every rank owns a GPU), rank 0 writing the rows.  Reported: loci/s over the slowest rank's wall time, every
    rank's busy time, and the table's sha256 against the same run on rank 0 alone."""
    import contextlib
    import hashlib
    import io
    import tempfile
    from vapor_amd import cli, pipeline, seqio
    from vapor_amd import dist as vdist
    from vapor_amd import workload as wl
    os.environ.setdefault("VAPOR_QC_SEED", "7")
    try:
        t0 = time.perf_counter()
        big, text, n_records = wl.at_size_input("cfg4", n_loci, base)
        t_world = time.perf_counter() - t0
        tmp = tempfile.mkdtemp(prefix="vapor_strong_r%d_" % rank)
        src = os.path.join(tmp, "in.vcf")
        open(src, "w").write(text)
        seqio.set_backend(seqio.MemorySamtools(big))
        pipeline.set_engine(eng)

        def run():
            with contextlib.redirect_stdout(io.StringIO()):
                vcf_list, _rec_hash = cli.vcf_list_readin(src)
                jobs = cli.vcf_jobs(vcf_list, 3, "x.bam", "ref.fa", tmp + "/", "s")
                scores = cli.score_jobs(jobs, 2048, None)
                rows = cli.output_rows([[j.key] for j in jobs], scores)[0]
            return jobs, rows
        vdist.adopt_process_group(backend)
        try:
            run()                                        # (warm: allocators, the second context of the chunk threads)
            barrier()
            t0 = time.perf_counter()
            jobs, rows = run()
            t_mine = time.perf_counter() - t0
            busy = dict(cli.last_timing)
            barrier()
            elapsed = time.perf_counter() - t0
        finally:
            vdist.release_process_group()
        dev = torch.device("cuda") if backend == "nccl" else torch.device("cpu")
        mine = torch.tensor([elapsed, t_mine, busy.get("score_s", 0.0), busy.get("gather_s", 0.0), float(busy.get("loci", 0)),
                             float(busy.get("cost", 0.0))], dtype=torch.float64, device=dev)
        allv = torch.empty(world * 6, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allv, mine)
        allv = allv.cpu().numpy().reshape(world, 6)
        sha = hashlib.sha256("\n".join(rows).encode()).hexdigest()
        out = None
        if rank == 0:
            t0 = time.perf_counter()
            _j, rows1 = run()                            # the same call set on this rank alone (vapor_amd.dist sees one process)
            t_one = time.perf_counter() - t0
            sha1 = hashlib.sha256("\n".join(rows1).encode()).hexdigest()
            wall = float(allv[:, 0].max())
            out = {"workload": "cfg4: `vapor vcf` call set of %d records (DEL / INV / INS, a third complex: DUP_INV, DISDUP, DEL_INV, Other=), "
                               "40 reads of 15 kb per locus, %d distinct loci tiled; in-memory world, figures off" % (n_records, base),
                   "scaling": "strong", "value": round(n_records / wall, 1), "unit": "loci/s", "seconds": round(wall, 4),
                   "ranks": world, "backend": backend, "shares": "greedy LPT on cli.job_cost",
                   "per_rank": [{"loci": int(r[4]), "estimated_cost_us": round(float(r[5]), 1), "busy_s": round(float(r[2]), 4),
                                 "gather_s": round(float(r[3]), 4), "wall_s": round(float(r[1]), 4)} for r in allv],
                   "busy_spread": round(float(allv[:, 2].max() / max(allv[:, 2].mean(), 1e-12)), 3),
                   "one_rank": {"value": round(n_records / t_one, 1), "seconds": round(t_one, 4)},
                   "speedup_over_one_rank": round(t_one / wall, 3),
                   "rows_sha256": sha, "rows_sha256_one_rank": sha1, "tables_equal": sha == sha1,
                   "world_build_s": round(t_world, 1)}
        seqio.set_backend(None)
        barrier()
        return out
    except Exception as e:      # noqa: BLE001 - a side record must not take the headline down
        import traceback
        return {"error": "%s: %s" % (type(e).__name__, e), "trace": traceback.format_exc()[-1200:]}


def inclusive_rate(eng, w, wl, reps: int = 6):
    """Everything a fresh batch costs: upload + packing of the bytes (the alt windows are descriptors: the device assembles
    them), the self dot plots window_size_refine looks at (k = 10, every ref and alt window), planning, join -> clean -> finish,
    records back on the host.  Two figures: one batch at a time (`ms_per_batch`, with its parts), and the rate with two batches
    in flight - a thread and a library context each, as cli.score_jobs keeps two chunks of a run in flight - which is `value`."""
    import threading
    from vapor_amd import _lib as L
    from vapor_amd.engine import Engine
    allele_idx = sorted(set(int(x) for x in w.pairs["seq2"]))
    selfp = np.zeros(len(allele_idx), dtype=L.PAIR_DTYPE)
    selfp["seq1"] = selfp["seq2"] = allele_idx
    selfp["k"] = 10
    table = wl.read_table(w)

    def batch(e, parts=None):
        t0 = time.perf_counter()
        ss = w.upload(e)
        t1 = time.perf_counter()
        pw = e.plan(ss, selfp)
        stw = pw.run()
        assert int(stw[:, 15].min()) == 0 and int(stw[:, 0].min()) > 0
        t2 = time.perf_counter()
        p = e.plan(ss, w.pairs)
        p.set_reads(table, w.n_loci)
        t3 = time.perf_counter()
        rec = p.run_loci()
        t4 = time.perf_counter()
        pw.close(); p.close(); ss.close()
        if parts is not None:
            parts += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
        return t4 - t0, rec

    times, parts = [], np.zeros(4)
    batch(eng)                                  # (the first repetition warms allocators up)
    for _ in range(reps):
        times.append(batch(eng, parts)[0])
    t_one = float(np.median(times))
    # several batches in flight: as many as cli.score_jobs keeps chunks of a run in flight (VAPOR_CHUNKS_IN_FLIGHT, default 3).
    # A run keeps its contexts for all of its chunks, so the figure wanted is the steady one: every context is warmed with a
    # few batches, the timed region lasts `in_flight_batches` batches per thread (about half a second), and the line carries
    # the region's first eight batches per thread beside it (new contexts: staging buffers and device pools still growing -
    # what the eight-batch region of rounds 3 and 4 measured, tools/inclusive_probe.py).
    n_fl = max(2, int(os.environ.get("VAPOR_CHUNKS_IN_FLIGHT", "3")))
    more = [Engine(eng.device) for _ in range(n_fl - 1)]
    for e in more:
        batch(e)
    n_each = max(reps, int(os.environ.get("VAPOR_BENCH_INFLIGHT_BATCHES", "150")))
    recs = [None] * n_fl
    stamps = [[] for _ in range(n_fl)]

    def worker(k, e):
        for _ in range(n_each):
            recs[k] = batch(e)[1].copy()
            stamps[k].append(time.perf_counter())
    th = [threading.Thread(target=worker, args=(k, e)) for k, e in enumerate([eng] + more)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    t_two = (time.perf_counter() - t0) / (n_fl * n_each)                     # the whole region, its start included
    n_first = min(8, n_each)
    t_first = (max(st[n_first - 1] for st in stamps) - t0) / (n_fl * n_first)
    half = n_each // 2
    t_late = ((max(st[-1] for st in stamps) - min(st[half - 1] for st in stamps)) / (n_fl * (n_each - half))) if half >= 1 and n_each > half else t_two
    same = all(np.array_equal(np.isnan(recs[0]), np.isnan(r)) and np.array_equal(recs[0][~np.isnan(recs[0])], r[~np.isnan(r)]) for r in recs[1:])
    for e in more:
        e.close()
    return {"value": round(w.n_loci / t_two, 2), "unit": "loci/s", "batches_in_flight": n_fl, "ms_per_batch_in_flight": round(t_two * 1e3, 3),
            "in_flight_batches_per_thread": n_each,
            "first_8_batches": {"value": round(w.n_loci / t_first, 2), "ms_per_batch": round(t_first * 1e3, 3)},
            "second_half": {"value": round(w.n_loci / t_late, 2), "ms_per_batch": round(t_late * 1e3, 3)},
            "one_at_a_time": {"value": round(w.n_loci / t_one, 2), "ms_per_batch": round(t_one * 1e3, 3),
                              "ms": dict(zip(("upload_pack", "window_selfplots", "plan", "join_clean_finish"), [round(x / reps * 1e3, 3) for x in parts]))},
            "records_equal": bool(same),
            "includes": "host bytes -> pinned staging -> H2D -> pack_kernel (the alt windows travel as segment descriptors and are assembled by "
                        "derive_kernel); %d self dot plots (k = 10) for window_size_refine's integer part; vapor_plan_create + set_reads; one "
                        "blocking join -> remap -> clean -> finish; records to host.  `value`: %d batches on each of %d threads, a library context "
                        "each (the way cli.score_jobs keeps chunks of a run in flight); `one_at_a_time`: median of %d batches" % (len(allele_idx), n_each, n_fl, reps)}


def pipeline_rate(n_loci: int = 400, span_dist=None):
    """The product path beside the kernel rate: the reference-named drivers (vapor_vali/vapor:334-367, SF:1701-1933) over a
    seeded in-memory world - BED parsing, read extraction and trimming, allele strings, window_size_refine, dot plots and
    scores on the device, result rows - in this process, figures off.  Host-bound: this is the rate a `vapor bed` run sees
    per process (tools/bench_pipeline.py, tools/run_at_size.py measure it at size and from FASTA/BAM files)."""
    import tempfile
    try:
        from vapor_amd import cli, pipeline, seqio, synth
        from vapor_amd import simple_function as SF
        from vapor_amd.finish import result_organize_ins
        # span_dist = "simulate": spans drawn from the reference's simulated truth sets (50 bp - 100 kb, ~7 % >= 10 kb: those loci
        # take the drivers' junction-window branch, the 5-10 kb ones windows the 9.5 kb reads barely span) instead of uniformly
        w = synth.make_world(seed=11, n_loci=n_loci, svtypes=("DEL", "DEL", "INV", "INS"), span_range=(100, 4000), read_len=9500, n_reads=20,
                             span_dist=span_dist)
        tmp = tempfile.mkdtemp(prefix="vapor_bench_")
        bed = os.path.join(tmp, "in.bed")
        open(bed, "w").write(synth.bed_text(w))
        seqio.set_backend(seqio.MemorySamtools(w))
        try:
            bed_info = cli.bed_info_readin(bed, tmp)

            def run():
                jobs = cli.bed_jobs(bed_info, 3, "x.bam", "ref.fa", tmp + "/", "s")
                scores = cli.score_jobs(jobs, 2048, None)
                return cli.output_rows([j.key.split(':') + [j.row_prefix] for j in jobs], scores)[0]   # (as cli.main writes them)
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):
                run()
                best, rows = 1e9, []
                for _ in range(3):
                    t0 = time.perf_counter(); rows = run(); best = min(best, time.perf_counter() - t0)
        finally:
            seqio.set_backend(None)
        scored = sum(1 for r in rows if "\tNA" not in r)
        return {"value": round(len(rows) / best, 1), "unit": "loci/s", "loci": len(rows), "loci_with_scores": scored,
                "spans": ("simulate/Structural_Variants_het distribution (vapor_amd/data/simulate_spans.json)" if span_dist else "uniform 100 - 4 000 bp"),
                "includes": "cli.bed_jobs -> drivers -> pipeline.run_batch (one sequence set and plan per round) -> result rows; in-memory "
                            "world of %d DEL/INV/INS loci x 20 reads of 9.5 kb, one process, figures off; best of 3" % n_loci}
    except Exception as e:      # noqa: BLE001 - a side record must not take the headline down
        return {"error": "%s: %s" % (type(e).__name__, e)}


def _quota_cores() -> int:
    try:
        from vapor_amd import pipeline
        return int(pipeline._usable_cores())
    except Exception:       # noqa: BLE001
        return len(os.sched_getaffinity(0))


def files_rate(n_loci: int = 500, repeat: int = 8):
    """the same drivers from FASTA + BAM FILES through the product CLI in this process - what a `vapor bed` run
    sees - with the read extraction on the device (vapor_bam_chop_device: the regions' BGZF blocks cross the link compressed, one
    wavefront inflates a block, one walks a region's records) and, beside it, on the host's prefetch threads (vapor_bam_chop);
    the two tables must be byte-identical.  The reference runs a samtools process per locus here (SF:340)."""
    import contextlib
    import hashlib
    import io
    import tempfile
    try:
        from vapor_amd import cli, synth
        w = synth.make_world(seed=11, n_loci=n_loci, svtypes=("DEL", "DEL", "INV", "INS"), span_range=(100, 4000), read_len=9500, n_reads=20)
        for c in w.reads:
            w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
        tmp = tempfile.mkdtemp(prefix="vapor_bench_files_")
        fa, bam = synth.write_world_files(w, tmp, block_size=0xFF00)
        bed = os.path.join(tmp, "in.bed")
        open(bed, "w").write(synth.bed_text(w) * repeat)        # (the file's loci `repeat` times over: a run of several chunks from a file that is written in seconds)
        n_file = n_loci
        n_loci *= repeat
        rec = {"unit": "loci/s", "loci": n_loci, "distinct_loci": n_file, "bam_mb": round(os.path.getsize(bam) / 1e6, 1)}
        shas = {}
        was = os.environ.get("VAPOR_BAM_DEVICE")
        try:
            for name, dev in (("device", "1"), ("host", "0")):
                os.environ["VAPOR_BAM_DEVICE"] = dev
                out = os.path.join(tmp, "o_%s.vapor" % name)
                args = ["bed", "--sv-input", bed, "--reference", fa, "--pacbio-input", bam, "--output-path", tmp + "/f", "--output-file", out, "--no-figures"]
                best = 1e9
                with contextlib.redirect_stdout(io.StringIO()):
                    cli.main(args)                                   # (warm: engines, pools, page cache)
                    for _ in range(3):
                        t0 = time.perf_counter(); cli.main(args); best = min(best, time.perf_counter() - t0)
                shas[name] = hashlib.sha256(open(out, "rb").read()).hexdigest()[:16]
                rec["value" if name == "device" else "host_extraction_value"] = round(n_loci / best, 1)
        finally:
            if was is None:
                os.environ.pop("VAPOR_BAM_DEVICE", None)
            else:
                os.environ["VAPOR_BAM_DEVICE"] = was
        # the extraction by itself: every locus's region (locus +- 500) through vapor_bam_chop_device and through the host reader's
        # chop_many; the inflate kernel's duration between two HIP events on its stream (vapor_bam_last_stats)
        try:
            import numpy as np
            from vapor_amd import pipeline, seqio
            be = seqio.InProcessBam()
            eng = pipeline.get_engine()
            rows = [l.split("\t") for l in synth.bed_text(w).strip().splitlines()]
            chroms = [r[0] for r in rows]
            st = np.asarray([max(int(r[1]) - 500, 1) for r in rows], dtype=np.int64)
            en = np.asarray([int(r[2]) + 500 for r in rows], dtype=np.int64)
            fl = np.full(len(rows), 500, dtype=np.int64)
            best = {"device": 1e9, "host": 1e9}
            kept = {}
            stats = None
            for _ in range(4):
                t0 = time.perf_counter(); got = be.chop_many_device(eng, bam, chroms, st, en, fl); dt = time.perf_counter() - t0
                for bt in got[5]:
                    bt.close()
                kept["device"] = (got[0].tolist(), got[3].tolist())
                if dt < best["device"]:
                    best["device"], stats = dt, eng.bam_last_stats()
                t0 = time.perf_counter(); got = be.chop_many(bam, chroms, st, en, fl); best["host"] = min(best["host"], time.perf_counter() - t0)
                kept["host"] = (got[0].tolist(), got[3].tolist())
            rec["extraction_alone"] = {
                "device_loci_per_s": round(len(rows) / best["device"], 1), "host_loci_per_s": round(len(rows) / best["host"], 1),
                "kept_reads_and_miss_bp_equal": kept["device"] == kept["host"], "blocks": stats["blocks"],
                "compressed_mb": round(stats["compressed_bytes"] / 1e6, 1), "inflated_mb": round(stats["inflated_bytes"] / 1e6, 1),
                "inflate_kernel_ms": round(stats["inflate_ms"], 3),
                "inflate_gb_per_s": round(stats["inflated_bytes"] / 1e9 / (stats["inflate_ms"] / 1e3), 1) if stats["inflate_ms"] > 0 else None,
                "note": "bgzf_inflate_kernel: one BGZF block per wavefront, twenty in flight per CU; bound by dependent table lookups of single "
                        "lanes, not by bytes (DESIGN.md 4.5) - no roofline fraction is claimed for it"}
        except Exception as e:      # noqa: BLE001
            rec["extraction_alone"] = {"error": "%s: %s" % (type(e).__name__, e)}
        rec["tables_equal"] = shas["device"] == shas["host"]
        rec["table_sha"] = shas["device"]
        rec["includes"] = ("cli.main bed from FASTA/.fai + BAM/.bai files (64 KB BGZF blocks, qualities 0xFF; the BED lists the file's loci %d times), one warm process, "
                           "figures off, best of 3; " % repeat +
                           "value = reads by device address (vapor_bam_chop_device), host_extraction_value = the host reader on the %d cores of this job's CPU quota"
                           % _quota_cores())
        return rec
    except Exception as e:      # noqa: BLE001 - a side record must not take the headline down
        return {"error": "%s: %s" % (type(e).__name__, e)}


def sub_record(eng, wl, name, torch, passes: int = 12, cpu_seconds: float = 0.0):
    """The resident measurement on another BASELINE shape, same build, same run: two plans in flight like the headline
    (the clean and finish kernels of one pass overlap the other plan's join), and one plan alone for comparison."""
    spec = wl.WORKLOADS[name]
    w = wl.make_workload(name, seed=3000, **spec)
    ss = w.upload(eng)
    plans = []
    for _ in range(2):
        p = eng.plan(ss, w.pairs)
        p.set_reads(wl.read_table(w), w.n_loci)
        p.run_loci(want_host=False)
        plans.append(p)

    def timed(ps):
        for i in range(2 * len(ps)):
            ps[i % len(ps)].run_loci_async()
        for p in ps:
            p.sync(want_host=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(passes):
            ps[i % len(ps)].run_loci_async()
        for p in ps:
            p.sync(want_host=False)
        return time.perf_counter() - t0, ps[0].timings()

    dt1, tm1 = timed(plans[:1])
    dt2, tm2 = timed(plans)
    st = plans[0].run().copy()
    alg, cells = plans[0].algorithmic()
    out = {"workload": "%s: %d loci x %d reads (%d bp) x 2 allele windows (%d bp), k=10, types %s; resident in HBM"
                       % (name, w.n_loci, spec["reads_per_locus"], spec["read_len"], spec["allele_len"], "/".join(sorted(set(w.svtypes)))),
           "value": round(w.n_loci * passes / dt2, 2), "unit": "loci/s", "passes": passes, "plans_in_flight": 2,
           "ms_per_pass": round(dt2 / passes * 1e3, 4), "cells_per_s": round(cells * passes / dt2, 1),
           "one_plan": {"value": round(w.n_loci * passes / dt1, 2), "ms_per_pass": round(dt1 / passes * 1e3, 4),
                        "kernel_ms": {"join": round(tm1["join_ms"], 4), "clean": round(tm1["clean_ms"], 4), "finish": round(tm1["finish_ms"], 4)}},
           "kernel_ms": {"join": round(tm2["join_ms"], 4), "clean": round(tm2["clean_ms"], 4), "finish": round(tm2["finish_ms"], 4)},
           # the same first-class record as the headline's (one basis: the two-plan pass and its live intervals)
           "roofline": roofline_record(name, alg, int(plans[0].record_counts().sum()), dt2 / passes * 1e3,
                                       {"join": tm2["join_ms"], "clean": tm2["clean_ms"], "finish": tm2["finish_ms"]},
                                       {"join": tm1["join_ms"], "clean": tm1["clean_ms"]}, kernel_source_id())}
    out["roofline_frac"] = out["roofline"]["frac"]
    for p in plans:
        p.close()
    ss.close()
    if cpu_seconds > 0:
        # the CPU port beside it on a bounded sample of THIS shape, and the reference : port ratio measured on this shape
        out["cpu_baseline"] = cpu_baseline(w, st, len(w.pairs), cpu_seconds, ratio_file="r03_cpu_ratio_%s.json" % name)
    return out


def host_cores() -> dict:
    """The box's core count and the share of it this process may use (north_star: 'core count stated')."""
    from vapor_amd import pipeline
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count()
    # (the affinity mask cut to the container's CPU quota: what the box really gives this job)
    return {"logical": os.cpu_count(), "affinity": affinity, "usable_by_this_process": pipeline._usable_cores()}


def cpu_baseline(w, st, n_pairs, seconds, ratio_file="r03_cpu_ratio_cfg2.json"):
    """The CPU port timed through the same C ABI: oracle/_build/libvapor_cpu.so (oracle/cpu_twin.cpp on vapor_oracle.c)
    exports include/vapor_hip.h's symbols; a bounded sample of the batch's pairs is run through vapor_plan_run there,
    chunk by chunk, one thread, and every record is compared with the GPU's."""
    import ctypes
    from oracle import oracle as orc
    from vapor_amd import _lib as L
    orc.build()
    tw = L.bind(ctypes.CDLL(orc.build_twin()))
    ctx = ctypes.c_void_p()
    assert tw.vapor_init(0, ctypes.byref(ctx)) == 0
    done = 0
    c_cells = 0
    ct = 0.0
    # A STRATIFIED sample: whole loci (all reads of a locus, both windows) taken in golden-ratio order over the batch, so that
    # whenever the time runs out the loci looked at are spread evenly over the whole batch and over its type cycle (VERDICT r04:
    # the first N pairs of cfg3 were 8 % of the batch from its front)
    first_pair = 2 * np.searchsorted(w.read_locus, np.arange(w.n_loci + 1))
    seen = set()
    loci_done = []
    j = 0
    while len(seen) < w.n_loci and ct < seconds:
        li = int((j * 0.6180339887498949) % 1.0 * w.n_loci)
        j += 1
        if li in seen:
            continue
        seen.add(li)
        lo, hi = int(first_pair[li]), int(first_pair[li + 1])
        if hi <= lo:
            continue
        pr = w.pairs[lo:hi].copy()
        ids = sorted(set(pr["seq1"].tolist()) | set(pr["seq2"].tolist()))
        remap = {s: t for t, s in enumerate(ids)}
        raw = [w.seqs[s].encode("ascii") for s in ids]
        ptrs = (ctypes.c_void_p * len(raw))(*[ctypes.cast(ctypes.c_char_p(x), ctypes.c_void_p).value for x in raw])
        lens = np.array([len(x) for x in raw], dtype=np.int32)
        pr["seq1"] = [remap[s] for s in pr["seq1"].tolist()]
        pr["seq2"] = [remap[s] for s in pr["seq2"].tolist()]
        out = np.zeros((len(pr), 16), dtype=np.int64)
        ss, pl = ctypes.c_void_p(), ctypes.c_void_p()
        t0 = time.perf_counter()
        assert tw.vapor_seqset_create_ptrs(ctx, len(raw), ptrs, L.ptr(lens, ctypes.c_int32), None, None, ctypes.byref(ss)) == 0
        assert tw.vapor_plan_create(ctx, ss, len(pr), pr.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pl)) == 0
        assert tw.vapor_plan_run(pl, L.ptr(out, ctypes.c_int64)) == 0
        ct += time.perf_counter() - t0
        tw.vapor_plan_destroy(pl); tw.vapor_seqset_destroy(ss)
        # statistics 0-12: counts, cleaning, and the directed statistics (float64 restatement there, integers here)
        assert np.array_equal(out[:, :13], st[lo:hi, :13]), "GPU / CPU-twin mismatch in locus %d (pairs %d..%d)" % (li, lo, hi)
        for q in w.pairs[lo:hi]:
            c_cells += len(w.seqs[q["seq1"]]) * (len(w.seqs[q["seq2"]]) - int(q["off2"]))
        done += hi - lo
        loci_done.append(li)
    tw.vapor_destroy(ctx)
    pairs_per_locus = n_pairs / w.n_loci
    cpu = {"value": round(done / pairs_per_locus / ct, 4), "unit": "loci/s", "cores": 1, "host_cores": host_cores(), "kind": "port",
           "sample": "%d of %d (read, allele) dot plots of the same batch - %d whole loci spread evenly over the batch (golden-ratio "
                     "order; types %s) - through libvapor_cpu.so, the CPU oracle behind the same C ABI (oracle/cpu_twin.cpp + "
                     "vapor_oracle.c: fill, C1/C2 clean, counts, directed statistics; gcc -O2, 1 thread, %.1f s); every record checked "
                     "equal to the GPU's" % (done, n_pairs, len(loci_done), "/".join("%s %d" % (t, sum(1 for li in loci_done if w.svtypes[li] == t))
                                                                                    for t in sorted(set(w.svtypes))), ct),
           "loci_checked": len(loci_done),
           "cells_per_s": round(c_cells / ct, 1)}
    rj = _load_json(ratio_file)
    if rj:
        # the reference itself cannot travel to the GPU box: its rate relative to this port was measured in the
        # development container on the same shape (tools/cpu_ratio.py)
        cpu["reference"] = {"port_over_reference_cython": round(rj["port_over_reference_cython"], 1),
                            "port_over_reference_python": round(rj["port_over_reference_python"], 1),
                            "reference_cython_loci_per_s_there": round(rj["reference_cython_loci_per_s"], 4),
                            "estimated_reference_cython_loci_per_s_here": round(cpu["value"] / rj["port_over_reference_cython"], 4),
                            "provenance": "profiles/" + ratio_file + ": " + rj["note"] + "; " + rj["shape"]}
    return cpu


if __name__ == "__main__":
    main()
