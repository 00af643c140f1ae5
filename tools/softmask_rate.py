"""The cfg2 batch (100 DEL/TANDUP loci x 20 reads of 10 kb x 20 kb windows, k = 10) as a run on a soft-masked genome sees
it (GPU box): allele windows with lower-case stretches (2 % of the bases, stretches of 20-300), and optionally N in the reads.
The pairs are laid out as vapor_amd/pipeline.py lays them out: abs_dis_m1b upper-cases ref and alt (SF:183-184), so a DEL read
is scored on the upper-cased windows (C1) AND on the windows as they are (C2) - four dot plots instead of two; a TANDUP read
on the windows as they are (directed distances).  Prints per variant: pairs, launches by symbol mode, dots, records, kernel
times of a blocking pass (median of 20).  usage: python tools/softmask_rate.py [n_frac ...]   (default 0 0.01)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vapor_amd import _lib as L
from vapor_amd import synth, workload as wl
from vapor_amd.engine import Engine


def soften(rng, s, frac):
    b = bytearray(s.encode())
    want = int(len(b) * frac)
    done = 0
    while done < want:
        n = int(rng.integers(20, 300))
        a = int(rng.integers(0, len(b) - n))
        b[a:a + n] = bytes(b[a:a + n]).lower()
        done += n
    return b.decode()


def with_n(rng, s, frac):
    if frac <= 0:
        return s
    b = bytearray(s.encode())
    for a in rng.integers(0, len(b), int(len(b) * frac)):
        b[a] = ord("N")
    return b.decode()


def variant(soft, n_frac, seed=1000):
    spec = wl.WORKLOADS["cfg2"]
    w = wl.make_workload("cfg2", seed=seed, **spec)
    rng = np.random.default_rng(seed + 7)
    rpl = spec["reads_per_locus"]
    seqs, upper, rows = [], [], []
    for li in range(w.n_loci):
        base = li * (2 + rpl)
        ref, alt = w.seqs[base], w.seqs[base + 1]
        if soft:
            ref, alt = soften(rng, ref, soft), soften(rng, alt, soft)
        ri = len(seqs)
        seqs += [ref, alt]; upper += [False, False]
        is_del = w.svtypes[li] == "DEL"
        ui = ri
        if is_del and soft:
            ui = len(seqs)
            seqs += [ref, alt]; upper += [True, True]
        for r in range(rpl):
            q = len(seqs)
            seqs.append(with_n(rng, w.seqs[base + 2 + r], n_frac)); upper.append(False)
            if is_del and ui != ri:
                rows += [(q, ui, 0, 10, L.PF_C1), (q, ui + 1, 0, 10, L.PF_C1), (q, ri, 0, 10, L.PF_C2), (q, ri + 1, 0, 10, L.PF_C2)]
            elif is_del:
                rows += [(q, ri, 0, 10, L.PF_C1 | L.PF_C2), (q, ri + 1, 0, 10, L.PF_C1 | L.PF_C2)]
            else:
                rows += [(q, ri, 0, 10, L.PF_C1 | L.PF_DIR), (q, ri + 1, 0, 10, L.PF_C1 | L.PF_DIR)]
    return seqs, upper, rows


def main():
    fracs = [float(a) for a in sys.argv[1:]] or [0.0, 0.01]
    eng = Engine(0)
    base_ms = None
    for name, soft, nf in [("plain", 0.0, 0.0)] + [("soft-masked 2 %%, N in reads %.3f" % f, 0.02, f) for f in fracs]:
        seqs, upper, rows = variant(soft, nf)
        ss = eng.seqset(seqs, upper)
        plan = eng.plan(ss, eng.make_pairs(rows))
        for _ in range(3):
            plan.run()
        tj, tc = [], []
        for _ in range(20):
            plan.run()
            t = plan.timings()
            tj.append(t["join_ms"]); tc.append(t["clean_ms"])
        st = plan.run()
        rec = plan.record_counts()
        ms = float(np.median(tj) + np.median(tc))
        base_ms = base_ms or ms
        print("%-36s pairs %5d  join launches %d  dots %9d  records %8d  join %.4f + clean %.4f = %.4f ms  (%.2f x plain, %.1f ns per pair)"
              % (name, len(rows), t["join_launches"], int(st[:, 0].sum()), int(rec.sum()), np.median(tj), np.median(tc), ms, ms / base_ms,
                 ms * 1e6 / len(rows)), flush=True)
        plan.close(); ss.close()


if __name__ == "__main__":
    main()
