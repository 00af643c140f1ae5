"""Randomised parity sweep of the HIP path against the CPU oracle (GPU box): odd lengths, every k, allele slices,
soft-masked and N-containing sequences, low-complexity repeats, reads that are exact copies (32-dot runs), alleles
around the tile size.  usage: fuzz_parity.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vapor_amd import synth
from vapor_amd.engine import Engine
from oracle import oracle as orc

def run(budget: float = 60.0, seed: int = 1, engine=None) -> str:
    orc.build()
    rng = np.random.default_rng(seed)
    eng = engine if engine is not None else Engine(0)
    t_end = time.time() + budget
    n_pairs = n_batches = n_dir = 0


    def spoil(s, rng):
        """lower-case stretches, N's, a repeat"""
        b = bytearray(s.encode())
        mode = rng.integers(0, 6)
        if mode == 1 and len(b) > 40:
            a = rng.integers(0, len(b) - 30); b[a:a + 25] = bytes(b[a:a + 25]).lower()
        elif mode == 2 and len(b) > 10:
            for _ in range(rng.integers(1, 4)):
                b[rng.integers(0, len(b))] = ord("N")
        elif mode == 3 and len(b) > 200:
            a = rng.integers(0, len(b) - 150); unit = bytes(b[a:a + rng.integers(1, 7)]); b[a:a + 120] = (unit * 120)[:120]
        elif mode == 4 and len(b) > 10:
            b[rng.integers(0, len(b))] = ord("n")
        return b.decode()


    while time.time() < t_end:
        seqs, rows = [], []
        for _ in range(int(rng.integers(4, 14))):
            la = int(rng.choice([rng.integers(12, 200), rng.integers(200, 3000), rng.integers(3000, 9000), rng.integers(24000, 26000),
                                 rng.integers(40500, 41500)], p=[0.25, 0.45, 0.24, 0.04, 0.02]))
            allele = synth.random_dna(rng, la)
            a_idx = len(seqs)
            seqs.append(spoil(allele, rng) if rng.random() < 0.4 else allele)
            for _r in range(int(rng.integers(1, 4))):
                lr = int(rng.integers(8, max(9, min(la, 6000))))
                st = int(rng.integers(0, la - lr + 1))
                seg = allele[st:st + lr]
                kind = rng.integers(0, 5)
                if kind == 0:
                    read = seg
                elif kind == 1:
                    read = synth.mutate(rng, seg, 0.01, 0.08, 0.04)[0]
                elif kind == 2:
                    read = synth.mutate(rng, seg, 0.001, 0.002, 0.002)[0]
                elif kind == 3:
                    h = len(seg) // 2
                    read = seg[:h // 2] + synth.revcomp(seg[h // 2:h]) + seg[h:]
                else:
                    read = synth.random_dna(rng, lr)
                if rng.random() < 0.25:
                    read = spoil(read, rng)
                if len(read) < 1:
                    continue
                r_idx = len(seqs)
                seqs.append(read)
                k = int(rng.choice([10, 20, 30, 40]))
                off2 = int(rng.integers(0, max(1, la // 3))) if rng.random() < 0.3 else 0
                rows.append((r_idx, a_idx, off2, k, int(rng.choice([1, 2, 3, 5, 7]))))
        upper = [bool(rng.random() < 0.15) for _ in seqs]
        ss = eng.seqset(seqs, upper)
        plan = eng.plan(ss, eng.make_pairs(rows))
        st = plan.run().copy()
        ok = [t for t in range(len(rows)) if st[t, 15] == 0]
        hits, fl, off = plan.fetch_hits(ok, want_flags=True)
        for q, t in enumerate(ok):
            s1, s2, off2, k, f = rows[t]
            a = seqs[s1].upper() if upper[s1] else seqs[s1]
            b = (seqs[s2].upper() if upper[s2] else seqs[s2])[off2:]
            exp = orc.dotdata_array(k, a, b)
            got = hits[off[q]:off[q + 1]]
            got = got[np.lexsort((got[:, 1], got[:, 0]))]
            assert got.tolist() == exp.tolist(), ("dots", seed, n_batches, t, rows[t], len(a), len(b))
            es = orc.pair_stats(k, a, b)
            if not f & 1:
                es[3] = es[4] = 0
            if not f & 2:
                es[5] = es[6] = es[9] = 0
            assert st[t, :10].tolist() == es[:10].tolist(), ("stats", seed, n_batches, t, rows[t], st[t, :10].tolist(), es[:10].tolist())
            if f & 4 and f & 1 and es[3] > 0 and len(exp) < 4000:
                _s, h, k1, _k2 = orc.pair_stats(k, a, b, want_hits=True)
                kept = [(int(x), int(y)) for x, y in h[k1 > 0]]
                c = orc.dis_to_diagnal_most_abundant_defined(list(kept))
                far = [d for d in ([x + c, y] for x, y in kept) if orc.eu_dis_single_dot(d) > 0.1]
                want = [int(round(2 * float(c))), len(far), int(round(2 * sum(d[0] - d[1] for d in far)))]
                assert st[t, 10:13].tolist() == want, ("dir", seed, n_batches, t, rows[t], st[t, 10:14].tolist(), want)
                n_dir += 1
        for t in range(len(rows)):
            if st[t, 15] != 0:
                s1, s2, off2, k, f = rows[t]
                try:
                    orc.pair_stats(k, seqs[s1].upper() if upper[s1] else seqs[s1], (seqs[s2].upper() if upper[s2] else seqs[s2])[off2:])
                    raise AssertionError(("status without KeyError", rows[t], st[t, 15]))
                except KeyError:
                    assert st[t, 15] == -3
        plan.close(); ss.close()
        n_pairs += len(rows); n_batches += 1
    return "fuzz ok: %d batches, %d pairs (%d with directed statistics checked), seed %d" % (n_batches, n_pairs, n_dir, seed)


if __name__ == "__main__":
    print(run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
