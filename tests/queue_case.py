"""A crafted (read, allele) pair that drives the join kernel's candidate queue through its edge: one read strip
whose positions 16*lane + 0 / + 4 / + 8 carry, over the 64 lanes, exactly 127, 129 and 129 candidates.  With a fast
path that admitted totals of 129 on a queue of 256 the third position wrote one candidate to the dump slot and a dot
went missing.  Shared by the CPU test (which checks the design with a model of the device hash) and
the GPU test (which compares with the oracle)."""
import numpy as np

_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
K = 10


def _key(s):
    v = 0
    for t, ch in enumerate(s):
        v |= _CODE[ch] << (2 * t)
    return v


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def bucket(kmer, nb_log2=15):
    """join_kernel's bucket of a k = 10 k-mer (2-bit keys, one word): canon_hash of min(key, revcomp)."""
    c = min(_key(kmer), _key(_rc(kmer)))
    return ((c * 0x9E3779B1) & 0xFFFFFFFF) >> (32 - nb_log2)


def build(seed=5):
    rng = np.random.default_rng(seed)

    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))

    while True:
        P, tailQ, f2 = rnd(10), rnd(4), rnd(2)
        Q = P[4:10] + tailQ
        P1 = P[:2] + rnd(2) + P[4:10]              # differs from P inside, shares what Q and R see of it
        R = Q[4:10] + f2 + P[:2]
        f2x = "".join("ACGT"[(_CODE[c] + 1) & 3] for c in f2)
        tq = "".join("ACGT"[(_CODE[c] + 2) & 3] for c in tailQ)
        kms = [P, P1, Q, R]
        if len({bucket(x) for x in kms}) == 4 and len(set(kms)) == 4:
            break
    # lane b owns read[16 b, 16 b + 16): P (or P1 in lane 63) at +0, Q (lanes < 43, another tail elsewhere) at +4,
    # R at +8 through the filler and the next block's first two bases (lanes < 43)
    blocks = []
    for b in range(64):
        head = P1 if b == 63 else P
        tail = tailQ if b < 43 else tq
        fill = f2 if b < 43 else f2x
        blocks.append(head + tail + fill)
    read = "".join(blocks) + P[:2] + rnd(40)
    # allele: the four k-mers (2, 1, 3, 3 copies) between random spacers that end / start with a base breaking
    # any longer match with the read's blocks
    def other(*forbidden):
        return [c for c in "ACGT" if c not in forbidden][0]

    # (k-mer, copies, bases the read has before / after it in some lane)
    plan = ((P, 2, (f2[1], f2x[1]), (tailQ[0], tq[0])), (P1, 1, (f2[1], f2x[1]), (tailQ[0], tq[0])),
            (Q, 3, (P[3], P1[3]), (f2[0], f2x[0])), (R, 3, (P[7],), (P[2], P1[2])))
    parts = [rnd(37)]
    for km, copies, before, after in plan:
        for _ in range(copies):
            parts += [other(*before), km, other(*after), rnd(39)]
    allele = "".join(parts)
    return read, allele, {"P": P, "P1": P1, "Q": Q, "R": R}


def candidate_totals(read, allele):
    """Per position offset t of the first strip: candidates summed over the 64 lanes (bucket sizes of the allele
    table, i.e. what the kernel calls `tot`)."""
    from collections import Counter
    sizes = Counter(bucket(allele[p:p + K]) for p in range(len(allele) - K + 1))
    tot = []
    for t in range(16):
        tot.append(sum(sizes.get(bucket(read[16 * b + t:16 * b + t + K]), 0) for b in range(64)
                       if 16 * b + t + K <= len(read)))
    return tot
