"""Recurrence-plot PNGs (make_event_figure_1, SF:1072-1089): ref x ref,
alt x alt, best read x ref, best read x alt, drawn from dot plots the device computes."""
from __future__ import annotations

import numpy as np


def _subplot(plt, hits: np.ndarray, title: str, pos: int) -> None:
    if len(hits) == 0:
        return
    x, y = hits[:, 0], hits[:, 1]
    mx = int(x.max())
    digits = len(str(mx))
    unit = 10 ** (digits - 1)
    n = int(float(mx) / float(unit)) + 1
    if n < 3:
        ticks = [(i + 1) * unit for i in range(n)]
        half = [ticks[0] / 2]
        for i in range(len(ticks) - 1):
            half.append(half[0] * (2 * (i + 1) + 1))
        ticks = sorted(ticks + half)
    elif n < 5:
        ticks = [(i + 1) * unit for i in range(n)]
    else:
        ticks = [(i + 1) * 2 * unit for i in range(int(n / 2 + 1) + 1)]
    plt.subplot(pos)
    plt.plot(x, y, '+', color='r')
    plt.xticks(ticks, [str(i) for i in ticks])
    plt.title(title)
    plt.grid(False)


def make_event_figure_1(req) -> None:
    """`req` is a drivers.Figure.  Nothing is drawn without a best read or when any of the four
    plots is empty, as in the reference."""
    best = req.best_read
    if best == '' or best == []:
        return
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    from . import pipeline
    eng = pipeline.get_engine()
    ss = eng.seqset([req.ref_seq, req.alt_seq, best[0]])
    try:
        k, miss = int(req.k), int(best[1])
        st, hits = eng.dotplots(ss, eng.make_pairs([(0, 0, 0, k, 0), (1, 1, 0, k, 0), (2, 0, miss, k, 0), (2, 1, miss, k, 0)]))
    finally:
        ss.close()
    for row in st:
        pipeline._raise_for_status(row)
    if any(len(h) == 0 for h in hits):
        return
    name = req.name
    base = name.split('/')[-1]
    if len(base) > 150:
        name = '/'.join(name.split('/')[:-1]) + '/' + base[:140] + '.' + name.split('.')[-1]
    fig = plt.figure()
    for h, title, pos in zip(hits, ('ref vs. ref', 'alt vs. alt', 'read vs. ref', 'read vs. alt'), (221, 222, 223, 224)):
        _subplot(plt, h, title, pos)
    plt.savefig(name)
    plt.close(fig)
