"""Recurrence-plot PNGs (make_event_figure_1, SF:1072-1089): ref x ref,
alt x alt, best read x ref, best read x alt, drawn from dot plots the device computes.

`figure_spec` decides everything the reference's make_event_figure_1 / makeDotplot_subfigure (SF:1041-1089) decide -
whether a figure is drawn at all, the four point sets in dotdata's order, the tick positions and labels, the titles,
the file name with its 150-character clamp - and `make_event_figure_1` hands that to matplotlib.  The parity tests
compare the specification with what the reference passes to matplotlib (tests/golden/figures.json.gz)."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

TITLES = ('ref vs. ref', 'alt vs. alt', 'read vs. ref', 'read vs. alt')
POSITIONS = (221, 222, 223, 224)


def x_ticks(max_x: int) -> list:
    """Tick positions of makeDotplot_subfigure (SF:1051-1062) for the largest x of a plot."""
    digits = len(str(max_x))
    unit = 10 ** (digits - 1)
    n = int(float(max_x) / float(unit)) + 1
    if n < 3:
        ticks = [(i + 1) * unit for i in range(n)]
        half = [ticks[0] / 2]
        for i in range(len(ticks) - 1):
            half.append(half[0] * (2 * (i + 1) + 1))
        return sorted(ticks + half)
    if n < 5:
        return [(i + 1) * unit for i in range(n)]
    return [(i + 1) * 2 * unit for i in range(int(n / 2 + 1) + 1)]


def clamp_name(name: str) -> str:
    """SF:1080-1081: a file name of more than 150 characters keeps its first 140 and its extension."""
    base = name.split('/')[-1]
    if len(base) > 150:
        return '/'.join(name.split('/')[:-1]) + '/' + base[:140] + '.' + name.split('.')[-1]
    return name


def figure_spec(req, engine=None) -> Optional[dict]:
    """None when the reference draws nothing (no best read, or one of the four plots is empty); else
    {"name": file name, "subplots": [{"pos", "title", "hits" (n, 2) int32 [x = j, y = i], "xticks", "xticklabels"}]}."""
    best = req.best_read
    if best == '' or best == []:
        return None
    from . import pipeline
    eng = engine or pipeline.get_engine()
    ss = eng.seqset([req.ref_seq, req.alt_seq, best[0]])
    try:
        k, miss = int(req.k), int(best[1])
        st, hits = eng.dotplots(ss, eng.make_pairs([(0, 0, 0, k, 0), (1, 1, 0, k, 0), (2, 0, miss, k, 0), (2, 1, miss, k, 0)]))
    finally:
        ss.close()
    for row in st:
        pipeline._raise_for_status(row)
    if any(len(h) == 0 for h in hits):
        return None
    subs: List[dict] = []
    for h, title, pos in zip(hits, TITLES, POSITIONS):
        h = np.asarray(h, dtype=np.int32).reshape(-1, 2)
        h = h[np.lexsort((h[:, 1], h[:, 0]))]            # dotdata's order: by j, then i
        ticks = x_ticks(int(h[:, 0].max()))
        subs.append({"pos": pos, "title": title, "hits": h, "xticks": ticks, "xticklabels": [str(i) for i in ticks]})
    return {"name": clamp_name(req.name), "subplots": subs}


def make_event_figure_1(req) -> None:
    """`req` is a drivers.Figure.  Nothing is drawn without a best read or when any of the four
    plots is empty, as in the reference."""
    spec = figure_spec(req)
    if spec is None:
        return
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    fig = plt.figure()
    for sp in spec["subplots"]:
        plt.subplot(sp["pos"])
        plt.plot(sp["hits"][:, 0], sp["hits"][:, 1], '+', color='r')
        plt.xticks(sp["xticks"], sp["xticklabels"])
        plt.title(sp["title"])
        plt.grid(False)
    plt.savefig(spec["name"])
    plt.close(fig)
