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


def _subplots_of(hits) -> List[dict]:
    subs: List[dict] = []
    for h, title, pos in zip(hits, TITLES, POSITIONS):
        h = np.asarray(h, dtype=np.int32).reshape(-1, 2)
        key = (h[:, 0].astype(np.int64) << 32) | h[:, 1].astype(np.int64)
        if key.size > 1 and not bool(np.all(key[1:] >= key[:-1])):     # (the engine hands them over sorted already)
            h = h[np.argsort(key, kind="stable")]        # dotdata's order: by j, then i
        ticks = x_ticks(int(h[:, 0].max()))
        subs.append({"pos": pos, "title": title, "hits": h, "xticks": ticks, "xticklabels": [str(i) for i in ticks]})
    return subs


def figure_specs(reqs, engine=None) -> List[Optional[dict]]:
    """figure_spec for many requests with one sequence set and one device pass for all their dot plots."""
    from . import pipeline
    todo = [t for t, r in enumerate(reqs) if not (r.best_read == '' or r.best_read == [])]
    out: List[Optional[dict]] = [None] * len(reqs)
    if not todo:
        return out
    eng = engine or pipeline.get_engine()
    seqs, rows = [], []
    for t in todo:
        r = reqs[t]
        b = len(seqs)
        k, miss = int(r.k), int(r.best_read[1])
        seqs += [r.ref_seq, r.alt_seq, r.best_read[0]]
        rows += [(b, b, 0, k, 0), (b + 1, b + 1, 0, k, 0), (b + 2, b, miss, k, 0), (b + 2, b + 1, miss, k, 0)]
    ss = eng.seqset(seqs)
    try:
        st, hits = eng.dotplots(ss, eng.make_pairs(rows))
    finally:
        ss.close()
    for n, t in enumerate(todo):
        for row in st[4 * n:4 * n + 4]:
            pipeline._raise_for_status(row)
        mine = hits[4 * n:4 * n + 4]
        if any(len(h) == 0 for h in mine):
            continue
        out[t] = {"name": clamp_name(reqs[t].name), "subplots": _subplots_of(mine)}
    return out


def figure_spec(req, engine=None) -> Optional[dict]:
    """None when the reference draws nothing (no best read, or one of the four plots is empty); else
    {"name": file name, "subplots": [{"pos", "title", "hits" (n, 2) int32 [x = j, y = i], "xticks", "xticklabels"}]}."""
    return figure_specs([req], engine)[0]


def render_fresh(spec: dict) -> None:
    """What make_event_figure_1 hands to matplotlib (SF:1072-1089), call for call, for one specification."""
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
    fig.savefig(spec["name"])        # (the method itself: pyplot.savefig draws the whole figure a second time after the file is written)
    plt.close(fig)


_canvas: dict = {}


def _fixed_positions(ax) -> None:
    """Every draw matplotlib places the (empty) axis labels and the title anew, for which it lays out all tick labels of the
    axes two more times - 40 % of a figure's time.  The axis labels are empty and a title above an axes whose ticks are at
    the bottom stays where it starts, so both are told to keep their positions (the switches `set_label_coords` and
    `set_title(y=...)` flip; private names: where a matplotlib lacks them nothing changes).  The PNGs stay what a fresh
    figure gives, byte for byte - tests/test_host_cpu.py compares them."""
    ax.xaxis._autolabelpos = False
    ax.yaxis._autolabelpos = False
    ax._autotitlepos = False


def render(spec: dict) -> None:
    """The same PNG, byte for byte (tests/test_host_cpu.py), from a figure that is kept between calls: the four axes, their
    titles and line objects are made once per process, a call sets the data, rescales, sets the ticks and saves - two
    thirds of the time of building the figure anew."""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    layout = tuple((sp["pos"], sp["title"]) for sp in spec["subplots"])
    if _canvas.get("layout") != layout:
        if _canvas:
            plt.close(_canvas["fig"])
            _canvas.clear()
        fig = plt.figure()
        axes, lines = [], []
        for sp in spec["subplots"]:
            ax = plt.subplot(sp["pos"])
            (ln,) = ax.plot([], [], '+', color='r')
            ax.set_title(sp["title"])
            ax.grid(False)
            _fixed_positions(ax)
            axes.append(ax)
            lines.append(ln)
        _canvas.update(layout=layout, fig=fig, axes=axes, lines=lines)
    for sp, ax, ln in zip(spec["subplots"], _canvas["axes"], _canvas["lines"]):
        ln.set_data(sp["hits"][:, 0], sp["hits"][:, 1])
        ax.relim()
        ax.autoscale(True)
        ax.autoscale_view()
        ax.set_xticks(sp["xticks"], sp["xticklabels"])
    _canvas["fig"].savefig(spec["name"])


def make_event_figure_1(req) -> None:
    """`req` is a drivers.Figure.  Nothing is drawn without a best read or when any of the four
    plots is empty, as in the reference."""
    spec = figure_spec(req)
    if spec is not None:
        render(spec)


# ------------------------------------------------------------------------------------------
# Many figures: the reference draws one PNG per locus inside its locus loop (a third of its time per locus);
# here the dot plots of a batch's figures are one device pass and the drawing - matplotlib, ~90 ms a figure - goes to a
# few worker processes (`hostpool.py`: fresh interpreters that never touch the GPU), so that it runs beside the scoring of the other
# loci.  pipeline._answer calls `make_event_figure_1.batch`, pipeline.run_batch `make_event_figure_1.wait` before it
# returns: when a batch is done its PNGs are on disk.
# ------------------------------------------------------------------------------------------
import threading

_pending: list = []
_plock = threading.Lock()                     # _pending / _first_error: the chunk threads of cli.score_jobs hand figures in together
_render_lock = threading.Lock()               # the kept figure of render() is one per process
_CHUNK = 64                                   # figures per device pass (their dots pass through host memory)


def make_figures(reqs, engine=None) -> None:
    """make_event_figure_1 for many requests: batched dot plots (on `engine`: the calling thread's library context),
    drawing handed to the worker processes of vapor_amd.hostpool (or done here when there are none: VAPOR_HOST_PROCS=0, a
    single core)."""
    from . import hostpool
    pool = hostpool.get()
    for a in range(0, len(reqs), _CHUNK):
        for spec in figure_specs(reqs[a:a + _CHUNK], engine):
            if spec is None:
                continue
            if pool is None:
                with _render_lock:
                    render(spec)
            else:
                # (a few figures per worker in flight: their dots wait in memory, half a megabyte a figure)
                while True:
                    with _plock:
                        old = _pending.pop(0) if len(_pending) >= 4 * pool.n else None
                    if old is None:
                        break
                    _settle(old)
                item = (pool.submit("vapor_amd.figures", "render", spec), spec)
                with _plock:
                    _pending.append(item)


_first_error: list = []


def _settle(item) -> None:
    from . import hostpool
    r, spec = item
    try:
        try:
            r.result()
        except hostpool.WorkerLost:
            with _render_lock:
                render(spec)            # (the worker is gone: drawn here)
    except Exception as e:              # noqa: BLE001 - the first one is raised by wait() once all are in
        with _plock:
            if not _first_error:
                _first_error.append(e)


def wait() -> None:
    """Returns when every figure handed out so far (by any thread) is on disk; raises what a drawing raised."""
    with _plock:
        todo = _pending[:]
        del _pending[:]
    for item in todo:
        _settle(item)
    with _plock:
        err = _first_error.pop() if _first_error else None
    if err is not None:
        raise err


def _warm() -> int:
    """In a worker: matplotlib imported and the kept figure of the usual layout made, before the first figure arrives."""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    if not _canvas:
        fig = plt.figure()
        axes, lines = [], []
        for pos, title in zip(POSITIONS, TITLES):
            ax = plt.subplot(pos)
            (ln,) = ax.plot([], [], '+', color='r')
            ax.set_title(title)
            ax.grid(False)
            _fixed_positions(ax)
            axes.append(ax)
            lines.append(ln)
        _canvas.update(layout=tuple(zip(POSITIONS, TITLES)), fig=fig, axes=axes, lines=lines)
    return 0


def warm() -> None:
    """Starts the worker processes and lets each import matplotlib now (half a second each, a second of a run's start when it
    waited for the first batch's figures); nothing waits for them."""
    from . import hostpool
    pool = hostpool.get()
    if pool is not None:
        for _ in range(pool.n):
            pool.submit("vapor_amd.figures", "_warm")


def shutdown() -> None:
    from . import hostpool
    try:
        wait()
    finally:
        hostpool.shutdown()


make_event_figure_1.batch = make_figures
make_event_figure_1.wait = wait
