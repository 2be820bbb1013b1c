#!/usr/bin/env python3
"""Concurrency picture of a rocprofv3 --kernel-trace run (rocpd SQLite): over the middle half of the trace, how much of the
wall time has 0 / 1 / 2 / ... kernels in flight, the busy time per kernel family, and the gap statistics per queue.
rocpd_timeline.py <results.db>"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
rows = list(db.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start"))
if not rows:
    sys.exit("no kernels")
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo, hi = t0 + (t1 - t0) // 4, t0 + 3 * (t1 - t0) // 4
win = [r for r in rows if r[1] >= lo and r[2] <= hi]
def fam(n):
    for k in ("engine_step_fast", "engine_step", "gcn_trunk_boards_mm", "gcn_trunk_boards", "gcn_heads_mm", "gcn_heads", "finish_move", "begin_move", "refill"):
        if k in n: return k
    return "other"
ev = []
busy = defaultdict(int); cnt = defaultdict(int)
for r in win:
    ev.append((r[1], 1)); ev.append((r[2], -1)); busy[fam(r[0])] += r[2] - r[1]; cnt[fam(r[0])] += 1
ev.sort()
depth = 0; last = lo; hist = defaultdict(int)
for t, d in ev:
    hist[depth] += t - last; last = t; depth += d
hist[depth] += hi - last
wall = hi - lo
print(f"window {wall/1e6:.2f} ms, {len(win)} kernels, columns {cols}")
for k in sorted(hist): print(f"  {k} kernels in flight: {100*hist[k]/wall:5.1f} %")
for k, v in sorted(busy.items(), key=lambda x: -x[1]): print(f"  {k:22s} calls {cnt[k]:7d}  avg {v/cnt[k]/1e3:7.2f} us  sum/wall {v/wall:5.2f}")
if qcol:
    byq = defaultdict(list)
    for r in win: byq[r[3]].append(r)
    for q, rs in byq.items():
        gaps = [b[1] - a[2] for a, b in zip(rs, rs[1:])]
        busyq = sum(r[2] - r[1] for r in rs)
        print(f"  queue {q}: {len(rs)} kernels, busy {100*busyq/wall:5.1f} %, median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us, mean gap {sum(gaps)/len(gaps)/1e3:.2f} us")
