cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p3
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/p3 -o f -- python3 $GRAFT_REPO_ROOT/tools/final_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r2_final_probe.log 2>&1
python3 - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/p3/*.db')[0])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
v = [t for t in tabs if t.startswith('kernels')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({v})")]
rows = cur.execute(f"select name, start, end from {v} order by start").fetchall()
seq = [(n[:40], (e - s) / 1000.0) for n, s, e in rows if 'train_' in n]
import collections
# pattern per iteration: board, final(compute), final(update), final(update)
fin = [d for n, d in seq if 'final' in n]
brd = [d for n, d in seq if 'board' in n]
import statistics as st
k = len(fin) // 3
print("board kernel mean us", st.mean(brd[5:]))
print("final compute-only mean us", st.mean(fin[15::3]), " update-only (1st) ", st.mean(fin[16::3]), " update-only (2nd) ", st.mean(fin[17::3]))
PY
