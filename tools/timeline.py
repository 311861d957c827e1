#!/usr/bin/env python3
"""rocprofv3 kernel trace (rocpd sqlite) -> timeline of the LAST search batch: every kernel after the previous batch's finalize, with start / end relative to it, duration, queue.  usage: timeline.py results.db out.md [batches_back]"""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = cur.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
short = lambda n: n.split('(')[0].replace('void ', '').replace('rsx::', '')
opens = [i for i, r in enumerate(rows) if "k_to_f32" in r[0] or "k_convert" in r[0]]
scans = [i for i, r in enumerate(rows) if "k_pq_scan" in r[0]]
last_scan = scans[-back]
closer = lambda n: "k_finalize" in n or "k_pq_final_tab" in n or "k_pq_rescore_all" in n          # the last kernel family of a batch
prev = [i for i in range(last_scan) if closer(rows[i][0])]
i0 = (prev[-1] + 1) if prev else 0
while i0 < last_scan and ("copyBuffer" in rows[i0][0] or "fillBuffer" in rows[i0][0]): i0 += 1
while last_scan - i0 > 14: i0 += 1          # (the first batch after the build: keep the tail)
i1 = last_scan
while i1 + 1 < len(rows) and rows[i1 + 1][1] - rows[i1][2] < 30000 and not closer(rows[i1][0]): i1 += 1
if i1 + 1 < len(rows) and "copyBuffer" in rows[i1 + 1][0] and rows[i1 + 1][1] - rows[i1][2] < 30000: i1 += 1
t0 = rows[i0][1]
with open(out, "w") as f:
    f.write("| kernel | queue | start us | end us | us |\n|---|---:|---:|---:|---:|\n")
    for n, s, e, q in rows[i0:i1 + 1]:
        f.write(f"| `{short(n)}` | {q} | {(s - t0) / 1e3:.1f} | {(e - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} |\n")
print(open(out).read())
