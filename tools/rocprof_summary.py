#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite) -> per-kernel summary text for profiles/.  usage: rocprof_summary.py results.db out.md "title" """
import sqlite3, sys
db, out, title = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(out, "w") as f:
    f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats (durations in microseconds, device time)\n\n")
    f.write("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for n, c, t, a, p in rows:
        f.write(f"| `{n.split('(')[0].replace('void ', '')}` | {c} | {t:.1f} | {a:.2f} | {p:.2f} |\n")
    f.write("\n## dispatch geometry / resources of the search kernels (first dispatch of each)\n\n")
    f.write("| kernel | grid (threads) | workgroup | LDS B | arch VGPR | accum VGPR | SGPR |\n|---|---|---|---:|---:|---:|---:|\n")
    seen = set()
    for r in cur.execute("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels order by start"):
        k = r[0].split('(')[0].replace('void ', '')
        if k in seen or not any(s in k for s in ("k_pq_scan", "k_pq_scan2", "k_select", "k_pq_lut", "k_gemm_exact", "k_finalize", "k_probe_setup", "k_flat_gemm", "k_list_scan")):
            continue
        seen.add(k)
        f.write(f"| `{k}` | {r[1]} x {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]} |\n")
print("wrote", out)
