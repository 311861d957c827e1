#!/usr/bin/env python3
"""Turn a `rocprofv3 --kernel-trace --pmc FETCH_SIZE` pass over bench.py into profiles-style pmc_traffic.json:
HBM bytes per launch of the IVF-PQ scan kernel, stamped with the hash of the sources it was measured on (bench.py
refuses the file when the hash no longer matches).  FETCH_SIZE is reported in KiB and, on gfx950, counts exactly half
of the bytes of 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM) -> bytes = value * 1024 * 2.

A second db — an SQ pass of the same command (`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`) — adds the matrix-core busy fraction
of the same kernel: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE cycles / 8 XCDs x 1024).

usage: update_pmc_traffic.py <results.db> <out.json> [n_vectors] [n_gpus] [sq_results.db]"""
import datetime, json, os, sqlite3, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import kernel_source_hash

db, out = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
ng = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select kernel_name, avg(value), count(*), avg(duration) from counters_collection "
                   "where counter_name = 'FETCH_SIZE' and (kernel_name like '%k_pq_scan_sl8%' or kernel_name like '%k_pq_scan_rot%' or kernel_name like '%k_pq_scan8%') "
                   "group by kernel_name order by avg(duration) desc").fetchall()
assert rows, "no IVF-PQ scan kernel in the PMC pass"
name, val, cnt, dur = rows[0]
kernel = "k_pq_scan_sl8" if "k_pq_scan_sl8" in name else "k_pq_scan_rot" if "k_pq_scan_rot" in name else "k_pq_scan8"
res = {"kernel": kernel, "kernel_name": name.split("(")[0].replace("void ", ""), "n": n, "n_gpus": ng,
       "hbm_bytes_per_launch": val * 1024.0 * 2.0, "fetch_size_kib_avg": val, "dispatches": cnt, "avg_us_under_pmc": dur / 1e3,
       "source_sha256": kernel_source_hash(), "date": datetime.date.today().isoformat(),
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs; "
                 "bytes = FETCH_SIZE[KiB] * 1024 * 2 (gfx950 halves 16-B/lane streaming reads)"}
if len(sys.argv) > 5:
    c2 = sqlite3.connect(sys.argv[5]).cursor()
    q = "select avg(value) from counters_collection where counter_name = ? and kernel_name = ?"
    busy = c2.execute(q, ("SQ_VALU_MFMA_BUSY_CYCLES", name)).fetchone()[0]
    act = c2.execute(q, ("GRBM_GUI_ACTIVE", name)).fetchone()[0]
    if busy and act:
        res["mfma_busy_cycles"] = busy; res["gui_active_cycles"] = act
        # GRBM_GUI_ACTIVE comes summed over the 8 XCDs (one GRBM each); the busy cycles summed over the 8 x 32 x 4 SIMDs
        res["mfma_busy_frac"] = busy / (act / 8.0 * 1024.0)
        res["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES (summed over 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
