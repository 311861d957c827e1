#!/bin/bash
# kernel trace of the pipelined phase of tools/wip/pipe_sweep.py (scan grids of 240 workgroups) -> gpurun_out/<tag>_pipe_trace.txt
TAG=${TAG:-pipe}; R=${RESERVE:-16}
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pt
timeout 300 rocprofv3 --kernel-trace -d /tmp/pt -o pt --output-format csv -- python $ROOT/tools/wip/pipe_sweep.py --reserves $R > /dev/null 2>&1
python3 - "$ROOT/gpurun_out/${TAG}_pipe_trace.txt" "$R" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/pt/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
want = str((256 - int(sys.argv[2])) * 1024)
idx = [i for i, r in enumerate(rows) if "k_pq_scan_rot" in r["Kernel_Name"] and r.get("Grid_Size_X", r.get("Grid_Size")) == want]
out = open(sys.argv[1], "w")
if not idx:
    out.write("no scan launch with grid %s\n" % want); sys.exit(0)
lo, hi = max(0, idx[-9] - 30 if len(idx) >= 9 else idx[0] - 30), min(len(rows), idx[-1] + 10)
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    out.write("%10.1f %10.1f %9.1f q%s %s grid %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                r.get("Queue_Id", "?"), r["Kernel_Name"][:50], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
