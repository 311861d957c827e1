#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc sqlite db per kernel (avg over dispatches) into text; usage: pmc_summary.py db out.txt [like-pattern ...]"""
import sqlite3, sys
db, out, pats = sys.argv[1], sys.argv[2], sys.argv[3:] or ['%']
cur = sqlite3.connect(db).cursor()
with open(out, 'a') as f:
    f.write(f"# {db}\n")
    for pat in pats:
        rows = cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                           "where kernel_name like ? group by kernel_name, counter_name order by kernel_name, counter_name", (pat,)).fetchall()
        for k, c, v, n, d in rows:
            f.write(f"{k.split('(')[0].replace('void ','')}\t{c}\t{v:.6g}\tdispatches={n}\tavg_us={d/1e3:.1f}\n")
