import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    st=r["stages"]
    print(f'{r["set"]:40s} L{r["layout"]} prof {r["ms_per_step"]:.4f} unprof {r["ms_per_step_unprofiled"]:.4f} tot {st["total"]:.4f} scan {st["scan"]:.4f} fixed {r["ms_per_step_unprofiled"]-st["scan"]:.4f} coarse {st["coarse"]:.3f} selp {st["select_probe"]:.3f} lut8 {st["lut8"]:.3f} grp {st["group"]:.3f} pre {st["scan0"]:.3f} sel {st["select"]:.3f} fin {st["finalize"]:.3f} same {r["same_as_first"]}')
