"""Kernel-trace timeline of the training step: how much of the wall time has NO kernel running on the GPU (gaps between launches), and how much
has kernels of BOTH streams running.  usage (on the GPU box): rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/prof_step.py two ; python3 tools/trace_gaps.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
# keep the last 90 ms of the trace (three steady-state steps; set-up, warm-up and their long gaps are before that)
t0 = max(e[1] for e in ev) - 90_000_000
ev = [e for e in ev if e[0] >= t0]
lo, hi = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, _, _ in ev: pts += [(s, 1), (e, -1)]
pts.sort()
busy = both = 0; depth = 0; last = lo
for t, d in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: both += t - last
    depth += d; last = t
wall = hi - lo
gaps = []
cur_end = ev[0][1]
for s, e, n, _ in ev[1:]:
    if s > cur_end: gaps.append((s - cur_end, n))
    cur_end = max(cur_end, e)
print(f"window {wall / 1e6:.2f} ms: some kernel running {busy / wall:.3f}, two or more running {both / wall:.3f}, idle {1 - busy / wall:.3f} ({(wall - busy) / 1e6:.2f} ms)")
gaps.sort(reverse=True)
print("largest gaps (us, next kernel):", [(round(g / 1e3, 1), n.split('(')[0][-40:]) for g, n in gaps[:8]])
print(f"gaps > 2 us: {sum(1 for g, _ in gaps if g > 2000)}, total {sum(g for g, _ in gaps if g > 2000) / 1e6:.2f} ms")
