"""GPU idle time inside the timed steps of a bench run, from a rocprofv3 --kernel-trace CSV: union of the kernels' busy
intervals vs wall time, and the idle gaps grouped by the kernel that FOLLOWS them (who was the GPU waiting for).
Caveat: under rocprofv3 every launch costs the host ~10 us more, which makes the ~3500-launch step host-bound (21 % idle in the trace);
without the profiler the host enqueues a step in 39 ms against 169 ms of GPU time (tools/enqueue_time.py), so read the gaps as an upper bound."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# keep the last 60 % of the trace (the timed steps; warm-up and baselines come first)
t_lo = rows[0][0] + int(0.4 * (rows[-1][1] - rows[0][0])) if len(sys.argv) < 3 else rows[0][0]
rows = [r for r in rows if r[0] >= t_lo]
wall = rows[-1][1] - rows[0][0]
busy, cur_end, gaps = 0, rows[0][0], collections.Counter()
gap_n = collections.Counter()
for s, e, name in rows:
    if s > cur_end:
        key = name.split("(")[0][-48:]
        gaps[key] += s - cur_end
        gap_n[key] += 1
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
print(f"kernels {len(rows)}, wall {wall / 1e6:.2f} ms, busy (union) {busy / 1e6:.2f} ms, idle {100.0 * (wall - busy) / wall:.2f} %")
for k, v in gaps.most_common(15):
    print(f"  idle before {k:50s} {v / 1e6:8.3f} ms in {gap_n[k]} gaps (avg {v / gap_n[k] / 1e3:.1f} us)")
