"""Developer tool: per-timestep timeline from a rocprofv3 --kernel-trace CSV (last full step).
usage: python tools/trace_step.py <kernel_trace.csv> [marker_kernel_substring] [--seq]
  --seq: additionally the kernels of that step in launch order: start offset, duration, gap to the previous kernel's end, stream"""
import csv, sys, collections
args = [a for a in sys.argv[1:] if not a.startswith("--")]
seq = "--seq" in sys.argv
path = args[0]
marker = args[1] if len(args) > 1 else "k_hh_update"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
a, b = starts[-2], starts[-1]
seg = rows[a:b]
t0, t1 = seg[0][0], rows[b][0]
busy = sum(r[1] - r[0] for r in seg)
print(f"step wall {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, launches {len(seg)}")
agg = collections.OrderedDict()
for s, e, n, _ in seg:
    k = n.split("(")[0][:70]
    c = agg.setdefault(k, [0, 0])
    c[0] += 1; c[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / 1e3:9.1f} us  {c:4d} x {t / c / 1e3:7.2f}  {k}")
gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
big = sorted(((g, seg[i][2].split('(')[0][:40], seg[i + 1][2].split('(')[0][:40]) for i, g in enumerate(gaps)), reverse=True)[:8]
print("largest gaps (us):", [(round(g / 1e3, 1), p, q) for g, p, q in big])
print("sum of gaps %.1f us" % (sum(gaps) / 1e3))
if seq:
    last_end = seg[0][0]
    for s_, e_, n, q in seg:
        print(f"{(s_ - t0) / 1e3:9.1f} us  dur {(e_ - s_) / 1e3:7.2f}  gap {(s_ - last_end) / 1e3:7.2f}  q{q}  {n.split('(')[0][:70]}")
        last_end = max(last_end, e_)
