"""Diagnostic: from a rocprofv3 --kernel-trace csv, the GPU-busy share (union of the kernel intervals over the wall time of the second
half of the trace) and the per-kernel totals.   python tests/diag/trace_busy.py <..._kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[len(rows) // 2:]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
t0, t1 = iv[0][0], max(b for _, b in iv)
u, (s, e) = 0, iv[0]
for a, b in iv[1:]:
    if a <= e:
        e = max(e, b)
    else:
        u += e - s
        s, e = a, b
u += e - s
tot, cnt = collections.Counter(), collections.Counter()
for r in sel:
    k = r["Kernel_Name"].split("(")[0][-48:]
    tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[k] += 1
print(f"wall {(t1 - t0) / 1e6:.2f} ms, GPU busy (union of kernels) {u / 1e6:.2f} ms = {u / (t1 - t0):.3f}, sum of kernel durations {sum(tot.values()) / 1e6:.2f} ms")
for k, v in tot.most_common(10):
    print(f"  {k:50s} n={cnt[k]:6d} total {v / 1e6:9.2f} ms  avg {v / cnt[k] / 1e3:8.1f} us")
