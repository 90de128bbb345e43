"""Diagnostic: a window of consecutive kernels from a rocprofv3 --kernel-trace csv (start offset, duration, gap behind the previous kernel's end).
   python tests/diag/trace_seq.py <..._kernel_trace.csv> [first row as a fraction of the trace = 0.6] [rows = 40] [start at the first kernel behind
   that point whose name contains this string]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = int(len(rows) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.6))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if len(sys.argv) > 4:
    while i0 < len(rows) - 1 and sys.argv[4] not in rows[i0]["Kernel_Name"]:
        i0 += 1
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for r in rows[i0:i0 + n]:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"{(a - prev_end) / 1e3:7.1f}"
    q = r.get("Queue_Id", "")
    print(f"{(a - t0) / 1e3:9.1f} us  dur {(b - a) / 1e3:7.1f}  gap {gap:>7s}  q{q:>3s}  {r['Kernel_Name'].split('(')[0][-44:]}  grid {r.get('Grid_Size', '')}")
    prev_end = b if prev_end is None else max(prev_end, b)
