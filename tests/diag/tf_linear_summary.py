"""Summarise tests/diag/tf_linear_prof.sh output: the last evaluation's kernels with their share of the fp32-MFMA peak."""
import csv, sys, glob
PEAK = 157.3e12
M = 128 * 279
for d in sorted(glob.glob(sys.argv[1] + "/rt*/")):
    rows = list(csv.DictReader(open(d + "p_kernel_trace.csv")))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) for r in rows))
    n = len(ks) // 23
    last = ks[-n:]
    tot = sum(k[1] for k in last) / 1e3
    lin = [k for k in last if "tf_linear" in k[2]]
    print(f"{d}: evaluation {tot:.1f} us, Linears {sum(k[1] for k in lin)/1e3:.1f} us, attention {sum(k[1] for k in last if 'attn' in k[2])/1e3:.1f} us")
    if len(sys.argv) > 2:
        for k in lin[1:5]:
            print(f"     {k[1]/1e3:7.1f} us  {k[3]:5d} workgroups  {k[2][21:50]}")
