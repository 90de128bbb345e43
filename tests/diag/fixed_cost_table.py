"""Diagnostic: the table of tests/diag/fixed_cost_table.sh's ablation timings (ab_time.py output): where the part of a lean-sampler
evaluation that is NOT matrix-pipe issue goes, per evaluation, in cycles at the measured clock.
    python tests/diag/fixed_cost_table.py gpurun_out/ab_fixed.log [clock_GHz=2.29]"""
import re, sys, collections

clk = float(sys.argv[2]) if len(sys.argv) > 2 else 2.29
t = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"lib_ab_(\w+)\.so\s+(B=\d+ \S+)\s+([\d.]+) ms", line)
    if m:
        t[m.group(1)][m.group(2)] = float(m.group(3))
NFE = 198
cases = [("B=256 n=32", 2), ("B=256 n=64", 4), ("B=256 n=96", 6), ("B=256 n=150", 10)]
def cyc(ms):  # per evaluation, one jet per CU (B = 256 = one round)
    return ms * 1e-3 / NFE * clk * 1e9
print(f"cycles per evaluation at {clk} GHz (one jet per CU; 100-step midpoint sample / 198); ideal = 13 Linears x 2 waves x 32 MFMAs x 32 cycles per tile\n")
print(f"{'':34s}" + "".join(f"{c[0][6:]:>12s}" for c in cases))
base = {c[0]: cyc(t["base"][c[0]]) for c in cases}
ideal = {c[0]: 13 * 2048.0 * c[1] for c in cases}
def row(name, vals, pct_of=None):
    s = f"{name:34s}"
    for c in cases:
        v = vals[c[0]]
        s += f"{v:9.0f}" + (f"{100 * v / pct_of[c[0]]:3.0f}%" if pct_of else "   ")
    print(s)
row("whole evaluation (measured)", base)
row("matrix-pipe issue (ideal)", ideal, base)
fixed = {k: base[k] - ideal[k] for k in base}
row("everything else", fixed, base)
print("\nremoved piece -> cycles saved (share of 'everything else'):")
pieces = [("nochain", "seven per-jet chains (incl. their barrier)"), ("nobar", "12 barriers between particle phases"), ("nochain_nobar", "chains + barriers together"),
          ("nohead", "fc_l3 head"), ("nol1", "fc_l1 (one MFMA per tile)"), ("noepi", "pair epilogues: lrelu, LDS write, pool sums"),
          ("nopoolfin", "pool tails: DPP row sum, mean, LDS write (7)"), ("nopf", "weight / table loads (all riders + tails)"),
          ("mfmaonly", "all of the above removed")]
for key, what in pieces:
    if key in t:
        row(what, {c[0]: base[c[0]] - cyc(t[key][c[0]]) for c in cases}, fixed)
if "mfmaonly" in t:
    left = {c[0]: cyc(t["mfmaonly"][c[0]]) - ideal[c[0]] for c in cases}
    row("left in the bare MFMA phases", left, fixed)
    print("  (= phase fill / drain: the first operand reads and the last accumulator's latency of 13 phases with both waves of a SIMD in step,\n"
          "   A-operand VGPR setup, s_nop / s_waitcnt; plus what the clock estimate is off by)")
print("\nbench mix (ms per 100-step sample):")
for key in t:
    print(f"  {key:14s} B=256 U{{30..150}} {t[key].get('B=256 U{30..150}', float('nan')):8.3f}   B=1024 {t[key].get('B=1024 U{30..150}', float('nan')):8.3f}")
