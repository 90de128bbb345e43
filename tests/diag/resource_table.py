"""Diagnostic (CPU container): VGPRs / scratch / SGPR spills per kernel from a `hipcc -Rpass-analysis=kernel-resource-usage` log.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iparticle_fm_amd/csrc -Rpass-analysis=kernel-resource-usage -c X.hip -o /tmp/x.o 2> log
    python tests/diag/resource_table.py log [name filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
KEY_S = r"ScratchSize \[bytes/lane\]"
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    if flt and flt not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    short = re.sub(r"^_ZN3pfm\d+", "", name)[:52]
    print("%-54s VGPR %4s  AGPR %3s  scratch %4s B/lane  SGPR spills %3s  occupancy %s" % (short, g("  VGPRs"), g("AGPRs"), g(KEY_S), g("SGPRs Spill"), g(r"Occupancy \[waves/SIMD\]")))
