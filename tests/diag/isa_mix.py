"""Diagnostic (CPU container): instruction mix of one kernel of a hipcc -save-temps assembly listing, per basic block.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iparticle_fm_amd/csrc -c particle_fm_amd/csrc/epic_kernels.hip -save-temps -o /tmp/ek.o
    python tests/diag/isa_mix.py epic_kernels-hip-amdgcn-amd-amdhsa-gfx950.s 'fast_kernelILi0ELb0ELb0E' [min_mfma_per_block]
Classes: mfma | valu (every other v_*: shares the fp32 matrix pipe's issue, tests/diag/mfma_coissue.hip) | lds (ds_*) | vmem (buffer_/global_) |
salu (s_* without waitcnt/barrier/nop) | wait (s_waitcnt, s_barrier, s_nop).  Prints the blocks with MFMAs and the totals."""
import re, sys, collections

path, pat = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\S*" + pat + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))

def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op in ("s_waitcnt", "s_barrier", "s_nop") or op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"

blocks, cur, name = [], collections.Counter(), "entry"
ops = collections.Counter()
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blocks.append((name, cur)); cur, name = collections.Counter(), m.group(1); continue
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if not m or l.lstrip().startswith((";", ".")): continue
    op = m.group(1)
    cur[cls(op)] += 1
    if cls(op) == "valu": cur["op:" + op] += 1
blocks.append((name, cur))
tot = collections.Counter()
for n, c in blocks:
    tot.update(c)
    if c["mfma"] >= min_mfma:
        top = sorted(((k[3:], v) for k, v in c.items() if k.startswith("op:")), key=lambda kv: -kv[1])[:8]
        print(f"{n:14s} mfma {c['mfma']:4d} valu {c['valu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} salu {c['salu']:3d} wait {c['wait']:3d}  " +
              " ".join(f"{k}:{v}" for k, v in top))
print("TOTAL (static)", {k: v for k, v in tot.items() if not k.startswith("op:")})
print("VALU ops (static)", sorted(((k[3:], v) for k, v in tot.items() if k.startswith("op:")), key=lambda kv: -kv[1])[:30])
