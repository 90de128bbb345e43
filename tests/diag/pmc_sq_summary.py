"""Per-kernel averages of the rocprofv3 --pmc passes collected by tests/diag/collect_pmc_sq.sh -> <dir>/pmc_sq_summary.json.

    python3 tests/diag/pmc_sq_summary.py gpurun_out/pmc_sq

Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots" and the cycle-constant table): SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count
shader cycles summed over the SIMDs / SQs that report; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles)
summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs (/ 8 = cycles of the dispatch); FETCH_SIZE / WRITE_SIZE are KiB, and
FETCH_SIZE reports half the bytes of wide streaming reads on gfx950 (HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE).
Derived per kernel:
    cycles          = GRBM_GUI_ACTIVE / 8                      (duration of one launch in shader cycles)
    mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)   (fraction of the chip's matrix-pipe cycles that were busy)
    mfma_busy_wg    = the same over the SIMDs the launch could occupy (min(workgroups, 256) CUs x 4) -- what a workgroup sees
    valu_per_wave_cycle = 4 x SQ_ACTIVE_INST_VALU / (4 x SQ_WAVE_CYCLES)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
grid = {}
for d in sorted(glob.glob(os.path.join(root, "*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            if not k.startswith("pfm::"):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
            try:  # the launch's own duration under the profiler (ns)
                acc[k]["duration_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                cnt[k]["duration_ns"] += 1
            except Exception:
                pass
            try:
                grid[k] = max(grid.get(k, 0), int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
            except Exception:
                pass
out = {}
for k in sorted(acc):
    e = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    e["launches_seen"] = max(v for c, v in cnt[k].items() if c != "duration_ns")
    if k in grid:
        e["workgroups_max"] = grid[k]
    if "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        e["cycles"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
            if k in grid:
                e["mfma_busy_wg"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * min(256, grid[k]) * cyc)
    if "SQ_ACTIVE_INST_VALU" in e and "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"] > 0:
        e["valu_active_share_of_wave_cycles"] = e["SQ_ACTIVE_INST_VALU"] / e["SQ_WAVE_CYCLES"]
    if "SQ_WAIT_INST_ANY" in e and "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"] > 0:
        e["issue_stall_share_of_wave_cycles"] = e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_BUSY_CYCLES", 0) > 0:
        e["lds_conflict_share_of_busy"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_BUSY_CYCLES"]
    out[k] = e
# whole-run totals (sampling-only runs of collect_pmc_rowmatrix.sh: PFM_PMC_EVALS = network evaluations of the profiled call)
tot_f = sum(acc[k].get("FETCH_SIZE", 0.0) for k in acc)
tot_w = sum(acc[k].get("WRITE_SIZE", 0.0) for k in acc)
if tot_f or tot_w:
    evals = int(os.environ.get("PFM_PMC_EVALS", "0"))
    out["_totals"] = {"FETCH_SIZE_KiB": tot_f, "WRITE_SIZE_KiB": tot_w, "hbm_bytes": (2 * tot_f + tot_w) * 1024.0,
                      "launches": int(sum(cnt[k].get("FETCH_SIZE", 0) for k in cnt)),
                      "duration_ns": sum(acc[k].get("duration_ns", 0.0) for k in acc) / max(1, len([c for c in ("a", "b", "c", "d") if os.path.isdir(os.path.join(root, c))])),
                      "evaluations": evals,
                      "hbm_bytes_per_evaluation": (2 * tot_f + tot_w) * 1024.0 / evals if evals else None,
                      "note": "sums over EVERY pfm:: launch of the run (HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE KiB, gfx950 correction)"}
out["_note"] = (("averages per launch over the launches of one `python3 bench_secondary.py --sample-only --steps 1 --warmup 0` call under "
                "rocprofv3 --pmc (tests/diag/collect_pmc_rowmatrix.sh; " if os.environ.get("PFM_PMC_EVALS") else
                "averages per launch over the launches of `python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline` under "
                "rocprofv3 --pmc (tests/diag/collect_pmc_sq.sh; ") +
                "dispatches serialised: every launch alone on the GPU).  Units and the derived fields: tests/diag/pmc_sq_summary.py")
json.dump(out, open(os.path.join(root, "pmc_sq_summary.json"), "w"), indent=1, sort_keys=True)
if "_totals" in out:
    print("_totals", out["_totals"])
for k, e in out.items():
    if k.startswith("_"):
        continue
    if any(s in k for s in ("sample_midpoint", "fm_loss", "epic_dw", "bwd_reduce")):
        print(k, {c: (round(v, 4) if v < 100 else round(v)) for c, v in e.items()})
