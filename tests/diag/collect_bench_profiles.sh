#!/bin/bash
# rocprofv3 evidence for bench.py's roofline object, written under gpurun_out/bench_prof/ (copy the summaries to profiles/):
#   1. --kernel-trace --stats of `python3 bench.py`            -> per-kernel average durations
#   2. two --pmc passes (FETCH_SIZE | WRITE_SIZE cannot share a pass) -> HBM bytes per launch (MI355X_MICROARCH.md: FETCH_SIZE x 2)
# Run on the GPU box:  bash tests/diag/collect_bench_profiles.sh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bench_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py > $O/bench_line.json 2> $O/stats.err
rm -f $O/stats/*trace.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o bench -- python3 $R/bench.py --steps 3 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o bench -- python3 $R/bench.py --steps 3 --warmup 1 > /dev/null 2> $O/pmc_write.err
python3 - <<PY
import csv, json, collections, re
out = collections.defaultdict(dict)
for name, d in (("FETCH_SIZE", "$O/pmc_fetch"), ("WRITE_SIZE", "$O/pmc_write")):
    acc, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f"{d}/bench_counter_collection.csv")):
        if r["Counter_Name"] != name:
            continue
        k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]  # template arguments kept: <matrix mode, time-term table>
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in acc:
        if k.startswith("pfm::"):
            out[k][name] = acc[k] / cnt[k]   # KiB per launch
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1, sort_keys=True)
out["_note"] = ("KiB per launch, averages over the launches of `python3 bench.py --steps 3 --warmup 1` (tests/diag/collect_bench_profiles.sh): two "
                "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE). gfx950: FETCH_SIZE reports half the bytes of wide streaming reads "
                "(MI355X_MICROARCH.md) -> HBM-side bytes = 2 * FETCH_SIZE + WRITE_SIZE. Template arguments: <matrix mode (0 fp32, 1 bf16, "
                "2 split fp16), time-term table>.")
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if "sample_midpoint" in k or "backward" in k or "epic_dw" in k}))
PY
head -8 $O/stats/bench_kernel_stats.csv | cut -c1-160
cat $O/bench_line.json | cut -c1-200
