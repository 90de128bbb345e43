#!/bin/bash
# Round-4 measurement set, in three gpurun calls (each well under the 20-minute limit):
#   bash tests/diag/round4_measure.sh bench      -> gpurun_out/final/: bench lines, kernel stats of bench.py and of the train step
#   bash tests/diag/round4_measure.sh secondary  -> gpurun_out/final/secondary_lines.jsonl (fp32 + bf16 of every secondary workload)
#   bash tests/diag/round4_measure.sh pmc A|B    -> gpurun_out/pmc_*: SQ / GRBM / TCC counter passes (A: bench.py + cfg 2; B: cfg 4 / 5 / cross-attention)
# Copy what is to be judged into profiles/round4_* (profiles/README.md lists the mapping).
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
case "$1" in
bench)
    python bench.py > $O/bench_line.json 2> $O/bench.err
    python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_steps20.json 2>> $O/bench.err
    python bench.py --batch 1024 --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_line_batch1024.json 2>> $O/bench.err
    python bench.py --overlap 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_line_overlap1.json 2>> $O/bench.err
    python tests/diag/train_time.py 256 40 > $O/train_time.txt 2>&1
    python tests/diag/train_time.py 1024 20 >> $O/train_time.txt 2>&1
    ( cd /tmp && export TMPDIR=/tmp && for b in 256 1024; do rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats_b$b -o train -- python3 $R/tests/diag/train_time.py $b 20 > /dev/null 2> $O/train_stats_b$b.err; rm -f $O/train_stats_b$b/*trace.csv; done )
    ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_line_rocprof.json 2> $O/stats.err; rm -f $O/stats/*trace.csv )
    grep "train step" $O/train_time.txt
    ;;
secondary)
    rm -f $O/secondary_lines.jsonl
    for p in fp32 bf16; do for w in jetnet30 lhco_transformer lhco_crossattention jetclass; do
        python bench_secondary.py --workload $w --precision $p >> $O/secondary_lines.jsonl 2>> $O/secondary.err
    done; done
    ;;
pmc)
    if [ "$2" = "A" ]; then
        bash tests/diag/collect_pmc_sq.sh
        bash tests/diag/collect_pmc_rowmatrix.sh jetnet30 bf16 100
        bash tests/diag/collect_pmc_rowmatrix.sh jetnet30 fp32 100
    else
        bash tests/diag/collect_pmc_rowmatrix.sh lhco_transformer fp32 6
        bash tests/diag/collect_pmc_rowmatrix.sh jetclass fp32 6
        bash tests/diag/collect_pmc_rowmatrix.sh lhco_crossattention fp32 6
    fi
    ;;
esac
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "final")
for f in ("bench_line","bench_line_steps20","bench_line_batch1024","bench_line_overlap1","bench_line_rocprof"):
    try:
        d=json.loads(open(f"{O}/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        continue
    r=d["roofline"]
    print(f, round(d["ms_per_step"],3), round(d["value"],1), "sample_ms", round(d["sample_ms"],2), "train_alone", round(d["train_ms_alone"],3), "frac", round(r["frac"],4), "mfma_busy", r.get("mfma_busy"), "alone", round(r["kernel_alone_ms"],2), "bf16", round(d["bf16_mfma_sampler"]["sample_ms"],2) if "bf16_mfma_sampler" in d else None, "cpu", round(d["cpu_baseline"]["value"],1) if "cpu_baseline" in d else None)
try:
    for l in open(f"{O}/secondary_lines.jsonl"):
        l=l.strip()
        if l.startswith("{"):
            d=json.loads(l); print(d["config"]["workload"][:48], d["dtype"], round(d["value"],1), round(d["ms_per_step"],2), round(d["roofline"]["frac"],3), d["roofline"]["kernel"][:44], "traffic", d["roofline"]["traffic"], "cpu", round(d["cpu_baseline"]["value"],2) if "cpu_baseline" in d else None)
except Exception:
    pass
PY
