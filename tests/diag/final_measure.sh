# The round's measurement run (one MI355X box): bench lines, secondary lines, kernel stats, SQ / TCC counters.
#   bash tests/diag/final_measure.sh        -> gpurun_out/final/ (copy what is to be judged into profiles/roundN_*)
set -x
O=gpurun_out/final
mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_steps20.json 2>> $O/bench.err
python bench.py --batch 1024 --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_line_batch1024.json 2>> $O/bench.err
python bench.py --overlap 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_line_overlap1.json 2>> $O/bench.err
python bench_secondary.py > $O/secondary_lines.jsonl 2> $O/secondary.err
python bench_secondary.py --workload jetnet30 --precision bf16 >> $O/secondary_lines.jsonl 2>> $O/secondary.err
# bf16 operands on the row-matrix paths (PFM_{TF,EW,CA}_F_BF16): cfg 4, cfg 5, cross-attention
for w in lhco_transformer jetclass lhco_crossattention; do python bench_secondary.py --workload $w --precision bf16 >> $O/secondary_lines.jsonl 2>> $O/secondary.err; done
python tests/diag/train_time.py 256 40 > $O/train_time.txt 2>&1
python tests/diag/train_time.py 1024 20 >> $O/train_time.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && for b in 256 1024; do rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/train_stats_b$b -o train -- python3 $GRAFT_REPO_ROOT/tests/diag/train_time.py $b 20 > /dev/null 2> $GRAFT_REPO_ROOT/$O/train_stats_b$b.err; rm -f $GRAFT_REPO_ROOT/$O/train_stats_b$b/*trace.csv; done )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_line_rocprof.json 2> $GRAFT_REPO_ROOT/$O/stats.err; rm -f $GRAFT_REPO_ROOT/$O/stats/*trace.csv )
python - <<'PY'
import json
for f in ("bench_line","bench_line_steps20","bench_line_batch1024","bench_line_overlap1","bench_line_rocprof"):
    try:
        d=json.loads(open(f"gpurun_out/final/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r=d["roofline"]
    print(f, round(d["ms_per_step"],3), round(d["value"],1), "sample_ms", round(d["sample_ms"],2), "train_alone", round(d["train_ms_alone"],3), "frac", round(r["frac"],4), "mfma_busy", r.get("mfma_busy"), "traffic", r["traffic"], "alone", round(r["kernel_alone_ms"],2), "bf16", round(d["bf16_mfma_sampler"]["sample_ms"],2) if "bf16_mfma_sampler" in d else None, "cpu", round(d["cpu_baseline"]["value"],1) if "cpu_baseline" in d else None)
for l in open("gpurun_out/final/secondary_lines.jsonl"):
    l=l.strip()
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["workload"][:48], d["dtype"], round(d["value"],1), round(d["ms_per_step"],2), round(d["roofline"]["frac"],3), "cpu", round(d["cpu_baseline"]["value"],2) if "cpu_baseline" in d else None)
PY
cat $O/train_time.txt | grep "train step"
