set -x
O=gpurun_out/final
mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_line_steps20.json 2>> $O/bench.err
python bench.py --batch 1024 --steps 12 --warmup 3 > $O/bench_line_batch1024.json 2>> $O/bench.err
python bench.py --overlap 1 --steps 20 --warmup 3 > $O/bench_line_overlap1.json 2>> $O/bench.err
python bench_secondary.py > $O/secondary_lines.jsonl 2> $O/secondary.err
python bench_secondary.py --workload jetnet30 --precision bf16 >> $O/secondary_lines.jsonl 2>> $O/secondary.err
PFM_MASKN=150 python tests/diag/stamps.py > $O/stamps_fast150.txt 2>&1
PFM_MASKN=32 python tests/diag/stamps.py > $O/stamps_fast32.txt 2>&1
python tests/diag/wide_time.py 256 100 > $O/cfg5_timings.txt 2>&1
python - <<'PY'
import json
for f in ("bench_line","bench_line_steps20","bench_line_batch1024","bench_line_overlap1"):
    d=json.loads(open(f"gpurun_out/final/{f}.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, round(d["ms_per_step"],3), round(d["value"],1), "sample_ms", round(d["sample_ms"],2), "train_alone", round(d["train_ms_alone"],3), "frac", round(r["frac"],4), "traffic", r["traffic"], "alone", round(r["kernel_alone_ms"],2), "cpu", round(d["cpu_baseline"]["value"],1) if "cpu_baseline" in d else None)
for l in open("gpurun_out/final/secondary_lines.jsonl"):
    l=l.strip()
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["workload"][:48], d["dtype"], round(d["value"],1), round(d["ms_per_step"],2), round(d["roofline"]["frac"],3))
PY
