#!/bin/bash
# Diagnostic: shader clock / power samples (rocm-smi) while bench.py's timed region runs, against one sampler launch at a time
# (tests/diag/overlap_l2.py, one stream).   bash tests/diag/clock_watch.sh
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Average Graphics" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_bench.txt &
W=$!
python bench.py --steps 300 --warmup 5 --no-cpu-baseline > gpurun_out/clock_bench_line.json 2>/dev/null
wait $W
echo "== bench (two launches in flight)"; sort gpurun_out/clock_bench.txt | uniq -c | sort -rn | head -8
( for i in $(seq 1 24); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Average Graphics" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_alone.txt &
W=$!
PFM_ONLY_ONE=1 python tests/diag/overlap_l2.py 200 > gpurun_out/clock_alone_run.txt 2>&1
wait $W
echo "== one launch at a time"; sort gpurun_out/clock_alone.txt | uniq -c | sort -rn | head -8
head -3 gpurun_out/clock_alone_run.txt
