#!/bin/bash
# SQ / GRBM counter evidence for bench.py's roofline object: matrix-pipe busy cycles of the lean sampler and of the
# train kernels on the CURRENT build, written under gpurun_out/pmc_sq/ (copy pmc_sq_summary.json to profiles/roundN_pmc_summary.json).
#   pass A: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
#   pass B: SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
#   pass C / D: FETCH_SIZE | WRITE_SIZE (TCC: cannot share a pass)
# Counter passes carry --kernel-trace only (no --stats / sys / hip traces: gpurun refuses the mix).  The program sits directly behind
# `--` (python3, no env / bash -c hop).  Under --pmc rocprofv3 serialises the dispatches: the numbers are per launch ALONE on the GPU.
# Run on the GPU box:  bash tests/diag/collect_pmc_sq.sh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--gpus 1 --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o bench -- python3 $R/bench.py $ARGS > $O/a.json 2> $O/a.err
echo "pass A done"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/b -o bench -- python3 $R/bench.py $ARGS > $O/b.json 2> $O/b.err
echo "pass B done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c -o bench -- python3 $R/bench.py $ARGS > $O/c.json 2> $O/c.err
echo "pass C done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/d -o bench -- python3 $R/bench.py $ARGS > $O/d.json 2> $O/d.err
echo "pass D done"
python3 $R/tests/diag/pmc_sq_summary.py $O > $O/pmc_sq_summary.txt
cat $O/pmc_sq_summary.txt
rm -f $O/*/bench_kernel_trace.csv $O/*/*agent_info.csv
