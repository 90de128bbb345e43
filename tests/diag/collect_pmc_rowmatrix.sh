#!/bin/bash
# SQ / GRBM counters of a row-matrix workload's kernels (the sampler of bench_secondary.py; every dispatch alone on the GPU under --pmc):
#   bash tests/diag/collect_pmc_rowmatrix.sh lhco_transformer  -> gpurun_out/pmc_<workload>/pmc_sq_summary.json
# Same passes, units and summary as collect_pmc_sq.sh / pmc_sq_summary.py; the program sits directly behind `--`.
set -e
W=${1:-lhco_transformer}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --steps 1 --warmup 1 --ode-steps 10 --no-cpu-baseline"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o s -- python3 $R/bench_secondary.py $ARGS > $O/a.json 2> $O/a.err
echo "pass A done"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/b -o s -- python3 $R/bench_secondary.py $ARGS > $O/b.json 2> $O/b.err
echo "pass B done"
python3 $R/tests/diag/pmc_sq_summary.py $O > $O/pmc_sq_summary.txt
rm -f $O/*/s_kernel_trace.csv $O/*/*agent_info.csv $O/*/*counter_collection.csv
