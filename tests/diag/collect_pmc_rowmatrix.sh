#!/bin/bash
# SQ / GRBM counters AND HBM traffic of a secondary workload's SAMPLER (bench_secondary.py --sample-only: every launch of the run is a
# sampler launch; every dispatch alone on the GPU under --pmc):
#   bash tests/diag/collect_pmc_rowmatrix.sh lhco_transformer [fp32|bf16]  -> gpurun_out/pmc_<workload>_<precision>/pmc_sq_summary.json
#   (copy to profiles/round4_pmc_<workload>_<precision>.json: bench_secondary.py reads roofline.traffic from there)
# Passes A / B: SQ + GRBM (as collect_pmc_sq.sh); C / D: FETCH_SIZE | WRITE_SIZE (TCC: one counter per pass).  Counter passes carry
# --kernel-trace only; the program sits directly behind `--`.
set -e
W=${1:-lhco_transformer}
P=${2:-fp32}
NFE_STEPS=${3:-6}   # ode_steps of the profiled call: 2 (ode_steps - 1) evaluations
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_${W}_${P}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --precision $P --steps 1 --warmup 0 --ode-steps $NFE_STEPS --no-cpu-baseline --sample-only --overlap 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o s -- python3 $R/bench_secondary.py $ARGS > $O/a.json 2> $O/a.err
echo "pass A done"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/b -o s -- python3 $R/bench_secondary.py $ARGS > $O/b.json 2> $O/b.err
echo "pass B done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c -o s -- python3 $R/bench_secondary.py $ARGS > $O/c.json 2> $O/c.err
echo "pass C done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/d -o s -- python3 $R/bench_secondary.py $ARGS > $O/d.json 2> $O/d.err
echo "pass D done"
PFM_PMC_EVALS=$((2 * (NFE_STEPS - 1))) python3 $R/tests/diag/pmc_sq_summary.py $O > $O/pmc_sq_summary.txt
rm -f $O/*/s_kernel_trace.csv $O/*/*agent_info.csv $O/*/*counter_collection.csv
tail -3 $O/pmc_sq_summary.txt
