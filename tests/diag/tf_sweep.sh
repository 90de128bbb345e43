# cfg 4 sampler rate under the diagnostic switches of launch_linear_kernel (one box, same seeded multiplicities)
#   bash tests/diag/tf_sweep.sh  -> gpurun_out/tf_sweep.txt
O=gpurun_out/tf_sweep.txt
: > $O
run() { echo "== $*" >> $O; env "$@" python bench_secondary.py --workload lhco_transformer --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['sample_ms'],2), round(d['roofline']['frac'],4))" >> $O; }
run A=1
run PFM_TF_BN=64
run PFM_TF_ROWTILE=64
run PFM_TF_ROWTILE=128
run PFM_TF_CPW=1
run PFM_TF_CPW=2
run A=2
cat $O
