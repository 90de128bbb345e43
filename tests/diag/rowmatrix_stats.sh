# Kernel statistics of one row-matrix workload's sampler (rocprofv3 serialises the kernels: per-kernel times, not the overlapped rate).
#   bash tests/diag/rowmatrix_stats.sh lhco_transformer   -> gpurun_out/rm_<workload>/
W=${1:-lhco_transformer}
O=$GRAFT_REPO_ROOT/gpurun_out/rm_$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 $GRAFT_REPO_ROOT/bench_secondary.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $O/line.json 2> $O/err.txt
rm -f $O/*kernel_trace.csv
