#!/bin/bash
# per-kernel durations of one cfg-4 evaluation under rocprofv3.  usage (on the GPU box): bash tests/diag/tf_linear_prof.sh <tag> <rowtile:panel>...
# (row tile 0 = the heuristic; panel 0 = tf_linear_kernel only, 1 = the launcher's choice, 32 / 64 = panel kernel with that row tile:
# the diagnostics-only PFM_TF_ROWTILE / PFM_TF_PANEL overrides)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift; mkdir -p gpurun_out/lin_$tag
for v in "${@:-0:1}"; do
  IFS=: read rtile panel cpw <<< "$v"; export PFM_TF_ROWTILE=$rtile PFM_TF_PANEL=$panel PFM_TF_CPW=${cpw:-0}
  rocprofv3 --kernel-trace --stats -d gpurun_out/lin_$tag/rt$v -o p --output-format csv -- python3 tests/diag/tf_time.py ${PFM_DIAG_B:-128} 1 > gpurun_out/lin_$tag/rt$v.log 2>&1 || exit 1
done
