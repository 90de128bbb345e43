#!/bin/bash
# Secondary-configuration timings + rocprofv3 kernel summaries (BASELINE cfg 4 and cfg 5), written under gpurun_out/.
# Run on the GPU box:  bash tests/diag/collect_profiles.sh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/secondary
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tests/diag/tf_time.py 128 100 train > $O/cfg4_timings.txt 2>&1; python3 $R/tests/diag/tf_time.py 128 100 valid >> $O/cfg4_timings.txt 2>&1
python3 $R/tests/diag/wide_time.py 256 100 train > $O/cfg5_timings.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4 -o cfg4 -- python3 $R/tests/diag/tf_time.py 128 3 train > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5 -o cfg5 -- python3 $R/tests/diag/wide_time.py 256 3 train > /dev/null 2>&1
rm -f $O/cfg4/*trace.csv $O/cfg5/*trace.csv
cat $O/cfg4_timings.txt $O/cfg5_timings.txt
