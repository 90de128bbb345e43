for sp in 0 1; do for bn in 64 128; do for rt in 32 64; do
  PFM_SPLIT_STREAMS=$sp PFM_TF_BN=$bn PFM_TF_ROWTILE=$rt python bench_secondary.py --workload jetclass --no-cpu-baseline > gpurun_out/sw.json 2> gpurun_out/sw.err
  python -c "
import json
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); print('split $sp bn $bn rt $rt:', round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['frac'],3))"
done; done; done
