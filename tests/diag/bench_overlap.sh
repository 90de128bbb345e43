for o in 2 3 4; do python bench.py --steps 30 --warmup 5 --overlap $o 2>/dev/null > gpurun_out/ovl_$o.json; python -c "
import json,sys;d=json.loads(open('gpurun_out/ovl_$o.json').read().strip().splitlines()[-1]);print('overlap $o',d['ms_per_step'],d['value'],d['sample_ms'],d['roofline']['frac'])"; done
