#!/bin/bash
mkdir -p gpurun_out
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/b_plain.json 2> gpurun_out/b_plain.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --force-collective > gpurun_out/b_coll.json 2> gpurun_out/b_coll.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload lrw-xt > gpurun_out/b_xt.json 2> gpurun_out/b_xt.err
for f in b_plain b_coll b_xt; do echo "== $f"; tail -1 gpurun_out/$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('collective'), d['final_loss']); print({k:(v['ms_per_step'],v['tflops']) for k,v in d['roofline']['per_kernel'].items()})" || tail -3 gpurun_out/$f.err; done
