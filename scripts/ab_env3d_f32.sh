#!/bin/bash
# Same-box A/B of cfg4 (3-D fp32 128^3 train step) between two settings of the SAME library: arm "prev" runs with the environment given on the command line, arm "new"
# without it; two interleaved rounds, one process per run.   bash scripts/ab_env3d_f32.sh ENV=VALUE [...]
cd "$(dirname "$0")/.."
for i in 1 2; do
for arm in prev new; do
  if [ $arm = prev ]; then env "$@" python bench.py --workload 3d --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab3f_$arm.json 2>/dev/null
  else python bench.py --workload 3d --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab3f_$arm.json 2>/dev/null; fi
  python -c "
import json; d=json.load(open('gpurun_out/ab3f_$arm.json')); k=d['kernels']; print('$arm', d['value'], d['ms_per_step'], {n.split('/')[0][:5] + n[-5:]: v['ms_per_step'] for n, v in k.items()}, 'mfma', d['mfma_kernel_ms_per_step'], 'loss', d['config'].get('final_loss'))"
done; done
