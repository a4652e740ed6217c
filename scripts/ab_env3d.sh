#!/bin/bash
# Same-box A/B of the cfg5-shaped 3-D train step (2 x 160^3 bf16) between two settings of the SAME library: arm "prev" runs with the environment given on the command line
# (e.g. MISAMD_GN_BWD_UNFUSED=1), arm "new" without it; two interleaved rounds, one process per run.   bash scripts/ab_env3d.sh ENV=VALUE [...]
cd "$(dirname "$0")/.."
for i in 1 2; do
for arm in prev new; do
  if [ $arm = prev ]; then env "$@" python bench.py --workload 3d --dtype bf16 --size 160 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab3_$arm.json 2>/dev/null
  else python bench.py --workload 3d --dtype bf16 --size 160 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab3_$arm.json 2>/dev/null; fi
  python -c "
import json; d=json.load(open('gpurun_out/ab3_$arm.json')); k=d['kernels']; print('$arm', d['value'], d['ms_per_step'], {n.split('/')[0][:5] + n[-5:]: v['ms_per_step'] for n, v in k.items()}, 'mfma', d['mfma_kernel_ms_per_step'])"
done; done
