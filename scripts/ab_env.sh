#!/bin/bash
# Same-box A/B of the whole 2-D train step between two settings of the SAME library: arm "prev" runs with the environment given on the command line (a dispatch switch
# that selects the older kernel, e.g. MIS_GEMM1_NOPP=1), arm "new" without it; two interleaved rounds, one process per run.   bash scripts/ab_env.sh ENV=VALUE [...]
cd "$(dirname "$0")/.."
for i in 1 2; do
for arm in prev new; do
  if [ $arm = prev ]; then env "$@" python bench.py --no-cpu-baseline --no-extra > gpurun_out/ab_$arm.json 2>/dev/null
  else python bench.py --no-cpu-baseline --no-extra > gpurun_out/ab_$arm.json 2>/dev/null; fi
  python -c "
import json; d=json.load(open('gpurun_out/ab_$arm.json')); k=d['kernels']; print('$arm', d['value'], d['ms_per_step'], 'conv', k['conv_igemm/bf16/k3/2d/bn128']['ms_per_step'], 'wgrad', k['wgrad/bf16/k3/2d']['ms_per_step'], 'k1', sum(v['ms_per_step'] for n, v in k.items() if n.startswith('conv_igemm') and '/k1/' in n), 'mfma', d['mfma_kernel_ms_per_step'])"
done; done
