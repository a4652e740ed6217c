#!/bin/bash
# Same-box A/B of cfg3's per-GPU shape (UNet(3,4), bs 32, 512^2 bf16) between two settings of the SAME library (arm "prev" = with the given environment).
cd "$(dirname "$0")/.."
for i in 1 2; do
for arm in prev new; do
  if [ $arm = prev ]; then env "$@" python bench.py --net 3x4 --no-cpu-baseline --no-extra > gpurun_out/ab34_$arm.json 2>/dev/null
  else python bench.py --net 3x4 --no-cpu-baseline --no-extra > gpurun_out/ab34_$arm.json 2>/dev/null; fi
  python -c "
import json; d=json.load(open('gpurun_out/ab34_$arm.json')); print('$arm', d['value'], d['ms_per_step'], d['loss_per_step'][-1])"
done; done
