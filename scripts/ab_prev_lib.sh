#!/bin/bash
# Same-box A/B of the whole 2-D train step between two builds of the library (GPU box): the tree's libmisamd.so against build_ab/libmisamd_prev.so
# (built by hand from another commit: `git archive <commit> mdeical_image_segmentation_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/.../csrc`, copied to build_ab/, outside the package, not next to
# the shipped library - *.so files are git-ignored but travel with gpurun), two interleaved rounds, one process per run.  Box-to-box variance is +-2 %: only this kind
# of comparison decides a kernel change.   bash scripts/ab_prev_lib.sh [ENV=VALUE ...]   (extra environment of the "prev" arm, e.g. MISAMD_REDUCE_PER_LAYER=1 when the older library lacks mis_wgrad_reduce_batch; without build_ab/libmisamd_prev.so the arm "prev" is this library under that environment)
cd "$(dirname "$0")/.."
PREV=$PWD/build_ab/libmisamd_prev.so
for i in 1 2; do
for arm in prev new; do
  if [ $arm = prev ]; then
    if [ -f "$PREV" ]; then env MISAMD_LIB=$PREV "$@" python bench.py --no-cpu-baseline --no-extra > gpurun_out/ab_$arm.json 2>/dev/null
    else env "$@" python bench.py --no-cpu-baseline --no-extra > gpurun_out/ab_$arm.json 2>/dev/null; fi
  else
    python bench.py --no-cpu-baseline --no-extra > gpurun_out/ab_$arm.json 2>/dev/null
  fi
  python -c "
import json; d=json.load(open('gpurun_out/ab_$arm.json')); k=d['kernels']; print('$arm', d['value'], d['ms_per_step'], 'conv', k['conv_igemm/bf16/k3/2d/bn128']['ms_per_step'], 'wgrad', k['wgrad/bf16/k3/2d']['ms_per_step'], 'bn64', k['conv_igemm/bf16/k3/2d/bn64']['ms_per_step'], 'mfma', d['mfma_kernel_ms_per_step'])"
done; done
