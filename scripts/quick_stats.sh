#!/bin/bash
# Kernel-trace statistics of a short bench run (GPU box; a development aid - the per-round profile set is scripts/collect_profiles.sh):
#   bash scripts/quick_stats.sh TAG 2d|3d [extra bench.py args]   -> gpurun_out/qs_TAG_{2d,3d}_kernel_stats.csv + the bench line
TAG=${1:-x}; WHAT=${2:-2d}; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = 3d ]; then ARGS="--workload 3d --dtype bf16 --size 160"; else ARGS=""; fi
rm -rf /tmp/qs_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs_$TAG -- python3 $ROOT/bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" > $OUT/qs_${TAG}_${WHAT}_bench.json 2> $OUT/qs_${TAG}_${WHAT}.err
cp $(ls /tmp/qs_$TAG/*/*kernel_stats.csv | head -1) $OUT/qs_${TAG}_${WHAT}_kernel_stats.csv
rm -rf /tmp/qs_$TAG
