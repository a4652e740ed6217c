#!/bin/bash
# Collect the per-round profile set on the GPU box (one gpurun call; repo root = $GRAFT_REPO_ROOT or the script's parent):
#   bash scripts/collect_profiles.sh r06         -> gpurun_out/prof_r06/{bench.json, kernel_stats.csv, bench_under_rocprof.json, pmc_summary.json, traffic.json,
#                                                   3d_bf16_160_*, 3d_f32_* (+ traffic_3d_f32.json), power_*.csv, noaug_delta.txt}
# rocprofv3 runs the program itself after `--` (python3 bench.py ...), counters in their own passes (--pmc with --kernel-trace only), as profiles/README.md prescribes.
# Order (round 6, VERDICT r5 #6: every file of a set from ONE tree, and the files must agree with each other): the PMC passes FIRST - their traffic*.json are installed
# into the box's profiles/ at once, so that every bench line printed afterwards (the lines under the kernel trace, the full default line) carries the traffic figure and
# the source hash of THIS collection; tests/test_profiles_consistency.py holds the set to that.
set -e
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B2D="--steps 3 --warmup 1 --no-cpu-baseline --no-extra"
B3D="--workload 3d --dtype bf16 --size 160 --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
F3D="--workload 3d --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
P2D="--steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing"
P3D="--workload 3d --dtype bf16 --size 160 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing"
PF3D="--workload 3d --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing"
echo "[1/14] 2-D PMC: SQ"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py $P2D > /dev/null 2> $OUT/pmc_sq.err
echo "[2/14] 2-D PMC: FETCH"; rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $P2D > /dev/null 2> $OUT/pmc_fetch.err
echo "[3/14] 2-D PMC: WRITE"; rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $P2D > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/scripts/pmc_summary.py $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.json
(cd $ROOT && python3 scripts/make_traffic.py $OUT/pmc_summary.json ${TAG}_pmc_summary.json > $OUT/traffic.json)
cp $OUT/traffic.json $ROOT/profiles/traffic.json
echo "[4/14] cfg4 PMC: SQ"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmcf32_sq -- python3 $ROOT/bench.py $PF3D > /dev/null 2> $OUT/pmcf32_sq.err
echo "[5/14] cfg4 PMC: FETCH / WRITE"; rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmcf32_fetch -- python3 $ROOT/bench.py $PF3D > /dev/null 2> $OUT/pmcf32_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcf32_write -- python3 $ROOT/bench.py $PF3D > /dev/null 2> $OUT/pmcf32_write.err
python3 $ROOT/scripts/pmc_summary.py $OUT/pmcf32_sq $OUT/pmcf32_fetch $OUT/pmcf32_write > $OUT/3d_f32_pmc_summary.json
(cd $ROOT && python3 scripts/make_traffic.py $OUT/3d_f32_pmc_summary.json ${TAG}_3d_f32_pmc_summary.json 3d_f32 > $OUT/traffic_3d_f32.json)
cp $OUT/traffic_3d_f32.json $ROOT/profiles/traffic_3d_f32.json
echo "[6/14] 3-D bf16 PMC: SQ"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc3d_sq -- python3 $ROOT/bench.py $P3D > /dev/null 2> $OUT/pmc3d_sq.err
echo "[7/14] 3-D bf16 PMC: FETCH / WRITE"; rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc3d_fetch -- python3 $ROOT/bench.py $P3D > /dev/null 2> $OUT/pmc3d_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc3d_write -- python3 $ROOT/bench.py $P3D > /dev/null 2> $OUT/pmc3d_write.err
python3 $ROOT/scripts/pmc_summary.py $OUT/pmc3d_sq $OUT/pmc3d_fetch $OUT/pmc3d_write > $OUT/3d_bf16_160_pmc_summary.json
echo "[8/14] 2-D kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks2d -- python3 $ROOT/bench.py $B2D > $OUT/bench_under_rocprof.json 2> $OUT/ks2d.err
cp $(ls $OUT/ks2d/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo "[9/14] 3-D bf16 kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks3d -- python3 $ROOT/bench.py $B3D > $OUT/3d_bf16_160_bench_under_rocprof.json 2> $OUT/ks3d.err
cp $(ls $OUT/ks3d/*/*kernel_stats.csv | head -1) $OUT/3d_bf16_160_kernel_stats.csv
echo "[10/14] cfg4 kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ksf32 -- python3 $ROOT/bench.py $F3D > $OUT/3d_f32_bench_under_rocprof.json 2> $OUT/ksf32.err
cp $(ls $OUT/ksf32/*/*kernel_stats.csv | head -1) $OUT/3d_f32_kernel_stats.csv
echo "[11/14] full default bench line (what the driver runs)"; python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
# round 6: socket power / shader clock beside 50 steps of each workload (VERDICT r5 "settle the clock"), and what the on-device augmentation costs the 3-D steps
echo "[12/14] power / sclk probes"
for w in 2d 3d_bf16 3d_f32; do python3 $ROOT/scripts/power_probe.py $w 50 > $OUT/power_$w.csv 2> $OUT/power_$w.err; done
echo "[13/14] MISAMD_BENCH_NOAUG A/B (cfg4, cfg5's shape; two runs each way)"
NA="--steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
( for i in 1 2; do
    for leg in "cfg4:--workload 3d --dtype f32" "cfg5shape:--workload 3d --dtype bf16 --size 160"; do
      name=${leg%%:*}; flags=${leg#*:}
      a=$(python3 $ROOT/bench.py $flags $NA 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'])")
      b=$(MISAMD_BENCH_NOAUG=1 python3 $ROOT/bench.py $flags $NA 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'])")
      echo "$name run $i: with augmentation (vol/s, ms/step) $a | MISAMD_BENCH_NOAUG=1 $b"
    done
  done ) > $OUT/noaug_delta.txt 2>&1
echo "[14/14] what a pure MFMA stream delivers (scripts/mfma_peak.hip)"
(cd $ROOT && bash scripts/mfma_peak.sh > /dev/null 2>&1 && cp gpurun_out/mfma_peak.txt $OUT/mfma_peak.txt) || echo "mfma_peak failed" > $OUT/mfma_peak.txt
# the raw counter dumps are large: keep the summaries only
rm -rf $OUT/ks2d $OUT/ks3d $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc3d_sq $OUT/pmc3d_fetch $OUT/pmc3d_write $OUT/ksf32 $OUT/pmcf32_sq $OUT/pmcf32_fetch $OUT/pmcf32_write
ls -la $OUT
