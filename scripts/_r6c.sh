python -m pytest tests/test_gpu_dispatch_parity.py tests/test_gpu_engine3d.py tests/test_gpu_deconv3d.py tests/test_gpu_fullsize.py tests/test_gpu_graph.py -q -m gpu -k "f32 or F32 or statistics or engine3d or deconv or fullsize or graph or tile_queue" > gpurun_out/r6c_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r6c_tests.log
B="--workload 3d --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
MISAMD_NO_EPI_STATS=1 python bench.py $B > gpurun_out/r6c_cfg4_noepi.json 2>gpurun_out/r6c_cfg4_noepi.err
python bench.py $B > gpurun_out/r6c_cfg4_new.json 2>gpurun_out/r6c_cfg4_new.err
MISAMD_NO_EPI_STATS=1 python bench.py $B > gpurun_out/r6c_cfg4_noepi2.json 2>gpurun_out/r6c_cfg4_noepi.err
python bench.py $B > gpurun_out/r6c_cfg4_new2.json 2>gpurun_out/r6c_cfg4_new.err
python - <<EOP
import json
for n in ("noepi","new","noepi2","new2"):
    try:
        o=json.load(open(f"gpurun_out/r6c_cfg4_{n}.json"))
        print(n, o["value"], o["ms_per_step"], o["mfma_kernel_ms_per_step"], json.dumps(o.get("kernels")))
    except Exception as e: print(n, "ERR", e)
EOP
