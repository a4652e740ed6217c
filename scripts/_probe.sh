mkdir -p gpurun_out/power
for w in 2d 3d_bf16 3d_f32; do python scripts/power_probe.py $w 50 > gpurun_out/power/r06_power_$w.csv 2> gpurun_out/power/$w.err; head -3 gpurun_out/power/r06_power_$w.csv; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/r6e_2d.json 2>gpurun_out/r6e_2d.err
MISAMD_BENCH_NO_CLOCK=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/r6e_2d_noclock.json 2>/dev/null
python - <<EOP
import json
for f in ("gpurun_out/r6e_2d.json","gpurun_out/r6e_2d_noclock.json"):
    o=json.load(open(f)); print(f, o["value"], o["ms_per_step"], o.get("clock"), {k:o["roofline"].get(k) for k in ("achieved","frac","peak_at_held_clock","frac_of_held_clock_peak")})
EOP
