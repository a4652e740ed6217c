"""The committed bench line and traffic profile keep the measurement contract: one JSON object with the driver's keys, a `roofline` and a `cpu_baseline` object, and a
`profiles/traffic.json` that was measured on THIS tree's conv / wgrad kernel sources (otherwise bench.py would print `traffic: null` on the driver's run: re-run the PMC
passes of profiles/README.md after touching those files)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_bench_line():
    """the newest round's full default line (scripts/collect_profiles.sh prints it after the traffic files of the same collection are in place)"""
    import glob
    import re
    rounds = [int(m.group(1)) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")) for m in [re.match(r"r(\d+)_bench\.json$", os.path.basename(f))] if m]
    return json.load(open(os.path.join(ROOT, "profiles", f"r{max(rounds):02d}_bench.json")))


def test_committed_bench_line_has_the_contract_keys():
    d = _newest_bench_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) < 0.5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_traffic_profile_belongs_to_this_tree():
    from mdeical_image_segmentation_amd import _lib
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t["source_hash"] == _lib.source_hash(_lib.TRAFFIC_SOURCES), "profiles/traffic.json was measured on other conv / wgrad kernel sources: re-run the PMC passes"
    ent = t["kernels"]["conv_igemm/bf16/k3/2d/bn128"]
    assert ent["hbm_read_bytes_per_launch"] > 0 and ent["hbm_write_bytes_per_launch"] > 0
    d = _newest_bench_line()
    assert d["roofline"]["traffic"] == ent["hbm_read_bytes_per_launch"] + ent["hbm_write_bytes_per_launch"]
    # cfg4 (the 3-D half of the metric): its own traffic file, hashed over the fp32 3-D kernel sources, and the committed line carries it next to its CPU baseline
    t3 = json.load(open(os.path.join(ROOT, "profiles", "traffic_3d_f32.json")))
    assert t3["source_hash"] == _lib.source_hash(_lib.TRAFFIC_SOURCES_3D_F32), "profiles/traffic_3d_f32.json was measured on other fp32 3-D kernel sources: re-run scripts/collect_profiles.sh"
    leg = d["extra"]["unet3d_cfg4_f32_128"]
    e3 = t3["kernels"][leg["roofline"]["key"]]
    assert leg["roofline"]["traffic"] == e3["hbm_read_bytes_per_launch"] + e3["hbm_write_bytes_per_launch"]
    assert leg["cpu_baseline"]["unit"] == "volumes/s" and leg["cpu_baseline"]["value"] > 0 and "kernels" in leg


def test_clock_sampler_never_fails_the_benchmark():
    """bench.ClockSampler reads the amdgpu hwmon files of the card while the timed steps run; where there is no readable hwmon (this container, a box with other
    permissions) it must quietly report nothing - and attach_clock must leave the roofline alone"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    s = bench.ClockSampler("cuda:0", period=0.005)
    s.start()
    out = s.stop()
    assert out is None or (out["samples"] >= 1 and out["sclk_mhz_mean"] > 0)
    roof = {"peak": 2500.0, "achieved": 1000.0}
    bench.LAST_CLOCK = None
    bench.attach_clock(roof)
    assert "peak_at_held_clock" not in roof
    assert roof["peak_measured_pure_mfma"] == 2105.0 and abs(roof["frac_of_measured_peak"] - 1000.0 / 2105.0) < 1e-3
    bench.LAST_CLOCK = {"sclk_mhz_mean": 2100.0}
    bench.attach_clock(roof)
    assert roof["peak_at_held_clock"] == 2187.5 and abs(roof["frac_of_held_clock_peak"] - 1000.0 / 2187.5) < 1e-3
    bench.LAST_CLOCK = None
