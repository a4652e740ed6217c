"""The committed profile set of the NEWEST round must belong to one tree and to itself: for every profiles/rNN_*bench_under_rocprof.json
  * the launches the bench line reports per key and step are the launches the rocprofv3 kernel-stats CSV of the SAME run counted for those kernel symbols,
  * the average launch time of the dominant key by HIP events agrees with the CSV's (the two clocks of one process),
  * a traffic figure / hash in the line is the one of the traffic file collected beside it,
and the traffic files carry the hash of the kernel sources in this tree (tests/test_bench_contract.py).  CPU only: reads files."""
import csv
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
TRAFFIC_OF = {"": "traffic.json", "3d_f32_": "traffic_3d_f32.json"}


def _newest_round():
    rounds = sorted({int(m.group(1)) for f in glob.glob(os.path.join(PROF, "r*_bench_under_rocprof.json")) for m in [re.match(r"r(\d+)_", os.path.basename(f))] if m})
    return rounds[-1]


def _legs():
    n = _newest_round()
    out = []
    for f in sorted(glob.glob(os.path.join(PROF, f"r{n:02d}_*bench_under_rocprof.json"))):
        prefix = os.path.basename(f)[len(f"r{n:02d}_"):-len("bench_under_rocprof.json")]
        out.append((n, prefix))
    return out


def _calls(csv_path):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(csv_path))}


def _match(calls, sym):
    """CSV rows of a kernel symbol as bench.py names it ("conv_ppc_kernel<8, 4, 0>"; "conv_ppd_head_kernel<C>" = any class count)"""
    base = sym.replace("<C>", "<")
    return [(n, v) for n, v in calls.items() if n.startswith("void " + base) or n.startswith(base)]


@pytest.mark.parametrize("leg", _legs(), ids=lambda l: f"r{l[0]:02d}_{l[1] or '2d_'}")
def test_bench_line_and_kernel_trace_are_one_run(leg):
    n, prefix = leg
    line = json.load(open(os.path.join(PROF, f"r{n:02d}_{prefix}bench_under_rocprof.json")))
    calls = _calls(os.path.join(PROF, f"r{n:02d}_{prefix}kernel_stats.csv"))
    steps_in_trace = line["steps"] + max(line["warmup"], 1)
    roof = line["roofline"]
    checked = 0
    for key, k in line["kernels"].items():
        syms = k.get("launches_by_symbol")
        assert syms, f"{key}: the line does not name its kernel symbols (collected with an older bench.py?)"
        rows = [r for s in syms for r in _match(calls, s)]
        if key != roof["key"] and (not rows or any(not _match(calls, s) for s in syms)):
            continue          # (keys served by conv_igemm.hip's template kernels print under another name in rocprofv3: only the dominant key is mandatory)
        assert rows, f"{key}: none of {list(syms)} is in the kernel trace"
        # a symbol may serve several keys (conv3d_f32_kernel<2>: forward and dgrad launches of one key; conv_ppc_kernel<8, 4, 0>: one key): compare per symbol set
        shared = [k2 for k2, v2 in line["kernels"].items() if k2 != key and set(v2.get("launches_by_symbol", {})) & set(syms)]
        per_step = k["launches_per_step"] + sum(line["kernels"][k2]["launches_per_step"] for k2 in shared)
        total_calls = sum(v[0] for _, v in {r[0]: r for r in rows}.values())
        assert total_calls == per_step * steps_in_trace, (key, list(syms), total_calls, per_step, steps_in_trace)
        checked += 1
        if key == roof["key"] and not shared:
            # HIP events (bracketed steps) against rocprofv3 (all steps of the trace, the warm-up's first launches included): the same kernel, the same process
            avg_csv = sum(v[1] for _, v in {r[0]: r for r in rows}.values()) / total_calls / 1e6
            assert abs(avg_csv - roof["avg_launch_ms"]) <= 0.08 * roof["avg_launch_ms"], (key, avg_csv, roof["avg_launch_ms"])
    assert checked >= 1
    tf = TRAFFIC_OF.get(prefix)
    if tf is not None:
        t = json.load(open(os.path.join(PROF, tf)))
        assert roof.get("traffic_source_hash") == t["source_hash"], f"{prefix}bench_under_rocprof.json was printed on other kernel sources than profiles/{tf} was measured on"


def test_newest_round_has_the_whole_set():
    n = _newest_round()
    have = {os.path.basename(f) for f in glob.glob(os.path.join(PROF, f"r{n:02d}_*"))}
    for need in ("bench.json", "bench_under_rocprof.json", "kernel_stats.csv", "pmc_summary.json", "3d_f32_bench_under_rocprof.json", "3d_f32_kernel_stats.csv",
                 "3d_f32_pmc_summary.json", "3d_bf16_160_bench_under_rocprof.json", "3d_bf16_160_kernel_stats.csv", "3d_bf16_160_pmc_summary.json"):
        assert f"r{n:02d}_{need}" in have, f"profiles/r{n:02d}_{need} is missing: scripts/collect_profiles.sh writes the whole set from one tree"
