"""profiles/traffic.json from a PMC summary (scripts/pmc_summary.py) of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing`:
HBM bytes per launch of the bench's dominant kernel KEYS, launch-weighted over the kernel symbols that make up a key, stamped with the hash of the
kernel sources (so bench.py only reports the figure for the build it was measured on).

    python scripts/make_traffic.py profiles/r03_pmc_summary.json [name of the committed summary] > profiles/traffic.json
    python scripts/make_traffic.py profiles/r05_3d_f32_pmc_summary.json r05_3d_f32_pmc_summary.json 3d_f32 > profiles/traffic_3d_f32.json      (cfg4: bench.py --workload 3d --dtype f32)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import _lib  # noqa: E402

KEYS = {   # bench key -> kernel symbols (prefix match on the rocprof name) whose launches carry that key at the benchmark shapes
    "conv_igemm/bf16/k3/2d/bn128": ["void conv_ppc_kernel<8, 4,", "void conv_pp_kernel<8>", "void conv_pp_kernel<4>"],
    "wgrad/bf16/k3/2d": ["void wgrad_pp_stream_kernel<", "wgrad_pp_stream_kernel", "void wgrad_pp_row_kernel<", "void wgrad_pp_wide_kernel<", "wgrad_pp_wide_kernel", "void wgrad_pp_kernel<2>"],
    "conv_igemm/bf16/k3/2d/bn64": ["void conv_ppd_kernel<", "conv64_ws_kernel", "void conv_ppc_kernel<8, 2,"],
    "conv_igemm/bf16/k1/2d/bn128": ["void gemm1_pp_kernel<"],
    "wgrad/bf16/k1/2d": ["wgrad1_pp_kernel"],
}
KEYS_3D_F32 = {   # cfg4: UNet3D(1,3), 2 x 128^3 fp32
    "conv_igemm/f32/k3/3d/bn64": ["void conv3d_f32_kernel<2>"],
    "conv_igemm/f32/k3/3d/bn128": ["void conv3d_f32_kernel<4>"],
    "conv_igemm/f32/k3/3d/bn32": ["void conv3d_f32_kernel<1>"],
    "wgrad/f32/k3/3d": ["void wgrad_f32_stream_kernel<true"],
}


def main(path, profile_name=None, preset="2d"):
    summ = json.load(open(path))
    if preset == "3d_f32":
        keys, sources, batch, size = KEYS_3D_F32, _lib.TRAFFIC_SOURCES_3D_F32, 2, 128
    else:
        keys, sources, batch, size = KEYS, _lib.TRAFFIC_SOURCES, 32, 512
    out = {"source_hash": _lib.source_hash(sources), "hashed_sources": list(sources), "batch": batch, "size": size,
           "profile": profile_name or os.path.basename(path), "kernels": {}}
    for key, syms in keys.items():
        rd = wr = n = 0.0
        used = []
        for e in summ:
            if any(e["kernel"].startswith(s) for s in syms) and "hbm_read_bytes_per_launch_x2_corrected" in e and "hbm_write_bytes_per_launch" in e:
                k = e["launches"]
                rd += e["hbm_read_bytes_per_launch_x2_corrected"] * k
                wr += e["hbm_write_bytes_per_launch"] * k
                n += k
                used.append(f"{e['kernel'][:40]} x{k}")
        if n:
            out["kernels"][key] = {"hbm_read_bytes_per_launch": int(rd / n), "hbm_write_bytes_per_launch": int(wr / n), "launches_profiled": int(n),
                                   "symbols": used, "note": "FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, separate --pmc passes"}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
