"""Summarise rocprofv3 --pmc passes (CSV counter_collection files) per kernel into one JSON.

    python scripts/pmc_summary.py gpurun_out/pmc_sq gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_summary.json

Each directory is the -d output of one `rocprofv3 --pmc ... --kernel-trace --output-format csv -- python3 bench.py ...` pass.
FETCH_SIZE is reported raw (KB) and as bytes with the gfx950 x2 correction (MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0][:110]


def main(dirs):
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(lambda: defaultdict(int))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                c = r["Counter_Name"]
                agg[k][c] += float(r["Counter_Value"])
                key = (r["Dispatch_Id"], c)
                if key not in seen:
                    seen.add(key)
                    launches[k][c] += 1
    out = []
    for k, cs in agg.items():
        e = {"kernel": k}
        for c, v in cs.items():
            n = max(launches[k][c], 1)
            e[c + "_per_launch"] = v / n
            e["launches"] = n
        if "FETCH_SIZE_per_launch" in e:
            e["hbm_read_bytes_per_launch_x2_corrected"] = e["FETCH_SIZE_per_launch"] * 1024 * 2
        if "WRITE_SIZE_per_launch" in e:
            e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE_per_launch"] * 1024
        if "SQ_WAVE_CYCLES_per_launch" in e:
            w = e["SQ_WAVE_CYCLES_per_launch"]
            for c, label in (("SQ_WAIT_ANY", "wait_any_pct"), ("SQ_WAIT_INST_ANY", "wait_inst_pct"), ("SQ_ACTIVE_INST_ANY", "active_pct")):
                if c + "_per_launch" in e:
                    e[label] = round(100 * e[c + "_per_launch"] / w, 1)
        out.append(e)
    out.sort(key=lambda e: -e.get("GRBM_GUI_ACTIVE_per_launch", 0) * e.get("launches", 1))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
