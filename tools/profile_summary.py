"""Condense rocprofv3 output (gpurun_out/<tag>_stats, <tag>_pmc_fetch, <tag>_pmc_write) into the small files kept under
profiles/:  <tag>_kernel_stats.csv (per-kernel calls / avg / share, short names) and <tag>_pmc_hbm.json (per-launch HBM
traffic per kernel, corrected as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE are in KiB, collected in
separate passes; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B, so it is doubled).

    python tools/profile_summary.py r01
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name.split("(")[0][-60:]


def find(tag, sub, pattern):
    hits = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{sub}", "**", pattern), recursive=True)
    return hits[0] if hits else None


def main(tag):
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stats = find(tag, "stats", "*kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats)))
        with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w") as f:
            f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
            for r in rows:
                f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.1f},{float(r['AverageNs']) / 1e3:.2f},"
                        f"{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f},{r['Percentage']}\n")
        print("wrote profiles/%s_kernel_stats.csv" % tag)
    pmc = {}
    for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        path = find(tag, sub, "*counter_collection.csv")
        if not path:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                agg[re.sub(r"<.*", "", short(r["Kernel_Name"]))].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc.setdefault(k, {})[counter + "_KiB_avg"] = sum(v) / len(v)
            pmc[k]["launches_" + counter] = len(v)
    for k, d in pmc.items():
        if "FETCH_SIZE_KiB_avg" in d and "WRITE_SIZE_KiB_avg" in d:
            d["hbm_read_bytes"] = 2.0 * d["FETCH_SIZE_KiB_avg"] * 1024.0   # gfx950: FETCH_SIZE reports 1/2 of wide reads
            d["hbm_write_bytes"] = d["WRITE_SIZE_KiB_avg"] * 1024.0
            d["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
    if pmc:
        out = {"_note": "per-launch averages; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of "
                        "`python bench.py --steps 20 --warmup 5 --no-cpu-baseline`; FETCH_SIZE doubled (gfx950 correction)",
               "kernels": {k: v for k, v in sorted(pmc.items()) if k.startswith("k_")}}
        json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm.json"), "w"), indent=1)
        print("wrote profiles/%s_pmc_hbm.json" % tag)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
