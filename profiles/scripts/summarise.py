"""Condenses the rocprofv3 CSV output of profile_set.sh into the small files committed under profiles/."""
import csv, glob, json, os, sys
from collections import defaultdict

out, summ, rnd = sys.argv[1], sys.argv[2], sys.argv[3]


def find(d, pat):
    r = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    for k in ("k_lookup_exact", "k_mask", "k_inexact", "k_finalize", "k_regions_easy", "k_region_order", "k_regions", "k_stitch", "k_enum", "k_myers_batch"):
        if k in name:
            return k
    if "index_elementwise" in name or "gather" in name.lower():
        return "torch_index"
    if "copy" in name.lower() or "Copy" in name:
        return "torch_copy"
    return name[:60]


def pmc(d, pat):
    """per kernel: launches, mean counter value per launch (summed over the agent's XCD rows of one dispatch)"""
    f = find(d, pat)
    res = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        res[short(r["Kernel_Name"])][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    o = {}
    for k, cs in res.items():
        o[k] = {c: {"launches": len(v), "mean_per_launch": sum(v.values()) / len(v), "max_per_launch": max(v.values())} for c, v in cs.items()}
    return o


# kernel stats
f = find("stats", "*kernel_stats.csv")
if f:
    rows = list(csv.reader(open(f)))
    with open(os.path.join(summ, rnd + "_kernel_stats.csv"), "w") as g:
        csv.writer(g).writerows(rows)
f = find("stats_serial", "*kernel_stats.csv")
if f:
    rows = list(csv.reader(open(f)))
    with open(os.path.join(summ, rnd + "_kernel_stats_serial.csv"), "w") as g:
        csv.writer(g).writerows(rows)
res = {"units": "FETCH_SIZE / WRITE_SIZE are KB as reported by rocprofv3; *_bytes fields are converted (x1024)"}
fe, wr, sq = pmc("fetch", "*counter_collection.csv"), pmc("write", "*counter_collection.csv"), pmc("sq", "*counter_collection.csv")
cf, cw = pmc("calib_fetch", "*counter_collection.csv"), pmc("calib_write", "*counter_collection.csv")
kern = {}
for k in ("k_lookup_exact", "k_mask", "k_inexact", "k_finalize", "k_enum", "k_region_order", "k_regions_easy", "k_regions", "k_stitch"):
    e = {}
    if k in fe and "FETCH_SIZE" in fe[k]:
        e["fetch_bytes_per_launch_raw"] = fe[k]["FETCH_SIZE"]["mean_per_launch"] * 1024.0
        e["launches"] = fe[k]["FETCH_SIZE"]["launches"]
    if k in wr and "WRITE_SIZE" in wr[k]:
        e["write_bytes_per_launch_raw"] = wr[k]["WRITE_SIZE"]["mean_per_launch"] * 1024.0
    if k in sq:
        e["sq"] = {c: v["mean_per_launch"] for c, v in sq[k].items()}
    kern[k] = e
res["kernels"] = kern
N = 1 << 26
cal = {"accesses": N}
if "torch_index" in cf and "FETCH_SIZE" in cf["torch_index"]:
    b = cf["torch_index"]["FETCH_SIZE"]["max_per_launch"] * 1024.0
    cal["gather_fetch_bytes_raw"] = b
    cal["gather_fetch_raw_bytes_per_element"] = b / N  # one random 8-byte read + 8 bytes of the coalesced index stream per element
if "torch_copy" in cf and "FETCH_SIZE" in cf["torch_copy"]:
    b = cf["torch_copy"]["FETCH_SIZE"]["max_per_launch"] * 1024.0
    cal["copy_fetch_bytes_raw"] = b
    cal["copy_fetch_raw_over_true"] = b / (8.0 * N)
if "torch_index" in cw and "WRITE_SIZE" in cw["torch_index"]:
    cal["gather_write_raw_over_true"] = cw["torch_index"]["WRITE_SIZE"]["max_per_launch"] * 1024.0 / (8.0 * N)
if "torch_copy" in cw and "WRITE_SIZE" in cw["torch_copy"]:
    cal["copy_write_raw_over_true"] = cw["torch_copy"]["WRITE_SIZE"]["max_per_launch"] * 1024.0 / (8.0 * N)
cal["reading"] = "coalesced 16 B/lane reads: FETCH_SIZE x 2 = bytes; random 8-byte reads: 64 B tallied per read; WRITE_SIZE = bytes for coalesced writes"
res["calibration"] = cal
json.dump(res, open(os.path.join(summ, rnd + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps(res["kernels"], indent=1)[:3000])
print(json.dumps({k: v for k, v in cal.items() if k != "all_calib_kernels"}, indent=1))
