#!/bin/bash
# configs[4] on DISTINCT reads under the profiler (round-5 review item 4: k_lookup_exact / k_inexact time and fetched bytes on tickets that are not one ticket repeated):
# bench_config4.py builds (or finds) the 3 Gb index, then rocprofv3 kernel stats and the two byte counters, each in its own pass, over four different tickets run once each.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WD=${1:-/tmp/rtk_c4}; OUT=$PWD/gpurun_out/r06_c4pmc; mkdir -p $OUT
timeout 1500 python bench_config4.py $OUT/config4.json 3000 30 16 128 $WD > $OUT/config4.log 2>&1
STEPS="python profiles/scripts/r06/config4_distinct_steps.py $WD 4"
timeout 600 $STEPS > $OUT/steps.log 2>&1; tail -4 $OUT/steps.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $STEPS > /dev/null 2> $OUT/stats.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $STEPS > /dev/null 2> $OUT/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $STEPS > /dev/null 2> $OUT/write.err
python - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
def files(sub, pat):
    return glob.glob(out + "/" + sub + "/**/" + pat, recursive=True)
res = collections.OrderedDict()
for f in files("stats", "*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Name", "").split("(")[0]
        if n.startswith("k_"): res.setdefault(n, {}).update({"calls": int(r["Calls"]), "avg_ms": round(float(r["AverageNs"]) / 1e6, 3), "min_ms": round(float(r["MinNs"]) / 1e6, 3)})
for sub, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    per = collections.defaultdict(lambda: collections.OrderedDict())
    for f in files(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == key:
                n = r["Kernel_Name"].split("(")[0]; d = int(r["Dispatch_Id"])
                per[n][d] = per[n].get(d, 0.0) + float(r["Counter_Value"])  # (one row per XCD of a dispatch: summed)
    for n, dd in per.items():
        if not n.startswith("k_"): continue
        v = [dd[d] for d in sorted(dd)]
        v = v[1:] if len(v) > 1 else v  # (the first launch of every kernel belongs to the warm ticket)
        res.setdefault(n, {})[key + "_KB_per_launch"] = round(sum(v) / len(v)); res[n][key + "_launches"] = len(v)
for n, d in res.items():
    if "FETCH_SIZE_KB_per_launch" in d and "WRITE_SIZE_KB_per_launch" in d:
        d["traffic_GB_2xFETCH_plus_WRITE"] = round((2.0 * d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * 1024 / 1e9, 2)
json.dump({"what": "configs[4] (3 Gb graph, 160.6 GB resident), four DIFFERENT 64 Mb tickets run once each, one at a time: rocprofv3 kernel stats and FETCH_SIZE / WRITE_SIZE passes (KB as rocprofv3 reports them; coalesced reads are tallied at half their bytes on gfx950, hence 2 x FETCH: profiles/scripts/summarise.py, calib_gather.py)", "kernels": res}, open(out + "/../r06_config4_distinct_pmc.json", "w"), indent=1)
print(json.dumps(res)[:2000])
PY
find $OUT -name "*.csv" -size +4M -delete
