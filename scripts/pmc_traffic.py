"""Turns a rocprofv3 `--pmc FETCH_SIZE` (and optionally WRITE_SIZE) pass over scripts/prof_tally.py into HBM bytes per
tally launch.  FETCH_SIZE is calibrated on the streaming probe kernel of the same run, whose byte count is known
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM: on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x;
calibrate on a known byte count in your own access pattern).  usage: pmc_traffic.py <pmc_counter_collection.csv>
<stream_bytes> [out.json]"""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
stream_bytes = float(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = "tally" if "tally_population" in r["Kernel_Name"] else ("probe" if "stream_probe" in r["Kernel_Name"] else None)
    if k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in agg.items():
    for c, v in d.items():
        out["%s_%s_avg" % (k, c)] = sum(v) / len(v)
if "probe_FETCH_SIZE_avg" in out and out["probe_FETCH_SIZE_avg"] > 0:
    raw_probe = out["probe_FETCH_SIZE_avg"] * 1024.0  # FETCH_SIZE is reported in KiB
    out["fetch_calibration_factor"] = stream_bytes / raw_probe
    out["hbm_bytes_per_launch"] = int(out["tally_FETCH_SIZE_avg"] * 1024.0 * out["fetch_calibration_factor"])
    out["algorithmic_bytes_per_launch"] = int(stream_bytes)
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
