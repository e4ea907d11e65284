"""Prints the handful of numbers of a bench.py line that a session's log should show."""
import json
import sys

for ln in open(sys.argv[1]):
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    r = d.get("roofline") or {}
    keys = ("value", "ms_per_step", "ms_per_step_min", "time_to_stable_cut_ms", "decided", "cut_size", "votes_winner", "n_ranks_seen")
    print({k: d.get(k) for k in keys})
    print({"kernel_ms": r.get("kernel_ms"), "frac": r.get("frac"), "traffic_over_bytes": r.get("traffic_over_bytes"), "kernel": r.get("kernel")})
    for k in ("parity_checked", "repetitions", "cpu_baseline"):
        if d.get(k) is not None:
            print(k, json.dumps(d[k])[:400])
