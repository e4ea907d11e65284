#!/bin/bash
# One `gpurun` call = one GPU session: scripts/gpu_session.sh <section> [<section> ...], sections run in the order given; everything a
# section prints that is worth keeping goes to gpurun_out/ (merged back by gpurun), a short digest to stdout.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash scripts/gpu_session.sh micro c5probe ab_c3b'
# (round 4 kept one script per session, gpu_r04_a.sh ... _q.sh; their sections live here now.)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
filter() { grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP\|amdgpu.ids"; }
for section in "$@"; do
  echo "==== $section"
  case "$section" in
    tests)        # the whole GPU suite, as the driver runs it
      timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; filter < gpurun_out/pytest_gpu.log | tail -4 ;;
    tests:*)      # a subset: tests:<-k expression>
      timeout 900 python -m pytest tests -m gpu -q -x -k "${section#tests:}" > gpurun_out/pytest_gpu_sub.log 2>&1; filter < gpurun_out/pytest_gpu_sub.log | tail -4 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | filter | tail -2 ;;
    bench)        # the driver's command
      timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; python scripts/bench_digest.py gpurun_out/bench.json ;;
    bench:*)      # bench:<config>[:extra flags]
      IFS=: read -r _ cfg extra <<< "$section"
      timeout 900 python bench.py --gpus 1 --config "$cfg" $extra > "gpurun_out/bench_$cfg.json" 2> "gpurun_out/bench_$cfg.err"; python scripts/bench_digest.py "gpurun_out/bench_$cfg.json"; tail -2 "gpurun_out/bench_$cfg.err" ;;
    micro)        # scripts/micro/gather_rate.hip: what a table lookup costs by where the table lives
      (cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate 2>/dev/null; timeout 200 ./gather_rate) | tee gpurun_out/gather_rate.txt ;;
    c5probe)      # the C5 tally kernel in every prepared build (what bounds it)
      RAPID_AB_ONLY=${C5_ONLY:-default,pnolook,pnoor,pstream} timeout 600 python scripts/c5_probe.py 1000000 1024 5 2> gpurun_out/c5_probe.err | tee gpurun_out/c5_probe.txt; tail -2 gpurun_out/c5_probe.err ;;
    c5phase)      # event counters and phase timers of the C5 tally (needs rapid_amd/librapid_mi355x_timers.so)
      timeout 600 python scripts/c5_phase.py 1000000 1024 2> gpurun_out/c5_phase.err | tee gpurun_out/c5_phase.txt; tail -2 gpurun_out/c5_phase.err ;;
    apply1m)      # where a view change at 10^6 members spends its time (phases with a stream synchronisation each), then untimed-phase totals
      RAPID_TIME_VIEW=1 timeout 300 python scripts/time_apply.py 1000000 > gpurun_out/time_apply_1m_phases.txt 2>&1; tail -14 gpurun_out/time_apply_1m_phases.txt
      timeout 300 python scripts/time_apply.py 1000000 2>&1 | tee gpurun_out/time_apply_1m.txt | tail -4 ;;
    apply1m_kstats)  # the kernels of a view change at 10^6 members (rocprofv3 kernel statistics of scripts/time_apply.py)
      cd /tmp; rm -rf "$R/gpurun_out/prof_apply"
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_apply" -o apply -- python "$R/scripts/time_apply.py" 1000000 > "$R/gpurun_out/apply_traced.txt" 2> "$R/gpurun_out/prof_apply.log"
      cd "$R"; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_apply/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:22]:
    print("%-60s calls %5s avg %10.1f us" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
      ;;
    ab_c3b)       # the C3b tally kernel in every prepared build, interleaved
      RAPID_AB_ONLY=${AB_ONLY:-default,lanedummy,skipdummy} timeout 600 python scripts/ab_variants.py C3b 3 20 2> gpurun_out/ab_c3b.err | tee gpurun_out/ab_c3b.txt | tail -12; tail -2 gpurun_out/ab_c3b.err ;;
    kstats)       # rocprofv3 kernel statistics of the bench command (driver form)
      cd /tmp; rm -rf "$R/gpurun_out/prof_bench"
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench" -o bench -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > "$R/gpurun_out/bench_traced.json" 2> "$R/gpurun_out/prof_bench.log"
      cd "$R"; head -12 gpurun_out/prof_bench/bench_kernel_stats.csv | cut -c1-200 ;;
    kstats:*)     # kstats:<config>[:extra flags] -- kernel statistics of another configuration's bench run
      IFS=: read -r _ cfg extra <<< "$section"
      cd /tmp; rm -rf "$R/gpurun_out/prof_bench_$cfg"
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench_$cfg" -o bench -- python "$R/bench.py" --gpus 1 --config "$cfg" $extra --no-cpu-baseline --no-pmc > "$R/gpurun_out/bench_traced_$cfg.json" 2> "$R/gpurun_out/prof_bench_$cfg.log"
      cd "$R"; python - "$cfg" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/prof_bench_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
for row in list(csv.DictReader(open(f[0])))[:12]:
    print("%-72s calls %6s avg %10.1f us  total %8.1f ms" % (row["Name"][:72], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6))
PY
      ;;
    gaps)         # the launch chain of a round (start / duration / gap of each kernel), from the kstats trace
      python scripts/trace_gaps.py gpurun_out/prof_bench > gpurun_out/trace_gaps.txt 2>&1; tail -15 gpurun_out/trace_gaps.txt ;;
    pmc_sq)       # SQ counters of the tally kernel, one group per pass (scripts/pmc_sq.sh) -> gpurun_out/sq_summary.txt
      bash scripts/pmc_sq.sh > gpurun_out/sq_summary.txt 2>&1; tail -24 gpurun_out/sq_summary.txt ;;
    pmc_traffic)  # FETCH_SIZE / WRITE_SIZE of the tally kernel, calibrated on the streaming probe (scripts/pmc_traffic.py)
      cd /tmp
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf "$R/gpurun_out/pmc_$c"
        timeout 300 rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/pmc_$c" -o pmc -- python "$R/scripts/prof_tally.py" C3b 3 > "$R/gpurun_out/pmc_$c.log" 2>&1
      done
      cd "$R"
      sb=$(grep -h "^workload" gpurun_out/pmc_FETCH_SIZE.log | tail -1 | sed 's/.*stream_bytes \([0-9]*\).*/\1/')
      python scripts/pmc_traffic.py "$(find gpurun_out/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$sb" gpurun_out/pmc_traffic.json | tail -12
      python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pmc_WRITE_SIZE/**/*counter_collection.csv", recursive=True):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "tally_population" in r["Kernel_Name"]]
    if v: print("tally WRITE_SIZE avg (KiB)", sum(v) / len(v), "launches", len(v))
PY
      ;;
    soak)         # random view builds / changes against the oracle on the device (product build, then poisoned test build)
      timeout 600 python scripts/soak_view.py --chunks 2 100 1000 2>&1 | filter | tee gpurun_out/soak_view_product.txt | tail -3
      RAPID_POISON=0xFF timeout 600 python scripts/soak_view.py --chunks 2 60 2000 --test-build 2>&1 | filter | tee gpurun_out/soak_view_poisoned.txt | tail -3 ;;
    tail_probe)   # tally kernel time against the number of receivers: steady rate + per-launch cost
      timeout 600 python scripts/tail_probe.py 10 2>&1 | filter | tee gpurun_out/tail_probe.txt | tail -12 ;;
    collective)   # the round's collective floor through a 1-rank communicator; the step at 1/2, 1/4, 1/8 of the receivers
      timeout 600 python scripts/collective_latency.py 50 2>&1 | filter | tee gpurun_out/collective_latency.txt | tail -8 ;;
    settle_ab)    # the fast round settled inside the tally launch (opt-in) against the default
      timeout 600 python scripts/step_ab_settle.py 60 2>&1 | filter | tee gpurun_out/step_ab_settle.txt | tail -4 ;;
    pmc_c5)       # FETCH_SIZE / WRITE_SIZE of the C5 tile kernels (generator, tally on boundary and on resolved records): scripts/c5_probe.py under rocprofv3
      cd /tmp
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf "$R/gpurun_out/pmc_c5_$c"
        RAPID_AB_ONLY=default timeout 600 rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/pmc_c5_$c" -o pmc -- python "$R/scripts/c5_probe.py" 1000000 1024 2 > "$R/gpurun_out/pmc_c5_$c.log" 2>&1
      done
      cd "$R"; python - <<'PY' | tee gpurun_out/pmc_traffic_c5.txt
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_c5_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("%-11s %-90s launches %4d  avg %12.0f KiB  max %12.0f KiB" % (c, k, len(v), sum(v) / len(v), max(v)))
for line in open("gpurun_out/pmc_c5_FETCH_SIZE.log"):
    if line.startswith("setup") or "records" in line:
        print(line.rstrip()[:200])
PY
      ;;
    *) echo "unknown section $section" ;;
  esac
done
