#!/bin/bash
# PMC pass: instruction-cache / scalar-cache / branch counters of the tally kernel
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -rf gpurun_out/pmc_f*
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN" \
           "SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$R/gpurun_out/pmc_f$i" -o pmc -- python "$R/scripts/prof_tally.py" C3b 2 > "$R/gpurun_out/pmc_f$i.log" 2>&1 || tail -3 "$R/gpurun_out/pmc_f$i.log"
done
cd "$R"
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_f*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tally_population' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, sum(v)/len(v), len(v))
PY
