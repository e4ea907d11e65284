#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -rf gpurun_out/pmc_so*
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$R/gpurun_out/pmc_so$i" -o pmc -- python "$R/scripts/prof_stream_only.py" > "$R/gpurun_out/pmc_so$i.log" 2>&1
done
cd "$R"
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_so*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tally_population' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, sum(v)/len(v), len(v))
PY
tail -1 gpurun_out/pmc_so1.log
