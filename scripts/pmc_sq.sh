#!/bin/bash
# PMC passes over the tally kernel (one counter group per pass; no tracing domains combined with --pmc).
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$R/gpurun_out/pmc_sq$i" -o pmc -- python "$R/scripts/prof_tally.py" C3b 2 > "$R/gpurun_out/pmc_sq$i.log" 2>&1
done
cd "$R"
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_sq*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tally_population' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, sum(v)/len(v), len(v))
PY
grep -h "^workload" gpurun_out/pmc_sq1.log | tail -1
