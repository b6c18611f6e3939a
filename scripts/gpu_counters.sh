#!/bin/bash
# SQ / TA / TCP counters of the SpMV kernels (separate --pmc passes, --kernel-trace only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/counters
cd /tmp
rocprofv3 -L > "$R/gpurun_out/counters/list.txt" 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_LDS_BANK_CONFLICT" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/counters/p$i" -o pmc -- python "$R/scripts/spmv_variants.py" 256 7 25,23 > "$R/gpurun_out/counters/p$i.log" 2>&1
  rm -f "$R/gpurun_out/counters/p$i/pmc_kernel_trace.csv"
done
cd "$R"
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/counters/p*/pmc_counter_collection.csv')):
    acc=collections.defaultdict(lambda: [0.0,0])
    for r in csv.DictReader(open(f)):
        if 'spmv_' in r['Kernel_Name']:
            k=(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])
            acc[k][0]+=float(r['Counter_Value']); acc[k][1]+=1
    for k,v in sorted(acc.items()): print(f.split('/')[2], k[0], k[1], '%.4g'%(v[0]/max(v[1],1)), v[1])
PY
grep -c . gpurun_out/counters/list.txt; tail -3 gpurun_out/counters/p1.log | cut -c1-200
