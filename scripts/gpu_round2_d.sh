#!/bin/bash
# Round 2, GPU call D: SOR statistics, template-SpMV counters, new tests (IPC halo, COO KATs, scale parity).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== sor stats"; HIPX_SOR_DEBUG=1 timeout 300 python scripts/config3_slab_proxy.py 2>&1 | grep -v amdgpu.ids | grep "hipx sor\|SOR" | head -30 | cut -c1-400 | tee "$O/r2d_sorstats.log"
echo "== new tests"; timeout 1500 python -m pytest tests/test_gpu_halo.py tests/test_gpu_plugin_mpi.py tests/test_gpu_plugin_kats.py tests/test_gpu_plugin.py tests/test_gpu_scale_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ksp.py -q --timeout=900 -p no:cacheprovider -rf > "$O/r2d_pytest.log" 2>&1; tail -30 "$O/r2d_pytest.log" | cut -c1-300
echo "== counters"
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE TCC_BUBBLE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/r2d_ctr/p$i" -o pmc -- python "$R/bench.py" --spmv-only 6 --variant 0 > "$O/r2d_ctr_p$i.log" 2>&1
  find "$O/r2d_ctr/p$i" -name "*kernel_trace.csv" -delete
done
cd "$R"
python - <<'PY' | tee gpurun_out/r2d_counters.txt
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/r2d_ctr/p*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: [0.0,0])
    for r in csv.DictReader(open(f)):
        kn=r['Kernel_Name']
        key='tmpl' if 'spmv_tmpl' in kn else 'axpy' if 'ew2_kernel' in kn else None
        if key:
            k=(key, r['Counter_Name']); acc[k][0]+=float(r['Counter_Value']); acc[k][1]+=1
    nd = {}
    for k,v in sorted(acc.items()): print(f.split('/')[2], k[0], k[1], '%.5g'%(v[0]/max(v[1],1)), 'samples', v[1])
PY
