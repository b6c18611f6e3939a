cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06c; mkdir -p $O
for cfg in "27 512 1" "27 512 3" "7 256 1"; do set -- $cfg
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$1_$2_f$3 -o t -- python $GRAFT_REPO_ROOT/scripts/per_rank_loopback.py --stencil $1 --grid $2 --fused $3 --its 30 > $O/run_$1_$2_f$3.txt 2>&1)
  f=$(find $O/tr_$1_$2_f$3 -name "*kernel_trace.csv" | head -1)
  python scripts/trace_timeline.py $f --last 45 --stats > $O/timeline_$1_$2_f$3.txt 2>&1
  rm -rf $O/tr_$1_$2_f$3
  tail -3 $O/run_$1_$2_f$3.txt
done
