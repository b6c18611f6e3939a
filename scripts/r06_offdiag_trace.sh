cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06al; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python $GRAFT_REPO_ROOT/scripts/per_rank_loopback.py --stencil 27 --grid 512 --fused 1 --its 30 > $O/run.txt 2>&1)
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python scripts/trace_timeline.py $f --last 20 --stats > $O/timeline_27_512_f1.txt 2>&1
rm -rf $O/tr
head -8 $O/timeline_27_512_f1.txt | cut -c1-120; tail -3 $O/run.txt
bash scripts/gpu_run.sh r06al "tests:tests/test_gpu_mpi_cgfuse.py"
