cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
HIPX_TMPL_TRACE=1 timeout 300 python bench.py --quick 2> gpurun_out/r03ac_trace.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d.get('roofline'))[:600])"
grep "tmpl trace" gpurun_out/r03ac_trace.err | grep -v "wg 8 \|wg 1032" | awk '/passes, start/ {print $0}' | sort -k10 -n | tail -3
grep "tmpl trace\] XCD" gpurun_out/r03ac_trace.err | head -3
python bench.py --help | grep -i "spmv-only" 
