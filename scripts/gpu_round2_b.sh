cd "$GRAFT_REPO_ROOT"; O=gpurun_out
for a in "7pt 12 dep 18" "7pt 12 strand 18" "7pt 12 strand 17" "7pt 16 strand 18" "7pt 12 strand 28"; do echo "== $a"; timeout 120 python scripts/sor_debug.py $a 2>&1 | grep -v amdgpu.ids | tail -8; done > $O/r2b_sordbg.log 2>&1
cat $O/r2b_sordbg.log
