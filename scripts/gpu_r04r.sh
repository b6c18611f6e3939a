#!/bin/bash
# round-4 run R: the CG prologue on the 27-entry kernels (HIPX_MARCH_CG27): parity of the 27-pt 512^3 leg against its golden, it/s with and without
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
leg() { python bench.py --no-plugin --no-cpu-baseline --no-general --no-traffic 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); o=d['other_configs']['cg_jacobi_27pt_512_strong']
print('27pt 512: %.1f it/s  %.4f ms/it  product %.4f ms  parity %s %.2e  | headline %.1f' % (o['iterations_per_s'], o['ms_per_step'], o['spmv_ms'], o['parity']['pass'], o['parity']['max_rel_diff'], d['value']))
c3=d['other_configs']['config3_solver_gmres30_sor_27pt_256']; print('   config3 %.1f it/s' % c3['iterations_per_s'])"; }
q() { python bench.py --quick "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%8.1f it/s  %.4f ms/it  product %.4f ms  %s' % (d['value'] or -1, d['ms_per_step'], r['avg_launch_ms'], r['kernel'][:24]))"; }
echo "27-pt prologue ON:";  HIPX_MARCH_CG27=1 leg
echo "27-pt prologue OFF:"; leg
for rep in 1 2; do
echo "27pt 256 quick, prologue ON:";  HIPX_MARCH_CG27=1 q --stencil 27 --grid 256 --steps 200
echo "27pt 256 quick, prologue OFF:"; q --stencil 27 --grid 256 --steps 200
done
