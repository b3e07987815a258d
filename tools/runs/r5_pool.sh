#!/bin/bash
# round 5: the LDS-DMA tile pool kernels -- parity check, then per-block times with the tile kernels on / off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_pool.txt; : > $O
timeout 600 python -c "
import sys; sys.path.insert(0,'tests')
import mvit_checks as mc
for l,e,t in mc.check_mvit_pool():
    print(('ok  ' if e<=t else 'FAIL'), l, '%.3e'%e, t)
" 2>&1 | grep -v Warn | tail -60 >> $O
for t in 1 0; do echo "== PVRL_POOL_TILE=$t" >> $O; PVRL_POOL_TILE=$t timeout 600 python tools/probe/mvit_pool_times.py 2>&1 | tail -17 >> $O; done
cat $O
