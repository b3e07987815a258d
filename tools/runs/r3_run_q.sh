#!/bin/bash
python -m pytest tests/test_mvit_gpu.py -m gpu -q > gpurun_out/r3_pytest_q.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_q.log
python - > gpurun_out/r3_q_im2col.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from procedurevrl_amd import ops_mvit as om
x = torch.randn(32, 3, 16, 224, 224, device='cuda')
for _ in range(3): om.im2col3d(x, (3,7,7), (2,4,4), (1,3,3), 512)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): om.im2col3d(x, (3,7,7), (2,4,4), (1,3,3), 512)
e1.record(); torch.cuda.synchronize()
print("im2col3d 32 clips: %.1f us" % (e0.elapsed_time(e1)/10*1e3))
PY
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
python bench.py $B --arch mvit > gpurun_out/r3_q_mvit1.json 2>/dev/null
python bench.py $B --arch mvit > gpurun_out/r3_q_mvit2.json 2>/dev/null
tail -3 gpurun_out/r3_pytest_q.log; cat gpurun_out/r3_q_im2col.txt; grep -h -o '"value": [0-9.]*' gpurun_out/r3_q_*.json
