timeout 900 python -m pytest tests/test_mvit_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for g in 1 0; do
PVRL_HIP_GRAPHS=$g python - <<'PY' 2>&1 | grep -v "^$\|UserWarning\|Consider\|not loading" | tail -5 | cut -c1-200
import sys, os, torch
sys.argv = ["bench_full_step.py", "--arch", "mvit", "--steps", "8", "--warmup", "4"]
import procedurevrl_amd.vit as vit
orig = vit.pretrain_loss
n = [0]
def pl(pred, teacher, mse, cfg):
    r = orig(pred, teacher, mse, cfg)
    n[0] += 1
    if n[0] >= 10:
        print("graphs", os.environ["PVRL_HIP_GRAPHS"], "step", n[0], "loss", float(r[0]))
    return r
vit.pretrain_loss = pl
import runpy
runpy.run_path("tools/bench_full_step.py", run_name="__main__")
PY
done
