export PVRL_DIST_BACKEND=gloo PVRL_SINGLE_DEVICE=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --batch 4 --no-cpu-baseline 2>&1 | grep -v "^W\|^\[W\|^$" | tail -3 | cut -c1-400
unset PVRL_DIST_BACKEND PVRL_SINGLE_DEVICE
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python tools/bench_full_step.py 2>&1 | tail -2 | cut -c1-300
