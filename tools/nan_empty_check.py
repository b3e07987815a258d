"""Debug aid: run training steps with every torch.empty / empty_like floating-point buffer pre-filled with NaN, so that a
kernel that reads memory nobody wrote shows up as a NaN loss / gradient deterministically (the caching allocator
otherwise hands back stale but finite data).  usage: python tools/nan_empty_check.py [vit|mvit] [full]"""
import os
import sys

os.environ["PVRL_HIP_GRAPHS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

_empty, _empty_like = torch.empty, torch.empty_like


def empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t


def empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t


torch.empty, torch.empty_like = empty, empty_like
arch = sys.argv[1] if len(sys.argv) > 1 else "vit"
if len(sys.argv) > 2:
    sys.argv = ["bench_full_step.py", "--arch", arch, "--steps", "2", "--warmup", "1", "--videos", "1"]
    import runpy
    runpy.run_path(os.path.join(os.path.dirname(__file__), "bench_full_step.py"), run_name="__main__")
else:
    sys.argv = ["bench.py", "--arch", arch, "--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline", "--no-kernel-timing", "--no-graphs"]
    import bench
    bench.main()
