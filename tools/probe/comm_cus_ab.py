"""Probe: what does leaving r CUs per XCD to RCCL cost on ONE GPU, and what does it avoid?  

A training step of bench.py's configuration (32 clips of 8x224^2, HIP-graph replays) is timed
  * alone, with the persistent grids sized for 32 - r CUs per XCD (PVRL_COMPUTE_CUS, read once per process: one child process per r);
  * under a "communication" kernel that HOLDS h CUs per XCD on another stream for the whole backward (tools/probe/cu_hog.hip: one
    512-thread / 128 KB-LDS workgroup per CU, what an RCCL channel kernel amounts to for this library's one-workgroup-per-CU kernels).
usage: python tools/probe/comm_cus_ab.py            (driver: spawns the children, prints a table)
       python tools/probe/comm_cus_ab.py --child h  (one measurement in this process; PVRL_COMPUTE_CUS from the environment)"""
import ctypes
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)


def build_hog():
    lib = os.path.join(HERE, "libcu_hog.so")
    src = os.path.join(HERE, "cu_hog.hip")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", lib], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
    return lib


def child(hog_per_xcd, steps=12):
    import torch
    from procedurevrl_amd.build import build_model
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd.datasets import synthetic_label_emb
    from procedurevrl_amd.functional import kl_topk_loss
    from procedurevrl_amd.optimizer import construct_optimizer, set_lr
    dll = ctypes.CDLL(build_hog())
    dll.pvrl_probe_cu_hog.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    cfg = get_cfg()
    cfg.MODEL.MODEL_NAME, cfg.MODEL.ARCH, cfg.MODEL.NUM_CLASSES, cfg.MODEL.PRETRAINED = "vit_base_patch16_224_develop", "vit", 9871, False
    cfg.MODEL.LOSS_FUNC, cfg.MODEL.DROP_PATH, cfg.DEV.MATCH_LANG_EMB, cfg.NUM_GPUS = "kldiv", 0.1, True, 1
    cfg.SOLVER.OPTIMIZING_METHOD = "adamw"
    cfg.TRAIN.LABEL_EMB = synthetic_label_emb(9871, 512, seed=0)
    torch.manual_seed(0)
    model = build_model(cfg, gpu_id=0).train()
    vt = model.model
    with torch.no_grad():
        for blk in vt.blocks:
            torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
    opt = construct_optimizer(model, cfg)
    set_lr(opt, 5e-5)
    dev = torch.device("cuda", 0)
    x = torch.randn(32, 3, 8, 224, 224, device=dev)
    teacher = torch.randn(32, 9871, device=dev) * 4
    hog = torch.cuda.Stream(device=dev, priority=-1 if os.environ.get("HOG_PRIO") else 0)
    sink = torch.zeros(1, dtype=torch.int32, device=dev)

    def step(hold_us):
        opt.zero_grad(set_to_none=True)
        pred = model(x)
        loss = kl_topk_loss(pred, teacher, 5)
        if hog_per_xcd > 0 and hold_us > 0:       # the "collective" starts with the backward and holds its CUs for its length
            hog.wait_stream(torch.cuda.current_stream())
            dll.pvrl_probe_cu_hog(8 * hog_per_xcd, float(hold_us), ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(hog.cuda_stream))
        loss.backward()
        opt.step()
        torch.cuda.current_stream().wait_stream(hog)

    for _ in range(6):
        step(0)
    torch.cuda.synchronize()

    def timed(hold_us):
        for _ in range(2):
            step(hold_us)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(hold_us)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    a, h = [], []
    for _ in range(3):          # interleaved: box drift shows up in both columns alike
        a.append(timed(0))
        h.append(timed(0.68 * a[-1] * 1e6))      # the backward + optimiser share of a step
    alone, held = sorted(a)[1], sorted(h)[1]
    print("RESULT " + json.dumps({"compute_cus": os.environ.get("PVRL_COMPUTE_CUS", "32"), "hog_per_xcd": hog_per_xcd,
                                  "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "hog_prio": os.environ.get("HOG_PRIO"),
                                  "ms_alone": round(1e3 * alone, 3), "ms_under_hog": round(1e3 * held, 3),
                                  "all_alone": [round(1e3 * v, 1) for v in a], "all_hog": [round(1e3 * v, 1) for v in h]}), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]))
    build_hog()
    rows = []
    quick = os.environ.get("COMM_CUS_CASES")
    cases = [tuple(int(v) for v in c.split(":")) for c in quick.split(",")] if quick else ((0, 0), (0, 1), (1, 1), (0, 2), (2, 2), (0, 4), (4, 4), (1, 0), (2, 0), (4, 0))
    for r, h in cases:
        env = dict(os.environ)
        if r:
            env["PVRL_COMPUTE_CUS"] = str(32 - r)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(h)], capture_output=True, text=True, env=env, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        rows.append(json.loads(line[-1][7:]) if line else {"compute_cus": 32 - r, "hog_per_xcd": h, "error": (out.stderr or "")[-300:]})
        print(rows[-1], flush=True)
    print("reserved r per XCD | CUs per XCD held by the 'collective' | ms / step alone | ms / step with the CUs held through the backward")
    for x in rows:
        print(f"{32 - int(x['compute_cus']):>3} | {x['hog_per_xcd']:>3} | {x.get('ms_alone')} | {x.get('ms_under_hog')}")


if __name__ == "__main__":
    main()
