"""The launcher surface for N > 1 and for evaluation, on ONE GPU.

* `tools/run_net.py`-style `launch_job` with NUM_GPUS 2 (two processes, gloo, both on cuda:0): `train(cfg)` builds its
  loader through `construct_loader`, whose DistributedSampler must hand the two ranks disjoint videos (reference:
  lib/datasets/utils.py:358-370, loader.py:85-92, train_net.py:503), with gradient accumulation over two
  micro-iterations (train_net.py:176-192) so the reducer's no-sync path runs under the real train loop; afterwards both
  ranks must hold identical weights.
* `test(cfg)` (reference signature, tools/test_net.py:161-221) builds its own multi-view loader; the video-level
  predictions and top-k accuracies must equal a recomputation from per-clip model outputs.
* `python bench.py --gpus 2` must become two ranks by itself and report n_gpus 2.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _base_cfg(tmp, extra=()):
    from procedurevrl_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_list(["MODEL.MODEL_NAME", "vit_base_patch16_224_develop", "MODEL.PRETRAINED", "False",
                         "MODEL.NUM_CLASSES", "64", "MODEL.LOSS_FUNC", "kldiv", "MODEL.DROP_PATH", "0.1",
                         "TIMESFORMER.DEPTH", "2", "DATA.TRAIN_CROP_SIZE", "32", "DATA.TEST_CROP_SIZE", "32",
                         "DEV.MATCH_LANG_EMB", "True", "SOLVER.BASE_LR", "1e-4", "SOLVER.OPTIMIZING_METHOD", "adamw",
                         "SOLVER.LR_POLICY", "steps_with_relative_lrs", "SOLVER.STEPS", "[0, 1]", "SOLVER.LRS", "[1, 0.1]",
                         "LOG_PERIOD", "2", "SYNTHETIC.ENABLE", "True", "SYNTHETIC.TEXT_LAYERS", "2",
                         "OUTPUT_DIR", str(tmp)] + list(extra))
    return cfg


def _train_job(cfg):
    """what launch_job runs in every rank: the real train(cfg), with the dataset's __getitem__ recording what it served"""
    import torch.distributed as dist
    from procedurevrl_amd import datasets
    from procedurevrl_amd.train_net import train
    seen = []
    orig = datasets.SyntheticHowTo100M.__getitem__

    def getitem(self, index):
        seen.append(int(index))
        return orig(self, index)

    datasets.SyntheticHowTo100M.__getitem__ = getitem
    model, _ = train(cfg)
    rank = dist.get_rank()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if "text_model" not in k}
    torch.save(dict(seen=seen, state=sd), os.path.join(cfg.OUTPUT_DIR, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_launch_job_shards_videos_and_accumulates(tmp_path, monkeypatch):
    monkeypatch.setenv("PVRL_SINGLE_DEVICE", "1")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import run_net
    from procedurevrl_amd.datasets import synthetic_label_emb
    cfg = _base_cfg(tmp_path, ["MODEL.TEXT_MODEL", "clip_vit_b_16", "DEV.ORDER_PRETRAIN_ENABLED", "True", "TRAIN.TEXT", "synthetic",
                               "NUM_GPUS", "2", "DIST_BACKEND", "gloo", "TRAIN.BATCH_SIZE", "2", "GLOBAL_BATCH_SIZE", "4",
                               "SOLVER.MAX_EPOCH", "1", "SYNTHETIC.NUM_VIDEOS", "8", "TRAIN.CHECKPOINT_PERIOD", "100",
                               "TEST.ENABLE", "False"])
    cfg.TRAIN.LABEL_EMB = synthetic_label_emb(64)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    run_net.launch_job(cfg=cfg, init_method=f"tcp://127.0.0.1:{port}", func=_train_job)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert len(r0["seen"]) == len(r1["seen"]) == 4                       # 8 videos / 2 ranks, 1 video per rank and iteration
    assert not set(r0["seen"]) & set(r1["seen"]) and sorted(r0["seen"] + r1["seen"]) == list(range(8))
    for k in r0["state"]:                                                  # same all-reduced update on both ranks
        assert torch.equal(r0["state"][k], r1["state"][k]), k


@pytest.mark.gpu
def test_multi_view_test_entry_matches_recomputation(tmp_path):
    from procedurevrl_amd import checkpoint as cu
    from procedurevrl_amd.build import build_model
    from procedurevrl_amd.datasets import SyntheticTestClips, synthetic_label_emb
    from procedurevrl_amd.optimizer import construct_optimizer
    from procedurevrl_amd.test_net import test
    from procedurevrl_amd.train_net import topks_correct
    cfg = _base_cfg(tmp_path, ["NUM_GPUS", "1", "TRAIN.ENABLE", "False", "TEST.ENABLE", "True", "TEST.BATCH_SIZE", "5",
                               "TEST.NUM_ENSEMBLE_VIEWS", "2", "TEST.NUM_SPATIAL_CROPS", "3", "MODEL.NUM_CLASSES", "16",
                               "SYNTHETIC.NUM_VIDEOS", "7"])
    cfg.DEV.TEST_LANG_EMB = synthetic_label_emb(16, seed=3)               # zero-shot head: classes = the language embeddings
    torch.manual_seed(3)
    model = build_model(cfg)
    with torch.no_grad():
        for blk in model.model.blocks:
            torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
    cu.save_checkpoint(str(tmp_path), model, construct_optimizer(model, cfg), 0, cfg)
    meter = test(cfg)                                                     # own model, own loader, the checkpoint above
    # recomputation: every clip through the saved model, sum over the 6 views of a video on the CPU
    ds = SyntheticTestClips(cfg, 7)
    model.eval()
    preds = torch.zeros(7, 16)
    labels = torch.zeros(7, dtype=torch.long)
    with torch.no_grad():
        for i in range(len(ds)):
            x, lab, idx, _ = ds[i]
            p = model(x.unsqueeze(0).cuda()).float().cpu()[0]
            assert abs(float(p.sum()) - 1.0) < 1e-4                       # eval forward = softmax probabilities (vit.py:355-356)
            preds[int(idx) // 6] += p
            labels[int(idx) // 6] = lab
    assert torch.equal(meter.clip_count, torch.full((7,), 6))
    assert torch.equal(meter.video_labels, labels)
    assert float((meter.video_preds - preds).abs().max()) < 2e-3, float((meter.video_preds - preds).abs().max())
    c1, c5 = topks_correct(preds, labels, (1, 5))
    assert meter.stats["top1_acc"] == "{:.2f}".format(float(c1) / 7 * 100.0)
    assert meter.stats["top5_acc"] == "{:.2f}".format(float(c5) / 7 * 100.0)


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    env = dict(os.environ, PVRL_DIST_BACKEND="gloo", PVRL_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--steps", "2",
                        "--warmup", "1", "--classes", "512", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["comm"]["ranks"] == 2
    assert out["value"] > 0 and out["scaling"] == "weak"


@pytest.mark.gpu
def test_bench_eight_ranks_the_drivers_command():
    """The driver's 8-GPU launch -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 ...` -- functionally on ONE GPU: eight processes share the device (PVRL_SINGLE_DEVICE) and reduce
    over gloo.  Covers the 8-rank rendezvous, per-block gradient hooks between staged HIP-graph replays at world 8, the all-gather of the
    InfoNCE embeddings, the max-over-ranks clock and the single JSON line; the reservation of CUs for RCCL is a no-op of the gloo
    backend apart from its bookkeeping fields."""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, PVRL_DIST_BACKEND="gloo", PVRL_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PVRL_COMM_CUS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PVRL_COMPUTE_CUS", "NCCL_MAX_NCHANNELS"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--batch", "2", "--steps", "2",
                        "--warmup", "1", "--classes", "512", "--no-cpu-baseline", "--no-kernel-timing"],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 16 and out["comm"]["ranks"] == 8
    assert out["config"]["parallelism"] == "dp8" and out["scaling"] == "weak" and out["value"] > 0
    assert out["comm"]["compute_cus_per_xcd"] == "31" and out["comm"]["nccl_max_nchannels"] == "8"
    # what the first real 8-GPU run needs to be diagnosable in one shot: the main stream's wait for the communication
    # stream per step (rank 0 and the slowest rank), the per-chunk collective times with their sizes, the hardware-queue setting, and
    # the 2-clip parity probe on rank 0
    c = out["comm"]
    assert c["steps"] == 2 and c["exposed_ms_per_step"] >= 0.0 and c["exposed_ms_per_step_max_over_ranks"] >= c["exposed_ms_per_step"] - 1e-9
    assert len(c["allreduce_ms_per_chunk"]) == len(c["chunk_mb"]) >= 2 and all(t >= 0.0 for t in c["allreduce_ms_per_chunk"])
    assert abs(sum(c["chunk_mb"]) - c["grad_allreduce_mb"]) < 1.0 and "hw_queues" in c and c["hook_group"] >= 1
    assert out["parity"]["meets_1e-3_on_logits_and_loss"] is True, out["parity"]
