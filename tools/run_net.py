"""python tools/run_net.py --cfg configs/HowTo100M/procedurevrl_adamw.yaml [--shard_id I --num_shards N --init_method URL] KEY VAL ...

Same command line as the reference (`tools/run_net.py:15-39`, `lib/utils/parser.py:12-93`): yaml config, trailing
KEY VAL overrides merged last, one process per GPU (`lib/utils/misc.py:272-300` launch_job).  With
`SYNTHETIC.ENABLE True` the data loader is the synthetic stand-in with the reference's batch contract
(real HowTo100M decoding is out of scope)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def parse_args():
    p = argparse.ArgumentParser(description="Provide ProcedureVRL (MI355X) training and testing pipeline.")
    p.add_argument("--shard_id", default=0, type=int)
    p.add_argument("--num_shards", default=1, type=int)
    p.add_argument("--init_method", default="tcp://127.0.0.1:9999", type=str)
    p.add_argument("--cfg", dest="cfg_file", default="", type=str)
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p.parse_args()


def load_config(args):
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd import checkpoint as cu
    cfg = get_cfg()
    if args.cfg_file:
        cfg.merge_from_file(args.cfg_file)
    if args.opts:
        cfg.merge_from_list(args.opts)
    cfg.NUM_SHARDS = args.num_shards
    cfg.SHARD_ID = args.shard_id
    cu.make_checkpoint_dir(cfg.OUTPUT_DIR)
    return cfg


def _run(local_rank, num_proc, func, init_method, shard_id, num_shards, backend, cfg):
    import torch
    from procedurevrl_amd import distributed as du
    single = bool(os.environ.get("PVRL_SINGLE_DEVICE"))     # functional test of the N-process path on one GPU (gloo)
    du.init_process_group(local_rank, num_proc, shard_id, num_shards, init_method, backend)
    torch.cuda.set_device(0 if single else local_rank)
    func(cfg)


def launch_job(cfg, init_method, func):
    import torch
    if cfg.NUM_GPUS > 1:
        torch.multiprocessing.spawn(_run, nprocs=cfg.NUM_GPUS, daemon=False,
                                    args=(cfg.NUM_GPUS, func, init_method, cfg.SHARD_ID, cfg.NUM_SHARDS, cfg.DIST_BACKEND, cfg))
    else:
        func(cfg)


def main():
    args = parse_args()
    cfg = load_config(args)
    from procedurevrl_amd.train_net import train
    from procedurevrl_amd.test_net import test
    if cfg.SYNTHETIC.ENABLE:        # offline stand-ins for the embedding files the yaml names (TRAIN.LABEL_EMB, DEV.TEST_LANG_EMB)
        from procedurevrl_amd.datasets import synthetic_label_emb
        if cfg.TRAIN.ENABLE and isinstance(cfg.TRAIN.LABEL_EMB, str) and not os.path.exists(cfg.TRAIN.LABEL_EMB):
            cfg.TRAIN.LABEL_EMB = synthetic_label_emb(cfg.MODEL.NUM_CLASSES)
        if isinstance(cfg.DEV.TEST_LANG_EMB, str) and not os.path.exists(cfg.DEV.TEST_LANG_EMB):
            cfg.DEV.TEST_LANG_EMB = synthetic_label_emb(cfg.MODEL.NUM_CLASSES, seed=1)
    # tools/run_net.py:25-31: train, then multi-clip testing
    if cfg.TRAIN.ENABLE:
        launch_job(cfg=cfg, init_method=args.init_method, func=train)
    if cfg.TEST.ENABLE:
        launch_job(cfg=cfg, init_method=args.init_method, func=test)


if __name__ == "__main__":
    main()
