"""Checkpoint interop against the REFERENCE's own save / load code (lib/utils/checkpoint.py), run in the build container
only (skipped where /root/reference is absent, e.g. on the GPU box): a `.pyth` written by the reference loads into this
implementation and vice versa, with identical tensors.  No GPU needed: state_dict plumbing only."""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


class _PathManager:
    exists = staticmethod(os.path.exists)
    ls = staticmethod(os.listdir)
    isfile = staticmethod(os.path.isfile)

    @staticmethod
    def mkdirs(p):
        os.makedirs(p, exist_ok=True)

    @staticmethod
    def open(p, mode="r"):
        return open(p, mode)


@pytest.fixture(scope="module")
def ref():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as mg
    defaults, vit, tfm, dist_mod, losses = mg.import_reference()
    fio = types.ModuleType("fvcore.common.file_io")
    fio.PathManager = _PathManager
    sys.modules["fvcore.common.file_io"] = fio
    import importlib
    ck = importlib.import_module("lib.utils.checkpoint")
    return types.SimpleNamespace(mg=mg, defaults=defaults, vit=vit, tfm=tfm, ck=ck)


def _my_model(cfg_ref, label_path):
    from procedurevrl_amd.build import MODEL_REGISTRY
    from procedurevrl_amd import vit  # noqa: F401
    from procedurevrl_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_list(["MODEL.MODEL_NAME", "vit_base_patch16_224_develop", "MODEL.PRETRAINED", "False", "MODEL.NUM_CLASSES",
                         str(cfg_ref.MODEL.NUM_CLASSES), "MODEL.TEXT_MODEL", "clip_vit_b_16", "TIMESFORMER.DEPTH",
                         str(cfg_ref.TIMESFORMER.DEPTH), "DATA.TRAIN_CROP_SIZE", str(cfg_ref.DATA.TRAIN_CROP_SIZE),
                         "DEV.MATCH_LANG_EMB", "True", "DEV.ORDER_PRETRAIN_ENABLED", "True", "NUM_GPUS", "0",
                         "SYNTHETIC.TEXT_LAYERS", "1"])
    cfg.TRAIN.LABEL_EMB = label_path
    return cfg, MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)


def test_reference_checkpoint_loads_here_and_ours_loads_there(ref, tmp_path):
    from procedurevrl_amd import checkpoint as cu
    cfg_r, model_r, _ = ref.mg.build_ref_model(ref.defaults, ref.vit, ref.tfm, depth=1, crop=32, K=16, text_layers=1,
                                               tmpdir=str(tmp_path))
    cfg_r.TRAIN.CHECKPOINT_PERIOD = 1
    opt_r = torch.optim.SGD([p for p in model_r.parameters() if p.requires_grad], lr=0.1, momentum=0.9)
    path = ref.ck.save_checkpoint(str(tmp_path / "job_ref"), model_r, opt_r, 4, cfg_r)
    assert path.endswith("checkpoints/checkpoint_epoch_00005.pyth")
    cfg_m, model_m = _my_model(cfg_r, cfg_r.TRAIN.LABEL_EMB)
    # same key set on both sides (the text tower is part of the reference's state_dict as well)
    assert set(model_m.state_dict().keys()) == set(model_r.state_dict().keys())
    epoch = cu.load_checkpoint(path, model_m, data_parallel=False, optimizer=None)
    assert epoch == 4
    sr, sm = model_r.state_dict(), model_m.state_dict()
    for k in sr:
        assert torch.equal(sr[k], sm[k]), k
    # ... and back: a checkpoint written here, read by the reference's load_checkpoint
    from procedurevrl_amd.optimizer import construct_optimizer
    with torch.no_grad():
        for p in model_m.parameters():
            p.add_(0.01)
    opt_m = construct_optimizer(model_m, cfg_m)
    path2 = cu.save_checkpoint(str(tmp_path / "job_mine"), model_m, opt_m, 7, cfg_m)
    assert os.path.basename(path2) == "checkpoint_epoch_00008.pyth"
    ck = torch.load(path2, map_location="cpu", weights_only=False)
    assert set(ck.keys()) == {"epoch", "model_state", "optimizer_state", "cfg"}
    epoch_r = ref.ck.load_checkpoint(path2, model_r, data_parallel=False, optimizer=None)
    assert epoch_r == 7
    sr, sm = model_r.state_dict(), model_m.state_dict()
    for k in sm:
        assert torch.equal(sr[k], sm[k]), k
