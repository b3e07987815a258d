"""Checkpoint save / load / auto-resume with the reference's `.pyth` schema.

Same functions and file layout as `lib/utils/checkpoint.py` (get_path_to_checkpoint :46-54, get_last_checkpoint
:57-70, is_checkpoint_epoch :84-104, save_checkpoint :107-136, load_checkpoint :190-400, load_train_checkpoint
:543-570): `OUTPUT_DIR/checkpoints/checkpoint_epoch_{epoch+1:05d}.pyth` holding
`{epoch, model_state, optimizer_state, cfg: cfg.dump()}`; loading keeps only keys whose shapes match
(strict=False) and nearest-resizes `model.time_embed`.  caffe2 / sub-BN conversions of the reference are out of
scope (ViT has no BN).  `load_pretrained` restates `lib/models/helpers.py:100-243` for LOCAL files
(`TIMESFORMER.PRETRAINED_MODEL`): `model.` prefix strip, classifier drop on size mismatch, pos/time-embed nearest
resize, and cloning of spatial `attn` / `norm1` weights into missing `temporal_attn` / `temporal_norm1` keys.
"""
import os
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import distributed as du


def make_checkpoint_dir(path_to_job):
    d = os.path.join(path_to_job, "checkpoints")
    if du.is_master_proc() and not os.path.exists(d):
        os.makedirs(d, exist_ok=True)
    return d


def get_checkpoint_dir(path_to_job):
    return os.path.join(path_to_job, "checkpoints")


def get_path_to_checkpoint(path_to_job, epoch):
    return os.path.join(get_checkpoint_dir(path_to_job), "checkpoint_epoch_{:05d}.pyth".format(epoch))


def _ls(d):
    return os.listdir(d) if os.path.exists(d) else []


def get_last_checkpoint(path_to_job):
    d = get_checkpoint_dir(path_to_job)
    names = [f for f in _ls(d) if "checkpoint" in f]
    assert len(names), "No checkpoints found in '{}'.".format(d)
    return os.path.join(d, sorted(names)[-1])


def has_checkpoint(path_to_job):
    return any("checkpoint" in f for f in _ls(get_checkpoint_dir(path_to_job)))


def is_checkpoint_epoch(cfg, cur_epoch, multigrid_schedule=None):
    if cur_epoch + 1 == cfg.SOLVER.MAX_EPOCH:
        return True
    return (cur_epoch + 1) % cfg.TRAIN.CHECKPOINT_PERIOD == 0


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def save_checkpoint(path_to_job, model, optimizer, epoch, cfg):
    if not du.is_master_proc(max(1, cfg.NUM_GPUS) * cfg.NUM_SHARDS):
        return None
    os.makedirs(get_checkpoint_dir(path_to_job), exist_ok=True)
    sd = OrderedDict((k, v.detach().cpu()) for k, v in _unwrap(model).state_dict().items())
    checkpoint = {"epoch": epoch, "model_state": sd, "optimizer_state": optimizer.state_dict(), "cfg": cfg.dump()}
    path = get_path_to_checkpoint(path_to_job, epoch + 1)
    with open(path, "wb") as f:
        torch.save(checkpoint, f)
    return path


def load_checkpoint(path_to_checkpoint, model, data_parallel=True, optimizer=None, inflation=False,
                    convert_from_caffe2=False, epoch_reset=False, clear_name_pattern=()):
    assert os.path.exists(path_to_checkpoint), "Checkpoint '{}' not found".format(path_to_checkpoint)
    if convert_from_caffe2 or inflation:
        raise NotImplementedError("caffe2 / 2D-inflation checkpoints belong to the reference's CNN zoo (out of scope)")
    ms = _unwrap(model)
    with open(path_to_checkpoint, "rb") as f:
        checkpoint = torch.load(f, map_location="cpu", weights_only=False)
    state = checkpoint["model_state"]
    for item in clear_name_pattern or ():
        state = OrderedDict((k.replace(item, "") if item in k else k, v) for k, v in state.items())
    model_dict = ms.state_dict()
    k = "model.time_embed"
    if k in state and k in model_dict and state[k].shape != model_dict[k].shape:
        v = state[k][0, :, :].unsqueeze(0).transpose(1, 2)
        state[k] = F.interpolate(v, size=(model_dict[k].size(1)), mode="nearest").transpose(1, 2)
    match = {k: v for k, v in state.items() if k in model_dict and v.size() == model_dict[k].size()}
    not_loaded = [k for k in model_dict if k not in match]
    for k in not_loaded:
        print("Network weights {} not loaded.".format(k))
    ms.load_state_dict(match, strict=False)
    epoch = -1
    if "epoch" in checkpoint and not epoch_reset:
        epoch = checkpoint["epoch"]
        if optimizer:
            optimizer.load_state_dict(checkpoint["optimizer_state"])
    return epoch


def load_train_checkpoint(cfg, model, optimizer):
    if cfg.TRAIN.AUTO_RESUME and has_checkpoint(cfg.OUTPUT_DIR):
        last = get_last_checkpoint(cfg.OUTPUT_DIR)
        print("Load from last checkpoint, {}.".format(last))
        return load_checkpoint(last, model, cfg.NUM_GPUS > 1, optimizer) + 1
    if cfg.TRAIN.CHECKPOINT_FILE_PATH != "":
        print("Load from given checkpoint file.")
        return load_checkpoint(cfg.TRAIN.CHECKPOINT_FILE_PATH, model, cfg.NUM_GPUS > 1, optimizer,
                               inflation=cfg.TRAIN.CHECKPOINT_INFLATE,
                               convert_from_caffe2=cfg.TRAIN.CHECKPOINT_TYPE == "caffe2",
                               epoch_reset=cfg.TRAIN.CHECKPOINT_EPOCH_RESET,
                               clear_name_pattern=cfg.TRAIN.CHECKPOINT_CLEAR_NAME_PATTERN) + 1
    return 0


def load_test_checkpoint(cfg, model):
    if cfg.TEST.CHECKPOINT_FILE_PATH != "":
        load_checkpoint(cfg.TEST.CHECKPOINT_FILE_PATH, model, cfg.NUM_GPUS > 1, None)
    elif has_checkpoint(cfg.OUTPUT_DIR):
        load_checkpoint(get_last_checkpoint(cfg.OUTPUT_DIR), model, cfg.NUM_GPUS > 1)
    elif cfg.TRAIN.CHECKPOINT_FILE_PATH != "":
        load_checkpoint(cfg.TRAIN.CHECKPOINT_FILE_PATH, model, cfg.NUM_GPUS > 1, None)
    else:
        print("Unknown way of loading checkpoint. Using with random initialization, only for debugging.")


def _read_state_dict(path):
    ck = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(ck, dict) and "state_dict" in ck:
        return OrderedDict((k[7:] if k.startswith("module") else k, v) for k, v in ck["state_dict"].items())
    if isinstance(ck, dict) and "model_state" in ck:
        return OrderedDict((k[6:] if k.startswith("model") else k, v) for k, v in ck["model_state"].items())
    if isinstance(ck, dict) and "model" in ck and isinstance(ck["model"], dict):
        return OrderedDict(ck["model"])
    return OrderedDict(ck)


def load_pretrained(model, cfg):
    """Initialise `model` (a VisionTransformer) from TIMESFORMER.PRETRAINED_MODEL (local file).  The reference falls
    back to downloading the ImageNet ViT from a URL (helpers.py:108-115); there is no network here, so an empty
    path is an error instead of a silent random init."""
    path = cfg.TIMESFORMER.PRETRAINED_MODEL
    if not path:
        raise FileNotFoundError("MODEL.PRETRAINED is True but TIMESFORMER.PRETRAINED_MODEL is empty: give a local "
                                "checkpoint (the reference's URL download needs network), or set MODEL.PRETRAINED False")
    state = _read_state_dict(path)
    head_w = state.get("head.weight")
    if head_w is not None and hasattr(model, "head") and head_w.size() != model.head.weight.size():
        print("Removing the last fully connected layer due to dimensions mismatch")
        state.pop("head.weight", None)
        state.pop("head.bias", None)
    n_tok = model.pos_embed.shape[1]
    if "pos_embed" in state and state["pos_embed"].size(1) != n_tok:
        pe = state["pos_embed"]
        cls_pe = pe[0, 0, :].unsqueeze(0).unsqueeze(1)
        other = F.interpolate(pe[0, 1:, :].unsqueeze(0).transpose(1, 2), size=(n_tok - 1), mode="nearest").transpose(1, 2)
        state["pos_embed"] = torch.cat((cls_pe, other), 1)
    nf = model.time_embed.shape[1]
    if "time_embed" in state and state["time_embed"].size(1) != nf:
        state["time_embed"] = F.interpolate(state["time_embed"].transpose(1, 2), size=(nf), mode="nearest").transpose(1, 2)
    new_state = OrderedDict(state)
    for key in state:                       # helpers.py:223-238
        if "blocks" in key and "attn" in key and "temporal_attn" not in key:
            new_state.setdefault(key.replace("attn", "temporal_attn"), state[key])
        if "blocks" in key and "norm1" in key and "temporal_norm1" not in key:
            new_state.setdefault(key.replace("norm1", "temporal_norm1"), state[key])
    missing, unexpected = model.load_state_dict(new_state, strict=False)
    print("\nMissing keys: ", missing, "\nUnexpected_keys: ", unexpected)


def convert_mvit_image_state(state, model_dict):
    """Released image MViTv2 checkpoint -> video encoder state (lib/models/helpers.py:124-142): every key gets the
    `video_encoder.` prefix; pooling / patch-embed conv weights are repeated over the new time axis (not divided by t, as in
    the reference); relative-position tables are linearly interpolated to the model's length."""
    new = OrderedDict()
    for key, v in state.items():
        tgt = "video_encoder." + key
        if "pool_" in key or "patch_embed.proj.weight" in key:
            t = model_dict[tgt].shape[2]
            new[tgt] = v.unsqueeze(2).repeat(1, 1, t, 1, 1)
        elif "rel_pos_" in key:
            n = model_dict[tgt].shape[0]
            r = F.interpolate(v.reshape(1, v.shape[0], -1).permute(0, 2, 1), size=n, mode="linear")
            new[tgt] = r.reshape(-1, n).permute(1, 0).squeeze()
        else:
            new[tgt] = v
    return new


def load_pretrained_mvit(model, cfg):
    """Initialise the MViT wrapper (`procedurevrl_amd.mvit.VisionTransformer`) from a LOCAL copy of the checkpoint the
    reference downloads (lib/models/mvit.py:41, helpers.py:108-142): an image MViTv2 `model_state` is converted, an already
    converted / video checkpoint (keys prefixed `video_encoder.`) is loaded as is; shape-matched, non-strict."""
    path = cfg.TIMESFORMER.PRETRAINED_MODEL
    if not path:
        raise FileNotFoundError("MODEL.PRETRAINED is True but TIMESFORMER.PRETRAINED_MODEL is empty: give a local copy of "
                                "MViTv2_S_in1k.pyth (the reference's URL download needs network), or set MODEL.PRETRAINED False")
    ck = torch.load(path, map_location="cpu", weights_only=False)
    state = ck["model_state"] if isinstance(ck, dict) and "model_state" in ck else ck
    model_dict = model.state_dict()
    if not any(k.startswith("video_encoder.") for k in state):
        state = convert_mvit_image_state(state, model_dict)
    hw = state.get("head.weight")
    if hw is not None and hw.size() != model.head.weight.size():
        state.pop("head.weight", None)
        state.pop("head.bias", None)
    missing, unexpected = model.load_state_dict(state, strict=False)
    print("\nMissing keys: ", missing, "\nUnexpected_keys: ", unexpected)
