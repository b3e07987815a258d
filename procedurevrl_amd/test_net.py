"""Multi-view evaluation (reference: tools/test_net.py `perform_test` :32-158, `test` :161-221; ensemble meter
lib/utils/meters.py `TestMeter` :21-203).  The model's eval forward returns softmax probabilities per clip
(vit.py:355-356); clips of one video are sum- (or max-) ensembled and top-k accuracy is computed on the video level.
The per-clip loop of the reference's `update_stats` (one Python iteration + tensor write per clip) is replaced by one
`index_add_` / `scatter_reduce_` per batch; `du.all_gather` gathers (preds, labels, video_idx) across ranks as in
test_net.py:113."""
import torch

from . import checkpoint as cu
from . import distributed as du
from .build import build_model
from .train_net import log_json_stats, topks_correct


class TestMeter:
    def __init__(self, num_videos, num_clips, num_cls, overall_iters=0, multi_label=False, ensemble_method="sum"):
        assert not multi_label, "multi-label (mAP) evaluation belongs to the reference's Charades/AVA paths (out of scope)"
        assert ensemble_method in ("sum", "max"), "Ensemble Method {} is not supported".format(ensemble_method)
        self.num_clips = num_clips
        self.ensemble_method = ensemble_method
        self.video_preds = torch.zeros((num_videos, num_cls))
        self.video_labels = torch.zeros((num_videos)).long()
        self.clip_count = torch.zeros((num_videos)).long()
        self.stats = {}

    def reset(self):
        self.clip_count.zero_()
        self.video_preds.zero_()
        self.video_labels.zero_()

    def update_stats(self, preds, labels, clip_ids):
        preds, labels, clip_ids = preds.detach().float().cpu(), labels.detach().cpu().long(), clip_ids.detach().cpu().long()
        vid = clip_ids // self.num_clips
        self.video_labels[vid] = labels
        if self.ensemble_method == "sum":
            self.video_preds.index_add_(0, vid, preds)
        else:
            self.video_preds.scatter_reduce_(0, vid[:, None].expand_as(preds), preds, reduce="amax", include_self=True)
        self.clip_count.index_add_(0, vid, torch.ones_like(vid))

    def finalize_metrics(self, ks=(1, 5)):
        self.stats = {"split": "test_final"}
        correct = topks_correct(self.video_preds, self.video_labels, ks)
        for k, c in zip(ks, correct):
            self.stats["top{}_acc".format(k)] = "{:.{prec}f}".format(float(c) / self.video_preds.size(0) * 100.0, prec=2)
        log_json_stats(self.stats)
        return self.stats


@torch.no_grad()
def perform_test(test_loader, model, test_meter, cfg):
    model.eval()
    dev = next(model.parameters()).device
    for inputs, labels, video_idx, _meta in test_loader:
        inputs = inputs.to(dev, non_blocking=True)
        labels = labels.to(dev).view(-1)
        video_idx = video_idx.to(dev).view(-1)
        preds = model(inputs)
        # gather every rank's predictions for the ensemble (test_net.py:112-113)
        if du.get_world_size() > 1 and not (cfg.TRAIN.LABEL_EMB == "" and cfg.TRAIN.TEXT != ""):
            preds, labels, video_idx = du.all_gather([preds, labels, video_idx])
        test_meter.update_stats(preds, labels, video_idx)
    test_meter.finalize_metrics(ks=(1, 5))
    return test_meter


def test(cfg, test_loader=None):
    """tools/test_net.py:161-221: `test(cfg)` builds the model, loads the test checkpoint, constructs its own "test"
    loader and multi-view meter and runs `perform_test`.  (`test_loader` may be injected by tests.)"""
    du.init_distributed_training(cfg)
    torch.manual_seed(cfg.RNG_SEED)
    model = build_model(cfg)
    cu.load_test_checkpoint(cfg, model)
    if test_loader is None:
        from .datasets import construct_loader
        test_loader = construct_loader(cfg, "test")
    num_clips = cfg.TEST.NUM_ENSEMBLE_VIEWS * cfg.TEST.NUM_SPATIAL_CROPS
    assert len(test_loader.dataset) % num_clips == 0
    num_cls = cfg.MODEL.NUM_CLASSES if cfg.TRAIN.LABEL_EMB == "" or cfg.TRAIN.TEXT == "" else 1059     # test_net.py:203
    meter = TestMeter(len(test_loader.dataset) // num_clips, num_clips, num_cls,
                      len(test_loader), cfg.DATA.MULTI_LABEL, cfg.DATA.ENSEMBLE_METHOD)
    perform_test(test_loader, model, meter, cfg)
    return meter
