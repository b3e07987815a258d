"""Synthetic stand-in for the HowTo100M loader with the reference's batch contract.

The reference dataset (`Howto100m_develop.__getitem__`, lib/datasets/howto100m.py:207-362, collated by
lib/datasets/loader.py:128-138) yields `(inputs, labels, index, meta)` with
    inputs  fp32 [b, 9, 3, T, H, W]   mean/std-normalised frames of 9 clips per video
    labels  int64 [b]
    index   int64 [b]
    meta    {clip_text_ids: int [b, 9, 1, 77] CLIP BPE ids, clip_vis_feat: fp32 [b, 9, 512], label: [b, 1]}
Real videos / ASR / CLIP features are not available offline, so this generator draws tensors of the same
shapes and statistics: frames ~ N(0,1); ids = [49406, U{1..49405}^L, 49407, 0...], L ~ U{8..40};
clip_vis_feat ~ N(0, 0.4^2).  Everything is created directly on `device` (inputs resident in HBM).
"""
import torch


def synthetic_label_emb(num_classes, dim=512, seed=0):
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(num_classes, dim, generator=g) * 0.38
    return e / e.norm(dim=1, keepdim=True)


def synthetic_text_ids(n, generator, ctx=77):
    ids = torch.zeros(n, ctx, dtype=torch.long)
    lens = torch.randint(8, 41, (n,), generator=generator)
    for i in range(n):
        L = int(lens[i])
        ids[i, 0] = 49406
        ids[i, 1:1 + L] = torch.randint(1, 49406, (L,), generator=generator)
        ids[i, 1 + L] = 49407
    return ids


class SyntheticHowTo100M(torch.utils.data.Dataset):
    def __init__(self, cfg, num_videos=16, seed=0, clips=9):
        self.cfg = cfg
        self.n = num_videos
        self.seed = seed
        self.clips = clips

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 100003 + index)
        T, S = self.cfg.DATA.NUM_FRAMES, self.cfg.DATA.TRAIN_CROP_SIZE
        frames = torch.randn(self.clips, 3, T, S, S, generator=g)
        ids = synthetic_text_ids(self.clips, g).view(self.clips, 1, 77)
        vis = torch.randn(self.clips, 512, generator=g) * 0.4
        meta = {"clip_text_ids": ids, "clip_vis_feat": vis, "label": torch.tensor([0])}
        return frames, torch.tensor(0), torch.tensor(index), meta


def construct_loader(cfg, split="train", num_videos=None, batch_size=None):
    ds = SyntheticHowTo100M(cfg, num_videos or int(cfg.SYNTHETIC.NUM_VIDEOS))
    bs = batch_size or max(1, int(cfg.TRAIN.BATCH_SIZE / max(1, cfg.NUM_GPUS)))
    return torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=False, num_workers=0, drop_last=True)
