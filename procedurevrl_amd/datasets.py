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
import torch.distributed as dist
from torch.utils.data.distributed import DistributedSampler


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


class SyntheticTestClips(torch.utils.data.Dataset):
    """Multi-view test split: entry `i` is view `i % num_clips` of video `i // num_clips` (the reference's test
    datasets index clips that way; `TestMeter.update_stats` recovers the video as `clip_id // num_clips`,
    lib/utils/meters.py:104-131).  Frames of one video share a per-video pattern so the ensemble is meaningful."""

    def __init__(self, cfg, num_videos=8, seed=0):
        self.cfg = cfg
        self.num_clips = cfg.TEST.NUM_ENSEMBLE_VIEWS * cfg.TEST.NUM_SPATIAL_CROPS
        self.n = num_videos * self.num_clips
        self.seed = seed

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        vid = index // self.num_clips
        T, S = self.cfg.DATA.NUM_FRAMES, self.cfg.DATA.TEST_CROP_SIZE
        gv = torch.Generator().manual_seed(self.seed * 100003 + 7919 * vid)
        base = torch.randn(3, T, S, S, generator=gv)
        g = torch.Generator().manual_seed(self.seed * 100003 + 31 * index + 1)
        frames = base + 0.5 * torch.randn(3, T, S, S, generator=g)
        label = int(torch.randint(0, max(1, int(self.cfg.MODEL.NUM_CLASSES)), (1,), generator=gv))
        return frames, torch.tensor(label), torch.tensor(index), {}


class SyntheticValClips(torch.utils.data.Dataset):
    """Validation split for `eval_epoch` (tools/train_net.py:251): labelled inputs in the layout the model's eval forward
    takes under `cfg` -- [m, 3, T, S, S] with m labels per video when DEV.ORDER_PRETRAIN_ENABLED regroups `b m c t h w`
    (vit.py:290-291), one [3, T, S, S] clip with one label otherwise."""

    def __init__(self, cfg, num_videos=8, seed=1, clips=9):
        self.cfg = cfg
        self.n = num_videos
        self.seed = seed
        self.clips = clips if cfg.DEV.ORDER_PRETRAIN_ENABLED else 0

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 100003 + 17 * index + 5)
        T, S = self.cfg.DATA.NUM_FRAMES, self.cfg.DATA.TRAIN_CROP_SIZE
        K = max(1, int(self.cfg.MODEL.NUM_CLASSES))
        if self.clips:
            return (torch.randn(self.clips, 3, T, S, S, generator=g), torch.randint(0, K, (self.clips,), generator=g),
                    torch.tensor(index), {})
        x = torch.randn(3, T, S, S, generator=g)
        if self.cfg.TRAIN.DATASET == "Epickitchens":            # lib/datasets/epickitchens.py: label = {'verb', 'noun'} (97 / 300 classes)
            return x, {"verb": torch.randint(0, 97, (1,), generator=g)[0], "noun": torch.randint(0, 300, (1,), generator=g)[0]}, \
                torch.tensor(index), {}
        return x, torch.randint(0, K, (1,), generator=g)[0], torch.tensor(index), {}


def create_sampler(dataset, shuffle, cfg):
    """lib/datasets/utils.py:358-370: a DistributedSampler whenever the job has more than one GPU process."""
    if cfg.NUM_GPUS * max(1, cfg.NUM_SHARDS) > 1 and dist.is_available() and dist.is_initialized():
        return DistributedSampler(dataset, shuffle=shuffle)
    return None


def construct_loader(cfg, split="train", num_videos=None, batch_size=None):
    """lib/datasets/loader.py:85-138 for the synthetic stand-in datasets: per-process batch = BATCH_SIZE / NUM_GPUS,
    train shuffles and drops the last ragged batch, test does neither; ranks see disjoint samples through the sampler."""
    assert split in ("train", "val", "test")
    n = num_videos or int(cfg.SYNTHETIC.NUM_VIDEOS)
    if split == "test":
        ds = SyntheticTestClips(cfg, n)
        bs = batch_size or max(1, int(cfg.TEST.BATCH_SIZE / max(1, cfg.NUM_GPUS)))
        shuffle, drop_last = False, False
    elif split == "val":
        ds = SyntheticValClips(cfg, max(1, n // 2))
        bs = batch_size or max(1, int(cfg.TRAIN.BATCH_SIZE / max(1, cfg.NUM_GPUS)))
        shuffle, drop_last = False, False
    else:
        # the pre-training tuple (clips + narration tokens) or, for the fine-tuning configs (TRAIN.LABEL_EMB / TRAIN.TEXT empty),
        # labelled clips
        pretrain = cfg.TRAIN.LABEL_EMB != "" and cfg.TRAIN.TEXT != ""
        ds = SyntheticHowTo100M(cfg, n) if pretrain else SyntheticValClips(cfg, n, seed=2)
        bs = batch_size or max(1, int(cfg.TRAIN.BATCH_SIZE / max(1, cfg.NUM_GPUS)))
        shuffle, drop_last = True, True
    sampler = create_sampler(ds, shuffle, cfg)
    return torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=(False if sampler else shuffle), sampler=sampler,
                                       num_workers=0, drop_last=drop_last,
                                       pin_memory=bool(cfg.DATA_LOADER.PIN_MEMORY) and torch.cuda.is_available())


class DevicePrefetcher:
    """Host-to-device overlap for the training loop (the reference pins loader memory and calls
    `.cuda(non_blocking=True)` per batch inside the iteration, tools/train_net.py:103-119, DATA_LOADER.PIN_MEMORY).
    Wraps a loader that yields `(inputs, labels, index, meta)`: batch i+1 is staged in pinned host memory and copied on a
    dedicated HIP stream while step i computes; the consumer's stream waits on the copy's event, nothing blocks the host.
    A 32-clip fp32 batch is 154 MB (2.4 ms at 63 GB/s of PCIe), a decoded uint8 batch 38 MB -- hidden either way."""

    def __init__(self, loader, device):
        self.loader = loader
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    @property
    def sampler(self):
        return self.loader.sampler

    @property
    def dataset(self):
        return self.loader.dataset

    def _to_device(self, obj):
        if torch.is_tensor(obj):
            if obj.device.type != "cpu":
                return obj.to(self.device, non_blocking=True)
            src = obj if obj.is_pinned() else obj.pin_memory()
            out = src.to(self.device, non_blocking=True)
            self._hold.append(src)                      # the pinned source must outlive the asynchronous copy
            return out
        if isinstance(obj, dict):
            return {k: self._to_device(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._to_device(v) for v in obj)
        return obj

    def _stage(self, batch):
        self._hold = []
        if self.stream is None:
            return self._to_device(batch), None, []
        with torch.cuda.stream(self.stream):
            dev_batch = self._to_device(batch)
            ev = self.stream.record_event()
        return dev_batch, ev, self._hold

    @staticmethod
    def _record(obj, stream):
        if torch.is_tensor(obj):
            obj.record_stream(stream)
        elif isinstance(obj, dict):
            for v in obj.values():
                DevicePrefetcher._record(v, stream)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                DevicePrefetcher._record(v, stream)

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            batch, ev, hold = nxt
            try:
                nxt = self._stage(next(it))             # next batch's copy is in flight while this one is consumed
            except StopIteration:
                nxt = None
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                self._record(batch, cur)                # allocated on the copy stream, consumed on the compute stream
            yield batch
            del hold


def shuffle_dataset(loader, cur_epoch):
    """lib/datasets/loader.py:140-157: DistributedSampler reshuffles per epoch through set_epoch."""
    if isinstance(loader.sampler, DistributedSampler):
        loader.sampler.set_epoch(cur_epoch)
