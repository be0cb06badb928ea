"""Frame-folder clip loader feeding the device: the part of ref/utils/dataset.py the Stage-2 loop needs -
`LitDataModule`'s per-dataset transforms and constants (:25-60), the folder walkers of `BAIRDataset` (:401-414) and
`CityScapesDataset` (:420-443) and the 95 / 5 train / validation split (:84-88), `ClipDataset` (:517-575), the `Vid*` transforms (:780-900) and the implicit DistributedSampler -
re-designed around the GPU instead of around torchvision:

  * worker threads decode with PIL and do the GEOMETRIC transforms (centre crop, resize, flips) on uint8 images;
  * a batch crosses PCIe as uint8 HWC (a quarter of the bytes of the reference's fp32 CHW tensors) from pinned memory on a
    copy stream, and `npvp_u8hwc_to_f32chw` (csrc/data.hip) does ToTensor + Normalize + the HWC->CHW transpose in one pass;
  * the sampler is rank-strided over a seeded permutation (one process per GPU, no data traffic between ranks).

torchvision / cv2 are not needed (and are absent in this image).  The reference's module cannot be imported here for the
same reason, so this file is pinned by its own property tests (tests/test_data.py), not by reference-generated vectors.
"""
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

from . import ops
from ._lib import lib

# per-dataset constants and geometric transforms of LitDataModule.__init__ (ref/utils/dataset.py:33-60).
# norm = VidNormalize(mean, std); renorm = VidReNormalize(mean, std) (KITTI's two sets differ in the reference, kept as is).
DATASETS = {
    "KTH": dict(color="grey_scale", norm=((0.6013795,), (2.7570653,)), renorm=((0.6013795,), (2.7570653,)),
                center_crop=(120, 120), resize=(64, 64), train_flips=True),
    "KITTI": dict(color="RGB", norm=((0.44812047, 0.47147775, 0.4677183), (1.5147436, 1.5871466, 1.5925455)),
                  renorm=((0.44811612, 0.47147346, 0.46771598), (1.5177081, 1.5897311, 1.5952978)),
                  center_crop=None, resize=(128, 128), train_flips=True),
    "SMMNIST": dict(color="grey_scale", norm=((0.0,), (1.0,)), renorm=((0.0,), (1.0,)), center_crop=None, resize=None,
                    train_flips=False),
    "BAIR": dict(color="RGB", norm=((0.61749697, 0.6050092, 0.52180636), (2.1824553, 2.1553133, 1.9115673)),
                 renorm=((0.61749697, 0.6050092, 0.52180636), (2.1824553, 2.1553133, 1.9115673)),
                 center_crop=None, resize=None, train_flips=True),
    "CityScapes": dict(color="RGB", norm=((0.31604213, 0.35114038, 0.3104223), (1.2172801, 1.3219808, 1.2082524)),
                       renorm=((0.31604213, 0.35114038, 0.3104223), (1.2172801, 1.3219808, 1.2082524)),
                       center_crop=None, resize=None, train_flips=False),
}


def frame_folder_clips(frames_dir, clip_length):
    """Non-overlapping clips of `clip_length` consecutive (name-sorted) frames from every sub-folder of `frames_dir`;
    a folder's remainder is dropped half at each end (ref BAIRDataset.__getClips__, dataset.py:401-414)."""
    root = Path(frames_dir).absolute()
    clips = []
    for folder in sorted(p for p in root.iterdir() if p.is_dir()):
        files = sorted(folder.glob('*'))
        n, rem = len(files) // clip_length, len(files) % clip_length
        files = files[rem // 2: rem // 2 + n * clip_length]
        clips += [files[i * clip_length:(i + 1) * clip_length] for i in range(n)]
    return clips


def cityscapes_clips(frames_dir, clip_length):
    """ref CityScapesDataset.__getClips__ (dataset.py:420-443): inside every sub-folder the (name-sorted) files are grouped by
    the sequence id in their name (`<city>_<sequence>_<frame>_...`: fields 1 and 2 of the FILE NAME split at '_'; the reference
    splits the full path string, which gives the same fields as long as no directory above has an underscore in its name),
    each sequence is cut into runs of consecutive frame numbers, and every run is cut into non-overlapping clips with the
    remainder dropped half at each end - a clip never straddles two sequences or a gap."""
    from itertools import groupby
    root = Path(frames_dir).absolute()
    clips = []
    for folder in (root / s_ for s_ in os.listdir(root)):           # the reference does not sort the folder list either
        if not folder.is_dir():
            continue
        by_seq = {}
        for f in sorted(folder.glob('*')):
            by_seq.setdefault(f.name.split('_')[1], []).append(f)
        for files in by_seq.values():
            for _, run in groupby(enumerate(files), lambda ix: ix[0] - int(ix[1].name.split('_')[2])):
                run = [f for _, f in run]
                n, rem = len(run) // clip_length, len(run) % clip_length
                run = run[rem // 2: rem // 2 + n * clip_length]
                clips += [run[i * clip_length:(i + 1) * clip_length] for i in range(n)]
    return clips


def train_val_split(dataset_len, train_ratio=0.95, seed=2021):
    """index lists of the reference's BAIR / SM-MNIST 95 / 5 train / validation split: torch random_split with
    Generator().manual_seed(2021) (dataset.py:84-88,100-103)"""
    n_train = int(dataset_len * train_ratio)
    perm = torch.randperm(dataset_len, generator=torch.Generator().manual_seed(seed)).tolist()
    return perm[:n_train], perm[n_train:]


class ClipDataset:
    """ref ClipDataset (dataset.py:517-575) up to the point where pixels become floats: __getitem__ gives the clip as ONE
    uint8 array (T, H, W, C) after the geometric transforms; ToTensor + Normalize happen on the device (ClipLoader)."""

    def __init__(self, num_past_frames, num_future_frames, clips, color_mode="RGB", center_crop=None, resize=None,
                 flips=False, seed=0):
        if color_mode not in ("RGB", "grey_scale"):
            raise ValueError("Unsupported color mode!!")
        self.num_past_frames, self.num_future_frames = num_past_frames, num_future_frames
        self.clips, self.color_mode = clips, color_mode
        self.center_crop, self.resize, self.flips, self.seed = center_crop, resize, flips, seed
        self.epoch = 0

    def __len__(self):
        return len(self.clips)

    def __getitem__(self, index):
        from PIL import Image
        rng = np.random.default_rng((self.seed, self.epoch, int(index)))
        hflip = self.flips and rng.random() < 0.5          # one draw per clip and axis (ref VidRandom*Flip, :813-833)
        vflip = self.flips and rng.random() < 0.5
        frames = []
        for path in self.clips[index]:
            img = Image.open(os.fspath(path)).convert('RGB' if self.color_mode == 'RGB' else 'L')
            if self.center_crop is not None:               # torchvision CenterCrop: top/left = round((size - crop) / 2)
                ch, cw = self.center_crop
                top, left = int(round((img.height - ch) / 2.0)), int(round((img.width - cw) / 2.0))
                img = img.crop((left, top, left + cw, top + ch))
            if self.resize is not None:                    # torchvision Resize((h, w)) on a PIL image: bilinear
                img = img.resize((self.resize[1], self.resize[0]), Image.BILINEAR)
            a = np.asarray(img, dtype=np.uint8)
            a = a[:, :, None] if a.ndim == 2 else a
            if hflip:
                a = a[:, ::-1]
            if vflip:
                a = a[::-1]
            frames.append(a)
        return np.ascontiguousarray(np.stack(frames, 0))


# folder walkers by dataset.  KTH (person / action splits, dataset.py:253-360), KITTI (test-folder ids + overlapping windows,
# :445-515) and SM-MNIST (generated digits, :578-700) are NOT folder-of-frame-folders trees: they have no walker here.
WALKERS = {"BAIR": frame_folder_clips, "CityScapes": cityscapes_clips}


def build_dataset(name, frames_dir, num_past_frames, num_future_frames, train=True, seed=0):
    """the reference's per-dataset recipe (LitDataModule, dataset.py:33-60) for the datasets whose clips come from a tree of
    frame folders: BAIR (frames_dir = <dir>/train or /test; the reference then splits train 95 / 5: train_val_split) and
    CityScapes (frames_dir = <dir>/train, /val or /test).  The per-dataset constants of KTH / KITTI / SM-MNIST are in DATASETS
    for ClipDataset users who bring their own clip lists; their walkers are not built."""
    d = DATASETS[name]
    if name not in WALKERS:
        raise NotImplementedError(f"{name}: the reference builds its clip list from dataset-specific structure (person / action "
                                  "splits, test-folder ids, generated digits), not from a tree of frame folders; pass your own "
                                  "clip lists to ClipDataset with DATASETS[name]'s constants")
    clips = WALKERS[name](frames_dir, num_past_frames + num_future_frames)
    return ClipDataset(num_past_frames, num_future_frames, clips, d["color"], d["center_crop"], d["resize"],
                       flips=train and d["train_flips"], seed=seed)


def shard_indices(n, batch_size, rank=0, world=1, shuffle=True, seed=0, epoch=0, drop_last=True):
    """This rank's index list for one epoch: a seeded permutation (the same on every rank), cut to a multiple of the
    global batch when drop_last (Lightning's DistributedSampler + DataLoader(drop_last=True)), rank-strided."""
    order = torch.randperm(n, generator=torch.Generator().manual_seed(seed + epoch)).tolist() if shuffle else list(range(n))
    if drop_last:
        order = order[: (n // (batch_size * world)) * batch_size * world]
    else:
        order = order + order[: (-len(order)) % world]        # pad so that every rank gets the same count
    return order[rank::world]


class VidNormalize:
    """(x - mean) / std per channel on a (..., C, H, W) tensor, out of place (ref :846-858 normalises in place)."""

    def __init__(self, mean, std):
        self.mean, self.std = tuple(np.atleast_1d(mean).tolist()), tuple(np.atleast_1d(std).tolist())

    def _mv(self, x):
        shape = (-1, 1, 1)
        return (torch.tensor(self.mean, dtype=x.dtype, device=x.device).view(shape),
                torch.tensor(self.std, dtype=x.dtype, device=x.device).view(shape))

    def __call__(self, x):
        m, s = self._mv(x)
        return (x - m) / s


class VidReNormalize(VidNormalize):
    """the inverse, x * std + mean (ref :860-886)"""

    def __call__(self, x):
        m, s = self._mv(x)
        return x * s + m


class ClipLoader:
    """Iterates (past, future) fp32 batches of shape (B, T, C, H, W) ON THE DEVICE.  `prefetch` batches are decoded ahead by
    `num_workers` threads; each batch is copied as uint8 from pinned memory on a private copy stream and normalised there;
    the consumer's stream waits for that batch's event only."""

    def __init__(self, dataset, batch_size, mean, std, device="cuda:0", shuffle=True, drop_last=True, rank=0, world=1, seed=0,
                 num_workers=8, prefetch=2):
        self.ds, self.bs, self.dev = dataset, batch_size, torch.device(device)
        self.mean = np.ascontiguousarray(np.atleast_1d(mean), dtype=np.float32)
        self.std = np.ascontiguousarray(np.atleast_1d(std), dtype=np.float32)
        self.shuffle, self.drop_last, self.rank, self.world, self.seed = shuffle, drop_last, rank, world, seed
        self.num_workers, self.prefetch = max(1, num_workers), max(1, prefetch)
        self.epoch = 0
        if self.dev.type != "cuda":
            raise RuntimeError("ClipLoader feeds an MI355X: device must be a cuda device (no CPU path)")
        self._copy = torch.cuda.Stream(device=self.dev)

    def set_epoch(self, epoch):
        self.epoch = epoch
        self.ds.epoch = epoch

    def _batches(self):
        idx = shard_indices(len(self.ds), self.bs, self.rank, self.world, self.shuffle, self.seed, self.epoch, self.drop_last)
        out = [idx[i:i + self.bs] for i in range(0, len(idx), self.bs)]
        if self.drop_last and out and len(out[-1]) < self.bs:
            out.pop()
        return out

    def __len__(self):
        return len(self._batches())

    def _upload(self, clips):
        """list of (T,H,W,C) uint8 arrays -> normalised fp32 (B,T,C,H,W) on the device + the event that marks it ready"""
        B, (T, H, W, C) = len(clips), clips[0].shape
        host = torch.empty((B, T, H, W, C), dtype=torch.uint8).pin_memory()
        hv = host.numpy()
        for i, c in enumerate(clips):
            if c.shape != (T, H, W, C):
                raise ValueError(f"clips of one batch differ in shape: {c.shape} vs {(T, H, W, C)}")
            hv[i] = c
        with torch.cuda.stream(self._copy):
            raw = host.to(self.dev, non_blocking=True)
            out = torch.empty((B, T, C, H, W), dtype=torch.float32, device=self.dev)
            ops.check(lib().npvp_u8hwc_to_f32chw(raw.data_ptr(), out.data_ptr(), B * T, H, W, C, self.mean.ctypes.data,
                                                 self.std.ctypes.data, self._copy.cuda_stream), "npvp_u8hwc_to_f32chw")
            ev = torch.cuda.Event()
            ev.record(self._copy)
        return out, ev, (host, raw)

    def __iter__(self):
        batches = self._batches()
        P = self.ds.num_past_frames
        with ThreadPoolExecutor(self.num_workers) as pool:
            pending = []
            nxt = 0
            while nxt < len(batches) or pending:
                while nxt < len(batches) and len(pending) < self.prefetch:
                    pending.append([pool.submit(self.ds.__getitem__, i) for i in batches[nxt]])
                    nxt += 1
                clips = [f.result() for f in pending.pop(0)]
                out, ev, keep = self._upload(clips)
                torch.cuda.current_stream(self.dev).wait_event(ev)
                out.record_stream(torch.cuda.current_stream(self.dev))
                yield out[:, :P], out[:, -self.ds.num_future_frames:]
                del keep
