"""Frame-folder clip loader feeding the device: the part of ref/utils/dataset.py the Stage-2 loop needs -
`LitDataModule`'s per-dataset transforms, constants and splits (:25-135), the clip builders of all five datasets - `KTHDataset`
(:267-360), `BAIRDataset` (:362-414), `CityScapesDataset` (:416-443), `KITTIDataset` (:445-515), `StochasticMovingMNIST` (:677-778) -,
`ClipDataset` (:517-575), the `Vid*` transforms (:780-900) and the implicit DistributedSampler -
re-designed around the GPU instead of around torchvision:

  * worker threads decode with PIL and do the GEOMETRIC transforms (centre crop, resize, flips) on uint8 images;
  * a batch crosses PCIe as uint8 HWC (a quarter of the bytes of the reference's fp32 CHW tensors) from pinned memory on a
    copy stream, and `npvp_u8hwc_to_f32chw` (csrc/data.hip) does ToTensor + Normalize + the HWC->CHW transpose in one pass;
  * the sampler is rank-strided over a seeded permutation (one process per GPU, no data traffic between ranks).

torchvision / cv2 are not needed (and are absent in this image).  Pinning: the five clip builders and the SM-MNIST trajectory
generator are checked against vectors the REFERENCE's own classes produced on synthetic trees (tests/golden/make_loader_golden.py:
its module imports with empty stand-ins for torchvision / cv2, which those code paths never touch); what does need torchvision -
`ClipDataset.__getitem__`'s PIL transforms, datasets.MNIST's file reader - is restated from the source text and covered by
property tests only (tests/test_data.py).
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


def _cut_clips(files, clip_length):
    """non-overlapping clips, the remainder dropped half at each end (the same four lines in every reference walker)"""
    n, rem = len(files) // clip_length, len(files) % clip_length
    files = files[rem // 2: rem // 2 + n * clip_length]
    return [files[i * clip_length:(i + 1) * clip_length] for i in range(n)]


KTH_ACTIONS = ('boxing', 'handclapping', 'handwaving', 'jogging_no_empty', 'running_no_empty', 'walking_no_empty')


def kth_clips(kth_dir, clip_length, person_ids, actions=KTH_ACTIONS):
    """ref KTHDataset.__getFramesFolder__ + __getClips__ (dataset.py:327-360): the frame folders `<action>/personNN_...` of all
    actions (entries with '.avi' in their name skipped), sorted by full path, kept when the person id - the last two characters
    of the folder name's first '_' field - is in `person_ids`; every folder cut into non-overlapping clips."""
    root = Path(kth_dir).absolute()
    folders = []
    for a in actions:
        folders.extend(root / a / s_ for s_ in os.listdir(root / a) if '.avi' not in s_)
    clips = []
    for ff in sorted(folders):
        if int(ff.name.strip().split('_')[0][-2:]) in person_ids:
            clips += _cut_clips(sorted(ff.glob('*')), clip_length)
    return clips


def kth_clip_lists(kth_dir, clip_length, train=True, val=True, val_person_ids=(5,), actions=KTH_ACTIONS):
    """ref KTHDataset.__init__ (dataset.py:273-313): persons 1..16 train, 17..25 test.  With `val_person_ids` given (LitDataModule
    passes [5], :38) the validation clips are those persons' - and, as in the reference, they are NOT taken out of the training
    ids (only the `val_person_ids is None` branch removes its randomly drawn person).  Returns {'train', 'val'} or {'test'}."""
    if not train:
        return {"test": kth_clips(kth_dir, clip_length, list(range(17, 26)), actions)}
    person_ids = list(range(1, 17))
    out = {}
    if val:
        if val_person_ids is None:
            import random
            val_person_ids = [random.randint(1, 17)]
            person_ids.remove(val_person_ids[0])
        out["val"] = kth_clips(kth_dir, clip_length, list(val_person_ids), actions)
    out["train"] = kth_clips(kth_dir, clip_length, person_ids, actions)
    return out


def kitti_clip_lists(kitti_dir, clip_length, test_folder_ids=(10, 11, 12, 13), train=True, val=True):
    """ref KITTIDataset (dataset.py:445-515): the drive folders sorted by name; the ones at `test_folder_ids` are the test set, of
    the rest the first two are validation and the others training; every folder cut into non-overlapping clips."""
    root = Path(kitti_dir).absolute()
    folders = sorted(os.listdir(root))
    cut = lambda names: [c for f in names for c in _cut_clips(sorted((root / f).glob('*')), clip_length)]
    if not train:
        return {"test": cut([folders[i] for i in test_folder_ids])}
    tr = [f for i, f in enumerate(folders) if i not in test_folder_ids]
    return {"val": cut(tr[0:2]), "train": cut(tr[2:])} if val else {"train": cut(tr)}


_DIGITS = {}          # (root, train, digit_size) -> array: the 60 000 PIL resizes are done once per process, not once per split


def load_mnist_digits(root, train=True, digit_size=32):
    """The digit images StochasticMovingMNIST draws from (ref dataset.py:693-700: torchvision datasets.MNIST(root, train,
    transform=Resize(32) + ToTensor)) without torchvision: the raw idx file `<root>/MNIST/raw/{train,t10k}-images-idx3-ubyte[.gz]`,
    every 28 x 28 image resized with PIL's bilinear filter (what Resize does to a PIL image) and scaled to [0, 1].
    Returns float32 (N, digit_size, digit_size)."""
    import gzip
    from PIL import Image
    key = (str(Path(root).resolve()), bool(train), int(digit_size))
    if key in _DIGITS:
        return _DIGITS[key]
    name = ("train" if train else "t10k") + "-images-idx3-ubyte"
    base = Path(root) / "MNIST" / "raw" / name
    if base.exists():
        raw = base.read_bytes()
    elif base.with_suffix(".gz").exists() or Path(str(base) + ".gz").exists():
        raw = gzip.open(str(base) + ".gz", "rb").read()
    else:
        raise FileNotFoundError(f"{base}[.gz]: MNIST idx file not found (the reference passes download=False too)")
    magic, n, h, w = np.frombuffer(raw[:16], dtype=">i4")
    if magic != 2051:
        raise ValueError(f"{base}: not an idx3-ubyte image file")
    imgs = np.frombuffer(raw, dtype=np.uint8, offset=16).reshape(n, h, w)
    out = np.empty((n, digit_size, digit_size), dtype=np.float32)
    for i in range(n):
        out[i] = np.asarray(Image.fromarray(imgs[i], mode='L').resize((digit_size, digit_size), Image.BILINEAR), dtype=np.float32) / 255.0
    out.setflags(write=False)
    _DIGITS[key] = out
    return out


class StochasticMovingMNIST:
    """ref StochasticMovingMNIST (dataset.py:677-778; after edenton/svg): `num_digits` digits bounce inside an image_size^2 frame,
    a new random velocity at every wall hit (or a mirrored one when `deterministic`); len = number of digit images, the clip for
    an index is DRAWN, not stored.  Random stream exactly as the reference's: numpy's legacy global generator seeded ONCE with the
    first index asked for (`set_seed`, per loader worker there) - here a private RandomState with the same stream, one per dataset
    object, so concurrent loader threads must go through `__getitem__` under its lock (it is a few hundred microseconds).
    `digits`: float32 (N, 32, 32) in [0, 1] (load_mnist_digits).  __getitem__ gives the clip as uint8 (T, H, W, 1) = what the
    reference's ToPILImage makes of the float frames (x 255, truncated), ready for ClipLoader (mean 0, std 1: VidToTensor only)."""

    sequential_draw = True      # every clip is a draw from ONE random stream: ClipLoader draws them in submission order on one thread,
                                # so a fixed seed gives the same clips run to run (the reference is deterministic per worker process)

    def __init__(self, digits, num_past_frames, num_future_frames, num_digits=2, image_size=64, deterministic=False):
        import threading
        self.digits = np.ascontiguousarray(digits, dtype=np.float32)
        self.num_past_frames, self.num_future_frames = num_past_frames, num_future_frames
        self.seq_len = num_past_frames + num_future_frames
        self.num_digits, self.image_size, self.digit_size = num_digits, image_size, 32
        self.deterministic = deterministic
        self.N = len(self.digits)
        self.epoch = 0
        self._rng = None
        self._lock = threading.Lock()

    def __len__(self):
        return self.N

    def clip_float(self, index):
        """the reference's __getnparray__: float32 (T, 1, H, W), overlapping digits clipped to 1"""
        with self._lock:
            if self._rng is None:
                self._rng = np.random.RandomState(int(index))
            rng, S, dsz = self._rng, self.image_size, self.digit_size
            x = np.zeros((self.seq_len, S, S, 1), dtype=np.float32)
            for _ in range(self.num_digits):
                digit = self.digits[rng.randint(self.N)]
                sx, sy = rng.randint(S - dsz), rng.randint(S - dsz)
                dx, dy = rng.randint(-4, 5), rng.randint(-4, 5)
                for t in range(self.seq_len):
                    if sy < 0:
                        sy = 0
                        if self.deterministic:
                            dy = -dy
                        else:
                            dy = rng.randint(1, 5); dx = rng.randint(-4, 5)
                    elif sy >= S - 32:
                        sy = S - 32 - 1
                        if self.deterministic:
                            dy = -dy
                        else:
                            dy = rng.randint(-4, 0); dx = rng.randint(-4, 5)
                    if sx < 0:
                        sx = 0
                        if self.deterministic:
                            dx = -dx
                        else:
                            dx = rng.randint(1, 5); dy = rng.randint(-4, 5)
                    elif sx >= S - 32:
                        sx = S - 32 - 1
                        if self.deterministic:
                            dx = -dx
                        else:
                            dx = rng.randint(-4, 0); dy = rng.randint(-4, 5)
                    x[t, sy:sy + 32, sx:sx + 32, 0] += digit
                    sy += dy
                    sx += dx
        x[x > 1] = 1.
        return x.transpose(0, 3, 1, 2)

    def __getitem__(self, index):
        x = self.clip_float(index).transpose(0, 2, 3, 1)          # (T, H, W, 1)
        return np.ascontiguousarray((x * np.float32(255.0)).astype(np.uint8))      # ToPILImage: mul(255).byte()


class ClipSubset:
    """torch.utils.data.Subset for ClipDataset-likes (the 95 / 5 splits): forwards the frame counts and the epoch"""

    def __init__(self, dataset, indices):
        self.dataset, self.indices = dataset, list(indices)
        self.num_past_frames, self.num_future_frames = dataset.num_past_frames, dataset.num_future_frames

    epoch = property(lambda self: self.dataset.epoch, lambda self, e: setattr(self.dataset, "epoch", e))

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        return self.dataset[self.indices[i]]


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


# folder walkers of the datasets whose clips come from ONE tree of frame folders
WALKERS = {"BAIR": frame_folder_clips, "CityScapes": cityscapes_clips}


def build_dataset(name, frames_dir, num_past_frames, num_future_frames, train=True, seed=0):
    """ClipDataset over one tree of frame folders with the dataset's constants (LitDataModule.__init__, dataset.py:33-60): BAIR
    (frames_dir = <dir>/train or /test) and CityScapes (frames_dir = <dir>/train, /val or /test).  The whole per-dataset recipe,
    splits included, for all five datasets: build_split."""
    d = DATASETS[name]
    if name not in WALKERS:
        raise ValueError(f"{name}: not a single tree of frame folders - use build_split(name, dataset_dir, ..., split)")
    clips = WALKERS[name](frames_dir, num_past_frames + num_future_frames)
    return ClipDataset(num_past_frames, num_future_frames, clips, d["color"], d["center_crop"], d["resize"],
                       flips=train and d["train_flips"], seed=seed)


def build_split(name, dataset_dir, num_past_frames, num_future_frames, split="train", seed=0):
    """LitDataModule.setup (ref/utils/dataset.py:62-135) for one of its five datasets and one split ('train' | 'val' | 'test'),
    `dataset_dir` = cfg.Dataset.dir:
      KTH         persons 1..16 train (person 5 validation, not removed from train), 17..25 test (:66-70,108-111)
      KITTI       drive folders 10..13 test, the first two others validation (:72-76,113-117)
      BAIR        <dir>/train split 95 / 5 by random_split(seed 2021), <dir>/test (:78-86,119-121)
      CityScapes  <dir>/train, <dir>/val, <dir>/test (:87-93,122-125)
      SMMNIST     clips drawn from the MNIST train / test digits; train split 95 / 5 (:95-101,127-128)
    Quirks kept: the validation sets get the TRAIN transform (flips), and so do the KITTI / CityScapes / SMMNIST test sets."""
    if split not in ("train", "val", "test"):
        raise ValueError(f"split must be 'train', 'val' or 'test', not {split!r}")
    d, L = DATASETS[name], num_past_frames + num_future_frames
    mk = lambda clips, flips: ClipDataset(num_past_frames, num_future_frames, clips, d["color"], d["center_crop"], d["resize"],
                                          flips=flips and d["train_flips"], seed=seed)
    if name == "KTH":
        lists = kth_clip_lists(dataset_dir, L, train=split != "test", val=True, val_person_ids=(5,))
        return mk(lists[split], flips=split != "test")
    if name == "KITTI":
        lists = kitti_clip_lists(dataset_dir, L, (10, 11, 12, 13), train=split != "test", val=True)
        return mk(lists[split], flips=True)
    if name == "BAIR":
        if split == "test":
            return mk(frame_folder_clips(Path(dataset_dir) / "test", L), flips=False)
        whole = mk(frame_folder_clips(Path(dataset_dir) / "train", L), flips=True)
        tr, va = train_val_split(len(whole))
        return ClipSubset(whole, tr if split == "train" else va)
    if name == "CityScapes":
        return mk(cityscapes_clips(Path(dataset_dir) / split, L), flips=True)
    if name == "SMMNIST":
        ds = StochasticMovingMNIST(load_mnist_digits(dataset_dir, train=split != "test"), num_past_frames, num_future_frames)
        if split == "test":
            return ds
        tr, va = train_val_split(len(ds))
        return ClipSubset(ds, tr if split == "train" else va)
    raise KeyError(name)


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
        # a dataset whose items are consecutive draws from one random stream (StochasticMovingMNIST) gets ONE worker thread: tasks
        # run in submission order, so the clips of a batch - and of an epoch - do not depend on thread scheduling
        workers = 1 if getattr(self.ds, "sequential_draw", False) else self.num_workers
        with ThreadPoolExecutor(workers) as pool:
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
