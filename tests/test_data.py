"""Frame-folder loader: folder walker, sampler and geometric transforms on the CPU; the uint8 -> normalised
fp32 kernel and the loader end to end on the GPU.  The five clip builders and the SM-MNIST generator are checked against vectors
produced by the reference's own classes (tests/golden/make_loader_golden.py); the PIL transforms of ClipDataset need torchvision on
the reference side (absent here), so those are restated-semantics and property tests, not reference-vector tests."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npvp_amd import data as D            # noqa: E402

DEV = "cuda:0"


def make_tree(root, folders=3, frames=(23, 20, 7), size=(140, 160), rgb=True, seed=0):
    """folders of PNG frames whose pixel values encode (folder, frame) so that order mistakes are visible"""
    from PIL import Image
    rng = np.random.default_rng(seed)
    truth = {}
    for f in range(folders):
        d = root / f"vid_{f:03d}"
        d.mkdir(parents=True)
        for t in range(frames[f]):
            a = rng.integers(0, 256, size=size + ((3,) if rgb else ()), dtype=np.uint8)
            a[0, 0] = f; a[0, 1] = t
            Image.fromarray(a, 'RGB' if rgb else 'L').save(d / f"frame_{t:04d}.png")
            truth[(f, t)] = a
    return truth


def test_frame_folder_clips_centres_the_remainder(tmp_path):
    make_tree(tmp_path, size=(8, 8))
    clips = D.frame_folder_clips(tmp_path, 10)
    # 23 frames -> 2 clips starting at frame 1 (remainder 3: one dropped in front, two behind); 20 -> 2 clips; 7 -> none
    assert len(clips) == 4
    assert [p.name for p in clips[0]] == [f"frame_{t:04d}.png" for t in range(1, 11)]
    assert [p.name for p in clips[1]] == [f"frame_{t:04d}.png" for t in range(11, 21)]
    assert clips[2][0].parent.name == "vid_001" and clips[3][-1].name == "frame_0019.png"


def test_cityscapes_walker_respects_sequences_and_gaps(tmp_path):
    """ref CityScapesDataset.__getClips__: clips never straddle two sequence ids or a gap in the frame numbers"""
    root = tmp_path / "cs"
    city = root / "aachen"
    city.mkdir(parents=True)
    names = ([f"aachen_000001_{t:06d}_leftImg8bit.png" for t in range(0, 13)]              # 13 consecutive frames
             + [f"aachen_000001_{t:06d}_leftImg8bit.png" for t in range(20, 26)]          # same sequence after a gap: 6 frames
             + [f"aachen_000002_{t:06d}_leftImg8bit.png" for t in range(5, 10)])          # another sequence: 5 frames
    for n in names:
        (city / n).write_bytes(b"")
    clips = D.cityscapes_clips(root, 5)
    got = [[p.name.split("_")[1] + ":" + str(int(p.name.split("_")[2])) for p in c] for c in clips]
    # 13 frames -> 2 clips from frame 1 (remainder 3: one dropped in front); 6 frames -> 1 clip from frame 20; 5 frames -> 1 clip
    assert got == [[f"000001:{t}" for t in range(1, 6)], [f"000001:{t}" for t in range(6, 11)],
                   [f"000001:{t}" for t in range(20, 25)], [f"000002:{t}" for t in range(5, 10)]]
    for name in ("KTH", "KITTI", "SMMNIST"):           # not single trees of frame folders: build_split owns their recipes
        with pytest.raises(ValueError):
            D.build_dataset(name, root, 4, 4)


# ---- the five clip builders against what the REFERENCE's own walker classes produced on synthetic trees
# ---- (tests/golden/make_loader_golden.py ran KTHDataset / KITTIDataset / BAIRDataset / CityScapesDataset; fixture = file names only)
def _golden_tree(tmp_path, name):
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "loader_walkers.json")))[name]
    root = tmp_path / name.lower()
    for f in g["files"]:
        (root / f).parent.mkdir(parents=True, exist_ok=True)
        (root / f).touch()
    rel = lambda clips: [[os.path.relpath(str(f), str(root.absolute())) for f in c] for c in clips]
    return g, root, rel


def test_kth_clip_lists_match_the_reference_walker(tmp_path):
    """ref KTHDataset (dataset.py:267-360): persons 1..16 train, [5] validation (and still in train), 17..25 test, '.avi' skipped"""
    g, root, rel = _golden_tree(tmp_path, "KTH")
    lists = D.kth_clip_lists(root, g["clip_length"], train=True, val=True, val_person_ids=[5])
    assert rel(lists["train"]) == g["train"] and rel(lists["val"]) == g["val"]
    assert rel(D.kth_clip_lists(root, g["clip_length"], train=False, val=False)["test"]) == g["test"]
    assert all(c in g["train"] for c in g["val"]), "the reference keeps the validation person in the training list"
    ds = D.build_split("KTH", root, 4, 6, "val")
    assert len(ds) == len(g["val"]) and ds.flips and ds.center_crop == (120, 120) and ds.resize == (64, 64)
    assert not D.build_split("KTH", root, 4, 6, "test").flips


def test_kitti_clip_lists_match_the_reference_walker(tmp_path):
    """ref KITTIDataset (dataset.py:445-515): sorted drive folders, ids 10..13 test, the first two others validation"""
    g, root, rel = _golden_tree(tmp_path, "KITTI")
    lists = D.kitti_clip_lists(root, g["clip_length"], (10, 11, 12, 13), train=True, val=True)
    assert rel(lists["train"]) == g["train"] and rel(lists["val"]) == g["val"]
    assert rel(D.kitti_clip_lists(root, g["clip_length"], (10, 11, 12, 13), train=False)["test"]) == g["test"]
    assert len(D.build_split("KITTI", root, 3, 4, "train")) == len(g["train"])
    assert D.build_split("KITTI", root, 3, 4, "test").flips, "the reference gives its KITTI test set the TRAIN transform"


def test_bair_and_cityscapes_walkers_match_the_reference(tmp_path):
    """ref BAIRDataset / CityScapesDataset.__getClips__ (dataset.py:401-443); the reference lists folders unsorted: compared sorted"""
    g, root, rel = _golden_tree(tmp_path, "BAIR")
    assert sorted(rel(D.frame_folder_clips(root, g["clip_length"]))) == g["clips"]
    g, root, rel = _golden_tree(tmp_path, "CityScapes")
    assert sorted(rel(D.cityscapes_clips(root, g["clip_length"]))) == g["clips"]


@pytest.mark.parametrize("tag", ["stochastic", "deterministic"])
def test_stochastic_moving_mnist_matches_the_reference_generator(tag):
    """ref StochasticMovingMNIST.__getnparray__ (dataset.py:716-778) run on seeded synthetic digits: the same clips, bit for bit
    (numpy's legacy stream seeded once with the first index), and the uint8 frames its ToPILImage would hand to VidToTensor"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "smmnist.npz"))
    ds = D.StochasticMovingMNIST(g["digits"], 5, 7, deterministic=tag == "deterministic")
    idx = g[f"idx_{tag}"].tolist()
    clips = np.stack([ds.clip_float(i) for i in idx], 0)
    assert clips.shape == g[f"clips_{tag}"].shape and np.array_equal(clips, g[f"clips_{tag}"])
    ds2 = D.StochasticMovingMNIST(g["digits"], 5, 7, deterministic=tag == "deterministic")
    u8 = ds2[idx[0]]
    assert u8.dtype == np.uint8 and u8.shape == (12, 64, 64, 1)
    assert np.array_equal(u8[..., 0], (g[f"clips_{tag}"][0][:, 0] * np.float32(255)).astype(np.uint8))


def test_mnist_idx_reader_and_95_5_split(tmp_path):
    """load_mnist_digits reads torchvision's raw idx layout; build_split('SMMNIST') = the generator over it, split 95 / 5"""
    raw = tmp_path / "MNIST" / "raw"
    raw.mkdir(parents=True)
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, size=(40, 28, 28), dtype=np.uint8)
    hdr = np.array([2051, 40, 28, 28], dtype=">i4").tobytes()
    (raw / "train-images-idx3-ubyte").write_bytes(hdr + imgs.tobytes())
    d = D.load_mnist_digits(tmp_path, train=True)
    assert d.shape == (40, 32, 32) and d.dtype == np.float32 and 0.0 <= d.min() and d.max() <= 1.0
    from PIL import Image
    assert np.array_equal(d[3], np.asarray(Image.fromarray(imgs[3], 'L').resize((32, 32), Image.BILINEAR), dtype=np.float32) / 255.0)
    tr, va = D.build_split("SMMNIST", tmp_path, 5, 15, "train"), D.build_split("SMMNIST", tmp_path, 5, 15, "val")
    assert len(tr) == 38 and len(va) == 2 and tr[0].shape == (20, 64, 64, 1)
    assert sorted(tr.indices + va.indices) == list(range(40))
    with pytest.raises(FileNotFoundError):
        D.load_mnist_digits(tmp_path, train=False)


def test_train_val_split_is_torch_random_split():
    from torch.utils.data import random_split
    n = 257
    tr, va = random_split(list(range(n)), [int(n * 0.95), n - int(n * 0.95)], generator=torch.Generator().manual_seed(2021))
    a, b = D.train_val_split(n)
    assert a == list(tr.indices) and b == list(va.indices)


def test_shard_indices_partition_the_epoch():
    n, bs, world = 103, 4, 3
    per_rank = [D.shard_indices(n, bs, r, world, seed=5, epoch=2) for r in range(world)]
    assert len({len(p) for p in per_rank}) == 1 and len(per_rank[0]) == (n // (bs * world)) * bs
    flat = sorted(i for p in per_rank for i in p)
    assert len(set(flat)) == len(flat) and set(flat) <= set(range(n))
    assert per_rank[0] != D.shard_indices(n, bs, 0, world, seed=5, epoch=3)           # reshuffled per epoch
    assert per_rank[1] == D.shard_indices(n, bs, 1, world, seed=5, epoch=2)           # and reproducible
    full = [D.shard_indices(10, 4, r, 4, shuffle=False, drop_last=False) for r in range(4)]
    assert all(len(p) == 3 for p in full) and sorted(set(i for p in full for i in p)) == list(range(10))


def test_clip_dataset_geometry(tmp_path):
    from PIL import Image
    truth = make_tree(tmp_path, folders=1, frames=(6,), size=(140, 160), rgb=False)
    clips = D.frame_folder_clips(tmp_path, 6)
    ds = D.ClipDataset(2, 4, clips, "grey_scale", center_crop=(120, 120), resize=(64, 64))
    clip = ds[0]
    assert clip.shape == (6, 64, 64, 1) and clip.dtype == np.uint8
    ref = Image.fromarray(truth[(0, 3)][10:130, 20:140], 'L').resize((64, 64), Image.BILINEAR)
    assert np.array_equal(clip[3, :, :, 0], np.asarray(ref))
    plain = D.ClipDataset(2, 4, clips, "grey_scale")[0]
    assert np.array_equal(plain[5, :, :, 0], truth[(0, 5)])
    # flips: one decision per clip and axis, reproducible per (seed, epoch, index), and all four outcomes occur
    seen = set()
    for s in range(24):
        f = D.ClipDataset(2, 4, clips, "grey_scale", flips=True, seed=s)
        a = f[0]
        assert np.array_equal(a, f[0])
        h = np.array_equal(a[:, :, ::-1], plain); v = np.array_equal(a[:, ::-1], plain); hv = np.array_equal(a[:, ::-1, ::-1], plain)
        assert np.array_equal(a, plain) or h or v or hv
        seen.add((np.array_equal(a, plain), h, v, hv))
    assert len(seen) == 4
    with pytest.raises(ValueError):
        D.ClipDataset(2, 4, clips, "YUV")


def test_dataset_table_matches_reference_constants():
    assert set(D.DATASETS) == {"KTH", "KITTI", "SMMNIST", "BAIR", "CityScapes"}
    assert D.DATASETS["KTH"]["norm"] == ((0.6013795,), (2.7570653,)) and D.DATASETS["KTH"]["center_crop"] == (120, 120)
    assert D.DATASETS["KITTI"]["norm"] != D.DATASETS["KITTI"]["renorm"]            # the reference's two sets differ
    x = torch.rand(2, 3, 4, 4)
    n, r = D.VidNormalize(*D.DATASETS["BAIR"]["norm"]), D.VidReNormalize(*D.DATASETS["BAIR"]["renorm"])
    assert torch.allclose(r(n(x)), x, atol=1e-6)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("C,H,W", [(1, 64, 64), (3, 128, 128), (3, 45, 70), (4, 5, 3)])
def test_u8_to_normalised_kernel(C, H, W):
    from npvp_amd import ops
    from npvp_amd._lib import lib
    F_ = 7
    rng = np.random.default_rng(C * 100 + H)
    src = rng.integers(0, 256, size=(F_, H, W, C), dtype=np.uint8)
    mean = np.linspace(0.2, 0.6, C).astype(np.float32); std = np.linspace(1.1, 2.3, C).astype(np.float32)
    s = torch.from_numpy(src).to(DEV)
    out = torch.empty(F_, C, H, W, device=DEV)
    ops.check(lib().npvp_u8hwc_to_f32chw(s.data_ptr(), out.data_ptr(), F_, H, W, C, mean.ctypes.data, std.ctypes.data,
                                         ops._stream()), "npvp_u8hwc_to_f32chw")
    ref = (src.astype(np.float32).transpose(0, 3, 1, 2) / 255.0 - mean[None, :, None, None]) / std[None, :, None, None]
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= 2e-6
    assert lib().npvp_u8hwc_to_f32chw(s.data_ptr(), out.data_ptr(), F_, H, W, 2, mean.ctypes.data, std.ctypes.data, None) < 0


@pytest.mark.gpu
def test_clip_loader_end_to_end(tmp_path):
    truth = make_tree(tmp_path, folders=3, frames=(23, 20, 7), size=(64, 64), rgb=True)
    ds = D.build_dataset("BAIR", tmp_path, 2, 8, train=False)
    assert len(ds) == 4
    mean, std = D.DATASETS["BAIR"]["norm"]
    got = {}
    for rank in range(2):
        ld = D.ClipLoader(ds, 1, mean, std, DEV, shuffle=True, rank=rank, world=2, seed=3, num_workers=3)
        ld.set_epoch(1)
        assert len(ld) == 2
        for (past, fut), idx in zip(ld, D.shard_indices(4, 1, rank, 2, seed=3, epoch=1)):
            assert past.shape == (1, 2, 3, 64, 64) and fut.shape == (1, 8, 3, 64, 64) and past.is_cuda
            got[idx] = torch.cat([past, fut], 1).cpu()
    assert sorted(got) == [0, 1, 2, 3]
    m = torch.tensor(mean).view(1, 1, 3, 1, 1); s = torch.tensor(std).view(1, 1, 3, 1, 1)
    for idx, clip in got.items():
        f = 0 if idx < 2 else 1
        t0 = (1 if f == 0 else 0) + 10 * (idx % 2)
        ref = np.stack([truth[(f, t0 + t)] for t in range(10)], 0)[None]
        ref = (torch.from_numpy(ref).float().permute(0, 1, 4, 2, 3) / 255.0 - m) / s
        assert torch.allclose(clip, ref, atol=2e-6)
        back = D.VidReNormalize(*D.DATASETS["BAIR"]["renorm"])(clip) * 255.0
        assert torch.equal(back.round().to(torch.uint8), torch.from_numpy(np.stack([truth[(f, t0 + t)] for t in range(10)], 0)[None]).permute(0, 1, 4, 2, 3))
    with pytest.raises(RuntimeError):
        D.ClipLoader(ds, 1, mean, std, "cpu")


def test_mnist_digits_are_resized_once_per_process(tmp_path):
    """build_split('SMMNIST', 'train') and (..., 'val') read the same idx file: the 60 000 PIL resizes are cached"""
    raw = tmp_path / "MNIST" / "raw"
    raw.mkdir(parents=True)
    imgs = np.random.default_rng(1).integers(0, 256, size=(12, 28, 28), dtype=np.uint8)
    (raw / "train-images-idx3-ubyte").write_bytes(np.array([2051, 12, 28, 28], dtype=">i4").tobytes() + imgs.tobytes())
    a, b = D.load_mnist_digits(tmp_path, train=True), D.load_mnist_digits(tmp_path, train=True)
    assert a is b and not a.flags.writeable
    assert D.StochasticMovingMNIST.sequential_draw


@pytest.mark.gpu
def test_smmnist_loader_is_reproducible():
    """StochasticMovingMNIST draws every clip from ONE random stream: ClipLoader hands such a dataset to one worker thread, so two
    loaders with the same seeds yield the same batches (with eight threads racing for the stream's lock they did not: ADVICE r4)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "smmnist.npz"))
    runs = []
    for _ in range(2):
        ds = D.StochasticMovingMNIST(g["digits"], 5, 7)
        ld = D.ClipLoader(ds, 4, 0.0, 1.0, DEV, shuffle=True, seed=11, num_workers=8, prefetch=3)
        ld.set_epoch(2)
        runs.append([torch.cat([p, f], 1).cpu() for (p, f), _ in zip(ld, range(3))])
    assert len(runs[0]) == 3
    for a, b in zip(*runs):
        assert torch.equal(a, b)
