"""Evaluation metrics: the oracle restatement against vectors written by the reference's own
utils/metrics.py (tests/golden/metrics.npz), and - on the GPU - npvp_amd.metrics (csrc/metrics.hip through the C ABI)
against both."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ops as O               # noqa: E402
from oracle import metrics as OM          # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "metrics.npz"))
CASES = {"g64": ((3, 1, 64, 64), 1.0), "rgb128": ((2, 3, 128, 128), 255.0), "odd": ((2, 3, 45, 70), 1.0)}


def images(tag):
    shape, rng = CASES[tag]
    a = torch.rand(shape, generator=torch.Generator().manual_seed(161)) * rng
    b = (a + 0.1 * rng * O.seeded_randn(shape, 162)).clamp(0, rng)
    return a, b, rng


class Shift(torch.nn.Module):
    def forward(self, past, fut, mask):
        return (fut * 0.9 + 0.05 * past[:, -1:],)


def loader(dev="cpu"):
    return [(O.synth_features((2, 2, 1, 32, 32), 163 + i).to(dev), O.synth_features((2, 3, 1, 32, 32), 173 + i).to(dev)) for i in range(2)]


def close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape and np.max(np.abs(a - b) / (np.abs(b) + 1e-6)) <= tol, (a, b)


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_metrics_match_reference_vectors(tag):
    a, b, rng = images(tag)
    close(OM.psnr_per_image(a, b, rng), G[f"{tag}_psnr"], 1e-6)
    close(OM.mse_per_image(a, b), G[f"{tag}_mse"], 1e-6)
    close(OM.ssim_per_image(a / rng, b / rng), G[f"{tag}_ssim"], 1e-6)
    if tag == "odd":
        close(OM.ssim_per_image(a, b, 7), G["odd_ssim_w7"], 1e-6)


def test_oracle_pred_ave_metrics_matches_reference_vector():
    got = OM.pred_ave_metrics(Shift(), loader(), lambda x, y: OM.psnr_per_image(x, y).mean(), lambda t: t * 0.5 + 0.25, 3)
    close(got, G["pred_ave_psnr"], 1e-6)


# ------------------------------------------------------------------------------------------------ GPU
DEV = "cuda:0"


@pytest.fixture(scope="module")
def M():
    import npvp_amd.metrics as M_
    return M_


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_hip_metrics_match_reference_vectors(M, tag):
    a, b, rng = images(tag)
    ad, bd = a.to(DEV), b.to(DEV)
    close(M.PSNR(ad, bd, data_range=rng, mean_flag=False).cpu(), G[f"{tag}_psnr"], 2e-5)
    close(M.MSEScore(ad, bd, mean_flag=False).cpu(), G[f"{tag}_mse"], 2e-5)
    close(M.SSIM()(ad / rng, bd / rng, mean_flag=False).cpu(), G[f"{tag}_ssim"], 2e-5)
    # mean_flag=True: python floats for PSNR / MSEScore, a 0-dim tensor for SSIM, as the reference returns
    p, m, s = M.PSNR(ad, bd, data_range=rng), M.MSEScore(ad, bd), M.SSIM()(ad / rng, bd / rng)
    assert isinstance(p, float) and isinstance(m, float) and s.dim() == 0
    close(p, G[f"{tag}_psnr_mean"], 2e-5); close(m, G[f"{tag}_mse_mean"], 2e-5); close(s.cpu(), G[f"{tag}_ssim_mean"], 2e-5)
    if tag == "odd":
        close(M.SSIM(window_size=7)(ad, bd, mean_flag=False).cpu(), G["odd_ssim_w7"], 2e-5)


@pytest.mark.gpu
def test_hip_metrics_edge_cases(M):
    # identical images: mse 0 -> PSNR = -10 log10(1e-8) = 80, SSIM = 1; a single tiny image; non-contiguous views
    a = torch.rand(2, 3, 17, 9, generator=torch.Generator().manual_seed(5)).to(DEV)
    assert M.MSEScore(a, a) == 0.0 and abs(M.PSNR(a, a) - 80.0) < 1e-4
    close(M.SSIM()(a, a, mean_flag=False).cpu(), np.ones(2), 1e-6)
    one = torch.rand(1, 1, 1, 1).to(DEV)
    close(M.SSIM()(one, one * 0.5, mean_flag=False).cpu(), OM.ssim_per_image(one.cpu(), one.cpu() * 0.5), 2e-5)
    big = torch.rand(4, 3, 40, 50, generator=torch.Generator().manual_seed(6))
    v, w = big[:, :, ::2, 5:37], big.flip(0)[:, :, ::2, 5:37]
    close(M.SSIM()(v.to(DEV), w.to(DEV), mean_flag=False).cpu(), OM.ssim_per_image(v.contiguous(), w.contiguous()), 2e-5)
    close(M.PSNR(v.to(DEV), w.to(DEV), mean_flag=False).cpu(), OM.psnr_per_image(v, w), 2e-5)
    with pytest.raises(RuntimeError):
        M.PSNR(big, big)                      # CPU tensors: no fallback
    with pytest.raises(ValueError):
        M.PSNR(a, a[:, :2])


@pytest.mark.gpu
def test_hip_metrics_full_size_properties(M):
    """one evaluation batch of 128 x 128 RGB frames (64 clips x 16 predicted steps): per-image results equal the oracle on a
    sample, SSIM is symmetric and invariant to a joint horizontal flip, PSNR follows 20 log10 of a rescaling"""
    g = torch.Generator().manual_seed(7)
    a = torch.rand(1024, 3, 128, 128, generator=g).to(DEV)
    b = (a + 0.05 * torch.randn(a.shape, generator=g).to(DEV)).clamp(0, 1)
    s = M.SSIM()(a, b, mean_flag=False)
    close(M.SSIM()(b, a, mean_flag=False).cpu(), s.cpu(), 1e-6)          # symmetric up to fma contraction
    assert torch.equal(s, M.SSIM()(a, b, mean_flag=False))               # and run-to-run identical
    close(M.SSIM()(a.flip(3), b.flip(3), mean_flag=False).cpu(), s.cpu(), 1e-5)
    idx = [0, 511, 1023]
    close(s[idx].cpu(), OM.ssim_per_image(a[idx].cpu(), b[idx].cpu()), 2e-5)
    p = M.PSNR(a, b, mean_flag=False)
    close(p[idx].cpu(), OM.psnr_per_image(a[idx].cpu(), b[idx].cpu()), 2e-5)
    close((M.PSNR(a * 255, b * 255, data_range=255.0, mean_flag=False)).cpu(), p.cpu(), 1e-5)
    assert torch.equal(p, M.PSNR(a, b, mean_flag=False))          # fixed-order reductions: run-to-run identical


@pytest.mark.gpu
def test_hip_pred_ave_metrics(M):
    got = M.pred_ave_metrics(Shift().to(DEV), loader(), M.PSNR, lambda t: t * 0.5 + 0.25, 3, device=DEV)
    close(got, G["pred_ave_psnr"], 2e-5)
    got = M.pred_ave_metrics(Shift().to(DEV), loader(), M.SSIM(), lambda t: t * 0.5 + 0.25, 3, device=DEV)
    ref = OM.pred_ave_metrics(Shift(), loader(), lambda x, y: OM.ssim_per_image(x, y).mean(), lambda t: t * 0.5 + 0.25, 3)
    close(got, ref, 2e-5)
