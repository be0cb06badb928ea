"""CPU: the oracle restatement reproduces every golden vector captured from the
imported reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import pytest
import torch

import oracle
import golden_cases as GC

TOL = 2e-5   # oracle vs reference-generated goldens, fp32 on CPU


@pytest.fixture(autouse=True)
def _threads():
    torch.set_num_threads(8)


@pytest.mark.parametrize("norm", ["layer", "instance"])
def test_posfuse(norm):
    GC.compare(GC.case_posfuse(oracle, "cpu", norm), GC.load(f"posfuse_{norm}"), TOL)


@pytest.mark.parametrize("fuse", ["Add", "SPADE"])
def test_nrmlp(fuse):
    GC.compare(GC.case_nrmlp(oracle, "cpu", fuse), GC.load(f"nrmlp_{fuse}"), TOL)


def test_slmhsa():
    GC.compare(GC.case_slmhsa(oracle, "cpu"), GC.load("slmhsa"), TOL)


def test_mlpdwbn():
    GC.compare(GC.case_mlpdwbn(oracle, "cpu"), GC.load("mlpdwbn"), TOL)


def test_block_enc_mask_quirk():
    GC.compare(GC.case_block_enc(oracle, "cpu"), GC.load("block_enc"), TOL)


def test_block_dec():
    GC.compare(GC.case_block_dec(oracle, "cpu"), GC.load("block_dec"), TOL)


def test_evtenc():
    GC.compare(GC.case_evtenc(oracle, "cpu"), GC.load("evtenc"), TOL)


def test_losses():
    GC.compare(GC.case_losses(oracle, "cpu"), GC.load("losses"), TOL)


@pytest.mark.parametrize("variant", ["D", "S"])
def test_predictor(variant):
    GC.compare(GC.case_predictor(oracle, "cpu", variant), GC.load(f"predictor_{variant}"), TOL)


@pytest.mark.parametrize("variant", ["D", "S"])
def test_train_step(variant):
    res = GC.case_train_step(oracle, "cpu", variant)
    g = GC.load(f"train_step_{variant}")
    # step 0 is tight; after the first AdamW update (~lr*sign(g)) only ~1e-3 is meaningful
    GC.compare({k: v for k, v in res.items() if k.endswith("_0")}, {k: v for k, v in g.items() if k.endswith("_0")}, TOL)
    GC.compare({k: v for k, v in res.items() if k.endswith("_1")}, {k: v for k, v in g.items() if k.endswith("_1")}, 2e-3)


def test_predictor_full_depth():
    GC.compare(GC.case_predictor_full(oracle, "cpu"), GC.load("predictor_full_D"), TOL)


def test_state_dict_keys_match_reference_census():
    """603 keys for full-depth NPVP-S (checkpoint row)."""
    h = torch.linspace(0, 7, 8)
    m = oracle.Predictor(8, 8, 20, h, h, torch.linspace(0, 9, 10), torch.linspace(10, 19, 10), 512, 'Add', 'layer',
                         256, 1, True, 8, evt_former=True, learn_evt_token=False, evt_former_num_layers=4)
    assert len(m.state_dict()) == 603
    assert m.EVT_Former.norm is m.transformer.norm
    n = sum(p.numel() for p in m.parameters())
    assert abs(n - 103.91e6) < 0.02e6


@pytest.mark.parametrize("tag", ["64", "128"])
def test_frozen_autoencoder_restatement(tag):
    """The frozen AE is stock torch on any device: its restatement (npvp_amd/models/ResNetAutoEncoder.py) is pinned
    to the reference's vectors on CPU here and on the GPU in test_hip_golden.py."""
    import npvp_amd
    GC.compare(GC.case_ae(npvp_amd, "cpu", tag), GC.load(f"ae_{tag}"), TOL)


def test_unified_random_context():
    GC.compare(GC.case_randctx(oracle, "cpu"), GC.load("predictor_randctx_S"), TOL)


def test_reset_pos_coor_fractional_times():
    GC.compare(GC.case_fractime(oracle, "cpu"), GC.load("predictor_fractime_D"), TOL)


def test_full_step_from_pixels():
    import npvp_amd
    GC.compare(GC.case_full_step(oracle, npvp_amd, "cpu"), GC.load("train_step_full_S"), TOL)


@pytest.mark.parametrize("variant", ["D", "S"])
def test_validation_step(variant):
    """LitPredictor.validation_step (ref Predictor.py:150-170): eval mode, NPVP-S with ground truth decodes from the prior."""
    import npvp_amd
    GC.compare(GC.case_val_step(oracle, npvp_amd, "cpu", variant), GC.load(f"val_step_{variant}"), TOL)
