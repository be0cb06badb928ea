"""CPU: the C-ABI library builds/loads and exports every symbol include/npvp_hip.h declares; the product
package fails loudly (no CPU fallback) and never imports the oracle."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from npvp_amd import build
    return build.build(verbose=False)


def test_header_symbols_exported(built):
    from npvp_amd._lib import lib, SIGNATURES
    hdr = open(os.path.join(ROOT, "include", "npvp_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(npvp_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    L = lib()
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in include/npvp_hip.h but not exported"
        assert n in SIGNATURES, f"{n} has no ctypes signature in npvp_amd/_lib.py"
    assert set(SIGNATURES) <= names | {"npvp_set_error"}
    assert L.npvp_version() >= 100


def test_argument_errors_do_not_need_a_gpu(built):
    """Bad arguments are rejected on the host before any launch, with a message."""
    from npvp_amd._lib import lib
    L = lib()
    rc = L.npvp_gemm_f32(1, 1, 128, 128, 33, None, 36, None, 36, None, 128, None, 0, None, None, None, 0, 0.0, 0, 1, 1,
                         None, 0, 1.0, 0, None, None, 0, None, 0, None)
    assert rc == -1 and b"multiple of 32" in L.npvp_last_error()
    assert L.npvp_layernorm_fwd(None, None, None, None, None, None, 4, 500, 1e-5, 0, None) == -1
    assert L.npvp_attn_fwd(None, 512, None, 512, None, 512, None, 512, 1, 1, 64, 8, 0, 40, 40, 8, 64, 0, 0.0, None, 0, None) == -1
    assert L.npvp_gemm_workspace_bytes(2048, 512, 20480) > 0 and L.npvp_gemm_workspace_bytes(20480, 512, 512) == 0


def test_no_cpu_fallback():
    import npvp_amd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        npvp_amd.ops.layernorm(torch.zeros(4, 512), torch.ones(512), torch.zeros(512))
    m = npvp_amd.MlpDWBN(8, 8, 512, 2048, 512)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 8, 8, 512))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "npvp_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_state_dict_keys_match_oracle():
    import npvp_amd, oracle
    h = torch.linspace(0, 7, 8)
    args = (8, 8, 20, h, h, torch.linspace(0, 9, 10), torch.linspace(10, 19, 10), 512, 'Add', 'layer', 256, 1, True, 8)
    kw = dict(evt_former=True, learn_evt_token=False, evt_former_num_layers=4)
    a, b = oracle.Predictor(*args, **kw), npvp_amd.Predictor(*args, **kw)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys()) and len(b.state_dict()) == 603
    assert all(x.shape == y.shape for x, y in zip(a.state_dict().values(), b.state_dict().values()))


def test_lr_schedule_matches_torch():
    import npvp_amd
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, 150, T_mult=1, eta_min=1e-7)
    for e in (0.0, 0.37, 12.5, 149.99, 150.0, 151.25, 449.5):
        sch.step(e)
        assert abs(opt.param_groups[0]["lr"] - npvp_amd.cosine_warm_restarts_lr(1e-4, 1e-7, 150, e)) < 1e-12
