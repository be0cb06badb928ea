"""GPU triage: where do the split-precision GEMM kernels differ from the fp32-MFMA kernel on the NRMLP backward?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import npvp_amd
from npvp_amd import ops
from oracle import ops as O
import oracle

dev = "cuda:0"
m = oracle.NRMLP(512, fuse_method="SPADE"); O.key_hashed_fill(m, 21)
coor = oracle.CoorGenerator(8, 8, 7)(torch.linspace(3, 6, 4), torch.linspace(0, 7, 8), torch.linspace(0, 7, 8))
feat = m.gaussian_mapping(coor).detach()
cot = O.seeded_randn((256, 512), 22)

def run(mode):
    ops.set_gemm_precision(mode)
    x = feat.to(dev).requires_grad_()
    acts = [x]
    h = x
    for l in m.MLP:
        if isinstance(l, torch.nn.Linear):
            h = ops.linear(h, l.weight.detach().to(dev), l.bias.detach().to(dev))
        else:
            h = torch.relu(h)
        h.retain_grad(); acts.append(h)
    b = ops.linear(h, m.mlp_beta.weight.detach().to(dev), m.mlp_beta.bias.detach().to(dev))
    g = ops.linear(h, m.mlp_gamma.weight.detach().to(dev), m.mlp_gamma.bias.detach().to(dev))
    ((b * cot.to(dev)).sum() + (g * cot.to(dev)).sum()).backward()
    return [a.detach().cpu() for a in acts], [a.grad.detach().cpu() for a in acts]

# CPU reference
x = feat.clone().requires_grad_(); acts_c = [x]; h = x
for l in m.MLP:
    h = l(h); h.retain_grad(); acts_c.append(h)
b = m.mlp_beta(h); g = m.mlp_gamma(h)
((b * cot).sum() + (g * cot).sum()).backward()
gc = [a.grad for a in acts_c]
for mode in ("f32", "bf16x6", "bf16x3"):
    a, gr = run(mode)
    print(mode)
    for i, (gg, ref) in enumerate(zip(gr, gc)):
        d = (gg - ref)
        rel = float(d.norm() / ref.norm())
        nbad = int((d.abs() > 1e-4 * ref.abs().max()).sum())
        print(f"   grad act[{i}] shape {tuple(ref.shape)} rel {rel:.3e}  #elements off by >1e-4*max: {nbad}  maxabs {float(d.abs().max()):.3e}")
    for i, (aa, ref) in enumerate(zip(a, acts_c)):
        d = aa - ref.detach()
        print(f"   act[{i}] rel {float(d.norm()/ref.detach().norm()):.3e}  sign flips at relu input: {int(((aa>0)!=(ref.detach()>0)).sum())}")
# direct dgrad test on the same operands: dy = cot, W = mlp_beta.weight
for mode in ("f32", "bf16x6"):
    ops.set_gemm_precision(mode)
    dx = ops.linear_dgrad(cot.to(dev), m.mlp_beta.weight.detach().to(dev)).cpu()
    ref = (cot.double() @ m.mlp_beta.weight.detach().double()).float()
    d = dx - ref
    print(mode, "direct dgrad rel", float(d.norm() / ref.norm()), "max", float(d.abs().max()))
