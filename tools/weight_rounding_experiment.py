"""Would a cheaper GEMM arithmetic do for north_star's 1e-3 bar?  CPU experiment on the oracle (no GPU): the full-depth NPVP-D
predictor (4 + 8 layers, 1 clip of 2 + 6 frames), forward + backward in fp32, against the same model with every GEMM WEIGHT rounded
to one fp16 (or bf16) term - what a one-plane weight operand (half the B bytes, two MFMAs per product instead of three) would
compute.  Prints rel-L2 and worst-row errors of the output, the input gradient and three weight gradients.
    python tools/weight_rounding_experiment.py          (~2 minutes on 8 cores; record: profiles/r05_weight_rounding.txt)"""
import sys, torch, copy
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle, golden_cases as GC
O=GC.O
torch.set_num_threads(8)
N,To,Tp=1,2,6
m=GC._small_predictor(oracle, False, 111, 'cpu', evt_layers=4, dec_layers=8, To=To, Tp=Tp)
m.train()
past=O.synth_features((N,To,512,8,8),112); cot=O.seeded_randn((N,Tp,512,8,8),95)
def run(mm):
    p=past.clone().requires_grad_()
    y=mm(p)
    loss=(y*y*cot).sum()
    mm.zero_grad(); loss.backward()
    sd=dict(mm.named_parameters())
    return dict(y=y.detach(), g_past=p.grad, g_lin1=sd['transformer.layers.1.linear1.weight'].grad, g_fc2=sd['transformer.layers.7.SpatialFFN1.fc2.weight'].grad if 'transformer.layers.7.SpatialFFN1.fc2.weight' in sd else None,
                g_enc_qkv=sd['EVT_Former.layers.0.temporal_MHSA.in_proj_weight'].grad)
ref=run(m)
def quantize(mm, mode):
    q=copy.deepcopy(mm)
    with torch.no_grad():
        for n,p in q.named_parameters():
            if p.dim()>=2 and ('weight' in n) and p.shape[-1]>=256 and 'norm' not in n:      # GEMM weights
                if mode=='fp16': p.copy_(p.half().float())
                elif mode=='bf16': p.copy_(p.bfloat16().float())
    return q
def rel(a,b): return float((a-b).norm()/b.norm())
def rowrel(a,b):
    a2=a.reshape(-1,a.shape[-1]); b2=b.reshape(-1,b.shape[-1])
    return float(((a2-b2).norm(dim=1)/b2.norm(dim=1).clamp_min(1e-30)).max())
for mode in ('fp16','bf16'):
    r=run(quantize(m,mode))
    print(mode, {k:(f"{rel(r[k],ref[k]):.2e}", f"{rowrel(r[k],ref[k]):.2e}") for k in ref if ref[k] is not None})
