import sys, os, hashlib; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, npvp_amd
import golden_cases as GC
from oracle import ops as O
DEV="cuda:0"
past = O.synth_features((2, 3, 512, 8, 8), 202).to(DEV); fut = O.synth_features((2, 4, 512, 8, 8), 203).to(DEV)
digs=[]
for r in range(int(os.environ.get("RUNS","12"))):
    m = GC._small_predictor(npvp_amd, True, 201, DEV, evt_layers=2, dec_layers=2, dropout=0.1, drop_path=0.1)
    m.train()
    opt = npvp_amd.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
    npvp_amd.ops.rng.manual_seed(9, torch.device(DEV)); torch.manual_seed(3)
    for s in range(3):
        npvp_amd.predictor_train_step(m, opt, past, fut, 0.01, 1e-6, 1.0, sync=False)
    torch.cuda.synchronize()
    digs.append(hashlib.sha256(opt.flat_p.cpu().numpy().tobytes()).hexdigest()[:8])
print(os.environ.get("TAG",""), "distinct digests:", len(set(digs)), digs)
