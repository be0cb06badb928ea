"""GPU rehearsal of the data-parallel path with W ranks (gloo or nccl): every rank runs the real HIP training
step on its shard of a global batch; rank 0 then re-runs the whole batch alone and compares the all-reduced flat
gradient, the loss and the EventEncoder BatchNorm running statistics (W ranks == 1 rank on the batch).

    NPVP_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dp_check.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import npvp_amd
from npvp_amd import dp, ops
from oracle import ops as O

rank, world, local = dp.init_distributed()
dev = torch.device("cuda", local % torch.cuda.device_count())
torch.cuda.set_device(dev)
h = torch.linspace(0, 7, 8)
To, Tp, B = 3, 4, 2 * world
args = (8, 8, To + Tp, h, h, torch.linspace(0, To - 1, To), torch.linspace(To, To + Tp - 1, Tp), 512, 'Add', 'layer', 256, 1, True, 2)
kw = dict(evt_former=True, learn_evt_token=False, evt_former_num_layers=2, dropout=0.0, drop_path=0.0)
past, fut = O.synth_features((B, To, 512, 8, 8), 1).to(dev), O.synth_features((B, Tp, 512, 8, 8), 2).to(dev)
eps = O.seeded_randn((B, 512, 8, 8), 3).to(dev)


def build(sync):
    m = npvp_amd.Predictor(*args, **kw)
    O.key_hashed_fill(m, 5)
    m = m.to(dev).train()
    if sync:
        dp.broadcast_module(m)
        dp.convert_sync_batchnorm(m)
    return m


def run(m, p, f, e, gsync):
    m.evt_prior.eps_fn = m.evt_posterior.eps_fn = lambda shape: e
    # lr = 0: the parameters stay put, so the gradients of the LATER (overlapped) steps are comparable to 1e-6 -
    # with lr > 0 AdamW's first updates are ~lr*sign(g) and amplify rounding-level gradient differences
    opt = npvp_amd.FlatAdamW(m, lr=0.0, clip_module=m.transformer)
    gs = dp.GradSync(opt.buf, bucket_bytes=8 << 20) if gsync else None
    # 3 steps: the first one teaches GradSync the per-parameter contribution counts (everything reduced in finish()),
    # the next ones take the overlapped path (buckets reduced while backward is still running)
    for _ in range(3):
        out = npvp_amd.predictor_train_step(m, opt, p, f, 0.01, 1e-6, 1.0, grad_sync=gs)
    return opt, out, gs


m = build(True)
opt, out, gs = run(m, dp.shard_batch(past, rank, world), dp.shard_batch(fut, rank, world), dp.shard_batch(eps, rank, world), True)
torch.cuda.synchronize()
loss = torch.tensor([out["loss"]], device=dev)
dist.all_reduce(loss)
if rank == 0:
    ref = build(False)
    ropt, rout, _ = run(ref, past, fut, eps, False)
    torch.cuda.synchronize()
    g, gr = opt.flat_g, ropt.flat_g
    rel = float((g - gr).norm() / gr.norm())
    rm = float((m.evt_posterior.conv1[1].running_mean - ref.evt_posterior.conv1[1].running_mean).abs().max())
    pe = float((opt.flat_p - ropt.flat_p).abs().max())
    print(f"[dp_check] world={world} backend={dist.get_backend()} buckets={len(gs.buckets)} launched={gs.launched} "
          f"grad rel-L2 {rel:.3e}  mean-loss {float(loss) / world:.6f} vs single {rout['loss']:.6f}  "
          f"BN running_mean max diff {rm:.2e}  param max diff after step {pe:.2e}", flush=True)
    assert pe == 0.0, "lr = 0: parameters must not move"
    assert rel < 1e-4 and abs(float(loss) / world - rout["loss"]) < 1e-5 * abs(rout["loss"]) + 1e-8 and rm < 1e-5, "DP != single"
    print("[dp_check] OK", flush=True)
dist.barrier()
dist.destroy_process_group()
