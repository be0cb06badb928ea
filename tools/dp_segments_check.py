"""The data-parallel step replayed as HIP-graph segments (trainer.GraphedTrainStep(grad_sync=...), trainer.StepTape) against the
EAGER data-parallel step: same model, same seeds, same batches, lr > 0, dropout and drop-path on - parameters, Adam state, the last
step's flat gradient and loss must be equal BIT FOR BIT after K optimiser steps, and the replay's host time per step is reported.

    NPVP_DP_FORCE=1 NPVP_DIST_BACKEND=nccl python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/dp_segments_check.py
    NPVP_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dp_segments_check.py

(one RCCL rank = what a one-GPU box allows; two gloo ranks on one card = the collectives really exchange data).  Workload: the c4 shard
(KITTI NPVP-D, 8 clips, 4 + 16 frames) by default; SEG_CHECK_CLIPS / SEG_CHECK_LAYERS shrink it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import npvp_amd
from npvp_amd import dp, ops
from npvp_amd.trainer import load_config

rank, world, local = dp.init_distributed()
dev = torch.device("cuda", local % torch.cuda.device_count())
torch.cuda.set_device(dev)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, To, Tp = int(os.environ.get("SEG_CHECK_CLIPS", 8)), 4, 16
K = int(os.environ.get("SEG_CHECK_STEPS", 6))
cfg = load_config(os.path.join(ROOT, "configs", "config_KITTI_VFP_NPVP-D.yaml"), B, To, Tp)
P = cfg["Predictor"]
if os.environ.get("SEG_CHECK_LAYERS"):
    P["transformer_layers"] = int(os.environ["SEG_CHECK_LAYERS"])
    P["evt_former_num_layers"] = max(1, int(os.environ["SEG_CHECK_LAYERS"]) // 2)
g = torch.Generator().manual_seed(100 + rank)
past = torch.relu(torch.randn(B, To, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
fut = torch.relu(torch.randn(B, Tp, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)


def build(name):
    ctx = ops.StepContext(name)
    with ops.use(ctx):
        torch.manual_seed(7)
        m = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev)
        dp.broadcast_module(m)
        dp.convert_sync_batchnorm(m)
        m.train()
        opt = npvp_amd.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0, ctx=ctx)
        gs = dp.GradSync(opt, bucket_bytes=16 << 20)
        ops.rng.manual_seed(1234 + rank, dev)
    return m, opt, gs


args = (P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"])
# ---- A: K eager data-parallel steps
mA, oA, gA = build("eager")
single = os.environ.get("SEG_CHECK_EAGER_SINGLE", "0") == "1"      # (diagnosis: the eager leg in the single-stream schedule too)
two = ops.WgradStream.enabled
if single:
    ops.WgradStream.enabled = False
tA = []
for i in range(K):
    t0 = time.perf_counter()
    outA = npvp_amd.predictor_train_step(mA, oA, past, fut, *args, sync=False, grad_sync=gA)
    tA.append(1000.0 * (time.perf_counter() - t0))
torch.cuda.synchronize()
ops.WgradStream.enabled = two
lossA = float(outA["loss"])
gA.remove()

# ---- B: 2 eager steps in the single-stream schedule (the recording's warm-up), then K - 2 replays of the recorded step
mB, oB, gB = build("segments")
step = npvp_amd.GraphedTrainStep(mB, oB, past, fut, *args, warmup=2, grad_sync=gB)
tape = step.tape
torch.cuda.synchronize()
host = []
for i in range(K - 2):
    t0 = time.perf_counter()
    outB = step()
    host.append(1000.0 * (time.perf_counter() - t0))
torch.cuda.synchronize()
lossB = float(outB["loss"])
n_seg, n_act = tape.segments, len(tape.items) - tape.segments

same = lambda a, b: bool(torch.equal(a, b))
res = {"params": same(oA.flat_p, oB.flat_p), "adam_m": same(oA.m, oB.m), "adam_v": same(oA.v, oB.v), "grad": same(oA.flat_g, oB.flat_g),
       "loss": lossA == lossB, "step_count": float(oA.hyper[1]) == float(oB.hyper[1]) == K}
rel = float((oA.flat_p - oB.flat_p).norm() / oA.flat_p.norm())
relg = float((oA.flat_g - oB.flat_g).norm() / oA.flat_g.norm().clamp_min(1e-30))
print(f"[dp_segments_check] rank {rank}/{world} backend={dist.get_backend()} comm={gB.comm}: {n_seg} graph segments + {n_act} eager actions, "
      f"{step.launches} library launches inside; buckets={len(gB.buckets)} launched={gB.launched}; host ms per replayed step: "
      + " ".join(f"{h:.2f}" for h in host) + "; host ms per EAGER step: " + " ".join(f"{h:.2f}" for h in tA) + f"; loss eager {lossA:.9g} / segments {lossB:.9g}; params rel diff {rel:.3e}, grad rel diff {relg:.3e}; "
      f"bitwise equal: {res}; nodes of the segments: {step.census}", flush=True)
ok = all(res.values())
t = torch.tensor([1.0 if ok else 0.0], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MIN)
assert gB.launched >= len(gB.buckets) * K, "the recorded step did not launch every bucket every replay"
if float(t) != 1.0:
    print("[dp_segments_check] FAILED: the segmented replay differs from the eager data-parallel step", flush=True)
    sys.exit(1)
if rank == 0:
    print(f"[dp_segments_check] OK host_ms_min={min(host):.2f}", flush=True)
dist.barrier()
dist.destroy_process_group()
