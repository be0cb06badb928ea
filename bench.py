#!/usr/bin/env python3
"""Benchmark of the NPVP Stage-2 predictor training step on MI355X (BASELINE.json metric:
"predictor train frames/sec at 1/2/4/8 MI355X; MFMA util %").

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one optimisation step of the predictor-only flavour of the reference's
training_step_no_gan: frozen-encoder feature grids in HBM -> predictor fwd (S: context +
target encoder passes, prior/posterior, decoder) -> feature-L1 + KL -> backward -> decoder-only
grad-norm clip -> AdamW -> cosine-warm-restart lr, with the reference's dropout 0.1 / drop-path 0.1
active.  Default workload = BASELINE.json configs[1]: KTH 64x64 NPVP-S, B=32 clips per GPU, To=Tp=10
(weak scaling: every rank gets its own 32 clips; gradients all-reduced over RCCL).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = forward GEMM gemm_f32_kernel<1,1>,
timed live with HIP event pairs around every launch inside the timed region) and, at N=1,
`cpu_baseline` (the CPU oracle restatement of the same step on a bounded sample, host cores stated).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

WORKLOADS = {   # name -> (config file, variant, per-GPU clips, To, Tp)
    "c1": ("config_KTH_VFP_NPVP-S.yaml", "KTH 64x64 NPVP-S", 32, 10, 10),
    "c2": ("config_BAIR_VFP_NPVP-D.yaml", "BAIR 64x64 NPVP-D", 64, 2, 28),
    "c2p": ("config_BAIR_VFP_NPVP-D.yaml", "BAIR 64x64 NPVP-D (T=20)", 64, 2, 18),
    "c0": ("config_SMMNIST_VFP_NPVP-S.yaml", "SM-MNIST 64x64 NPVP-S", 4, 5, 15),
    "c3": ("config_Cityscapes_VFP_NPVP-S.yaml", "Cityscapes 128x128 NPVP-S (per-GPU shard)", 8, 2, 12),
    "c4": ("config_KITTI_VFP_NPVP-D.yaml", "KITTI 128x128 NPVP-D (per-GPU shard)", 8, 4, 16),
}
# MI355X_MICROARCH.md dense matrix peaks: v_mfma_f32_32x32x2_f32 (fp32 in) and v_mfma_f32_32x32x16_bf16
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16x6": 2500.0, "bf16x6pc": 2500.0, "bf16x6db": 2500.0,
                    "bf16x3db": 2500.0}
KERNEL_NAME = {"f32": "gemm_f32_kernel<true,true> (forward GEMMs, v_mfma_f32_32x32x2_f32)",
               "bf16x3": "gemm_split_kernel<2,true,true> (forward GEMMs; 3 x v_mfma_f32_32x32x16_bf16 per product: "
                         "achieved = ALGORITHMIC fp32-equivalent flops, so frac <= 1/3 of the bf16 peak)",
               "bf16x6": "gemm_split_kernel<3,true,true> (forward GEMMs; 6 x v_mfma_f32_32x32x16_bf16 per product: "
                         "achieved = ALGORITHMIC fp32-equivalent flops, so frac <= 1/6 of the bf16 peak)",
               "bf16x6db": "gemm_split_db_kernel<3,true,true> (forward GEMMs; 6 x v_mfma_f32_32x32x16_bf16 per product: "
                           "achieved = ALGORITHMIC fp32-equivalent flops, so frac <= 1/6 of the bf16 peak)",
               "bf16x3db": "gemm_split_db_kernel<2,true,true> (forward GEMMs; 3 x v_mfma_f32_32x32x16_bf16 per product)",
               "bf16x6pc": "gemm_split_pc_kernel<3,true,true> (forward GEMMs; 6 x v_mfma_f32_32x32x16_bf16 per product: "
                           "achieved = ALGORITHMIC fp32-equivalent flops, so frac <= 1/6 of the bf16 peak)"}


def forward_macs_per_clip(To, Tp, stochastic):
    """Algorithmic MACs of one Predictor.forward per clip (SURVEY 8d formula; train = 3x forward)."""
    C, hid, ff = 512, 2048, 1024

    def E(T):
        return 4 * C * C + 2 * 16 * C + (2 * C * hid + 9 * hid) + 4 * C * C + 2 * T * C + 2 * C * ff

    def D(To_, Tp_):
        return E(Tp_) + (2 * C * hid + 9 * hid) + 2 * C * C + 2 * To_ * C + 2 * C * C * To_ / Tp_

    evt = 9 * 512 + 9 * 512 * 256 + 256 * 256 + 256 * 512 * (2 if stochastic else 1)
    macs = 64 * (To * 4 * E(To) + (Tp * 4 * E(Tp) if stochastic else 0) + Tp * 8 * D(To, Tp))
    macs += 64 * evt * (2 if stochastic else 1) + 64 * (To + Tp) * 393984
    return macs


def cpu_baseline(cfg_file, To, Tp, clips=2, steps=2):
    """The reference CPU path = the oracle restatement (pinned to the reference by tests/golden), timed on the
    host cores on a bounded sample of the same workload: `clips` clips of the same To/Tp, full depth."""
    import oracle
    from npvp_amd.trainer import load_config
    # the GPU box gives this job a 16-core share of a much larger host: more threads than that only thrash
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    cfg = load_config(os.path.join(ROOT, "configs", cfg_file), clips, To, Tp)
    P = cfg["Predictor"]
    m = oracle.build_predictor_from_cfg(oracle.Predictor, P, To, Tp)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=P["predictor_lr"])
    past, fut = oracle.synth_features((clips, To, 512, 8, 8), 3047), oracle.synth_features((clips, Tp, 512, 8, 8), 3048)
    oracle.predictor_train_step(m, opt, past, fut, P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"])      # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        oracle.predictor_train_step(m, opt, past, fut, P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"])
    dt = (time.perf_counter() - t0) / steps
    return {"value": clips * (To + Tp) / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{clips} clips x (To={To},Tp={Tp}), full-depth predictor train step, 1 warm-up + {steps} timed "
                      f"steps of the CPU oracle (torch {torch.__version__}, {dt:.2f} s/step)"}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c1", choices=sorted(WORKLOADS))
    ap.add_argument("--gemm", default=os.environ.get("NPVP_GEMM", "bf16x6db"), choices=["f32", "bf16x3", "bf16x6", "bf16x6pc", "bf16x6db", "bf16x3db"],
                    help="GEMM arithmetic: exact fp32 MFMA, or 2-/3-term bf16 split-precision MFMA (see npvp_amd/ops.py)")
    ap.add_argument("--flavour", default="predictor", choices=["predictor", "full"],
                    help="predictor: feature grids resident in HBM (the BASELINE metric's step); full: SURVEY 8d's second "
                         "flavour, pixels -> frozen encoder -> predictor -> frozen decoder -> image L1 (AE = stock PyTorch-ROCm)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from one captured HIP graph (N=1, predictor flavour; no per-kernel probe: roofline null)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    args = ap.parse_args()

    import npvp_amd
    from npvp_amd import dp, ops
    from npvp_amd.trainer import load_config, cosine_warm_restarts_lr

    ops.set_gemm_precision(args.gemm)
    rank, world, local = dp.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X devices"
    local = local % torch.cuda.device_count()         # NPVP_DIST_BACKEND=gloo rehearsal: ranks may share a card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg_file, name, B, To, Tp = WORKLOADS[args.workload]
    cfg = load_config(os.path.join(ROOT, "configs", cfg_file), B, To, Tp)
    P = cfg["Predictor"]
    torch.manual_seed(cfg["Env"]["rand_seed"])
    model = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev)       # dropout/drop-path 0.1 defaults
    if world > 1:
        dp.broadcast_module(model)
        dp.convert_sync_batchnorm(model)
    model.train()
    log(f"model built: {name}, {B} clips/GPU, To={To}, Tp={Tp}, world={world}")
    opt = npvp_amd.FlatAdamW(model, lr=P["predictor_lr"], clip_module=model.transformer, max_grad_norm=P["max_grad_norm"])
    gsync = dp.GradSync(opt.buf) if world > 1 else None
    ops.rng.manual_seed(cfg["Env"]["rand_seed"] + rank, dev)

    g = torch.Generator().manual_seed(cfg["Env"]["rand_seed"] + rank)
    past = torch.relu(torch.randn(B, To, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
    fut = torch.relu(torch.randn(B, Tp, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
    iters_per_epoch = 100
    if args.flavour == "full":
        D = cfg["Dataset"]
        enc, dec = npvp_amd.build_frozen_autoencoder(cfg["AE"], D["img_channels"])
        enc, dec = npvp_amd.to_device_layout(enc, dec, dev)
        S = D["img_size"]
        past_px = torch.rand(B, To, D["img_channels"], S, S, generator=g).to(dev)
        fut_px = torch.rand(B, Tp, D["img_channels"], S, S, generator=g).to(dev)
        log(f"frozen autoencoder built (ngf={cfg['AE']['ngf']}, {S}x{S}x{D['img_channels']} pixels)")

    def step(i):
        opt.set_lr(cosine_warm_restarts_lr(P["predictor_lr"], P["scheduler_eta_min"], P["scheduler_T0"], i / iters_per_epoch))
        if args.flavour == "full":
            return npvp_amd.full_train_step(model, opt, enc, dec, past_px, fut_px, P["lam_PF_L1"], P["KL_beta"],
                                            P["max_grad_norm"], sync=False, grad_sync=gsync)
        return npvp_amd.predictor_train_step(model, opt, past, fut, P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"],
                                             sync=False, grad_sync=gsync)

    if args.graph:
        assert world == 1 and args.flavour == "predictor", "--graph: single process, predictor-only flavour"
        args.no_probe = True
        opt.set_lr(P["predictor_lr"])
        gstep = npvp_amd.GraphedTrainStep(model, opt, past, fut, P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"])
        log("step captured into a HIP graph")
        step = lambda i: gstep(lr=cosine_warm_restarts_lr(P["predictor_lr"], P["scheduler_eta_min"], P["scheduler_T0"], i / iters_per_epoch))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    if not args.no_probe:
        ops.GemmProbe.arm(1, 1)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    t_host = time.perf_counter() - t0          # Python + launch time: the host must stay ahead of the GPU
    fence()
    dt = time.perf_counter() - t0
    ops.GemmProbe.disarm()
    log(f"{args.steps} timed steps: {1000.0 * dt / args.steps:.2f} ms/step (host enqueue {1000.0 * t_host / args.steps:.2f} ms/step)")
    loss = float(out["loss"])
    assert loss == loss, "loss is NaN"

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    ms = 1000.0 * dt / args.steps
    frames = world * B * (To + Tp)
    flops_step = 3 * 2 * forward_macs_per_clip(To, Tp, P["stochastic"]) * B          # per GPU, fwd+bwd

    roof = None
    if not args.no_probe:
        n, pms, pfl = ops.GemmProbe.summary()
        if n:
            ach = pfl / (pms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.gemm]
            # HBM-side bytes per launch of the same kernel from the committed PMC passes of this command
            # (profiles/r01_hbm_traffic_c1.*: separate FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 rule)
            traffic = None
            tj = os.path.join(ROOT, "profiles", "r01_hbm_traffic_c1.json")
            if args.workload == "c1" and args.gemm == "bf16x6db" and args.flavour == "predictor" and os.path.exists(tj):
                # the forward GEMM kernel has two instantiations (plain epilogue / frame-statistics epilogue): pool them
                ents = [v for k, v in json.load(open(tj)).get("pooled", {}).items()
                        if k.startswith("npvp::gemm_split_db_kernel<3, true, true, false")]
                nd = sum(e["dispatches"] for e in ents)
                traffic = round(sum(e["hbm_bytes_per_dispatch"] * e["dispatches"] for e in ents) / nd) if nd else None
            roof = {"bound": "mfma", "kernel": KERNEL_NAME[args.gemm],
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": round(ops.GemmProbe.bytes / n),
                    # NPVP-S runs its two encoder passes on two streams: those GEMMs share the device pairwise and look
                    # slower one by one (same as in a rocprofv3 trace); the decoder's launches have the device alone
                    "unshared": None if ops.GemmProbe.unshared[0] in (0, n) else {
                        "launches": ops.GemmProbe.unshared[0],
                        "achieved": round(ops.GemmProbe.unshared[2] / (ops.GemmProbe.unshared[1] * 1e-3) / 1e12, 2),
                        "avg_launch_us": round(1000.0 * ops.GemmProbe.unshared[1] / ops.GemmProbe.unshared[0], 2)},
                    "launches": n, "avg_launch_us": round(1000.0 * pms / n, 2),
                    "whole_step_tflops": round(flops_step / (ms * 1e-3) / 1e12, 2)}

    if rank == 0:
        res = {"metric": "predictor train frames/sec", "value": round(frames / (ms * 1e-3), 2), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.gemm == "f32" else
                        f"f32 ({args.gemm} split-precision MFMA, fp32 accumulate"
                        + ("; weight-gradient GEMMs bf16x3db" if (args.gemm == "bf16x6db" and ops.WGRAD_PRECISION == 5) else "") + ")",
               "data": "synthetic",
               "config": {"workload": f"{name} " + ("predictor-only train step (features in HBM)" if args.flavour == "predictor"
                                                     else "FULL train step from pixels (frozen AE enc/dec in stock PyTorch-ROCm)")
                                      + (" [HIP-graph replay]" if args.graph else "") + f", {B} clips/GPU, To={To}, "
                                      f"Tp={Tp}, dropout=drop_path=0.1, AdamW+clip",
                          "global_batch": world * B, "frames_per_clip": To + Tp, "parallelism": f"dp{world}",
                          "algorithmic_tflop_per_step_per_gpu": round(flops_step / 1e12, 3), "final_loss": round(loss, 6),
                          "host_enqueue_ms_per_step": round(1000.0 * t_host / args.steps, 2)},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline and args.flavour == "predictor":
            log("timing the CPU oracle on a bounded sample ...")
            res["cpu_baseline"] = cpu_baseline(cfg_file, To, Tp)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
