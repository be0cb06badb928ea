#!/usr/bin/env python3
"""Benchmark of the NPVP Stage-2 predictor training step on MI355X (BASELINE.json metric:
"predictor train frames/sec at 1/2/4/8 MI355X; MFMA util %").

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one optimisation step of the predictor-only flavour of the reference's
training_step_no_gan: frozen-encoder feature grids in HBM -> predictor fwd (S: context +
target encoder passes, prior/posterior, decoder) -> feature-L1 + KL -> backward -> decoder-only
grad-norm clip -> AdamW -> cosine-warm-restart lr, with the reference's dropout 0.1 / drop-path 0.1
active and every GEMM (forward, dgrad AND weight gradient) in fp32-grade split arithmetic (default f16x3:
two fp16 terms per operand, three MFMAs per product, operands scaled by their amax; GEMMs the fp16 kernels
do not take - fewer than 128 rows, no weight planes - run the three-term bf16 split).

PRIMARY workload (the `value` line) = BASELINE.json configs[2], the largest single-GPU configuration:
BAIR 64x64 NPVP-D, 64 clips per GPU, To=2, Tp=28 (T=30).  At N > 1 it is WEAK scaling: every rank gets
its own 64 clips, gradients are all-reduced over RCCL.  SECONDARY workloads (shorter runs; `secondary` is a compact
{key: [ms_per_step, frames_per_s]} map, the details go to stderr): at N = 1 every other BASELINE configuration -
north_star's "BAIR B=64 T=20" target line (c2p), configs[1] (c1), configs[0] (c0), the per-GPU shards of configs[3] /
configs[4] (c3s / c4s: 8 clips of 128x128 Cityscapes / KITTI), their FULL global batches on one GPU (c3full: 32 clips,
c4full: 64 clips - the honest denominators of a strong-scaling ratio) and the FULL step from pixels through the frozen
autoencoder on a 64x64 (full64 = c1) and a 128x128 (full128 = the c4 shard) workload; at N = 2 / 4 / 8 the BASELINE
data-parallel configuration for that GPU count (c3 on 4, c4 on 8 and - as a rehearsal - on 2; 8 clips per GPU), and in the SAME
run rank 0 alone times that configuration's global batch and its shard on one GPU, so that `scaling_dp` carries both the
strong ratio (N GPUs / the whole batch on 1) and the shard efficiency (N GPUs / N x one shard on 1).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the GEMM layout - forward or dgrad -
with the largest share of the timed region, timed live with HIP event pairs around every launch on the
stream it is launched on), `roofline_hbm` (the three kernels that lead the HBM-bound half of the step: algorithmic bytes,
event-pair time in the step, TB/s against 8 TB/s) and, at N=1, `cpu_baseline` (the CPU oracle restatement of the same step
on a bounded sample, host cores stated).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# --graph-packets fast keeps the ROCm runtime's prepared-packet replay path (NPVP_GRAPH_PACKET_CAPTURE=1).  npvp_amd/__init__.py switches it
# off by default: on ROCm 7.2 it does not execute a graph's memset nodes reliably (profiles/r06_graph_alloc_hazard.txt).  The
# package's own step contains no memset node (`graph_nodes` in the record) and `replay_check` compares the timed replays with the same
# steps taken eagerly - equal to the bit in both modes.  The choice has to be in the environment before the HIP runtime initialises, hence here.
if "--graph-packets" in sys.argv and sys.argv[sys.argv.index("--graph-packets") + 1:][:1] == ["fast"]:
    os.environ["NPVP_GRAPH_PACKET_CAPTURE"] = "1"

import torch
import torch.distributed as dist

WORKLOADS = {   # name -> (config file, variant, per-GPU clips, To, Tp)
    "c2": ("config_BAIR_VFP_NPVP-D.yaml", "BAIR 64x64 NPVP-D B=64 T=30 (BASELINE configs[2])", 64, 2, 28),
    "c2p": ("config_BAIR_VFP_NPVP-D.yaml", "BAIR 64x64 NPVP-D B=64 T=20 (north_star target line)", 64, 2, 18),
    "c1": ("config_KTH_VFP_NPVP-S.yaml", "KTH 64x64 NPVP-S B=32 T=20 (BASELINE configs[1])", 32, 10, 10),
    "c0": ("config_SMMNIST_VFP_NPVP-S.yaml", "SM-MNIST 64x64 NPVP-S B=4 T=20 (BASELINE configs[0])", 4, 5, 15),
    "c3": ("config_Cityscapes_VFP_NPVP-S.yaml", "Cityscapes 128x128 NPVP-S B=32 T=14 over 4 GPUs (BASELINE configs[3], per-GPU shard)", 8, 2, 12),
    "c4": ("config_KITTI_VFP_NPVP-D.yaml", "KITTI 128x128 NPVP-D B=64 T=20 over 8 GPUs (BASELINE configs[4], per-GPU shard)", 8, 4, 16),
    # the same two configurations with their WHOLE global batch on one GPU: the denominators of a strong-scaling ratio
    "c3full": ("config_Cityscapes_VFP_NPVP-S.yaml", "Cityscapes 128x128 NPVP-S B=32 T=14 (BASELINE configs[3]) whole batch on ONE GPU", 32, 2, 12),
    "c4full": ("config_KITTI_VFP_NPVP-D.yaml", "KITTI 128x128 NPVP-D B=64 T=20 (BASELINE configs[4]) whole batch on ONE GPU", 64, 4, 16),
}
HBM_PEAK_TBS = 8.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured by a float4 copy)
MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16); fp32-input MFMA: 157.3
LAYOUTS = {(1, 1): "forward", (1, 0): "dgrad", (0, 0): "wgrad"}


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


def host_cores():
    """the GPU box gives a 1-GPU job a 16-core share of a much larger host: more threads than that only thrash"""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))


def host_cpu():
    """(model name, cores this process may run on): the host-bound secondaries depend on it (c0 / c3s / c4s differ by 25 % from box to box)"""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return model, cores


def cpu_baseline_one(cfg_file, To, Tp, clips, steps, cores):
    import oracle
    from npvp_amd.trainer import load_config
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
    return clips * (To + Tp) / dt, dt


def cpu_baseline(primary):
    """The reference CPU path = the oracle restatement (pinned to the reference by tests/golden), timed on the host
    cores: (1) a bounded sample of the PRIMARY workload (2 clips of the same To/Tp, full depth; the step is linear in
    the clip count), (2) BASELINE configs[0] in full (SM-MNIST, 4 clips, the reference's own CPU-runnable case).
    1 warm-up + 3 timed steps each."""
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg_file, name, _, To, Tp = WORKLOADS[primary]
    v, dt = cpu_baseline_one(cfg_file, To, Tp, 2, 3, cores)
    f0, _, B0, To0, Tp0 = WORKLOADS["c0"]
    v0, dt0 = cpu_baseline_one(f0, To0, Tp0, B0, 3, cores)
    return {"value": round(v, 2), "unit": "frames/s", "cores": cores, "kind": "port", "extrapolated_from_sample": True,
            "host_logical_cpus": host_cpu()[1],
            "sample": f"2 clips x (To={To},Tp={Tp}) of the primary workload, full-depth predictor train step, 1 warm-up + 3 "
                      f"timed steps of the CPU oracle (torch {torch.__version__}, {dt:.2f} s/step)",
            "c0": {"value": round(v0, 2), "unit": "frames/s",
                   "sample": f"BASELINE configs[0] in full: {B0} clips x (To={To0},Tp={Tp0}), 1 warm-up + 3 timed steps "
                             f"({dt0:.2f} s/step)"}}


def library_hash():
    """sha256 of the loaded libnpvp_hip.so (what the committed counter passes of profiles/ are tied to)"""
    import hashlib
    from npvp_amd._lib import LIB_PATH
    return hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def run_workload(key, *a, **kw):
    """one workload = one trainer = one scheduling context (npvp_amd.sched.StepContext: dropout stream, gradient-stream queue, deferred
    reductions, data-parallel listener, range guard): nothing of it outlives the workload or is shared with the next one"""
    from npvp_amd import ops
    with ops.use(ops.StepContext(key)):
        return _run_workload(key, *a, **kw)


def _run_workload(key, steps, warmup, args, rank, world, dev, probe, flavour=None, clips=None, mode="eager", check=False):
    """Build the workload's model / optimiser / synthetic batch, run `warmup` untimed and `steps` timed steps
    (barrier + synchronize on both sides, MAX over ranks), free everything.  -> result dict.  world = 1 inside a multi-rank job =
    a SOLO run of the calling rank (no collectives, no barrier): the one-GPU denominators of `scaling_dp`.
    mode: "eager" (two streams, host enqueues every launch), "graph" (the step captured single-stream into ONE HIP graph and replayed:
    no host in the loop) or "auto" = whichever of the two is faster in a short timed trial after the warm-up - only tried when the eager
    step is plausibly host bound (host enqueue time >= 60 % of the step), single process, predictor flavour.  The record says which."""
    import npvp_amd
    from npvp_amd import dp, ops
    from npvp_amd.trainer import load_config, cosine_warm_restarts_lr

    cfg_file, name, B, To, Tp = WORKLOADS[key]
    if clips is not None:
        B = clips
    cfg = load_config(os.path.join(ROOT, "configs", cfg_file), B, To, Tp)
    P = cfg["Predictor"]
    torch.manual_seed(cfg["Env"]["rand_seed"])
    model = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev)       # dropout/drop-path 0.1 defaults
    dp_on = world > 1 or (dp.FORCE and dp.active())        # (NPVP_DP_FORCE=1: the data-parallel machinery on a group of one rank)
    _stage = os.environ.get("NPVP_DP_STAGE", "")            # (measurement: how far the data-parallel machinery is switched on)
    if _stage == "init":
        dp_on = False
    if dp_on:
        if _stage not in ("convert", "one"):
            dp.broadcast_module(model)
        if _stage != "bcast":
            dp.convert_sync_batchnorm(model)
        if _stage == "one":               # (one collective on one small tensor instead of the model broadcast)
            dist.broadcast(torch.zeros(4, device=dev), 0)
    if _stage in ("model", "bcast", "convert", "one"):
        dp_on = False
    model.train()
    log(f"[{key}] model built: {name}, {B} clips/GPU, To={To}, Tp={Tp}, world={world}")
    opt = npvp_amd.FlatAdamW(model, lr=P["predictor_lr"], clip_module=model.transformer, max_grad_norm=P["max_grad_norm"])
    gsync = dp.GradSync(opt) if dp_on else None
    ops.rng.manual_seed(cfg["Env"]["rand_seed"] + rank, dev)

    g = torch.Generator().manual_seed(cfg["Env"]["rand_seed"] + rank)
    past = torch.relu(torch.randn(B, To, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
    fut = torch.relu(torch.randn(B, Tp, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
    iters_per_epoch = 100
    full = (flavour or args.flavour) == "full"
    if full:
        D = cfg["Dataset"]
        enc, dec = npvp_amd.build_frozen_autoencoder(cfg["AE"], D["img_channels"])
        enc, dec = npvp_amd.to_device_layout(enc, dec, dev)
        S = D["img_size"]
        past_px = torch.rand(B, To, D["img_channels"], S, S, generator=g).to(dev)
        fut_px = torch.rand(B, Tp, D["img_channels"], S, S, generator=g).to(dev)
        log(f"[{key}] frozen autoencoder built (ngf={cfg['AE']['ngf']}, {S}x{S}x{D['img_channels']} pixels)")

    def step(i):
        opt.set_lr(cosine_warm_restarts_lr(P["predictor_lr"], P["scheduler_eta_min"], P["scheduler_T0"], i / iters_per_epoch))
        if full:
            return npvp_amd.full_train_step(model, opt, enc, dec, past_px, fut_px, P["lam_PF_L1"], P["KL_beta"],
                                            P["max_grad_norm"], sync=False, grad_sync=gsync)
        return npvp_amd.predictor_train_step(model, opt, past, fut, P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"],
                                             sync=False, grad_sync=gsync)

    eager_step = step
    lr_at = lambda i: cosine_warm_restarts_lr(P["predictor_lr"], P["scheduler_eta_min"], P["scheduler_T0"], i / iters_per_epoch)

    def arm_probes():
        # forward / dgrad kernels of either arithmetic (ids 2, 4, 5, 7: the candidates for the dominant kernel; 8 = the fused dgrad +
        # weight-gradient launch); --probe-all also brackets the weight-gradient and small-shape launches (their event pairs cost ~1 %)
        ops.GemmProbe.arm(None if args.probe_all else {2, 4, 5, 7, 8})

    check_state = {}
    fence_t = torch.zeros(1, dtype=torch.float32, device=dev)

    def state_tensors():
        return [opt.flat_p, opt.m, opt.v, opt.hyper, opt.ctx.rng.seed_tensor(dev)] + [b for b in model.buffers() if b.is_cuda]

    def graphed():
        opt.set_lr(P["predictor_lr"])
        if check and not check_state:
            # buffers of `replay_check` (below), allocated BEFORE the capture: nothing may be allocated between replays
            check_state["snap"] = [torch.empty_like(t) for t in state_tensors()]
            check_state["result"] = torch.empty_like(opt.flat_p)
        if probe:
            # the capture carries the probes: library events recorded inside it are external event-record nodes that every replay
            # stamps again (sched.ProbeEvent), so the roofline figures of a replayed step are measured IN the timed replays - read after
            # the last one, i.e. over the launches of the last timed step
            arm_probes()
            ops.HbmProbe.armed, ops.HbmProbe.records = True, []
        two = args.graph_streams == 2                # (measurement: the gradient stream inside the capture, profiles/r05_graph_modes.txt)
        gs = npvp_amd.GraphedTrainStep(model, opt, past, fut, P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"], single_stream=not two,
                                       grad_sync=gsync)
        if gs.tape is not None:
            log(f"[{key}] data-parallel step recorded as {gs.tape.segments} HIP-graph segments + {len(gs.tape.items) - gs.tape.segments} "
                f"eager collectives ({gs.launches} library launches inside the segments)")
        else:
            log(f"[{key}] step captured into a HIP graph ({'two streams' if two else 'single stream'}, {gs.launches} library launches)")
        return gs, (lambda i: gs(lr=lr_at(i)))

    TRIAL_STEPS = args.trial_steps

    def trial(fn, n=TRIAL_STEPS):
        """-> (ms per step, fastest host enqueue ms) over n steps measured like the timed region: three untimed steps (a fresh graph's
        first replays are slow: they fault its private pool in), a device synchronisation, n steps enqueued back to back, a device
        synchronisation"""
        for i in range(3):
            fn(0)
        torch.cuda.synchronize()
        t0, hmin = time.perf_counter(), 1e9
        for i in range(n):
            ti = time.perf_counter()
            fn(1 + i)
            hmin = min(hmin, time.perf_counter() - ti)
        torch.cuda.synchronize()
        return 1000.0 * (time.perf_counter() - t0) / n, 1000.0 * hmin

    def prefer_replay(e_ms, e_host, g_ms):
        """The rule.  A step whose host enqueue time is >= 60 % of its duration runs at the mercy of the host: its
        trial can look fine and the timed region - a noisy neighbour on the host's cores later - 30 % worse (driver record r05: eager
        trial 30.2 ms, timed eager steps 41.6 ms, replay 31.7).  A replay does not depend on the host, so it is taken unless the eager
        step beat it by more than 5 % over the trial's 10 steps."""
        return g_ms is not None and e_host >= 0.6 * e_ms and not (e_ms < 0.95 * g_ms)

    if args.graph:
        mode = "graph"
    used, trial_ms = "eager", None
    ops.FusedLinearBwd.with_gradient_stream = False
    if mode in ("graph", "auto") and full:
        assert mode == "auto", "--graph: predictor-only flavour"
        mode = "eager"
    if dp_on and not full:
        # Data parallel.  The collectives are never captured into a graph, but the step can still be taken off the host: recorded as a
        # chain of single-stream graph SEGMENTS cut at every collective (trainer.StepTape), replayed with the collectives issued eagerly
        # between them (~30 host touches instead of ~1 000 launches).  Candidates, each timed over the same 10 steps on every rank, every
        # decision from the MAX over the ranks so that all ranks take the same one:
        #   eager        two streams, two launches per layer backward
        #   eager_fused  (if the eager step is host bound) dgrad + weight gradient of a layer as ONE launch beside the gradient stream
        #   segments     (host-bound or short steps; --dp-graph never skips it, always forces it) the segmented replay
        for i in range(warmup):
            eager_step(i)

        def agreed(*v, op=dist.ReduceOp.MAX):
            t = torch.tensor(v, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=op)
            return t.tolist()

        e_ms, e_host = agreed(*trial(eager_step))
        trial_ms = {"eager": round(e_ms, 2), "eager_host": round(e_host, 2)}
        host_bound = e_host >= 0.6 * e_ms
        if args.dp_fused_trial == "always" or (args.dp_fused_trial == "auto" and host_bound):
            ops.FusedLinearBwd.with_gradient_stream = True
            f_ms, f_host = agreed(*trial(eager_step))
            trial_ms["eager_fused"] = round(f_ms, 2)
            if f_ms < e_ms:
                used, e_ms, e_host = "eager_fused", f_ms, f_host
            else:
                ops.FusedLinearBwd.with_gradient_stream = False
        if mode == "graph" or args.dp_graph == "always" or (args.dp_graph == "auto" and (host_bound or e_ms < 60.0)):
            # (auto: host-bound or short steps - the shards; c2's segmented replay loses to its eager step in the safe replay mode, 238 vs 231 ms)
            fused_was = ops.FusedLinearBwd.with_gradient_stream
            try:
                gstep, gfn = graphed()
                ok = 1.0
            except Exception as e:          # (a capture that fails on some rank must not take the job down - nor leave the ranks in disagreement)
                log(f"[{key}] segmented capture failed ({type(e).__name__}: {e}); staying eager")
                gstep = gfn = None
                ok = 0.0
                ops.GemmProbe.disarm(); ops.HbmProbe.armed = False
                torch.cuda.synchronize()
            (ok,) = agreed(ok, op=dist.ReduceOp.MIN)
            if ok:
                g_ms, g_host = agreed(*trial(gfn))
                trial_ms["segments"], trial_ms["segments_host"] = round(g_ms, 2), round(g_host, 2)
                forced = mode == "graph" or args.dp_graph == "always"
                if forced or prefer_replay(e_ms, e_host, g_ms) or g_ms < e_ms:
                    step, used = gfn, "graph_segments"
            else:
                trial_ms["segments"] = None
                gstep = gfn = None
            if used != "graph_segments":
                gstep = gfn = None
                ops.GemmProbe.disarm(); ops.HbmProbe.armed = False
                gc.collect()
                torch.cuda.empty_cache()
                ops.FusedLinearBwd.with_gradient_stream = fused_was
                gsync.relearn()             # (the recording's warm-up taught GradSync the single-stream schedule's counts)
        mode = "eager"
        log(f"[{key}] mode trial: {trial_ms} -> {used}")
    if mode == "graph":
        gstep, step = graphed()
        used = "graph"
    elif mode == "auto":
        for i in range(warmup):
            eager_step(i)
        e_ms, e_host = trial(eager_step)
        trial_ms = {"eager": round(e_ms, 2), "eager_host": round(e_host, 2)}
        if e_host >= 0.6 * e_ms or e_ms < 60.0 or args.graph_packets == "fast":
            # the replay is tried for every step that is host bound or short (the 8-clip shards, c0: a few seconds of trial); the large
            # workloads' replays lose to their eager two-stream steps in the safe replay mode (c2 236 - 242 against 229 - 236 ms, c1 92 - 96
            # against 84 - 86: DESIGN section 5), and a capture of c2 next to its eager pool is 130 GiB of allocator churn for nothing
            try:
                gstep, gfn = graphed()
                g_ms, g_host = trial(gfn)
            except Exception as e:          # (a capture that fails on some box must not take the whole record down: stay eager)
                log(f"[{key}] graph capture / replay failed ({type(e).__name__}: {e}); staying eager")
                trial_ms["graph"] = None
                gstep = gfn = None
                ops.GemmProbe.disarm(); ops.HbmProbe.armed = False
                torch.cuda.synchronize()
            else:
                trial_ms["graph"], trial_ms["graph_host"] = round(g_ms, 2), round(g_host, 2)
                if prefer_replay(e_ms, e_host, g_ms) or g_ms < e_ms:
                    step, used = gfn, "graph"
                else:
                    del gstep, gfn
                    ops.GemmProbe.disarm(); ops.HbmProbe.armed = False
                    gc.collect()
                    torch.cuda.empty_cache()    # (the graph's private pool goes back to the device: the eager warm-up below grows its own)
        log(f"[{key}] mode trial: {trial_ms} -> {used}")

    def fence():
        torch.cuda.synchronize()
        if dp_on:
            dist.all_reduce(fence_t)        # (a barrier on a tensor that exists since before any capture: dist.barrier() allocates one)
        torch.cuda.synchronize()

    # Warm-up WITHOUT a synchronisation per step: the host must run ahead of the GPU here as it does in the timed region.  Blocks that
    # the gradient stream still uses are not reusable until its events complete, so a host that is a step or two ahead needs a
    # larger pool than a synchronised one - and with a sync after every warm-up step that pool was first grown INSIDE the timed
    # steps: each new multi-GB hipMalloc blocked the host for 1.5 - 3 s on a box whose memory an earlier process had just released
    # (back-to-back runs: the driver's N = 1, 2, 4, 8 sequence).  The pool also gets slack (allocated and returned to the cache).
    for i in range(warmup):
        step(i)
        log(f"[{key}] warm-up step {i} enqueued")
    torch.cuda.synchronize()
    # (never more than the device has free: an allocation the device cannot serve makes the caching allocator release its WHOLE pool
    #  and retry - the pool that the warm-up had just grown: the timed steps then grew it again with fresh allocations, each of
    #  which can block for seconds on a box whose memory an earlier process has just released; profiles/r05_back_to_back.txt)
    free_b, _ = torch.cuda.mem_get_info(dev)
    # (an eager step's business only: a replayed step allocates nothing - and must not find freshly mapped memory being written
    #  between its replays, npvp_amd/__init__.py)
    slack_n = 1 if used in ("graph", "graph_segments") else max(1, min(int(0.2 * torch.cuda.memory_reserved(dev)), free_b - (8 << 30)))
    slack = torch.empty(slack_n, dtype=torch.uint8, device=dev) if slack_n > 1 else torch.empty(0, dtype=torch.uint8, device=dev)
    # ... and touched before it goes back to the cache (its first use happens here, not in a timed step)
    t_sl = time.perf_counter()
    slack.zero_()
    torch.cuda.synchronize()
    log(f"[{key}] {slack.numel() / 2 ** 30:.1f} GiB of slack allocated and touched in {time.perf_counter() - t_sl:.2f} s")
    del slack
    free_b, total_b = torch.cuda.mem_get_info(dev)
    log(f"[{key}] before the timed steps: reserved {torch.cuda.memory_reserved(dev) / 2 ** 30:.1f} GiB, allocated "
        f"{torch.cuda.memory_allocated(dev) / 2 ** 30:.1f} GiB, device free {free_b / 2 ** 30:.1f} of {total_b / 2 ** 30:.1f} GiB, "
        f"allocator retries so far {torch.cuda.memory_stats(dev).get('num_alloc_retries', 0)}")
    replayed = used in ("graph", "graph_segments")
    if probe and not replayed:              # (a replayed step carries its probes since its capture)
        arm_probes()
    if check and replayed and check_state:
        for a_, b_ in zip(check_state["snap"], state_tensors()):      # (`replay_check`: the state the timed replays start from)
            a_.copy_(b_)
        check_state["rng"] = torch.cuda.get_rng_state(dev)            # (NPVP-S: the position of torch's generator - host state, no device work)
    fence()
    L0 = npvp_amd._lib.lib().npvp_launch_count()
    t0 = time.perf_counter()
    host_max, host_min, host_all = 0.0, 1e9, []
    for i in range(steps):
        ti = time.perf_counter()
        out = step(warmup + i)
        host_all.append(time.perf_counter() - ti)
        host_max, host_min = max(host_max, host_all[-1]), min(host_min, host_all[-1])
    # Python + launch time of one step.  The FASTEST step is the host's own cost (the first one after the fence finds an idle
    # GPU); the others also contain the time the launch queue makes the host wait for the GPU once it is a few thousand
    # launches ahead, i.e. they converge to the GPU's step time and say nothing about the host.
    t_host = host_min * steps
    launches = (npvp_amd._lib.lib().npvp_launch_count() - L0) / steps       # (library launches; a replayed graph enqueues none: taken from its capture)
    graph_nodes = None
    if used in ("graph", "graph_segments"):
        launches = gstep.launches
        graph_nodes = {k: v for k, v in gstep.census.items() if k != "memset_bytes"}        # (node census of the replayed step: no "memset")
    fence()
    dt = time.perf_counter() - t0
    if check and replayed and check_state:
        check_state["result"].copy_(opt.flat_p)
    log(f"[{key}] after the timed steps: reserved {torch.cuda.memory_reserved(dev) / 2 ** 30:.1f} GiB, allocator retries "
        f"{torch.cuda.memory_stats(dev).get('num_alloc_retries', 0)}")
    ops.GemmProbe.disarm()
    ops.HbmProbe.armed = False
    loss = float(out["loss"])
    assert loss == loss, "loss is NaN"

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dp_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    ms = 1000.0 * dt / steps
    frames = world * B * (To + Tp)
    flops_step = 3 * 2 * forward_macs_per_clip(To, Tp, P["stochastic"]) * B          # per GPU, fwd+bwd
    log(f"[{key}] {steps} timed steps: {ms:.2f} ms/step (host enqueue {1000.0 * t_host / steps:.2f} ms/step, slowest {1000.0 * host_max:.2f}), "
        f"{frames / (ms * 1e-3):.0f} frames/s; host ms per step: " + " ".join(f"{1000.0 * h:.0f}" for h in host_all))

    roof, roof_hbm = None, None
    if probe:
        hbm_steps = 1                       # (a replayed step: the launches of the last timed replay)
        if not replayed:
            # the HBM-bound family, timed in two EXTRA steps after the clock stopped (the event packets fence their neighbours)
            ops.HbmProbe.armed, ops.HbmProbe.records = True, []
            hbm_steps = 2
            for i in range(hbm_steps):
                step(warmup + steps + i)
            fence()
            ops.HbmProbe.armed = False
        roof_hbm = []
        for kname, (n, pms, pby) in sorted(ops.HbmProbe.summary().items(), key=lambda kv: -kv[1][1]):
            tbs = pby / (pms * 1e-3) / 1e12
            roof_hbm.append({"kernel": kname, "bound": "hbm", "launches_per_step": n // hbm_steps, "algorithmic_bytes_per_launch": round(pby / n),
                             "avg_launch_us": round(1000.0 * pms / n, 1), "achieved": round(tbs, 3), "peak": HBM_PEAK_TBS, "unit": "TB/s",
                             "frac": round(tbs / HBM_PEAK_TBS, 3)})
        ops.HbmProbe.records = []
    if probe:
        per = ops.GemmProbe.summary()           # {((a_kc,b_kc), kernel id): (launches, ms, flops, bytes)}
        groups = {}            # "<layout>:<kernel id>" -> [launches, average us, algorithmic TFLOP/s]
        for (lay, kid), (n, pms, pfl, pby) in per.items():
            groups[f"{LAYOUTS[lay]}:{kid}"] = [n, round(1000.0 * pms / n, 1), round(pfl / (pms * 1e-3) / 1e12, 1)]
            log(f"[{key}] probe {LAYOUTS[lay]}:{ops.GemmProbe.KERNELS[kid]}: {n} launches, {1000.0 * pms / n:.1f} us average, "
                f"{pfl / (pms * 1e-3) / 1e12:.1f} TF algorithmic, {pby / n / 1e6:.0f} MB algorithmic per launch, {pms:.1f} ms in all")
        # dominant kernel = the one with the largest total time among the kernels of the critical path (forward and dgrad
        # launches: same stream as the step).  The weight-gradient GEMMs run on the low-priority gradient stream UNDER other
        # kernels, so their event-pair durations include time-sharing; they are reported but not used for the fraction.
        crit = {}
        for (lay, kid), (n, pms, pfl, pby) in per.items():
            if LAYOUTS[lay] != "wgrad":
                a = crit.setdefault(kid, [0, 0.0, 0.0, 0.0])
                a[0] += n; a[1] += pms; a[2] += pfl; a[3] += pby
        if crit:
            kid = max(crit, key=lambda k: crit[k][1])
            n, pms, pfl, pby = crit[kid]
            kname = ops.GemmProbe.KERNELS[kid]
            ach = pfl / (pms * 1e-3) / 1e12
            # Figures that cannot be measured inside the timed process (PMC passes serialise the dispatches): read from the committed
            # counter passes of this very command - and ONLY when those passes ran on the library that is loaded now (the profile
            # records its sha256).  A kernel change without new passes shows as null + "stale".
            lib_hash = library_hash()
            traffic, traffic_src = None, None

            def committed(pattern):
                """-> (record, file) of the newest round's profile whose library hash is the loaded library's, else (None, newest file)"""
                newest = None
                for r_ in range(9, 2, -1):
                    q = os.path.join(ROOT, "profiles", pattern.format(r=r_))
                    if os.path.exists(q):
                        newest = newest or q
                        rec = json.load(open(q))
                        if rec.get("lib_sha256") == lib_hash:
                            return rec, q
                return None, newest

            if not full:
                rec, q = committed("r0{r}_hbm_traffic_" + key + ".json")
                if rec is not None:
                    ent = rec.get("pooled", {}).get(kname)
                    traffic, traffic_src = (round(ent["hbm_bytes_per_dispatch"]) if ent else None), os.path.relpath(q, ROOT)
                elif q is not None:
                    traffic_src = {"stale": True, "file": os.path.relpath(q, ROOT), "why": "measured on another build of libnpvp_hip.so"}
            f16 = kid in (5, 6, 7, 8)
            mfmas = 1 if kid == 0 else (3 if f16 else 6)
            roof = {"bound": "mfma", "kernel": kname + " (forward + dgrad launches)",
                    "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "mfma_per_product": mfmas,
                    "mfma_pipe_busy_frac": round(mfmas * ach / MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(pby / n), "launches": n,
                    "avg_launch_us": round(1000.0 * pms / n, 2), "by_layout_and_kernel_id": groups,
                    "note": f"achieved = algorithmic 2MNK flops / event-pair time of every launch; {mfmas} MFMAs per fp32-grade product, "
                            f"so frac <= 1/{mfmas} by construction",
                    "sample": ("the launches of the LAST timed replay (the event pairs are nodes of the replayed HIP graph; every replay "
                               "stamps them again)" if replayed else "every launch of the timed steps"),
                    "whole_step_tflops": round(flops_step / (ms * 1e-3) / 1e12, 2)}
            # BASELINE's "MFMA util %" as a MEASURED step-level number: sum of SQ_VALU_MFMA_BUSY_CYCLES over every dispatch of the timed
            # steps / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8), from a committed rocprofv3 --pmc pass over this very command
            # (tools/profile_r05.sh, tools/rocpd_mfma_util.py; counter passes cannot run inside the timed process - like `traffic`)
            for tag, wk in (("mfma_util_step", key), ("mfma_util_step_c2p", "c2p")):
                if full or not (tag == "mfma_util_step" or key == "c2"):
                    continue
                u, q = committed("r0{r}_mfma_util_" + wk + ".json")
                if u is not None:
                    roof[tag] = {"value": u["mfma_util_step"], "vs_unprofiled_step": u.get("util_vs_unprofiled_step"),
                                 "clock_ghz": u["effective_clock_ghz"], "source": os.path.relpath(q, ROOT).replace(".json", ".md")}
                elif q is not None:
                    roof[tag] = {"value": None, "stale": True, "file": os.path.relpath(q, ROOT), "why": "measured on another build of libnpvp_hip.so"}

    replay_check = None
    if check and replayed and check_state:
        # Did the TIMED replays compute what eager steps compute?  The state the timed region started from (saved above) is put
        # back and the same K steps are taken eagerly (two streams, every launch enqueued by the host): same batch, same learning
        # rates, same dropout stream (the seed lives in device memory and is part of the state), same reparameterisation noise
        # (NPVP-S: torch's generator is put back to where the timed region found it).
        st, snap, pa = state_tensors(), check_state["snap"], check_state["result"]
        for a_, b_ in zip(snap, st):
            b_.copy_(a_)
        torch.cuda.set_rng_state(check_state["rng"], dev)             # (a replay advances the generator exactly as the eager step does)
        ops.WeightPlanes.refresh_all(opt.flat_p)          # (the planes mirror the parameters that were just put back)
        if gsync is not None:
            gsync.relearn()
        for i in range(steps):
            out_e = eager_step(warmup + i)
        fence()
        loss_e = float(out_e["loss"])
        rel = float((opt.flat_p - pa).norm() / pa.norm())
        upd = float((pa - snap[0]).norm() / pa.norm())
        replay_check = {"steps": steps, "params_rel_l2_replay_vs_eager": rel, "update_rel_l2_over_the_steps": upd,
                        "loss_replay": loss, "loss_eager": loss_e, "bit_equal": bool(rel == 0.0 and loss == loss_e),
                        "ok": bool(rel < 5e-5 and abs(loss - loss_e) <= 2e-4 * abs(loss_e)),
                        "packet_capture": npvp_amd.graph_packet_capture(),
                        "what": "the timed replays against the same steps taken eagerly from the same saved state, dropout seed and "
                                "generator position (expected: 0.0 = equal to the bit)"}
        log(f"[{key}] replay check: " + json.dumps(replay_check))
        check_state.clear()

    peak_gb = torch.cuda.max_memory_allocated(dev) / 2.0 ** 30
    log(f"[{key}] peak device memory {peak_gb:.1f} GiB")
    torch.cuda.reset_peak_memory_stats(dev)
    events = ops.RangeGuard.poll(dev)           # weight-gradient launches that met a feature 2^18 below its tensor's bound (0 expected)
    res = {"key": key, "name": name, "B": B, "To": To, "Tp": Tp, "ms": ms, "frames_per_s": frames / (ms * 1e-3), "peak_gb": peak_gb,
           "loss": loss, "host_ms": 1000.0 * t_host / steps, "flops_step": flops_step, "roof": roof, "roof_hbm": roof_hbm, "steps": steps,
           "warmup": warmup, "range_events": events, "launches": launches, "mode": used, "mode_trial": trial_ms, "replay_check": replay_check,
           "graph_nodes": graph_nodes}
    if gsync is not None:
        # how many ranks REALLY reduce together: the mean over the ranks of (rank + 1), through the very exchange the gradient buckets
        # take (ProcessGroup or the library's npvp_dp_*), must be (world + 1) / 2 - one number the driver can check against --gpus
        chk = torch.full((4,), float(dist.get_rank() + 1), dtype=torch.float32, device=dev)
        if gsync.comm == "c":
            L_ = npvp_amd._lib.lib()
            npvp_amd._lib.check(L_.npvp_dp_allreduce_async(chk.data_ptr(), chk.numel(), torch.cuda.current_stream().cuda_stream), "npvp_dp_allreduce_async")
            lib_world = L_.npvp_dp_world()
        else:
            dist.all_reduce(chk)
            chk /= dist.get_world_size()
            lib_world = None
        ranks_seen = round(2.0 * float(chk[0]) - 1.0, 3)
        assert ranks_seen == dist.get_world_size() == world or (world == 1 and dp.FORCE), f"all-reduce spans {ranks_seen} ranks, WORLD_SIZE={world}"
        assert lib_world in (None, dist.get_world_size()), f"npvp_dp_world() = {lib_world}, torch world size {dist.get_world_size()}"
        res["dp"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_in_allreduce": ranks_seen,
                     "npvp_dp_world": lib_world, "comm": gsync.comm, "buckets": len(gsync.buckets), "last_bucket_mb": round((gsync.buckets[0]["hi"] - gsync.buckets[0]["lo"]) * 4 / 2 ** 20, 1),
                     "allreduces_launched": gsync.launched, "exposed_allreduce_ms_per_step": round(gsync.exposed_ms(), 3)}
        log(f"[{key}] data parallel: " + json.dumps(res["dp"]))
        gsync.remove()
    del model, opt, gsync, past, fut, out, step, eager_step
    gstep = gfn = None
    gc.collect()
    # The cache goes back to the driver: with c2's ~100 GB of cached blocks in the allocator the host-bound 8-clip workloads that
    # follow ran 10 % slower (every torch.empty walks a long free list).  The multi-second stalls this once caused inside a LATER
    # workload's timed steps (its pool grew there, and hipMalloc right after a large hipFree is slow) are closed at the other end:
    # the warm-up now grows the pool, with slack, before the clock starts (see run_workload).
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="primary workload (default c2 = BASELINE configs[2], the largest single-GPU configuration)")
    ap.add_argument("--gemm", default=os.environ.get("NPVP_GEMM", "f16x3"), choices=["f32", "bf16x6", "f16x3"],
                    help="GEMM arithmetic: f16x3 = two-term fp16 split with amax-scaled operands, 3 MFMAs per product (fp32-grade, "
                         "default; small shapes run as bf16x6); bf16x6 = three-term bf16 split, 6 MFMAs per product (fp32-grade); "
                         "f32 = exact fp32-input MFMA (parity triage)")
    ap.add_argument("--flavour", default="predictor", choices=["predictor", "full"],
                    help="predictor: feature grids resident in HBM (the BASELINE metric's step); full: SURVEY 8d's second "
                         "flavour, pixels -> frozen encoder -> predictor -> frozen decoder -> image L1 (AE = stock PyTorch-ROCm)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from one captured HIP graph (N=1, predictor flavour; no per-kernel probe: roofline null)")
    ap.add_argument("--dp-fused-trial", default="auto", choices=["auto", "always", "never"],
                    help="data parallel: when to time the one-launch-per-layer-backward eager step against the two-launch one "
                         "(auto: if the slowest rank's host enqueue time is >= 60 %% of its step; always: the rehearsals)")
    ap.add_argument("--trial-steps", type=int, default=10, help="timed steps per candidate of a mode trial (the rehearsals on a shared card use fewer)")
    ap.add_argument("--dp-graph", default="auto", choices=["auto", "always", "never"],
                    help="data parallel: when to record the step as HIP-graph segments with the collectives issued eagerly between them "
                         "(trainer.StepTape): auto = for host-bound or short steps, time it against the eager step and take the replay unless "
                         "eager is faster (by > 5 %% when host bound); always = take it; never = eager only")
    ap.add_argument("--graph-packets", default="safe", choices=["fast", "safe"],
                    help="how the ROCm runtime replays a HIP graph: safe (default) = DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, the package's default; "
                         "fast = the runtime's prepared-packet path (less host time per replay; refuses a step with memset nodes, which "
                         "that path does not execute reliably on ROCm 7.2).  `replay_check` in the record compares the replays with eager steps either way")
    ap.add_argument("--graph-streams", type=int, default=1, choices=[1, 2],
                    help="streams inside a captured step: 1 (default: one chain of nodes, what replays fast) or 2 (measurement only)")
    ap.add_argument("--mode", default="auto", choices=["eager", "graph", "auto"],
                    help="primary workload: eager (two streams, the host enqueues every launch), graph (the step replayed from one "
                         "single-stream HIP graph; the roofline probes ride inside it) or auto (default) = a timed trial of both after "
                         "the warm-up: the replay unless the eager step beat it by more than 5 %% over 10 steps (host-bound steps) / is "
                         "simply faster")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--probe-all", action="store_true", help="bracket every GEMM launch with events, not only forward / dgrad")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads")
    args = ap.parse_args()

    import npvp_amd
    from npvp_amd import dp, ops

    ops.set_gemm_precision(args.gemm)
    rank, world, local = dp.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X devices"
    local = local % torch.cuda.device_count()         # NPVP_DIST_BACKEND=gloo rehearsal: ranks may share a card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    main_res = run_workload(args.workload, args.steps, args.warmup, args, rank, world, dev, probe=not args.no_probe, mode=args.mode, check=True)

    secondary, detail, scaling_dp, modes, strong = {}, {}, None, {}, None
    sec_ok = not args.no_secondary and not args.graph and args.flavour == "predictor"
    if sec_ok and (args.workload == "c2" or world > 1):
        sec_steps, sec_warm = max(4, args.steps // 2), min(3, args.warmup) or 1

        def note(name, r, flav=None, solo=False):
            # [ms per step, frames/s, host enqueue ms per step, library launches per step, whole-step TFLOP/s]
            secondary[name] = [round(r["ms"], 2), round(r["frames_per_s"], 1), round(r["host_ms"], 1), round(r["launches"]),
                               round(r["flops_step"] / (r["ms"] * 1e-3) / 1e12, 1)]
            if r["mode"] != "eager":
                modes[name] = r["mode"] + ("" if r.get("replay_check") is None else (" (replay_check ok)" if r["replay_check"]["ok"] else " (replay_check FAILED)"))
            detail[name] = {"workload": r["name"] + (" - FULL step from pixels through the frozen autoencoder" if flav == "full" else
                                                     " - predictor-only step") + (" [rank 0 alone]" if solo else ""),
                            "clips_per_gpu": r["B"], "To": r["To"], "Tp": r["Tp"], "frames_per_s": round(r["frames_per_s"], 2),
                            "ms_per_step": round(r["ms"], 3), "steps": r["steps"], "warmup": r["warmup"],
                            "whole_step_tflops_per_gpu": round(r["flops_step"] / (r["ms"] * 1e-3) / 1e12, 2),
                            "host_enqueue_ms_per_step": round(r["host_ms"], 2), "peak_device_memory_gib": round(r["peak_gb"], 1),
                            "launches_per_step": round(r["launches"]), "mode": r["mode"], "mode_trial_ms": r["mode_trial"]}
            log(f"secondary {name}: " + json.dumps(detail[name]))

        if world == 1:
            # (key in the JSON, workload, flavour): every BASELINE configuration the primary line does not cover
            for name, k, flav in [("c2p", "c2p", None), ("c1", "c1", None), ("c0", "c0", None), ("c3s", "c3", None), ("c4s", "c4", None),
                                  ("c3full", "c3full", None), ("c4full", "c4full", None), ("full64", "c1", "full"), ("full128", "c4", "full")]:
                note(name, run_workload(k, sec_steps, sec_warm, args, rank, world, dev, probe=False, flavour=flav,
                                        mode="eager" if flav == "full" else "auto", check=True), flav)
            # what N GPUs can at best make of these shards: N x shard / whole batch on one GPU (the step is not linear in the clip
            # count - an 8-clip shard is bound by kernel count)
            scaling_dp = {"c3_on_4_upper_bound": round(4 * secondary["c3s"][1] / secondary["c3full"][1], 2),
                          "c4_on_8_upper_bound": round(8 * secondary["c4s"][1] / secondary["c4full"][1], 2),
                          "note": "N x (8-clip shard on 1 GPU) / (whole global batch on 1 GPU), frames/s: the strong-scaling ratio a free "
                                  "all-reduce would give; the measured ratio is in the N = 4 / 8 records"}
        else:
            # the BASELINE data-parallel configuration for this GPU count on all ranks, then - rank 0 alone, the others waiting at the
            # barrier below - the same configuration's whole global batch and its 8-clip shard on ONE GPU
            k = {4: "c3"}.get(world, "c4")
            # (the primary run IS that configuration when the caller asked for it - the 2-rank rehearsal does: not run twice)
            rd = main_res if args.workload == k else run_workload(k, sec_steps, sec_warm, args, rank, world, dev, probe=False, check=True)
            note(k, rd)
            if rank == 0:
                rs = run_workload(k, sec_steps, sec_warm, args, 0, 1, dev, probe=False)
                rf = run_workload(k, sec_steps, sec_warm, args, 0, 1, dev, probe=False, clips=8 * world)
                note(k + "s@1", rs, solo=True); note(k + "full@1", rf, solo=True)
                scaling_dp = {"config": f"{k}: {8 * world} clips over {world} GPUs (8 per GPU)",
                              "dp_frames_per_s": round(rd["frames_per_s"], 1), "whole_batch_on_1_gpu_frames_per_s": round(rf["frames_per_s"], 1),
                              "shard_on_1_gpu_frames_per_s": round(rs["frames_per_s"], 1),
                              "strong_ratio": round(rd["frames_per_s"] / rf["frames_per_s"], 3),
                              "shard_efficiency": round(rd["frames_per_s"] / (world * rs["frames_per_s"]), 3),
                              "note": "strong_ratio = N GPUs / the whole batch on one GPU; shard_efficiency = N GPUs / (N x one shard on one GPU)"}
                # the honest multi-GPU figure, top level: `value` above is WEAK scaling (64 clips per GPU at c2, reads
                # ~N x on any machine); this is BASELINE's data-parallel configuration for this GPU count - a FIXED global batch cut
                # into 8-clip shards - against the same global batch on one GPU, both measured in this run
                strong = {"config": WORKLOADS[k][1].replace(" (BASELINE", f" - here {8 * world} clips over {world} GPUs (BASELINE"),
                          "frames_per_s": round(rd["frames_per_s"], 1), "ms_per_step": round(rd["ms"], 3),
                          "whole_batch_on_1_gpu_frames_per_s": round(rf["frames_per_s"], 1), "whole_batch_on_1_gpu_ms_per_step": round(rf["ms"], 3),
                          "speedup_vs_1_gpu": round(rd["frames_per_s"] / rf["frames_per_s"], 3), "n_gpus": world,
                          "ideal": world, "mode": rd["mode"], "host_enqueue_ms_per_step": round(rd["host_ms"], 2),
                          "exposed_allreduce_ms_per_step": (rd.get("dp") or {}).get("exposed_allreduce_ms_per_step")}
            dist.barrier()

    if rank == 0:
        r = main_res
        wg = "" if ops.WGRAD_PRECISION is None else "; weight gradients two-term bf16 (NPVP_WGRAD=bf16x3 opt-in)"
        res = {"metric": "predictor train frames/sec", "value": round(r["frames_per_s"], 2), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(r["ms"], 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": {"f32": "f32 (fp32-input MFMA)",
                         "bf16x6": f"f32 (three-term bf16 split, 6 MFMAs per product, fp32 accumulate: fp32-grade{wg})",
                         "f16x3": "f32 (two-term fp16 split of amax-scaled operands, 3 MFMAs per product, fp32 accumulate: fp32-grade; "
                                  "small GEMMs: three-term bf16 split)"}[args.gemm],
               "data": "synthetic",
               "config": {"workload": f"{r['name']} " + ("predictor-only train step (features in HBM)" if args.flavour == "predictor"
                                                         else "FULL train step from pixels (frozen AE)")
                                      + (" [HIP-graph replay]" if r["mode"] == "graph" else " [HIP-graph segments + eager collectives]"
                                         if r["mode"] == "graph_segments" else "") + f", {r['B']} clips/GPU, To={r['To']}, "
                                      f"Tp={r['Tp']}, dropout=drop_path=0.1, AdamW+clip",
                          "global_batch": world * r["B"], "frames_per_clip": r["To"] + r["Tp"], "parallelism": f"dp{world}",
                          "algorithmic_tflop_per_step_per_gpu": round(r["flops_step"] / 1e12, 3), "final_loss": round(r["loss"], 6),
                          "host_enqueue_ms_per_step": round(r["host_ms"], 2), "launches_per_step": round(r["launches"]),
                          "peak_device_memory_gib": round(r["peak_gb"], 1), "f16_range_events": r["range_events"]},
               "roofline": r["roof"], "roofline_hbm": r["roof_hbm"], "secondary": secondary or None,
               "secondary_fields": ["ms_per_step", "frames_per_s", "host_enqueue_ms", "library_launches_per_step", "whole_step_tflops"] if secondary else None,
               "secondary_mode": modes or None, "scaling_dp": scaling_dp, "strong_scaling": strong, "dp": r.get("dp"),
               "replay_check": r.get("replay_check"), "graph_nodes": r.get("graph_nodes"), "graph_packet_capture": npvp_amd.graph_packet_capture(),
               "host": dict(zip(("cpu", "cores"), host_cpu()))}
        if world == 1 and not args.no_cpu_baseline and args.flavour == "predictor":
            log("timing the CPU oracle on a bounded sample ...")
            res["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
