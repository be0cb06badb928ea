"""Data parallelism for the Stage-2 step: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" for the CPU tests).

What the reference does implicitly through Lightning's DDP strategy + sync_batchnorm=True
(ref/train_Predictor_lightning.py:40-42), done explicitly here:

  C1  gradient all-reduce (mean): the FLAT gradient buffer of FlatBuffers is cut into contiguous
      buckets; a bucket is all-reduced IN PLACE on a side stream as soon as autograd has accumulated the
      last of its parameters (post-accumulate-grad hooks), so the reduction of the decoder's gradients
      overlaps the backward pass of the encoder.  No gradient copies, few large messages (xGMI is
      point-to-point: ring collectives are per-link bound, so buckets are tens of MB, not DDP's 25 MB
      default tuned for NVSwitch).  `finish()` makes the compute stream wait for the side stream before
      the decoder-only clip_grad_norm_ (ref/models/Predictor.py:135), which needs reduced gradients.
      "Filled" = every gradient CONTRIBUTION of every parameter of the bucket has been written: the HIP backward
      kernels accumulate most parameter gradients straight into the flat buffer (ops.GradSink) and report each
      write, the rest arrive through autograd's post-accumulate-grad hooks.  The per-parameter contribution
      counts of a step are learned in the first step (which reduces everything in finish()) and checked after.
  C2  SyncBatchNorm2d for the EventEncoder's three BatchNorm layers: ONE all-reduce of [sum, sum_sq, count]
      per layer forward and one of [sum_dy, sum_dy_xhat] backward.
  C3  parameter/buffer broadcast from rank 0 at construction.
  C4  rank-strided sharding of the global batch (shard_batch).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


# NPVP_DP_FORCE=1: take the data-parallel code path with ONE rank too - process group, model broadcast, SyncBatchNorm2d, GradSync
# with its side stream and asynchronous all-reduces.  With backend "nccl" this executes ProcessGroupNCCL / RCCL itself (init, stream
# semantics of async_op collectives, work.wait() ordering) on a box that has a single GPU (tests/test_dp_gpu.py::test_rccl_one_rank).
FORCE = os.environ.get("NPVP_DP_FORCE", "0") == "1"


# NPVP_DP_COMM=c: the gradient buckets travel through the library's own exchange (include/npvp_hip.h npvp_dp_*: one RCCL communicator
# per process, created here from an id that rank 0 draws and the process group carries) instead of ProcessGroupNCCL's
# all_reduce(async_op=True).  Same buckets, same order, same side stream; the mean is RCCL's ncclAvg instead of a pre-scale + sum.
# What it is for: the proof that the C ABI alone carries a data-parallel step (a host without torch.distributed calls the same
# four functions).  SyncBatchNorm's statistics, the model broadcast and the barrier stay on the process group either way.
COMM = os.environ.get("NPVP_DP_COMM", "torch")


# A segmented replay of the data-parallel step (trainer.GraphedTrainStep(grad_sync=...)): while a trainer.StepTape is RECORDING, every
# collective of the step - a gradient bucket's all-reduce, SyncBatchNorm's statistics, the closing wait - is not issued but handed to
# the tape as a host action that CUTS the HIP-graph capture in two; a replay then alternates graph launches and those actions.
_TAPE = None


def set_tape(tape):
    global _TAPE
    prev, _TAPE = _TAPE, tape
    return prev


def _collective(fn):
    """issue fn() now, or - while a step tape records - make it the host action between two graph segments"""
    if _TAPE is not None:
        _TAPE.cut(fn)
    else:
        fn()


def c_comm_init(group=None):
    """one library communicator over the ranks of `group` (idempotent); the 128-byte id travels by a broadcast on the group"""
    from ._lib import lib, check
    L = lib()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if L.npvp_dp_world() > 0:
        # one communicator per process: a second GradSync on ANOTHER group must not silently reuse the first group's
        if (L.npvp_dp_world(), L.npvp_dp_rank()) != (world, rank):
            raise RuntimeError(f"npvp_dp: the library's communicator spans {L.npvp_dp_world()} ranks (this is rank {L.npvp_dp_rank()}), "
                               f"the requested group has {world} (rank {rank}); npvp_dp_finalize() first")
        return
    on_gpu = dist.get_backend(group) == "nccl"
    idt = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        import ctypes
        raw = ctypes.create_string_buffer(128)
        check(L.npvp_dp_unique_id(ctypes.addressof(raw)), "npvp_dp_unique_id")
        idt = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
    if on_gpu:
        idt = idt.cuda()
    dist.broadcast(idt, 0, group=group)
    import ctypes
    raw = ctypes.create_string_buffer(bytes(idt.cpu().tolist()), 128)
    check(L.npvp_dp_init(rank, world, ctypes.addressof(raw)), "npvp_dp_init")


def active(group=None):
    """the data-parallel machinery is on: more than one rank, or one rank with NPVP_DP_FORCE=1"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE)


def init_distributed(backend=None):
    """Initialise from the torchrun / torch.distributed.run environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or FORCE) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("NPVP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.is_available():
            # gloo + CUDA tensors (several ranks sharing one card) is the single-GPU rehearsal of the RCCL path
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(x, rank, world):
    """Rank-strided partition of the global batch (DistributedSampler semantics)."""
    return x[rank::world].contiguous()


def broadcast_module(module, src=0):
    """parameters and buffers start identical on every rank."""
    if not active():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                dist.broadcast(t.data, src)


class GradSync:
    """Bucketed, overlapped all-reduce(mean) of a FlatBuffers gradient buffer."""

    def __init__(self, buf, bucket_bytes=64 << 20, group=None, last_bucket_bytes=8 << 20, ctx=None, comm=None):
        """buf: the trainer's optimiser (FlatAdamW: its flat buffers AND its scheduling context are taken - the form to use) or a bare
        FlatBuffers (CPU / gloo tests; then ctx = the trainer's sched.StepContext, default: the context current at construction).  The
        context matters: its gradient sink reports the backward kernels' in-place contributions to this object and the bucket
        all-reduces are ordered after its gradient stream - a GradSync listening on another context than the one the step runs
        under would learn too few contributions and reduce buckets early (trainer.predictor_train_step refuses such a pair).
        comm: "torch" (ProcessGroupNCCL / gloo) or "c" (the library's npvp_dp_* exchange; GPU buffers only); default NPVP_DP_COMM"""
        if hasattr(buf, "buf") and hasattr(buf, "ctx"):        # a FlatAdamW
            if ctx is not None and ctx is not buf.ctx:
                raise ValueError("GradSync: ctx differs from the optimiser's own scheduling context")
            buf, ctx = buf.buf, buf.ctx
        self.buf, self.group = buf, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.on = active(group)                 # (one rank with NPVP_DP_FORCE=1 runs the whole machinery on a group of one)
        self.comm = comm or COMM
        assert self.comm in ("torch", "c"), self.comm
        if self.comm == "c" and self.on:
            if not buf.flat_g.is_cuda:
                raise RuntimeError("GradSync(comm='c'): the library's exchange is RCCL - it needs the gradient buffer on a GPU")
            c_comm_init(group)
            from ._lib import lib
            self._L = lib()
        from . import ops
        self.ctx = ctx if ctx is not None else ops.current()
        if self.on and ops.AuxStream.enabled:
            # a bucket's all-reduce is ordered after the compute stream and the gradient stream only: contributions
            # produced on the auxiliary encoder stream (NPVP_DUAL_ENCODER=1) could land after it
            raise RuntimeError("GradSync: NPVP_DUAL_ENCODER=1 (two-stream encoder passes) is not supported under data parallelism")
        self.cuda = buf.flat_g.is_cuda
        if self.on and self.cuda:
            from .sched import WgradStreamState
            WgradStreamState.use_normal_priority()      # (a low-priority queue beside RCCL's streams slows every eager dispatch)
        self.side = torch.cuda.Stream() if self.cuda else None
        # Contiguous buckets over the flat buffer, each owning whole parameters, cut in AUTOGRAD order: the buffer is laid out in
        # module order (coordinate MLP, encoder, event encoders, decoder last), backward fills it from the END, so the buckets are
        # cut walking from the end - full `bucket_bytes` ones first - and what is left at the FRONT (the coordinate MLP and the first
        # encoder layers, whose gradients complete last: every layer's positional tables feed them) is the bucket that is reduced
        # last.  That one is the only all-reduce nothing can hide behind, so it is kept small (`last_bucket_bytes`).
        cap, cap_last = max(1, bucket_bytes // 4), max(1, last_bucket_bytes // 4)
        ends = [buf.offsets[i + 1][0] if i + 1 < len(buf.offsets) else buf.total for i in range(len(buf.offsets))]
        cuts, hi = [], buf.total                     # bucket boundaries (element offsets), found from the end
        for i in range(len(buf.params) - 1, -1, -1):
            lo = buf.offsets[i][0]
            if hi - lo >= cap and lo > 0:
                cuts.append(lo); hi = lo
        if hi > cap_last:                            # the front remainder: split off its head
            for i in range(len(buf.params)):
                if ends[i] >= cap_last and ends[i] < hi:
                    cuts.append(ends[i]); break
        cuts = sorted(set(cuts))
        self.buckets, self.param_bucket = [], {}
        bounds = [0] + cuts + [buf.total]
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            self.buckets.append({"lo": lo, "hi": hi, "n": 0, "ready": 0, "work": None})
        bi = 0
        for p, (off, n) in zip(buf.params, buf.offsets):
            while off >= self.buckets[bi]["hi"]:
                bi += 1
            self.param_bucket[id(p)] = bi
            self.buckets[bi]["n"] += 1
        self._exposed = []                           # (event before, event after) the compute stream's wait for the side stream
        self._handles = []
        self.count = {id(p): 0 for p in buf.params}     # contributions seen this step
        self.expected = None                            # learned in the first step
        if self.on:
            for p in buf.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._hook))
            self.ctx.grad_sink.listener = self._hook
        self.launched = 0

    def _hook(self, p):
        """one gradient contribution of parameter p is in the flat buffer (autograd hook or ops.GradSink)"""
        k = id(p)
        if k not in self.count:
            return
        self.count[k] += 1
        if self.expected is None or self.count[k] != self.expected[k]:
            return
        b = self.buckets[self.param_bucket[k]]
        b["ready"] += 1
        if b["ready"] == b["n"]:
            self._launch(b)

    def _launch(self, b):
        if _TAPE is not None:
            b["work"] = "taped"                 # (finish() of the recording pass must not launch it again)
            _TAPE.cut(lambda b=b: self._launch_now(b))
            return
        self._launch_now(b)

    def _launch_now(self, b):
        g = self.buf.flat_g[b["lo"]:b["hi"]]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.side.wait_event(ev)
            gs = self.ctx.wgrad.pending_stream()            # weight gradients are accumulated on their own stream
            if gs is not None:
                self.side.wait_stream(gs)
            if self.comm == "c":
                from ._lib import check
                check(self._L.npvp_dp_allreduce_async(g.data_ptr(), g.numel(), self.side.cuda_stream), "npvp_dp_allreduce_async")
                b["work"] = True
            else:
                with torch.cuda.stream(self.side):
                    g.mul_(1.0 / self.world)
                    b["work"] = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            g.mul_(1.0 / self.world)
            b["work"] = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.launched += 1

    def finish(self):
        """Call after backward(): reduce any bucket whose hooks did not all fire (unused parameters), then
        order the compute stream after every reduction."""
        if not self.on:
            return
        if self.expected is None:
            self.expected = dict(self.count)
            for b in self.buckets:                      # a bucket is complete when its CONTRIBUTING parameters are
                b["n"] = 0
            for p in self.buf.params:
                if self.expected[id(p)] > 0:
                    self.buckets[self.param_bucket[id(p)]]["n"] += 1
        elif self.count != self.expected:
            # a different graph ran (e.g. an eval-mode forward with gradients, or other parameters in use): buckets may
            # have been reduced before their last contribution.  Drain what is in flight and reset so that the NEXT step
            # re-learns the counts, then fail loudly - this step's gradients are not trustworthy.
            for b in self.buckets:
                if b["work"] is not None and b["work"] is not True and b["work"] != "taped":
                    if self.cuda:
                        with torch.cuda.stream(self.side):
                            b["work"].wait()
                    else:
                        b["work"].wait()
                b["work"], b["ready"] = None, 0
            for k in self.count:
                self.count[k] = 0
            self.expected = None
            if self.cuda:
                torch.cuda.current_stream().wait_stream(self.side)
            raise RuntimeError("GradSync: the per-parameter gradient contribution counts changed between steps "
                               "(a bucket may have been reduced before its last contribution); state was reset, call "
                               "relearn() before changing the training graph on purpose")
        for k in self.count:
            self.count[k] = 0
        if _TAPE is not None:
            # recording: the buckets no hook launched and the closing wait are ONE host action of the replay
            late = [b for b in self.buckets if b["work"] is None]
            for b in self.buckets:
                if b["work"] == "taped":            # (launched by an action of the replay, not now)
                    b["work"] = None
                b["ready"] = 0
            _TAPE.cut(lambda late=late: self._finish_now(late))
            return
        self._finish_now([b for b in self.buckets if b["work"] is None])

    def _finish_now(self, late):
        for b in late:
            self._launch_now(b)
        for b in self.buckets:
            if b["work"] is True:                       # (the library's exchange: stream-ordered on `side`, nothing to wait for here)
                pass
            elif self.cuda:
                with torch.cuda.stream(self.side):
                    b["work"].wait()
            else:
                b["work"].wait()
            b["work"], b["ready"] = None, 0
        if self.cuda:
            # what the compute stream waits here is the EXPOSED part of the step's all-reduces (the tail of the last bucket)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if self.comm == "c":
                from ._lib import check
                check(self._L.npvp_dp_wait(torch.cuda.current_stream().cuda_stream), "npvp_dp_wait")
            else:
                torch.cuda.current_stream().wait_stream(self.side)
            e1.record()
            self._exposed.append((e0, e1))
            del self._exposed[:-64]

    def exposed_ms(self):
        """average time the compute stream spent waiting for the all-reduce stream in finish() over the recorded steps (call after
        a device synchronisation; the first step reduces everything in finish() and is left out)"""
        pairs = self._exposed[1:] if len(self._exposed) > 1 else self._exposed
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs) if pairs else 0.0

    def relearn(self):
        """Call before a step whose autograd graph differs from the previous one (e.g. a random-context batch with a
        different number of context frames does NOT change it - parameter use is the same - but switching the predictor
        between NPVP-S training with / without ground truth does): the next step reduces everything in finish() and
        learns the new contribution counts."""
        self.expected = None
        for k in self.count:
            self.count[k] = 0
        for b in self.buckets:
            b["ready"] = 0

    def remove(self):
        for h in self._handles:
            h.remove()
        if self.ctx.grad_sink.listener == self._hook:
            self.ctx.grad_sink.listener = None


_SYNCBN_GROUP = None


def syncbn_group():
    """The EventEncoder's SyncBatchNorm statistics travel on their OWN communicator: on the gradient communicator the tiny,
    latency-critical [sum, sum_sq, n] all-reduces of the forward pass would queue behind 64 MB bucket reductions that are
    still in flight from the previous step's tail / this step's backward."""
    global _SYNCBN_GROUP
    if _SYNCBN_GROUP is None and active():
        _SYNCBN_GROUP = dist.new_group(ranks=list(range(dist.get_world_size())))
    return _SYNCBN_GROUP


def _chan_sum(t, dims):
    """per-channel sum of a SyncBatchNorm operand.  The predictor hands channels-last ROWS [R, C] on the device: those go through the
    library's fixed-order column sum (ops.colsum) - deterministic, and without the 4 - 8 byte semaphore memset a multi-block torch
    reduction brings into a captured graph segment (trainer._require_memset_free).  Anything else (host tensors of the gloo tests,
    NCHW callers) takes torch's sum."""
    if t.is_cuda and t.dim() == 2 and t.dtype == torch.float32 and t.shape[1] % 4 == 0:
        from . import ops
        return ops.colsum(t if t.is_contiguous() else t.contiguous())
    return t.sum(dims)


class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, training, group):
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        if training:
            cnt = x.new_full((1,), x.numel() / C)             # (a fill kernel, not a host-to-device copy: capturable)
            stat = torch.cat([_chan_sum(x, dims), _chan_sum(x * x, dims), cnt])
            _collective(lambda: dist.all_reduce(stat, group=group))
            n = stat[-1]
            mean = stat[:C] / n
            var = stat[C:2 * C] / n - mean * mean
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(momentum * mean)
                running_var.mul_(1 - momentum).add_(momentum * var * (n / (n - 1)))
        else:
            mean, var, n = running_mean, running_var, None
        shape = [1, C] + [1] * (x.dim() - 2)
        rstd = torch.rsqrt(var + eps)
        xhat = (x - mean.view(shape)) * rstd.view(shape)
        ctx.save_for_backward(xhat, weight, rstd)
        ctx.cfg = (dims, shape, training, group, n)
        return xhat * weight.view(shape) + bias.view(shape)

    @staticmethod
    def backward(ctx, dy):
        xhat, weight, rstd = ctx.saved_tensors
        dims, shape, training, group, n = ctx.cfg
        C = weight.shape[0]
        dw, db = _chan_sum(dy * xhat, dims), _chan_sum(dy, dims)
        g = dy * weight.view(shape)
        if training:
            s = torch.cat([_chan_sum(g, dims), _chan_sum(g * xhat, dims)])
            _collective(lambda: dist.all_reduce(s, group=group))
            dx = rstd.view(shape) * (g - (s[:C] / n).view(shape) - xhat * (s[C:] / n).view(shape))
        else:
            dx = g * rstd.view(shape)
        return dx, dw, db, None, None, None, None, None, None


class SyncBatchNorm2d(nn.BatchNorm2d):
    """BatchNorm2d whose training statistics span every rank (what Lightning's sync_batchnorm=True turns the
    EventEncoder's BatchNorm2d layers into, ref/train_Predictor_lightning.py:41, ref/models/submodules.py:373-383).
    Same parameters / buffers / state-dict keys as nn.BatchNorm2d; works on gloo (CPU) and nccl (RCCL)."""

    def forward(self, x):
        if not (active() and self.training):
            return super().forward(x)
        if self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        return _SyncBNFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum,
                               True, syncbn_group())


def convert_sync_batchnorm(module):
    """Swap every nn.BatchNorm2d under `module` for SyncBatchNorm2d in place (parameters and buffers are shared).
    Collective on first use: every rank must call it (it creates the SyncBatchNorm process group)."""
    syncbn_group()
    for name, child in list(module.named_children()):
        if type(child) is nn.BatchNorm2d:
            new = SyncBatchNorm2d(child.num_features, child.eps, child.momentum, child.affine, child.track_running_stats)
            new.weight, new.bias = child.weight, child.bias
            new.running_mean, new.running_var, new.num_batches_tracked = child.running_mean, child.running_var, child.num_batches_tracked
            new.train(child.training)
            setattr(module, name, new)
        else:
            convert_sync_batchnorm(child)
    return module
