"""Step-level scheduling state of the HIP path: which stream a launch goes to, what is deferred to the end of the backward pass,
where gradients are accumulated, the dropout seed, the range guard of the fp16 arithmetic.

All of it is PER TRAINER: a `StepContext` owns one instance of every stateful piece (`rng`, `grad_sink`, `wgrad`, `chain`, `reduce`,
`range_guard`); a trainer holds one (`FlatAdamW(ctx=StepContext())`; without the argument it adopts the context current at its
construction - the process's default one in a single-trainer process), the training step runs under `with use(opt.ctx)`, and every
autograd node remembers the context its forward ran under and re-enters it in backward.  The current context is a global of the
PROCESS (the autograd engine's thread must see what the thread waiting in backward() set): steps of different trainers can be
interleaved, not run concurrently from two threads.  Two predictors with their own contexts in one process therefore do not share a dropout stream, a gradient listener, a
pending join or a sticky fallback (round 4 kept all of that in class attributes: one backward pass, one data-parallel listener,
one seed per PROCESS); bench.py runs every workload under a context of its own.  Code that uses the ops without a trainer (the op
tests, tools) runs under the process's default context.

The module-level names the rest of the package uses - `rng`, `GradSink`, `WgradStream`, `WgradChain`, `ReduceQueue`, `RangeGuard` -
are proxies: attribute reads go to the CURRENT context's instance (and from there to the class for configuration: `enabled`,
`BATCH`, `mode` ... are process-wide knobs), attribute writes go to the instance when it is run state and to the class when it is
configuration.  What stays process-wide by nature: the amax slot chunks (keyed by device and stream), the weight-plane caches (kept
on the tensors they mirror), the gradient stream of a device, bench.py's probes, the opt-in stream experiments.
"""
import contextlib
import ctypes
import os

import torch

from ._lib import lib, check

_p = ctypes.c_void_p


def _ptr(t):
    """device address as a plain int (None -> NULL): the ctypes prototypes declare c_void_p, which takes either"""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_dev = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device      # (the C call, without the lazy-init checks)


def _stream():
    """the calling thread's current HIP stream as a raw handle (one C call: this runs once per kernel launch)"""
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


def _ws(nbytes, dev):
    n = max(int(nbytes), 16)
    return torch.empty((n + 3) // 4, dtype=torch.float32, device=dev), n


# --------------------------------------------------------------------------- process-wide pieces
class AmaxSlot:
    """One amax slot (2 KB of device memory: 32 words, 64 bytes apart, see include/npvp_hip.h): the bound of |x| over a tensor
    that feeds a precision-6 GEMM.  Slots are cut from zero-filled chunks (one torch.zeros per 1024 slots); a slot keeps its
    chunk alive, so a saved-for-backward slot is valid until the node that holds it is freed.

    One current chunk per (device, STREAM): the chunk's zero fill is enqueued on the stream that cuts the slots, so it is ordered
    before every producer's atomic max and every consumer's read on that stream (autograd replays a node on the stream of its
    forward; streams that consume a tensor produced elsewhere are ordered behind its producer by the caller's wait_stream, which
    covers the slot too).  A chunk is also protected from allocator reuse on every side stream this module runs (gradient stream,
    auxiliary stream) and on the device's default stream - its slots may be read there after the cutting stream has moved on."""
    __slots__ = ("ptr", "chunk")
    CHUNK, BYTES, FLOATS = 1024, 2048, 512
    _cur = {}                 # (device index, raw stream) -> [chunk, next slot]

    def __init__(self, ptr, chunk):
        self.ptr, self.chunk = ptr, chunk

    def data_ptr(self):
        return self.ptr

    def read(self):
        """host value (synchronises; tests and diagnostics only)"""
        i = (self.ptr - self.chunk.data_ptr()) // self.BYTES
        return float(self.chunk.view(-1, self.FLOATS)[i].max())

    @classmethod
    def reset_chunks(cls):
        """forget the current chunks (a HIP-graph capture cuts its slots from chunks created INSIDE the capture)"""
        cls._cur = {}

    @classmethod
    def new(cls, dev):
        key = (dev.index, _stream())             # (one C call; the tensors of this module live on the current device)
        st = cls._cur.get(key)
        if st is None or st[1] >= cls.CHUNK:
            cur = torch.cuda.current_stream(dev)
            # (zeroed by a fill KERNEL, explicitly: a hipMemsetAsync would be a memset node of a captured step, and memset nodes are
            #  what the ROCm 7.2 prepared-packet replay does not execute reliably - profiles/r06_graph_alloc_hazard.txt)
            ch = torch.empty(cls.CHUNK, cls.FLOATS, dtype=torch.float32, device=dev).fill_(0.0)
            others = [torch.cuda.default_stream(dev)]
            if WgradStreamState.enabled:
                others.append(WgradStreamState.stream(dev))      # weight-gradient GEMMs read slots on the gradient stream
            if AuxStream.enabled:
                others.append(AuxStream.stream(dev))
            for o in others:
                if o.cuda_stream != cur.cuda_stream:
                    ch.record_stream(o)
            st = cls._cur[key] = [ch, 0]
        s = cls(st[0].data_ptr() + cls.BYTES * st[1], st[0])
        st[1] += 1
        return s


class ProbeEvent:
    """A timing event of the library (include/npvp_hip.h npvp_event_*): recorded on the calling thread's current stream; inside a
    stream capture the record becomes an external event-record node, so the pair bracketing a launch reads that launch's time in
    the LAST replay of the graph (torch.cuda.Event cannot: "External events are disallowed in rocm").  Same interface as the
    torch event the probes used before (record / elapsed_time)."""
    __slots__ = ("h",)

    def __init__(self):
        self.h = lib().npvp_event_create()
        if not self.h:
            raise RuntimeError("npvp_event_create failed")

    def record(self):
        check(lib().npvp_event_record(self.h, _stream()), "npvp_event_record")

    def elapsed_time(self, other):
        """ms from this event to `other` (both complete); negative when one of them was never stamped"""
        return float(lib().npvp_event_elapsed_ms(self.h, other.h))

    def __del__(self):
        h, self.h = self.h, None
        if h:
            try:
                lib().npvp_event_destroy(h)
            except Exception:
                pass


def probe_pair():
    """two fresh timing events, the first already recorded"""
    e0, e1 = ProbeEvent(), ProbeEvent()
    e0.record()
    return e0, e1


class GemmProbe:
    """bench.py's live roofline probe: when armed, every GEMM launch is bracketed by a pair of HIP events on the stream
    it is launched on (no synchronisation; read after the timed region), keyed by (layout, kernel): layout (1,1) forward,
    (1,0) dgrad, (0,0) weight gradient; kernel = npvp_gemm_kernel_id (so the groups line up with the per-kernel rows of a
    rocprofv3 trace of the same command)."""
    armed = False
    records = []          # (start_event, end_event, flops, bytes, (layout, kernel id))
    KERNELS = {0: "npvp::gemm_f32_kernel", 1: "npvp::gemm_split_db_kernel", 2: "npvp::gemm_wide_kernel<2, 4, 2, 2>",
               3: "npvp::gemm_wgrad_wide_kernel", 4: "npvp::gemm_wide_kernel<2, 2, 2, 2>", 5: "npvp::gemm_f16_kernel<2, 4, 2, 2>",
               6: "npvp::gemm_wgrad_f16_kernel", 7: "npvp::gemm_f16_kernel<2, 2, 2, 2>",
               8: "npvp::gemm_f16_group_kernel (dgrad + weight gradient in one launch)"}

    only = None           # set of kernel ids to bracket (None = every GEMM launch)

    @classmethod
    def arm(cls, only=None):
        """only = kernel ids to time: every event pair is a pair of marker packets that fences the launches around it, so the
        benchmark brackets the critical-path (forward / dgrad) kernels by default and the gradient stream's on request"""
        cls.armed, cls.records, cls.only = True, [], (None if only is None else set(only))

    @classmethod
    def disarm(cls):
        cls.armed = False

    @classmethod
    def summary(cls):
        """{(layout, kernel id): (launches, total_ms, total_flops, total_algorithmic_bytes)} - after torch.cuda.synchronize()"""
        out = {}
        for e0, e1, fl, by, key in cls.records:
            dt = e0.elapsed_time(e1)
            if dt < 0.0:                # (a pair that was never stamped: recorded while capturing a graph that was not replayed)
                continue
            n, ms, f, b = out.get(key, (0, 0.0, 0.0, 0.0))
            out[key] = (n + 1, ms + dt, f + fl, b + by)
        return out


class HbmProbe:
    """bench.py's live probe of the HBM-bound family: when armed, the three kernels that lead the non-GEMM time of a step (the
    fused MlpDWBN middle backward, the token LayerNorm backward, the frame-LayerNorm backward apply pass) are bracketed by HIP
    event pairs on the stream they run on, with their ALGORITHMIC bytes (every operand read once, every result
    written once, fp32).  Armed for a few extra steps AFTER the timed region, so the event packets do not perturb `value`."""
    armed = False
    records = []          # (start_event, end_event, kernel name, algorithmic bytes)

    @classmethod
    def begin(cls):
        if not cls.armed:
            return None
        e0 = ProbeEvent()
        e0.record()
        return e0

    @classmethod
    def end(cls, e0, name, nbytes):
        if e0 is None:
            return
        e1 = ProbeEvent()
        e1.record()
        cls.records.append((e0, e1, name, float(nbytes)))

    @classmethod
    def summary(cls):
        """{kernel: (launches, total_ms, total_algorithmic_bytes)} - after torch.cuda.synchronize()"""
        out = {}
        for e0, e1, name, by in cls.records:
            dt = e0.elapsed_time(e1)
            if dt < 0.0:
                continue
            n, ms, b = out.get(name, (0, 0.0, 0.0))
            out[name] = (n + 1, ms + dt, b + by)
        return out


class AuxStream:
    """A second compute stream for INDEPENDENT sub-graphs of the forward pass (the two encoder passes of NPVP-S
    training).  autograd runs each backward node on the stream of its forward, so the two backward chains overlap
    as well; MFMA-bound GEMMs of one chain fill the gaps of the HBM-bound kernels of the other."""
    # opt-in (NPVP_DUAL_ENCODER=1): measured -2.6 ms (2 %) on a c1 step.  Off by default so that (a) every kernel has
    # the device to itself in the forward pass and per-kernel timings agree between bench.py's live probe and a
    # rocprofv3 trace (which serialises the two streams), and (b) under data parallelism the SyncBatchNorm collectives
    # of the two passes are issued from ONE stream in program order.
    enabled = os.environ.get("NPVP_DUAL_ENCODER", "0") == "1"
    active = False           # inside a two-stream region (GemmProbe skips launches there: their durations overlap)
    _streams = {}

    @classmethod
    def stream(cls, dev):
        key = (dev.type, dev.index)
        if key not in cls._streams:
            cls._streams[key] = torch.cuda.Stream(device=dev)
            # gradients of the few parameters that still go through autograd's AccumulateGrad are produced on two
            # streams on purpose; the engine synchronises them, the advisory warning about it is noise here
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
        return cls._streams[key]


# --------------------------------------------------------------------------- per-trainer state
class RngState:
    """Device seed + per-call-site salt counter for the in-kernel counter-hash dropout masks.
    `begin_step()` bumps the device seed (a captured graph replays that bump) and restarts the
    salt counter, so eager and graph-replayed steps draw identical mask streams."""

    def __init__(self, ctx=None):
        self.ctx = ctx
        self.seed = None
        self.salt = 0

    def seed_tensor(self, dev):
        if self.seed is None or self.seed.device != dev:
            self.seed = torch.full((1,), 0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF, dtype=torch.int64, device=dev)
        return self.seed

    def manual_seed(self, s, dev):
        self.seed_tensor(dev).fill_(int(s) & 0x7FFFFFFFFFFFFFFF)
        self.salt = 0

    def begin_step(self, dev):
        self.seed_tensor(dev).add_(0x632BE5AB)
        self.salt = 0

    def next_salt(self):
        self.salt += 1
        return self.salt


class GradSinkState:
    """Parameter gradients go STRAIGHT into the flat gradient buffer.  When a weight / bias / LayerNorm parameter is a
    FlatBuffers parameter (or a contiguous view into one, e.g. the q|k rows of an in_proj_weight or a 1x1 conv weight
    seen as [N, K]), the backward kernels accumulate into the matching slice of its .grad (`accumulate=1` on the C
    entry points) and the autograd Function returns None for it.  This removes the temporary gradient tensors, the
    slice-backward zero+copy kernels and autograd's per-parameter accumulate adds (about 1000 small kernels and 7 ms of
    a 160 ms c1 step).  The flat buffer is zeroed once per step by FlatBuffers.zero_grad(), so every contribution is a
    plain accumulate.  `listener(param)` is called after each contribution (npvp_amd.dp.GradSync counts them to know
    when a bucket is complete)."""
    enabled = True                # (configuration: process-wide)

    def __init__(self, ctx):
        self.ctx = ctx
        self.listener = None      # this trainer's data-parallel listener

    def slot(self, t):
        """-> (grad slice shaped like t, owning parameter) or None"""
        if not self.enabled or t is None or not t.requires_grad:
            return None
        base = t if t.is_leaf else t._base
        if base is None or not base.is_leaf:
            return None
        d = base.__dict__
        if not d.get("_npvp_flat", False):
            return None
        g = base.grad
        if g is None or not t.is_contiguous():
            return None
        # the slot of a given (offset, shape) view never changes while .grad is the same flat-buffer view: cache it on
        # the parameter (this runs ~750 times per step)
        off = t.storage_offset() - base.storage_offset()
        key = (off, t.shape)
        cache = d.get("_npvp_slots")
        if cache is None or cache[0] is not g:
            cache = d["_npvp_slots"] = (g, {})
        hit = cache[1].get(key)
        if hit is None:
            if not g.is_contiguous() or off < 0 or off + t.numel() > g.numel():
                return None
            hit = cache[1][key] = (g.view(-1)[off:off + t.numel()].view(t.shape), base)
        return hit

    def wrote(self, *slots):
        if self.listener is not None:
            for s in slots:
                if s is not None:
                    self.listener(s[1])


class WgradStreamState:
    """Weight-gradient GEMMs run on a SECOND HIP stream.  In backward a layer's dgrad feeds the next layer, but its
    wgrad feeds nobody until the optimiser: when it is accumulated in place (GradSink) it has no consumer in the
    autograd graph at all.  Launched on a side stream (ordered after the producer of dy), the MFMA-bound wgrad GEMMs
    overlap the HBM-bound backward kernels of the following layers (LayerNorm / frame-LN / depthwise / attention
    backward, ~35 ms of a c1 step) and fill the tail of the dgrad GEMMs.  The main stream re-joins the side stream when
    the backward pass finishes (autograd engine callback), so .grad is complete wherever it is read.

    Per trainer: the queue of deferred launches, the pending join, the tensors held for the gradient stream.  Process-wide: the
    gradient stream of a device (one low-priority stream, shared: work of two trainers is serialised on it) and the knobs."""
    enabled = os.environ.get("NPVP_WGRAD_STREAM", "1") == "1"
    # Priority of the gradient stream.  Lowest device priority (critical-path kernels are dispatched first: -1 ms of a c2 step) - but
    # NOT in a process that holds an RCCL communicator: with ProcessGroupNCCL's streams alive beside a low-priority queue every
    # dispatch of the eager two-stream step takes ~50 us longer on the device (c4 shard 29 -> 64 - 72 ms, c2 233 -> 266 ms; a
    # normal-priority gradient stream or GPU_MAX_HW_QUEUES <= 3 restores it: profiles/r06_dp_eager_bisect.txt).  npvp_amd.dp switches
    # this off (`use_normal_priority`) when the data-parallel machinery comes up; NPVP_WGRAD_PRIORITY=low|normal overrides.
    low_priority = os.environ.get("NPVP_WGRAD_PRIORITY", "low") != "normal"
    HOLD_BYTES = 2048 << 20
    BATCH = 16
    _side = {}               # (device type, index) -> the device's gradient stream

    def __init__(self, ctx):
        self.ctx = ctx
        self._pending = None     # (device, side stream) while a backward pass has work in flight on the side stream
        self.in_flush = False    # inside flush(): the current stream is the gradient stream (WgradChain defers reductions there)
        self._held, self._held_bytes, self._held_storages = [], 0, set()
        self._queue = []         # deferred (fn, keep_alive tensors, gradient slots to report) - see run()

    @classmethod
    def stream(cls, dev):
        key = (dev.type, dev.index)
        if key not in cls._side:
            st = None
            # lowest device priority (torch only offers normal / high): critical-path kernels are dispatched first
            with torch.cuda.device(dev):
                h = lib().npvp_stream_create_low_priority(None, None) if cls.low_priority else None
            if h:
                st = torch.cuda.ExternalStream(h, device=dev)
            cls._side[key] = st if st is not None else torch.cuda.Stream(device=dev)
        return cls._side[key]

    @classmethod
    def use_normal_priority(cls):
        """data parallel (an RCCL communicator in the process): gradient streams from now on at normal priority; the low-priority
        ones created so far are retired once the device is idle (no trainer may be inside a backward pass)"""
        if os.environ.get("NPVP_WGRAD_PRIORITY") == "low":
            return
        cls.low_priority = False
        if cls._side:
            torch.cuda.synchronize()
            cls._side.clear()

    def run(self, fn, *keep_alive, wrote=None, urgent=False):
        """fn() on the side stream, after everything already enqueued on the current stream; keep_alive tensors are protected
        from allocator reuse until the side stream has consumed them; `wrote` = gradient slots to report to the GradSink
        listener once fn is enqueued.  Calls are QUEUED and handed to the side stream BATCH at a time (16 since round 4: on host-bound
        shards 3 -> 16 measured 0 .. -4.7 ms per step depending on the box's CPU, c1 -1 ms; and when the backward
        pass ends): one event record / wait and one stream switch per batch instead of per call - 300 of them were 8 ms of an
        8-clip step's 42 ms of host time (c3 shard 48.5 -> 43 ms).  `urgent` hands the queue over at once: large GEMMs, whose
        early start is worth more than the host time (c2: 257 vs 260 ms).  The inputs of fn are never written again on the main stream (they are already read
        concurrently with later main-stream kernels), so starting it a few launches later changes no result.  fn runs under this
        trainer's context (a closure made in one context is never run in another)."""
        self._queue.append((fn, keep_alive, wrote))
        self.begin(keep_alive[0].device)
        if urgent or len(self._queue) >= self.BATCH:
            self.flush()

    def begin(self, dev):
        """this backward pass has (or is about to have) work on the gradient stream: note it and have the pass's end re-join"""
        if self._pending is None:
            self._pending = (dev, self.stream(dev))
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.join)
            except RuntimeError:                    # not inside a backward pass: the caller joins (FlatAdamW.step / zero_grad do)
                pass

    def flush(self):
        if not self._queue:
            return
        q, self._queue = self._queue, []
        dev, side = self._pending
        c = self.ctx
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), use(c):       # (a closure made under this trainer's context runs under it)
            self.in_flush = True
            try:
                for fn, _, _ in q:
                    fn()
                if c.grad_sink.listener is not None:
                    # data parallel: the slots reported below must be COMPLETE on this stream when a bucket's all-reduce is
                    # ordered behind it - a split-K reduction still waiting for its next launch is not
                    c.chain.flush()
                if c.reduce.pending() and (c.grad_sink.listener is not None or c.reduce.due()):
                    c.reduce._launch(side)      # a launch's worth of deferred parameter-gradient reductions (or, data parallel, all)
            finally:
                self.in_flush = False
        for _, keep, slots in q:
            for t in keep:
                self.hold(t)
            if slots is not None:
                c.grad_sink.wrote(*slots)       # (on the caller's stream: the listener orders its collective after both streams)

    def hold(self, t):
        """The gradient stream reads t after the caller may have dropped it: either HOLD a reference until the join (after which the
        compute stream is ordered behind everything the gradient stream did - freeing is then safe without any allocator
        bookkeeping) or, past a byte budget (the large workloads: tens of GB of dy per backward pass), record the stream on the
        block (an event per block when it is freed: 500 of them were ~1 ms of a shard's step).  The budget counts what a reference
        really pins - the tensor's whole STORAGE, once (a small view of a large activation holds all of it: ADVICE r4)."""
        st = t.untyped_storage()
        key = st.data_ptr()
        if key in self._held_storages:
            self._held.append(t)
            return
        nb = st.nbytes()
        if self._held_bytes + nb <= self.HOLD_BYTES:
            self._held.append(t)
            self._held_storages.add(key)
            self._held_bytes += nb
        else:
            t.record_stream(self._pending[1] if self._pending is not None else self.stream(t.device))

    def pending_stream(self):
        """the gradient stream if this backward pass has work in flight on it, else None (npvp_amd.dp orders a bucket's
        all-reduce after it)"""
        return self._pending[1] if self._pending is not None else None

    def join(self):
        """the caller's current stream waits for the gradient stream (the autograd engine runs its final callbacks
        under the streams that were current when backward() was called)"""
        if self._pending is not None:
            self.flush()
            dev, side = self._pending
            c = self.ctx
            if c.reduce.pending():
                side.wait_stream(torch.cuda.current_stream(dev))      # (the partials' producers ran on the caller's stream)
            with torch.cuda.stream(side):
                c.chain.flush()                  # the last weight gradient's split-K reduction has no launch to ride in
                if c.reduce.pending():
                    c.reduce._launch(side)
            torch.cuda.current_stream(dev).wait_stream(side)
            self._pending = None
            self._held, self._held_bytes, self._held_storages = [], 0, set()


class ReduceQueueState:
    """Deferred parameter-gradient reductions (include/npvp_hip.h, npvp_sum_rows_multi / npvp_splitk_reduce_multi).  The LayerNorm /
    frame-LayerNorm / fused MlpDWBN-middle backward kernels leave per-block partial sums of their parameter gradients in a
    workspace; summing them into the flat gradient buffer has no consumer before the optimiser.  One launch per site was ~150
    launches of ~10 us per 8-clip step (a tenth of its launches).  Here a site only writes a 48-byte job record into a host buffer;
    the records are run 40 per launch
      * on the gradient stream, by WgradStream.flush() once 40 have gathered (the large workloads: the reductions keep overlapping
        the backward pass, the workspaces - 2 GB per c2 step - do not pile up) and by WgradStream.join();
      * on the current stream when the backward pass ends (autograd engine callback), if there is no gradient stream (single-stream
        capture of the step into a HIP graph: 146 graph nodes become 4).
    The split-K reductions of the fused dgrad + weight-gradient launches (ops.linear_bwd) go the same way as 64-byte records.
    Two jobs that write the same gradient slice (a LayerNorm applied twice per step: the tied final norm, the encoder of NPVP-S
    training) never share a launch: the queue is run before the second one is added.  Same summation order whoever runs it."""
    enabled = True            # (False: one reduction launch per site, as in round 4 - the tests compare the two)
    JOB, CAP, LAUNCH = 48, 480, 40
    SKJOB, SKCAP, SKLAUNCH = 64, 256, 16      # split-K reductions of fused dgrad + weight-gradient launches: 64-byte records
    DP_LAUNCH = 8                             # data parallel: run the queue every so many jobs (see _dp_progress)
    LAUNCH_BYTES = 192 << 20                  # ... or once this many bytes of partials wait (see due)

    def __init__(self, ctx):
        self.ctx = ctx
        self._buf = self._addr = self._skbuf = self._skaddr = None
        self._n = self._skn = 0
        self._bytes = 0
        self._keep, self._wrote, self._outs = [], [], set()
        self._armed = False

    def pending(self):
        return self._n + self._skn

    def due(self):
        """a launch's worth has gathered - by count, or by the bytes of partials waiting (WgradStream.flush runs the queue then, so
        that the reductions keep overlapping the pass and a large workload's partials - 33 MB per frame LayerNorm at c2 - are not
        read back 1.3 GB at a time: with the count alone the c2 step lost 0.9 ms to the queue, profiles/r05_ab_knobs.txt)"""
        return self._n >= self.LAUNCH or self._skn >= self.SKLAUNCH or self._bytes >= self.LAUNCH_BYTES

    def splitk_slot(self, out_ptrs):
        """host address for the next 64-byte split-K job record (the C call that leaves the partial slabs writes it)"""
        if self._skbuf is None:
            self._skbuf = ctypes.create_string_buffer(self.SKJOB * self.SKCAP)
            self._skaddr = ctypes.addressof(self._skbuf)
        if self._skn == self.SKCAP or not self._outs.isdisjoint(out_ptrs):
            self.run_pending()
        return self._skaddr + self.SKJOB * self._skn

    def splitk_added(self, out_ptrs, keep, wrote):
        self._skn += 1
        self._bytes += keep.numel() * 4
        self._keep.append(keep)
        self._outs.update(out_ptrs)
        if wrote is not None:
            self._wrote.append(wrote)
        self._arm()
        self._dp_progress()

    def _dp_progress(self):
        """Data parallel: a gradient slot is reported to the listener (GradSync) when its reduction has been enqueued, and a bucket's
        all-reduce starts when all its slots are reported - so the queue must not sit on them until the pass ends (on an 8-clip
        shard nearly every gradient goes through this queue: the all-reduce would start after backward).  Every DP_LAUNCH jobs the
        queue is run; always on the gradient stream, where every in-place gradient write is serialised."""
        if self.ctx.grad_sink.listener is not None and self._n + self._skn >= self.DP_LAUNCH:
            self.run_pending()

    def _arm(self):
        if not self._armed:
            self._armed = True
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.finish)
            except RuntimeError:                    # not inside a backward pass (an op test calling the wrappers directly):
                self._armed = False                 # the caller runs finish() itself

    def add(self, filler, name, args, out_ptrs, keep, wrote):
        """filler(*args, job address) = one of the npvp_*_reduce_job entry points; out_ptrs: device addresses the job writes"""
        if self._buf is None:
            self._buf = ctypes.create_string_buffer(self.JOB * self.CAP)
            self._addr = ctypes.addressof(self._buf)
        if self._n == self.CAP or not self._outs.isdisjoint(out_ptrs):
            self.run_pending()
        check(filler(*args, self._addr + self.JOB * self._n), name)
        self._n += 1
        self._bytes += keep.numel() * 4
        self._keep.append(keep)
        self._outs.update(out_ptrs)
        if wrote is not None:
            self._wrote.append(wrote)
        self._arm()
        self._dp_progress()

    def run_pending(self):
        """the queued jobs, now, on the stream where in-place gradient writes belong: the gradient stream whenever there is one
        (also before the pass's first weight gradient has gone there), else the current stream"""
        if self._n + self._skn == 0:
            return
        w = self.ctx.wgrad
        if w.enabled and not w.in_flush:
            w.begin(torch.device("cuda", torch.cuda.current_device()))
            dev, side = w._pending
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._launch(side)
        else:
            self._launch(None)              # (inside WgradStream.flush the current stream IS the gradient stream)

    def _launch(self, side):
        n, skn, keep, wrote = self._n, self._skn, self._keep, self._wrote
        self._n, self._skn, self._bytes, self._keep, self._wrote, self._outs = 0, 0, 0, [], [], set()
        if n:
            check(lib().npvp_sum_rows_multi(self._addr, n, _stream()), "npvp_sum_rows_multi")
        if skn:
            check(lib().npvp_splitk_reduce_multi(self._skaddr, skn, _stream()), "npvp_splitk_reduce_multi")
        w = self.ctx.wgrad
        if side is not None or w.in_flush:
            for t in keep:                         # read on the gradient stream after the caller drops them (see WgradStream.flush)
                w.hold(t)
        for sk in wrote:
            self.ctx.grad_sink.wrote(*sk)

    def finish(self):
        """end of the backward pass (autograd engine callback; also FlatAdamW.step / zero_grad)"""
        self._armed = False
        w = self.ctx.wgrad
        if not w.enabled:
            self.ctx.chain.flush()                  # (single stream: the last fused launch's split-K reduction has no launch to ride in)
        if self._n + self._skn == 0:
            return
        if w.enabled:
            w.begin(torch.device("cuda", torch.cuda.current_device()))
            w.join()                                # (runs the queue on the gradient stream before the streams re-join)
        else:
            self._launch(None)


class WgradChainState:
    """Split-K reductions of the fp16 weight gradients, handed from launch to launch (include/npvp_hip.h, npvp_wgrad_f16_chained):
    a weight gradient accumulated in place on the gradient stream leaves its `splits` partial slabs in a workspace and a 64-byte
    job; the NEXT weight-gradient launch on that stream does the sum with extra workgroups (no launch of its own: 110 of the 170
    reduction launches of an 8-clip step; HBM-bound work beside MFMA-bound work), WgradStream.join() runs the last one.  Same
    summation order as the stand-alone reduction, so results are bit-identical.  `enabled = False`: one reduction launch each."""
    enabled = True
    _ok, _wsb = {}, {}           # (shape caches: pure functions of the shape, process-wide)

    def __init__(self, ctx):
        self.ctx = ctx
        self._pending = {}       # raw stream -> (job bytes, workspace, dw, db, sink slots to report): alive until the job is handed on

    @classmethod
    def takes(cls, M, N, K):
        key = (M, N, K)
        v = cls._ok.get(key)
        if v is None:
            v = cls._ok[key] = bool(lib().npvp_wgrad_f16_chainable(M, N, K))
            cls._wsb[key] = lib().npvp_wgrad_f16_chain_workspace_bytes(M, N, K)
        return v

    def launch(self, dy, x, dw, db, dy_amax, x_amax, a_drop, flag):
        """dw (+)= dy^T x, db (+)= colsum(dy), both ACCUMULATED (GradSink slices), reduction deferred"""
        R, N = dy.shape
        K = x.shape[1]
        st = _stream()
        ws, wsn = _ws(self._wsb[(N, K, R)], dy.device)
        job = ctypes.create_string_buffer(64)
        prev = self._pending.pop(st, None)
        seed = self.ctx.rng.seed_tensor(dy.device) if a_drop.on else None
        probe = GemmProbe.armed and (GemmProbe.only is None or 6 in GemmProbe.only)
        if probe:
            e0, e1 = probe_pair()
        check(lib().npvp_wgrad_f16_chained(N, K, R, _ptr(dy), dy.stride(0), _ptr(x), x.stride(0), _ptr(dw), dw.stride(0), _ptr(db), 1,
                                           _ptr(dy_amax), _ptr(x_amax), _ptr(flag), a_drop.p, a_drop.g1, a_drop.g2, a_drop.salt,
                                           _ptr(seed), ctypes.addressof(prev[0]) if prev is not None else None, ctypes.addressof(job),
                                           _ptr(ws), wsn, st), "npvp_wgrad_f16_chained")
        if probe:
            e1.record()
            GemmProbe.records.append((e0, e1, 2.0 * N * K * R, 4.0 * (N * R + K * R + N * K), ((0, 0), 6)))
        self._pending[st] = (job, ws, dw, db, None)

    def flush(self):
        """the pending job of the CURRENT stream, as a launch of its own"""
        st = _stream()
        prev = self._pending.pop(st, None)
        if prev is not None:
            check(lib().npvp_splitk_reduce_job(ctypes.addressof(prev[0]), st), "npvp_splitk_reduce_job")
            if prev[4] is not None:
                self.ctx.grad_sink.wrote(*prev[4])


class RangeGuardState:
    """The per-ROW range of the two-term fp16 arithmetic (include/npvp_hip.h, `range_flag`).  Forward / dgrad GEMMs repair a tile
    whose rows lie 2^18 or more below the operand's bound themselves (a second pass with per-row scales, inside the kernel).  The
    weight-gradient kernel only DETECTS a feature (a column of dy = a row of dW) that far below dy's bound and raises a device
    counter; what happens then:
      strict (NPVP_RANGE_GUARD=strict, RangeGuard.strict = True: tests, audits): linear_wgrad reads the counter right after the
          launch (a device synchronisation per weight gradient) and re-runs THAT gradient in the six-term bf16 arithmetic, which
          has fp32's exponent range;
      default: nothing is read inside the step.  Whoever drives the steps reads the counter: trainer.predictor_train_step(sync=True)
          where it reads its loss scalars anyway ('f16_range_events'), trainer.GraphedTrainStep every `poll_every` replays without
          blocking (a replayed graph cannot be switched: it is captured again), and a sync=False loop of its own making must call
          RangeGuard.poll() (or poll_async()) every so often - nothing else will.  From the first event on, every weight gradient
          of THIS trainer runs as bf16x6 (sticky; RangeGuard.reset() re-arms).  One step's smallest feature rows are then late by
          one step, never silently wrong for long.
    NPVP_RANGE_GUARD=off passes no counter (the kernel then skips the column maxima).
    Per trainer: the counter, the fallback and the event total (one trainer's event does not slow another's weight gradients).
    The fallback is not for ever (the watch fires per K-chunk, so one chunk in which a feature was momentarily tiny
    would cost the rest of the run): after `retry_after` polls in fallback the fp16 weight gradients are tried again, and the
    interval doubles every time the event comes back within PROBATION polls of a retry (512, 1 024, ... polls, at most RETRY_MAX).  A replayed graph is not switched
    back (it would have to be captured again for the try).
    Under data parallelism each rank decides for itself (its own counter): ranks may run different weight-gradient kernels for a
    while - slower by the slowest rank, never wrong.  Making the ranks agree needs a collective inside poll(), and a poll that one
    rank takes alone (rank 0's single-GPU runs inside a multi-rank job; a reference trainer on the same context) would hang the
    job: a hang is the worse failure, so there is none."""
    mode = os.environ.get("NPVP_RANGE_GUARD", "on")
    strict = mode == "strict"
    RETRY_AFTER, RETRY_MAX, PROBATION = 512, 1 << 16, 64

    def __init__(self, ctx):
        self.ctx = ctx
        self.fallback = False        # weight gradients run as bf16x6
        self.events = 0              # total raised so far (host view)
        self.retry_after = self.RETRY_AFTER
        self._polls_in_fallback = self._probation = 0
        self._flags = {}
        self._async = None           # (pinned host word, event) of a poll_async() in flight

    def flag(self, dev):
        if self.mode == "off":
            return None
        f = self._flags.get(dev)
        if f is None:
            f = self._flags[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
            if WgradStreamState.enabled:
                f.record_stream(WgradStreamState.stream(dev))
        return f

    def _note(self, f, n):
        if n:
            f.zero_()
            self.events += n
            if self._probation > 0:                 # the event came back soon after a retry: wait twice as long next time
                self.retry_after = min(2 * self.retry_after, self.RETRY_MAX)
            self.fallback, self._polls_in_fallback, self._probation = True, 0, 0
        elif self.fallback:
            self._polls_in_fallback += 1
            if self._polls_in_fallback >= self.retry_after:
                self.fallback, self._polls_in_fallback, self._probation = False, 0, self.PROBATION
        elif self._probation > 0:
            self._probation -= 1                    # a clean poll on fp16 again
            if self._probation == 0:
                self.retry_after = self.RETRY_AFTER
        return n

    def poll(self, dev):
        """read and clear the device counter (synchronises); arms the fallback if it was raised"""
        f = self._flags.get(torch.device(dev) if not isinstance(dev, torch.device) else dev)
        if f is None:
            return 0
        return self._note(f, int(f.item()))

    def poll_async(self, dev):
        """the same without blocking, for loops that never synchronise: looks at the copy issued by the PREVIOUS call (if it has
        landed) and issues the next one.  -> events seen by this call (they were raised at least one call ago)"""
        dev = torch.device(dev) if not isinstance(dev, torch.device) else dev
        f = self._flags.get(dev)
        if f is None:
            return 0
        n = 0
        if self._async is not None and self._async[1].query():
            n = self._note(f, int(self._async[0][0]))
            self._async = None
        if self._async is None:
            host = torch.zeros(1, dtype=torch.int32).pin_memory()
            host.copy_(f, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._async = (host, ev)
        return n

    def reset(self):
        self.fallback, self.events, self._async = False, 0, None
        self.retry_after, self._polls_in_fallback, self._probation = self.RETRY_AFTER, 0, 0
        for f in self._flags.values():
            f.zero_()


# --------------------------------------------------------------------------- the context
class StepContext:
    """Everything a training step's launches consult that must not be shared between two trainers."""
    __slots__ = ("rng", "grad_sink", "wgrad", "chain", "reduce", "range_guard", "name")

    def __init__(self, name="default"):
        self.name = name
        self.rng = RngState(self)
        self.grad_sink = GradSinkState(self)
        self.wgrad = WgradStreamState(self)
        self.chain = WgradChainState(self)
        self.reduce = ReduceQueueState(self)
        self.range_guard = RangeGuardState(self)


_default = StepContext()
_cur = _default          # the CURRENT context: a plain module global (npvp_amd.ops reads it as sched._cur.<field> on every launch)


def current():
    """the current context (the process's default one outside `use`)"""
    return _cur


@contextlib.contextmanager
def use(ctx):
    """Run the enclosed launches under `ctx` (None = leave the current one in place).  The current context is one global of the
    process, not of the thread: the autograd engine runs backward nodes in a thread of its own while the thread that called
    backward() waits inside its `use` block, and both must see the same context.  Steps of different trainers may therefore be
    INTERLEAVED in one process (each enters its context), not run concurrently from two Python threads."""
    global _cur
    if ctx is None or ctx is _cur:
        yield
        return
    prev, _cur = _cur, ctx
    try:
        yield
    finally:
        _cur = prev


def scoped(backward):
    """decorator of an autograd Function's backward: run it under the context the node's forward ran under (`ctx.scope`, set by
    `remember`) - a backward pass started outside the trainer's `use` block (a caller's own loss.backward()) still queues its
    weight gradients, reductions and mask replays on the right trainer"""
    def wrapper(ctx, *grads):
        global _cur
        sc = getattr(ctx, "scope", None)
        if sc is None or sc is _cur:
            return backward(ctx, *grads)
        prev, _cur = _cur, sc
        try:
            return backward(ctx, *grads)
        finally:
            _cur = prev
    wrapper.__doc__ = backward.__doc__
    return staticmethod(wrapper)


def remember(ctx):
    """forward of an autograd Function: note the step context on the node"""
    ctx.scope = _cur


class _Scoped:
    """module-level stand-in for one field of the current StepContext (see the module docstring)"""
    __slots__ = ("_field", "_cls")

    def __init__(self, field, cls):
        object.__setattr__(self, "_field", field)
        object.__setattr__(self, "_cls", cls)

    def __getattr__(self, name):
        return getattr(getattr(_cur, self._field), name)

    def __setattr__(self, name, value):
        inst = getattr(_cur, self._field)
        if name in inst.__dict__:
            setattr(inst, name, value)              # run state of the current trainer
        else:
            setattr(self._cls, name, value)         # configuration: process-wide (a test's `ops.WgradStream.enabled = False`)


rng = _Scoped("rng", RngState)
GradSink = _Scoped("grad_sink", GradSinkState)
WgradStream = _Scoped("wgrad", WgradStreamState)
WgradChain = _Scoped("chain", WgradChainState)
ReduceQueue = _Scoped("reduce", ReduceQueueState)
RangeGuard = _Scoped("range_guard", RangeGuardState)
