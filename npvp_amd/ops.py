"""torch.autograd glue over the C ABI of libnpvp_hip.so.  PyTorch supplies device memory,
the current HIP stream and the autograd graph; every piece of arithmetic on the hot path is
a hand-written gfx950 kernel reached through ctypes.  There is no CPU / eager fallback:
tensors must be fp32 CUDA(ROCm) tensors and the library must be built.
"""
import ctypes
import os

import torch

from ._lib import lib, check
# scheduling state (per trainer: sched.StepContext) and the process-wide pieces beside it; re-exported here because the rest of the
# package, the tests and the tools reach them as ops.<name>
from . import sched as _S            # (hot paths read the current context as _S._cur.<field>: one global + two attribute loads)
from .sched import (_p, _ptr, _stream, _ws, AmaxSlot, GemmProbe, HbmProbe, ProbeEvent, probe_pair, AuxStream, StepContext, current, use, scoped,
                    remember, rng, GradSink, WgradStream, WgradChain, ReduceQueue, RangeGuard)


def amax_of(t, slot=None):
    """the amax slot of a 2-D fp32 matrix: the one its producer attached (`t._npvp_amax`), else a fresh slot filled by
    the stand-alone reduction kernel (one read of t)"""
    if slot is not None:
        return slot
    for cand in (t, t._base):               # a view is bounded by the slot of the tensor it is a view of
        tag = getattr(cand, "_npvp_amax", None) if cand is not None else None
        if tag is not None and tag[1] == cand._version:     # (an in-place update after the slot was filled voids it)
            return tag[0]
    s = AmaxSlot.new(t.device)
    t2 = t if t.dim() == 2 else t.reshape(-1, t.shape[-1])
    if t2.stride(1) != 1:
        t2 = t2.contiguous()
    check(lib().npvp_amax(_ptr(t2), t2.shape[0], t2.shape[1], t2.stride(0), _ptr(s), _stream()), "npvp_amax")
    t._npvp_amax = (s, t._version)
    return s


def tag_amax(t, slot):
    """attach a producer-filled slot to the tensor object that carries the values on (also across autograd nodes: the
    engine hands the next node the same Python object unless it had to sum two gradients)"""
    if slot is not None:
        t._npvp_amax = (slot, t._version)
    return t


def _row(t, i):
    """device address of row i of a contiguous fp32 matrix (t[i] without building the view: ~1 000 of them per step)"""
    return t.data_ptr() + 4 * i * t.stride(0)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("npvp_amd ops need tensors on an MI355X device (no CPU fallback)")
        if t.dtype != torch.float32:
            raise RuntimeError("npvp_amd ops are fp32")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------- dropout state
class Drop:
    """A dropout / drop-path site: probability, keying mode (0 element, 1 row group) and salt."""
    __slots__ = ("p", "mode", "g1", "g2", "salt")

    def __init__(self, p=0.0, mode=0, g1=1, g2=1):
        self.p, self.mode, self.g1, self.g2 = float(p), mode, g1, g2
        self.salt = _S._cur.rng.next_salt() if p > 0 else 0

    @property
    def on(self):
        return self.p > 0.0

    @classmethod
    def _like(cls, other, mode, g1, g2):
        """same probability and salt (= same mask stream), other keying geometry"""
        d = cls.__new__(cls)
        d.p, d.mode, d.g1, d.g2, d.salt = other.p, mode, g1, g2, other.salt
        return d


NO_DROP = Drop(0.0)


class DropRecorder:
    """Test hook: while `sites` is a list, every forward kernel that applies a dropout / drop-path mask appends
    (Drop, kind, count) - kind "elem": one decision per flat element index 0..count-1 (nn.Dropout sites and the
    attention-probability dropout, whose key is the flat index of the [groups, heads, L, S] weights); kind "group": one
    decision per row group 0..count-1 (DropPath).  `mask(site)` replays the site's mask through npvp_drop_apply on ones,
    so a parity test can inject the very masks the kernels drew into the oracle (tests/test_hip_dropout.py)."""
    sites = None

    @classmethod
    def note(cls, drop, kind, count):
        if cls.sites is not None and drop.on:
            cls.sites.append((drop, kind, int(count)))

    @staticmethod
    def mask(site, dev):
        """the site's keep-scale per decision (0 or 1/(1-p)) as a flat tensor, from the current device seed"""
        drop, kind, count = site
        if kind == "elem":
            pad = (-count) % 4
            ones = torch.ones((count + pad) // 4, 4, dtype=torch.float32, device=dev)
            return drop_apply(ones, Drop._like(drop, 0, 1, 1)).reshape(-1)[:count]
        ones = torch.ones(count, 4, dtype=torch.float32, device=dev)          # group g = row g (g1 = 1, g2 = count)
        return drop_apply(ones, Drop._like(drop, 1, 1, count))[:, 0].contiguous()


# GEMM arithmetic (NPVP_GEMM=bf16x6|f32), fp32 accumulation on the matrix cores:
#   bf16x6 (default) every fp32 operand is split into three bf16 terms and the six leading cross products run on
#          v_mfma_f32_32x32x16_bf16: ~2^-23 per product, fp32-grade.  Kernels: gemm_wide_kernel (256-row tiles, weights
#          from pre-split planes: the large forward / dgrad shapes) and gemm_split_db_kernel (128 x 128 tiles: small
#          shapes, weight gradients)
#   f32    exact fp32-input MFMA (v_mfma_f32_32x32x2_f32), 1/16 of the bf16 rate: parity triage
#   bf16x3 two-term split, 3 MFMAs per product, ~2^-16: only ever used for weight gradients, opt-in (NPVP_WGRAD=bf16x3)
#   f16x3  two fp16 terms per operand, 3 MFMAs per product, ~2^-22: fp32-grade at half the matrix work of bf16x6.  Operands
#          are scaled by the power of two of their amax slot (csrc/gemm_f16.hip); taken by the large forward / dgrad shapes
#          (weights as scaled fp16 planes) and the weight gradients, everything else runs as bf16x6
GEMM_MODES = {"f32": 0, "bf16x6": 4, "bf16x6db": 4, "bf16x3": 5, "bf16x3db": 5, "f16x3": 6}
GEMM_PRECISION = GEMM_MODES[os.environ.get("NPVP_GEMM", "f16x3")]


# Arithmetic of the WEIGHT-GRADIENT GEMMs.  Default: the same six-term split as forward / dgrad (fp32-grade, what the
# reference's fp32 wgrad delivers).  NPVP_WGRAD=bf16x3 is an OPT-IN fast mode (two bf16 terms, 3 MFMAs per product, 2^-16
# per product): a weight gradient is a leaf of the backward graph, so its rounding error does not propagate through the
# layers, and the reference's small-N vectors are met (weight-gradient rel-L2 4e-6..7e-6), but there is no full-depth,
# full-batch evidence for it - bench.py never measures it unless asked to, and says so in `dtype` when it does.
WGRAD_PRECISION = {"": None, "bf16x6": None, "same": None, "bf16x3": 5}[os.environ.get("NPVP_WGRAD", "")]


def set_gemm_precision(name):
    global GEMM_PRECISION
    GEMM_PRECISION = GEMM_MODES[name]


class WeightPlanes:
    """The pre-split planes of every weight that is a GEMM B operand: F planes for the forward GEMM (B = w as [N][K]), D planes
    for dgrad (B = w as [K][N]).  A weight changes once per optimiser step but is staged by every tile of two GEMMs, so it
    is split ONCE per step, into the exact layout of the GEMM's LDS image - the wide GEMM kernels copy it HBM -> LDS by
    LDS-DMA and spend no VALU or VGPR on it.  Format by GEMM mode: bf16x6 = three bf16 term planes (npvp_split_weight),
    f16x3 = two fp16 planes scaled by the power of two of the weight's amax, which lives in the entry's amax slot
    (npvp_split_weight_f16).
      * A view registers itself on first use (split there and then).
      * `refresh_all()` (FlatAdamW.step, right after the AdamW kernel) re-splits EVERY registered view in one batched call
        per format over a device table - not ~350 small launches per step, and no per-call bookkeeping.
      * An in-place torch update (load_state_dict, a test's fill) is caught by the tensor version counter: that view is
        re-split lazily.
    The cache lives ON the owning tensor object (the Parameter, or the flat parameter buffer it is a view of), never in a
    table keyed by device address alone: a freed weight's address is reused by the next model's weights.  The key holds the
    view's device address too: re-pointing `p.data` (FlatBuffers does) makes a NEW entry, and entries whose storage is no longer
    the owner's are dropped.
    `enabled = False` (tests) makes every GEMM split both operands on the fly (bf16x6 128 x 128 kernel only)."""
    enabled = True
    _owners = []            # weak references to tensors that carry a `_npvp_planes` store
    _tables = None          # [(fmt, device table, amax table or None, [entries])] - rebuilt when `_dirty`
    _group_tensors = {}     # (storage address, fmt, device) -> (entry identities, device table, amax table): see refresh_all
    _dirty = True
    _gen = object()         # identity token of the current generation of entries

    class Entry:
        __slots__ = ("fmt", "version", "planes", "w", "amax", "amax_t", "F", "D")

    @classmethod
    def _owner_died(cls, _ref):
        cls._dirty = True

    @classmethod
    def invalidate(cls, owner=None):
        """the parameters changed under the planes (optimiser step): re-split now - every registered weight, or those that are
        views of `owner` (a trainer's flat parameter buffer: another trainer's planes are still current)"""
        cls.refresh_all(owner)

    @classmethod
    def forget(cls, owner):
        """drop the planes cached on `owner` (FlatBuffers re-points parameter storage: the old planes mirror dead memory)"""
        if owner.__dict__.pop("_npvp_planes", None) is not None:
            cls._dirty = True
            cls._gen = object()         # voids every remembered entry (`_npvp_ent`) at once

    @staticmethod
    def _split(ent):
        w = ent.w
        N, K = w.shape
        F, D = _p(ent.planes[0].data_ptr()), _p(ent.planes[1].data_ptr())
        if ent.fmt == 6:
            check(lib().npvp_split_weight_f16(_ptr(w), w.stride(0), N, K, F, D, _ptr(ent.amax), _stream()), "npvp_split_weight_f16")
        else:
            check(lib().npvp_split_weight(_ptr(w), w.stride(0), N, K, F, D, _stream()), "npvp_split_weight")
        ent.version = w._version

    @classmethod
    def get(cls, w, want):
        """want = 'F' (forward, B = w as [N][K]) or 'D' (dgrad, B = w as [K][N]); returns (planes buffer, amax slot or None)
        or None when this weight has no planes."""
        fmt = GEMM_PRECISION
        # fast path (a Parameter that went through the slow path before: the entry and its address are remembered on the tensor
        # object; ~350 lookups per step)
        hit = w.__dict__.get("_npvp_ent")
        if hit is not None:
            ent = hit[0]
            if ent.fmt == fmt and hit[1] == w.data_ptr() and ent.version == w._version and hit[2] is cls._gen and cls.enabled:
                return (ent.F if want == "F" else ent.D), ent.amax
        if not cls.enabled or fmt not in (4, 6) or w.dim() != 2 or w.shape[0] % 8 or w.shape[1] % 8 or w.stride(1) != 1:
            return None
        owner = w._base if w._base is not None else w
        if not owner.is_leaf and owner.grad_fn is not None:
            return None                     # a temporary (e.g. a permuted conv weight): nothing persistent to cache on
        store = owner.__dict__.get("_npvp_planes")
        if store is None:
            import weakref
            store = owner.__dict__["_npvp_planes"] = {}
            cls._owners.append(weakref.ref(owner, cls._owner_died))
        key = (fmt, w.data_ptr(), tuple(w.shape), w.stride(0))
        ent = store.get(key)
        if ent is None:
            live = owner.untyped_storage().data_ptr()
            for k in [k for k, e in store.items() if e.w.untyped_storage().data_ptr() != live]:
                del store[k]                # entries of a storage this owner no longer has (p.data was re-pointed)
            N, K = w.shape
            ent = cls.Entry()
            ent.fmt, ent.w = fmt, w.detach()
            if fmt == 6:
                ent.planes = torch.empty(2, 2 * N * K, dtype=torch.float16, device=w.device)
                ent.amax_t = torch.zeros(AmaxSlot.FLOATS, dtype=torch.float32, device=w.device)
                ent.amax = AmaxSlot(ent.amax_t.data_ptr(), ent.amax_t)
            else:
                ent.planes = torch.empty(2, 3 * N * K, dtype=torch.bfloat16, device=w.device)
                ent.amax = ent.amax_t = None
            ent.F, ent.D = ent.planes[0], ent.planes[1]
            cls._split(ent)
            store[key] = ent
            cls._dirty = True
        elif ent.version != w._version:
            cls._split(ent)
        if w._base is None and tuple(w.shape) == tuple(ent.w.shape):
            w.__dict__["_npvp_ent"] = (ent, w.data_ptr(), cls._gen)
        return (ent.F if want == "F" else ent.D), ent.amax

    @classmethod
    def _entries(cls):
        live = []
        for ref in cls._owners:
            owner = ref()
            if owner is None:
                continue
            live.append(ref)
            for ent in owner.__dict__.get("_npvp_planes", {}).values():
                yield ent
        cls._owners = live

    @classmethod
    def refresh_if_stale(cls):
        """re-split everything if any registered weight was updated in place behind the planes' back (GraphedTrainStep calls
        this before a replay: the captured step refreshes the planes only after ITS optimiser step)"""
        if any(ent.version != ent.w._version for ent in cls._entries()):
            cls.refresh_all()

    epoch = 0               # bumped whenever the parameters changed under us (caches derived from parameters key on it)

    @classmethod
    def refresh_all(cls, owner=None):
        cls.epoch += 1
        if not cls.enabled:
            return
        if cls._dirty or cls._tables is None:
            owner = None                # (the rebuild below hands every entry a fresh, zeroed amax slot: everything is split again)
            groups = {}
            for ent in cls._entries():
                if ent.w.is_cuda:       # (grouped by STORAGE: the weights of one trainer are views of its flat parameter buffer)
                    groups.setdefault((ent.w.untyped_storage().data_ptr(), ent.fmt, ent.w.device), []).append(ent)
            cls._tables = []
            kept = {}
            for gkey, ents in groups.items():
                own_id, fmt, dev = gkey
                # A group whose entries are the ones it had at the last rebuild KEEPS its device table and its amax slots: the rebuild
                # was caused by something else (another model's weights registered or garbage collected), and a captured HIP graph of
                # this trainer's step has the addresses of both baked in - a fresh pair would leave the graph re-splitting into, and
                # its GEMMs reading scales from, memory that the allocator hands to the next caller (round 6: a replayed step went
                # wrong as soon as anything was allocated and written between two replays).
                ids = tuple(id(e) for e in ents)
                old = cls._group_tensors.get(gkey)
                if old is not None and old[0] == ids and (fmt != 6 or all(e.amax_t is old[2] for e in ents)):
                    table, amax_t = old[1], old[2]
                else:
                    amax_t = None
                    if fmt == 6:             # one contiguous slot table per device: zeroed by ONE memset in the batched call
                        amax_t = torch.zeros(len(ents), AmaxSlot.FLOATS, dtype=torch.float32, device=dev)
                        for i, ent in enumerate(ents):
                            ent.amax_t = amax_t
                            ent.amax.ptr, ent.amax.chunk = amax_t.data_ptr() + AmaxSlot.BYTES * i, amax_t
                    rows = []
                    for ent in ents:
                        N, K = ent.w.shape
                        r = [ent.w.data_ptr(), ent.w.stride(0), N, K, ent.planes[0].data_ptr(), ent.planes[1].data_ptr()]
                        rows.append(r + [ent.amax.ptr, 0] if fmt == 6 else r)
                    table = torch.tensor(rows, dtype=torch.int64).to(dev)
                kept[gkey] = (ids, table, amax_t)
                cls._tables.append((fmt, table, amax_t, ents, own_id))
            cls._group_tensors = kept
            cls._dirty = False
        only = None if owner is None else owner.untyped_storage().data_ptr()
        for fmt, table, amax_t, ents, own_id in cls._tables:
            if only is not None and own_id != only:
                continue
            with torch.cuda.device(table.device):
                if fmt == 6:
                    check(lib().npvp_split_weights_f16(_p(table.data_ptr()), table.shape[0], _p(amax_t.data_ptr()),
                                                       amax_t.numel() * 4, _stream()), "npvp_split_weights_f16")
                else:
                    check(lib().npvp_split_weights_batched(_p(table.data_ptr()), table.shape[0], _stream()), "npvp_split_weights_batched")
            for ent in ents:
                ent.version = ent.w._version


# --------------------------------------------------------------------------- raw kernel wrappers


def gemm(a_kc, b_kc, M, N, K, A, lda, B, ldb, out, bias=None, act=0, aux_in=None, aux_out=None, residual=None,
         drop=NO_DROP, alpha=1.0, colsum_a=None, b_pre=None, accumulate=False, rowstats=None, precision=None,
         replay=False, a_amax=None, b_amax=None, c_amax=None, a_drop=NO_DROP, range_flag=None):
    """replay=True: `drop` replays a forward site's mask in backward (not a new site for ops.DropRecorder).
    b_pre = WeightPlanes.get(...) = (planes, weight amax slot) or None; a_amax / b_amax: the operands' amax slots (f16x3; a
    missing slot of A - or of B for a weight gradient - is filled by the stand-alone reduction); c_amax: slot that receives
    the bound of the values stored to `out`.  a_drop: a row-group (DropPath) mask on the rows of A - fp16 kernels only, see
    `masked_grad`."""
    # (this wrapper runs for a third of a step's launches: the optional arguments are tested in line instead of through _ptr / _chk,
    # ~3 us of its ~9 us of interpreter time)
    for t in (A, B, out, bias, aux_in, aux_out, residual, colsum_a):
        if t is not None and not (t.is_cuda and t.dtype is torch.float32):
            _chk(t)
    prec = GEMM_PRECISION if precision is None else precision
    planes = None
    if b_pre is not None:
        planes, b_amax = b_pre
    if prec == 6:
        kid = _gemm_kernel_id(a_kc, b_kc, M, N, K, 6, planes is not None)
        if kid == 5 or kid == 7:                # fp16 forward / dgrad kernel
            if a_amax is None:
                a_amax = amax_of(A)
        elif kid == 6:                          # fp16 weight-gradient kernel
            if a_amax is None:
                a_amax = amax_of(A)
            if b_amax is None:
                b_amax = amax_of(B)
        else:                                   # runs as bf16x6 without planes
            planes = None
    wsb = _gemm_ws_bytes(M, N, K)
    ws, wsn = (None, 0)
    if wsb > 0:
        ws, wsn = _ws(wsb, A.device)
    masked = drop.p > 0.0 or a_drop.p > 0.0
    seed = _S._cur.rng.seed_tensor(A.device) if masked else None
    if DropRecorder.sites is not None and not replay:
        DropRecorder.note(drop, "elem" if drop.mode == 0 else "group", M * N if drop.mode == 0 else drop.g2)
    # every launch is timed on the stream it runs on, also those that share the device with a kernel of another stream:
    # the population (and the average duration) is then the same as in a rocprofv3 kernel trace of the same command
    probe = GemmProbe.armed
    if probe:
        kid = _gemm_kernel_id(a_kc, b_kc, M, N, K, prec, planes is not None)
        probe = GemmProbe.only is None or kid in GemmProbe.only
    if probe:
        e0, e1 = probe_pair()
    rc = lib().npvp_gemm_f32(
        a_kc, b_kc, M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), out.stride(0),
        None if bias is None else bias.data_ptr(), act, None if aux_in is None else aux_in.data_ptr(),
        None if aux_out is None else aux_out.data_ptr(), None if residual is None else residual.data_ptr(),
        0 if residual is None else residual.stride(0), drop.p, drop.mode, drop.g1, drop.g2,
        None if seed is None else seed.data_ptr(), drop.salt, alpha, prec, None if colsum_a is None else colsum_a.data_ptr(),
        None if planes is None else planes.data_ptr(), 1 if accumulate else 0, None if rowstats is None else rowstats.data_ptr(),
        None if a_amax is None else a_amax.data_ptr(), None if b_amax is None else b_amax.data_ptr(),
        None if c_amax is None else c_amax.data_ptr(), None if range_flag is None else range_flag.data_ptr(),
        a_drop.p, a_drop.g1, a_drop.g2, a_drop.salt, None if ws is None else ws.data_ptr(), wsn, _stream())
    if rc:
        check(rc, "npvp_gemm_f32")
    if probe:
        e1.record()
        GemmProbe.records.append((e0, e1, 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N), ((a_kc, b_kc), kid)))
    return out


_KID_CACHE, _WSB_CACHE = {}, {}


def _gemm_kernel_id(a_kc, b_kc, M, N, K, prec, has_planes):
    """npvp_gemm_kernel_id, remembered per shape (a pure function of its arguments; one C call less per GEMM)"""
    key = (a_kc, b_kc, M, N, K, prec, has_planes)
    v = _KID_CACHE.get(key)
    if v is None:
        v = _KID_CACHE[key] = lib().npvp_gemm_kernel_id(a_kc, b_kc, M, N, K, prec, int(has_planes))
    return v


def _gemm_ws_bytes(M, N, K):
    v = _WSB_CACHE.get((M, N, K))
    if v is None:
        v = _WSB_CACHE[(M, N, K)] = lib().npvp_gemm_workspace_bytes(M, N, K)
    return v


def _planes(w, want, R):
    return WeightPlanes.get(w, want) if R >= 256 else None


def _new_slot(dev, want=True):
    """a fresh amax slot for a tensor a kernel is about to produce, when the GEMM mode uses them"""
    return AmaxSlot.new(dev) if (want and GEMM_PRECISION == 6) else None


def linear_fwd(x, w, b, act=0, aux_out=None, residual=None, drop=NO_DROP, rowstats=None, x_amax=None, y_amax=None):
    """y[R,N] = epilogue(x[R,K] w[N,K]^T); rowstats [R/64, N/64, 2] receives the frame-statistics partials of y;
    x_amax: x's amax slot if its producer filled one; y_amax: slot to receive the bound of y"""
    R, K = x.shape
    N = w.shape[0]
    y = torch.empty(R, N, dtype=torch.float32, device=x.device)
    return gemm(1, 1, R, N, K, x, x.stride(0), w, w.stride(0), y, bias=b, act=act, aux_out=aux_out, residual=residual,
                drop=drop, b_pre=_planes(w, "F", R), rowstats=rowstats, a_amax=x_amax, c_amax=y_amax)


def linear_frame_stats_supported(R, N):
    """the forward GEMM can emit the frame-LayerNorm statistics of its output (frames of 64 token rows)"""
    return GEMM_PRECISION in (4, 6) and R % 64 == 0 and N % 128 == 0


def linear_dgrad(dy, w, act=0, aux_in=None, drop=NO_DROP, residual=None, dy_amax=None, dx_amax=None, a_drop=NO_DROP, out=None,
                 accumulate=False):
    """dx[R,K] = epilogue((a_drop mask on the rows of dy) dy[R,N] w[N,K]); out / accumulate: into (added to) an existing [R,K] buffer"""
    R, N = dy.shape
    K = w.shape[1]
    dx = torch.empty(R, K, dtype=torch.float32, device=dy.device) if out is None else out
    return gemm(1, 0, R, K, N, dy, dy.stride(0), w, w.stride(0), dx, act=act, aux_in=aux_in, drop=drop, residual=residual,
                b_pre=_planes(w, "D", R), replay=True, a_amax=dy_amax, c_amax=dx_amax, a_drop=a_drop, accumulate=accumulate)


def masked_grad(dy2, drop, w):
    """The gradient that flows into a linear whose output went through `drop` (dropout or DropPath): -> (dz, a_drop).  A DropPath
    (row-group) mask is handed to the dgrad and weight-gradient GEMMs as `a_drop` when both run on the fp16 kernels - they fold
    it into the scale of the rows they stage, so no masked copy of dy is written or read (2 passes over [R, C] and one launch
    per site); every other case applies the mask here (npvp_drop_apply)."""
    if not drop.on:
        return dy2, NO_DROP
    R, N = dy2.shape
    K = w.shape[1]
    if (drop.mode == 1 and drop.g1 % 16 == 0 and drop.p < 0.5 and GEMM_PRECISION == 6 and R <= DROP_PATH_IN_GEMM_ROWS
            and _planes(w, "D", R) is not None and _gemm_kernel_id(1, 0, R, K, N, 6, True) in (5, 7)
            and _gemm_kernel_id(0, 0, N, K, R, 6, False) == 6):
        return dy2, drop
    return drop_apply(dy2, drop), NO_DROP


# Up to this many token rows.  Rounds 3 - 4 limited it to 32 768 (the launch-bound shards: c4 36.4 -> 35.1 ms; on c2 the step did
# not change then, 245.6 vs 246.2 ms, while the dgrad launches' event-pair rate fell - the masked epilogues still divided per
# element).  With the multiply-shift group index (round 4, end) the masked launch costs what the unmasked one does, and round 5's
# in-process A/B at c2 (tools/ab_bench.py: A, B, A, B on one box) reads 233.4 / 232.8 ms with the limit against 232.4 / 232.2
# without: no limit any more (20 drop_apply passes over [114 688 x 512] per step gone).
DROP_PATH_IN_GEMM_ROWS = 1 << 30


def linear_wgrad(dy, x, with_bias_grad=False, into=None, into_b=None, dy_amax=None, x_amax=None, a_drop=NO_DROP):
    """dw[N,K] = dy[R,N]^T x[R,K]; with_bias_grad also returns db[N] = column sums of dy, accumulated by the same
    kernel while it stages dy (no separate reduction pass).  into / into_b: ACCUMULATE into these existing
    gradient slices instead of allocating results (GradSink).  Range of the fp16 arithmetic: RangeGuard."""
    R, N = dy.shape
    K = x.shape[1]
    acc = into is not None
    dw = into if acc else torch.empty(N, K, dtype=torch.float32, device=dy.device)
    db = (into_b if acc else torch.empty(N, dtype=torch.float32, device=dy.device)) if with_bias_grad else None
    prec = WGRAD_PRECISION if GEMM_PRECISION == 4 else None
    watch = GEMM_PRECISION == 6 and _gemm_kernel_id(0, 0, N, K, R, 6, False) == 6
    if watch and _S._cur.range_guard.fallback and not a_drop.on:
        prec, watch = 4, False                     # (a launch that carries a row-group mask needs the fp16 kernel: it stays there)
    if watch and _S._cur.range_guard.strict and not a_drop.on:
        # strict: into a scratch result first, so that a flagged launch leaves the accumulation target untouched
        flag = _S._cur.range_guard.flag(dy.device)
        tw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
        tb = torch.empty(N, dtype=torch.float32, device=dy.device) if with_bias_grad else None
        gemm(0, 0, N, K, R, dy, dy.stride(0), x, x.stride(0), tw, colsum_a=tb, a_amax=dy_amax, b_amax=x_amax, range_flag=flag)
        if _S._cur.range_guard.poll(dy.device):
            _S._cur.range_guard.fallback = False             # (strict mode repairs launch by launch)
            gemm(0, 0, N, K, R, dy, dy.stride(0), x, x.stride(0), tw, colsum_a=tb, precision=4)
        if acc:
            dw.add_(tw)
            if tb is not None:
                db.add_(tb)
        else:
            dw, db = tw, tb
        return (dw, db) if with_bias_grad else dw
    flag = _S._cur.range_guard.flag(dy.device) if watch else None
    # chained: on the gradient stream, or - without one (the single-stream step that is captured into a graph) - on the current stream,
    # where the pending reduction rides in the next weight-gradient or fused launch and ReduceQueue.finish() runs the last one.  Not
    # without a gradient stream under data parallelism: a bucket's all-reduce must not overtake a reduction that is not enqueued yet.
    on_side = _S._cur.wgrad.in_flush
    if (watch and acc and prec is None and _S._cur.chain.enabled and _S._cur.chain.takes(N, K, R)
            and (on_side or (not _S._cur.wgrad.enabled and _S._cur.grad_sink.listener is None))
            and dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(1) == 1):
        _S._cur.chain.launch(dy, x, dw, db, amax_of(dy, dy_amax), amax_of(x, x_amax), a_drop, flag)
        if not on_side:
            _S._cur.reduce._arm()
        return (dw, db) if with_bias_grad else dw
    gemm(0, 0, N, K, R, dy, dy.stride(0), x, x.stride(0), dw, colsum_a=db, accumulate=acc, precision=prec, a_amax=dy_amax, b_amax=x_amax,
         a_drop=a_drop, range_flag=flag)
    return (dw, db) if with_bias_grad else dw


class FusedLinearBwd:
    """dgrad + weight gradient of one linear layer as ONE launch (include/npvp_hip.h, npvp_linear_bwd_f16) - for the shapes on which
    each of the two alone leaves half the chip idle: small-tile dgrads with at least 1 024 and at most MAX_ROWS token rows (the
    8-clip shards of the data-parallel configurations; the large workloads keep the two-stream schedule).  The weight gradient's
    split-K reduction is the only in-place gradient write: with a gradient stream it is queued for that stream (ReduceQueue), where
    every in-place write is serialised; without one (the step captured single-stream into a HIP graph) it rides in the next fused
    launch on the same stream (WgradChain)."""
    enabled = True
    MAX_ROWS = 16384
    # Beside a gradient stream the two launches on two streams are the better schedule wherever the GPU is the bound (in-process
    # A/Bs, profiles/r05_ab_knobs.txt: c4 shard eager 30.5 ms unfused / 31.3 fused, c2 233.3 / 234.3, c1 with 20 480-row layers
    # fused 89.8 against 86.2): the fused launch is for the step WITHOUT a gradient stream - the single-stream capture that is
    # replayed from a graph.  with_gradient_stream = True takes it in the eager two-stream step too: fewer launches (934 instead of
    # 1 160 per c4 step), which is what a host-bound step wants (bench.py tries both under data parallelism, where the step
    # cannot be replayed from a graph).
    with_gradient_stream = False
    _ok = {}

    @classmethod
    def takes(cls, R, N, K):
        key = (R, N, K)
        v = cls._ok.get(key)
        if v is None:
            v = cls._ok[key] = bool(lib().npvp_linear_bwd_f16_takes(R, N, K)) and _S._cur.chain.takes(N, K, R)
        return v and R <= cls.MAX_ROWS


def linear_bwd(dy, x, w, b, sk, act=0, aux_in=None, drop=NO_DROP, residual=None, dy_amax=None, dx_amax=None, a_drop=NO_DROP):
    """The backward of y = x w^T + b given dy [R, N]: -> dx = epilogue((a_drop mask) dy w), gw, gb - the weight (+ bias) gradient goes
    into the sink `sk` (-> None, None) or is returned.  One launch where FusedLinearBwd takes the shape, else the dgrad on this
    stream and the weight gradient on the gradient stream as before."""
    R, N = dy.shape
    K = w.shape[1]
    has_b = b is not None
    if (FusedLinearBwd.enabled and GEMM_PRECISION == 6 and sk and has_b == (sk[1] is not None) and FusedLinearBwd.takes(R, N, K)
            and (FusedLinearBwd.with_gradient_stream or not _S._cur.wgrad.enabled) and not _S._cur.range_guard.fallback and not _S._cur.range_guard.strict and dy.stride(1) == 1 and x.stride(1) == 1):
        pl = _planes(w, "D", R)
        gw = sk[0][0]
        if pl is not None and gw.stride(1) == 1:
            planes, w_amax = pl
            gb = sk[1][0] if has_b else None
            dev = dy.device
            dy_amax, x_amax = amax_of(dy, dy_amax), amax_of(x)
            dx = torch.empty(R, K, dtype=torch.float32, device=dev)
            ws, wsn = _ws(_S._cur.chain._wsb[(N, K, R)], dev)
            st = _stream()
            seed = _S._cur.rng.seed_tensor(dev) if (drop.on or a_drop.on) else None
            outs = (gw.data_ptr(),) if gb is None else (gw.data_ptr(), gb.data_ptr())
            chain = not _S._cur.wgrad.enabled
            if chain:
                job = ctypes.create_string_buffer(64)
                prev = _S._cur.chain._pending.pop(st, None)
                job_addr, prev_addr = ctypes.addressof(job), (ctypes.addressof(prev[0]) if prev is not None else None)
            else:
                job_addr, prev_addr = _S._cur.reduce.splitk_slot(outs), None
            probe = GemmProbe.armed and (GemmProbe.only is None or 8 in GemmProbe.only)
            if probe:
                e0, e1 = probe_pair()
            check(lib().npvp_linear_bwd_f16(R, N, K, dy.data_ptr(), dy.stride(0), dy_amax.data_ptr(), planes.data_ptr(), w_amax.data_ptr(),
                                            dx.data_ptr(), dx.stride(0), act, _ptr(aux_in), _ptr(residual),
                                            0 if residual is None else residual.stride(0), drop.p, drop.mode, drop.g1, drop.g2, drop.salt,
                                            _ptr(dx_amax), x.data_ptr(), x.stride(0), x_amax.data_ptr(), gw.data_ptr(), gw.stride(0),
                                            _ptr(gb), _ptr(_S._cur.range_guard.flag(dev)), a_drop.p, a_drop.g1, a_drop.g2, a_drop.salt, _ptr(seed),
                                            prev_addr, job_addr, ws.data_ptr(), wsn, st), "npvp_linear_bwd_f16")
            if probe:
                e1.record()
                GemmProbe.records.append((e0, e1, 4.0 * R * N * K, 4.0 * (2 * R * N + 2 * R * K + 2 * N * K), ((1, 0), 8)))
            if chain:
                if prev is not None and prev[4] is not None:
                    _S._cur.grad_sink.wrote(*prev[4])            # (its reduction rode in the launch just enqueued)
                _S._cur.chain._pending[st] = (job, ws, gw, gb, sk)
                _S._cur.reduce._arm()                      # (the backward pass's end runs the last one: ReduceQueue.finish)
            else:
                _S._cur.reduce.splitk_added(outs, ws, sk)
            return dx, None, None
    dx = linear_dgrad(dy, w, act=act, aux_in=aux_in, drop=drop, residual=residual, dy_amax=dy_amax, dx_amax=dx_amax, a_drop=a_drop)
    gw, gb = _lin_grads(dy, x, w, b, sk, a_drop=a_drop)
    return dx, gw, gb


def colsum(x):
    R, N = x.shape
    L = lib()
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    ws, wsn = _ws(L.npvp_colsum_workspace_bytes(R, N), x.device)
    check(L.npvp_colsum(_ptr(x), R, N, x.stride(0), _ptr(out), 0, _ptr(ws), wsn, _stream()), "npvp_colsum")
    return out


def drop_apply(x2d, drop):
    out = torch.empty_like(x2d)
    slot = _new_slot(x2d.device)
    check(lib().npvp_drop_apply(_ptr(x2d), _ptr(out), x2d.shape[0], x2d.shape[1], drop.p, drop.mode, drop.g1, drop.g2,
                                _ptr(_S._cur.rng.seed_tensor(x2d.device)), drop.salt, _ptr(slot), _stream()), "npvp_drop_apply")
    return tag_amax(out, slot)


def transpose(x):
    """[B,R,C] -> [B,C,R]"""
    _chk(x)
    x = _c(x)
    B, R, C = x.shape
    out = torch.empty(B, C, R, dtype=torch.float32, device=x.device)
    check(lib().npvp_transpose(_ptr(x), _ptr(out), B, R, C, _stream()), "npvp_transpose")
    return out


def reduce_mid(x, scale=1.0, out=None, accumulate=False):
    """[A,B,C] -> [A,C]: scale * sum over B (into `out`, added to it with accumulate)"""
    _chk(x)
    x = _c(x)
    A, B, C = x.shape
    if out is None:
        out = torch.empty(A, C, dtype=torch.float32, device=x.device)
    check(lib().npvp_reduce_mid(_ptr(x), _ptr(out), A, B, C, scale, 1 if accumulate else 0, _stream()), "npvp_reduce_mid")
    return out


class ActSink:
    """The gradient of an activation that feeds MANY sub-layers - a positional table (~30 per step), the event latent (16), the
    decoder's memory and its fused key (8 each) - summed IN PLACE by its consumers' backward kernels (the `accumulate` flag of
    npvp_posfuse_bwd / npvp_reduce_mid / the dgrad GEMM) instead of by one autograd add kernel per consumer (75 launches per
    step).  `fan_out(t)` returns t as seen through an identity node that owns a sink; a consumer that finds `t._npvp_sink` writes
    its share into `sink.target(...)` and returns None for that input; the identity node's backward hands the buffer on (plus
    whatever consumers without the in-place route returned the usual way)."""
    __slots__ = ("buf",)
    enabled = True

    def __init__(self):
        self.buf = None

    def target(self, shape, dev):
        """-> (buffer, accumulate): the first contribution of a backward pass writes, the others add"""
        if self.buf is None:
            self.buf = torch.empty(shape, dtype=torch.float32, device=dev)
            return self.buf, False
        if self.buf.numel() != torch.Size(shape).numel():
            raise RuntimeError(f"ActSink: a consumer contributes a gradient of shape {tuple(shape)} to a buffer of shape {tuple(self.buf.shape)}")
        return self.buf, True

    @staticmethod
    def of(t):
        return None if t is None else getattr(t, "_npvp_sink", None)


class _FanOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, sink):
        ctx.sink = sink
        ctx.set_materialize_grads(False)
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        buf, ctx.sink.buf = ctx.sink.buf, None
        if buf is None:
            return g, None
        if g is not None:
            buf.add_(g.reshape(buf.shape))
        return buf, None


def fan_out(t):
    """t for consumers that can sum their gradient contributions in place (ActSink); anything else sees a plain tensor"""
    if t is None or not (ActSink.enabled and t.is_cuda and t.requires_grad and torch.is_grad_enabled()):
        return t
    sink = ActSink()
    out = _FanOut.apply(t, sink)
    out._npvp_sink = sink
    return out


def broadcast_mid(x, B, scale=1.0):
    """[A,C] -> [A,B,C]"""
    _chk(x)
    x = _c(x)
    A, C = x.shape
    out = torch.empty(A, B, C, dtype=torch.float32, device=x.device)
    check(lib().npvp_broadcast_mid(_ptr(x), _ptr(out), A, B, C, scale, _stream()), "npvp_broadcast_mid")
    return out


# --------------------------------------------------------------------------- autograd Functions
class _Transpose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        remember(ctx)
        return transpose(x)

    @scoped
    def backward(ctx, g):
        return transpose(g)


class _MeanMid(torch.autograd.Function):
    """[A,B,C] -> [A,C] mean over B (event coding = mean over T, ref/models/Predictor.py:346)"""

    @staticmethod
    def forward(ctx, x):
        remember(ctx)
        ctx.B = x.shape[1]
        return reduce_mid(x, 1.0 / x.shape[1])

    @scoped
    def backward(ctx, g):
        return broadcast_mid(g, ctx.B, 1.0 / ctx.B)


class _L1Mean(torch.autograd.Function):
    """lam * mean|a - b| (ref/models/criterion.py:99-121 with norm_dim None); the gradient goes to `a` only - the second operand of
    the Stage-2 losses is the ground truth.  One fused pass forward, one backward, and no torch reduction: a captured step must not
    contain memset nodes (npvp_hip.h npvp_l1_mean)."""

    @staticmethod
    def forward(ctx, a, b, lam):
        remember(ctx)
        _chk(a, b)
        if a.shape != b.shape:
            raise RuntimeError(f"l1_mean: shapes differ: {tuple(a.shape)} / {tuple(b.shape)}")
        a, b = _c(a), _c(b)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws, wsn = _ws(4096, a.device)
        check(lib().npvp_l1_mean(_ptr(a), _ptr(b), a.numel(), float(lam), _ptr(out), _ptr(ws), wsn, _stream()), "npvp_l1_mean")
        ctx.save_for_backward(a, b)
        ctx.lam = float(lam)
        return out

    @scoped
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        _chk(g)
        da = torch.empty_like(a)
        check(lib().npvp_l1_mean_bwd(_ptr(a), _ptr(b), a.numel(), _ptr(_c(g)), ctx.lam, _ptr(da), _stream()), "npvp_l1_mean_bwd")
        return da, None, None


def l1_mean(a, b, lam=1.0):
    if b.requires_grad:
        raise RuntimeError("l1_mean: only the first operand takes a gradient (the Stage-2 losses compare against ground truth)")
    return _L1Mean.apply(a, b, lam)


class _SumAll(torch.autograd.Function):
    """sum of every element -> device scalar, in a fixed order (npvp_hip.h npvp_sum_all)"""

    @staticmethod
    def forward(ctx, x):
        _chk(x)
        ctx.shape = x.shape
        x = _c(x)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        ws, wsn = _ws(4096, x.device)
        check(lib().npvp_sum_all(_ptr(x), x.numel(), _ptr(out), _ptr(ws), wsn, _stream()), "npvp_sum_all")
        return out

    @staticmethod
    def backward(ctx, g):
        return g.expand(ctx.shape)


def sum_all(x):
    return _SumAll.apply(x)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, relu):
        remember(ctx)
        _chk(x, w, b)
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        slot = _new_slot(x.device)
        check(lib().npvp_layernorm_fwd(_ptr(x2), _ptr(w), _ptr(b), _ptr(y), _ptr(mean), _ptr(rstd), rows, C, eps,
                                       int(relu), _ptr(slot), _stream()), "npvp_layernorm_fwd")
        tag_amax(y, slot)
        ctx.save_for_backward(x2, w, b, mean, rstd)
        ctx.relu, ctx.shape = int(relu), x.shape
        ctx.sink = _ln_sink(w, b)
        return y.reshape(x.shape)

    @scoped
    def backward(ctx, dy):
        x2, w, b, mean, rstd = ctx.saved_tensors
        rows, C = x2.shape
        dy2 = _c(dy).reshape(rows, C)
        L = lib()
        dx = torch.empty_like(x2)
        sk = ctx.sink
        dw, db = (sk[0][0], sk[1][0]) if sk else (torch.empty_like(w), torch.empty_like(b))
        ws, wsn = _ws(L.npvp_layernorm_bwd_workspace_bytes(rows, C), x2.device)
        slot = _new_slot(x2.device)
        check(L.npvp_layernorm_bwd(_ptr(dy2), _ptr(x2), _ptr(w), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dw),
                                   _ptr(db), rows, C, ctx.relu, _p(0), _sink_mode(sk), _ptr(slot), _ptr(ws), wsn, _stream()),
              "npvp_layernorm_bwd")
        tag_amax(dx, slot)
        if sk:
            _sunk_ln_reduce(sk, ws, rows, C)
            return dx.reshape(ctx.shape), None, None, None, None
        return dx.reshape(ctx.shape), dw, db, None, None


def _ln_sink(w, b):
    sw, sb = _S._cur.grad_sink.slot(w), _S._cur.grad_sink.slot(b)
    return (sw, sb) if (sw is not None and sb is not None) else None


def _sink_mode(sk):
    """`accumulate` argument of the norm backward entry points: 0 plain outputs, 1 accumulate into the gradient slots
    on this stream, 2 leave the partial sums in the workspace - their reduction into the slots then runs on the
    gradient stream (WgradStream), where EVERY in-place gradient write is serialised."""
    return 0 if not sk else (2 if (_S._cur.wgrad.enabled or _S._cur.reduce.enabled) else 1)


def _sunk_ln_reduce(sk, ws, rows, C):
    gw, gb = sk[0][0], sk[1][0]
    if _S._cur.reduce.enabled:
        _S._cur.reduce.add(lib().npvp_layernorm_bwd_reduce_job, "npvp_layernorm_bwd_reduce_job", (_ptr(ws), _ptr(gw), _ptr(gb), rows, C, 1),
                        (gw.data_ptr(), gb.data_ptr()), ws, sk)
    elif _S._cur.wgrad.enabled:
        # (deferred: every value is bound NOW, the closure runs a few launches later)
        _S._cur.wgrad.run(lambda ws=ws, gw=gw, gb=gb, rows=rows, C=C: check(
            lib().npvp_layernorm_bwd_reduce(_ptr(ws), _ptr(gw), _ptr(gb), rows, C, 1, _stream()), "npvp_layernorm_bwd_reduce"), ws, wrote=sk)
    else:
        _S._cur.grad_sink.wrote(*sk)


def _sunk_fln_reduce(sk, ws, dw, db, frames, PF):
    """the frame-LayerNorm parameter-gradient partials left in `ws` (accumulate mode 2) -> the gradient slices"""
    if _S._cur.reduce.enabled:
        _S._cur.reduce.add(lib().npvp_frameln_act_bwd_reduce_job, "npvp_frameln_act_bwd_reduce_job", (_ptr(ws), _ptr(dw), _ptr(db), frames, PF, 1),
                        (dw.data_ptr(), db.data_ptr()), ws, sk)
    elif _S._cur.wgrad.enabled:
        _S._cur.wgrad.run(lambda ws=ws, dw=dw, db=db, frames=frames, PF=PF: check(
            lib().npvp_frameln_act_bwd_reduce(_ptr(ws), _ptr(dw), _ptr(db), frames, PF, 1, _stream()), "npvp_frameln_act_bwd_reduce"),
            ws, wrote=sk)
    else:
        _S._cur.grad_sink.wrote(*sk)


def layernorm(x, w, b, eps=1e-5, relu=False):
    return _LayerNorm.apply(x, w, b, eps, relu)


class _LayerNormNchw(torch.autograd.Function):
    """LayerNorm(C) (+ReLU) over the token rows of canonical x [F, 64, C], result in the reference's (N,T,C,H,W) layout
    (ref VidHRFormer.py:150-159): one forward kernel instead of LayerNorm + transpose."""

    @staticmethod
    def forward(ctx, x, w, b, eps, relu, N, T, H, W):
        remember(ctx)
        _chk(x, w, b)
        C = x.shape[-1]
        x3 = _c(x).reshape(N * T, H * W, C)
        out = torch.empty(N, T, C, H, W, dtype=torch.float32, device=x.device)
        mean = torch.empty(N * T * H * W, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        check(lib().npvp_layernorm_nchw_fwd(_ptr(x3), _ptr(w), _ptr(b), _ptr(out), _ptr(mean), _ptr(rstd), N * T, H * W, C, eps,
                                            int(relu), _stream()), "npvp_layernorm_nchw_fwd")
        ctx.save_for_backward(x3, w, b, mean, rstd)
        ctx.relu, ctx.shape = int(relu), x.shape
        ctx.sink = _ln_sink(w, b)
        return out

    @scoped
    def backward(ctx, dy):
        # dy (N,T,C,H,W) -> canonical rows (the LDS-tiled transpose), then the row-wise LayerNorm backward: a fused backward
        # through the forward kernel's tile was 3x slower than these two kernels
        x3, w, b, mean, rstd = ctx.saved_tensors
        F_, P, C = x3.shape
        L = lib()
        dy = _c(dy)
        dy2 = torch.empty(F_ * P, C, dtype=torch.float32, device=dy.device)
        check(L.npvp_transpose(_ptr(dy), _ptr(dy2), F_, C, P, _stream()), "npvp_transpose")
        rows = F_ * P
        x2 = x3.reshape(rows, C)
        dx = torch.empty_like(x2)
        sk = ctx.sink
        dw, db = (sk[0][0], sk[1][0]) if sk else (torch.empty_like(w), torch.empty_like(b))
        ws, wsn = _ws(L.npvp_layernorm_bwd_workspace_bytes(rows, C), x2.device)
        slot = _new_slot(x2.device)
        check(L.npvp_layernorm_bwd(_ptr(dy2), _ptr(x2), _ptr(w), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dw), _ptr(db),
                                   rows, C, ctx.relu, _p(0), _sink_mode(sk), _ptr(slot), _ptr(ws), wsn, _stream()), "npvp_layernorm_bwd")
        tag_amax(dx, slot)
        if sk:
            _sunk_ln_reduce(sk, ws, rows, C)
            return (dx.reshape(ctx.shape),) + (None,) * 8
        return (dx.reshape(ctx.shape), dw, db) + (None,) * 6


def layernorm_nchw_supported(x, H, W):
    return x.is_cuda and H * W == 64 and x.shape[-1] in (256, 512)


def layernorm_nchw(x, w, b, eps, relu, N, T, H, W):
    """canonical (N,T,H,W,C) -> LayerNorm (+ReLU) -> (N,T,C,H,W)"""
    return _LayerNormNchw.apply(x, w, b, eps, relu, N, T, H, W)


class _LayerNormRes(torch.autograd.Function):
    """(x, LN(x)) for the pre-norm residual pattern  x + f(LN(x))  of every sub-layer (ref VidHRFormer.py:87-112).
    Returning x through the Function lets backward fold the residual branch's gradient into the LayerNorm backward
    kernel (dx = d_residual + LN'(dy)) instead of leaving a separate [R, C] add to autograd."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        remember(ctx)
        _chk(x, w, b)
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        slot = _new_slot(x.device)
        check(lib().npvp_layernorm_fwd(_ptr(x2), _ptr(w), _ptr(b), _ptr(y), _ptr(mean), _ptr(rstd), rows, C, eps, 0,
                                       _ptr(slot), _stream()), "npvp_layernorm_fwd")
        tag_amax(y, slot)
        ctx.save_for_backward(x2, w, b, mean, rstd)
        ctx.shape = x.shape
        ctx.sink = _ln_sink(w, b)
        return x.view_as(x), y.reshape(x.shape)

    @scoped
    def backward(ctx, dres, dy):
        x2, w, b, mean, rstd = ctx.saved_tensors
        rows, C = x2.shape
        L = lib()
        if dy is None:
            return dres, None, None, None
        sk = ctx.sink
        dx = torch.empty_like(x2)
        dw, db = (sk[0][0], sk[1][0]) if sk else (torch.empty_like(w), torch.empty_like(b))
        ws, wsn = _ws(L.npvp_layernorm_bwd_workspace_bytes(rows, C), x2.device)
        dy2 = _c(dy).reshape(rows, C)
        dr2 = None if dres is None else _c(dres).reshape(rows, C)
        slot = _new_slot(x2.device)
        check(L.npvp_layernorm_bwd(_ptr(dy2), _ptr(x2), _ptr(w), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dw),
                                   _ptr(db), rows, C, 0, _ptr(dr2), _sink_mode(sk), _ptr(slot), _ptr(ws), wsn, _stream()),
              "npvp_layernorm_bwd")
        tag_amax(dx, slot)
        if sk:
            _sunk_ln_reduce(sk, ws, rows, C)
            return dx.reshape(ctx.shape), None, None, None
        return dx.reshape(ctx.shape), dw, db, None


def layernorm_res(x, w, b, eps=1e-5):
    """returns (x, LN(x)); use the returned x as the residual operand"""
    return _LayerNormRes.apply(x, w, b, eps)


class _PosFuse(torch.autograd.Function):
    """y = GroupNorm1(x + add) * (1 + gamma) + beta on [N*T, PF] frames (ref submodules.py:432-454)."""

    @staticmethod
    def forward(ctx, x, add, beta, gamma, N, T):
        remember(ctx)
        _chk(x, add, beta, gamma)
        x = _c(x)
        F_, PF = N * T, x.numel() // (N * T)
        beta = _c(beta)
        add_c = None if add is None else _c(add)
        gamma_c = None if gamma is None else _c(gamma)
        y = torch.empty_like(x)
        mean = torch.empty(F_, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        slot = _new_slot(x.device)
        check(lib().npvp_posfuse_fwd(_ptr(x), _ptr(add_c), _ptr(beta), _ptr(gamma_c), _ptr(y), _ptr(mean), _ptr(rstd), N, T,
                                     PF, 1e-5, _ptr(slot), _stream()), "npvp_posfuse_fwd")
        tag_amax(y, slot)
        ctx.save_for_backward(x, add_c, gamma_c, mean, rstd)
        ctx.N, ctx.T, ctx.PF = N, T, PF
        ctx.beta_shape = beta.shape
        return y

    @scoped
    def backward(ctx, dy):
        x, add, gamma, mean, rstd = ctx.saved_tensors
        N, T, PF = ctx.N, ctx.T, ctx.PF
        dy = _c(dy)
        du, dbeta, dgamma = _posfuse_bwd_call(dy, x, add, gamma, mean, rstd, N, T, PF, ctx.beta_shape, ctx.needs_input_grad[2],
                                              gamma is not None and ctx.needs_input_grad[3])
        dadd = reduce_mid(du.view(N, T, PF)).view(add.shape) if (add is not None and ctx.needs_input_grad[1]) else None
        return du, dadd, dbeta, dgamma, None, None


def _posfuse_bwd_call(dy, x, add, gamma, mean, rstd, N, T, PF, beta_shape, want_beta, want_gamma, beta_sink=None, gamma_sink=None):
    """-> du, dbeta, dgamma: one entry point; the sums over the batch come out of the apply pass when the shape allows
    (npvp_posfuse_bwd_fused), else through a dy*uhat scratch and two reductions inside the library.  With sinks (ActSink) the
    table gradients are accumulated in place and None is returned for them."""
    L = lib()
    du = torch.empty_like(x)
    acc = False
    sunk = beta_sink is not None and want_beta and (not want_gamma or gamma_sink is not None)
    if sunk:
        # (one accumulate flag for both tables: they are sunk together - a fresh pair is written, an existing pair added to)
        dbeta, acc = beta_sink.target(beta_shape, x.device)
        dgamma = None
        if want_gamma:
            dgamma, acc_g = gamma_sink.target(gamma.shape, x.device)
            if acc_g != acc:
                # the kernel takes ONE accumulate flag: a pair out of step (one table sunk alone by some consumer) would have its
                # second table accumulated into uninitialised memory - bring the fresh buffer to zero and accumulate into both
                (dgamma if not acc_g else dbeta).zero_()
                acc = True
    else:
        dbeta = torch.empty(beta_shape, dtype=torch.float32, device=x.device) if want_beta else None
        dgamma = torch.empty(gamma.shape, dtype=torch.float32, device=x.device) if want_gamma else None
    dyxh = torch.empty_like(x) if (want_gamma and not L.npvp_posfuse_bwd_fused(N, T, PF)) else None
    ws, wsn = _ws(8 * N * T, x.device)
    mp, rp = (mean, rstd) if isinstance(mean, int) else (_ptr(mean), _ptr(rstd))        # (tensors or device addresses)
    check(L.npvp_posfuse_bwd(_ptr(dy), _ptr(x), _ptr(add), _ptr(gamma), mp, rp, _ptr(du), _ptr(dyxh), _ptr(dbeta),
                             _ptr(dgamma), N, T, PF, 1 if acc else 0, _ptr(ws), wsn, _stream()), "npvp_posfuse_bwd")
    return (du, None, None) if sunk else (du, dbeta, dgamma)


def posfuse(x, add, beta, gamma, N, T):
    return _PosFuse.apply(x, add, beta, gamma, N, T)


class _PosFuseInstance(torch.autograd.Function):
    """PosFeatFuser with param_free_norm_type='instance' (ref submodules.py:427-431): statistics per (frame, channel) over the
    H*W pixels.  x (N,T,H,W,C) channels-last; beta / gamma [T*H*W, C]; add (N,H,W,C) or None."""

    @staticmethod
    def forward(ctx, x, add, beta, gamma, N, T):
        remember(ctx)
        _chk(x, add, beta, gamma)
        x = _c(x)
        C = x.shape[-1]
        P = x.numel() // (N * T * C)
        beta = _c(beta)
        add_c = None if add is None else _c(add)
        gamma_c = None if gamma is None else _c(gamma)
        y = torch.empty_like(x)
        st = torch.empty(2, N * T, C, dtype=torch.float32, device=x.device)
        slot = _new_slot(x.device)
        check(lib().npvp_posfuse_instance_fwd(_ptr(x), _ptr(add_c), _ptr(beta), _ptr(gamma_c), _ptr(y), _row(st, 0), _row(st, 1), N, T, P,
                                              C, 1e-5, _ptr(slot), _stream()), "npvp_posfuse_instance_fwd")
        tag_amax(y, slot)
        ctx.save_for_backward(x, add_c, gamma_c, st)
        ctx.cfg = (N, T, P, C, beta.shape)
        return y

    @scoped
    def backward(ctx, dy):
        x, add, gamma, st = ctx.saved_tensors
        N, T, P, C, beta_shape = ctx.cfg
        dy = _c(dy)
        du = torch.empty_like(x)
        dyxh = torch.empty_like(x) if (gamma is not None and ctx.needs_input_grad[3]) else None
        check(lib().npvp_posfuse_instance_bwd(_ptr(dy), _ptr(x), _ptr(add), _ptr(gamma), _row(st, 0), _row(st, 1), _ptr(du), _ptr(dyxh),
                                              N, T, P, C, _stream()), "npvp_posfuse_instance_bwd")
        PF = P * C
        dadd = reduce_mid(du.view(N, T, PF)).view(add.shape) if (add is not None and ctx.needs_input_grad[1]) else None
        dbeta = reduce_mid(dy.view(1, N, T * PF)).view(beta_shape) if ctx.needs_input_grad[2] else None
        dgamma = reduce_mid(dyxh.view(1, N, T * PF)).view(gamma.shape) if dyxh is not None else None
        return du, dadd, dbeta, dgamma, None, None


def posfuse_instance(x, add, beta, gamma, N, T):
    return _PosFuseInstance.apply(x, add, beta, gamma, N, T)


class _Linear(torch.autograd.Function):
    """y = residual + drop(x w^T + b): nn.Linear / 1x1 conv / MHA projections with the residual add
    and the dropout / drop-path that follows them in the reference fused into the GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, w, b, residual, drop, frame_stats=False):
        remember(ctx)
        _chk(x, w, b, residual)
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 4 != 0:
            x2 = x2.contiguous()
        w = w if (w.stride(1) == 1 and w.stride(0) % 4 == 0) else w.contiguous()
        r2 = None if residual is None else _c(residual).reshape(-1, w.shape[0])
        ctx.save_for_backward(x2, w)
        ctx.drop, ctx.has_b, ctx.has_r, ctx.xshape = drop, b is not None, residual is not None, x.shape
        ctx.sink = _wb_sink(w, b)
        if frame_stats:
            # the GEMM's epilogue leaves per-(frame, 64-column block) partial statistics of y; a tiny kernel merges them
            R, N = x2.shape[0], w.shape[0]
            frames = R // 64
            part = torch.empty(frames * (N // 64) * 2, dtype=torch.float32, device=x.device)
            y = linear_fwd(x2, w, b, rowstats=part)
            mean = torch.empty(frames, dtype=torch.float32, device=x.device)
            rstd = torch.empty_like(mean)
            check(lib().npvp_frame_stats_finalize(_ptr(part), N // 64, 4096.0, _ptr(mean), _ptr(rstd), frames, 1e-5, _stream()),
                  "npvp_frame_stats_finalize")
            ctx.mark_non_differentiable(mean, rstd)
            ctx.set_materialize_grads(False)        # no zero-filled gradient tensors for the two statistics outputs
            return y.reshape(*x.shape[:-1], N), mean, rstd
        y = linear_fwd(x2, w, b, residual=r2, drop=drop)
        return y.reshape(*x.shape[:-1], w.shape[0])

    @scoped
    def backward(ctx, dy, *_stats_grads):
        x2, w = ctx.saved_tensors
        N = w.shape[0]
        dy2 = _c(dy).reshape(-1, N)
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dz, ad = masked_grad(dy2, ctx.drop, w)          # (a DropPath mask rides in the two GEMMs when they can take it)
        else:
            dz, ad = (drop_apply(dy2, ctx.drop) if ctx.drop.on else dy2), NO_DROP
        dx = linear_dgrad(dz, w, a_drop=ad).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw = db = None
        want_b = ctx.has_b and ctx.needs_input_grad[2]
        sk = ctx.sink
        if sk and ctx.needs_input_grad[1] and want_b == (sk[1] is not None):
            _sunk_wgrad(dz, x2, want_b, sk, a_drop=ad)
        elif ctx.needs_input_grad[1]:
            dw = linear_wgrad(dz, x2, want_b, a_drop=ad)
            if want_b:
                dw, db = dw
        elif want_b:
            db = colsum(dz)
        dres = dy if ctx.has_r else None
        return dx, dw, db, dres, None, None


def _wgrad_slots(dy, x, dy_amax=None, x_amax=None):
    """the amax slots of a weight-gradient GEMM's operands when the fp16 kernel will take it, filled on the CURRENT stream
    (the GEMM itself may run on the gradient stream, which is ordered after this one; a slot first filled over there would
    be read here without any ordering)"""
    if GEMM_PRECISION == 6 and _gemm_kernel_id(0, 0, dy.shape[1], x.shape[1], dy.shape[0], 6, False) == 6:
        return amax_of(dy, dy_amax), amax_of(x, x_amax)
    return None, None


def _sunk_wgrad(dy, x, with_b, sk, dy_amax=None, x_amax=None, a_drop=NO_DROP):
    """accumulate dW (and db) of one linear into its gradient slots - on the wgrad stream when enabled"""
    dy_amax, x_amax = _wgrad_slots(dy, x, dy_amax, x_amax)
    fn = lambda dy=dy, x=x, gw=sk[0][0], gb=(sk[1][0] if with_b else None), a1=dy_amax, a2=x_amax, ad=a_drop: \
        linear_wgrad(dy, x, with_b, into=gw, into_b=gb, dy_amax=a1, x_amax=a2, a_drop=ad)
    if _S._cur.wgrad.enabled:
        _S._cur.wgrad.run(fn, dy, x, wrote=sk, urgent=2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] >= 3e10)
    else:
        fn()
        _S._cur.grad_sink.wrote(*sk)


def _wb_sink(w, b):
    """(weight slot, bias slot or None) when BOTH gradients can be accumulated in place, else None"""
    sw = _S._cur.grad_sink.slot(w)
    if sw is None:
        return None
    if b is None or not b.requires_grad:
        return (sw, None)
    sb = _S._cur.grad_sink.slot(b)
    return (sw, sb) if sb is not None else None


def linear(x, w, b=None, residual=None, drop=NO_DROP, frame_stats=False):
    """frame_stats=True (plain bias-only linear whose rows come in frames of 64): returns (y, mean, rstd), the statistics
    a following frame LayerNorm needs, computed in the GEMM's epilogue"""
    if frame_stats:
        assert residual is None and not drop.on
    return _Linear.apply(x, w, b, residual, drop, bool(frame_stats))


class _FFN(torch.autograd.Function):
    """out = x + drop3(linear2(drop2(GELU(linear1(xn)))))  (ref/models/VidHRFormer.py:110-112,224-226):
    two GEMMs; bias+GELU+dropout and bias+dropout+residual live in their epilogues, and in backward the
    GELU'/dropout product is the epilogue of the dgrad GEMM."""

    @staticmethod
    def forward(ctx, xn, x, w1, b1, w2, b2, p):
        remember(ctx)
        _chk(xn, x, w1, b1, w2, b2)
        C = xn.shape[-1]
        xn2, x2 = _c(xn).reshape(-1, C), _c(x).reshape(-1, C)
        R, Fh = xn2.shape[0], w1.shape[0]
        d2, d3 = Drop(p), Drop(p)
        h = torch.empty(R, Fh, dtype=torch.float32, device=xn.device)
        a_slot = _new_slot(xn.device)
        a = tag_amax(linear_fwd(xn2, w1, b1, act=1, aux_out=h, drop=d2, y_amax=a_slot), a_slot)
        y = linear_fwd(a, w2, b2, residual=x2, drop=d3)
        ctx.save_for_backward(xn2, h, a, w1, w2)
        ctx.d2, ctx.d3, ctx.shape = d2, d3, x.shape
        s1, s2 = _wb_sink(w1, b1), _wb_sink(w2, b2)
        ctx.sink = (s1, s2) if (s1 and s2 and s1[1] is not None and s2[1] is not None) else None
        return y.reshape(x.shape)

    @scoped
    def backward(ctx, dy):
        xn2, h, a, w1, w2 = ctx.saved_tensors
        C = xn2.shape[1]
        dy2 = _c(dy).reshape(-1, C)
        dz2, ad = masked_grad(dy2, ctx.d3, w2)
        sk = ctx.sink
        if sk:
            dh = linear_dgrad(dz2, w2, act=3, aux_in=h, drop=ctx.d2, a_drop=ad)
            _sunk_wgrad(dz2, a, True, sk[1], a_drop=ad)
            dxn = linear_dgrad(dh, w1)
            _sunk_wgrad(dh, xn2, True, sk[0])
            return dxn.reshape(ctx.shape), dy, None, None, None, None, None
        dw2, db2 = linear_wgrad(dz2, a, True, a_drop=ad)
        dh = linear_dgrad(dz2, w2, act=3, aux_in=h, drop=ctx.d2, a_drop=ad)
        dw1, db1 = linear_wgrad(dh, xn2, True)
        dxn = linear_dgrad(dh, w1)
        return dxn.reshape(ctx.shape), dy, dw1, db1, dw2, db2, None


def ffn(xn, x, w1, b1, w2, b2, p):
    return _FFN.apply(xn, x, w1, b1, w2, b2, p)


class AttnCfg:
    __slots__ = ("mode", "dim0", "P", "W", "ws", "Tq", "Tk", "heads", "mask_mode", "drop")

    def __init__(self, mode, dim0, P, W, ws, Tq, Tk, heads, mask_mode, p_drop):
        self.mode, self.dim0, self.P, self.W, self.ws, self.Tq, self.Tk = mode, dim0, P, W, ws, Tq, Tk
        self.heads, self.mask_mode = heads, mask_mode
        self.drop = Drop(p_drop)


def _attn_fwd(q, k, v, o, cfg):
    hd = o.shape[1] // cfg.heads
    seed = _S._cur.rng.seed_tensor(q.device) if cfg.drop.on else None
    if DropRecorder.sites is not None and cfg.drop.on:      # weights tensor [groups, heads, L, S]
        L_, S_ = (cfg.ws * cfg.ws, cfg.ws * cfg.ws) if cfg.mode == 0 else (cfg.Tq, cfg.Tk)
        groups = cfg.dim0 * (cfg.P // (cfg.ws * cfg.ws)) if cfg.mode == 0 else cfg.dim0 * cfg.P
        DropRecorder.note(cfg.drop, "elem", groups * cfg.heads * L_ * S_)
    slot = _new_slot(q.device)
    check(lib().npvp_attn_fwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(o), o.stride(0),
                              cfg.mode, cfg.dim0, cfg.P, cfg.W, cfg.ws, cfg.Tq, cfg.Tk, cfg.heads, hd, cfg.mask_mode,
                              cfg.drop.p, _ptr(seed), cfg.drop.salt, _ptr(slot), _stream()), "npvp_attn_fwd")
    tag_amax(o, slot)


def _attn_bwd(q, k, v, go, dq, dk, dv, cfg, packed=None):
    """packed: the [R, 2C] tensor dq and dk are the halves of (one amax slot for both, tagged on it)"""
    hd = go.shape[1] // cfg.heads
    seed = _S._cur.rng.seed_tensor(q.device) if cfg.drop.on else None
    sq = _new_slot(q.device)
    sk = sq if packed is not None else _new_slot(q.device)
    sv = _new_slot(q.device)
    check(lib().npvp_attn_bwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(go), go.stride(0),
                              _ptr(dq), dq.stride(0), _ptr(dk), dk.stride(0), _ptr(dv), dv.stride(0), cfg.mode, cfg.dim0,
                              cfg.P, cfg.W, cfg.ws, cfg.Tq, cfg.Tk, cfg.heads, hd, cfg.mask_mode, cfg.drop.p, _ptr(seed),
                              cfg.drop.salt, _ptr(sq), _ptr(sk), _ptr(sv), _stream()), "npvp_attn_bwd")
    if packed is not None:
        tag_amax(packed, sq)
    else:
        tag_amax(dq, sq); tag_amax(dk, sk)
    tag_amax(dv, sv)


class _AttnPacked(torch.autograd.Function):
    """Self-attention core on a packed [R, 2C] q|k projection and a [R, C] v projection."""

    @staticmethod
    def forward(ctx, qk, v, cfg):
        remember(ctx)
        _chk(qk, v)
        C = v.shape[1]
        o = torch.empty_like(v)
        _attn_fwd(qk[:, :C], qk[:, C:], v, o, cfg)
        ctx.save_for_backward(qk, v)
        ctx.cfg = cfg
        return o

    @scoped
    def backward(ctx, go):
        qk, v = ctx.saved_tensors
        C = v.shape[1]
        go = _c(go)
        dqk, dv = torch.empty_like(qk), torch.empty_like(v)
        _attn_bwd(qk[:, :C], qk[:, C:], v, go, dqk[:, :C], dqk[:, C:], dv, ctx.cfg, packed=dqk)
        return dqk, dv, None


class _Attn(torch.autograd.Function):
    """Attention core with separate q [Rq,C], k [Rk,C], v [Rk,C] (the enc-dec site)."""

    @staticmethod
    def forward(ctx, q, k, v, cfg):
        remember(ctx)
        _chk(q, k, v)
        o = torch.empty_like(q)
        _attn_fwd(q, k, v, o, cfg)
        ctx.save_for_backward(q, k, v)
        ctx.cfg = cfg
        return o

    @scoped
    def backward(ctx, go):
        q, k, v = ctx.saved_tensors
        go = _c(go)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _attn_bwd(q, k, v, go, dq, dk, dv, ctx.cfg)
        return dq, dk, dv, None


def attn_packed(qk, v, cfg):
    return _AttnPacked.apply(qk, v, cfg)


def attn(q, k, v, cfg):
    return _Attn.apply(q, k, v, cfg)


class _FrameLnAct(torch.autograd.Function):
    """out = res + droppath_n(drop(GELU(LayerNorm((Ch,H,W))(h))))   (ref VidHRFormer.py:381-382,384-386,388-390)"""

    @staticmethod
    def forward(ctx, h, w_cl, b_cl, res, frames, p_drop, p_dp, frames_per_sample, mean=None, rstd=None):
        remember(ctx)
        _chk(h, w_cl, b_cl, res)
        h = _c(h)
        PF = h.numel() // frames
        w_cl, b_cl = _c(w_cl), _c(b_cl)
        res_c = None if res is None else _c(res)
        L = lib()
        if mean is None:        # statistics not supplied by the producer of h (dwconv3x3(..., want_stats=True) emits them)
            mean = torch.empty(frames, dtype=torch.float32, device=h.device)
            rstd = torch.empty_like(mean)
            check(L.npvp_frame_stats(_ptr(h), _p(0), _ptr(mean), _ptr(rstd), frames, 1, PF, 1e-5, _stream()), "npvp_frame_stats")
        d, dp = Drop(p_drop), Drop(p_dp, 1)
        DropRecorder.note(d, "elem", h.numel())
        DropRecorder.note(dp, "group", frames // max(1, frames_per_sample))
        out = torch.empty_like(h)
        seed = _S._cur.rng.seed_tensor(h.device) if (d.on or dp.on) else None
        slot = _new_slot(h.device)
        check(L.npvp_frameln_act_fwd(_ptr(h), _ptr(mean), _ptr(rstd), _ptr(w_cl), _ptr(b_cl), _ptr(res_c), _ptr(out), frames,
                                     PF, d.p, d.salt, dp.p, dp.salt, frames_per_sample, _ptr(seed), _ptr(slot), _stream()),
              "npvp_frameln_act_fwd")
        tag_amax(out, slot)
        ctx.save_for_backward(h, mean, rstd, w_cl, b_cl)
        ctx.cfg = (frames, PF, d, dp, frames_per_sample, res is not None)
        ctx.sink = _ln_sink(w_cl, b_cl)
        return out

    @scoped
    def backward(ctx, dout):
        h, mean, rstd, w_cl, b_cl = ctx.saved_tensors
        frames, PF, d, dp, fps, has_res = ctx.cfg
        dout = _c(dout)
        L = lib()
        dh = torch.empty_like(h)
        sk = ctx.sink
        dw, db = (sk[0][0], sk[1][0]) if sk else (torch.empty_like(w_cl), torch.empty_like(b_cl))
        ws, wsn = _ws(L.npvp_frameln_act_bwd_workspace_bytes(frames, PF), h.device)
        seed = _S._cur.rng.seed_tensor(h.device) if (d.on or dp.on) else None
        slot = _new_slot(h.device)
        check(L.npvp_frameln_act_bwd(_ptr(dout), _ptr(h), _ptr(mean), _ptr(rstd), _ptr(w_cl), _ptr(b_cl), _ptr(dh), _ptr(dw),
                                     _ptr(db), frames, PF, d.p, d.salt, dp.p, dp.salt, fps, _ptr(seed), _sink_mode(sk), _ptr(slot),
                                     _ptr(ws), wsn, _stream()), "npvp_frameln_act_bwd")
        tag_amax(dh, slot)
        if sk:
            _sunk_fln_reduce(sk, ws, dw, db, frames, PF)
            dw = db = None
        return dh, dw, db, (dout if has_res else None), None, None, None, None, None, None


def frameln_act(h, w_cl, b_cl, res, frames, p_drop=0.0, p_dp=0.0, frames_per_sample=1, stats=None):
    """stats = (mean, rstd) per frame when the producer of h already has them"""
    mean, rstd = stats if stats is not None else (None, None)
    return _FrameLnAct.apply(h, w_cl, b_cl, res, frames, p_drop, p_dp, frames_per_sample, mean, rstd)


class _DwConv(torch.autograd.Function):
    """Depthwise 3x3 on [F, H*W, Ch]; wtb = [10, Ch]: 9 tap-major weight rows + the bias row."""

    @staticmethod
    def forward(ctx, a, wtb, frames, H, W, want_stats):
        remember(ctx)
        _chk(a, wtb)
        a, wtb = _c(a), _c(wtb)
        Ch = wtb.shape[1]
        out = torch.empty_like(a)
        ctx.save_for_backward(a, wtb)
        ctx.cfg = (frames, H, W, Ch)
        if want_stats:
            mean = torch.empty(frames, dtype=torch.float32, device=a.device)
            rstd = torch.empty_like(mean)
            ws, wsn = _ws(frames * (Ch // 1024) * 8, a.device)
            check(lib().npvp_dwconv3x3_stats(_ptr(a), _ptr(wtb), _row(wtb, 9), _ptr(out), _ptr(mean), _ptr(rstd), frames, H, W,
                                             Ch, 1e-5, _ptr(ws), wsn, _stream()), "npvp_dwconv3x3_stats")
            ctx.mark_non_differentiable(mean, rstd)
            ctx.set_materialize_grads(False)
            return out, mean, rstd
        check(lib().npvp_dwconv3x3(_ptr(a), _ptr(wtb), _row(wtb, 9), _ptr(out), frames, H, W, Ch, 0, _stream()),
              "npvp_dwconv3x3")
        return out

    @scoped
    def backward(ctx, dout, *_unused):
        a, wtb = ctx.saved_tensors
        frames, H, W, Ch = ctx.cfg
        dout = _c(dout)
        L = lib()
        da = torch.empty_like(a)
        check(L.npvp_dwconv3x3(_ptr(dout), _ptr(wtb), _p(0), _ptr(da), frames, H, W, Ch, 1, _stream()), "npvp_dwconv3x3(bwd)")
        dwtb = torch.empty_like(wtb)
        ws, wsn = _ws(L.npvp_dwconv3x3_wgrad_workspace_bytes(frames, Ch), a.device)
        check(L.npvp_dwconv3x3_wgrad(_ptr(a), _ptr(dout), _ptr(dwtb), frames, H, W, Ch, _ptr(ws), wsn, _stream()),
              "npvp_dwconv3x3_wgrad")
        return da, dwtb, None, None, None, None


def dwconv3x3(a, wtb, frames, H, W, want_stats=False):
    """want_stats (8x8 grid, Ch % 1024 == 0): returns (out, mean, rstd) - the frame-LayerNorm statistics of the output"""
    return _DwConv.apply(a, wtb, frames, H, W, bool(want_stats))


class _MlpDwbn(torch.autograd.Function):
    """The whole conv feed-forward sub-layer body of the reference's MlpDWBN (ref/models/VidHRFormer.py:374-392) as ONE
    autograd node:  out = res + droppath(drop(GELU(norm3(fc2(drop(GELU(norm2(dw3x3(GELU(norm1(fc1(x))))))))))))
    Forward = 6 launches + 3 tiny statistics merges: fc1 GEMM (frame statistics in its epilogue) -> fused middle (norm1 +
    GELU + depthwise 3x3 + norm2's statistics: a1 is never written) -> norm2 + GELU + dropout -> fc2 GEMM (statistics) ->
    norm3 + GELU + dropout + residual + drop-path.  Backward mirrors it; the fused middle's backward recomputes a1 from h1 for
    the depthwise weight gradient and emits norm1's backward statistics, so norm1's backward is one pass.
    Passes over the [R, 2048] hidden tensor: forward 6 (was 8), backward 15 (was 18); one Python autograd node instead of 9."""

    @staticmethod
    def forward(ctx, x, res, w1, b1, n1w, n1b, dww, dwb, n2w, n2b, w2, b2, n3w, n3b, frames, T, p_drop, p_dp):
        remember(ctx)
        _chk(x, res, w1, b1, w2, b2)
        L = lib()
        R, C = x.shape
        hid, Co = w1.shape[0], w2.shape[0]
        dev, f32 = x.device, torch.float32
        st = _stream()
        PFh, PFo = 64 * hid, 64 * Co
        # fc1 (+ frame statistics of h1)
        h1 = torch.empty(R, hid, dtype=f32, device=dev)
        part = torch.empty(frames * (hid // 64) * 2, dtype=f32, device=dev)
        gemm(1, 1, R, hid, C, x, x.stride(0), w1, w1.stride(0), h1, bias=b1, b_pre=_planes(w1, "F", R),
             rowstats=part)
        stats = torch.empty(6, frames, dtype=f32, device=dev)               # mean1, rstd1, mean2, rstd2, mean3, rstd3
        # (no statistics launches: each consumer below merges the partial (mean, M2) pairs its producer left and writes
        # mean / rstd into `stats` for backward)
        # tap-major depthwise weights [9][hid] + bias row: rebuilt when the parameters changed (optimiser step / in-place update),
        # not per call (one launch per MlpDWBN and step)
        key = (WeightPlanes.epoch, dww._version, dwb._version, dww.data_ptr(), dwb.data_ptr())
        hit = dww.__dict__.get("_npvp_wtb")
        if hit is not None and hit[0] == key:
            wtb = hit[1]
        else:
            wtb = torch.empty(10, hid, dtype=f32, device=dev)
            check(L.npvp_dwtb_build(_ptr(_c(dww)), _ptr(dwb), _ptr(wtb), hid, st), "npvp_dwtb_build")
            dww.__dict__["_npvp_wtb"] = (key, wtb)
        # fused middle
        h2 = torch.empty(R, hid, dtype=f32, device=dev)
        part2 = torch.empty(frames * (hid // 512) * 2, dtype=f32, device=dev)
        check(L.npvp_mlpdw_mid_fwd_parts(_ptr(h1), _ptr(part), hid // 64, 4096.0, _row(stats, 0), _row(stats, 1), _ptr(n1w), _ptr(n1b),
                                         _ptr(wtb), _row(wtb, 9), _ptr(h2), _ptr(part2), frames, 8, 8, hid, 1e-5, st),
              "npvp_mlpdw_mid_fwd_parts")
        d2, d3, dp = Drop(p_drop), Drop(p_drop), Drop(p_dp, 1)
        DropRecorder.note(d2, "elem", R * hid)
        seed = _S._cur.rng.seed_tensor(dev) if (d2.on or dp.on) else None
        a2 = torch.empty(R, hid, dtype=f32, device=dev)
        a2_slot = _new_slot(dev)
        check(L.npvp_frameln_act_fwd_parts(_ptr(h2), _ptr(part2), hid // 512, 32768.0, 1e-5, _row(stats, 2), _row(stats, 3), _ptr(n2w),
                                           _ptr(n2b), None, _ptr(a2), frames, PFh, d2.p, d2.salt, 0.0, 0, 1, _ptr(seed), _ptr(a2_slot),
                                           st), "npvp_frameln_act_fwd_parts")
        tag_amax(a2, a2_slot)
        # fc2 (+ statistics), norm3 + GELU + dropout + residual + drop-path
        h3 = torch.empty(R, Co, dtype=f32, device=dev)
        part3 = torch.empty(frames * (Co // 64) * 2, dtype=f32, device=dev)
        gemm(1, 1, R, Co, hid, a2, hid, w2, w2.stride(0), h3, bias=b2, b_pre=_planes(w2, "F", R),
             rowstats=part3)
        DropRecorder.note(d3, "elem", R * Co)
        DropRecorder.note(dp, "group", frames // max(1, T))
        out = torch.empty(R, Co, dtype=f32, device=dev)
        check(L.npvp_frameln_act_fwd_parts(_ptr(h3), _ptr(part3), Co // 64, 4096.0, 1e-5, _row(stats, 4), _row(stats, 5), _ptr(n3w),
                                           _ptr(n3b), _ptr(res), _ptr(out), frames, PFo, d3.p, d3.salt, dp.p, dp.salt, T, _ptr(seed),
                                           None, st), "npvp_frameln_act_fwd_parts")
        ctx.save_for_backward(x, h1, h2, a2, h3, stats, wtb, w1, w2, n1w, n1b, n2w, n2b, n3w, n3b)
        ctx.cfg = (frames, T, d2, d3, dp, res is not None, b1 is not None, b2 is not None)
        ctx.sinks = (_wb_sink(w1, b1), _wb_sink(w2, b2), _ln_sink(n1w, n1b), _ln_sink(n2w, n2b), _ln_sink(n3w, n3b))
        sd = _wb_sink(dww, dwb) if (dww.is_contiguous() and dwb is not None) else None
        ctx.sink_dw = sd if (sd and sd[1] is not None) else None
        return out

    @staticmethod
    def _fln_bwd(L, dout, h, mean, rstd, w, b, frames, PF, d, dp, T, sk, psum=None, nparts=0, want_amax=True):
        """frame-LN backward (with its own statistics pass, or with the producer's `psum`); parameter gradients into the
        sink (partials reduced on the gradient stream) or returned; want_amax: dh feeds a GEMM (gets an amax slot)"""
        dev = h.device
        dh = torch.empty_like(h)
        slot = _new_slot(dev, want_amax)
        tag_amax(dh, slot)
        dw, db = (sk[0][0], sk[1][0]) if sk else (torch.empty_like(w), torch.empty_like(b))
        ws, wsn = _ws(L.npvp_frameln_act_bwd_workspace_bytes(frames, PF), dev)
        mode = _sink_mode(sk)
        if psum is None:
            seed = _S._cur.rng.seed_tensor(dev) if (d.on or dp.on) else None
            check(L.npvp_frameln_act_bwd(_ptr(dout), _ptr(h), _ptr(mean), _ptr(rstd), _ptr(w), _ptr(b), _ptr(dh), _ptr(dw), _ptr(db),
                                         frames, PF, d.p, d.salt, dp.p, dp.salt, T, _ptr(seed), mode, _ptr(slot), _ptr(ws), wsn,
                                         _stream()), "npvp_frameln_act_bwd")
        else:
            pe = HbmProbe.begin()
            check(L.npvp_frameln_act_bwd_apply(_ptr(dout), _ptr(h), _ptr(mean), _ptr(rstd), _ptr(w), _ptr(b), _ptr(psum), nparts,
                                               _ptr(dh), _ptr(dw), _ptr(db), frames, PF, mode, _ptr(slot), _ptr(ws), wsn, _stream()),
                  "npvp_frameln_act_bwd_apply")
            HbmProbe.end(pe, "npvp::frameln_act_bwd_fused_kernel", 4.0 * 3 * frames * PF)            # reads dout, h; writes dh
        if sk:
            _sunk_fln_reduce(sk, ws, dw, db, frames, PF)
            return dh, None, None
        return dh, dw, db

    @staticmethod
    def _fln_pgrad(L, dout, h, mean, rstd, w, b, frames, PF, d, sk):
        """frame-LN backward without the input gradient (npvp_frameln_act_bwd_pgrad): -> psum [frames][PF/1024][2], dw, db
        (None, None when they went into the sink); mean / rstd are device addresses"""
        dev = h.device
        psum = torch.empty(frames * (PF // 1024) * 2, dtype=torch.float32, device=dev)
        dw, db = (sk[0][0], sk[1][0]) if sk else (torch.empty_like(w), torch.empty_like(b))
        ws, wsn = _ws(L.npvp_frameln_act_bwd_workspace_bytes(frames, PF), dev)
        seed = _S._cur.rng.seed_tensor(dev) if d.on else None
        check(L.npvp_frameln_act_bwd_pgrad(_ptr(dout), _ptr(h), mean, rstd, _ptr(w), _ptr(b), _ptr(psum), _ptr(dw), _ptr(db),
                                           frames, PF, d.p, d.salt, 0.0, 0, 1, _ptr(seed), _sink_mode(sk), _ptr(ws), wsn, _stream()),
              "npvp_frameln_act_bwd_pgrad")
        if sk:
            _sunk_fln_reduce(sk, ws, dw, db, frames, PF)
            return psum, None, None
        return psum, dw, db

    @staticmethod
    def _lin_bwd(dy, x, w, sk, has_b):
        """dgrad on this stream, weight (+bias) gradient into the sink on the gradient stream or returned"""
        return linear_bwd(dy, x, w, True if has_b else None, sk)      # (`b` only says whether there is a bias gradient to take)

    @scoped
    def backward(ctx, dout):
        x, h1, h2, a2, h3, stats, wtb, w1, w2, n1w, n1b, n2w, n2b, n3w, n3b = ctx.saved_tensors
        frames, T, d2, d3, dp, has_res, has_b1, has_b2 = ctx.cfg
        s_fc1, s_fc2, s_n1, s_n2, s_n3 = ctx.sinks
        L = lib()
        R, hid = h1.shape
        Co = h3.shape[1]
        dev = x.device
        dout = _c(dout)
        F_ = _MlpDwbn
        dh3, gn3w, gn3b = F_._fln_bwd(L, dout, h3, stats[4], stats[5], n3w, n3b, frames, 64 * Co, d3, dp, T, s_n3)
        da2, gw2, gb2 = F_._lin_bwd(dh3, a2, w2, s_fc2, has_b2)
        fuse_n2 = MID_BWD_N2 and hid // 16 <= 256
        if fuse_n2:
            # norm2's backward WITHOUT its input gradient: frame sums (partials) + parameter gradients in one pass over da2 / h2;
            # dh2 is evaluated inside the fused middle's backward below and never written (2 passes over [R, hidden] less)
            psum2, gn2w, gn2b = F_._fln_pgrad(L, da2, h2, _row(stats, 2), _row(stats, 3), n2w, n2b, frames, 64 * hid, d2, s_n2)
        else:
            dh2, gn2w, gn2b = F_._fln_bwd(L, da2, h2, stats[2], stats[3], n2w, n2b, frames, 64 * hid, d2, NO_DROP, 1, s_n2,
                                          want_amax=False)
            del da2
        # fused middle backward: da1, depthwise weight / bias gradient (a1 recomputed from h1), norm1's backward statistics
        da1 = torch.empty_like(h1)
        dwtb = torch.empty(10, hid, dtype=torch.float32, device=dev)
        nparts = hid // 256                     # partials per frame of norm1's backward statistics (one per block of the kernel below)
        psum = torch.empty(frames * nparts * 2, dtype=torch.float32, device=dev)
        ws, wsn = _ws(L.npvp_mlpdw_mid_bwd_workspace_bytes(frames, hid), dev)
        sk_dw = ctx.sink_dw if ctx.needs_input_grad[6] and ctx.needs_input_grad[7] else None
        wmode = 2 if sk_dw else 0          # 2: the depthwise gradient partials stay in `ws` for the gradient stream (below)
        if fuse_n2:
            seed = _S._cur.rng.seed_tensor(dev) if d2.on else None
            pe = HbmProbe.begin()
            check(L.npvp_mlpdw_mid_bwd_n2(_ptr(da2), _ptr(h2), _row(stats, 2), _row(stats, 3), _ptr(n2w), _ptr(n2b), _ptr(psum2),
                                          hid // 16, d2.p, d2.salt, _ptr(seed), _ptr(h1), _row(stats, 0), _row(stats, 1), _ptr(n1w),
                                          _ptr(n1b), _ptr(wtb), _ptr(da1), _ptr(dwtb), _ptr(psum), frames, 8, 8, hid, wmode, _ptr(ws), wsn,
                                          _stream()), "npvp_mlpdw_mid_bwd_n2")
            HbmProbe.end(pe, "npvp::mlpdw_mid_bwd_kernel<1, true>", 4.0 * 4 * frames * 64 * hid)     # reads da2, h2, h1; writes da1
            del da2, psum2
        else:
            check(L.npvp_mlpdw_mid_bwd(_ptr(dh2), _ptr(h1), _row(stats, 0), _row(stats, 1), _ptr(n1w), _ptr(n1b), _ptr(wtb), _ptr(da1),
                                       _ptr(dwtb), _ptr(psum), frames, 8, 8, hid, wmode, _ptr(ws), wsn, _stream()), "npvp_mlpdw_mid_bwd")
            del dh2
        if sk_dw:
            # straight into the gradient slots (on the gradient stream, like every in-place gradient write): the chunk partials
            # are reduced, transposed and accumulated by ONE launch there - no reduction on this stream, no transpose, none of
            # autograd's accumulate adds
            fn = lambda ws=ws, gw=sk_dw[0][0], gb=sk_dw[1][0], F=frames, C=hid: check(
                L.npvp_mlpdw_mid_bwd_reduce_into(_ptr(ws), _ptr(gw), _ptr(gb), F, C, _stream()), "npvp_mlpdw_mid_bwd_reduce_into")
            if _S._cur.reduce.enabled:
                gw, gb = sk_dw[0][0], sk_dw[1][0]
                _S._cur.reduce.add(L.npvp_mlpdw_mid_bwd_reduce_job, "npvp_mlpdw_mid_bwd_reduce_job", (_ptr(ws), _ptr(gw), _ptr(gb), frames, hid),
                                (gw.data_ptr(), gb.data_ptr()), ws, sk_dw)
            elif _S._cur.wgrad.enabled:
                _S._cur.wgrad.run(fn, ws, wrote=sk_dw)
            else:
                fn()
                _S._cur.grad_sink.wrote(*sk_dw)
            gdww = gdwb = None
        else:
            gdww = torch.empty(hid, 1, 3, 3, dtype=torch.float32, device=dev)
            check(L.npvp_transpose(_ptr(dwtb), _ptr(gdww), 1, 9, hid, _stream()), "npvp_transpose")
            gdwb = dwtb[9]
        dh1, gn1w, gn1b = F_._fln_bwd(L, da1, h1, stats[0], stats[1], n1w, n1b, frames, 64 * hid, NO_DROP, NO_DROP, 1, s_n1,
                                      psum=psum, nparts=nparts)
        del da1
        dx, gw1, gb1 = F_._lin_bwd(dh1, x, w1, s_fc1, has_b1)
        return (dx, dout if has_res else None, gw1, gb1, gn1w, gn1b, gdww, gdwb, gn2w, gn2b, gw2, gb2, gn3w, gn3b,
                None, None, None, None)


MID_BWD_N2 = True        # (False: norm2's backward as two kernels of its own - the op test compares the two routes)


def mlpdwbn_fused_supported(R, C, hid, Co, H, W):
    return (GEMM_PRECISION in (4, 6) and H == 8 and W == 8 and R % 64 == 0 and hid % 512 == 0 and Co % 128 == 0 and hid % 128 == 0
            and C % 32 == 0)


def mlpdwbn(x, res, w1, b1, n1w, n1b, dww, dwb, n2w, n2b, w2, b2, n3w, n3b, frames, T, p_drop, p_dp):
    return _MlpDwbn.apply(x, res, w1, b1, n1w, n1b, dww, dwb, n2w, n2b, w2, b2, n3w, n3b, frames, T, p_drop, p_dp)


# --------------------------------------------------------------------------- sub-layer nodes
# One autograd node per residual sub-layer of a VidHRFormer block (ref/models/VidHRFormer.py:87-112,210-243): the pre-norm
# LayerNorm, the positional fuse, the projections, the attention core / FFN and the residual epilogue are a fixed sequence
# of kernel launches - nothing in between needs autograd's bookkeeping.  Against one node per kernel this removes ~3/4 of the
# Python / autograd work per step (the 8-clip shards of c3 / c4 were bound by it) and the [R, C] gradient adds autograd
# inserted where LN(x) feeds two consumers (they become the `residual` input of a dgrad GEMM's epilogue).
def _raw_ln_fwd(x2, w, b, eps):
    rows, C = x2.shape
    y = torch.empty_like(x2)
    st = torch.empty(2, rows, dtype=torch.float32, device=x2.device)
    slot = _new_slot(x2.device)
    check(lib().npvp_layernorm_fwd(_ptr(x2), _ptr(w), _ptr(b), _ptr(y), _row(st, 0), _row(st, 1), rows, C, eps, 0, _ptr(slot),
                                   _stream()), "npvp_layernorm_fwd")
    return tag_amax(y, slot), st


def _raw_ln_bwd(dy2, x2, w, b, st, dres, sk):
    """-> dx, dw, db (None, None when the parameter gradients went into the sink)"""
    L = lib()
    rows, C = x2.shape
    dx = torch.empty_like(x2)
    dw, db = (sk[0][0], sk[1][0]) if sk else (torch.empty_like(w), torch.empty_like(b))
    ws, wsn = _ws(L.npvp_layernorm_bwd_workspace_bytes(rows, C), x2.device)
    slot = _new_slot(x2.device)
    pe = HbmProbe.begin()
    check(L.npvp_layernorm_bwd(_ptr(dy2), _ptr(x2), _ptr(w), _ptr(b), _row(st, 0), _row(st, 1), _ptr(dx), _ptr(dw), _ptr(db), rows, C,
                               0, _ptr(dres), _sink_mode(sk), _ptr(slot), _ptr(ws), wsn, _stream()), "npvp_layernorm_bwd")
    HbmProbe.end(pe, "npvp::ln_bwd_kernel<2>", 4.0 * (3 + (dres is not None)) * rows * C)            # reads dy, x (+ dres); writes dx
    tag_amax(dx, slot)
    if sk:
        _sunk_ln_reduce(sk, ws, rows, C)
        return dx, None, None
    return dx, dw, db


def _raw_posfuse_fwd(x, add, beta, gamma, N, T):
    PF = x.numel() // (N * T)
    y = torch.empty_like(x)
    st = torch.empty(2, N * T, dtype=torch.float32, device=x.device)
    slot = _new_slot(x.device)
    check(lib().npvp_posfuse_fwd(_ptr(x), _ptr(add), _ptr(beta), _ptr(gamma), _ptr(y), _row(st, 0), _row(st, 1), N, T, PF, 1e-5,
                                 _ptr(slot), _stream()), "npvp_posfuse_fwd")
    return tag_amax(y, slot), st


LN_POSFUSE_ONE_KERNEL = True          # (False: the two-kernel route, kept for the op tests)


def _raw_ln_posfuse_fwd(x2, lw, lb, eps, add, beta, gamma, N, T):
    """LayerNorm then positional fuse of its output: -> x1 = LN(x2), its statistics [2, rows], fused, its statistics [2, N*T].
    One kernel for frames of 64 token rows x 512 channels (npvp_ln_posfuse_fwd), the two-kernel route otherwise."""
    rows, C = x2.shape
    if not (LN_POSFUSE_ONE_KERNEL and C == 512 and rows == N * T * 64):
        x1, lst = _raw_ln_fwd(x2, lw, lb, eps)
        fused, pst = _raw_posfuse_fwd(x1, add, beta, gamma, N, T)
        return x1, lst, fused, pst
    x1, fused = torch.empty_like(x2), torch.empty_like(x2)
    lst = torch.empty(2, rows, dtype=torch.float32, device=x2.device)
    pst = torch.empty(2, N * T, dtype=torch.float32, device=x2.device)
    s1, s2 = _new_slot(x2.device), _new_slot(x2.device)
    check(lib().npvp_ln_posfuse_fwd(_ptr(x2), _ptr(lw), _ptr(lb), eps, _ptr(x1), _row(lst, 0), _row(lst, 1), _ptr(add), _ptr(beta),
                                    _ptr(gamma), _ptr(fused), _row(pst, 0), _row(pst, 1), N, T, 64, C, 1e-5, _ptr(s1), _ptr(s2),
                                    _stream()), "npvp_ln_posfuse_fwd")
    return tag_amax(x1, s1), lst, tag_amax(fused, s2), pst


def _raw_posfuse_bwd(dy, x, add, beta_shape, gamma, st, N, T, want_add, sinks=(None, None, None)):
    """-> du [like x], dadd, dbeta, dgamma; sinks = the ActSinks of (beta, gamma, add) where the caller found them: those gradients
    are accumulated in place and come back as None"""
    PF = x.numel() // (N * T)
    du, dbeta, dgamma = _posfuse_bwd_call(dy, x, add, gamma, _row(st, 0), _row(st, 1), N, T, PF, beta_shape, True, gamma is not None,
                                          sinks[0], sinks[1])
    dadd = None
    if add is not None and want_add:
        if sinks[2] is not None:
            buf, acc = sinks[2].target(add.shape, x.device)
            reduce_mid(du.view(N, T, PF), out=buf.view(N, PF), accumulate=acc)
        else:
            dadd = reduce_mid(du.view(N, T, PF)).view(add.shape)
    return du, dadd, dbeta, dgamma


def _lin_grads(dy, x, w, b, sk, a_drop=NO_DROP):
    """weight (+ bias) gradient of y = x w^T + b: into the sink on the gradient stream (-> None, None) or returned"""
    has_b = b is not None
    if sk and (has_b == (sk[1] is not None)):
        _sunk_wgrad(dy, x, has_b, sk, a_drop=a_drop)
        return None, None
    g = linear_wgrad(dy, x, has_b, a_drop=a_drop)
    return (g[0], g[1]) if has_b else (g, None)


class _SelfAttnSublayer(torch.autograd.Function):
    """y = x + drop(out_proj(attn(q = k = fuse(LN(x) [+ add]), v = LN(x))))   - spatial-window or temporal self-attention
    (ref/models/VidHRFormer.py:87-88,94-107,210-212,217-221): LN, positional fuse (2 kernels), q|k GEMM, v GEMM, attention
    core, out-projection GEMM with the dropout / drop-path + residual epilogue."""

    @staticmethod
    def forward(ctx, x, lw, lb, eps, beta, gamma, add, wqk, bqk, wv, bv, wo, bo, cfg, drop, N, T):
        remember(ctx)
        # wqk / wv are the [:2C] / [2C:] row slices of in_proj_weight, sliced by the caller WITH grad mode on: a slice taken in
        # here (grad mode off) would not be recognised as a view of a flat-buffer parameter by GradSink
        _chk(x, lw, lb, beta, gamma, add, wqk, bqk, wv, bv, wo, bo)
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        beta_c = _c(beta)
        gamma_c = None if gamma is None else _c(gamma)
        add_c = None if add is None else _c(add)
        x1, lst, fused, pst = _raw_ln_posfuse_fwd(x2, lw, lb, eps, add_c, beta_c, gamma_c, N, T)
        qk = linear_fwd(fused, wqk, bqk)
        v = linear_fwd(x1, wv, bv)
        o = torch.empty_like(v)
        _attn_fwd(qk[:, :C], qk[:, C:], v, o, cfg)
        y = linear_fwd(o, wo, bo, residual=x2, drop=drop)
        ctx.save_for_backward(x2, x1, lst, fused, pst, qk, v, o, lw, lb, gamma_c, add_c, wqk, bqk, wv, bv, wo, bo)
        ctx.act_sinks = (ActSink.of(beta), ActSink.of(gamma), ActSink.of(add))
        ctx.cfg = (cfg, drop, N, T, x.shape, beta.shape)
        ctx.sinks = (_ln_sink(lw, lb), _wb_sink(wqk, bqk), _wb_sink(wv, bv), _wb_sink(wo, bo))
        return y.view(x.shape)

    @scoped
    def backward(ctx, dy):
        x2, x1, lst, fused, pst, qk, v, o, lw, lb, gamma, add, wqk, bqk, wv, bv, wo, bo = ctx.saved_tensors
        cfg, drop, N, T, xshape, beta_shape = ctx.cfg
        s_ln, s_qk, s_v, s_o = ctx.sinks
        C = x2.shape[1]
        dy2 = _c(dy).reshape(-1, C)
        dz, ad = masked_grad(dy2, drop, wo)
        do, gwo, gbo = linear_bwd(dz, o, wo, bo, s_o, a_drop=ad)
        dqk, dv = torch.empty_like(qk), torch.empty_like(v)
        _attn_bwd(qk[:, :C], qk[:, C:], v, do, dqk[:, :C], dqk[:, C:], dv, cfg, packed=dqk)
        dfused, gwqk, gbqk = linear_bwd(dqk, fused, wqk, bqk, s_qk)
        du, dadd, dbeta, dgamma = _raw_posfuse_bwd(dfused, x1, add, beta_shape, gamma, pst, N, T, ctx.needs_input_grad[6], ctx.act_sinks)
        # dx1 = dv Wv + du: the second consumer's gradient rides in as the dgrad GEMM's residual input (no separate add)
        dx1, gwv, gbv = linear_bwd(dv, x1, wv, bv, s_v, residual=du)
        dx, glw, glb = _raw_ln_bwd(dx1, x2, lw, lb, lst, dy2, s_ln)
        return (dx.view(xshape), glw, glb, None, dbeta, dgamma, dadd, gwqk, gbqk, gwv, gbv, gwo, gbo, None, None, None, None)


class _CrossAttnSublayer(torch.autograd.Function):
    """y = x + droppath_t(out_proj(attn(q = fuse(LN(x) + add), k = key, v = memory)))   - the decoder's encoder-decoder
    attention (ref/models/VidHRFormer.py:229-239); key = fuse(memory) is layer invariant and supplied by the caller."""

    @staticmethod
    def forward(ctx, x, lw, lb, eps, beta, gamma, add, key, memory, wq, bq, wk, bk, wv, bv, wo, bo, cfg, drop, N, T):
        remember(ctx)
        _chk(x, lw, lb, beta, gamma, add, key, memory, wq, bq, wk, bk, wv, bv, wo, bo)
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        k2, m2 = _c(key).reshape(-1, C), _c(memory).reshape(-1, C)
        beta_c = _c(beta)
        gamma_c = None if gamma is None else _c(gamma)
        add_c = None if add is None else _c(add)
        x1, lst, query, pst = _raw_ln_posfuse_fwd(x2, lw, lb, eps, add_c, beta_c, gamma_c, N, T)
        q = linear_fwd(query, wq, bq)
        k = linear_fwd(k2, wk, bk)
        v = linear_fwd(m2, wv, bv)
        o = torch.empty_like(q)
        _attn_fwd(q, k, v, o, cfg)
        y = linear_fwd(o, wo, bo, residual=x2, drop=drop)
        ctx.save_for_backward(x2, x1, lst, query, pst, q, k, v, o, k2, m2, lw, lb, gamma_c, add_c, wq, bq, wk, bk, wv, bv, wo, bo)
        ctx.act_sinks = (ActSink.of(beta), ActSink.of(gamma), ActSink.of(add))
        ctx.km_sinks = (ActSink.of(key), ActSink.of(memory))
        ctx.cfg = (cfg, drop, N, T, x.shape, beta.shape, key.shape, memory.shape)
        ctx.sinks = (_ln_sink(lw, lb), _wb_sink(wq, bq), _wb_sink(wk, bk), _wb_sink(wv, bv), _wb_sink(wo, bo))
        return y.view(x.shape)

    @scoped
    def backward(ctx, dy):
        x2, x1, lst, query, pst, q, k, v, o, k2, m2, lw, lb, gamma, add, wq, bq, wk, bk, wv, bv, wo, bo = ctx.saved_tensors
        cfg, drop, N, T, xshape, beta_shape, kshape, mshape = ctx.cfg
        s_ln, s_q, s_k, s_v, s_o = ctx.sinks
        C = x2.shape[1]
        dy2 = _c(dy).reshape(-1, C)
        dz, ad = masked_grad(dy2, drop, wo)
        do, gwo, gbo = linear_bwd(dz, o, wo, bo, s_o, a_drop=ad)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _attn_bwd(q, k, v, do, dq, dk, dv, cfg)
        dquery, *gq = linear_bwd(dq, query, wq, bq, s_q)
        du, dadd, dbeta, dgamma = _raw_posfuse_bwd(dquery, x1, add, beta_shape, gamma, pst, N, T, ctx.needs_input_grad[6], ctx.act_sinks)
        dkey = dmem = None
        for need, g_, w_, sink, shape, which in ((ctx.needs_input_grad[7], dk, wk, ctx.km_sinks[0], kshape, 0),
                                                 (ctx.needs_input_grad[8], dv, wv, ctx.km_sinks[1], mshape, 1)):
            if not need:
                continue
            if sink is not None:        # the decoder's 8 layers share key / memory: their gradients are summed by the dgrad GEMMs' epilogues
                buf, acc = sink.target(shape, g_.device)
                linear_dgrad(g_, w_, out=buf.view(g_.shape[0], C), accumulate=acc)
            elif which == 0:
                dkey = linear_dgrad(g_, w_).view(shape)
            else:
                dmem = linear_dgrad(g_, w_).view(shape)
        gk = _lin_grads(dk, k2, wk, bk, s_k)
        gv = _lin_grads(dv, m2, wv, bv, s_v)
        dx, glw, glb = _raw_ln_bwd(du, x2, lw, lb, lst, dy2, s_ln)
        return (dx.view(xshape), glw, glb, None, dbeta, dgamma, dadd, dkey, dmem, gq[0], gq[1], gk[0], gk[1], gv[0], gv[1], gwo, gbo,
                None, None, None, None)


class _FfnSublayer(torch.autograd.Function):
    """y = x + drop3(linear2(drop2(GELU(linear1(LN(x))))))   (ref/models/VidHRFormer.py:110-112,224-226)"""

    @staticmethod
    def forward(ctx, x, lw, lb, eps, w1, b1, w2, b2, p):
        remember(ctx)
        _chk(x, lw, lb, w1, b1, w2, b2)
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        xn, lst = _raw_ln_fwd(x2, lw, lb, eps)
        R, Fh = x2.shape[0], w1.shape[0]
        d2, d3 = Drop(p), Drop(p)
        h = torch.empty(R, Fh, dtype=torch.float32, device=x.device)
        a_slot = _new_slot(x.device)
        a = tag_amax(linear_fwd(xn, w1, b1, act=1, aux_out=h, drop=d2, y_amax=a_slot), a_slot)
        y = linear_fwd(a, w2, b2, residual=x2, drop=d3)
        ctx.save_for_backward(x2, xn, lst, h, a, lw, lb, w1, b1, w2, b2)
        ctx.cfg = (d2, d3, x.shape)
        ctx.sinks = (_ln_sink(lw, lb), _wb_sink(w1, b1), _wb_sink(w2, b2))
        return y.view(x.shape)

    @scoped
    def backward(ctx, dy):
        x2, xn, lst, h, a, lw, lb, w1, b1, w2, b2 = ctx.saved_tensors
        d2, d3, xshape = ctx.cfg
        s_ln, s1, s2 = ctx.sinks
        C = x2.shape[1]
        dy2 = _c(dy).reshape(-1, C)
        dz2, ad = masked_grad(dy2, d3, w2)
        dh_slot = _new_slot(dy2.device)
        dh, gw2, gb2 = linear_bwd(dz2, a, w2, b2, s2, act=3, aux_in=h, drop=d2, dx_amax=dh_slot, a_drop=ad)
        tag_amax(dh, dh_slot)
        dxn, gw1, gb1 = linear_bwd(dh, xn, w1, b1, s1)
        dx, glw, glb = _raw_ln_bwd(dxn, x2, lw, lb, lst, dy2, s_ln)
        return dx.view(xshape), glw, glb, None, gw1, gb1, gw2, gb2, None


def self_attn_sublayer(x, norm, beta, gamma, add, mha, cfg, drop, N, T):
    C = mha.embed_dim
    w, b = mha.in_proj_weight, mha.in_proj_bias
    return _SelfAttnSublayer.apply(x, norm.weight, norm.bias, norm.eps, beta, gamma, add, w[:2 * C], b[:2 * C], w[2 * C:], b[2 * C:],
                                   mha.out_proj.weight, mha.out_proj.bias, cfg, drop, N, T)


def cross_attn_sublayer(x, norm, beta, gamma, add, key, memory, mha, cfg, drop, N, T):
    C = mha.embed_dim
    w, b = mha.in_proj_weight, mha.in_proj_bias
    return _CrossAttnSublayer.apply(x, norm.weight, norm.bias, norm.eps, beta, gamma, add, key, memory, w[:C], b[:C], w[C:2 * C],
                                    b[C:2 * C], w[2 * C:], b[2 * C:], mha.out_proj.weight, mha.out_proj.bias, cfg, drop, N, T)


def ffn_sublayer(x, norm, lin1, lin2, p):
    return _FfnSublayer.apply(x, norm.weight, norm.bias, norm.eps, lin1.weight, lin1.bias, lin2.weight, lin2.bias, p)


class _Im2Col(torch.autograd.Function):
    """[F, H*W, C] -> [F*H*W, 9*C] patches of a 3x3 / pad-1 conv (tap-major columns); backward = col2im."""

    @staticmethod
    def forward(ctx, x, frames, H, W):
        remember(ctx)
        _chk(x)
        x = _c(x)
        C = x.shape[-1]
        out = torch.empty(frames * H * W, 9 * C, dtype=torch.float32, device=x.device)
        check(lib().npvp_im2col3x3(_ptr(x), _ptr(out), frames, H, W, C, 0, _stream()), "npvp_im2col3x3")
        ctx.cfg = (frames, H, W, C, x.shape)
        return out

    @scoped
    def backward(ctx, g):
        frames, H, W, C, shape = ctx.cfg
        g = _c(g)
        out = torch.empty(shape, dtype=torch.float32, device=g.device)
        check(lib().npvp_im2col3x3(_ptr(g), _ptr(out), frames, H, W, C, 1, _stream()), "npvp_im2col3x3(col2im)")
        return out, None, None, None


def im2col3x3(x, frames, H, W):
    return _Im2Col.apply(x, frames, H, W)


def nchw_to_canonical(x):
    """(N,T,C,H,W) -> canonical [N*T, H*W, C].  A channels-last feature tensor (what the frozen encoder emits when it
    runs in torch.channels_last) already IS the canonical layout: it is viewed, not transposed."""
    if x.dim() == 5 and x.permute(0, 1, 3, 4, 2).is_contiguous() and not x.requires_grad:
        N, T, C, H, W = x.shape
        return x.permute(0, 1, 3, 4, 2).reshape(N * T, H * W, C)
    return _nchw_to_canonical(x)


def _nchw_to_canonical(x):
    """(N,T,C,H,W) -> canonical [N*T, H*W, C]"""
    N, T, C, H, W = x.shape
    return _Transpose.apply(x.reshape(N * T, C, H * W))


def canonical_to_nchw(x, N, T, H, W):
    """[N*T, H*W, C] -> (N,T,C,H,W)"""
    C = x.shape[-1]
    return _Transpose.apply(x.reshape(N * T, H * W, C)).reshape(N, T, C, H, W)


def mean_mid(x):
    return _MeanMid.apply(x)


# --------------------------------------------------------------------------- frozen autoencoder epilogues (csrc/ae.hip)
class _BiasAct(torch.autograd.Function):
    """out = act(x + bias[c]) (+ residual) on an (N,C,H,W) tensor in contiguous or channels_last memory; bias is frozen.
    Gradient w.r.t. x only (the decoder's input-gradient path), through the saved OUTPUT."""

    @staticmethod
    def forward(ctx, x, bias, residual, act):
        remember(ctx)
        _chk(x, bias, residual)
        need_x, need_res = ctx.needs_input_grad[0], ctx.needs_input_grad[2]     # (read before x may be rebound to a no-grad copy)
        N, C, H, W = x.shape
        cl = (not x.is_contiguous()) and x.is_contiguous(memory_format=torch.channels_last)
        if not cl and not x.is_contiguous():
            x = x.contiguous()
        if residual is not None and residual.stride() != x.stride():
            residual = residual.contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
        out = torch.empty_like(x)
        outer, inner, layout = (N * H * W, C, 0) if cl else (N * C, H * W, 1)
        check(lib().npvp_bias_act(_ptr(x), _ptr(bias), _ptr(residual), _ptr(out), outer, inner, C, layout, act, _stream()),
              "npvp_bias_act")
        ctx.act, ctx.cl, ctx.has_res = act, cl, residual is not None
        if need_x or need_res:
            if residual is not None:
                raise NotImplementedError("bias_act: backward with a fused skip-add is not on the Stage-2 path")
            ctx.save_for_backward(out)
        return out

    @scoped
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous(memory_format=torch.channels_last if ctx.cl else torch.contiguous_format)
        dx = torch.empty_like(y)
        check(lib().npvp_act_bwd(_ptr(g), _ptr(y), _ptr(dx), y.numel(), ctx.act, _stream()), "npvp_act_bwd")
        return dx, None, None, None


def bias_act(x, bias, act=0, residual=None):
    return _BiasAct.apply(x, bias, residual, act)
