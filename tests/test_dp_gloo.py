"""CPU, world_size 2 and 4, gloo: the data-parallel layer (npvp_amd.dp) is correct by construction -
W ranks on shards of a global batch produce the gradients / BatchNorm statistics / parameters of one
rank on the whole batch, with gradients reduced in place in the flat bucket buffer."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Net(nn.Module):
    """Shaped like the predictor's coupling structure: a shared (tied) norm used twice, a BatchNorm head
    (the EventEncoder), a `transformer` sub-module that forms the clip range, and an unused parameter."""

    def __init__(self):
        super().__init__()
        norm = nn.LayerNorm(16)
        self.enc = nn.Linear(16, 16)
        self.enc_norm = norm
        self.head = nn.Sequential(nn.Conv2d(4, 6, 3, 1, 1, bias=False), nn.BatchNorm2d(6), nn.ReLU(True))
        self.transformer = nn.Sequential(nn.Linear(16, 32), nn.GELU(), nn.Linear(32, 16))
        self.transformer.norm = norm
        self.unused = nn.Parameter(torch.ones(5))

    def forward(self, x, img):
        h = self.enc_norm(self.enc(x))
        z = self.head(img).mean(dim=(1, 2, 3), keepdim=False).unsqueeze(1)
        return self.transformer.norm(self.transformer(h) + z)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class _ImmediateTape:
    """stands in for trainer.StepTape on CPU (no stream capture there): `cut(action)` runs the action at once and logs it - the step
    then takes dp's RECORDING code paths (GradSync._launch -> cut -> _launch_now, finish -> one closing action, SyncBatchNorm's
    statistics as actions) and must give the results of the direct path"""

    def __init__(self):
        self.log = []

    def cut(self, action):
        self.log.append(action)
        action()


def _worker(rank, world, port, out, taped=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from npvp_amd import dp
    from npvp_amd.trainer import FlatBuffers
    dp.init_distributed("gloo")
    torch.manual_seed(0)
    m = Net()
    if rank != 0:                       # de-synchronise, then C3 broadcast must repair it
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
    dp.broadcast_module(m)
    dp.convert_sync_batchnorm(m)
    buf = FlatBuffers(m, m.transformer)
    sync = dp.GradSync(buf, bucket_bytes=1024)       # tiny buckets: several per step
    assert len(sync.buckets) >= 3
    g = torch.Generator().manual_seed(1)
    X, IMG = torch.randn(8, 16, generator=g), torch.randn(8, 4, 5, 5, generator=g)
    x, img = dp.shard_batch(X, rank, world), dp.shard_batch(IMG, rank, world)
    m.train()
    for it in range(2):
        tape = _ImmediateTape() if (taped and it == 1) else None        # (step 0 teaches GradSync its counts, as the recording's warm-up does)
        prev = dp.set_tape(tape)
        buf.zero_grad()
        y = m(x, img)
        # loss = mean over the GLOBAL batch -> per-rank mean, gradients averaged by GradSync
        (y ** 2).mean().backward()
        sync.finish()
        dp.set_tape(prev)
        if tape is not None:
            # one SyncBatchNorm forward + one backward collective, every contributing bucket from a hook, ONE closing action
            assert len(tape.log) >= 2 + 3 + 1, len(tape.log)
            assert all(b["work"] is None and b["ready"] == 0 for b in sync.buckets)
    if rank == 0:
        torch.save({"flat_g": buf.flat_g.clone(), "rm": m.head[1].running_mean.clone(), "rv": m.head[1].running_var.clone(),
                    "tail": (buf.tail_begin, buf.tail_end), "launched": sync.launched}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,taped", [(2, False), (4, False), (2, True)])
def test_two_rank_gloo_matches_single_process(tmp_path, world, taped):
    """(2 ranks, and 4: the shards of the 8-sample batch are then 2 samples each - SyncBatchNorm's statistics and the bucket means
    must still be those of the whole batch; taped: the second step through the recording code paths of a segmented replay)"""
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, out, taped), nprocs=world, join=True)
    got = torch.load(out)
    # single process, whole batch
    sys.path.insert(0, ROOT)
    from npvp_amd.trainer import FlatBuffers
    torch.manual_seed(0)
    m = Net()
    buf = FlatBuffers(m, m.transformer)
    g = torch.Generator().manual_seed(1)
    X, IMG = torch.randn(8, 16, generator=g), torch.randn(8, 4, 5, 5, generator=g)
    m.train()
    for it in range(2):
        buf.zero_grad()
        (m(X, IMG) ** 2).mean().backward()
    assert torch.allclose(got["flat_g"], buf.flat_g, rtol=1e-4, atol=1e-6), (got["flat_g"] - buf.flat_g).abs().max()
    assert torch.allclose(got["rm"], m.head[1].running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(got["rv"], m.head[1].running_var, rtol=1e-4, atol=1e-6)
    assert got["tail"] == (buf.tail_begin, buf.tail_end) and buf.tail_end - buf.tail_begin > 0
    assert got["launched"] >= 6            # every bucket, both steps
    # the tied norm sits in the clip (transformer) range, the unused parameter's gradient stays zero
    assert float(m.unused.grad.abs().max()) == 0.0


def test_buckets_are_cut_in_autograd_order_with_a_small_last_one():
    """The flat buffer is laid out in module order and filled from the END in backward: buckets are cut walking from the end, and
    the bucket at the FRONT - reduced last, the only all-reduce with nothing left to hide behind - is kept small."""
    sys.path.insert(0, ROOT)
    from npvp_amd import dp
    from npvp_amd.trainer import FlatBuffers

    class Stack(nn.Module):
        def __init__(self):
            super().__init__()
            self.first = nn.Linear(8, 8)                                            # completes last in backward
            self.enc = nn.Sequential(*[nn.Linear(64, 64) for _ in range(6)])
            self.transformer = nn.Sequential(*[nn.Linear(64, 64) for _ in range(10)])
    m = Stack()
    buf = FlatBuffers(m, m.transformer)
    cap, cap_last = 4 * 3 * 4160, 4 * 4160                                          # three layers per bucket; one layer in the last one
    gs = dp.GradSync(buf, bucket_bytes=cap, last_bucket_bytes=cap_last)
    b = gs.buckets
    assert b[0]["lo"] == 0 and b[-1]["hi"] == buf.total and all(x["hi"] == y["lo"] for x, y in zip(b[:-1], b[1:]))
    sizes = [x["hi"] - x["lo"] for x in b]
    assert all(sz >= cap // 4 for sz in sizes[2:]), sizes                           # full buckets from the end
    assert sizes[0] <= cap_last // 4 + 4160, sizes                                  # the front (last-reduced) bucket is small
    assert sum(x["n"] for x in b) == len(buf.params)
    for p, (off, n) in zip(buf.params, buf.offsets):
        bk = b[gs.param_bucket[id(p)]]
        assert bk["lo"] <= off and off + n <= bk["hi"]
    assert gs.param_bucket[id(m.first.weight)] == 0 and gs.param_bucket[id(m.transformer[9].weight)] == len(b) - 1
