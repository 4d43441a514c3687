"""CPU, world_size 2 over gloo: the sample-sharding arithmetic and the all-gather of finished images
(the N>1 path of bench.py / rangeldm_amd.distributed)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rangeldm_amd import distributed as D
from rangeldm_amd.synth import latent_noise


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sampler(x_T):
    """Stands in for the GPU pipeline: a deterministic per-sample function of x_T (no cross-sample mixing)."""
    return torch.tanh(x_T * 0.5) + x_T.roll(1, dims=2) * 0.1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    B, shape = 3, (4, 8, 2)
    outs = []
    for it in range(2):
        idx = D.global_sample_indices(it, B, rank, world)
        x = torch.from_numpy(np.stack([latent_noise(7, j, shape) for j in idx]))
        outs.append(D.all_gather_images(_fake_sampler(x)))
    full = torch.cat(outs)
    # strong scaling (bench.py --scaling strong, BASELINE config 3): ONE global batch split into contiguous per-rank slices
    GB = 4
    lo, hi = D.shard_range(GB, rank, world)
    xs = torch.from_numpy(np.stack([latent_noise(7, 100 + j, shape) for j in range(lo, hi)]))
    strong = D.all_gather_images(_fake_sampler(xs))
    t = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
    info = D.comm_info(torch.device("cpu"))              # the `comm` block of the bench line: every rank answered a collective
    D.barrier()
    if rank == 0:
        q.put((full.numpy(), t, strong.numpy(), info))
    dist.destroy_process_group()


def test_two_rank_sharding_and_allgather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, tmax, strong, info = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0
    assert info["world"] == 2 and info["ranks_seen"] == [0, 1] and info["backend"] == "gloo" and info["collective"] == "torch"
    # every rank holds all images, ordered by GLOBAL sample index (ldm/inference.py:174-183 file-index arithmetic):
    # iteration i, rank r, slot j -> index (r + world*i)*B + j
    B, shape = 3, (4, 8, 2)
    ref = torch.from_numpy(np.stack([latent_noise(7, j, shape) for j in range(2 * world * B)]))
    assert np.array_equal(full, _fake_sampler(ref).numpy())
    # strong split: the gathered batch is the global batch in global order, whatever the rank count
    ref = torch.from_numpy(np.stack([latent_noise(7, 100 + j, shape) for j in range(4)]))
    assert np.array_equal(strong, _fake_sampler(ref).numpy())


def test_index_arithmetic_single_process():
    assert D.global_sample_indices(0, 16, 0, 1) == list(range(16))
    assert D.global_sample_indices(1, 16, 3, 8) == list(range((3 + 8) * 16, (3 + 8) * 16 + 16))
    assert D.shard_range(16, 7, 8) == (14, 16) and D.shard_range(5, 3, 4) == (5, 5)
    x = torch.arange(6.).view(2, 3)
    assert D.all_gather_images(x) is x                   # world 1: no collective


# ---------------------------------------------------------------------------------------------------------------------------------
# Bootstrap of the C-ABI RCCL communicator (rangeldm_amd.distributed.Communicator) with a fake library: the unique-id exchange through
# a TCPStore (no process group), rank ordering, the 128-byte id, and the agreement rounds -- a failure on ONE rank must make EVERY rank
# raise, before anything collective (the first real multi-GPU run is the driver's: this is its rehearsal without hardware).
# ---------------------------------------------------------------------------------------------------------------------------------
class _FakeRccl:
    def __init__(self, rank, fail_bind=False, fail_create=False, short_uid=False):
        self.rank, self.fail_bind, self.fail_create, self.short_uid = rank, fail_bind, fail_create, short_uid
        self.created, self.destroyed, self.ids_made, self.binds = [], 0, 0, 0

    def rldm_comm_bind(self):
        self.binds += 1
        return 1 if self.fail_bind else 0

    def rldm_comm_unique_id(self, buf, n):
        if self.fail_bind:
            return 1
        import ctypes
        self.ids_made += 1
        ctypes.memmove(buf, bytes([0x40 + self.rank]) * n, n)
        return 0

    def rldm_comm_create(self, uid, rank, world, out):
        self.created.append((bytes(uid.raw), rank, world))
        if self.fail_create:
            return 1
        out._obj.value = 0x1000 + rank                              # (byref(c_void_p): a non-null handle)
        return 0

    def rldm_comm_destroy(self, h):
        self.destroyed += 1


class _ShortUidExchange(D._StoreExchange):
    """rank `bad` receives a truncated id (a broken side channel)"""
    bad = 1

    def broadcast_uid(self, raw):
        got = super().broadcast_uid(raw)
        return got[:100] if self.rank == self.bad else got


def _bootstrap_ranks(world, port, fail_bind=(), fail_create=(), exchange_cls=None, rounds=1):
    import threading
    out = [None] * world

    def run(rank):
        lib = _FakeRccl(rank, rank in fail_bind, rank in fail_create)
        try:
            store = dist.TCPStore("127.0.0.1", port, world, rank == 0)
            for _ in range(rounds):                                 # (rounds > 1: several communicators over ONE store)
                ex = (exchange_cls or D._StoreExchange)(rank, world, store, timeout=60.0)
                c = D.Communicator(rank=rank, world=world, lib=lib, exchange=ex)
            out[rank] = ("ok", lib, c)
        except D.CommunicatorUnavailable as e:
            out[rank] = ("unavailable", lib, str(e))                # (nothing kept alive: rank 0's finish() waits for the peers)
        except Exception as e:                                      # pragma: no cover
            out[rank] = ("error", lib, repr(e))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
        assert not t.is_alive(), "a rank hung in the bootstrap"
    return out


def test_cabi_bootstrap_exchanges_rank0_id_in_rank_order():
    world = 3
    res = _bootstrap_ranks(world, _free_port())
    for rank, (status, lib, c) in enumerate(res):
        assert status == "ok", (rank, c)
        assert lib.created == [(bytes([0x40]) * 128, rank, world)]   # rank 0's 128-byte id, own rank, world -- on every rank
        assert (c.rank, c.world) == (rank, world)
        assert lib.ids_made == (1 if rank == 0 else 0) and lib.binds == (0 if rank == 0 else 1)   # only rank 0 makes an id
        c._h = None                                                  # (nothing to destroy in the fake)


def test_cabi_bootstrap_bind_failure_on_one_rank_stops_every_rank():
    res = _bootstrap_ranks(2, _free_port(), fail_bind=(1,))
    assert [r[0] for r in res] == ["unavailable", "unavailable"]
    assert all(r[1].created == [] for r in res)                      # nobody entered the collective init
    assert "another rank" in res[0][2] and "this rank" in res[1][2]


def test_cabi_bootstrap_create_failure_is_agreed_and_cleaned_up():
    res = _bootstrap_ranks(2, _free_port(), fail_create=(0,))
    assert [r[0] for r in res] == ["unavailable", "unavailable"]
    assert len(res[0][1].created) == 1 and len(res[1][1].created) == 1
    assert res[1][1].destroyed == 1                                  # the rank that did get a communicator gave it back


def test_cabi_bootstrap_malformed_id_on_one_rank_is_agreed_not_raised_alone():
    """a rank that receives a bad id must not raise between the agreement rounds (its peers would sit in ncclCommInitRank):
    it skips the init, reports the failure, and every rank gives its communicator back"""
    res = _bootstrap_ranks(2, _free_port(), exchange_cls=_ShortUidExchange)
    assert [r[0] for r in res] == ["unavailable", "unavailable"]
    assert len(res[0][1].created) == 1 and res[0][1].destroyed == 1   # rank 0 did initialise -- and gave it back
    assert res[1][1].created == []                                    # the rank with the bad id never entered the init
    assert "expected 128" in res[1][2] and "another rank" in res[0][2]


def test_cabi_bootstrap_twice_on_one_store_uses_fresh_keys():
    """generation-tagged keys: a second Communicator over the same store waits for its peers instead of reading the first one's answers"""
    world = 2
    res = _bootstrap_ranks(world, _free_port(), rounds=2)
    for rank, r in enumerate(res):
        assert r[0] == "ok", r
        assert [c[1:] for c in r[1].created] == [(rank, world)] * 2
        assert r[2]._exchange.gen == 2
        r[2]._h = None


def test_cabi_bootstrap_missing_library_is_an_outcome_not_an_exception(monkeypatch):
    """the .so cannot be loaded on rank 1: both ranks raise CommunicatorUnavailable together, nobody initialises"""
    import threading
    from rangeldm_amd import _lib
    world, port, out = 2, _free_port(), [None, None]

    def boom():
        raise RuntimeError("librangeldm_hip.so not found")
    monkeypatch.setattr(_lib, "lib", boom)

    def run(rank):
        lib = _FakeRccl(rank) if rank == 0 else None                 # rank 1 goes through _lib.lib()
        try:
            ex = D._StoreExchange(rank, world, dist.TCPStore("127.0.0.1", port, world, rank == 0), timeout=60.0)
            D.Communicator(rank=rank, world=world, lib=lib, exchange=ex)
            out[rank] = ("ok",)
        except D.CommunicatorUnavailable as e:
            out[rank] = ("unavailable", str(e), lib)
        except Exception as e:                                       # pragma: no cover
            out[rank] = ("error", repr(e))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
        assert not t.is_alive()
    assert [o[0] for o in out] == ["unavailable", "unavailable"], out
    assert out[0][2].created == [] and "not found" in out[1][1]


def test_cabi_bootstrap_default_store_is_reused_by_a_second_communicator(monkeypatch):
    """The DEFAULT path (no store passed: a TCPStore at MASTER_ADDR : MASTER_PORT + 1, hosted by rank 0 and kept while peers may still
    read it): a second Communicator built later in the same process must not try to bind the port again (EADDRINUSE on rank 0, its peers
    then waiting a full timeout for generation-2 keys nobody writes) -- it reuses the process's store, with fresh generation-tagged keys."""
    import threading
    port = _free_port()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(port - 1))
    world = 2
    for rnd in (1, 2):
        out = [None] * world

        def run(rank):
            lib = _FakeRccl(rank)
            try:
                c = D.Communicator(rank=rank, world=world, lib=lib)
                out[rank] = ("ok", lib, c)
            except Exception as e:                                   # pragma: no cover
                out[rank] = ("error", lib, repr(e))
        ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
            assert not t.is_alive(), "a rank hung in the bootstrap"
        for rank, r in enumerate(out):
            assert r[0] == "ok", r
            assert r[2]._exchange.gen == rnd and r[1].created == [(bytes([0x40]) * 128, rank, world)]
            r[2]._h = None
    assert len([k for k in D._LIVE_STORES if k[1] == port]) == world      # one store per (endpoint, rank): created once, used twice
