"""Sample-sharded multi-GPU inference.  One process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).  The reference shards by file-index arithmetic with no collective
(ldm/inference.py:159,174-183: image index = (rank + nproc*i)*B + j); here the same index arithmetic picks each
rank's global sample indices, x_T is a function of the GLOBAL index (so 1/2/4/8-GPU runs produce identical images),
and the finished range images are all-gathered once per batch."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local %= torch.cuda.device_count()              # (more ranks than visible devices: the one-GPU rehearsal of an N-rank run)
    if world > 1 and not dist.is_initialized():
        backend = os.environ.get("RLDM_DIST_BACKEND", backend)      # e.g. gloo: RCCL refuses two ranks on one device
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def global_sample_indices(iteration, batch_size, rank, world):
    """Indices of the images rank `rank` produces in outer iteration `iteration` -- ldm/inference.py:174-183."""
    base = (rank + world * iteration) * batch_size
    return list(range(base, base + batch_size))


def shard_range(total, rank, world):
    """Contiguous [lo, hi) slice of `total` samples for strong-scaling a fixed global batch."""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


class _GroupExchange:
    """Bootstrap traffic of the C-ABI communicator over an initialised torch.distributed process group (any backend)."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def agree(self, ok, tag):
        box = [None] * self.world
        dist.all_gather_object(box, bool(ok))
        return all(box)

    def broadcast_uid(self, raw):
        box = [raw if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def finish(self):
        pass                                             # (the collectives above are their own rendezvous)


# The TCPStore of the default path (MASTER_ADDR : MASTER_PORT + 1), one per process and endpoint: rank 0 HOSTS it and must keep it while
# a slower peer may still be reading its last key, and a second Communicator built later in the same process reuses it (the keys carry a
# generation, so the bootstraps never read each other's answers) instead of trying to bind the port a second time.
_LIVE_STORES = {}


class _StoreExchange:
    """... without a process group: a TCPStore at MASTER_ADDR:MASTER_PORT + 1 (rank 0 hosts it).  Every wait is bounded by
    `timeout` seconds: a rank that never arrives is a bootstrap failure on all the others, not a hang.
    Keys carry a GENERATION (this rank's n-th bootstrap on this store: `store.add` on a per-rank counter; the ranks build their
    communicators in the same order, so generation n is the same bootstrap everywhere) -- a second Communicator on the same store
    never reads the first one's answers.  `finish()` is the last thing a bootstrap does, on success and on failure: every rank
    posts a `done` key and rank 0, which hosts the store, waits for all of them before it returns or raises, so the server cannot
    disappear under a peer's pending read."""

    def __init__(self, rank, world, store=None, timeout=120.0):
        import datetime
        self.rank, self.world, self.timeout = rank, world, float(timeout)
        if store is None:
            addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + 1
            store = _LIVE_STORES.get((addr, port, rank, world))
            if store is None:
                store = dist.TCPStore(addr, port, world, rank == 0, timeout=datetime.timedelta(seconds=timeout))
                _LIVE_STORES[(addr, port, rank, world)] = store
        self.store = store
        self.gen = int(self.store.add(f"rldm_comm_gen_r{rank}", 1))

    def _key(self, what, rank=None):
        return f"rldm_comm_g{self.gen}_{what}" + ("" if rank is None else f"_{rank}")

    def agree(self, ok, tag):
        self.store.set(self._key(tag, self.rank), b"1" if ok else b"0")
        return all(self.store.get(self._key(tag, r)) == b"1" for r in range(self.world))

    def broadcast_uid(self, raw):
        if self.rank == 0:
            self.store.set(self._key("uid"), raw)
        return self.store.get(self._key("uid"))

    def finish(self):
        self.store.set(self._key("done", self.rank), b"1")
        if self.rank == 0:
            # ONE deadline for all ranks (a failed bootstrap must not cost world x timeout before its error surfaces)
            import datetime
            import time
            keys = [self._key("done", r) for r in range(self.world)]
            left = max(1.0, self.timeout)
            t0 = time.monotonic()
            try:
                self.store.wait(keys, datetime.timedelta(seconds=left))
            except Exception:
                pass                                     # (a rank never arrived: the caller already holds the error to raise)
            finally:
                _ = time.monotonic() - t0


class CommunicatorUnavailable(RuntimeError):
    """The C-ABI communicator cannot be used -- on THIS rank or on a peer; every rank raises it together (see Communicator)."""


class Communicator:
    """RCCL communicator behind the C ABI (include/rangeldm_hip.h: rldm_comm_*).  Replaces the reference's rank arithmetic over
    files (ldm/inference.py:56,159-183) with one all-gather.  Bootstrap, in three steps so that the ranks never disagree about the
    path they use (round 4; before, a rank whose librccl could not be bound fell back to torch.distributed ALONE while its peers
    blocked inside ncclCommInitRank or issued RCCL collectives against its torch ones):
      1. local and fallible: every rank loads librangeldm_hip and binds RCCL (`rldm_comm_bind`); rank 0 alone makes the unique
         id (`rldm_comm_unique_id`: ncclGetUniqueId opens a listening socket and a root thread, which only the rank whose id is
         used should own).  A missing .so, a missing librccl, a failed id: all are outcomes of THIS step, not exceptions;
      2. agreement: the ranks exchange their step-1 outcome (`agree`); if ANY failed, ALL raise CommunicatorUnavailable -- nobody
         enters the collective init;
      3. collective: rank 0's 128-byte id is broadcast, every rank calls `rldm_comm_create` (ncclCommInitRank) -- a rank that
         received a malformed id does NOT raise in between (its peers would block in the init) but skips the init and reports a
         failure -- and a second agreement round confirms that all of them hold a communicator (else all destroy theirs and raise).
    Nothing between the first agreement round and the last can raise on one rank alone; a failure of the exchange itself (the
    TCPStore cannot be reached, a peer never arrives within the bounded wait) surfaces as CommunicatorUnavailable on every rank
    that is waiting.  The traffic goes through torch.distributed's object collectives (any backend; gloo is enough) or, without a
    process group, through a TCPStore at MASTER_ADDR:MASTER_PORT + 1.  Collectives are issued on the CURRENT torch stream of the
    device, i.e. stream-ordered behind the sampler / trainer launches.  `lib` / `exchange` are injectable for the CPU tests."""

    def __init__(self, rank=None, world=None, store=None, lib=None, exchange=None):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        self._cdll = lib
        self._injected = lib is not None
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else int(os.environ.get("RANK", "0"))
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else int(os.environ.get("WORLD_SIZE", "1"))
        self.rank, self.world = rank, world
        self._h = None
        err = None
        try:
            self._bootstrap(rank, world, store, exchange)
        except CommunicatorUnavailable as e:
            err = e
        except Exception as e:                           # (the exchange itself failed: store unreachable, a peer timed out)
            self.close()
            err = CommunicatorUnavailable(f"rank {rank}: bootstrap exchange failed: {e!r}")
        ex = getattr(self, "_exchange", None)
        if ex is not None:
            try:
                ex.finish()                              # (rank 0 keeps its store until every rank is through, also on failure)
            except Exception:
                pass
        if err is not None:
            raise err

    def _bootstrap(self, rank, world, store, exchange):
        C = self._C
        # 1. local: load the library, bind RCCL, (rank 0) make an id -- every failure here is an OUTCOME the peers learn in step 2
        uid = C.create_string_buffer(128)
        err = None
        try:
            if self._cdll is None:
                self._cdll = self._lib.lib()
            if rank == 0:
                self._check(self._cdll.rldm_comm_unique_id(uid, 128), "rldm_comm_unique_id")
            else:
                self._check(self._cdll.rldm_comm_bind(), "rldm_comm_bind")
        except Exception as e:
            err = e
        if exchange is None and world > 1:
            exchange = _GroupExchange(rank, world) if dist.is_initialized() else _StoreExchange(rank, world, store)
        self._exchange = exchange                        # (rank 0 hosts the TCPStore: it must outlive the peers' last reads)
        # 2. agreement before anything collective
        if world > 1 and not exchange.agree(err is None, "bind"):
            raise CommunicatorUnavailable(f"rank {rank}: RCCL could not be bound on "
                                          f"{'this rank: ' + str(err) if err else 'another rank'}; no rank creates a communicator")
        if err is not None:
            raise CommunicatorUnavailable(str(err))
        # 3. collective init with rank 0's id, then confirm
        if world > 1:
            raw = exchange.broadcast_uid(uid.raw if rank == 0 else None)
            if not isinstance(raw, (bytes, bytearray)) or len(raw) != 128:
                err = RuntimeError(f"unique id of {len(raw) if hasattr(raw, '__len__') else '?'} bytes (expected 128)")
            else:
                uid = C.create_string_buffer(bytes(raw), 128)
        if err is None:
            h = C.c_void_p()
            try:
                self._check(self._cdll.rldm_comm_create(uid, rank, world, C.byref(h)), "rldm_comm_create")
                self._h = h
            except Exception as e:
                err = e
        if world > 1 and not exchange.agree(err is None, "init"):
            self.close()
            raise CommunicatorUnavailable(f"rank {rank}: rldm_comm_create failed on "
                                          f"{'this rank: ' + str(err) if err else 'another rank'}; every rank falls back together")
        if err is not None:
            raise CommunicatorUnavailable(str(err))

    def _check(self, rc, what):
        if rc == 0:
            return
        if self._injected:                               # (tests: a fake library has no rldm_last_error)
            raise RuntimeError(f"{what} failed ({rc})")
        self._lib.check(rc, what)

    def rccl_origin(self):
        buf = self._C.create_string_buffer(256)
        self._check(self._cdll.rldm_comm_info(self._h, None, None, buf, 256), "rldm_comm_info")
        return buf.value.decode()

    def all_gather_images(self, local_images):
        """fp32 on the wire (the C ABI moves floats: rldm_allgather_images); other dtypes are cast for the transfer and cast back, so
        the result has the input's dtype like the torch.distributed path."""
        x = local_images.contiguous().float()
        out = torch.empty((self.world * x.shape[0], *x.shape[1:]), dtype=torch.float32, device=x.device)
        self._check(self._cdll.rldm_allgather_images(self._h, self._C.c_void_p(x.data_ptr()),
                                                              self._C.c_void_p(out.data_ptr()), x.numel(),
                                                              self._lib.stream_ptr(x.device)), "rldm_allgather_images")
        return out if local_images.dtype == torch.float32 else out.to(local_images.dtype)

    def all_reduce_grads(self, flat, average=True):
        """in place over a contiguous fp32 slice of the flat gradient buffer (one bucket)"""
        assert flat.is_contiguous() and flat.dtype == torch.float32
        self._check(self._cdll.rldm_allreduce_grads(self._h, self._C.c_void_p(flat.data_ptr()), flat.numel(),
                                                             1 if average else 0, self._lib.stream_ptr(flat.device)),
                        "rldm_allreduce_grads")
        return flat

    def info(self):
        """(rank, world, origin of the RCCL copy) as the communicator itself reports them (rldm_comm_info)."""
        C = self._C
        r, w, buf = C.c_int(-1), C.c_int(-1), C.create_string_buffer(256)
        self._check(self._cdll.rldm_comm_info(self._h, C.byref(r), C.byref(w), buf, 256), "rldm_comm_info")
        return r.value, w.value, buf.value.decode()

    def close(self):
        """Destroy the communicator NOW (drivers call this before they exit: at interpreter shutdown torch may already have torn
        down RCCL / HIP under a destructor)."""
        if getattr(self, "_h", None):
            self._cdll.rldm_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_COMM = None
_COMM_FAILED = None


def _env_world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else int(os.environ.get("WORLD_SIZE", "1"))


def collective_choice():
    """"cabi" (rldm_allgather_images / rldm_allreduce_grads: RCCL on the CURRENT stream, i.e. stream-ordered behind the sampler's /
    trainer's launches with no hand-off to a communication stream) or "torch" (torch.distributed).  Default: cabi whenever more
    than one rank drives real GPUs over RCCL; torch for one rank, CPU tensors and the gloo rehearsals (RCCL refuses two ranks on
    one device).  RLDM_COLLECTIVE=cabi|torch overrides."""
    env = os.environ.get("RLDM_COLLECTIVE")
    if env in ("cabi", "torch"):
        return env
    if not torch.cuda.is_available() or _env_world() <= 1:
        return "torch"
    if dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl":
        return "torch"
    return "cabi"


def cabi_communicator():
    """The process-wide Communicator of the C-ABI path, or None when collective_choice() says torch.  Works without a process
    group too (RANK / WORLD_SIZE / MASTER_* from the environment: the id travels through a TCPStore).  If RCCL cannot be bound or
    bootstrapped ON ANY RANK, every rank learns it in the bootstrap's agreement rounds (Communicator), the failure is printed once
    and ALL ranks use torch.distributed (`comm_info()["collective"]` says which) -- unless RLDM_COLLECTIVE=cabi or
    RLDM_REQUIRE_CABI=1 (bench.py sets it for --gpus N > 1) ask for an error instead of a fall-back."""
    global _COMM, _COMM_FAILED
    if collective_choice() != "cabi" or not torch.cuda.is_available() or _COMM_FAILED:
        return None
    if _COMM is None:
        try:
            _COMM = Communicator()
        except Exception as e:                          # (a missing librccl, a bootstrap error: never silently, never on one rank alone)
            _COMM_FAILED = str(e)
            import sys
            print(f"rangeldm_amd.distributed: C-ABI RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
            if os.environ.get("RLDM_COLLECTIVE") == "cabi" or os.environ.get("RLDM_REQUIRE_CABI") == "1":
                raise
            return None
    return _COMM


def close():
    """Tear the C-ABI communicator down explicitly (see Communicator.close)."""
    global _COMM
    if _COMM is not None:
        _COMM.close()
        _COMM = None


def all_gather_images(local_images, world=None):
    """All-gather finished (B_local, C, W, H) tensors along dim 0; every rank gets the full batch in rank order.
    One collective per batch: 8 x 1 MiB for BASELINE config 2/3 -- latency-bound, negligible vs the sampling time."""
    if _env_world() == 1:
        return local_images
    comm = cabi_communicator() if local_images.is_cuda else None
    if comm is not None:
        return comm.all_gather_images(local_images)
    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("all_gather_images: WORLD_SIZE > 1 but neither a torch.distributed process group nor the C-ABI "
                           "communicator is available")
    world = dist.get_world_size()
    local_images = local_images.contiguous()
    if local_images.is_cuda and dist.get_backend() == "gloo":
        # the one-GPU rehearsal of an N-rank run (RLDM_DIST_BACKEND=gloo: RCCL refuses two ranks on one device); gloo gathers host tensors
        host = local_images.cpu()
        out = torch.empty((world * host.shape[0], *host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host)
        return out.to(local_images.device)
    out = torch.empty((world * local_images.shape[0], *local_images.shape[1:]), dtype=local_images.dtype,
                      device=local_images.device)
    dist.all_gather_into_tensor(out, local_images)
    return out


def comm_info(device=None):
    """What a scaling record needs to prove that N ranks really met: the world the process group reports, the ranks that answered
    a collective (every rank contributes its id to an all-gather through the SAME path the images take; `ranks_seen` lists them),
    which path that was, and -- C-ABI path -- rank / world / library origin as the RCCL communicator itself reports them."""
    world = _env_world()
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else int(os.environ.get("RANK", "0"))
    info = {"world": world, "rank": rank, "backend": dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None,
            "collective": "none" if world == 1 else "torch", "ranks_seen": [rank]}
    if world == 1:
        return info
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                                             else torch.device("cpu"))
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        dev_t = torch.device("cpu")
    else:
        dev_t = dev
    mine = torch.full((1, 1), float(rank), dtype=torch.float32, device=dev_t)
    comm = cabi_communicator() if mine.is_cuda else None
    got = all_gather_images(mine)
    info["ranks_seen"] = [int(v) for v in got.flatten().tolist()]
    if comm is not None:
        r, w, origin = comm.info()
        info.update(collective="cabi", rccl_rank=r, rccl_world=w, rank0_rccl_origin=origin)
    elif _COMM_FAILED:
        info["cabi_unavailable"] = _COMM_FAILED
    return info


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
