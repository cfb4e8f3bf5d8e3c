"""Sharding an agent population over several engine instances (one process per GPU).

SURVEY.md section 8e: agents are split so that every shard holds ~1/G of every age; state transitions
are local; the only per-day coupling is that a contact is a uniform member of an age range over the
WHOLE population.  Each shard therefore counts, per (destination shard, contact age range,
variant), the transmissible contacts it aims at other shards; one small all-reduce (2048 int32
over RCCL/xGMI, latency-bound) sums these "infection pressure" histograms, and each shard realises
the pressure aimed at it on uniformly drawn local agents.  Beds and ICU units are ONE pool over the
shards, handed out in one global order (priority bucket, shard, priority, agent) from per-bucket event
maps that ride on the same all-reduce (csrc/k_remote.inc); import and vaccination
quotas are partitioned 1/G per shard.

Infector links across shards.  EXACT attribution (the default): every link
field holds a global id (shard, index); a cross-shard contact is completed at its SOURCE (every shard's age_start table is
global knowledge, so the source draws the target and knows its age) and travels as an 8-byte record to the target's shard,
which claims the target under the source's own key; the shard of an infection returns (source, infectee) so that the
source's count and infectee list are true the next morning; contact tracing sends (candidate, tracer) requests to the
candidate's shard, one exchange per level.  The exchanges are fixed-capacity all-to-alls (RCCL: ncclAllToAll queued on the day
stream) at the points reina_step_phase names.

"Mirror attribution" (comm.attribution = 'mirror': no exchange but the one all-reduce): the true infector of a cross-shard infection
lives on another shard and is never shipped.  Shards are statistically exchangeable, so an
infection realised from incoming pressure in cell (range, variant) takes as its infector a LOCAL
source that aimed an attempt of the same cell at another shard today (sampled through a
day-tagged hash table of the outgoing attempts).  Contact tracing and `r` then see a link
structure with the same distribution as the unsharded model, with no extra communication.  The
deviation that remains: links are stand-ins, not the true pairs.
"""
import numpy as np


def split_count(total, rank, world):
    """Deterministic integer partition: ranks < total % world get one more."""
    total = int(total)
    return total // world + (1 if rank < total % world else 0)


def split_population(age_counts, rank, world):
    return np.asarray([split_count(c, rank, world) for c in age_counts], dtype=np.int64)


class DirectRccl:
    """An RCCL communicator of our own, driven through ctypes: its `ncclAllReduce` is handed to the
    engine (reina_set_collective), which queues the per-day pressure all-reduce on the DAY STREAM
    itself -- torch.distributed's all_reduce runs on a stream of its own and costs two cross-stream
    event waits plus Python time per day.  The library is the RCCL PyTorch itself loaded
    (torch/lib/librccl.so); the unique id travels through the existing torch.distributed group."""

    def __init__(self, dist, group, rank, world, lib_loader=None):
        """Collective: every rank of `group` must call this together.  No rank raises before ALL ranks
        have left the last collective of the construction, so a local failure (library not loadable,
        ncclGetUniqueId error) can never leave the others inside a mismatched collective: the phases are
        (1) local preparation, failures recorded; (2) the unique id -- or a failure marker -- is broadcast,
        every rank joins; (3) an all-reduce (MIN) agrees whether EVERY rank is ready; only then
        (4) ncclCommInitRank, entered by all ranks or by none; (5) a second agreement on its outcome."""
        import ctypes
        import os
        import torch
        self.comm = None
        self.lib = None
        err = None

        class UniqueId(ctypes.Structure):
            _fields_ = [('internal', ctypes.c_char * 128)]

        uid = UniqueId()
        try:   # (1) nothing here communicates
            path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
            self.lib = ctypes.CDLL(path) if lib_loader is None else lib_loader()   # (lib_loader: test hook)
            if rank == 0:
                rc = self.lib.ncclGetUniqueId(ctypes.byref(uid))
                if rc != 0:
                    raise RuntimeError('ncclGetUniqueId failed: %d' % rc)
        except Exception as e:   # noqa: BLE001 -- reported after the agreement below
            err = e
        # (2) also with a single rank: the one-GPU test box then exercises the call
        box = [(bytes(bytearray(uid)) if err is None else None) if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if box[0] is None and err is None:
            err = RuntimeError('rank 0 could not create an RCCL unique id')

        def agree(ok):   # (3) / (5): MIN over the ranks, on the backend's own device
            dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return int(t.item()) == 1

        if not agree(err is None):
            raise err if err is not None else RuntimeError('another rank could not prepare its RCCL communicator')
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        comm = ctypes.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        rc = self.lib.ncclCommInitRank(ctypes.byref(comm), int(world), uid, int(rank))   # (4)
        if rc == 0:
            self.comm = comm
        if not agree(rc == 0):
            self.close()
            raise RuntimeError('ncclCommInitRank failed: %d' % rc if rc != 0 else 'ncclCommInitRank failed on another rank')
        self.fn_ptr = ctypes.cast(self.lib.ncclAllReduce, ctypes.c_void_p).value
        # exact attribution's exchanges: RCCL's own all-to-all (grouped ncclSend / ncclRecv inside the library), same stream
        self.a2a_ptr = ctypes.cast(self.lib.ncclAllToAll, ctypes.c_void_p).value
        self.comm_ptr = self.comm.value

    def count(self):
        """ranks RCCL itself reports for this communicator (ncclCommCount)"""
        import ctypes
        n = ctypes.c_int(0)
        self.lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        rc = self.lib.ncclCommCount(self.comm, ctypes.byref(n))
        return int(n.value) if rc == 0 else -1

    def close(self):
        import ctypes
        if getattr(self, 'comm', None) is not None and self.comm.value:
            self.lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


# Cross-shard infector links: the ONE place the default is decided.  'exact': the true
# infector of every cross-shard infection, as in the reference (main.pyx:219-233) -- contact / feedback / tracing records through
# all-to-all segments, two collectives a day (four on contact-tracing days; round 6: the shards' capacity words and event maps ride in the
# segments' trailers, no all-reduce); 'mirror': stand-in infectors, ONE all-reduce a day (north_star's exchange).
# The library's comm objects default to it; model.Context reads the comm's `attribution` (a comm object without one gets it too and
# must then provide all_to_all: checked at construction); bench.py --gpus N measures 'mirror' unless told otherwise and labels
# its line with the mode it ran (config.attribution) -- the one-collective day is what BASELINE.json's north_star describes.
DEFAULT_ATTRIBUTION = 'exact'


class TorchComm:
    """torch.distributed wrapper: `nccl` (= RCCL on ROCm) for HBM tensors, `gloo` for host arrays.
    With the nccl backend the per-day exchange bypasses torch (DirectRccl, REINA_DIRECT_RCCL=0 turns
    that off); counter reductions at export time keep using torch.distributed."""

    def __init__(self, group=None, attribution=DEFAULT_ATTRIBUTION):
        import torch
        import torch.distributed as dist
        self.attribution = attribution
        self.torch = torch
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._nccl = dist.get_backend(group) == 'nccl'   # looked up once: this sits on the per-day path
        self.direct = None
        import os
        if self._nccl and os.environ.get('REINA_DIRECT_RCCL', '1') != '0':
            # DirectRccl's construction is itself collective and ends with an agreement: either every rank
            # holds a communicator afterwards or every rank has raised, so all take the same route (one rank
            # on torch.distributed while the others sit in the direct communicator's all-reduce would hang
            # the day loop)
            try:
                self.direct = DirectRccl(dist, group, self.rank, self.world)
            except Exception as e:   # noqa: BLE001 -- all ranks land here together
                import sys
                print('reina: direct RCCL communicator unavailable (%s); using torch.distributed' % e, file=sys.stderr)
                self.direct = None

    def _as_tensor(self, buf):
        if isinstance(buf, np.ndarray):
            return self.torch.from_numpy(buf)  # shares memory: the reduction lands in the array
        return buf

    def _all_reduce(self, buf, op):
        t = self._as_tensor(buf)
        if self._nccl and not t.is_cuda:
            # RCCL reduces device memory only: stage host-side counter blocks through HBM
            tmp = t.to(self.torch.device('cuda', self.torch.cuda.current_device()))
            self.dist.all_reduce(tmp, op=op, group=self.group)
            t.copy_(tmp.cpu())
        else:
            self.dist.all_reduce(t, op=op, group=self.group)

    def all_reduce_sum(self, buf):
        self._all_reduce(buf, self.dist.ReduceOp.SUM)

    def all_reduce_max(self, buf):
        self._all_reduce(buf, self.dist.ReduceOp.MAX)

    def all_to_all(self, send, recv):
        """segment d of every rank's `send` -> segment (sender's rank) of rank d's `recv` (exact attribution's record
        exchanges when the engine does not queue them itself: gloo, or REINA_DIRECT_RCCL=0)"""
        s, r = self._as_tensor(send), self._as_tensor(recv)
        if s.dtype == self.torch.uint64:   # (host arrays of the CPU checker: gloo moves signed words)
            s, r = s.view(self.torch.int64), r.view(self.torch.int64)
        if self._nccl and not s.is_cuda:
            # RCCL moves device memory only: stage host buffers through HBM
            dev = self.torch.device('cuda', self.torch.cuda.current_device())
            ts, tr = s.to(dev), r.to(dev)
            self.dist.all_to_all_single(tr, ts, group=self.group)
            r.copy_(tr.cpu())
        elif not self._nccl and s.is_cuda:
            # gloo exchanges host memory: HBM buffers go through the host (two ranks sharing one GPU: a plumbing check)
            ts, tr = s.cpu(), r.cpu()
            self.dist.all_to_all_single(tr, ts, group=self.group)
            r.copy_(tr.to(r.device))
        else:
            self.dist.all_to_all_single(r, s, group=self.group)


class InProcessComm:
    """All G shards live in ONE process and are stepped in lock-step by `step_shards_together`
    (tests; single-GPU emulation of a sharded run).  Collectives are plain sums over the members."""

    def __init__(self, rank, world, members, attribution=DEFAULT_ATTRIBUTION):
        self.rank = rank
        self.world = world
        self.members = members  # shared list of Contexts, filled by the driver
        self.attribution = attribution


def step_shards_together(contexts, pool=None):
    """One day for G in-process shards, phase by phase (include/reina_hip.h: reina_step_phase): every shard runs the phase,
    then the collectives the phase asked for are carried out between them -- the pressure blocks summed, the exchange
    segments swapped.  `pool`: an executor whose map() runs the shards' phases side by side (host engines: the C calls
    release the GIL)."""
    from . import engine as eng
    days = []
    for c in contexts:
        d, changed = c._build_day()
        if changed:
            c._upload_tables()
        days.append(d)
    for ph in range(eng.PH_NR):
        need = set((pool.map if pool is not None else map)(lambda cd: cd[0].engine.step_phase(cd[1], ph), list(zip(contexts, days))))
        assert len(need) == 1, 'the shards disagree about the collectives of phase %d: %s' % (ph, need)
        need = need.pop()
        if need & eng.X_ALLREDUCE:
            bufs = [c.engine.tensors['pressure'] for c in contexts]
            host = [np.array(c.engine.alloc.to_host(b), dtype=np.int64) for c, b in zip(contexts, bufs)]
            total = np.sum(host, axis=0).astype(np.int32)
            for c, b in zip(contexts, bufs):
                if isinstance(b, np.ndarray):
                    b[:] = total
                else:
                    b.copy_(c.engine.alloc.torch.from_numpy(total).to(b.device))
        if need & eng.X_ALLTOALL:
            G = len(contexts)
            send = [np.asarray(c.engine.alloc.to_host(c.engine.tensors['xsend'])).reshape(G, -1) for c in contexts]
            for r, c in enumerate(contexts):
                got = np.ascontiguousarray(np.stack([send[s][r] for s in range(G)]).reshape(-1))
                b = c.engine.tensors['xrecv']
                if isinstance(b, np.ndarray):
                    b[:] = got.view(b.dtype)
                else:
                    b.copy_(c.engine.alloc.torch.from_numpy(got.view(np.int64)).to(b.device))
    for c in contexts:
        c.day += 1


def reduce_counters(contexts):
    """Global counter block of G in-process shards (what TorchComm's all-reduces produce)."""
    from . import engine as eng
    base = eng.C_NR * eng.MAX_AGES
    rows = np.array([c.engine.read_counters() for c in contexts], dtype=np.int64)
    out = rows.sum(axis=0)
    out[base + eng.S_PROBLEM] = rows[:, base + eng.S_PROBLEM].max()
    out[base + eng.S_DAY] = rows[0, base + eng.S_DAY]
    return out.astype(np.int32)
