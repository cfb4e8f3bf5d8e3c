"""Sharding an agent population over several engine instances (one process per GPU).

SURVEY.md section 8e: agents are split so that every shard holds ~1/G of every age; state transitions
are local; the only per-day coupling is that a contact is a uniform member of an age range over the
WHOLE population.  Each shard therefore counts, per (destination shard, contact age range,
variant), the transmissible contacts it aims at other shards; one small all-reduce (2048 int32
over RCCL/xGMI, latency-bound) sums these "infection pressure" histograms, and each shard realises
the pressure aimed at it on uniformly drawn local agents.  Global scarce resources (beds, ICU
units, import and vaccination quotas) are partitioned 1/G per shard.

Infector links across shards ("mirror attribution"): the true infector of a cross-shard infection
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


class TorchComm:
    """torch.distributed wrapper: `nccl` (= RCCL on ROCm) for HBM tensors, `gloo` for host arrays."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._nccl = dist.get_backend(group) == 'nccl'   # looked up once: this sits on the per-day path

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


class InProcessComm:
    """All G shards live in ONE process and are stepped in lock-step by `step_shards_together`
    (tests; single-GPU emulation of a sharded run).  Collectives are plain sums over the members."""

    def __init__(self, rank, world, members):
        self.rank = rank
        self.world = world
        self.members = members  # shared list of Contexts, filled by the driver


def step_shards_together(contexts):
    """One day for G in-process shards: all first halves, pressure summed, all second halves."""
    days = []
    for c in contexts:
        d, changed = c._build_day()
        if changed:
            c._upload_tables()
        c.engine.step_day_begin(d)
        days.append(d)
    bufs = [c.engine.tensors['pressure'] for c in contexts]
    host = [np.array(c.engine.alloc.to_host(b), dtype=np.int64) for c, b in zip(contexts, bufs)]
    total = np.sum(host, axis=0).astype(np.int32)
    for c, b in zip(contexts, bufs):
        if isinstance(b, np.ndarray):
            b[:] = total
        else:
            b.copy_(c.engine.alloc.torch.from_numpy(total).to(b.device))
    for c, d in zip(contexts, days):
        c.engine.step_day_end(d)
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
