"""Test-side helpers for sharded populations: the infector / infectee links of a set of in-process shards as global pairs,
whatever the engine (HIP or the CPU checker) and whatever the attribution mode."""
import numpy as np

from reina_model_amd import engine as eng


def _host(ctx, name):
    t = ctx.engine.tensors[name]
    return np.asarray(ctx.engine.alloc.to_host(t)) if not isinstance(t, np.ndarray) else t


def list_pairs(ctx):
    """sorted (infector index, infectee id) pairs of every infectee list of one engine: the inline slots (-1 = empty) and the
    overflow list -- threaded through the infectees' own records, or (exact attribution) through the nodes of the pool"""
    inline = _host(ctx, 'infectees').reshape(-1, eng.INLINE_INFECTEES)
    first = _host(ctx, 'first_infectee')
    o, k = np.nonzero(inline >= 0)
    pairs = [o.astype(np.int64) * (1 << 32) + inline[o, k].astype(np.int64)]
    owner = np.nonzero(first >= 0)[0].astype(np.int64)
    cur = first[owner].astype(np.int64)
    exact = bool(ctx.engine.config.exact_attribution)
    if exact:
        pool = _host(ctx, 'infectee_pool').view(np.int32).reshape(-1, 2)
    else:
        nxt_sib = _host(ctx, 'next_sibling')
    for _ in range(70):
        if len(cur) == 0:
            break
        if exact:
            pairs.append(owner * (1 << 32) + pool[cur, 0].astype(np.int64))
            nxt = pool[cur, 1].astype(np.int64)
        else:
            pairs.append(owner * (1 << 32) + cur)
            nxt = nxt_sib[cur].astype(np.int64)
        keep = nxt >= 0
        owner, cur = owner[keep], nxt[keep]
    assert len(cur) == 0, 'an infectee list longer than 64 entries (or a cycle)'
    return np.sort(np.concatenate(pairs))


def assert_links_are_true(contexts):
    """EXACT attribution's defining property (the reference: person_infect records the true infector and appends to ITS
    infectee array, main.pyx:219-233): over all shards together, an agent's infection count equals the number of agents that
    name it as their infector, and every entry of an infectee list is such an agent, listed once."""
    G = len(contexts)
    named = []     # (infector gid, infectee gid) from the infectees' side
    for r, c in enumerate(contexts):
        infector = _host(c, 'infector')
        idx = np.nonzero(infector >= 0)[0]
        named.append(infector[idx].astype(np.int64) * (1 << 32) + ((r << eng.GID_SHIFT) | idx).astype(np.int64))
    named = np.sort(np.concatenate(named))
    assert len(np.unique(named)) == len(named)
    sources, counts = np.unique(named >> 32, return_counts=True)
    got = {}
    for r, c in enumerate(contexts):
        n_inf = _host(c, 'n_infected')
        i = np.nonzero(n_inf)[0]
        for g, n in zip(((r << eng.GID_SHIFT) | i).tolist(), n_inf[i].tolist()):
            got[g] = n
    want = dict(zip(sources.tolist(), counts.tolist()))
    assert got == want, 'infection counts differ from the number of agents naming the source'
    listed = []
    for r, c in enumerate(contexts):
        p = list_pairs(c)
        listed.append(((p >> 32) | (r << eng.GID_SHIFT)) * (1 << 32) + (p & 0xFFFFFFFF))
    listed = np.concatenate(listed)
    assert len(np.unique(listed)) == len(listed), 'an infectee listed twice'
    assert np.isin(listed, named).all(), 'an infectee list holds an agent that does not name its owner as infector'
    cross = int(((named >> 32 >> eng.GID_SHIFT) != ((named & 0xFFFFFFFF) >> eng.GID_SHIFT)).sum())
    return dict(links=len(named), cross_shard=cross, listed=len(listed))
