"""Worker for tests/test_distributed.py: one rank of a sharded run over torch.distributed (gloo on
CPU with the oracle-B engine; nccl on GPUs with the HIP engine).  Rank 0 also recomputes the same
sharded run with all shards in-process and checks that the distributed result is identical."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import torch
    import torch.distributed as dist
    backend = sys.argv[1] if len(sys.argv) > 1 else 'gloo'
    days = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    total = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank)))
        dist.init_process_group('nccl', device_id=torch.device('cuda', int(os.environ.get('LOCAL_RANK', rank))))
    else:
        dist.init_process_group('gloo')
    from reina_model_amd import datasets, sharding, simulation
    from reina_model_amd import engine as eng
    if len(sys.argv) > 4 and sys.argv[4] == 'rccl_fail':
        # DirectRccl's construction is collective: a failure on ONE rank (library not loadable / unique id
        # not created) must make EVERY rank raise, with the process group still usable afterwards
        class FakeLib:
            def __init__(self, uid_rc):
                self.uid_rc = uid_rc

            def ncclGetUniqueId(self, uid):
                return self.uid_rc

        def failing_loader():
            raise OSError('librccl.so: cannot open shared object file (injected)')

        cases = [('loader fails on the last rank', lambda: failing_loader() if rank == world - 1 else FakeLib(0)),
                 ('loader fails on rank 0', lambda: failing_loader() if rank == 0 else FakeLib(0)),
                 ('unique id fails on rank 0', lambda: FakeLib(7))]
        for name, loader in cases:
            try:
                sharding.DirectRccl(dist, None, rank, world, lib_loader=loader)
                raise AssertionError('%s: rank %d did not raise' % (name, rank))
            except (RuntimeError, OSError):
                pass
            t = torch.tensor([rank + 1], dtype=torch.int32)   # the group is intact: same collective on every rank
            dist.all_reduce(t)
            assert int(t.item()) == world * (world + 1) // 2, name
        if rank == 0:
            print('RCCL_FAIL_OK world=%d' % world, flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    import par_backend
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    v.update(hospital_beds=25, icu_units=3)
    ages = datasets.scaled_population(total)
    factory = None if backend == 'nccl' else par_backend.par_engine_factory
    device = 'cuda:%d' % int(os.environ.get('LOCAL_RANK', rank)) if backend == 'nccl' else 'cpu'
    attribution = os.environ.get('REINA_TEST_ATTRIBUTION', 'exact')   # cross-shard infector links (sharding.py)
    comm = sharding.TorchComm(attribution=attribution)
    comm.always_collective = True  # exercise the phases and their collectives even when world == 1
    if len(sys.argv) > 4 and sys.argv[4] == 'instream' and backend == 'gloo':
        # the engine's OWN day loop (step_day: the phases in C, the collectives called through the function pointers of
        # reina_set_collective / reina_set_alltoall) across real processes: the pointers are ctypes callbacks that run the gloo
        # collectives on the host buffers -- what DirectRccl's ncclAllReduce / ncclAllToAll are to the HIP engine
        import ctypes

        def _allreduce(send, recv, count, dtype, op, comm_, stream):
            assert dtype == 2 and op == 0 and send == recv
            buf = np.ctypeslib.as_array(ctypes.cast(recv, ctypes.POINTER(ctypes.c_int32)), shape=(count,))
            t = torch.from_numpy(buf)
            dist.all_reduce(t)
            return 0

        def _alltoall(send, recv, count, dtype, comm_, stream):
            assert dtype == 4
            s_ = torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(send, ctypes.POINTER(ctypes.c_int64)), shape=(count * world,)))
            r_ = torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(recv, ctypes.POINTER(ctypes.c_int64)), shape=(count * world,)))
            dist.all_to_all_single(r_, s_)
            return 0

        class FakeDirect:   # (the attributes model.Context reads of sharding.DirectRccl)
            pass
        fd = FakeDirect()
        fd._keep = (ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)(_allreduce),
                    ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)(_alltoall))
        fd.fn_ptr = ctypes.cast(fd._keep[0], ctypes.c_void_p).value
        fd.a2a_ptr = ctypes.cast(fd._keep[1], ctypes.c_void_p).value
        fd.comm_ptr = 1
        fd.count = lambda: world
        comm.direct = fd
    if rank == 0:
        print('direct_rccl=%s' % (comm.direct is not None), flush=True)
    if comm.direct is not None and backend == 'nccl':
        # the exchange of exact attribution as the engine queues it: RCCL's ncclAllToAll through the function pointer
        # reina_set_alltoall receives (count per peer, ncclInt64 = 4), on the day stream, with this rank's own communicator --
        # every rank's segment k must arrive as segment (rank) at rank k
        import ctypes
        n = 1000
        send = (torch.arange(world * n, dtype=torch.int64, device='cuda') + 1000003 * rank)
        recv = torch.zeros(world * n, dtype=torch.int64, device='cuda')
        fn = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)(comm.direct.a2a_ptr)
        rc = fn(send.data_ptr(), recv.data_ptr(), n, 4, comm.direct.comm_ptr, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0, rc
        for k in range(world):
            want = torch.arange(rank * n, (rank + 1) * n, dtype=torch.int64, device='cuda') + 1000003 * k
            assert torch.equal(recv[k * n:(k + 1) * n], want), 'ncclAllToAll through the engine hook: segment %d' % k
        if rank == 0:
            print('direct_alltoall=ok', flush=True)
    # TorchComm.all_to_all as model.Context._step calls it when the engine does not queue the exchange itself: every rank's
    # segment k arrives as segment (rank) at rank k -- HBM buffers under nccl, host arrays (uint64) under gloo
    n = 257
    if backend == 'nccl':
        snd = torch.arange(world * n, dtype=torch.int64, device='cuda') + 1000003 * rank
        rcv = torch.zeros(world * n, dtype=torch.int64, device='cuda')
    else:
        snd = (np.arange(world * n, dtype=np.uint64) + np.uint64(1000003 * rank))
        rcv = np.zeros(world * n, dtype=np.uint64)
    comm.all_to_all(snd, rcv)
    got = rcv.cpu().numpy() if backend == 'nccl' else rcv.astype(np.int64)
    for k in range(world):
        assert np.array_equal(got[k * n:(k + 1) * n], np.arange(rank * n, (rank + 1) * n) + 1000003 * k), 'TorchComm.all_to_all: segment %d' % k
    ctx = simulation.make_context(v, age_counts=ages, seed=21, engine_factory=factory, device=device, comm=comm)
    if len(sys.argv) > 4 and sys.argv[4] == 'ensemble':
        # config 5 shape: seeds partitioned over the ranks, gathered on rank 0
        from reina_model_amd import ensemble
        seeds = list(range(40, 47))
        ens = ensemble.run_ensemble_distributed(v, seeds, days, age_counts=ages, engine_factory=factory, device=device)
        if rank == 0:
            for k, sd in enumerate(seeds):
                one = simulation.make_context(v, age_counts=ages, seed=sd, engine_factory=par_backend.par_engine_factory).run(days)
                assert np.array_equal(ens[k], one), 'ensemble member (seed %d) differs from its single run' % sd
            print('ENSEMBLE_OK world=%d members=%d' % (world, len(seeds)), flush=True)
        else:
            assert ens is None
        dist.barrier()
        dist.destroy_process_group()
        return
    if len(sys.argv) > 4 and sys.argv[4] == 'instream':
        assert ctx._in_stream, 'the run must go through the engine\'s own day loop'
    hist = ctx.run(days)
    final = ctx.generate_state()
    if rank == 0:
        members = []
        ref = [simulation.make_context(v, age_counts=ages, seed=21, engine_factory=par_backend.par_engine_factory,
                                       comm=sharding.InProcessComm(r, world, members, attribution=attribution)) for r in range(world)]
        A = eng.MAX_AGES
        for d in range(days):
            expect = sharding.reduce_counters(ref)
            assert np.array_equal(hist[d], expect), 'day %d: distributed history != in-process sharded run' % d
            sharding.step_shards_together(ref)
        expect = sharding.reduce_counters(ref)
        s2 = ref[0].state_from_counters(expect)
        for k in ('susceptible', 'infected', 'dead', 'all_detected', 'new_infections'):
            assert np.array_equal(final[k], s2[k]), k
        assert final['r'] == s2['r'] and final['available_hospital_beds'] == s2['available_hospital_beds']
        n = int(np.asarray(ages).sum())
        tot = lambda name: hist[:, eng.C_NAMES.index(name) * A:(eng.C_NAMES.index(name) + 1) * A].sum(axis=1)
        assert np.all(tot('susceptible') + tot('infected') + tot('recovered') + tot('dead') == n)
        assert tot('all_infected')[-1] > 1000
        print('DIST_OK world=%d days=%d all_infected=%d attribution=%s' % (world, days, tot('all_infected')[-1], ctx.attribution), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
