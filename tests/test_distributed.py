"""N>1 path: world_size-2 `gloo` run on CPU (oracle-B engines behind the product's host code and
`sharding.TorchComm`) must equal the same sharded run stepped in one process."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, backend, days, total, timeout=600, extra_env=None, mode=None):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', 'dist_worker.py'), backend, str(days), str(total)] + ([mode] if mode else [])
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='1')
    env.update(extra_env or {})
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.slow
@pytest.mark.parametrize('world,attribution', [(2, 'exact'), (3, 'exact'), (8, 'exact'), (2, 'mirror'), (3, 'mirror')])
def test_gloo_sharded_run_equals_in_process_sharded_run(world, attribution):
    """world 8 = the rank count of BASELINE configs[3] (one process per GPU of the node): eight ranks' pressure blocks
    through one all-reduce per day, beds / ICU units / imports split eight ways.  exact: the contact, feedback and tracing
    records through torch.distributed.all_to_all_single between the phases of every day (gloo == in-process)"""
    r = _launch(world, 'gloo', 130 if world < 8 else 70, 40000, timeout=900, extra_env={'REINA_TEST_ATTRIBUTION': attribution})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'DIST_OK world=%d' % world in r.stdout
    assert 'attribution=%s' % (attribution if world > 1 else 'none') in r.stdout


@pytest.mark.parametrize('attribution', ['exact', 'mirror'])
def test_the_engines_own_day_loop_with_the_collectives_behind_function_pointers(attribution):
    """reina_step_day / reina_run_days run the phases of a day in C and call the collectives through the pointers of
    reina_set_collective / reina_set_alltoall (RCCL's ncclAllReduce / ncclAllToAll on the day stream in production).  Here, with the
    CPU checker's same loop (par_step_day) and ctypes callbacks that run the gloo collectives: world 2, 130 days incl. two weeks of
    contact tracing == the same sharded run stepped phase by phase in one process."""
    r = _launch(2, 'gloo', 130, 30000, mode='instream', timeout=600, extra_env={'REINA_TEST_ATTRIBUTION': attribution})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'DIST_OK world=2' in r.stdout and 'attribution=%s' % attribution in r.stdout


def test_gloo_ensemble_is_partitioned_over_the_ranks():
    """BASELINE config 5 (replicas only): 7 seeds over 2 ranks, gathered on rank 0, every member identical
    to its single run"""
    r = _launch(2, 'gloo', 40, 8000, mode='ensemble')
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'ENSEMBLE_OK world=2 members=7' in r.stdout


def test_direct_rccl_construction_fails_on_every_rank_together():
    """a rank that cannot load RCCL or create the unique id must not leave its peers inside a
    mismatched collective: every rank raises, and the process group keeps working (gloo, world 2, faults
    injected through DirectRccl's lib_loader hook)."""
    r = _launch(2, 'gloo', 0, 0, mode='rccl_fail', timeout=180)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'RCCL_FAIL_OK world=2' in r.stdout


def test_population_split_is_a_partition():
    import numpy as np
    from reina_model_amd import datasets, sharding
    ages = datasets.get_population_for_area()
    for world in (2, 3, 8):
        parts = [sharding.split_population(ages, r, world) for r in range(world)]
        assert np.array_equal(np.sum(parts, axis=0), ages)
        assert max(p.sum() for p in parts) - min(p.sum() for p in parts) <= len(ages)
        assert sum(sharding.split_count(2600, r, world) for r in range(world)) == 2600


@pytest.mark.gpu
def test_nccl_single_rank_exercises_the_collective_path():
    """RCCL path on the one GPU of the test box: init_process_group('nccl'), the per-day pressure
    all-reduce on an HBM int32 tensor, host counter blocks staged through HBM.  world_size 1, so
    the result must equal the CPU oracle's unsharded run (checked by the worker)."""
    r = _launch(1, 'nccl', 80, 40000)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'DIST_OK world=1' in r.stdout
    # ... and it went through the engine's in-stream collective (our own RCCL communicator,
    # ncclAllReduce queued on the day stream by reina_step_day)
    assert 'direct_rccl=True' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # ... and RCCL's all-to-all answered through the pointer reina_set_alltoall takes (exact attribution's in-stream exchange)
    assert 'direct_alltoall=ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_nccl_single_rank_torch_distributed_fallback():
    """the same with REINA_DIRECT_RCCL=0: the per-day exchange through torch.distributed.all_reduce"""
    r = _launch(1, 'nccl', 80, 40000, extra_env={'REINA_DIRECT_RCCL': '0'})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'DIST_OK world=1' in r.stdout and 'direct_rccl=False' in r.stdout
