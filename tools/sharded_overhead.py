"""Cost of the sharded day path on one GPU: world-1 NCCL group, the per-day pressure all-reduce and
the begin / all-reduce / end split, vs the plain path.  (8-GPU runs are the round driver's.)"""
import copy, os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29617')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from reina_model_amd import datasets, sharding, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS); ages = datasets.get_population_for_area()
for mode in ('plain', 'collective', 'collective'):
    comm = None
    if mode == 'collective':
        comm = sharding.TorchComm(); comm.always_collective = True
    ctx = simulation.make_context(v, age_counts=ages, seed=0, comm=comm)
    ctx.run(30, record_history=False); ctx.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.run(365, record_history=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%-10s %.1f us/day (host returned at %.1f)' % (mode, (t2 - t0) / 365 * 1e6, (t1 - t0) / 365 * 1e6), flush=True)
dist.destroy_process_group()
