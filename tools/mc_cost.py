"""Where does run_monte_carlo spend its time? (64 seeds x HUS x 365 days)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reina_model_amd import simulation
simulation.run_monte_carlo('default', seeds=range(4), days=60, write_csv=False)   # warm up
t0 = time.perf_counter()
cProfile.run("df = simulation.run_monte_carlo('default', seeds=range(64), write_csv=False)", '/tmp/mc.prof')
print('64 seeds x 365 d: %.2f s' % (time.perf_counter() - t0))
pstats.Stats('/tmp/mc.prof').sort_stats('cumtime').print_stats(14)
