import sys, os, time
sys.path.insert(0, os.getcwd())
import bench
r = bench.ensemble_line(128, 365, 'cuda:0')
print('standalone ensemble_line:', r['ms_per_step'], r['kernels'])
r = bench.ensemble_line(128, 365, 'cuda:0')
print('second ensemble_line:', r['ms_per_step'], r['kernels'])
