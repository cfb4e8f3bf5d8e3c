"""hipGraph replay of a chunk of days against the eager day loop (HUS by default): does capturing the launch-bound
inner loop buy anything?  The round-1 verdict asked for this measurement (item 6).

A stretch of 64 consecutive days without a contact-table change is taken from the plan of the default scenario
(days >= 150, mid-epidemic).  EAGER: reina_run_days_hist issues its launches on the stream.  GRAPH: the very same
call is captured once (torch.cuda.graph: hipStreamBeginCapture on the day stream; day descriptors are by-value
kernel arguments, so they live in the graph's nodes) and replayed.  A replay re-runs the same day NUMBERS on the
state the previous replay left -- not a valid simulation (same-day claim tags repeat), but the same kernels with
the same kind of work: a timing experiment only.  Reported: GPU wall per day (sync to sync over `reps` x 64
days) and host time per day to issue the work.

usage: python tools/graph_replay.py [agents (0 = HUS)] [reps]"""
import copy
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

import bench
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
v = copy.deepcopy(VARIABLE_DEFAULTS)
if n:
    v, ages = bench.scaled_scenario(v, n)
else:
    ages = datasets.get_population_for_area()
ctx = simulation.make_context(v, age_counts=ages, seed=0)
plan = ctx.make_plan(365)
eng = ctx.engine
# run the scenario eagerly up to the first >= 64-day stretch of unchanged tables that starts at day >= 150
done, chunk = 0, None
for tables, arr, cnt in plan['segments']:
    if tables is not None:
        eng.upload_contact_tables(*tables)
    skip = max(0, 150 - done)
    if chunk is None and cnt - skip >= 64:
        if skip:
            eng.run_day_array((type(arr[0]) * skip)(*arr[:skip]), skip, None)
        chunk = (type(arr[0]) * 64)(*arr[skip:skip + 64])
        first_day = done + skip
        break
    eng.run_day_array(arr, cnt, None)
    done += cnt
assert chunk is not None, 'no 64-day stretch without a table change after day 150'
assert [d.day for d in chunk] == list(range(first_day, first_day + 64)), 'chunk is not 64 consecutive real days'
torch.cuda.synchronize()
print('agents %d; chunk = days %d..%d' % (int(sum(ages)), first_day, first_day + 63))


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t_host, time.perf_counter() - t0


eager = lambda: eng.run_day_array(chunk, 64, None)
# replays evolve the state (the epidemic burns out under repeated days), and with it the work per day: replay until
# the day time has settled, then ALTERNATE eager and graph rounds so that what drift is left hits both alike
prev = None
for k in range(12):
    h, w = timed(eager, reps)
    cur = w / reps / 64 * 1e6
    print('settling round %d: eager %.2f us/day' % (k, cur))
    if prev is not None and abs(cur - prev) < 0.01 * prev and k >= 3:
        break
    prev = cur

s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.stream(s):
        eng.run_day_array(chunk, 64, None)   # warm-up on the capture stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.graph(g, stream=s):
        eng.run_day_array(chunk, 64, None)
    t_cap = time.perf_counter() - t0
except Exception as e:   # a launch the runtime cannot capture
    print('capture failed: %r' % (e,))
    raise SystemExit(0)
print('capture + instantiate of the 64-day graph: %.2f ms = %.1f us per captured day' % (t_cap * 1e3, t_cap / 64 * 1e6))
timed(g.replay, 2)
E, Gr = [], []
for k in range(4):
    h, w = timed(eager, reps)
    E.append(w / reps / 64 * 1e6)
    he = h / reps / 64 * 1e6
    h, w = timed(g.replay, reps)
    Gr.append(w / reps / 64 * 1e6)
    print('round %d: eager %.2f us/day (host issue %.2f)   graph %.2f us/day (host issue %.2f)' % (k, E[-1], he, Gr[-1], h / reps / 64 * 1e6))
print('mean of the alternating rounds: eager %.2f, graph %.2f us/day GPU wall' % (sum(E) / 4, sum(Gr) / 4))
