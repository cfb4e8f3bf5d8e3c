"""CPU calibration (BASELINE.md section 3 / SURVEY 8d): the REAL cythonsim (both build modes) and oracle A -- its bit-exact C
restatement, the CPU baseline that travels to the GPU box -- timed on ONE core of the build container over the same days of the
HUS default scenario (seed 0).  Needs /root/reference (build container only); prints one JSON object that bench.py carries
as the recorded constant cpu_baseline.calibration.
    python tools/cpu_calibration.py [cythonsim|cythonsim_noexcept|oracle_a]   (no argument: all three, each in a process of its own)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HUS = 1685983


def time_days(ctx_iter, spans=(100, 365)):
    out, done, t0 = {}, 0, time.perf_counter()
    for n in spans:
        while done < n:
            ctx_iter()
            done += 1
        out[str(n)] = time.perf_counter() - t0
    return out


def run(what):
    if what == 'oracle_a':
        sys.path.insert(0, ROOT)
        import copy
        from oracle import seq_oracle
        from reina_model_amd import datasets
        from reina_model_amd.variables import VARIABLE_DEFAULTS
        ctx = seq_oracle.make_context(copy.deepcopy(VARIABLE_DEFAULTS), datasets.get_population_for_area(), 0)
        return time_days(ctx.iterate)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden', '_harness'))
    if what == 'cythonsim_noexcept':
        # the build the reference pins (cython 3.0a6: cdef nogil functions are noexcept): same results, no GIL round trip per call
        os.environ['REINA_PYXBLD'] = '/tmp/reina_pyxbld_noexcept'
        from Cython.Compiler import Options
        Options.get_directive_defaults()['legacy_implicit_noexcept'] = True
    import ref_harness as rh
    ctx = rh.make_context(0)
    return time_days(ctx.iterate)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        print(json.dumps(run(sys.argv[1])))
        sys.exit(0)
    res = {}
    for what in ('oracle_a', 'cythonsim', 'cythonsim_noexcept'):
        p = subprocess.run(['taskset', '-c', '2', sys.executable, os.path.abspath(__file__), what], capture_output=True, text=True)
        try:
            t = json.loads(p.stdout.strip().splitlines()[-1])
            res[what] = {k: dict(seconds=round(v, 2), agent_days_per_s=round(HUS * int(k) / v, 1)) for k, v in t.items()}
        except Exception:
            res[what] = dict(error=(p.stderr or p.stdout)[-400:])
    cpu = next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), '?')
    print(json.dumps(dict(cpu_model=cpu, cores=1, workload='HUS 1 685 983 agents, default scenario, seed 0, the first 100 / 365 days', results=res), indent=1))
