# Convenience targets; everything is plain python underneath.
PY ?= python

.PHONY: build test test-gpu bench golden clean

build:            ## HIP engine (gfx950) + the two C oracles (test infrastructure)
	$(PY) -c "import __graft_entry__ as g; g.build()"

test: build       ## CPU suite: oracles vs goldens, host logic, ABI, gloo sharding (about 2 min)
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu: build   ## on an MI355X: HIP engine == oracle B bit-exact, statistical tier, scale properties
	$(PY) -m pytest tests -x -q -m gpu

bench: build      ## one JSON line: agent-days/s, roofline, cpu_baseline, large, ensemble
	$(PY) bench.py

golden:           ## regenerate tests/golden from the real reference (needs /root/reference; build container only)
	cd tests/golden && $(PY) make_golden.py

clean:
	rm -f reina_model_amd/csrc/libreina_hip.so oracle/libreina_seq.so oracle/libreina_par.so
