# convenience targets (the driver calls __graft_entry__.build(), pytest and bench.py directly)
PY ?= python

build:
	$(PY) __graft_entry__.py

test-cpu: build
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build
	$(PY) -m pytest tests -q -m gpu

smoke: build
	$(PY) __graft_entry__.py --smoke

bench: build
	$(PY) bench.py

bench-build: build
	$(PY) bench.py --workload build --steps 5000 --warmup 20

golden:
	$(PY) tools/gen_golden.py --all

profiles:
	bash tools/collect_profiles.sh r01 && $(PY) tools/publish_profiles.py r01

c-example: build
	gcc -O2 -Iinclude examples/c_caller.c -Lavlmaps_amd/lib -lavlmaps_hip -Wl,-rpath,$(CURDIR)/avlmaps_amd/lib -lm -o examples/c_caller

.PHONY: build test-cpu test-gpu smoke bench bench-build golden profiles c-example
