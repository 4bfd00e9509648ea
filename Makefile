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
	$(PY) bench.py --workload build --steps 10000 --warmup 20

bench-config5: build
	$(PY) bench.py --feat-dim 1536 --queries 128

golden:
	$(PY) tools/gen_golden.py --all

TAG ?= r04
profiles:
	bash tools/collect_profiles.sh $(TAG) && $(PY) tools/publish_profiles.py $(TAG)

c-example: build
	gcc -O2 -Iinclude examples/c_caller.c -Lavlmaps_amd/lib -lavlmaps_hip -Wl,-rpath,$(CURDIR)/avlmaps_amd/lib -lm -o examples/c_caller

.PHONY: build test-cpu test-gpu smoke bench bench-build bench-config5 golden profiles c-example
