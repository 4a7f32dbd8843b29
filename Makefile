# Convenience targets; the driver's entry points are __graft_entry__.py (build/smoke), bench.py and pytest.
.PHONY: build test test-gpu bench clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test:
	python -m pytest tests/ -x -q -m "not gpu"
test-gpu:
	python -m pytest tests/ -x -q -m gpu
bench:
	python bench.py
clean:
	$(MAKE) -C handposeestimation-with-3d-cnns_amd/csrc clean
	$(MAKE) -C oracle clean
