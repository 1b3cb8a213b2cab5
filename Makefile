# `make demo` mirrors the reference's target (Makefile:23-25) with the headless, data-set-free pipeline.
.PHONY: build demo test test-gpu bench
build:
	python -m structure_from_motion_amd.build
demo: build
	python main.py
test:
	python -m pytest tests -q -m "not gpu"
test-gpu:
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
