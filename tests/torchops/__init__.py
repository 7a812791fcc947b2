"""torch-op restatements used only by tests (see operators.py, sampler.py, wpe.py)."""
