"""Batched (leading utterance axis), device-capable torch-op forms of the restatements in ``oracle/`` -- operators, loss, sampler hooks, WPE -- used only by the
tests (host logic on the CPU, autograd cross-checks of the HIP operator on the GPU).  Test infrastructure like the rest of ``oracle/``; they share the
oracle's helper functions (hilbert, minimum phase, linear interpolation)."""
