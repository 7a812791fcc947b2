"""CPU oracle for the reverse-diffusion dereverberation sampler path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``buddy_amd/`` imports this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the checker /
reported baseline, never as the product path.

What it is: a plain PyTorch fp32 (CPU) *restatement* of the reference algorithm for the hot path
(SURVEY.md section 8(a)): NCSN++ score network incl. STFT/iSTFT (``ncsnpp_ref``), EDM preconditioning
(``edm_ref``), reverb operators + DSP utils + reconstruction loss (``operators_ref``), Euler-Heun and
Euler-Heun-DPS samplers (``sampler_ref``).  Every function cites the reference file:line it follows.

Parity pin: the reference is pure Python/PyTorch and imports in the build container, so the oracle is
pinned against outputs of the reference itself (fixtures under ``tests/golden/`` produced by
``tests/golden/make_golden.py``, which imports ``/root/reference`` read-only).  The reference has no
tests/golden vectors of its own.  Two third-party pieces are absent from
``/root/reference`` and restated from their API semantics -- parity for these two is UNPINNED:
``torchcde`` (unpinned in reference ``requirements.txt:17``; linear interpolation in
``operators_ref.linear_interp``) and ``nara_wpe`` (WPE warm start, ``wpe_ref``: numpy complex128, written from the package's
published algorithm and STFT conventions independently of the product's kernels, so the HIP warm start and ``wpe_scaled`` runs
have an oracle counterpart; against nara_wpe itself parity stays unpinned -- see DESIGN.md).
"""
