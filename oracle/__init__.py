"""CPU oracle for the MODE disparity-stage hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU and in plain PyTorch tensor ops, the algorithm of the
reference's ``ModeDisparity`` path (reference files cited per function as file:line,
relative to the upstream repo root).  It exists to *check* the HIP product path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import it;
  * nothing under ``mode-2022_amd/`` imports it, and the product never falls back to it.

Pin status: the reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md section 4), and its native op cannot be compiled here (needs nvcc + THC
headers that no longer exist).  The Python layers of the reference *are* importable in
the development container; ``tests/golden/make_golden.py`` runs them (with the native
op bound to ``oracle.sphere_conv_ref``) and commits the resulting vectors, so every
layer except the native sphere_conv op is pinned against the reference itself.  For the
native op (a7/a8 in SURVEY.md section 8) the oracle is a line-by-line restatement of the CUDA
kernels, cross-checked against an independent ``grid_sample`` formulation:
**parity unpinned by the reference's own tests** for that op.
"""
