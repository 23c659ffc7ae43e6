"""CPU oracle for the Tiny-SD hot path (UNet denoise loop + VAE decoder/encoder).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker, never as the thing measured or shipped.

PARITY PIN STATUS: **parity unpinned by the reference.**  The reference
(lrmantovani10/Stable-Diffusion.mojo) ships no tests, no golden vectors and cannot be run
(Mojo toolchain absent; the literal code reads uninitialised memory — SURVEY.md Appendix A).
This restatement follows the reference's architecture, constants and scalar formulas
(file:line cited per function) and takes tensor semantics from the PyTorch op each struct is
named after wherever the Mojo body is undefined.  It is pinned
instead against ``torch.nn.functional`` for the ops whose formula equals PyTorch's
(tests/test_oracle_pins.py), against hand-derived values for the formulas that differ
(GroupNorm ``sigma+eps``, tanh-GELU), and against committed fixtures in ``tests/golden``.

``oracle/cref/cref.c`` (+ ``oracle/cref.py``) states the same ops a second time in C, as the reference's own loop nests
with OpenMP where the reference calls ``parallelize`` (``cpu_ref``); the two statements are
held against each other in tests/test_cref_cpu.py, and the C one is what bench.py's ``cpu_baseline`` times.
"""
from . import rng, ops, spec, models, sampler  # noqa: F401
