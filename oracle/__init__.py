"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

A plain numpy (float64 / complex128) CPU restatement of the reference's NEGF
energy-grid hot path (wliverno/GauNEGF), used as the parity checker for the HIP
engine in ``gaunegf_amd``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``gaunegf_amd``) never imports, links or executes anything from here.

Pinning status (see DESIGN.md section "Oracle"):
  * grid / weight / index bookkeeping ........ PINNED bit-exact against vectors
    produced by executing the reference's own numpy-only functions here
    (tests/golden/make_golden.py; fixtures under tests/golden/).
  * G(E), G<, transmission, DOS .............. PINNED against the reference's own
    numpy restatements (tests/test_computation_consistency.py:80-118,
    tests/jax_optimization_suite.py:165-194) executed here on the reference's
    seeded generators.
  * 1-D chain fixed point .................... PINNED (zero-start variant) against
    the reference's tests/test_surface_green_jit.py:47-68 manual_iteration; the
    production variant (start = inv(A), surfG1D.py:287) differs only in g_init.
  * Bethe-lattice fixed point ................ parity UNPINNED by reference output
    (needs jax to run); restated from surfGBethe.py:958-1108 as text and pinned
    only by physical invariants.
"""
from .negf_oracle import *  # noqa: F401,F403
