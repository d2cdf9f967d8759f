"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

A plain numpy (float64 / complex128) CPU restatement of the reference's NEGF
energy-grid hot path (wliverno/GauNEGF), used as the parity checker for the HIP
engine in ``gaunegf_amd``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``gaunegf_amd``) never imports, links or executes anything from here.

Pinning status (fixtures under tests/golden/, all produced by tests/golden/make_golden.py
EXECUTING the reference's own numpy-only functions in this container, bit-reproducible):
  * grid / weight / index bookkeeping, Fermi function, ANT points, formSigma, SigmaCalculator, the (Elist, weights,
    ind) arrays the density / current front-ends hand to GrInt / GrLessInt / calculate_transmission ... PINNED bit-exact.
  * G(E), G<, transmission (restricted), DOS ... PINNED to the reference's numpy restatements
    (tests/test_computation_consistency.py:80-118, tests/jax_optimization_suite.py:165-194) on its seeded generators.
  * 1-D chain fixed point ... PINNED for BOTH starts: manual_iteration (tests/test_surface_green_jit.py:47-68) from zeros
    and from g0 = solve(A, I), the production start of surfG1D.py:287 (sweep counts, flag, iterate to 1e-10);
    sigma / sigmaTot to compute_sigma_for_energy (tests/benchmark_sigma_parallelization.py:63-119).
  * Bethe lattice ... PINNED to the reference's numpy twin gauNEGF/surfG3D.py: parameter parser, neighbour generator,
    Slater-Koster blocks, the SURFACE fixed point (surfGAt.sigma), the 13-site cluster assembly and the contact
    assembly surfG3.sigma / sigmaTot (neighbour lists inside 0..8, spin r / u / g).
  * Fermi searches and integration-limit fitting (density.py:821-1515) ... PINNED probe for probe (ref_fermi_search.npz).
  * closed-form density() / bisectFermi() (density.py:276-382) ... PINNED.
  * UNPINNED (no runnable numpy counterpart in the reference; restated from the text, checked device-vs-oracle only):
    the Bethe BULK loop surfGBAt.sigmaK (surfGBethe.py:958-1030; the twin's bulk loop is a different, Jacobi-type
    iteration), the spin-block transmission kernel (transport.py:159-181; its uu / dd blocks are anchored indirectly
    on block-diagonal systems), and the a16 corner cases (attached directions outside 0..8, the Xi Sigma Xi branch).
"""
from .negf_oracle import *  # noqa: F401,F403
