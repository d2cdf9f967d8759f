"""
Default parameters of the engine; names and values follow gauNEGF/config.py:8-33 so
that host code written against the reference finds the same knobs.  (The values in
the reference's code differ from its README -- ETA is 1e-6 in config.py:9 -- the
code values are the ones mirrored here.)
"""
# Physical parameters
TEMPERATURE = 0.0
ETA = 1e-6
ENERGY_STEP = 0.001

# Contact tolerances
FERMI_CALCULATION_TOL = 1e-3
FERMI_SEARCH_CYCLES = 10
SURFACE_GREEN_CONVERGENCE = 1e-5
SURFACE_RELAXATION_FACTOR = 0.1

# Integration parameters
ADAPTIVE_INTEGRATION_TOL = 1e-4
N_KT = 10
ENERGY_MIN = -1e6
MAX_CYCLES = 1000
MAX_GRID_POINTS = 1000

# SCF parameters (kept for API completeness; the SCF driver is out of scope)
SCF_DAMPING = 0.02
SCF_CONVERGENCE_TOL = 1e-3
SCF_MAX_CYCLES = 100
PULAY_MIXING_SIZE = 4

LOG_LEVEL = 'DEBUG'
LOG_PERFORMANCE = True

# hard-coded in the reference's kernels
SURFACE_GREEN_MAX_ITER = 2000     # surfG1D.py:265
BETHE_MAX_ITER = 1000             # surfGBethe.py:998,1077
BETHE_MIX = 0.5                   # surfGBethe.py:958
