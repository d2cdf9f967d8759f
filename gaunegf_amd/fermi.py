"""Contact Fermi-level search used by surfGBethe (re-exported from density.py, where the
reference keeps it: gauNEGF/density.py:969-1003)."""
from .density import getFermiContact, getFermi1DContact, calcFermi  # noqa: F401
