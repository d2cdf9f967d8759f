"""
SURVEY.md section 8 f-1: the Fermi-level searches and the integration-limit fitting of gaunegf_amd.density
against the reference's own function bodies (gauNEGF/density.py:821-1515), probe for probe.

tests/golden/make_golden.py executed the reference functions (AST-extracted, unmodified) on the analytic spies of
tests/fermi_probe_harness.py and committed every probe they made -- (mu, grid size, temperature) of each density
call, each DOS sample, each g.setF -- plus the returned scalars (ref_fermi_search.npz).  Here the product's
functions run on the same spies: the probe sequences must be IDENTICAL (bit-exact chemical potentials), so the
search logic walks exactly the reference's path whatever engine serves the integrals.
"""
import contextlib
import io
import os

import numpy as np
import pytest

import fermi_probe_harness as H
import gaunegf_amd.density as D

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_fermi_search.npz"))


@pytest.mark.parametrize("tag,name,call", H.CASES, ids=[c[0] for c in H.CASES])
def test_search_probe_sequence_equals_reference(monkeypatch, tag, name, call):
    rec = []
    for k, fn in H.make_spies(rec).items():
        monkeypatch.setattr(D, k, fn)
    with contextlib.redirect_stdout(io.StringIO()):
        ret = call(getattr(D, name), H.ProbeG(rec))
    got = H.pack(rec)
    ref = GOLD[f"{tag}_probes"]
    # the polynomial-fit search goes through scipy.optimize.least_squares: same probes to rounding; all others bit-exact
    if name == "calcFermiPolyFit":
        assert got.shape == ref.shape and np.array_equal(got[:, 0], ref[:, 0])
        assert np.max(np.abs(got - ref)) < 1e-9
        assert np.allclose(H.scalars(ret), GOLD[f"{tag}_ret"], rtol=0, atol=1e-9, equal_nan=True)
    else:
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert np.array_equal(got, ref), np.argwhere(got != ref)[:5]
        assert np.array_equal(H.scalars(ret), GOLD[f"{tag}_ret"], equal_nan=True)
