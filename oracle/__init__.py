"""CPU oracle for the HPFG hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32) restatement of the arithmetic the
reference performs on its training hot path (U-Net forward/backward, CE/Dice/
MSE losses, pseudo-labels, EMA, SGD and the host-side scalar laws).  It is a
checker: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  Nothing under ``hpfg_amd/`` imports it and
the product path never falls back to it.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, imported file-by-file in the build container by
``oracle/make_golden.py``; the resulting vectors are committed under
``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``.
"""
