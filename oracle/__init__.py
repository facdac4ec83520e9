"""CPU oracle for the DisenLink hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference's
``model.py`` (Disentangle_layer.forward model.py:55-77, Disentangle.forward
model.py:105-114) and of the loss / AUC lines of ``main_disentangled.py``
(:195, :202-204, :217-219).

Parity status: PINNED.  ``tests/golden/case_*.npz`` were produced by importing the
reference's own ``model.py`` in the build container (``tests/golden/make_golden.py``);
``tests/test_oracle_golden.py`` checks every function here against those vectors, and
the AUC routine against ``sklearn.metrics.roc_auc_score`` vectors.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker.  Nothing under ``disenlink_amd/`` imports
it; the product path fails loudly when the HIP library is missing.
"""
