"""CPU oracle for the EEG hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker.  The product path (``isd_amd``) never
imports this package and raises if the HIP library is missing.

Parity status (see DESIGN.md "Oracle"):

* ``oracle.cnn``  -- restates ``src/fast/models/fast.py`` (Conv4Layers, Head,
  EEGNet_Encoder, forward_head, train_head) and ``src/fast/train/trainer.py``
  (CE loss, argmax predict, cosine schedule).  PINNED by golden vectors
  captured from the importable reference (``tests/golden/make_golden.py``).
* ``oracle.dsp``  -- restates the scipy.signal (1.15.3, reference pins
  ``scipy>=1.10``) algorithms the reference calls at
  ``scripts/global_shap_analysis.py:132-156`` (stft + band aggregation) plus
  the build-defined Butterworth filterbank (spec S).  PINNED by
  golden vectors produced by scipy itself.
* MNE ``filter_data`` (``notebooks/svm_baseline.ipynb:238``): source absent,
  not installed -> PARITY UNPINNED for that one call; nothing here claims it.
"""
