"""CPU oracle for the NeurIPS-CellSeg 3-class U-Net hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product path
(``neurips-cellseg_amd/``) never imports this package and fails loudly when
the HIP library is missing.

Parity status (see DESIGN.md "Oracle pinning"):

* post-process (rows P1-P4), label contract, input normalisation and the F1
  harness are PINNED: fixtures under ``tests/golden/`` were generated in the
  build container by running real scikit-image 0.18.3 and by importing the
  reference's own ``baseline/compute_metric.py`` / ``data/pre_process_3class.py``
  (generator scripts: ``tests/golden/make_*.py``).
* U-Net topology (row M), sliding-window tiler (row T) and DiceCE (row L)
  live in MONAI, which is neither vendored in the reference nor installable
  here: those restatements are "parity unpinned" against MONAI itself and are
  anchored on the reference's call sites plus structural cross-checks
  (parameter count 1 626 072, 63 state-dict tensors, MONAI key names).
"""
