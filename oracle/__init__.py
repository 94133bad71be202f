"""CPU oracle for the 3D-SSD hot path — TEST INFRASTRUCTURE ONLY.

This package is a plain-torch / numpy CPU restatement of what the reference
(Medical-Image-Analysis-Laboratory/MSLesions3D, ``lesions3d/ssd3d.py`` + ``mobilenet.py`` +
``utils.py:42-396``) computes on the hot path.  Every function cites the reference file:line it
follows.  It exists so that the HIP kernels in ``mslesions3d_amd/csrc`` can be checked against
something that is itself pinned to the reference:

* pinned by ``tests/golden/*.npz`` — vectors minted in the build container by importing the
  reference's own modules (``tests/golden/make_golden.py``; third-party Lightning/MONAI/wandb,
  none of which touch the arithmetic, replaced by inert stand-ins) — see
  ``tests/test_oracle_golden.py``.  The reference ships no tests or golden vectors of its own
  (SURVEY.md §4), so these minted vectors are the only pin there is.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``mslesions3d_amd``) never imports it and has no CPU fallback.
"""
