# -*- coding: utf-8 -*-
"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

A CPU (numpy / scipy) restatement of the tasoc/photometry hot path, written from the
reference's published behaviour.  Every function cites the reference ``file:line`` it
follows (paths relative to the reference repository root).

Who may import this package:  ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the *checker* / timed CPU
baseline.  Nothing under ``photometry_amd/`` imports it; the product path fails loudly
when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):

* A1 (TPF branch), A5b, A6, A7, P2, P3, P4, quality, utilities, the light-curve diagnostics
  (``diagnostics.py``), the stamp cutter (``cutout.py``, rank 2): pinned against golden
  vectors produced by *executing the reference's own code* in the dev container
  (``tests/golden/make_golden.py``; fixtures in ``tests/golden/*.npz``).
* A2-A5 (K2P2): control flow pinned by executing the reference's own
  ``k2p2FixFromSum`` with stand-ins for the four third-party functions that cannot be
  installed here (statsmodels KDE / bandwidth, scikit-image peak_local_max / watershed);
  DBSCAN, Powell, trim1 and the ndimage filters are pinned against the real scikit-learn /
  scipy installed here.  The statsmodels / scikit-image restatements themselves are
  written from the published algorithms of the pinned versions (statsmodels 0.13.2,
  scikit-image 0.19.2) and are **parity unpinned** against those packages.
* B* (stamp-level background) has no reference symbol; it is build-defined and pinned
  only by the reference's constant-image known answer (tests/test_background.py:36-54).
  **parity unpinned** against photutils/astropy.
"""

__all__ = ['quality', 'utilities', 'sumimage', 'aperture', 'kde', 'powell', 'k2p2',
	'backgrounds', 'psf', 'linpsf', 'diagnostics', 'cutout']
