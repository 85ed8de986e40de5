#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""
Import the *reference* package (tasoc/photometry, mounted read-only at
``/root/reference``) in the dev container, where most of its third-party
dependencies are absent.  ONLY used by ``make_golden.py`` to generate the
golden vectors committed under ``tests/golden/`` -- it never runs on the GPU box
and nothing in the product imports it.

Recipe:

1. numpy-2 shims (``np.NaN``, ``np.str``, ``np.int``): the reference is pinned to
   numpy 1.21.6 (``requirements.txt:8``) and uses e.g. ``np.NaN``
   (``AperturePhotometry/photometry.py:183``), ``np.str``
   (``linpsf_photometry.py:105``).
2. a ``sys.meta_path`` finder returning ``MagicMock`` package modules for the
   missing top-level packages.
3. the few names that must be real classes/functions are patched
   (warning categories, base classes, the bottleneck reductions -> numpy).
"""

import sys
import types
import importlib.abc
import importlib.machinery
import warnings
from unittest import mock
import numpy as np

REFERENCE_PATH = '/root/reference'

STUBBED = {'astropy', 'h5py', 'erfa', 'bottleneck', 'photutils', 'statsmodels', 'skimage',
	'cv2', 'halophot', 'spiceypy', 'mpi4py', 'psycopg2', 'tqdm', 'requests', 'matplotlib_inline'}


class _StubLoader(importlib.abc.Loader):
	def create_module(self, spec):
		m = mock.MagicMock(name=spec.name)
		m.__name__ = spec.name
		m.__path__ = []
		m.__spec__ = spec
		m.__loader__ = self
		return m

	def exec_module(self, module):
		pass


class _StubFinder(importlib.abc.MetaPathFinder):
	def find_spec(self, fullname, path, target=None):
		top = fullname.split('.')[0]
		if top in STUBBED:
			try:
				# Only stub what is really missing:
				if top not in _really_missing:
					return None
			except NameError:
				pass
			return importlib.machinery.ModuleSpec(fullname, _StubLoader(), is_package=True)
		return None


def _probe_missing():
	missing = set()
	for name in STUBBED:
		try:
			__import__(name)
		except Exception:
			missing.add(name)
	return missing


_really_missing = _probe_missing()


def _bn_move_median(x, window, min_count=None):
	"""numpy stand-in for bottleneck.move_median (trailing window, NaN-aware)."""
	x = np.asarray(x, dtype='float64')
	if min_count is None:
		min_count = window
	y = np.full_like(x, np.nan)
	for i in range(len(x)):
		w = x[max(0, i - window + 1):i + 1]
		w = w[~np.isnan(w)]
		if len(w) >= min_count:
			y[i] = np.median(w)
	return y


def import_reference():
	"""Returns the imported reference ``photometry`` package."""
	if 'photometry' in sys.modules and getattr(sys.modules['photometry'], '__file__', '').startswith(REFERENCE_PATH):
		return sys.modules['photometry']

	# 1. numpy-2 shims
	if not hasattr(np, 'NaN'):
		np.NaN = np.nan
	if not hasattr(np, 'str'):
		np.str = str
	if not hasattr(np, 'int'):
		np.int = int
	if not hasattr(np, 'RankWarning'):
		np.RankWarning = np.exceptions.RankWarning

	# 2. stub finder
	sys.meta_path.insert(0, _StubFinder())

	# 3. patch names that must be real
	import erfa
	erfa.ErfaWarning = type('ErfaWarning', (Warning,), {})
	import astropy.wcs
	astropy.wcs.FITSFixedWarning = type('FITSFixedWarning', (Warning,), {})
	import astropy.nddata
	astropy.nddata.CCDData = type('CCDData', (object,), {})
	import photutils
	photutils.BackgroundBase = type('BackgroundBase', (object,), {})

	import bottleneck
	with warnings.catch_warnings():
		warnings.simplefilter('ignore')
		bottleneck.allnan = lambda x: bool(np.all(np.isnan(x)))
		bottleneck.anynan = lambda x: bool(np.any(np.isnan(x)))
		bottleneck.nansum = np.nansum
		bottleneck.nanmedian = np.nanmedian
		bottleneck.nanmean = np.nanmean
		bottleneck.nanvar = np.nanvar
		bottleneck.nanstd = np.nanstd
		bottleneck.nanargmin = np.nanargmin
		bottleneck.nanargmax = np.nanargmax
		bottleneck.nanmin = np.nanmin
		bottleneck.nanmax = np.nanmax
		bottleneck.move_median = _bn_move_median

		def _replace(a, old, new):
			if np.isnan(old):
				a[np.isnan(a)] = new
			else:
				a[a == old] = new
		bottleneck.replace = _replace

	sys.path.insert(0, REFERENCE_PATH)
	with warnings.catch_warnings():
		warnings.simplefilter('ignore')
		import photometry
	assert photometry.__file__.startswith(REFERENCE_PATH)
	return photometry


class FakeCatalog(object):
	"""
	Tiny stand-in for the ``astropy.table.Table`` catalog that the reference plugins use
	(``BasePhotometry.catalog``, BasePhotometry.py:1094-1181): supports ``cat['col']``,
	``cat['col'] = values``, iteration over row-dicts, boolean / list row indexing, ``len``.
	"""
	def __init__(self, **cols):
		self.cols = {k: np.asarray(v) for k, v in cols.items()}

	def __len__(self):
		return len(next(iter(self.cols.values())))

	def __bool__(self):
		return True

	def __getitem__(self, key):
		if isinstance(key, str):
			return self.cols[key]
		if isinstance(key, (int, np.integer)):
			return {k: v[key] for k, v in self.cols.items()}
		if isinstance(key, slice):
			return FakeCatalog(**{k: v[key] for k, v in self.cols.items()})
		key = np.asarray(key)
		if key.size == 0:
			key = key.astype('int64')
		return FakeCatalog(**{k: v[key] for k, v in self.cols.items()})

	def sort(self, key):
		"""astropy ``Table.sort(key)``: in place, ascending (psf_photometry.py:126)."""
		order = np.argsort(self.cols[key])
		self.cols = {k: v[order] for k, v in self.cols.items()}

	def __setitem__(self, key, value):
		self.cols[key] = np.asarray(value)

	def __iter__(self):
		for i in range(len(self)):
			yield {k: v[i] for k, v in self.cols.items()}

	def copy(self):
		return FakeCatalog(**{k: v.copy() for k, v in self.cols.items()})
