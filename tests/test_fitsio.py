# -*- coding: utf-8 -*-
"""CPU tests of the minimal FITS writer / reader behind save_lightcurve (BasePhotometry.py:1417-1730)."""
import numpy as np
from photometry_amd import fitsio


def test_round_trip_and_checksums(tmp_path):
	rng = np.random.default_rng(1)
	T = 77
	cols = [
		{'name': 'TIME', 'format': 'D', 'array': np.linspace(1325, 1352, T), 'unit': 'BJD - 2457000, days', 'disp': 'D14.7'},
		{'name': 'TIMECORR', 'format': 'E', 'array': rng.random(T).astype('float32')},
		{'name': 'CADENCENO', 'format': 'J', 'array': np.arange(T) + 4697},
		{'name': 'FLUX_RAW', 'format': 'D', 'array': np.where(rng.random(T) < 0.1, np.nan, rng.normal(1e4, 10, T))},
	]
	prim = [fitsio.card('ORIGIN', 'TASOC/Aarhus', 'institution responsible for creating this file'), fitsio.card('TICID', 260795451, 'id'),
		fitsio.card('PMRA', fitsio.Undefined(), 'undefined value'), fitsio.card('TESSMAG', 10.25), fitsio.card('OBJECT', "it's", 'quote'),
		fitsio.card('CRSPOC', False), fitsio.card('EQUINOX', 2000.0), fitsio.card('TINY', 1.5e-300), fitsio.card('NANVAL', float('nan'))]
	img = rng.random((11, 13))
	aper = (rng.random((11, 13)) > 0.5).astype('int32') * 11
	for name in ('t.fits', 't.fits.gz'):
		path = str(tmp_path / name)
		fitsio.write(path, [fitsio.primary_hdu(prim), fitsio.bintable_hdu('LIGHTCURVE', cols, [fitsio.card('INHERIT', True)]),
			fitsio.image_hdu('SUMIMAGE', img), fitsio.image_hdu('APERTURE', aper), fitsio.image_hdu('FLAGS', aper.astype('uint8'))])
		hdus = fitsio.read(path)
		assert [h.get('EXTNAME') for h, _ in hdus] == [None, 'LIGHTCURVE', 'SUMIMAGE', 'APERTURE', 'FLAGS']
		assert all(h['__checksum_ok__'] and h['__datasum_ok__'] for h, _ in hdus)
		h0 = hdus[0][0]
		assert h0['SIMPLE'] is True and h0['TICID'] == 260795451 and h0['PMRA'] is None and h0['OBJECT'] == "it's"
		assert h0['CRSPOC'] is False and h0['TESSMAG'] == 10.25 and h0['TINY'] == 1.5e-300 and h0['NANVAL'] is None
		tab = hdus[1][1]
		for c in cols:
			np.testing.assert_array_equal(tab[c['name']], c['array'])
		assert hdus[1][0]['TUNIT1'] == 'BJD - 2457000, days' and hdus[1][0]['NAXIS2'] == T and hdus[1][0]['NAXIS1'] == 8 + 4 + 4 + 8
		np.testing.assert_array_equal(hdus[2][1], img)
		np.testing.assert_array_equal(hdus[3][1], aper)
		assert hdus[4][1].dtype == np.uint8


def test_file_structure(tmp_path):
	"""Blocks of 2880 bytes, 80-character cards, END card, big-endian data, NAXIS1 = fastest axis."""
	path = str(tmp_path / 's.fits')
	a = np.arange(6, dtype='float64').reshape(2, 3)
	fitsio.write(path, [fitsio.primary_hdu([]), fitsio.image_hdu('X', a)])
	raw = open(path, 'rb').read()
	assert len(raw) % 2880 == 0 and raw[:30] == b'SIMPLE  =                    T'
	hdr2 = raw[2880:5760].decode('ascii')
	cards = [hdr2[i:i+80] for i in range(0, 2880, 80)]
	assert cards[0].startswith("XTENSION= 'IMAGE   '") and any(c.startswith('END') for c in cards)
	d = dict((c[:8].strip(), c[10:30].strip()) for c in cards if c[8:10] == '= ')
	assert d['BITPIX'] == '-64' and d['NAXIS1'] == '3' and d['NAXIS2'] == '2'
	assert np.frombuffer(raw[5760:5760 + 48], dtype='>f8').tolist() == [0, 1, 2, 3, 4, 5]
