# -*- coding: utf-8 -*-
"""
Minimal FITS writer / reader for the TASOC light-curve files (no astropy in this image).

Implements exactly what ``BasePhotometry.save_lightcurve`` needs (photometry/BasePhotometry.py:1417-1730): a primary
HDU without data, one binary table (``TFORM`` D / E / J scalar columns), image extensions (float64 / int32 / uint8),
80-character header cards in 2880-byte blocks, big-endian data, the ``DATASUM`` / ``CHECKSUM`` keywords of the FITS
checksum convention (the reference writes with ``checksum=True``, :1720), optional gzip.  The reader understands the
same subset and is used by the tests for round trips.
"""

import gzip
import io
import numpy as np

BLOCK = 2880
_FORMATS = {'D': '>f8', 'E': '>f4', 'J': '>i4', 'K': '>i8', 'B': 'u1', 'I': '>i2'}
_BITPIX = {'uint8': 8, 'int16': 16, 'int32': 32, 'int64': 64, 'float32': -32, 'float64': -64}


class Undefined(object):
	"""A keyword with an undefined value (astropy ``fits.card.Undefined``, BasePhotometry.py:1479-1491)."""


def _fmt_value(v):
	if isinstance(v, Undefined) or v is None:
		return ' ' * 20
	if isinstance(v, (bool, np.bool_)):
		return ('T' if v else 'F').rjust(20)
	if isinstance(v, (int, np.integer)):
		return str(int(v)).rjust(20)
	if isinstance(v, (float, np.floating)):
		if not np.isfinite(v):
			return ' ' * 20 # FITS has no NaN header values: undefined
		s = repr(float(v)).upper()
		if 'E' not in s and '.' not in s:
			s += '.0'
		return s.rjust(20)
	s = str(v).replace("'", "''")
	return ("'" + s.ljust(8) + "'").ljust(20)


def card(key, value=None, comment=None):
	"""One 80-character header card."""
	if key in ('COMMENT', 'HISTORY', ''):
		return (key.ljust(8) + str(value or ''))[:80].ljust(80)
	c = key.upper().ljust(8)[:8] + '= ' + _fmt_value(value)
	if comment:
		c += ' / ' + comment
	return c[:80].ljust(80)


def _header_bytes(cards):
	txt = ''.join(cards) + 'END'.ljust(80)
	txt += ' ' * ((-len(txt)) % BLOCK)
	return txt.encode('ascii')


def _pad(data, fill=b'\0'):
	return data + fill * ((-len(data)) % BLOCK)


def _sum32(data, start=0):
	"""1's complement sum of big-endian 32-bit words (FITS checksum convention)."""
	if len(data) % 4:
		data = data + b'\0' * (4 - len(data) % 4)
	words = np.frombuffer(data, dtype='>u4').astype(np.uint64)
	total = int(words.sum()) + int(start)
	while total >> 32:
		total = (total & 0xFFFFFFFF) + (total >> 32)
	return total


def _encode_checksum(value):
	"""ASCII encoding of the complement of a 32-bit checksum (16 characters, rotated right by one)."""
	value = (~value) & 0xFFFFFFFF
	exclude = [0x3a, 0x3b, 0x3c, 0x3d, 0x3e, 0x3f, 0x40, 0x5b, 0x5c, 0x5d, 0x5e, 0x5f, 0x60]
	asc = [0] * 16
	for i in range(4):
		byte = (value >> (24 - 8 * i)) & 0xFF
		quotient, remainder = byte // 4 + 0x30, byte % 4
		ch = [quotient + remainder, quotient, quotient, quotient]
		check = True
		while check:
			check = False
			for k in (0, 2):
				while ch[k] in exclude or ch[k + 1] in exclude:
					ch[k] += 1
					ch[k + 1] -= 1
					check = True
		for j in range(4):
			asc[4 * j + i] = ch[j]
	s = ''.join(chr(c) for c in asc)
	return s[-1] + s[:-1]


def _hdu_bytes(cards, data):
	"""Header + data with DATASUM / CHECKSUM filled in."""
	datasum = _sum32(data) if data else 0
	cards = list(cards) + [card('CHECKSUM', '0' * 16, 'HDU checksum'), card('DATASUM', str(datasum), 'data unit checksum')]
	hdr = _header_bytes(cards)
	total = _sum32(hdr, datasum)
	cards[-2] = card('CHECKSUM', _encode_checksum(total), 'HDU checksum')
	return _header_bytes(cards) + data


def primary_hdu(cards):
	base = [card('SIMPLE', True, 'conforms to FITS standard'), card('BITPIX', 8, 'array data type'),
		card('NAXIS', 0, 'number of array dimensions'), card('EXTEND', True)]
	return _hdu_bytes(base + list(cards), b'')


def image_hdu(name, array, cards=()):
	a = np.asarray(array)
	if a.dtype == bool:
		a = a.astype('uint8')
	bitpix = _BITPIX[a.dtype.name]
	base = [card('XTENSION', 'IMAGE', 'Image extension'), card('BITPIX', bitpix, 'array data type'),
		card('NAXIS', a.ndim, 'number of array dimensions')]
	for i, n in enumerate(reversed(a.shape)):
		base.append(card(f'NAXIS{i+1}', int(n)))
	base += [card('PCOUNT', 0, 'number of parameters'), card('GCOUNT', 1, 'number of groups')]
	base += list(cards) + [card('EXTNAME', name, 'extension name')]
	data = _pad(a.astype(a.dtype.newbyteorder('>')).tobytes())
	return _hdu_bytes(base, data)


def bintable_hdu(name, columns, cards=()):
	"""``columns``: list of dicts ``name, format (D/E/J/...), array`` and optional ``unit, disp, comments`` (dict of TTYPE/TFORM/... comments)."""
	nrows = len(columns[0]['array'])
	dt = np.dtype([(c['name'], _FORMATS[c['format']]) for c in columns])
	rec = np.zeros(nrows, dtype=dt)
	for c in columns:
		rec[c['name']] = np.asarray(c['array'])
	base = [card('XTENSION', 'BINTABLE', 'binary table extension'), card('BITPIX', 8, 'array data type'),
		card('NAXIS', 2, 'number of array dimensions'), card('NAXIS1', dt.itemsize, 'length of dimension 1'),
		card('NAXIS2', nrows, 'length of dimension 2'), card('PCOUNT', 0, 'number of group parameters'),
		card('GCOUNT', 1, 'number of groups'), card('TFIELDS', len(columns), 'number of table fields')]
	base += list(cards)
	for i, c in enumerate(columns, start=1):
		cm = c.get('comments', {})
		base.append(card(f'TTYPE{i}', c['name'], cm.get('TTYPE')))
		base.append(card(f'TFORM{i}', c['format'], cm.get('TFORM')))
		if c.get('unit'):
			base.append(card(f'TUNIT{i}', c['unit'], cm.get('TUNIT')))
		if c.get('disp'):
			base.append(card(f'TDISP{i}', c['disp'], cm.get('TDISP')))
	base.append(card('EXTNAME', name, 'extension name'))
	return _hdu_bytes(base, _pad(rec.tobytes()))


def write(path, hdus):
	"""``hdus``: list of byte strings from :func:`primary_hdu`, :func:`bintable_hdu`, :func:`image_hdu`."""
	blob = b''.join(hdus)
	if str(path).endswith('.gz'):
		with gzip.open(path, 'wb') as fh:
			fh.write(blob)
	else:
		with open(path, 'wb') as fh:
			fh.write(blob)


#--------------------------------------------------------------------------------------------------
def _parse_value(s):
	s = s.strip()
	if not s:
		return None
	if s.startswith("'"):
		end = s.rfind("'")
		return s[1:end].replace("''", "'").rstrip()
	if s in ('T', 'F'):
		return s == 'T'
	try:
		return int(s)
	except ValueError:
		return float(s.replace('D', 'E'))


def read(path):
	"""Returns a list of ``(header dict, data)``; data is None, an ndarray (image) or a dict of column arrays (table)."""
	opener = gzip.open if str(path).endswith('.gz') else open
	with opener(path, 'rb') as fh:
		blob = fh.read()
	out = []
	pos = 0
	while pos < len(blob):
		header = {}
		raw_cards = []
		done = False
		while not done:
			block = blob[pos:pos + BLOCK].decode('ascii')
			pos += BLOCK
			for i in range(0, BLOCK, 80):
				c = block[i:i + 80]
				raw_cards.append(c)
				key = c[:8].strip()
				if key == 'END':
					done = True
					break
				if c[8:10] == '= ':
					val = c[10:]
					if val.lstrip().startswith("'"):
						q = val.index("'")
						end = q + 1
						while True:
							end = val.index("'", end)
							if end + 1 < len(val) and val[end + 1] == "'":
								end += 2
								continue
							break
						header[key] = _parse_value(val[:end + 1])
					else:
						header[key] = _parse_value(val.split('/')[0])
		naxis = header.get('NAXIS', 0)
		shape = [header[f'NAXIS{i}'] for i in range(naxis, 0, -1)]
		nbytes = (abs(header.get('BITPIX', 8)) // 8) * int(np.prod(shape)) if naxis else 0
		data = None
		raw = blob[pos:pos + nbytes]
		if header.get('XTENSION') == 'BINTABLE':
			dt = np.dtype([(header[f'TTYPE{i}'], _FORMATS[header[f'TFORM{i}'].strip()]) for i in range(1, header['TFIELDS'] + 1)])
			rec = np.frombuffer(raw, dtype=dt, count=header['NAXIS2'])
			data = {n: rec[n].astype(rec[n].dtype.newbyteorder('=')) for n in dt.names}
		elif naxis:
			code = {8: 'u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}[header['BITPIX']]
			data = np.frombuffer(raw, dtype=code).reshape(shape)
			data = data.astype(data.dtype.newbyteorder('='))
		header['__datasum_ok__'] = (str(_sum32(_pad(raw))) == str(header.get('DATASUM', '0')).strip()) if 'DATASUM' in header else None
		hdr_bytes = ''.join(raw_cards).encode('ascii')
		hdr_bytes += b' ' * ((-len(hdr_bytes)) % BLOCK)
		header['__checksum_ok__'] = (_sum32(hdr_bytes, _sum32(_pad(raw)) if raw else 0) == 0xFFFFFFFF) if 'CHECKSUM' in header else None
		pos += len(_pad(raw)) if nbytes else 0
		out.append((header, data))
	return out


#--------------------------------------------------------------------------------------------------
# TDB -> UTC calendar strings for DATE-OBS / DATE-END (astropy ``Time(..., scale='tdb').utc.isot``, BasePhotometry.py:1607-1629)
#--------------------------------------------------------------------------------------------------
#: (first MJD of validity, TAI - UTC in seconds): the leap seconds since 1972 (IERS Bulletin C; none after 2017-01-01)
_LEAP_SECONDS = ((41317, 10), (41499, 11), (41683, 12), (42048, 13), (42413, 14), (42778, 15), (43144, 16), (43509, 17), (43874, 18),
	(44239, 19), (44786, 20), (45151, 21), (45516, 22), (46247, 23), (47161, 24), (47892, 25), (48257, 26), (48804, 27), (49169, 28),
	(49534, 29), (50083, 30), (50630, 31), (51179, 32), (53736, 33), (54832, 34), (56109, 35), (57204, 36), (57754, 37))


def _tai_minus_utc(mjd_utc):
	dat = 10
	for first, value in _LEAP_SECONDS:
		if mjd_utc >= first:
			dat = value
	return dat


def tdb_to_utc_isot(jd1, jd2=0.0):
	"""
	``'YYYY-MM-DDTHH:MM:SS.sss'`` (UTC) of the TDB Julian date ``jd1 + jd2``.

	TT = TDB - (0.001657 sin g + 0.000014 sin 2g) s with g the Earth's mean anomaly (the two leading terms of the series
	astropy / ERFA ``dtdb`` evaluate; the rest is below 30 microseconds, i.e. below the millisecond the string carries unless the
	instant falls within that distance of a rounding boundary), TAI = TT - 32.184 s, UTC = TAI - leap seconds.
	"""
	# split into integer day number and seconds of day, keeping the precision of the two-part input
	big, small = (jd1, jd2) if abs(jd1) >= abs(jd2) else (jd2, jd1)
	day = np.floor(big + 0.5) # JD of the preceding midnight + 0.5, as an MJD-like integer below
	frac = (big - (day - 0.5)) + small # days since that midnight (TDB)
	mjd_day = int(day - 0.5 - 2400000.5 + 0.5) # = day - 2400001
	g = np.deg2rad(357.53 + 0.98560028 * ((big - 2451545.0) + small))
	sec = frac * 86400.0 - (0.001657 * np.sin(g) + 0.000014 * np.sin(2 * g)) - 32.184 # TAI seconds of the TDB day
	# leap seconds: decide with the UTC day the instant falls into
	sec_utc = sec - _tai_minus_utc(mjd_day)
	while sec_utc < 0:
		mjd_day -= 1
		sec_utc = sec + 86400.0 * 1 - _tai_minus_utc(mjd_day)
		sec += 86400.0
	while sec_utc >= 86400.0:
		mjd_day += 1
		sec -= 86400.0
		sec_utc = sec - _tai_minus_utc(mjd_day)
	ms = int(np.floor(sec_utc * 1000.0 + 0.5))
	if ms >= 86400000:
		ms -= 86400000
		mjd_day += 1
	# civil date of the MJD (Fliegel & Van Flandern)
	jdn = mjd_day + 2400001
	a = jdn + 32044
	b = (4 * a + 3) // 146097
	c = a - 146097 * b // 4
	d = (4 * c + 3) // 1461
	e = c - 1461 * d // 4
	m = (5 * e + 2) // 153
	dd = e - (153 * m + 2) // 5 + 1
	mm = m + 3 - 12 * (m // 10)
	yy = 100 * b + d - 4800 + m // 10
	hh, rem = divmod(ms, 3600000)
	mi, rem = divmod(rem, 60000)
	ss, ms = divmod(rem, 1000)
	return f'{yy:04d}-{mm:02d}-{dd:02d}T{hh:02d}:{mi:02d}:{ss:02d}.{ms:03d}'
