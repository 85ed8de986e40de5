# -*- coding: utf-8 -*-
"""CPU-side checks of the C-ABI library: it builds, loads, and exports every declared symbol."""
import os
import re
import ctypes
import pytest
import conftest

HEADER = os.path.join(conftest.ROOT, 'include', 'tessphot_hip.h')


@pytest.fixture(scope='session')
def built_lib():
	import __graft_entry__ as g
	g.build()
	from photometry_amd import _lib
	return _lib


def header_functions():
	src = open(HEADER).read()
	src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
	return sorted(set(re.findall(r'\b(tp_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol(built_lib):
	lib = ctypes.CDLL(built_lib.LIB_PATH)
	names = header_functions()
	assert len(names) >= 20
	for name in names:
		assert hasattr(lib, name), f"{name} declared in include/tessphot_hip.h but not exported"
	# and the ctypes table covers the header exactly
	assert sorted(built_lib.SIGNATURES.keys()) == names


def test_version_and_names(built_lib):
	lib = built_lib.load()
	assert lib.tp_version() >= 100
	n = lib.tp_kernel_count()
	names = [lib.tp_kernel_name(k).decode() for k in range(n)]
	assert 'tp_aperture_kernel' in names and 'tp_sumimage_kernel' in names
	assert lib.tp_kernel_name(n + 5) == b''


def test_missing_library_fails_loudly(monkeypatch):
	from photometry_amd import _lib
	monkeypatch.setattr(_lib, '_lib', None)
	monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libtessphot_hip.so')
	with pytest.raises(_lib.TessphotLibraryError):
		_lib.load()


def test_no_gpu_is_an_error_not_a_fallback(built_lib):
	"""Without a GPU, creating a context must raise (there is no CPU fallback)."""
	lib = built_lib.load()
	n = ctypes.c_int(0)
	lib.tp_device_count(ctypes.byref(n))
	if n.value > 0:
		pytest.skip("a GPU is visible")
	from photometry_amd.device import Context
	from photometry_amd._lib import TessphotError
	with pytest.raises(TessphotError):
		Context(0)


def test_ctypes_arity_matches_the_header(built_lib):
	"""Every ctypes prototype has as many arguments as the declaration in include/tessphot_hip.h (catches signature drift
	between the header, the library and the Python layer for the long argument lists)."""
	src = open(HEADER).read()
	src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
	for name, (restype, argtypes) in built_lib.SIGNATURES.items():
		m = re.search(r'\b' + name + r'\s*\((.*?)\)\s*;', src, flags=re.S)
		assert m, name
		params = m.group(1).strip()
		n = 0 if params in ('', 'void') else len([p for p in params.split(',') if p.strip()])
		assert n == len(argtypes), f"{name}: header has {n} parameters, ctypes table has {len(argtypes)}"


def test_struct_layouts_match_the_header(tmp_path):
	"""The ctypes mirrors of the header's structs (photometry_amd/_lib.py) have the size and the field offsets a C compiler gives the
	header's own definitions: a C program that includes include/tessphot_hip.h prints them (gcc; no GPU, no library needed)."""
	import shutil, subprocess
	from photometry_amd import _lib
	if shutil.which('gcc') is None:
		pytest.skip('no C compiler')
	structs = {name: getattr(_lib, name) for name in ('tp_cube_desc', 'tp_k2p2_params', 'tp_zoom_image', 'tp_radial_image', 'tp_frames_stack')}
	lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "tessphot_hip.h"', 'int main(void) {']
	for name, cls in structs.items():
		lines.append(f'	printf("{name} size %zu\\n", sizeof({name}));')
		for field, _ in cls._fields_:
			lines.append(f'	printf("{name} {field} %zu\\n", offsetof({name}, {field}));')
	lines += ['	return 0;', '}']
	src = tmp_path / 'layout.c'
	src.write_text('\n'.join(lines))
	exe = tmp_path / 'layout'
	subprocess.run(['gcc', '-I', os.path.dirname(HEADER), str(src), '-o', str(exe)], check=True)
	out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
	seen = 0
	for line in out.splitlines():
		name, what, value = line.split()
		cls = structs[name]
		if what == 'size':
			assert ctypes.sizeof(cls) == int(value), (name, ctypes.sizeof(cls), value)
		else:
			assert getattr(cls, what).offset == int(value), (name, what, getattr(cls, what).offset, value)
		seen += 1
	assert seen == sum(len(c._fields_) + 1 for c in structs.values())
