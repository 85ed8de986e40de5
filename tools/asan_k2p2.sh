#!/bin/bash
# AddressSanitizer + UBSan over the K2P2 kernel logic (csrc/k2p2_core.h compiled for the host, tests/hostsim): the GPU pool has no
# device sanitizer, so the index arithmetic of the mask builder is checked on the CPU build.  Usage: bash tools/asan_k2p2.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/tp_asan
g++ -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -Itests/hostsim -o /tmp/tp_asan/k2p2_hostsim_asan.so tests/hostsim/k2p2_hostsim.cpp
cat > /tmp/tp_asan/run.py <<'PY'
import sys, os, ctypes
root = os.getcwd()
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import test_k2p2_hostsim as t
from k2p2_common import make_cases, oracle_batch, compare
lib = ctypes.CDLL('/tmp/tp_asan/k2p2_hostsim_asan.so')
lib.hostsim_k2p2.restype = ctypes.c_int
for kind, seed in [('faint15', 1), ('small11', 2), ('crowded', 3), ('bright', 4), ('tiny', 5), ('large', 21)]:
	s, S = make_cases(kind, seed)
	print(kind, compare(s, S, t.run_hostsim(lib, s, S), oracle_batch(s, S)), flush=True)
PY
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python /tmp/tp_asan/run.py
