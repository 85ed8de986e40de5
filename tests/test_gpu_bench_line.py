# -*- coding: utf-8 -*-
"""bench.py end to end on the GPU box, small: the stdout of a run is ONE short line the driver can parse (benchlib.line.check_line),
at N = 1 and -- two ranks sharing the one device: a control-flow run through the host group -- at N = 2, started both ways
(spawned by bench.py itself, and by torch.distributed.run as the driver does)."""
import json
import os
import subprocess
import sys
import pytest
from benchlib import line as bl

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['--steps', '3', '--warmup', '1', '--targets', '512', '--cadences', '200', '--cpu-sample', '0', '--no-extra']


def _run(cmd, tmp_path):
	env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'TESSPHOT_RDZV_ID')}
	r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
	assert r.returncode == 0, r.stderr.decode()[-3000:]
	out = r.stdout.decode().strip().splitlines()
	assert len(out) == 1, out                       # nothing but the line on stdout
	return bl.check_line(out[0])


def test_single_gpu_line(tmp_path):
	d = _run([sys.executable, 'bench.py', '--gpus', '1'] + SMALL, tmp_path)
	assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['value'] > 0
	assert d['config']['workload'].startswith('configs[2]') and d['roofline']['frac'] > 0
	full = json.load(open(os.path.join(ROOT, bl.LEGS_FILE)))
	assert full['value'] == pytest.approx(d['value'], rel=1e-5) and 'kernels' in full


@pytest.mark.parametrize('launcher', ['own', 'torchrun'])
def test_two_ranks_on_one_device_line(tmp_path, launcher):
	if launcher == 'own':
		cmd = [sys.executable, 'bench.py', '--gpus', '2'] + SMALL
	else:
		pytest.importorskip('torch')
		cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
			'--master-port', str(26500 + os.getpid() % 2000), 'bench.py', '--gpus', '2'] + SMALL
	d = _run(cmd, tmp_path)
	assert d['n_gpus'] == 2 and d['config']['workload'].startswith('configs[4]') and d['config']['targets_total'] == 1024
	g = d['gather']
	assert g['mode'].startswith('host') and g['issued_short'] in ('step', 'final') and g['measured_8gpu'] is False
	assert g['step_ms_without_gather'] > 0 and (g['mean_ms'] or g['final_ms'])
	assert 'warning' in d                           # two ranks on one GPU: said so in the line
