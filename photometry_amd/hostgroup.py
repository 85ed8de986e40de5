# -*- coding: utf-8 -*-
"""
The host-side group of a multi-rank run: rendezvous, barrier, max / min over ranks, a small broadcast (the 128-byte id of the
RCCL communicator) and -- only where the ranks have no device of their own -- the gather of the output blocks through host
memory.  It carries control messages; the data path between GPUs is ``tp_comm_gather`` (RCCL over xGMI).

The reference's counterpart is the mpi4py world of ``run_tessphot_mpi.py:74-209`` (``comm.send`` / ``comm.recv`` of pickled
task dicts between a master and its workers).

Two implementations of one interface (``rank``, ``world``, ``barrier()``, ``max(x)``, ``min(x)``, ``broadcast_bytes(b, src)``,
``allgather_int(i)``, ``gather_array(a, dst)``, ``close()``):

* :class:`SocketGroup` -- plain TCP sockets in a star around rank 0, standard library only: **a multi-GPU run imports no
  PyTorch**.  Two ways to find rank 0:

  - ONE NODE (the default): rank 0 listens on an ephemeral port of ``MASTER_ADDR`` and announces it, with a random secret,
    in a rendezvous file (mode 0600, in the local temporary directory) that the ranks of one launcher share -- they are
    children of one process: ``python -m torch.distributed.run``'s agent, or ``sharded.spawn_ranks``.  ``MASTER_PORT`` itself
    is NOT used as a listening port: under torch.distributed.run it belongs to the launcher's own store.  This way cannot
    span nodes (the file is local).
  - SEVERAL NODES: with ``TESSPHOT_RDZV_PORT`` set -- or, when the launcher's environment says the run spans nodes
    (``WORLD_SIZE > LOCAL_WORLD_SIZE``), ``MASTER_PORT + 1`` -- rank 0 listens on that port on all interfaces and the other
    ranks connect to ``MASTER_ADDR`` at it; no file.  The handshake token is then ``TESSPHOT_RDZV_SECRET`` if the launcher
    exported one to every rank, else the run id (guessable: use the secret on a shared network).
* :class:`TorchGroup` -- ``torch.distributed`` with the gloo backend (what the CPU tests of the sharded run also exercise).
"""

import os
import socket
import struct
import sys
import tempfile
import time
import numpy as np

_MAGIC = b'TPHG0001'


class GroupError(RuntimeError):
	pass


def _recv_exact(sock, n, into=None):
	buf = into if into is not None else bytearray(n)
	view = memoryview(buf)
	got = 0
	while got < n:
		k = sock.recv_into(view[got:n], min(n - got, 1 << 24))
		if k == 0:
			raise GroupError('a rank closed its connection (it has probably failed)')
		got += k
	return buf


def _send_msg(sock, payload):
	sock.sendall(struct.pack('<Q', len(payload)))
	if len(payload):
		sock.sendall(payload)


_MAX_CONTROL = 64 << 20      # no control message comes near this (the largest is the allgather blob of small payloads)


def _recv_msg(sock, limit=_MAX_CONTROL):
	"""A length-prefixed message.  The length comes from the peer: it is checked against ``limit`` BEFORE anything is allocated."""
	n, = struct.unpack('<Q', bytes(_recv_exact(sock, 8)))
	if n > limit:
		raise GroupError('a peer announced a message of %d bytes (limit %d): not one of ours' % (n, limit))
	return bytes(_recv_exact(sock, n)) if n else b''


def network_rendezvous_port():
	"""The port of the several-node rendezvous (module docstring), or None for the one-node rendezvous file."""
	p = os.environ.get('TESSPHOT_RDZV_PORT')
	if p:
		return int(p)
	try:
		world, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_WORLD_SIZE', '0'))
	except ValueError:
		return None
	if local > 0 and world > local and os.environ.get('MASTER_PORT'):
		return int(os.environ['MASTER_PORT']) + 1
	return None


def rendezvous_file():
	"""
	The file in which rank 0 announces its port: named after what the ranks of ONE launch share and another launch does not --
	``TESSPHOT_RDZV_ID`` when the launcher set it (``sharded.spawn_ranks``), else the launcher's process id (the ranks' common
	parent) with ``MASTER_PORT`` and the torchelastic run id.
	"""
	tag = os.environ.get('TESSPHOT_RDZV_ID')
	if not tag:
		tag = '%d_%s_%s' % (os.getppid(), os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'))
	tag = ''.join(c if c.isalnum() or c in '_-' else '_' for c in tag)
	return os.path.join(tempfile.gettempdir(), 'tessphot_rdzv_%d_%s' % (os.getuid(), tag))


class SocketGroup(object):
	"""TCP star around rank 0 (see the module docstring).  Every collective is entered by all ranks in the same order."""

	def __init__(self, rank, world, addr=None, timeout=1800.0, rendezvous_timeout=300.0):
		self.rank, self.world = int(rank), int(world)
		self.addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
		self.timeout = float(timeout)
		self.peers = {}          # rank 0: rank -> socket
		self.sock = None         # other ranks: the connection to rank 0
		self._path = None
		if self.world <= 1:
			return
		net_port = network_rendezvous_port()
		path = None if net_port is not None else rendezvous_file()
		if net_port is not None:
			secret = os.environ.get('TESSPHOT_RDZV_SECRET') or '%s_%s' % (os.environ.get('TORCHELASTIC_RUN_ID', 'none'), os.environ.get('MASTER_PORT', '0'))
			token = ('net:%s:%d' % (secret, self.world)).encode()
		if self.rank == 0:
			srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
			srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
			if net_port is not None:
				srv.bind(('', net_port))
			else:
				srv.bind((self.addr, 0))
			srv.listen(self.world + 8)
			port = srv.getsockname()[1]
			if path is not None:
				secret = os.urandom(16).hex()
				token = ('%s:%s:%d' % (path, secret, self.world)).encode()
				tmp = '%s.%d.tmp' % (path, os.getpid())
				fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
				with os.fdopen(fd, 'w') as fh:
					fh.write('%d %d %s\n' % (port, os.getpid(), secret))
				os.replace(tmp, path)               # atomic: a reader sees the old file or the whole new one
				self._path = path
			srv.settimeout(rendezvous_timeout)
			try:
				while len(self.peers) < self.world - 1:
					try:
						c, _ = srv.accept()
					except socket.timeout:
						raise GroupError('rendezvous: %d of %d ranks arrived within %.0f s' % (len(self.peers) + 1, self.world, rendezvous_timeout))
					c.settimeout(30.0)
					try:
						hello = _recv_msg(c, limit=1 << 20)
						if not hello.startswith(_MAGIC) or hello[len(_MAGIC) + 4:] != token:
							c.close()               # not one of ours (a stale file pointed a stranger here)
							continue
						r, = struct.unpack('<i', hello[len(_MAGIC):len(_MAGIC) + 4])
						if not (0 < r < self.world) or r in self.peers:
							c.close()
							continue
						_send_msg(c, _MAGIC)
					except (GroupError, OSError, MemoryError, OverflowError, struct.error):
						c.close()                   # a stray connection must not take rank 0 down
						continue
					c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
					c.settimeout(self.timeout)
					self.peers[r] = c
			finally:
				srv.close()
		else:
			deadline = time.monotonic() + rendezvous_timeout
			last = 'no rendezvous file' if path is not None else 'no connection'
			while self.sock is None:
				if time.monotonic() > deadline:
					raise GroupError('rendezvous: rank %d could not reach rank 0 (%s; %s)' % (self.rank, last,
						('file %s' % path) if path is not None else ('%s port %d' % (self.addr, net_port))))
				if path is not None:
					try:
						fields = open(path).read().split()
						port = int(fields[0])
						token = ('%s:%s:%d' % (path, fields[2], self.world)).encode()
					except (OSError, ValueError, IndexError):
						time.sleep(0.05)
						continue
				else:
					port = net_port
				s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
				s.settimeout(10.0)
				try:
					s.connect((self.addr, port))
					_send_msg(s, _MAGIC + struct.pack('<i', self.rank) + token)
					if _recv_msg(s, limit=1 << 20) != _MAGIC:
						raise GroupError('handshake refused')
				except (OSError, GroupError) as e:   # a stale file of an earlier launch, or rank 0 not listening yet
					last = str(e)
					s.close()
					time.sleep(0.1)
					continue
				s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
				s.settimeout(self.timeout)
				self.sock = s
		self.barrier()
		if self.rank == 0 and path is not None:     # every rank is connected: the file has done its work
			try:
				os.unlink(path)
			except OSError:
				pass
			self._path = None

	# ---- the one primitive: everybody's small payload to everybody ----------------------------------
	def _allgather(self, payload):
		if self.world <= 1:
			return [payload]
		if self.rank == 0:
			parts = [payload] + [None] * (self.world - 1)
			for r, c in self.peers.items():
				parts[r] = _recv_msg(c)
			blob = b''.join(struct.pack('<Q', len(p)) + p for p in parts)
			for c in self.peers.values():
				_send_msg(c, blob)
			return parts
		_send_msg(self.sock, payload)
		blob = _recv_msg(self.sock)
		parts, off = [], 0
		for _ in range(self.world):
			n, = struct.unpack_from('<Q', blob, off)
			parts.append(blob[off + 8:off + 8 + n])
			off += 8 + n
		return parts

	def barrier(self):
		self._allgather(b'')

	def max(self, value):
		return max(struct.unpack('<d', p)[0] for p in self._allgather(struct.pack('<d', float(value))))

	def min(self, value):
		return min(struct.unpack('<d', p)[0] for p in self._allgather(struct.pack('<d', float(value))))

	def allgather_int(self, value):
		return [struct.unpack('<q', p)[0] for p in self._allgather(struct.pack('<q', int(value)))]

	def broadcast_bytes(self, payload, src=0):
		return self._allgather(bytes(payload) if self.rank == src else b'')[src]

	def gather_array(self, array, dst=0):
		"""Every rank's array (same shape and dtype) to rank 0: a list in rank order there, None elsewhere.  Rank 0 only as ``dst``."""
		if dst != 0:
			raise ValueError('SocketGroup gathers to rank 0')
		a = np.ascontiguousarray(array)
		if self.world <= 1:
			return [a]
		if self.rank != 0:
			self.sock.sendall(struct.pack('<Q', a.nbytes))
			self.sock.sendall(memoryview(a).cast('B'))
			return None
		out = [a] + [None] * (self.world - 1)
		for r, c in self.peers.items():
			n, = struct.unpack('<Q', bytes(_recv_exact(c, 8)))
			if n != a.nbytes:
				raise GroupError('gather: rank %d sent %d bytes, rank 0 holds %d' % (r, n, a.nbytes))
			buf = np.empty(a.shape, dtype=a.dtype)
			_recv_exact(c, n, into=memoryview(buf).cast('B'))
			out[r] = buf
		return out

	def close(self):
		for c in list(self.peers.values()) + ([self.sock] if self.sock is not None else []):
			try:
				c.close()
			except OSError:
				pass
		self.peers, self.sock = {}, None
		if self._path:
			try:
				os.unlink(self._path)
			except OSError:
				pass
			self._path = None


class TorchGroup(object):
	"""The same interface over an initialised ``torch.distributed`` group (gloo)."""

	def __init__(self, dist, torch, rank=None, world=None):
		self.dist, self.torch = dist, torch
		self.rank = dist.get_rank() if rank is None else int(rank)
		self.world = dist.get_world_size() if world is None else int(world)

	@classmethod
	def init(cls, rank, world, timeout_s=1800):
		"""Import torch (BEFORE the HIP library, so that one HIP runtime is shared) and start the gloo group.  gloo announces its
		connections on stdout at C level: stdout is kept clean for the caller's one result line."""
		import datetime
		import torch
		import torch.distributed as dist
		os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
		sys.stdout.flush()
		saved = os.dup(1)
		os.dup2(2, 1)
		try:
			dist.init_process_group(backend='gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
		finally:
			os.dup2(saved, 1)
			os.close(saved)
		return cls(dist, torch, rank, world)

	def barrier(self):
		self.dist.barrier()

	def _reduce(self, value, op):
		t = self.torch.tensor([float(value)], dtype=self.torch.float64)
		self.dist.all_reduce(t, op=op)
		return float(t[0])

	def max(self, value):
		return self._reduce(value, self.dist.ReduceOp.MAX)

	def min(self, value):
		return self._reduce(value, self.dist.ReduceOp.MIN)

	def allgather_int(self, value):
		out = [None] * self.world
		self.dist.all_gather_object(out, int(value))
		return [int(v) for v in out]

	def broadcast_bytes(self, payload, src=0):
		obj = [bytes(payload) if self.rank == src else None]
		self.dist.broadcast_object_list(obj, src=src)
		return obj[0]

	def gather_array(self, array, dst=0):
		h = self.torch.from_numpy(np.ascontiguousarray(array))
		out = [self.torch.empty_like(h) for _ in range(self.world)] if self.rank == dst else None
		self.dist.gather(h, out, dst=dst)
		return [o.numpy() for o in out] if self.rank == dst else None

	def close(self):
		if self.dist.is_initialized():
			self.dist.destroy_process_group()


class SingleGroup(object):
	"""One rank: nothing to exchange."""
	rank, world = 0, 1

	def barrier(self):
		pass

	def max(self, value):
		return float(value)

	min = max

	def allgather_int(self, value):
		return [int(value)]

	def broadcast_bytes(self, payload, src=0):
		return bytes(payload)

	def gather_array(self, array, dst=0):
		return [np.ascontiguousarray(array)]

	def close(self):
		pass


def open_group(rank, world, kind='socket'):
	"""``kind``: ``'socket'`` (no PyTorch; ONE node unless ``TESSPHOT_RDZV_PORT`` / a multi-node launcher environment selects the
	network rendezvous, see :class:`SocketGroup`) or ``'gloo'`` (``torch.distributed`` over MASTER_ADDR / MASTER_PORT)."""
	if world <= 1:
		return SingleGroup()
	if kind == 'gloo':
		return TorchGroup.init(rank, world)
	if kind != 'socket':
		raise ValueError("host group kind must be 'socket' or 'gloo'")
	return SocketGroup(rank, world)
