"""Ghost-value import over RCCL / xGMI.

Replaces the three ``rel_* = *`` owned->relevant imports at cracks.cc:2147-2154 (Trilinos
Import => MPI point-to-point).  Pattern = neighbour halo exchange, not a reduction: per
peer one packed message of ``(dim+3)`` doubles per interface node (u, phi, phi_old,
phi_oldold), all peers posted together with ``batch_isend_irecv`` so that every xGMI link
of the GPU carries its own pair concurrently.  Packing/unpacking are HIP kernels behind
the C ABI (``pfm_halo_pack_all`` / ``pfm_halo_unpack_all``: one launch each for all peers).

Transport.  With one GPU per rank (``torch.distributed`` backend ``nccl``) the whole exchange is ONE library call,
``pfm_halo_exchange``: pack -> ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd -> unpack on the context's stream,
on a communicator the library creates (``pfm_comm_create``; torch.distributed only broadcasts the 128-byte unique id).
That is the code path a C++ host (the deal.II application, no torch) uses.  ``PFM_HALO_TORCH=1`` moves the buffers with
``torch.distributed.batch_isend_irecv`` instead; ``gloo`` (CPU tests, single-GPU smoke runs) always does, staging
device buffers through the host.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Sequence

import numpy as np


class HaloExchange:
    def __init__(self, dim: int, peers: Sequence[int], send_ptr, send_nodes, recv_ptr, recv_nodes,
                 device, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.dim = dim
        self.peers = list(peers)
        self.send_ptr = np.asarray(send_ptr, np.int64)
        self.recv_ptr = np.asarray(recv_ptr, np.int64)
        self.send_nodes = np.asarray(send_nodes, np.int32)
        self.recv_nodes = np.asarray(recv_nodes, np.int32)
        self.group = group
        self.rec = dim + 3  # PFM_HALO_DOUBLES_PER_NODE
        ns = np.diff(self.send_ptr)
        nr = np.diff(self.recv_ptr)
        # one contiguous buffer per direction; peer k's message is the slice [rec * ptr[k], rec * ptr[k + 1])
        self.send_all = torch.empty(int(ns.sum()) * self.rec, dtype=torch.float64, device=device)
        self.recv_all = torch.empty(int(nr.sum()) * self.rec, dtype=torch.float64, device=device)
        so = (self.send_ptr - self.send_ptr[0]) * self.rec
        ro = (self.recv_ptr - self.recv_ptr[0]) * self.rec
        self.send_bufs = [self.send_all[int(so[k]):int(so[k + 1])] for k in range(len(self.peers))]
        self.recv_bufs = [self.recv_all[int(ro[k]):int(ro[k + 1])] for k in range(len(self.peers))]
        self._registered = None
        # gloo moves host memory only: device buffers are staged through the host (smoke tests of the multi-process
        # flow on a box without one GPU per rank; the product path is backend "nccl" = RCCL, device to device)
        self._stage_host = self.send_all.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo"
        # in-library RCCL transport (pfm_halo_exchange) whenever every rank has its own GPU
        self._use_lib = (self.send_all.is_cuda and dist.is_initialized() and dist.get_backend(group) == "nccl"
                         and os.environ.get("PFM_HALO_TORCH") != "1")
        self._comm = None
        self._comm_lib = None

    @property
    def bytes_per_exchange(self) -> int:
        return int(sum(b.numel() for b in self.send_bufs) * 8)

    def register(self, ctx):
        ctx.halo_register(self.send_ptr, self.send_nodes, self.recv_ptr, self.recv_nodes)
        self._registered = ctx

    def _post(self):
        dist = self.dist
        if self._stage_host:
            send_h, recv_h = self.send_all.cpu(), self.torch.empty(self.recv_all.shape, dtype=self.torch.float64)
            so = (self.send_ptr - self.send_ptr[0]) * self.rec
            ro = (self.recv_ptr - self.recv_ptr[0]) * self.rec
            ops = [dist.P2POp(dist.irecv, recv_h[int(ro[k]):int(ro[k + 1])], peer, group=self.group)
                   for k, peer in enumerate(self.peers) if ro[k + 1] > ro[k]]
            ops += [dist.P2POp(dist.isend, send_h[int(so[k]):int(so[k + 1])], peer, group=self.group)
                    for k, peer in enumerate(self.peers) if so[k + 1] > so[k]]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            self.recv_all.copy_(recv_h)
            return
        ops = []
        for k, peer in enumerate(self.peers):
            if self.recv_bufs[k].numel():
                ops.append(dist.P2POp(dist.irecv, self.recv_bufs[k], peer, group=self.group))
        for k, peer in enumerate(self.peers):
            if self.send_bufs[k].numel():
                ops.append(dist.P2POp(dist.isend, self.send_bufs[k], peer, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def _ensure_comm(self, ctx):
        """ncclCommInitRank through the C ABI; torch.distributed only carries the unique id to the other ranks."""
        if self._comm is not None:
            return
        torch, dist = self.torch, self.dist
        from . import capi

        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        uid = np.zeros(capi.COMM_ID_BYTES + 1, np.uint8)  # last byte: rank 0 obtained an id
        if rank == 0:
            rc0 = ctx.lib.pfm_comm_unique_id(capi.np_ptr(uid, np.uint8))
            uid[-1] = 1 if rc0 == capi.PFM_OK else 0
        # every rank takes part in the broadcast whatever rank 0's outcome was: nobody is left waiting in a collective
        t = torch.from_numpy(uid).to(self.send_all.device)
        dist.broadcast(t, 0, group=self.group)
        uid = np.ascontiguousarray(t.cpu().numpy())
        if uid[-1] == 0:
            import warnings

            warnings.warn("pfm_comm_unique_id failed on rank 0 (RCCL unavailable?): ghost exchange uses torch.distributed P2P")
            self._use_lib = False
            return
        uid = np.ascontiguousarray(uid[:-1])
        h = C.c_void_p()
        rc = ctx.lib.pfm_comm_create(C.byref(h), capi.np_ptr(uid, np.uint8), world, rank, self.send_all.device.index)
        # the choice of transport must be the same on every rank: agree on the outcome
        ok = torch.tensor([1 if rc == capi.PFM_OK else 0], device=self.send_all.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            if rc == capi.PFM_OK:
                ctx.lib.pfm_comm_destroy(h)
            import warnings

            warnings.warn("pfm_comm_create failed on some rank: ghost exchange falls back to torch.distributed P2P")
            self._use_lib = False
            return
        self._comm, self._comm_lib = h, ctx.lib

    def library_comm(self, ctx):
        """The ncclComm_t handle of the in-library transport (made on first use), or None when the exchange goes through
        torch.distributed (gloo smoke runs, PFM_HALO_TORCH=1, RCCL unavailable)."""
        if not self._use_lib:
            return None
        if self._registered is not ctx:
            self.register(ctx)
        self._ensure_comm(ctx)
        return self._comm.value if (self._use_lib and self._comm is not None) else None

    def rccl_info(self):
        """What RCCL reports for the communicator the exchange uses: {"rccl_nranks": ncclCommCount, "rccl_rank":
        ncclCommUserRank, "rccl_version": ncclGetVersion}, or None when the exchange does not go through the library's
        RCCL transport (gloo runs, PFM_HALO_TORCH=1, one rank)."""
        if self._comm is None:
            return None
        n, r, ver = C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
        if self._comm_lib.pfm_comm_info(self._comm, C.byref(n), C.byref(r), C.byref(ver)) != 0:
            return None
        return {"rccl_nranks": int(n.value), "rccl_rank": int(r.value), "rccl_version": int(ver.value)}

    def close(self):
        if self._comm is not None:
            self._comm_lib.pfm_comm_destroy(self._comm)
            self._comm = None

    def exchange(self, ctx):
        """pack (HIP) -> RCCL send/recv -> unpack (HIP), all on torch's current stream."""
        if self._registered is not ctx:
            self.register(ctx)
        if self._use_lib:
            self._ensure_comm(ctx)
        if self._use_lib:
            # Every local, non-blocking precondition (library loaded, communicator made, lists registered) was agreed on
            # by all ranks in _ensure_comm BEFORE any RCCL work was enqueued.  From here on a failure is fatal: peers may
            # already have enqueued the matching send/receive, so there is no fall-back -- the library aborts the
            # communicator (pfm_halo_exchange) and the error propagates.
            ctx.halo_exchange(self._comm.value, self.peers)
            return
        if self.send_all.numel():
            ctx.halo_pack_all(self.send_all.data_ptr())  # one launch for all peers
        self._post()
        if self.recv_all.numel():
            ctx.halo_unpack_all(self.recv_all.data_ptr())

    def exchange_with(self, pack: Callable[[int, np.ndarray], "object"], unpack: Callable[[int, np.ndarray, "object"], None]):
        """Same exchange with caller-supplied pack/unpack (CPU/gloo tests of the lists)."""
        for k in range(len(self.peers)):
            nodes = self.send_nodes[self.send_ptr[k]:self.send_ptr[k + 1]]
            if nodes.size:
                self.send_bufs[k].copy_(pack(k, nodes).reshape(-1))
        self._post()
        for k in range(len(self.peers)):
            nodes = self.recv_nodes[self.recv_ptr[k]:self.recv_ptr[k + 1]]
            if nodes.size:
                unpack(k, nodes, self.recv_bufs[k])
