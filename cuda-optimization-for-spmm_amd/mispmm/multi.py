"""Single-process, multi-device row-sharded CSR SpMM: Python plumbing over mispmm_multi_csr_f32
(include/mispmm.h, "multi-GPU" section).  One host thread drives every device through per-device streams;
torch supplies the device buffers and stream handles, the C ABI does the work.

New capability: the reference selects one device (src/main.cu:176).  The one-process-per-GPU variant over
torch.distributed is mispmm/dist.py; this module is the path the C++ host (`cuspmm --gpus n`) uses.
"""
import ctypes

import numpy as np
import torch

from . import capi, ops
from .dist import csr_row_slice

GATHER_MODES = {"none": capi.GATHER_NONE, "first": capi.GATHER_TO_FIRST, "peer": capi.GATHER_ALL_PEER,
                "rccl": capi.GATHER_ALL_RCCL, "rccl-equal": capi.GATHER_ALL_RCCL_EQUAL}


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t.numel() else 0
    return arr


class MultiCsrSpmm:
    """devices: list of ordinals, one per slot (the same ordinal may appear twice: a rehearsal of the bookkeeping
    on one card -- two slabs, two streams, real copies -- RCCL itself refuses duplicate devices)."""

    def __init__(self, csr, n_cols, devices, gather="first", kernel=0, acc="reference", ldc=None):
        l = capi.lib()
        self.l = l
        self.devices = [int(d) for d in devices]
        self.ndev = len(self.devices)
        self.n = int(n_cols)
        self.kernel, self.acc, self.gather = int(kernel), capi.ACC_MODES[acc], GATHER_MODES[gather]
        self.num_rows, self.num_cols = csr.num_rows, csr.num_cols
        self.bounds = ops.shard_rows_by_nnz(csr.row_ptrs, self.ndev)
        self.ldc = int(ldc) if ldc else self.n          # ldc > n: a strided C (gap columns must survive the gather)
        c_rows = self.num_rows
        if self.gather == capi.GATHER_ALL_RCCL_EQUAL:
            # equal row chunks (the last may be short), C padded to ndev * chunk rows: one ncclAllGather per device
            chunk = -(-self.num_rows // self.ndev)
            self.bounds = np.minimum(np.arange(self.ndev + 1, dtype=np.int64) * chunk, self.num_rows).astype(np.uint32)
            c_rows = chunk * self.ndev
        self.slices, self.streams, self.b, self.c = [], [], [], []
        for d, dev in enumerate(self.devices):
            tdev = torch.device("cuda", dev)
            local = csr_row_slice(csr, int(self.bounds[d]), int(self.bounds[d + 1]))
            self.slices.append(ops.DeviceCSR.from_host(local, device=tdev))
            self.streams.append(torch.cuda.Stream(device=tdev))
            self.b.append(torch.zeros((self.num_cols, self.n), dtype=torch.float32, device=tdev))
            self.c.append(torch.zeros((c_rows, self.ldc), dtype=torch.float32, device=tdev))
        distinct = sorted(set(self.devices))
        if len(distinct) > 1:
            arr = (ctypes.c_int * len(distinct))(*distinct)
            capi.check(l.mispmm_enable_peer_access(len(distinct), arr))
        self.comm = ctypes.c_void_p()
        if self.gather in (capi.GATHER_ALL_RCCL, capi.GATHER_ALL_RCCL_EQUAL):
            arr = (ctypes.c_int * self.ndev)(*self.devices)
            capi.check(l.mispmm_comm_create(ctypes.byref(self.comm), self.ndev, arr))
        # argument arrays (host arrays of device pointers), built once
        self._devices = (ctypes.c_int * self.ndev)(*self.devices)
        self._streams = (ctypes.c_void_p * self.ndev)(*[s.cuda_stream for s in self.streams])
        self._bounds = (ctypes.c_uint32 * (self.ndev + 1))(*[int(x) for x in self.bounds])
        self._row_ptrs = _ptr_array([s.row_ptrs for s in self.slices])
        self._col_idxs = _ptr_array([s.col_idxs for s in self.slices])
        self._vals = _ptr_array([s.data for s in self.slices])
        self._nnz = (ctypes.c_uint32 * self.ndev)(*[s.nnz for s in self.slices])
        self._uniform = (ctypes.c_uint32 * self.ndev)(*[s.uniform_row_nnz for s in self.slices])
        self._b = _ptr_array(self.b)
        self._c = _ptr_array(self.c)

    def set_b(self, b_host):
        """Replicate the dense operand: one H2D copy per device slot (outside any timed region)."""
        src = torch.from_numpy(np.ascontiguousarray(b_host, dtype=np.float32))
        for t in self.b:
            t.copy_(src)
        self.sync()

    def step(self):
        capi.check(self.l.mispmm_multi_csr_f32(self.ndev, self._devices, self._streams, self._bounds, self.num_cols,
                                               self._row_ptrs, self._col_idxs, self._vals, self._nnz, self._uniform, self._b,
                                               self.n, self.n, self._c, self.ldc, self.kernel, self.acc, self.gather, self.comm))

    def sync(self):
        for s in self.streams:
            s.synchronize()

    def full_c(self, slot=0):
        """C as device slot `slot` holds it after sync(): complete for slot 0 (gather first) or any slot (peer, rccl)."""
        return self.c[slot][:self.num_rows, :self.n]

    def sharded_c(self):
        """Every slot's own rows, concatenated on the host (valid for every gather mode, including none)."""
        return np.concatenate([self.c[d][int(self.bounds[d]):int(self.bounds[d + 1]), :self.n].cpu().numpy() for d in range(self.ndev)])

    def close(self):
        if self.comm:
            capi.check(self.l.mispmm_comm_destroy(self.comm))
            self.comm = ctypes.c_void_p()
