"""Single-process, multi-device row-sharded SpMM: Python plumbing over mispmm_multi_csr_f32, mispmm_multi_ell_f32 (ELL by
rows) and mispmm_multi_bsrc_slots_bf16 (bf16 BSR-16 by block rows) (include/mispmm.h, "multi-GPU" section).  One host thread drives every device through per-device streams;
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


class _MultiBase:
    """What the sharded formats share: device slots, streams, peer access, the communicator, C buffers and their views."""

    def _setup(self, num_rows, num_cols, n_cols, devices, gather, bounds_rows, ldc, c_dtype=torch.float32):
        self.l = capi.lib()
        self.devices = [int(d) for d in devices]
        self.ndev = len(self.devices)
        self.n = int(n_cols)
        self.gather = GATHER_MODES[gather]
        self.num_rows, self.num_cols = num_rows, num_cols
        self.bounds = np.asarray(bounds_rows, dtype=np.uint32)            # in C rows
        self.ldc = int(ldc) if ldc else self.n
        self.streams = [torch.cuda.Stream(device=torch.device("cuda", d)) for d in self.devices]
        self.c = [torch.zeros((num_rows, self.ldc), dtype=c_dtype, device=torch.device("cuda", d)) for d in self.devices]
        distinct = sorted(set(self.devices))
        if len(distinct) > 1:
            arr = (ctypes.c_int * len(distinct))(*distinct)
            capi.check(self.l.mispmm_enable_peer_access(len(distinct), arr))
        self.comm = ctypes.c_void_p()
        if self.gather in (capi.GATHER_ALL_RCCL, capi.GATHER_ALL_RCCL_EQUAL):
            arr = (ctypes.c_int * self.ndev)(*self.devices)
            capi.check(self.l.mispmm_comm_create(ctypes.byref(self.comm), self.ndev, arr))
        self._devices = (ctypes.c_int * self.ndev)(*self.devices)
        self._streams = (ctypes.c_void_p * self.ndev)(*[s.cuda_stream for s in self.streams])
        self._c = _ptr_array(self.c)

    def sync(self):
        for s in self.streams:
            s.synchronize()

    def full_c(self, slot=0):
        return self.c[slot][:self.num_rows, :self.n]

    def sharded_c(self):
        return torch.cat([self.c[d][int(self.bounds[d]):int(self.bounds[d + 1]), :self.n].cpu() for d in range(self.ndev)]).numpy()

    def close(self):
        if self.comm:
            capi.check(self.l.mispmm_comm_destroy(self.comm))
            self.comm = ctypes.c_void_p()


class MultiEllSpmm(_MultiBase):
    """ELL sharded by rows: `ell` is a formats.ELLRowMajor (or the reference's column-major ELL,
    converted); rows are cut into nnz-balanced contiguous ranges (an ELL row's nnz = its occupied slots)."""

    def __init__(self, ell, n_cols, devices, gather="first", kernel=0, acc="reference", ldc=None):
        from . import formats
        if isinstance(ell, formats.ELLColMajor):
            ell = ops.colmajor_ell_to_rowmajor(ell)
        self.kernel, self.acc, self.width = int(kernel), capi.ACC_MODES[acc], int(ell.width)
        cols = np.ascontiguousarray(ell.col_idxs, dtype=np.uint32).reshape(ell.num_rows, -1)
        vals = np.ascontiguousarray(ell.data, dtype=np.float32).reshape(ell.num_rows, -1)
        occupied = np.concatenate([[0], np.cumsum((cols != 0xFFFFFFFF).sum(axis=1))]).astype(np.uint32)
        bounds = ops.shard_rows_by_nnz(occupied, len(devices))
        self._setup(ell.num_rows, ell.num_cols, n_cols, devices, gather, bounds, ldc)
        self.cols, self.vals, self.b = [], [], []
        for d, dev in enumerate(self.devices):
            tdev = torch.device("cuda", dev)
            r0, r1 = int(bounds[d]), int(bounds[d + 1])
            self.cols.append(ops._dev_u32(cols[r0:r1].reshape(-1), tdev))
            self.vals.append(ops._dev_f32(vals[r0:r1].reshape(-1), tdev))
            self.b.append(torch.zeros((self.num_cols, self.n), dtype=torch.float32, device=tdev))
        self._bounds = (ctypes.c_uint32 * (self.ndev + 1))(*[int(x) for x in bounds])
        self._cols, self._vals, self._b = _ptr_array(self.cols), _ptr_array(self.vals), _ptr_array(self.b)

    def set_b(self, b_host):
        src = torch.from_numpy(np.ascontiguousarray(b_host, dtype=np.float32))
        for t in self.b:
            t.copy_(src)
        self.sync()

    def step(self):
        capi.check(self.l.mispmm_multi_ell_f32(self.ndev, self._devices, self._streams, self._bounds, self.num_cols, self.width, self._cols,
                                               self._vals, self._b, self.n, self.n, self._c, self.ldc, self.kernel, self.acc, self.gather,
                                               self.comm))


def bsr_block_row_slice(bsr, b0, b1):
    """Block rows [b0, b1) of a host BSR as a stand-alone BSR (block row pointers rebased, block columns untouched)."""
    from . import formats
    s, e = int(bsr.block_row_ptrs[b0]), int(bsr.block_row_ptrs[b1])
    ptrs = (bsr.block_row_ptrs[b0:b1 + 1].astype(np.int64) - s).astype(np.uint32)
    data = np.asarray(bsr.data).reshape(bsr.num_blocks, bsr.block_row_size, bsr.block_col_size)[s:e].copy()
    return formats.BSR((b1 - b0) * bsr.block_row_size, bsr.num_cols, int(data.size), bsr.block_row_size, bsr.block_col_size, ptrs,
                       bsr.block_col_idxs[s:e].copy(), data)


class MultiBsrcSlotsSpmm(_MultiBase):
    """bf16 BSR-16 sharded by BLOCK rows (BASELINE config 4's kernel on every shard): block rows are
    cut into ranges balanced by block count, every shard is compacted into fixed step slots on its own."""

    def __init__(self, bsr, n_cols, devices, gather="first", out_bf16=False, ldc=None):
        if bsr.block_row_size != 16 or bsr.block_col_size != 16:
            raise ValueError("16 x 16 blocks (BASELINE config 4)")
        self.out_bf16 = bool(out_bf16)
        bb = ops.shard_rows_by_nnz(bsr.block_row_ptrs, len(devices))
        self.block_bounds = bb
        self._setup(bsr.num_rows, bsr.num_cols, n_cols, devices, gather, bb.astype(np.int64) * 16, ldc,
                    c_dtype=torch.int16 if out_bf16 else torch.float32)
        self.shards, self.b = [], []
        for d, dev in enumerate(self.devices):
            tdev = torch.device("cuda", dev)
            self.shards.append(ops.DeviceBSRCSlots.from_host(bsr_block_row_slice(bsr, int(bb[d]), int(bb[d + 1])), device=tdev))
            self.b.append(torch.zeros((self.num_cols, self.n), dtype=torch.int16, device=tdev))
        self._bounds = (ctypes.c_uint32 * (self.ndev + 1))(*[int(x) for x in bb])
        self._nsteps = (ctypes.c_uint32 * self.ndev)(*[s.num_steps for s in self.shards])
        self._extra = _ptr_array([s.extra_ptrs for s in self.shards])
        self._cols = _ptr_array([s.cols for s in self.shards])
        self._tiles = _ptr_array([s.tiles for s in self.shards])
        self._b = _ptr_array(self.b)

    def set_b(self, b_host):
        """b_host: fp32 [K, N]; rounded to bf16 on the first device, replicated."""
        first = ops.f32_to_bf16(torch.from_numpy(np.ascontiguousarray(b_host, dtype=np.float32)).to(self.b[0].device))
        for t in self.b:
            t.copy_(first)
        self.sync()
        torch.cuda.synchronize()

    def step(self):
        capi.check(self.l.mispmm_multi_bsrc_slots_bf16(self.ndev, self._devices, self._streams, self._bounds, self.num_cols, self._nsteps,
                                                       self._extra, self._cols, self._tiles, self._b, self.n, self.n, self._c, self.ldc,
                                                       int(self.out_bf16), self.gather, self.comm))
