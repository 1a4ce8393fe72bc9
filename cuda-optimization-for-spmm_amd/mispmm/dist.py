"""Row-sharded SpMM across the GPUs of one node (one process per GPU, RCCL over xGMI).

New capability: the reference is single-GPU (`cudaSetDevice(7)`, src/main.cu:176).  Rows of A --
hence rows of C -- are independent, so A is cut into `world` contiguous row ranges of near-equal
nnz (mispmm_shard_rows_by_nnz_host), B is replicated once by broadcast, every rank multiplies its
slab with the single-GPU kernel, and the only exchange is the gather of C row slabs.  Slabs are
padded to the tallest slab so one `all_gather_into_tensor` moves a whole bucket of steps; the
collective runs on its own stream and overlaps the next bucket's kernels (xGMI is point to point:
few, large messages).

Two ways the slabs travel (`exchange`):
  allgather  one `all_gather_into_tensor` (RCCL) per bucket on a second stream, overlapped with the next bucket;
  peer       no collective on the data path: every rank maps the other ranks' gather buffers into its own
             address space (IPC handles exchanged once through torch.distributed) and ONE kernel per bucket
             (mispmm_slab_scatter, part of the bucket's hipGraph) stores the rank's slabs straight into every
             peer's buffer over xGMI; a one-element all-reduce per bucket tells a rank that all its peers'
             stores have landed.  A rank may read bucket i's gathered C once `finish()` or the wait for that
             bucket has returned, and must be done with it before it contributes to bucket i + 1's barrier
             (the peers overwrite the buffer two buckets later, gated on that barrier).

Why a reader of the peer exchange sees its peers' stores (the argument, link by link):
  1. writer: the scatter kernel's stores go to hipMalloc'ed (coarse-grained) memory of another device through an
     IPC mapping; whatever the writer's caches still hold is written back by the END-OF-KERNEL RELEASE of that launch.
     The writer then records an event on its compute stream and makes the communication stream wait for it before it
     enqueues its contribution to the one-element all-reduce: an event record is a barrier packet with a system-scope
     release, so the contribution cannot be sent before the slab bytes have left the writer for the destination's HBM
     (xGMI writes are posted in order per link; the all-reduce's own data travels behind them).
  2. the all-reduce completes on a reader only after EVERY rank has contributed, i.e. after every writer passed 1.
  3. reader: it consumes the gathered buffer in a LATER kernel (or copy) ordered behind that all-reduce
     (`_wait(buf, stream)` / `finish()`); a kernel launch begins with an acquire that invalidates the reader's L1s and
     the non-coherent lines of its L2s, so no stale copy of the buffer from two buckets ago can be read.
  4. re-use: a writer overwrites ring buffer `buf` two buckets later and only after waiting for the all-reduce of the
     bucket in between, which every reader enters after it is done with `buf` (program order on the reader).
Final equality with the unsharded product checks the data; `debug_sentinel=True` (bench: MISPMM_DIST_DEBUG=1) checks the
ORDER: behind every slab scatter each rank stores the bucket's sequence number into a 16-byte sentinel slot of every
peer (same stream, so it is written last), and a reader that has waited for the bucket asserts that all of its
sentinel slots already carry that number -- a missing release or a too-early reader shows up as a stale sentinel even
when the slab bytes happen to be identical from bucket to bucket, as they are in a benchmark.

Kernel-only steps (`run(..., gather=False)`: C left row-sharded) write ONE resident slab, as the single-GPU bench loop
overwrites one C: with a slot of its own per step (what an exchange needs) the slabs stream to fresh addresses beyond
the Infinity Cache, and round 3's N = 1 distributed line read 6-9 % above the plain line for that reason alone.

`batch = b` (kernel-only steps): b dense operands in buffers of their own (replicas of B) are multiplied per launch
(mispmm_csr_batch_f32) -- the ~1.2 us between two dependent launches is paid once per b products, the only form in which
a 788-row shard of this problem can scale.

The compute step is injectable so the partition / bucket / gather logic is exercised on CPU with
the gloo backend (tests/test_dist_cpu.py); on a GPU the default is the HIP kernel via the C ABI.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import capi, formats, ops


def shard_bounds(row_ptrs, parts):
    return ops.shard_rows_by_nnz(row_ptrs, parts).astype(np.int64)


def csr_row_slice(csr, r0, r1):
    """Rows [r0, r1) of a CSR as a stand-alone CSR (row pointers rebased, columns untouched)."""
    s, e = int(csr.row_ptrs[r0]), int(csr.row_ptrs[r1])
    ptrs = (csr.row_ptrs[r0:r1 + 1].astype(np.int64) - s).astype(np.uint32)
    return formats.CSR(r1 - r0, csr.num_cols, ptrs, csr.col_idxs[s:e].copy(), csr.data[s:e].copy())


class ShardedCsrSpmm:
    """CSR sharded by rows.  ShardedEllSpmm (ELL by rows) and ShardedBsrcSlotsSpmm (bf16 BSR-16 by block rows) below reuse
    everything but the four hooks `_partition`, `_make_local`, `_alloc_b` and `_hip_compute`."""
    b_dtype = torch.float32

    # -- hooks -------------------------------------------------------------------------------------
    @staticmethod
    def _partition(csr, world):
        """Row bounds of the shards, nnz-balanced (int64 array of world + 1 C-row indices)."""
        return shard_bounds(csr.row_ptrs, world)

    def _make_local(self, csr, r0, r1):
        local = csr_row_slice(csr, r0, r1)
        self.local_nnz = local.nnz
        return ops.DeviceCSR.from_host(local, device=self.device)

    def __init__(self, csr, n_cols, device, kernel=0, acc="reference", bucket=16, compute=None, exchange="allgather",
                 debug_sentinel=False, batch=1):
        if exchange not in ("allgather", "peer"):
            raise ValueError(f"unknown exchange mode {exchange!r}")
        self.exchange = exchange
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.n = int(n_cols)
        self.kernel, self.acc = kernel, acc
        self.bucket = max(1, int(bucket))
        self.batch = max(1, int(batch))
        if self.batch > 16:
            raise ValueError("at most 16 dense operands per launch (mispmm_csr_batch_f32)")
        self.bucket = -(-self.bucket // self.batch) * self.batch     # whole launches per bucket
        self.num_rows, self.num_cols = csr.num_rows, csr.num_cols
        self.bounds = np.asarray(self._partition(csr, self.world), dtype=np.int64)
        self.r0, self.r1 = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        self.rows = self.r1 - self.r0
        self.slab_rows = max(1, int(np.diff(self.bounds).max()))
        self.a = self._make_local(csr, self.r0, self.r1)
        self.b = torch.zeros((self.num_cols, self.n), dtype=self.b_dtype, device=self.device)
        # two bucket-sized slab rings and gather targets: one being filled, one being gathered
        self.ring = [torch.zeros((self.bucket, self.slab_rows, self.n), dtype=torch.float32, device=self.device)
                     for _ in range(2)]
        self.gathered = [torch.zeros((self.world, self.bucket, self.slab_rows, self.n), dtype=torch.float32,
                                     device=self.device) for _ in range(2)]
        # kernel-only steps: one resident slab per operand of a batch (slab 0 for single launches)
        self.local_c = torch.zeros((self.batch, self.slab_rows, self.n), dtype=torch.float32, device=self.device)
        self.b_replicas = [self.b]    # batch > 1: filled by broadcast_b
        self.last_local = None        # "resident" after a kernel-only step, (buf, slot) after a step into the ring
        self.pending = [None, None]
        self.step_count = 0
        self.last = None   # (buffer index, slot) of the most recent gathered step
        self.compute = compute if compute is not None else self._hip_compute
        if self.on_gpu:
            self.compute_stream = torch.cuda.Stream(device=self.device)
            self.comm_stream = torch.cuda.Stream(device=self.device)
        else:
            self.compute_stream = self.comm_stream = None
        # One hipGraph per ring buffer holding a whole bucket of kernel launches (slots 0..bucket-1):
        # a step is a few microseconds, so launching each from Python would be host-bound.
        self.bucket_graphs = {}
        self.use_graphs = self.on_gpu and compute is None
        self.peer_dst = None
        self.debug_sentinel = bool(debug_sentinel) and exchange == "peer"
        self.sentinel_checks = 0
        if exchange == "peer":
            if not self.on_gpu:
                raise ValueError("peer exchange needs device buffers")
            self._map_peer_buffers()

    # -- peer exchange: map every rank's gather buffers once ------------------------------------------
    def _map_peer_buffers(self):
        from torch.multiprocessing.reductions import reduce_tensor
        # debug sentinels: per ring buffer one 16-byte slot per source rank (int32 x 4), written behind the slab scatter
        self.sentinel = [torch.zeros((self.world, 4), dtype=torch.int32, device=self.device) for _ in range(2)]
        self.seq_src = [torch.zeros(4, dtype=torch.int32, device=self.device) for _ in range(2)]
        self.seq = [0, 0]
        mine = [reduce_tensor(g) for g in self.gathered + self.sentinel]   # (rebuild function, picklable IPC arguments)
        everyone = [None] * self.world
        dist.all_gather_object(everyone, [m[1] for m in mine])
        rebuild = mine[0][0]
        self.peer_gathered, self.peer_sentinel = [], []
        for r in range(self.world):
            self.peer_gathered.append(self.gathered if r == self.rank else [rebuild(*everyone[r][b]) for b in (0, 1)])
            self.peer_sentinel.append(self.sentinel if r == self.rank else [rebuild(*everyone[r][2 + b]) for b in (0, 1)])
        ordinals = sorted({t.device.index for pg in self.peer_gathered for t in pg} | {self.device.index})
        if len(ordinals) > 1:
            arr = (ctypes.c_int * len(ordinals))(*ordinals)
            capi.check(capi.lib().mispmm_enable_peer_access(len(ordinals), arr))
        # destination pointers of this rank's slot in every rank's buffer, per ring buffer
        self.peer_dst = []
        for b in (0, 1):
            arr = (ctypes.c_void_p * self.world)()
            for r in range(self.world):
                arr[r] = self.peer_gathered[r][b][self.rank].data_ptr()
            self.peer_dst.append(arr)
        self.peer_sentinel_dst = []
        for b in (0, 1):
            arr = (ctypes.c_void_p * self.world)()
            for r in range(self.world):
                arr[r] = self.peer_sentinel[r][b][self.rank].data_ptr()
            self.peer_sentinel_dst.append(arr)
        self.flag = torch.zeros(1, dtype=torch.float32, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.barrier()                                            # nobody stores before everybody has mapped

    def _next_sequence(self, buf):
        """Debug sentinels: the number the NEXT scatter of ring buffer `buf` will leave in every peer (enqueued on the
        compute stream in front of that scatter; never captured into a bucket graph)."""
        self.seq[buf] += 1
        with torch.cuda.stream(self.compute_stream):
            self.seq_src[buf].fill_(self.seq[buf])

    def _scatter(self, buf, capturing=False):
        nbytes = self.ring[buf].numel() * 4
        sp = ctypes.c_void_p(self.compute_stream.cuda_stream)
        if self.debug_sentinel and not capturing:
            self._next_sequence(buf)
        capi.check(capi.lib().mispmm_slab_scatter(sp, ctypes.c_void_p(self.ring[buf].data_ptr()), nbytes, self.peer_dst[buf],
                                                  self.world))
        if self.debug_sentinel:   # behind the slabs on the same stream: the sentinel is the last thing a peer receives
            capi.check(capi.lib().mispmm_slab_scatter(sp, ctypes.c_void_p(self.seq_src[buf].data_ptr()), 16,
                                                      self.peer_sentinel_dst[buf], self.world))

    def _check_sentinels(self, buf):
        """After the wait for bucket `buf`: every peer's sentinel slot must already carry this bucket's sequence number."""
        got = self.sentinel[buf][:, 0].cpu().tolist()
        if any(g != self.seq[buf] for g in got):
            raise RuntimeError(f"peer exchange ordering violated on rank {self.rank}: ring buffer {buf} expected sequence "
                               f"{self.seq[buf]} from every rank, sentinels hold {got}")
        self.sentinel_checks += 1

    # -- the compute step ------------------------------------------------------------------------
    def _hip_compute(self, a, b, out):
        ops.spmm_csr(a, b, out=out, kernel=self.kernel, acc=self.acc, stream=self.compute_stream)

    # -- one-time B replication ------------------------------------------------------------------
    def _load_b(self, b_host):
        """rank 0: the fp32 host operand into this rank's device copy (the bf16 shards round it on the device)"""
        self.b.copy_(torch.from_numpy(np.ascontiguousarray(b_host, dtype=np.float32)))

    def broadcast_b(self, b_host):
        if self.rank == 0:
            self._load_b(b_host)
        wire = self.b.view(torch.uint8) if self.b.dtype == torch.int16 else self.b    # bf16 bit patterns travel as bytes (RCCL has no int16)
        if self.on_gpu and dist.get_backend() != "nccl":
            staged = wire.cpu()                   # a CPU-only backend (gloo rehearsal on one card): broadcast on the host
            dist.broadcast(staged, src=0)
            wire.copy_(staged)
        else:
            dist.broadcast(wire, src=0)
        self.b_replicas = [self.b] + [self.b.clone() for _ in range(self.batch - 1)]
        if self.on_gpu:
            torch.cuda.synchronize(self.device)

    # -- steady state ----------------------------------------------------------------------------
    def _gather(self, buf, scattered=False):
        if self.exchange == "peer":
            if not scattered:
                self._scatter(buf)
            if dist.get_backend() == "nccl":
                # every rank's scatter precedes its contribution on its own stream: the reduction completing on
                # this rank means every peer's stores into this rank's buffer have been issued AND completed
                done = torch.cuda.Event()
                done.record(self.compute_stream)
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(done)
                    self.pending[buf] = dist.all_reduce(self.flag, async_op=True)
            else:
                self.compute_stream.synchronize()
                dist.barrier()
                if self.debug_sentinel:
                    torch.cuda.synchronize(self.device)
                    self._check_sentinels(buf)
            return
        if self.on_gpu:
            done = torch.cuda.Event()
            done.record(self.compute_stream)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(done)
                self.pending[buf] = dist.all_gather_into_tensor(self._flat(buf), self.ring[buf], async_op=True)
        else:
            self.pending[buf] = dist.all_gather_into_tensor(self._flat(buf), self.ring[buf], async_op=True)

    def _flat(self, buf):
        # all_gather_into_tensor wants the output as the inputs concatenated along dim 0
        return self.gathered[buf].view(self.world * self.bucket, self.slab_rows, self.n)

    def _wait(self, buf, on_stream=None):
        w = self.pending[buf]
        if w is None:
            return
        if self.on_gpu and on_stream is not None:
            with torch.cuda.stream(on_stream):
                w.wait()
            if self.debug_sentinel:
                on_stream.synchronize()
        else:
            w.wait()
        self.pending[buf] = None
        if self.debug_sentinel:
            self._check_sentinels(buf)

    def _bucket_graph(self, buf, scatter, resident=False):
        key = ("resident", self.batch) if resident else (buf, bool(scatter))
        if key not in self.bucket_graphs:
            l = capi.lib()
            sp = ctypes.c_void_p(self.compute_stream.cuda_stream)
            self.compute_stream.synchronize()
            capi.check(l.mispmm_graph_begin(sp))
            if resident and self.batch > 1:
                outs = [self.local_c[i, :self.rows] for i in range(self.batch)]
                for _ in range(self.bucket // self.batch):
                    ops.spmm_csr_batch(self.a, self.b_replicas, outs=outs, acc=self.acc, stream=self.compute_stream)
            else:
                for slot in range(self.bucket):
                    self.compute(self.a, self.b, self.local_c[0, :self.rows] if resident else self.ring[buf][slot, :self.rows])
            if scatter:
                self._scatter(buf, capturing=True)  # the bucket's slabs leave for every peer inside the same graph
            g = ctypes.c_void_p()
            capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
            self.bucket_graphs[key] = g
        return self.bucket_graphs[key]

    def run(self, steps, gather=True):
        done = 0
        while done < steps:
            i = self.step_count
            buf, slot = (i // self.bucket) % 2, i % self.bucket
            if self.use_graphs and self.rows and slot == 0 and steps - done >= self.bucket:
                # a whole bucket at once: replay its graph, then hand the ring buffer to the collective
                self._wait(buf, self.compute_stream)
                if self.exchange == "peer" and gather:
                    self._wait(1 - buf, self.compute_stream)   # peers are done with the bucket before the last one
                scatter = gather and self.exchange == "peer"
                graph = self._bucket_graph(buf, scatter, resident=not gather)
                if scatter and self.debug_sentinel:
                    self._next_sequence(buf)
                capi.check(capi.lib().mispmm_graph_launch(graph,
                                                          ctypes.c_void_p(self.compute_stream.cuda_stream)))
                self.step_count += self.bucket
                done += self.bucket
                self.last_local = (buf, self.bucket - 1) if gather else "resident"
                if gather:
                    self._gather(buf, scattered=self.exchange == "peer")
                    self.last = (buf, self.bucket - 1)
                continue
            self._run_eager(1, gather)
            done += 1

    def _run_eager(self, steps, gather):
        for _ in range(steps):
            i = self.step_count
            buf, slot = (i // self.bucket) % 2, i % self.bucket
            if slot == 0:
                self._wait(buf, self.compute_stream)      # the gather that last read this ring is done
                if self.exchange == "peer" and gather:
                    self._wait(1 - buf, self.compute_stream)
            if self.rows:
                self.compute(self.a, self.b, self.ring[buf][slot, :self.rows] if gather else self.local_c[0, :self.rows])
            self.last_local = (buf, slot) if gather else "resident"
            self.step_count += 1
            if gather and slot == self.bucket - 1:
                self._gather(buf)
                self.last = (buf, slot)

    def finish(self, gather=True):
        """Gather a partially filled bucket and wait for every collective in flight."""
        i = self.step_count
        if gather and i % self.bucket:
            buf = (i // self.bucket) % 2
            self._gather(buf)
            self.last = (buf, (i - 1) % self.bucket)
            self.step_count = (i // self.bucket + 1) * self.bucket     # next run starts a fresh bucket
        for buf in (0, 1):
            self._wait(buf, self.comm_stream)
        if self.on_gpu:
            self.compute_stream.synchronize()
            self.comm_stream.synchronize()

    def close(self):
        for g in self.bucket_graphs.values():
            capi.check(capi.lib().mispmm_graph_destroy(g))
        self.bucket_graphs = {}

    # -- results ---------------------------------------------------------------------------------
    def gathered_c(self):
        """Full C [num_rows, n] of the most recent gathered step (every rank holds it)."""
        if self.last is None:
            raise RuntimeError("no gathered step yet")
        buf, slot = self.last
        parts = [self.gathered[buf][r, slot, :int(self.bounds[r + 1] - self.bounds[r])] for r in range(self.world)]
        return torch.cat(parts, dim=0)

    def local_slab(self, operand=0):
        """This rank's C rows of the most recent step (kernel-only steps: of operand `operand` of the last launch)."""
        if self.last_local is None:
            raise RuntimeError("no step yet")
        if self.last_local == "resident":
            return self.local_c[operand, :self.rows]
        buf, s = self.last_local
        return self.ring[buf][s, :self.rows]


class ShardedEllSpmm(ShardedCsrSpmm):
    """ELL sharded by rows: `ell` is a formats.ELLRowMajor (or the reference's column-major ELL,
    converted once); rows are cut into contiguous ranges balanced by occupied slots; every rank multiplies its rows with
    mispmm_ell_f32 -- the reference's ELL arithmetic (fp32 product, fp32 add in slot order), so the gathered C equals the
    unsharded ELL product bit for bit."""

    def __init__(self, ell, n_cols, device, kernel=0, acc="reference", bucket=16, compute=None, exchange="allgather", debug_sentinel=False):
        if isinstance(ell, formats.ELLColMajor):
            ell = ops.colmajor_ell_to_rowmajor(ell)
        super().__init__(ell, n_cols, device, kernel=kernel, acc=acc, bucket=bucket, compute=compute, exchange=exchange,
                         debug_sentinel=debug_sentinel, batch=1)

    @staticmethod
    def _partition(ell, world):
        cols = np.asarray(ell.col_idxs, dtype=np.uint32).reshape(ell.num_rows, -1)
        occupied = np.concatenate([[0], np.cumsum((cols != 0xFFFFFFFF).sum(axis=1))]).astype(np.uint32)
        return shard_bounds(occupied, world)

    def _make_local(self, ell, r0, r1):
        cols = np.asarray(ell.col_idxs, dtype=np.uint32).reshape(ell.num_rows, -1)[r0:r1]
        vals = np.asarray(ell.data, dtype=np.float32).reshape(ell.num_rows, -1)[r0:r1]
        self.local_nnz = int((cols != 0xFFFFFFFF).sum())
        local = formats.ELLRowMajor(r1 - r0, ell.num_cols, self.local_nnz, ell.width, cols, vals)
        return ops.DeviceELL.from_host(local, device=self.device, compact=False) if self.device.type == "cuda" else local

    def _hip_compute(self, a, b, out):
        ops.spmm_ell(a, b, out=out, kernel=self.kernel, acc=self.acc, stream=self.compute_stream)


class ShardedBsrcSlotsSpmm(ShardedCsrSpmm):
    """bf16 BSR-16 sharded by BLOCK rows (BASELINE config 4's kernel on every shard): block rows are
    cut into ranges balanced by block count, every shard is compacted into fixed step slots on its own (the same columns per
    block row as the unsharded compaction, so the gathered C equals the unsharded kernel's C bit for bit).  B is rounded to
    bf16 once on rank 0 and broadcast as bf16; C is fp32."""
    b_dtype = torch.int16

    def __init__(self, bsr, n_cols, device, bucket=16, compute=None, exchange="allgather", debug_sentinel=False):
        if bsr.block_row_size != 16 or bsr.block_col_size != 16:
            raise ValueError("16 x 16 blocks (BASELINE config 4)")
        super().__init__(bsr, n_cols, device, kernel=0, acc="reference", bucket=bucket, compute=compute, exchange=exchange,
                         debug_sentinel=debug_sentinel, batch=1)

    @staticmethod
    def _partition(bsr, world):
        return shard_bounds(bsr.block_row_ptrs, world) * 16              # C rows

    def _make_local(self, bsr, r0, r1):
        from .multi import bsr_block_row_slice
        local = bsr_block_row_slice(bsr, r0 // 16, r1 // 16)
        self.local_nnz = int(local.num_blocks) * 256
        return ops.DeviceBSRCSlots.from_host(local, device=self.device) if self.device.type == "cuda" else local

    def _load_b(self, b_host):
        self.b.copy_(ops.f32_to_bf16(torch.from_numpy(np.ascontiguousarray(b_host, dtype=np.float32)).to(self.device)))

    def _hip_compute(self, a, b, out):
        ops.spmm_bsrc_slots_bf16(a, b, out_bf16=False, out=out, stream=self.compute_stream)

