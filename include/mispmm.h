/*
 * mispmm.h -- C ABI of libmispmm.so, the MI355X (gfx950) SpMM engine.
 *
 * This is the drop-in boundary for the reference's SpMM hot path
 * (mli43/Cuda-Optimization-for-SpMM @ 2024-12-18).  The reference has no FFI:
 * its engine calls C++ templates `spmm<F>Wrapper<N>(a, b, ref)` that launch CUDA
 * kernels on raw device pointers held by its format classes.  Each entry point
 * below is what such a wrapper body binds to (INTEGRATION.md shows the binding);
 * the reference interface it replaces is cited per function as file:line under
 * /root/reference.
 *
 * Conventions
 *   - plain C: pointers, sizes, ints.  No HIP, torch or C++ types.
 *   - every `const T *` / `T *` operand of a compute call is a DEVICE pointer
 *     (hipMalloc'd or any memory the GPU can address) unless the name ends in
 *     `_host`.  Index type is uint32_t (the reference's MT), values float
 *     (its DT) -- the only instantiation the reference's kernels have
 *     (e.g. src/spmm/csr/spmm_csr_k1.cu:86).
 *   - dense operands are ROW-major with a leading dimension in elements
 *     (ldb >= N, ldc >= N).  Column-major B (the layout the reference's K2/K4
 *     ask DenseMatrix::toOrdering for, src/formats/dense.cu:139-191) is
 *     converted on the device with mispmm_dense_transpose_f32.
 *   - `stream` is a hipStream_t passed as void*; NULL is the null stream.
 *     Compute calls only ENQUEUE work: no allocation, no synchronisation, so
 *     they can be captured into a hipGraph (mispmm_graph_*).
 *   - return value: MISPMM_OK (0) or a negative mispmm_status; the library
 *     never calls exit().  mispmm_last_error() gives the detail string of the
 *     calling thread's last failure.  (The reference prints and exit()s:
 *     include/cuda_utils.hpp:13-22 -- the host layer above this ABI keeps that
 *     CLI behaviour.)
 *   - C is overwritten (beta = 0); it does not need to be zeroed first.
 */
#ifndef MISPMM_H
#define MISPMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISPMM_VERSION 100 /* 0.1.0 */

typedef void *mispmm_stream_t; /* hipStream_t */
typedef void *mispmm_event_t;  /* hipEvent_t */
typedef void *mispmm_graph_t;  /* hipGraphExec_t */

enum mispmm_status {
    MISPMM_OK = 0,
    MISPMM_ERR_INVALID_ARG = -1, /* null pointer, bad leading dimension, ... */
    MISPMM_ERR_UNSUPPORTED = -2, /* this kernel id declines the shape (cf. spmm_csr_k4.cu:97-101) */
    MISPMM_ERR_HIP = -3,         /* a HIP runtime call failed */
    MISPMM_ERR_NO_DEVICE = -4,
    MISPMM_ERR_ALLOC = -5
};

/* How products are accumulated.
 *  REFERENCE: the rounding sequence of the reference's sequential CPU engine, so
 *    results are bit-identical to it:
 *      CSR  fp32 product, widened and summed in a double accumulator in storage
 *           order, rounded to fp32 once (src/spmm/csr/spmm_csr.cpp:20-25,
 *           AccT = double from src/main.cu:196);
 *      COO / ELL / BSR  fp32 product then fp32 add into C, in storage order
 *           (spmm_coo.cpp:16-24, spmm_ell.cpp:16-29, spmm_bsr.cpp:17-38).
 *  FAST: fp32 fused multiply-add chains (<= 1e-5 relative to sum|a||b| of the
 *    reference result; the MFMA kernels always use this).  One chain per output
 *    element in storage order, except CSR kernel 6, which adds 8 chains per row
 *    (per chunk of a shared row) in a fixed order: run to run identical, but
 *    not bit for bit what another kernel id or entry point returns. */
enum mispmm_acc_mode { MISPMM_ACC_REFERENCE = 0, MISPMM_ACC_FAST = 1 };

/* Kernel selector common to all formats: 0 lets the library choose. */
#define MISPMM_KERNEL_AUTO 0

/* ------------------------------------------------------------------ runtime */
int mispmm_version(void);
const char *mispmm_status_string(int status);
const char *mispmm_last_error(void);
/* Tag of the device kernel (template instance + XCD tiling) the calling thread's last compute call
 * enqueued, e.g. "row_gather<G16,V4,ref64,uniform,B128,U8,roll,S14> xcd 4x2".  Diagnostic: lets a
 * measurement tie a rocprofv3 counter figure to the kernel that actually ran. */
const char *mispmm_last_kernel(void);

/* replaces cudaSetDevice(7) (src/main.cu:176) */
int mispmm_device_count(int *count);
int mispmm_set_device(int ordinal);
int mispmm_get_device(int *ordinal);
/* name buffer >= 256 bytes; cu_count / hbm_bytes may be NULL */
int mispmm_device_info(int ordinal, char *name, int *cu_count, size_t *hbm_bytes);
/* PCI bus id of a device ("0000:c1:00.0"): lets a multi-process run prove that its ranks sit on distinct devices */
int mispmm_device_bus_id(int ordinal, char *bus_id, int len);

/* replace cudaMalloc+cudaMemset / cudaMallocHost / cudaFree / cudaFreeHost as
 * the format classes use them (src/formats/dense.cu:234-262): device and pinned
 * host allocations come back zero-filled. */
int mispmm_malloc(void **dev_ptr, size_t bytes);
int mispmm_free(void *dev_ptr);
int mispmm_host_alloc(void **host_ptr, size_t bytes);
int mispmm_host_free(void *host_ptr);

enum mispmm_copy_kind { MISPMM_H2H = 0, MISPMM_H2D = 1, MISPMM_D2H = 2, MISPMM_D2D = 3 };
/* blocking copy, as cudaMemcpy in copy2Device/copy2Host (dense.cu:92-137) */
int mispmm_memcpy(void *dst, const void *src, size_t bytes, int kind);
int mispmm_memcpy_async(void *dst, const void *src, size_t bytes, int kind, mispmm_stream_t stream);
int mispmm_memset_async(void *dev_ptr, int value, size_t bytes, mispmm_stream_t stream);

int mispmm_stream_create(mispmm_stream_t *stream);
int mispmm_stream_destroy(mispmm_stream_t stream);
int mispmm_stream_sync(mispmm_stream_t stream);
int mispmm_device_sync(void); /* cudaDeviceSynchronize after each launch, e.g. spmm_csr_k3.cu:83 */

int mispmm_event_create(mispmm_event_t *event);
int mispmm_event_destroy(mispmm_event_t event);
int mispmm_event_record(mispmm_event_t event, mispmm_stream_t stream);
int mispmm_event_sync(mispmm_event_t event);
int mispmm_event_elapsed_ms(mispmm_event_t start, mispmm_event_t stop, float *ms);

/* Capture everything enqueued on `stream` between begin and end into an
 * executable graph; replay it with mispmm_graph_launch (launch-bound loops). */
int mispmm_graph_begin(mispmm_stream_t stream);
int mispmm_graph_end(mispmm_stream_t stream, mispmm_graph_t *graph);
int mispmm_graph_launch(mispmm_graph_t graph, mispmm_stream_t stream);
int mispmm_graph_destroy(mispmm_graph_t graph);

/* -------------------------------------------------------------- CSR x dense */
/* C[M x N] = A_csr[M x K] * B[K x N].
 * Replaces the bodies of spmmCSRWrapper1..4 (src/spmm/csr/spmm_csr_k1.cu:36-84,
 * _k2.cu:60-107, _k3.cu:58-105, _k4.cu:81-142) and their kernels.
 * kernel: 0 auto; 1 wave-per-row, lane-shuffle broadcast of (col,val);
 *         2 row-block workgroup, (col,val) staged through LDS;
 *         3 wave-per-row, (col,val) on the scalar path, B rows by SGPR base;
 *         4 as 3 with two rows in flight per wave;
 *         5 as 1 with a 2-D (row part x column part) XCD tiling and write-through C stores (auto: rows of
 *           24 entries or more on average take kernel 6);
 *         6 one workgroup per row, the row's entries split over its lane groups (long or very uneven rows).
 *           REFERENCE mode stays bit-exact: an output element is summed in split order only where that
 *           provably cannot round differently, else sequentially (csrc/csr_split.hpp).
 * Any kernel id handles any M, K, nnz, N, ragged and empty rows. */
int mispmm_csr_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                   const uint32_t *colIdxs, const float *vals, const float *B, uint32_t N, uint32_t ldb, float *C,
                   uint32_t ldc, int kernel, int acc_mode);
#define MISPMM_CSR_NUM_KERNELS 6

/* Structure hint: a CSR whose rows ALL hold exactly rowNnz entries (rowPtrs[r] == r * rowNnz, e.g. the
 * headline matrix n4c6-b13 with 14, or any ELL-shaped CSR).  Same arithmetic and results as
 * mispmm_csr_f32 (kernel 5), but the row pointer array is never read: one dependent memory hop less
 * per wave.  The caller vouches for the structure (the host layers check it once when A is copied to
 * the device, an O(M) scan).  Returns MISPMM_ERR_UNSUPPORTED for a B of 2 GiB or more. */
int mispmm_csr_uniform_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t rowNnz, const uint32_t *colIdxs,
                           const float *vals, const float *B, uint32_t N, uint32_t ldb, float *C, uint32_t ldc,
                           int acc_mode);

/* Kernel 6 walking a SPAN LIST instead of the rows in order.  spans (device, 4 * numSpans uint32, 16-byte aligned)
 * is what mispmm_csr_spans_by_length_host builds once per matrix (an O(M + longest) counting sort; the host layers do it
 * when A is copied to the device): the rows as (row, rowPtrs[row], rowPtrs[row + 1], 0) sorted by decreasing length,
 * preceded by the rows of more than share_len entries (0 = 128) as 4 chunks (row, start, end, 1) each, one aligned group
 * of 4 positions per row -- the 4 waves of one workgroup sum one chunk each and add the four in entry order.
 * Same arithmetic and results as mispmm_csr_f32 with kernel 6 (REFERENCE mode: the reference's bits); the list only
 * decides WHEN and by how many waves a row is summed -- a matrix sorted by row length (GL7d25: every row of more
 * than 128 entries among the last 93 of 2798) otherwise starts its long rows last, and its longest row alone is as long
 * as the rest of the kernel.  spans may be NULL (rows in order; rowPtrs needed, numSpans ignored); rowPtrs may be NULL
 * when spans is given.  The caller vouches that the list describes this matrix.
 * Returns MISPMM_ERR_UNSUPPORTED (nothing launched) when B or C rows are not 16-byte vectors or B spans 2 GiB or
 * more: use mispmm_csr_f32.  New capability: the reference has no analysis phase.
 * mispmm_csr_spans_by_length_host: *count_out = number of spans (M + 3 per shared row); spans_out_host NULL = size query. */
int mispmm_csr_split_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                         const uint32_t *colIdxs, const float *vals, const uint32_t *spans, uint32_t numSpans, const float *B,
                         uint32_t N, uint32_t ldb, float *C, uint32_t ldc, int acc_mode);
int mispmm_csr_spans_by_length_host(uint32_t M, const uint32_t *rowPtrs_host, uint32_t share_len, uint32_t *count_out,
                                    uint32_t *spans_out_host);

/* The same span list in ONE launch by two bodies (csr_hybrid.hpp): span positions [0, numLongSpans) -- the long rows --
 * through the split kernel's body, the other positions -- the short rows, the tail of the list -- through the row-gather
 * body (a lane group per row, the row's entries and its row of C named by its span).  On GL7d25 the split kernel spends
 * 7.0 us where its 437 rows of more than 32 entries alone need 4.9 and the 2361 others 3.7 through the row-gather kernel
 * (tools/probe/hybrid_longrows_probe.py); two launches would pay the launch boundary twice.  Same results as
 * mispmm_csr_split_f32 in REFERENCE mode (the reference's bits: the row-gather body adds a row in entry order), FAST
 * within its bound.  numLongSpans: a multiple of 4 that covers every 4-chunk group of the list; rows whose span lies
 * behind it must be short enough for a lane group (mispmm_csr_spans_long_count_host: the first position whose row has
 * at most `threshold` entries, rounded up to 4; the host layers use 32).
 * Returns MISPMM_ERR_UNSUPPORTED, without a message and with nothing launched, for shapes without such a launch (more
 * than 256 columns, column parts that are not one lane group wide, rows that are not 16-byte vectors, B or C of 2 GiB or
 * more, no short rows): the caller takes mispmm_csr_split_f32.  New capability (no reference counterpart). */
int mispmm_csr_hybrid_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *colIdxs, const float *vals,
                          const uint32_t *spans, uint32_t numSpans, uint32_t numLongSpans, const float *B, uint32_t N, uint32_t ldb,
                          float *C, uint32_t ldc, int acc_mode);
int mispmm_csr_spans_long_count_host(uint32_t numSpans, const uint32_t *spans_host, uint32_t threshold, uint32_t *numLong_out);

/* Plan order: a CSR whose rows were PERMUTED once at upload so that rows which read the same B rows sit together --
 * the row-gather kernel gives each XCD a contiguous range of array rows and walks it in order, so a B row fetched for
 * one row of a cluster is still in that XCD's L2 for the others (n4c6-b13 x K=512, where a 6.5 MB B slice per XCD
 * competes for 4 MiB of L2: 13.6 -> 13.0 us; where the slice fits, the scattered C rows make it a loss: K=128 +6 %).  Array row i produces row rowMap[i] of C; every row keeps its entries in
 * storage order, so results are bit-identical to mispmm_csr_f32 on the unpermuted arrays.
 *   mispmm_csr_cluster_rows_host   greedy clustering into `parts` equal clusters (the format objects use 8);
 *                                  order_out[i] = original row at position i; *natural / *clustered_distinct_out = sum
 *                                  over the parts of the distinct columns a part touches, before and after (the caller
 *                                  keeps the plan only if the figure drops)
 *   mispmm_csr_permute_rows_host   the permuted arrays (rowPtrs_out[M + 1], colIdxs_out[nnz], vals_out[nnz])
 *   mispmm_csr_plan_f32            C_list[i] = A * B_list[i], i < batch (HOST arrays of device pointers, as for
 *                                  mispmm_csr_batch_f32; batch = 1 for a single product); uniformRowNnz > 0 = every
 *                                  row holds that many entries (rowPtrs may be NULL); rowMap NULL = identity.
 * MISPMM_ERR_UNSUPPORTED for a B of 2 GiB or more (multiply from the unpermuted arrays).  New capability: the
 * reference has no analysis phase (src/formats/sparse_csr.cu copies the file's arrays as they are). */
int mispmm_csr_cluster_rows_host(uint32_t M, uint32_t K, const uint32_t *rowPtrs_host, const uint32_t *colIdxs_host,
                                 uint32_t parts, uint32_t *order_out_host, uint64_t *natural_distinct_out,
                                 uint64_t *clustered_distinct_out);
int mispmm_csr_permute_rows_host(uint32_t M, const uint32_t *rowPtrs_host, const uint32_t *colIdxs_host, const float *vals_host,
                                 const uint32_t *order_host, uint32_t *rowPtrs_out_host, uint32_t *colIdxs_out_host,
                                 float *vals_out_host);
int mispmm_csr_plan_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                        const uint32_t *colIdxs, const float *vals, uint32_t uniformRowNnz, const uint32_t *rowMap,
                        uint32_t batch, const float *const *B_list_host, uint32_t N, uint32_t ldb, float *const *C_list_host,
                        uint32_t ldc, int acc_mode);

/* Autotune (round 4): whether a product of THIS matrix with a dense operand of N columns is faster from the plan-order arrays
 * or from the storage-order ones is MEASURED, once per (matrix, N), instead of guessed from a footprint rule (the rule the
 * measurements of round 3 gave -- plan where the B slice of an XCD exceeds its L2 -- misses e.g. ACTIVSg10K x N = 256:
 * 12.6 -> 11.1 us with the plan).  Scratch B (ones) and C are allocated, each candidate is warmed and then timed over 3 rounds
 * of `launches` plain launches (0 = 48) between HIP events on `stream` (kernels of a few microseconds run at the host's launch
 * rate this way and tie, which keeps the default: the plan only wins on products of 10 us and more), the scratch is freed; times_us_out[0] = storage
 * order (mispmm_csr_uniform_f32 / mispmm_csr_f32 kernel auto), [1] = plan order, a candidate the shape does not support = +inf.
 * *use_plan_out = mispmm_autotune_pick(times, 2, 0.02) == 1.  Synchronises `stream`; never call it while the stream is being
 * captured.  The format objects (SparseMatrixCSR::copy2Device's caller, DeviceCSR) call it on the first product of a width and
 * remember the answer.  New capability: the reference's only shape rule is K4's shared-memory guard (spmm_csr_k4.cu:97-101).
 *   mispmm_autotune_pick   the decision alone, a pure function of the timings (so that it can be tested without a GPU): the
 *                          index of the smallest time, except that candidate 0 (the default) is kept unless another is at
 *                          least `min_gain` (e.g. 0.02 = 2 %) faster; non-finite / non-positive times never win; -1 if n = 0. */
int mispmm_csr_autotune_plan_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                                 const uint32_t *colIdxs, const float *vals, uint32_t uniformRowNnz, const uint32_t *planRowPtrs,
                                 const uint32_t *planColIdxs, const float *planVals, const uint32_t *planRowMap, uint32_t N,
                                 int acc_mode, uint32_t launches, int *use_plan_out, float *times_us_out);
int mispmm_autotune_pick(const float *times_us, uint32_t n, float min_gain);

/* LDS tiles (round 4): the kernel shape BASELINE.json's north_star names for the row-parallel kernels -- the B rows of a GROUP
 * of rows staged once into LDS, every row of the group summing from there (ancestor: the shared-memory staging of
 * src/spmm/csr/spmm_csr_k4.cu:26-62).  Built to have the number; rows of ONE width of <= 16 entries.
 *   mispmm_csr_tiles_host   HOST, once per upload: greedy grouping into tiles of <= maxRows (<= 16) rows whose DISTINCT columns
 *                           number <= maxCols (the kernel's LDS budget: 128).  Outputs NULL = size query.  tileRowPtrs[T + 1] /
 *                           tileColPtrs[T + 1]: a tile's plan rows and its column list in tileCols; order[i] = the row at plan
 *                           position i (= the kernel's rowMap; permute vals with mispmm_csr_permute_rows_host); slots[nnz]: per
 *                           entry, in plan order, the position of its column in its tile's list.
 *   mispmm_csr_lds_tile_f32 C = A * B from those arrays; REFERENCE mode bit-exact (a row keeps its entries in storage order).
 *                           MISPMM_ERR_UNSUPPORTED for other shapes (column parts that are not whole 64-column groups, ...). */
int mispmm_csr_tiles_host(uint32_t M, uint32_t K, const uint32_t *rowPtrs_host, const uint32_t *colIdxs_host, uint32_t maxRows,
                          uint32_t maxCols, uint32_t *numTiles_out, uint32_t *numListed_out, uint32_t *tileRowPtrs_out_host,
                          uint32_t *tileColPtrs_out_host, uint32_t *tileCols_out_host, uint32_t *order_out_host, uint8_t *slots_out_host);
int mispmm_csr_lds_tile_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t rowNnz, uint32_t numTiles,
                            const uint32_t *tileRowPtrs, const uint32_t *tileColPtrs, const uint32_t *tileCols, const uint8_t *slots,
                            const float *vals, const uint32_t *rowMap, const float *B, uint32_t N, uint32_t ldb, float *C, uint32_t ldc,
                            int acc_mode);

/* Several products with the same A in ONE launch: C_list[i] = A * B_list[i], i < batch (HOST arrays of device
 * pointers; every operand N columns wide with leading dimensions ldb / ldc).  Same arithmetic and results as
 * `batch` calls of mispmm_csr_f32 (kernel 5) / mispmm_csr_uniform_f32 (uniformRowNnz > 0: rowPtrs may be NULL),
 * but the time between two dependent launches -- about 1.2 us on MI355X, a third of a headline-sized product --
 * is paid once per 16 operands.  New capability: the reference multiplies by one dense.in per process
 * (src/main.cu:185).  Shapes the batched kernel does not take are issued as one launch each. */
int mispmm_csr_batch_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                         const uint32_t *colIdxs, const float *vals, uint32_t uniformRowNnz, uint32_t batch,
                         const float *const *B_list_host, uint32_t N, uint32_t ldb, float *const *C_list_host, uint32_t ldc,
                         int acc_mode);

/* -------------------------------------------------------------- ELL x dense */
/* Row-major ELL: colIdxs/vals are [M x width], padding index 0xFFFFFFFF.
 * Replaces spmmELLWrapper1/2 (src/spmm/ell/spmm_ell_k1.cu:38-63,
 * spmm_ell_k2.cu:57-83), which scatter a column-major ELL with atomics; the host
 * layer converts the reference's column-major arrays once with
 * mispmm_ell_colmajor_to_rowmajor_host.  kernel: 0 auto, 1 row-group gather. */
int mispmm_ell_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t width, const uint32_t *colIdxs,
                   const float *vals, const float *B, uint32_t N, uint32_t ldb, float *C, uint32_t ldc, int kernel,
                   int acc_mode);
#define MISPMM_ELL_NUM_KERNELS 1

/* HOST helper.  rowIdxs_host/vals_host: the reference's SparseMatrixELL arrays,
 * [numCols x maxColNnz] (src/formats/sparse_ell.cu:36-47).  Call once with
 * colIdxs_out_host == NULL to get *width_out (longest row), then with buffers of
 * numRows * width.  Entries of a row keep ascending column, then slot order --
 * the order spmmELLCpu accumulates them in (spmm_ell.cpp:16-29). */
int mispmm_ell_colmajor_to_rowmajor_host(uint32_t numRows, uint32_t numCols, uint32_t maxColNnz,
                                         const uint32_t *rowIdxs_host, const float *vals_host, uint32_t *width_out,
                                         uint32_t *colIdxs_out_host, float *vals_out_host);

/* -------------------------------------------------------------- BSR x dense */
/* blocks: numBlocks x (bR x bC) row-major, block-CSR order, any block-column
 * order inside a block row.  M = numBlockRows * bR.
 * Replaces spmmBSRWrapper1 (src/spmm/bsr/spmm_bsr_k1.cu:44-91).
 * kernel: 0 auto; 1 VALU, any bR/bC (REFERENCE or FAST accumulate);
 *         2 fp32-input MFMA v_mfma_f32_16x16x4_f32, bR = bC = 16 (FAST numerics). */
int mispmm_bsr_f32(mispmm_stream_t stream, uint32_t numBlockRows, uint32_t K, uint32_t bR, uint32_t bC,
                   uint32_t numBlocks, const uint32_t *blockRowPtrs, const uint32_t *blockColIdxs,
                   const float *blocks, const float *B, uint32_t N, uint32_t ldb, float *C, uint32_t ldc, int kernel,
                   int acc_mode);
#define MISPMM_BSR_NUM_KERNELS 2

/* ELL without its padding.  An ELL is as wide as its longest row: GL7d25 (mean 29, longest 422 entries) is 93 %
 * padding, ACTIVSg10K 81 %, and the padded kernel walks every slot.  The host helper lists the occupied slots of the
 * row-major view as (rowPtrs[M+1], colIdxs, vals) in slot order -- the order spmmELLCpu adds them; a padding slot adds
 * nothing there (`if (row >= 0)`, spmm_ell.cpp:21), so leaving it out changes no sum: call it with the three outputs
 * NULL for *nnz_out, then with arrays of that size, once per upload (the host layers do when more than half of the
 * slots are padding).  The device call multiplies from that list with ELL's arithmetic (REFERENCE = fp32 product, fp32
 * add in list order; same bits as mispmm_ell_f32); rows of 24 entries or more on average take the split kernel's shape
 * with ordered sums.  Returns MISPMM_ERR_UNSUPPORTED for a B of 2 GiB or more: use mispmm_ell_f32. */
int mispmm_ell_compact_host(uint32_t M, uint32_t width, const uint32_t *rmColIdxs_host, const float *rmVals_host, uint32_t *nnz_out,
                            uint32_t *rowPtrs_out_host, uint32_t *colIdxs_out_host, float *vals_out_host);
int mispmm_ell_compact_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                           const uint32_t *colIdxs, const float *vals, const float *B, uint32_t N, uint32_t ldb, float *C,
                           uint32_t ldc, int acc_mode);

/* Rows with the fp32 arithmetic of COO / ELL / BSR (fp32 product, fp32 add in list order) on the split kernel's shape,
 * walked LONGEST FIRST.  spans (device, 4 * M uint32, 16-byte aligned) = mispmm_csr_spans_by_length_host with
 * share_len = 0xFFFFFFFF: one (row, start, end, 0) per row -- a sum of this arithmetic cannot be dealt to several
 * waves -- built once per upload from the row pointers of a sorted COO, of the BSR non-zero list or of the compact ELL.
 * Same bits as mispmm_coo_f32 / mispmm_ell_compact_f32 / mispmm_bsr_nonzeros_f32; the row boundaries are in the spans,
 * so a COO needs no boundary pass.  MISPMM_ERR_UNSUPPORTED when B or C rows are not 16-byte vectors or B spans
 * 2 GiB or more: use the format's own entry point. */
int mispmm_rows_split_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *colIdxs, const float *vals,
                          const uint32_t *spans, uint32_t numSpans, const float *B, uint32_t N, uint32_t ldb, float *C,
                          uint32_t ldc, int acc_mode);
/* mispmm_rows_split_f32's list in one launch by two bodies, as mispmm_csr_hybrid_f32 does for CSR: span positions
 * [0, numLongSpans) on the split kernel's shape, the others by lane groups of the row-gather body -- the same fp32
 * product and fp32 add in list order either way, so the same bits.  numLongSpans from
 * mispmm_csr_spans_long_count_host (a multiple of 4).  MISPMM_ERR_UNSUPPORTED (no message, nothing launched) for shapes
 * without such a launch: take mispmm_rows_split_f32. */
int mispmm_rows_hybrid_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *colIdxs, const float *vals,
                           const uint32_t *spans, uint32_t numSpans, uint32_t numLongSpans, const float *B, uint32_t N, uint32_t ldb,
                           float *C, uint32_t ldc, int acc_mode);

/* Zero-skipping BSR.  At BSR-16 the SuiteSparse matrices of data/ are ~98 % explicit zeros (ACTIVSg10K: 33 100 blocks
 * for 137 736 non-zeros): the dense block arithmetic of mispmm_bsr_f32 kernel 1 spends 84 us where the non-zeros
 * need 7.  The host helper lists the block entries that are not zero as (rowPtrs[M+1], colIdxs, vals), per C row in
 * the order spmmBSRCpu adds them (blocks in storage order, ascending column inside a block, spmm_bsr.cpp:17-38): call
 * it with the three outputs NULL for *nnz_out, then with arrays of that size, once per upload.  The device call
 * multiplies from that list; REFERENCE = fp32 product, fp32 add in list order.
 * NUMERICS: identical bits to spmmBSRCpu whenever no explicit zero of A meets an Inf or NaN of B -- a skipped term
 * 0 * b adds +-0, which never changes a sum that started at +0; but 0 * Inf = NaN, which the reference (and kernel 1)
 * propagate and this path does not.  That is why it is a separate entry point and not the default. */
int mispmm_bsr_nonzeros_host(uint32_t numBlockRows, uint32_t bR, uint32_t bC, uint32_t numBlocks,
                             const uint32_t *blockRowPtrs_host, const uint32_t *blockColIdxs_host, const float *blocks_host,
                             uint32_t *nnz_out, uint32_t *rowPtrs_out_host, uint32_t *colIdxs_out_host, float *vals_out_host);
int mispmm_bsr_nonzeros_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                            const uint32_t *colIdxs, const float *vals, const float *B, uint32_t N, uint32_t ldb, float *C,
                            uint32_t ldc, int acc_mode);

/* bf16 blocks and B (raw bf16 bit patterns), fp32 accumulate on
 * v_mfma_f32_16x16x32_bf16; bR = bC = 16 (two blocks per instruction) or 32.  C is fp32 (c_bf16 = 0) or bf16.
 * New capability (BASELINE.json config 4); the reference has no bf16 path. */
int mispmm_bsr_bf16(mispmm_stream_t stream, uint32_t numBlockRows, uint32_t K, uint32_t bR, uint32_t bC,
                    uint32_t numBlocks, const uint32_t *blockRowPtrs, const uint32_t *blockColIdxs,
                    const uint16_t *blocks, const uint16_t *B, uint32_t N, uint32_t ldb, void *C, uint32_t ldc,
                    int c_bf16);

/* Column-compacted block rows for the bf16 MFMA path.  A 16-row block row touches few of the columns its blocks span
 * (ACTIVSg10K at 16 x 16: 71 of 424 on average), so instead of one B panel per block it keeps, per block row, the
 * list of columns that hold a non-zero (padded with 0xFFFFFFFF to a multiple of 32) and its values gathered to those
 * columns as bf16 tiles [16 rows][32 k]: one v_mfma_f32_16x16x32_bf16 step per 32 occupied columns.
 * HOST helper (once per upload; bR must be 16, values rounded to bf16 RNE here): outputs NULL = size query for
 * *nSteps_out; then stepPtrs[numBlockRows + 1], cols[nSteps * 32], tiles[nSteps * 512] (raw bf16 bits).
 * Device call: B and C as for mispmm_bsr_bf16 (bf16 bits; C fp32 or bf16), N / ldb / ldc multiples of 8.
 * Same results as mispmm_bsr_bf16 up to the order of the fp32 sums; like the zero-skipping fp32 path it never forms
 * 0 * b for columns a block row does not occupy.  New capability (BASELINE.json config 4); the reference has no bf16. */
int mispmm_bsr_compact_bf16_host(uint32_t numBlockRows, uint32_t bR, uint32_t bC, uint32_t numBlocks,
                                 const uint32_t *blockRowPtrs_host, const uint32_t *blockColIdxs_host, const float *blocks_host,
                                 uint32_t *nSteps_out, uint32_t *stepPtrs_out_host, uint32_t *cols_out_host,
                                 uint16_t *tiles_out_host);
int mispmm_bsrc_bf16(mispmm_stream_t stream, uint32_t numBlockRows, uint32_t K, uint32_t nSteps, const uint32_t *stepPtrs,
                     const uint32_t *cols, const uint16_t *tiles, const uint16_t *B, uint32_t N, uint32_t ldb, void *C, uint32_t ldc,
                     int c_bf16);

/* The same compacted operand laid out for ONE WORKGROUP per (block row, 128 output columns): a block row's first 4 MFMA K
 * steps sit in fixed slots -- step R * 4 + w belongs to wave w of block row R, slots past the row's step count hold a
 * padding column list -- so no pointer is read in front of a column list and the steps of a block row run on 4 waves
 * at once; steps past the fourth ("extra" steps, block rows with more than 128 occupied columns) follow the slots at
 * index 4 * numBlockRows + extraPtrs[R] .. extraPtrs[R + 1].  The 4 partial tiles are added in wave order through LDS
 * (fixed order: deterministic); where no block row has extra steps and C is fp32 (config 4) each of the 4 waves adds and
 * stores 4 of the block row's 16 rows, else wave 0 does both -- the same bits either way.  mispmm_bsrc_bf16 walks the same steps
 * with one wave per block row: 3 dependent memory hops per step on 1250 waves (config 4: 6.2 us); this layout needs 2 hops
 * and runs 5000 waves (4.5 us).
 * HOST helper (once per upload): outputs NULL = size query; *nSteps_out = 4 * numBlockRows + extra steps (array
 * extents: cols[nSteps * 32], tiles[nSteps * 512]), *nUsedSteps_out (may be NULL) = steps that hold values (what the
 * kernel reads in full: an empty slot costs its 128-byte column list only); extraPtrs[numBlockRows + 1].
 * Device call: operands as for mispmm_bsrc_bf16.  nSteps is the extent of cols / tiles, so nSteps == 4 * numBlockRows
 * says there are NO extra steps: extraPtrs must then be all zeros (as the helper writes it) and is not read; with extra
 * steps every extraPtrs[R + 1] <= nSteps - 4 * numBlockRows.  Column indices of a live column must be < K (loads of B are
 * range-checked against K * ldb by the buffer descriptor: an index past it reads zeros, never memory).  Replaces spmmBSRWrapper1 (src/spmm/bsr/spmm_bsr_k1.cu:44-91) for
 * BASELINE.json config 4; bf16 is a new capability, the reference has none. */
int mispmm_bsr_compact_slots_bf16_host(uint32_t numBlockRows, uint32_t bR, uint32_t bC, uint32_t numBlocks,
                                       const uint32_t *blockRowPtrs_host, const uint32_t *blockColIdxs_host,
                                       const float *blocks_host, uint32_t *nSteps_out, uint32_t *nUsedSteps_out,
                                       uint32_t *extraPtrs_out_host, uint32_t *cols_out_host, uint16_t *tiles_out_host);
int mispmm_bsrc_slots_bf16(mispmm_stream_t stream, uint32_t numBlockRows, uint32_t K, uint32_t nSteps, const uint32_t *extraPtrs,
                           const uint32_t *cols, const uint16_t *tiles, const uint16_t *B, uint32_t N, uint32_t ldb, void *C,
                           uint32_t ldc, int c_bf16);

/* -------------------------------------------------------------- COO x dense */
/* Row-major-sorted COO (the order convert_mtx.py:172-190 writes).  PRECONDITION: rowIdxs is
 * non-decreasing -- an unsorted array makes the row boundaries meaningless and the kernel read out of
 * range; callers holding arbitrary COO data run mispmm_coo_sort_by_row_host first (the host layer's
 * SparseMatrixCOO::copy2Device does).  Replaces spmmCOOWrapper1 (src/spmm/coo/spmm_coo_k1.cu:50-103)
 * without atomics.  rowPtrs_workspace: device scratch of (M + 1) uint32 the call
 * fills with row boundaries first; NULL makes every row group binary-search its
 * range instead (slower, no scratch).  kernel: 0 auto, 1 row-group gather,
 * 2 the same with the boundaries ALREADY in rowPtrs_workspace (written by an
 * earlier kernel-1 call or by mispmm_coo_row_bounds for the same rowIdxs): one
 * launch per SpMM instead of two -- the analysis step a format object does once
 * at upload (SparseMatrixCOO::copy2Device in the host layer). */
int mispmm_coo_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowIdxs,
                   const uint32_t *colIdxs, const float *vals, const float *B, uint32_t N, uint32_t ldb, float *C,
                   uint32_t ldc, uint32_t *rowPtrs_workspace, int kernel, int acc_mode);
#define MISPMM_COO_NUM_KERNELS 2
/* HOST helper.  The COO kernels need entries grouped by non-decreasing row (the reference's kernel, one atomicAdd
 * per entry, does not: spmm_coo_k1.cu:8-27).  *was_sorted says whether the input already is; with the three
 * output arrays given (each nnz long, may not alias the inputs) the entries are copied in a STABLE row order:
 * every row keeps its entries in storage order, the order spmmCOOCpu adds them in (spmm_coo.cpp:16-24), so
 * REFERENCE results stay bit-identical to it.  Outputs NULL = query only. */
int mispmm_coo_sort_by_row_host(uint32_t M, uint32_t nnz, const uint32_t *rowIdxs_host, const uint32_t *colIdxs_host,
                                const float *vals_host, uint32_t *rowIdxs_out_host, uint32_t *colIdxs_out_host,
                                float *vals_out_host, int *was_sorted);
/* rowPtrs_out[r] = index of the first entry of row r (r = 0..M; rowPtrs_out[M] = nnz) for a COO sorted by row. */
int mispmm_coo_row_bounds(mispmm_stream_t stream, uint32_t M, uint32_t nnz, const uint32_t *rowIdxs,
                          uint32_t *rowPtrs_out);

/* ------------------------------------------------------- vendor cross-check */
/* rocSPARSE generic SpMM (alpha 1, beta 0, fp32) into C, timed as prolog (handle, descriptors,
 * buffer, preprocess) / kernel (compute + sync) / epilog (teardown), all in microseconds.
 * Replaces cusparseTest (src/engine/cusparse.cu:9-57; algorithms chosen like
 * sparse_csr.cu:182-185 / sparse_coo.cu:97-100: CSR row split, COO segmented).  Synchronous and
 * allocating: not for graph capture.  For CSR ptrs_or_rows = rowPtrs, for COO = rowIdxs, for BSR
 * = blockRowPtrs with nnz = numBlocks and square blocks of block_dim (ignored otherwise). */
enum mispmm_vendor_format { MISPMM_VENDOR_CSR = 0, MISPMM_VENDOR_COO = 1, MISPMM_VENDOR_BSR = 2 };
int mispmm_vendor_spmm_f32(mispmm_stream_t stream, int format, uint32_t M, uint32_t K, uint32_t nnz, uint32_t block_dim,
                           const uint32_t *ptrs_or_rows, const uint32_t *cols, const float *vals, const float *B,
                           uint32_t N, uint32_t ldb, float *C, uint32_t ldc, double *pro_us, double *kernel_us,
                           double *epi_us);

/* ------------------------------------------------------------ dense helpers */
/* dst[cols x rows] = transpose(src[rows x cols]); both dense row-major buffers.
 * Replaces the host round trip of DenseMatrix::toOrdering (dense.cu:139-191). */
int mispmm_dense_transpose_f32(mispmm_stream_t stream, uint32_t rows, uint32_t cols, const float *src, float *dst);
/* round-to-nearest-even fp32 -> bf16 bit patterns */
int mispmm_f32_to_bf16(mispmm_stream_t stream, size_t n, const float *src, uint16_t *dst);
int mispmm_bf16_to_f32(mispmm_stream_t stream, size_t n, const uint16_t *src, float *dst);

/* ------------------------------------------------------- multi-GPU sharding */
/* HOST helper: split rows 0..M into `parts` contiguous ranges of near-equal nnz
 * (prefix search over rowPtrs_host); bounds_out_host has parts + 1 entries,
 * bounds[0] = 0, bounds[parts] = M.  New capability (the reference is
 * single-GPU, src/main.cu:176). */
int mispmm_shard_rows_by_nnz_host(uint32_t M, const uint32_t *rowPtrs_host, uint32_t parts,
                                  uint32_t *bounds_out_host);

/* ------------------------------------------------ multi-GPU: single process, arrays of per-device pointers */
/* New capability (the reference selects ONE device, src/main.cu:176, and has no exchange step).  Device slot d
 * (ordinal devices[d], stream streams[d]) owns rows [rowBounds[d], rowBounds[d+1]) of A and C:
 *   rowPtrs[d]  its row pointers, REBASED to start at 0 (rows + 1 entries);  colIdxs[d] / vals[d] its entries
 *               (column indices untouched: B is replicated);  nnz_host[d] their count;
 *   uniformRowNnz_host[d]  > 0 if every row of the slice holds exactly that many entries (may be NULL);
 *   B[d]        its replica of the dense operand [K x N], leading dimension ldb;
 *   C[d]        its buffer for the FULL C [M x N] (M = rowBounds[ndev]), leading dimension ldc: the device writes
 *               its own rows at their global position, the gather fills in the others'.
 * All pointer arrays are HOST arrays of DEVICE pointers.  Work is only enqueued, on the given streams; the
 * result is complete once every stream has been synchronised (C[0] for GATHER_TO_FIRST, every C[d] for the
 * ALL modes, each device's own rows for GATHER_NONE).  The calling thread's current device is preserved. */
#define MISPMM_MAX_SCATTER_DSTS 16
typedef struct mispmm_comm_s *mispmm_comm_t; /* one RCCL communicator per device slot (ncclCommInitAll) */
enum mispmm_gather_mode {
    MISPMM_GATHER_NONE = 0,     /* C stays row-sharded */
    MISPMM_GATHER_TO_FIRST = 1, /* slabs copied into C[0] over xGMI (hipMemcpyPeerAsync on the owner's stream) */
    MISPMM_GATHER_ALL_PEER = 2, /* every slab copied into every other device's C */
    MISPMM_GATHER_ALL_RCCL = 3, /* slabs of equal height (rowBounds[d] = d * M / ndev): ONE in-place ncclAllGather per
                                 * device; uneven slabs (nnz-balanced ranges): grouped in-place ncclBroadcast of each
                                 * slab from its owner (all-gather-v) */
    MISPMM_GATHER_ALL_RCCL_EQUAL = 4 /* rowBounds[d] = min(M, d * ceil(M / ndev)) REQUIRED, and every C[d] must hold
                                 * ndev * ceil(M / ndev) rows (rows past M are scratch): always ONE ncclAllGather */
};
/* Strided C (ldc > N): the peer gathers copy N columns per row and leave the gap columns of every destination
 * untouched; the RCCL gathers move contiguous runs and return MISPMM_ERR_UNSUPPORTED unless ldc == N. */
int mispmm_enable_peer_access(uint32_t ndev, const int *devices);
/* loads librccl.so on first use; MISPMM_ERR_UNSUPPORTED when it is not there */
int mispmm_comm_create(mispmm_comm_t *comm, uint32_t ndev, const int *devices);
int mispmm_comm_destroy(mispmm_comm_t comm);
int mispmm_multi_csr_f32(uint32_t ndev, const int *devices, const mispmm_stream_t *streams, const uint32_t *rowBounds_host,
                         uint32_t K, const uint32_t *const *rowPtrs, const uint32_t *const *colIdxs,
                         const float *const *vals, const uint32_t *nnz_host, const uint32_t *uniformRowNnz_host,
                         const float *const *B, uint32_t N, uint32_t ldb, float *const *C, uint32_t ldc, int kernel,
                         int acc_mode, int gather_mode, mispmm_comm_t comm);
/* The same sharding for the other two formats SURVEY.md section 8(e) names ("BSR shards by block-rows; ELL by rows"; the
 * reference has no sharded counterpart of src/spmm/ell/spmm_ell_k1.cu:10-35 or src/spmm/bsr/spmm_bsr_k1.cu:9-41).  Same contract
 * as mispmm_multi_csr_f32: host arrays of device pointers, per-device streams, every gather mode, strided C through the peer
 * gathers only.
 *   ELL: device slot d holds rows [rowBounds[d], rowBounds[d+1]) of the ROW-MAJOR ELL ([rows x width] colIdxs / vals of its own).
 *   BSR (BASELINE config 4's layout): device slot d holds BLOCK rows [blockRowBounds[d], blockRowBounds[d+1]) as column-compacted
 *   bf16 block rows in fixed step slots -- mispmm_bsr_compact_slots_bf16_host run on ITS block rows (nSteps_host[d], extraPtrs[d],
 *   cols[d], tiles[d]) --, a replica of the bf16 B and the full C (fp32, or bf16 with c_bf16 = 1); at most 64 slots. */
int mispmm_multi_ell_f32(uint32_t ndev, const int *devices, const mispmm_stream_t *streams, const uint32_t *rowBounds_host, uint32_t K,
                         uint32_t width, const uint32_t *const *colIdxs, const float *const *vals, const float *const *B, uint32_t N,
                         uint32_t ldb, float *const *C, uint32_t ldc, int kernel, int acc_mode, int gather_mode, mispmm_comm_t comm);
int mispmm_multi_bsrc_slots_bf16(uint32_t ndev, const int *devices, const mispmm_stream_t *streams, const uint32_t *blockRowBounds_host,
                                 uint32_t K, const uint32_t *nSteps_host, const uint32_t *const *extraPtrs, const uint32_t *const *cols,
                                 const uint16_t *const *tiles, const uint16_t *const *B, uint32_t N, uint32_t ldb, void *const *C, uint32_t ldc,
                                 int c_bf16, int gather_mode, mispmm_comm_t comm);
/* One launch copies `bytes` (a multiple of 16; 16-byte aligned pointers) from src to each of ndst <= 16
 * destinations, which may be another device's memory mapped into this process (peer access, or an IPC handle
 * opened by a one-process-per-GPU host: mispmm/dist.py).  Enqueue only; capturable into a graph. */
int mispmm_slab_scatter(mispmm_stream_t stream, const void *src, size_t bytes, void *const *dsts_host, uint32_t ndst);

#ifdef __cplusplus
}
#endif
#endif /* MISPMM_H */
