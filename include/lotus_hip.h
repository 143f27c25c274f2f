/* lotus_hip.h — C-ABI of liblotus_hip.so: the MI355X-native (gfx950) 3D-LOTUS hot path.
 *
 * The reference (vlc-robot/robot-3dlotus) has no FFI layer: its hot path calls three un-vendored
 * native packages (spconv, flash_attn, torch_scatter) and ATen/cuBLAS from Python.  Each entry
 * point below replaces one of those call sites; the citation names the reference file:line
 * (relative to genrobo3d/models/) whose arithmetic it implements.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative LOTUS_E_* code; lotus_last_error() returns a
 *     thread-local message.  Nothing throws, allocates device memory or synchronises the stream.
 *   - all pointers are DEVICE pointers borrowed for the duration of the enqueue, unless marked
 *     "host".  Tensors are dense row-major (activations `lotus_act_t`, everything else fp32); indices are int32; curve codes are int64.
 *   - `stream` is a hipStream_t (the caller's current stream; 0 = default stream).
 *   - workspaces are caller-allocated; sizes come from the matching *_workspace() function.
 *   - process-wide state: none besides the thread-local error message and the environment switches listed here, each read
 *     ONCE per process on first use (they select between kernels / tilings that all meet the same parity bars; the defaults
 *     are what every number in DESIGN.md was measured with; tests/test_capi.py checks this list against the library's strings):
 *       LOTUS_GEMM_DMA=0              dense products never use the LDS-DMA kernels (gemm_dma.hip)
 *       LOTUS_GEMM_DMA_MINROWS=n      ... only from n activation rows (default 16384)
 *       LOTUS_GEMM_DMA_MINBLOCKS=n    ... only when their 128-row tiles give at least n blocks (default 400)
 *       LOTUS_GEMM_DMA_WGRAD_BLOCKS=n blocks a weight gradient is split into on those kernels (default 256; 0: not used)
 *       LOTUS_SPLITK_FUSED=0          split-K products finish in a second launch instead of the last-arriving block
 *       LOTUS_CONV_TAP=0              deep sparse-convolution levels stay on the pair-compacted kernel
 *       LOTUS_CONV_TAP_MINC=c         tap-grouped dense path from c channels (default 64)
 *       LOTUS_CONV_TAP_SLAB_MB=m      ... while its partial slab stays below m MB (default 2047)
 *       LOTUS_CONV_WG_CHUNK=n         points per split of the sparse-convolution weight gradient (default 1024)
 *       LOTUS_CONV_OS=0               bf16 operand modes use the pair-compacted convolution kernel
 *       LOTUS_CONV_OS_F32=1|2|3       exact-fp32 products on the output-stationary convolution kernel (opt-in)
 *       LOTUS_XQ=0|2|3                cross attention on the tile kernels / patch attention on the per-query kernels / the
 *                                     cross-attention backward on the round-5 one-lane-per-query kernel
 *   - dropout masks are a stateless function of (seed, element index): one 32-bit hash decides two consecutive elements (since
 *     ABI version 2: the mask stream of a seed differs from version-1 builds), so the drop probability of the dense / LayerNorm /
 *     elementwise entry points is quantised to t / 65536, t >= 1, and kept elements are scaled by 1 / (1 - t / 65536) — the
 *     probability actually applied (round 6); the attention entry points use one hash per element and 1 / (1 - p).
 *   - lotus_abi_version() changes whenever an existing entry point changes its arguments (3 since round 6: lotus_adamw_step took the usage mask; 2 in round 5); bindings check it.
 */
#ifndef LOTUS_HIP_H
#define LOTUS_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LOTUS_OK 0
#define LOTUS_E_ARG (-1)
#define LOTUS_E_LAUNCH (-2)
#define LOTUS_E_UNSUPPORTED (-3)
#define LOTUS_E_WORKSPACE (-4)
/* Activation storage type.  Every [rows][channels] activation tensor (layer inputs / outputs, saved pre-activations, their
 * gradients) is `lotus_act_t`: float in this header's entry points.  The same sources are compiled a second time with bf16
 * storage; those twins (lotus_b16_<name>, pointers to 16-bit bf16) are declared in lotus_hip_b16.h.  Weights, biases,
 * parameter gradients, statistics and accumulation are fp32 in both. */
#ifdef LOTUS_ACT_BF16 /* internal: the second compilation of the library */
typedef unsigned short lotus_act_t;
#else
typedef float lotus_act_t;
#endif

#define LOTUS_ACT_NONE 0
#define LOTUS_ACT_GELU 1  /* nn.GELU() exact erf form */
#define LOTUS_ACT_LEAKY 2 /* nn.LeakyReLU(0.02), simple_policy_ptv3.py:42 */

const char* lotus_last_error(void);
int lotus_abi_version(void);

/* ---- stream link: order `to_stream` after everything enqueued on `from_stream` so far (event record + stream
 * wait from a caller-owned ring of timing-less events; no host synchronisation).  The Python host uses it to fork the
 * weight-gradient stream from / join it to the stream autograd runs on — what `side.wait_stream(main)` does in
 * torch, at ~2 us instead of ~15 us per fork.  create returns an opaque handle (0 on failure). */
unsigned long long lotus_streamlink_create(int nevents);
int lotus_streamlink_wait(unsigned long long link, void* from_stream, void* to_stream);
int lotus_streamlink_destroy(unsigned long long link);

/* ---- collectives of the data-parallel step (csrc/comm.cpp): RCCL all-reduces issued straight into the caller's stream on a
 * communicator of the library's own — ONE kernel in the stream the producer and the consumer of the message run on, where a
 * blocking ProcessGroupNCCL collective adds ~11 us of work-object / end-event overhead per message (26 us through its own
 * stream; tools/dbg/msg_cost.py).  Replaces the NCCL calls behind DistributedDataParallel and SyncBatchNorm
 * (genrobo3d/train/utils/distributed.py:196-205, train/train_simple_policy.py:116-117).  RCCL is dlopen()ed, not linked:
 *   lotus_comm_load(path)        host string or null: open `path`, else librccl.so.1 / librccl.so; idempotent
 *   lotus_comm_unique_id(id)     rank 0: 128 host bytes for the other ranks (the caller broadcasts them, e.g. over
 *                                torch.distributed — the bootstrap stays there)
 *   lotus_comm_create(id, n, r)  collective over the n ranks -> opaque handle, 0 on failure (device = the caller's current one)
 *   lotus_comm_allreduce(...)    in place on `buf`; dtype 0 = f32, 1 = f64, 2 = i32; op 0 = sum, 1 = max, 2 = average.
 *                                Collectives of one communicator must be issued in the same order on every rank; RCCL
 *                                runs them in that order even when they are issued from different streams (parallel.py keeps
 *                                one communicator per sending stream so that the two streams do NOT wait for each other,
 *                                after lotus_stream_probe below has shown that they cannot block each other either). */
int lotus_comm_load(const char* path);
int lotus_comm_version(void);
int lotus_comm_unique_id(void* id128_host);
unsigned long long lotus_comm_create(const void* id128_host, int nranks, int rank);
int lotus_comm_allreduce(unsigned long long comm, void* buf, size_t count, int dtype, int op, void* stream);
int lotus_comm_destroy(unsigned long long comm);
/* Two communicators in flight (gradient buckets on the communication stream, statistics on the training stream) are only safe
 * when a collective kernel that waits for its peers on one stream cannot hold back the other stream — which is a property of
 * the hardware queues the two HIP streams were mapped to, not of the program (csrc/stream_probe.hip has the deadlock picture).
 * lotus_stream_probe parks a kernel on `blocked` (it spins on a host flag, and leaves by itself after timeout_ms + 500 ms),
 * launches a second one on `other` and reports whether that one completed while the first was parked:
 * 1 = independent, 0 = `other` waits for `blocked`, < 0 = error.  Synchronises both streams before it returns; 1..2000 ms.
 * parallel.GradReducer calls it once per process before it creates the second communicator. */
int lotus_stream_probe(void* blocked, void* other, int timeout_ms);

/* ---- operand precision: a PER-CALL argument (`precision`) of the dense, sparse-convolution and attention entry points
 * (no process-wide state: two models with different precisions can share a process).  0 = fp32 MFMA, exact products
 * (the 1e-4 logit parity mode); 1 = bf16 operands, fp32 accumulate (the bf16 compute mode of BASELINE configs[4]);
 * 3 = bf16x3 split products (a = hi + lo; hi*hi + hi*lo + lo*hi).  Packed convolution weights must be used with the
 * precision they were packed for.  In the bf16-storage twins (lotus_hip_b16.h) lotus_b16_linear_fwd / _dgrad and the
 * composite entry points additionally accept 5 = bf16 products whose weight pointers address bf16 SHADOWS of the fp32
 * master weights (half the weight bytes per block, nothing converted while staging); attention / convolution entry points
 * reached through a composite see 1. */

/* ---- front end (integer, bit-exact) ------------------------------------------------------- */
/* Point.serialization grid step, PointTransformerV3/model.py:96-98: grid = int32(trunc((coord -
 * coord.min(0)) / grid_size)) with an IEEE fp32 divide; gmax = max grid coordinate.  coord rows have
 * stride ld floats (xyz = first three columns of pc_fts).  scratch: 4 x uint32. */
int lotus_fe_grid(const float* coord, long ld, int n, float grid_size, int* grid, int* gmax, unsigned* scratch,
                  void* stream);
/* serialization.encode for the 4 curves, serialization/default.py:9-24, z_order.py:40-101,
 * hilbert.py:91-198; depth = bit_length(gmax) (model.py:102).  Slot j of code[4][slot_stride] holds
 * curve perm4[j] (host int[4]; 0 z, 1 z-trans, 2 hilbert, 3 hilbert-trans) — the shuffle_orders
 * permutation of model.py:130-134 applied at the source.  state int32[8]: [0] |= error bits,
 * [1] = depth. */
int lotus_fe_encode(const int* grid, const int* batch, int n, const int* gmax, const int* perm4, int depth_bound,
                    int* state, long long* code, long slot_stride, void* stream);
/* torch.argsort(code) + inverse scatter, model.py:121-128, as a stable LSD radix sort of the 4
 * rows (ties by index).  n is read from n_ptr (device int); slot_stride of every buffer == n_max. */
size_t lotus_fe_sort_workspace(int n_max);
int lotus_fe_sort(const long long* code, long slot_stride, const int* n_ptr, int n_max, int key_bits,
                  long long* skeys, int* order, int* inverse, void* workspace, size_t workspace_bytes, void* stream);
/* Index part of SerializedPooling.forward, model.py:726-772 (torch.unique + sort + head gather):
 * cluster[n], CSR seg_start[n_child+1] into order0, n_child (device int), child code (rows
 * permuted by perm4, host int[4]), child grid/batch and per-cloud child counts.  n_dup (optional device int,
 * zeroed by the caller) receives the number of parent points that share their voxel with a lower-indexed point
 * (SURVEY.md Trap 5: augmented real batches have 1-7 % of them at level 0). */
int lotus_fe_pool(const long long* pcode, const long long* skey0, const int* order0, const int* pgrid,
                  const int* pbatch, const int* n_ptr, int n_max, const int* perm4, int nbatch, int* cluster,
                  int* seg_start, int* n_child, long long* ccode, int* cgrid, int* cbatch, int* ccounts,
                  int* n_dup, void* stream);
/* torch_scatter.segment_csr(coord, reduce="mean"), model.py:763-765 */
int lotus_fe_pool_coord(const float* pcoord, const int* order0, const int* seg_start, int n_child, float* ccoord,
                        void* stream);
/* SerializedAttention.get_padding_and_inverse, model.py:410-466, composed with order/inverse:
 * gidx[i] = order[pad[i]], owner[i] = (unpad[inverse[gidx[i]]] == i); kext[i] = -1 for owners, else the
 * compact index e of the borrowed copy, ext_pos[e] = i.  off/offp: int32 [B+1] exclusive prefix sums of
 * the per-cloud counts / padded counts. */
int lotus_fe_patch(const int* order, const int* off, const int* offp, int B, int K, int npad, int* gidx, int* owner,
                   int* kext, int* ext_pos, void* stream);
/* spconv submanifold neighbour lookup (SubMConv3d call sites model.py:615-622, :844-853): tap-major
 * nbr int32 [ksize^3][n], -1 = absent, duplicates -> lowest index (SURVEY.md Trap 5).  workspace: the hash table,
 * lotus_fe_neighbours_workspace(n) bytes, 16-byte aligned (16-byte {key, index} slots). */
size_t lotus_fe_neighbours_workspace(int n);
int lotus_fe_neighbours(const int* grid, const int* batch, int n, int ksize, int* nbr, void* workspace,
                        size_t workspace_bytes, void* stream);
/* Tap plan of a 3^3 table (consumed by lotus_subm_conv's tap-grouped path): plan int32 [lotus_fe_tap_plan_ints(n)] =
 * [27 pair counts (32 ints) | per tap the neighbour rows of the rows that have one, in processing order rowidx (optional),
 * segments of n64 = roundup(n, 64) | per tap the segment position of every row or -1]. */
size_t lotus_fe_tap_plan_ints(int n);
int lotus_fe_tap_plan(const int* nbr27, const int* rowidx, int n, int* plan, void* stream);

/* ---- dense layers (fp32 MFMA) ------------------------------------------------------------- */
/* nn.Linear (+GELU/LeakyReLU, +Dropout, +residual): y = dropout(act(x w^T + bias)) + residual;
 * `pre` (optional) receives x w^T + bias.  Call sites: model.py:386-387,572-574,623,707,804-805;
 * model_ca.py:27-31; simple_policy_ptv3.py:40-68,387. */
size_t lotus_linear_workspace(int M, int N, int K); /* optional split-K scratch for fwd / dgrad */
/* `counters` (optional, all three entry points): lotus_splitk_counters_bytes() bytes of device memory that are ZERO when
 * first passed and owned by one stream at a time.  With it a split-K product is ONE launch: every block stores its partial
 * tile, the last block to arrive at a tile sums the partials in fixed order and applies the epilogue, and resets the
 * tile's counter (so the buffer stays zero between calls).  Without it the reduction is a second launch. */
size_t lotus_splitk_counters_bytes(void);
int lotus_linear_fwd(const lotus_act_t* x, const float* w, const float* bias, const lotus_act_t* residual, lotus_act_t* y,
                     lotus_act_t* pre, int M, int N, int K, int act, float drop_p, unsigned long long drop_seed,
                     int precision, void* workspace, size_t workspace_bytes, void* counters, void* stream);
/* dx = (dy w) * act'(pre) * dropmask + add : chain rule through the producer of this layer's input */
int lotus_linear_dgrad(const lotus_act_t* dy, const float* w, lotus_act_t* dx, const lotus_act_t* pre, const lotus_act_t* add, int M,
                       int N, int K, int act, float drop_p, unsigned long long drop_seed, int precision, void* workspace,
                       size_t workspace_bytes, void* counters, void* stream);
size_t lotus_linear_wgrad_workspace(int M, int N, int K);
/* dw (+)= dy^T x, db (+)= colsum(dy); deterministic split-K */
int lotus_linear_wgrad(const lotus_act_t* dy, const lotus_act_t* x, float* dw, float* db, int M, int N, int K,
                       int accumulate, int precision, void* workspace, size_t workspace_bytes, void* counters,
                       void* stream);

/* ---- submanifold sparse convolution (spconv.SubMConv3d, model.py:615-622, :844-853) -------- */
/* mode 0 fwd: y[n][cout] = sum_t W[:,t,:] x[nbr[t][.]] + bias (+ add); mode 1 dgrad: x = dy, y = dx.
 * w is (cout, k, k, k, cin) row-major; rowidx (optional) = processing order of the rows.  w_t
 * (optional; 2 * cout*T*cin floats written by lotus_conv_weight_transpose: MFMA-fragment-packed copies of w
 * for both modes, cin and cout multiples of 32) and workspace (optional) enable the pair-compacted,
 * tap-split fast path for the 3^3 convolutions.  For thin inputs (mode 0, cin <= 8, e.g. the 5^3 stem) a workspace
 * of >= T*cin*cout floats selects the active-pair VALU kernel. */
/* tap_plan (optional; lotus_fe_tap_plan of the level's 3^3 table): for few rows and wide layers (lotus_conv_tap_eligible:
 * fp32 storage, exact products, n <= 32768, cin, cout >= 256) the convolution runs as 27 gathered dense products in one
 * launch + a fixed-order sum over the taps of every output row (workspace: lotus_subm_conv_workspace covers it) — the
 * pair-compacted kernel re-streams the weight tensor once per 64-row tile there. */
size_t lotus_subm_conv_workspace(int n, int cin, int cout);
int lotus_conv_tap_eligible(int n, int cin, int cout);
int lotus_conv_weight_transpose(const float* w, float* w_t, int cout, int T, int cin, int precision, void* stream);
int lotus_subm_conv(int mode, const lotus_act_t* x, const float* w, const float* w_t, const float* bias, const lotus_act_t* add,
                    lotus_act_t* y, const int* nbr, const int* rowidx, int n, int T, int cin, int cout, int precision,
                    const int* tap_plan, void* workspace, size_t workspace_bytes, void* stream);
/* Duplicate voxels (several points in one cell; the neighbour tables name the lowest index, `rep[p]` = centre tap
 * row nbr[T/2][p]): the true input gradient of mode 0 is dx[q] = [rep[q] == q] * sum_t W_t^T sum_{p in voxel(q) - d_t} dy[p].
 * lotus_conv_dup_fold writes dyr[r] = sum of dy over the points of r's voxel for representatives r (0 elsewhere), walking
 * the points in sorted (code, index) order `order0` (code0 = curve code per point; fixed order -> deterministic); mode 1
 * on dyr then yields the gradient for representatives, and lotus_conv_dup_mask resets the other rows to `add` (or 0). */
int lotus_conv_dup_fold(const lotus_act_t* dy, const long long* code0, const int* order0, int n, int C, lotus_act_t* dyr,
                        void* stream);
int lotus_conv_dup_mask(lotus_act_t* dx, const lotus_act_t* add, const int* rep, int n, int C, void* stream);
size_t lotus_subm_conv_wgrad_workspace(int n, int T, int cin, int cout);
/* precision (0 fp32 | 1 bf16 | 3 bf16x3): operand mode of the products, as lotus_linear_wgrad; accumulation, the split
 * partials and the bias gradient (summed from the fp32 rows) stay fp32. */
int lotus_subm_conv_wgrad(const lotus_act_t* dy, const lotus_act_t* x, float* dw, float* db, const int* nbr, int n, int T,
                          int cin, int cout, int accumulate, int precision, void* workspace, size_t workspace_bytes,
                          void* stream);

/* ---- normalisation ------------------------------------------------------------------------ */
/* nn.LayerNorm (model.py:624,627,645; model_ca.py:114,124): y = LN(x) (+ res) */
int lotus_layernorm_fwd(const lotus_act_t* x, const lotus_act_t* res, const float* gamma, const float* beta, lotus_act_t* y,
                        float* mean, float* rstd, int M, int C, float eps, void* stream);
size_t lotus_layernorm_bwd_workspace(int M, int C);
/* dz (optional, with drop_p > 0): second output dz = dx * mask(drop_seed), the nn.Dropout mask (same element index and
 * hash as the forward epilogue) of the layer that produced this block's input — its backward then needs no mask pass. */
int lotus_layernorm_bwd(const lotus_act_t* dy, const lotus_act_t* x, const float* mean, const float* rstd, const float* gamma,
                        const lotus_act_t* add, lotus_act_t* dx, float* dgamma, float* dbeta, int M, int C, int accumulate,
                        lotus_act_t* dz, float drop_p, unsigned long long drop_seed, void* workspace, size_t workspace_bytes,
                        void* stream);
/* number of partial rows lotus_layernorm_bwd(dgamma == NULL) leaves in its workspace */
int lotus_layernorm_bwd_parts(int M, int C);
/* the reduction of lotus_layernorm_bwd_params over an explicit number of partial rows [nparts][2][C] */
int lotus_layernorm_bwd_params_n(const void* workspace, int nparts, int C, float* dgamma, float* dbeta, int accumulate, void* stream);
/* nn.Linear input gradient + the backward of the nn.LayerNorm that feeds the layer (model.py:659-680: norm1 -> attn.qkv,
 * norm2 -> mlp.fc1; model_ca.py:114-124): dn = dy w [M,K] (scratch), dx = LN'(dn) + add, dz (optional) = dx * dropout mask;
 * *nparts (HOST int) = partial rows left in ln_workspace for lotus_layernorm_bwd_params_n.  One kernel where a 128-wide
 * tile covers the rows (K = 64 / 128, >= 16 k rows), otherwise lotus_linear_dgrad + lotus_layernorm_bwd. */
int lotus_linear_dgrad_ln(const lotus_act_t* dy, const float* w, const lotus_act_t* x, const float* mean, const float* rstd,
                          const float* gamma, const lotus_act_t* add, lotus_act_t* dx, lotus_act_t* dn, lotus_act_t* dz, float dz_p,
                          unsigned long long dz_seed, int M, int N, int K, int precision, void* workspace, size_t workspace_bytes,
                          void* counters, void* ln_workspace, size_t ln_workspace_bytes, int* nparts, void* stream);
/* second half of the above when it was called with dgamma == NULL: reduce the column partials left in workspace */
int lotus_layernorm_bwd_params(const void* workspace, int M, int C, float* dgamma, float* dbeta, int accumulate,
                               void* stream);
/* nn.BatchNorm1d(eps=1e-3, momentum=0.01) (+GELU), model_ca.py:226: statistics are exposed as
 * double sums[2*C+1] = (sum, sumsq, row count) so that a caller can all-reduce them across ranks (SyncBatchNorm,
 * train_simple_policy.py:116-117) between _stats and _finalize. */
size_t lotus_batchnorm_workspace(int M, int C);
int lotus_batchnorm_stats(const lotus_act_t* x, double* sums, int M, int C, void* workspace, size_t workspace_bytes,
                          void* stream);
int lotus_batchnorm_finalize(const double* sums, float* mean, float* invstd, float* running_mean,
                             float* running_var, int C, float eps, float momentum, void* stream);
/* One-launch statistics (round 4): the last block to arrive reduces the per-block partials in a fixed order (two levels of
 * 16) and, forward, finishes mean / invstd / running averages — lotus_batchnorm_stats + _finalize, or _bwd_stats, without
 * their second and third launch.  `counter`: 64 zeroed unsigned of the launching stream, left at zero (the tail of the
 * lotus_splitk_counters_bytes() buffer, at byte offset lotus_bn_counters_offset()).  SyncBatchNorm needs the sums between the
 * two halves: lotus_batchnorm_stats_fused with mean == invstd == NULL leaves the sums alone (round 5), the message follows, and
 * lotus_batchnorm_apply_sums finishes the statistics inside the apply pass. */
size_t lotus_bn_counters_offset(void);
int lotus_batchnorm_stats_fused(const lotus_act_t* x, double* sums, float* mean, float* invstd, float* running_mean, float* running_var,
                                int M, int C, float eps, float momentum, void* workspace, size_t workspace_bytes, void* counter,
                                void* stream);
int lotus_batchnorm_bwd_stats_fused(const lotus_act_t* dy, const lotus_act_t* x, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, double* sums, int M, int C, int act, void* workspace, size_t workspace_bytes,
                                    void* counter, void* stream);
/* ... and dbeta = sum dz, dgamma = sum dz * xhat of the LOCAL rows as fp32 vectors: what a SyncBatchNorm backward keeps before its
 * sums are all-reduced (round 5: no conversion kernels between the statistics and the message). */
int lotus_batchnorm_bwd_stats_fused_params(const lotus_act_t* dy, const lotus_act_t* x, const float* mean, const float* invstd,
                                           const float* gamma, const float* beta, double* sums, float* dgamma, float* dbeta, int M,
                                           int C, int act, void* workspace, size_t workspace_bytes, void* counter, void* stream);
int lotus_batchnorm_eval_stats(const float* running_mean, const float* running_var, float* mean, float* invstd, int C,
                               float eps, void* stream);
int lotus_batchnorm_apply(const lotus_act_t* x, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, lotus_act_t* y, int M, int C, int act, void* stream);
/* y = act(BN(x)) straight from the statistics sums (sum x, sum x^2, count; all-reduced across ranks for SyncBatchNorm): no
 * finalisation launch between the message and the apply pass; mean / invstd are written for backward, the running
 * averages updated (null = not tracked).  M == 0: only the statistics are finished. */
int lotus_batchnorm_apply_sums(const lotus_act_t* x, const double* sums, const float* gamma, const float* beta, lotus_act_t* y,
                               float* mean, float* invstd, float* running_mean, float* running_var, int M, int C, int act,
                               float eps, float momentum, void* stream);
int lotus_batchnorm_bwd_stats(const lotus_act_t* dy, const lotus_act_t* x, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, double* sums, int M, int C, int act,
                              void* workspace, size_t workspace_bytes, void* stream);
int lotus_batchnorm_bwd_apply(const lotus_act_t* dy, const lotus_act_t* x, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, const double* sums, lotus_act_t* dx, float* dgamma,
                              float* dbeta, int M, int C, int act, int train, int accumulate, void* stream);

/* ---- attention ---------------------------------------------------------------------------- */
/* flash_attn_varlen_qkvpacked_func (model.py:543-549) and flash_attn_varlen_kvpacked_func
 * (model_ca.py:62-66) with the preceding q_norm/k_norm LayerNorm(d, eps) (model.py:532-533,
 * model_ca.py:52-53), in fp32.  tiles: int32 [ntiles][4] = q_start, q_len, k_start, k_len (<= 128).
 * Row r of q lives at q + r*q_ld + q_off + h*d; k/v rows at kv + r*kv_ld + {k_off, v_off} + h*d.
 * drop_p / drop_seed: dropout on the probabilities (flash-attn dropout_p), regenerated in backward.
 * k_max: the caller's upper bound of k_len over all tiles (0 = unknown).  With identity-indexed rows (qidx = kidx = owner =
 * null) and 0 < k_max <= 32 — the point <-> instruction cross attention, whose key side is a cloud's 6-19 tokens — the call
 * takes the short-key kernels (one lane per query, keys / values as LDS broadcast rows, exact fp32 whatever `precision`).
 * k_max is a HARD contract: every tile's k_len must be <= k_max.  The backward kernel of that family keeps per-query state in
 * registers across a tile's key chunks and traps (GPU fault, loud) on a block that walks several tiles with k_len > 32. */
int lotus_attention_fwd(const lotus_act_t* q, long q_ld, int q_off, const lotus_act_t* kv, long kv_ld, int k_off, int v_off,
                        const int* qidx, const int* kidx, const int* owner, const int* tiles, int ntiles,
                        const float* qn_w, const float* qn_b, const float* kn_w, const float* kn_b, lotus_act_t* out,
                        long out_ld, float* lse, int H, int d, float scale, float eps, float drop_p,
                        unsigned long long drop_seed, int precision, int k_max, void* stream);
size_t lotus_attention_bwd_workspace(int nblocks, int H);
/* blocks: int32 [nblocks][6] = first_tile, n_tiles, tile_step, part_slot, k_start, k_len.  kext / ext_pos /
 * dkv_extra (optional, from lotus_fe_patch): k/v gradients of the borrowed tail-patch copies go to a side
 * buffer [n_extra][2*H*d] and are added to their point afterwards (no atomics, no zero-fill of dkv). */
int lotus_attention_bwd(const lotus_act_t* q, long q_ld, int q_off, const lotus_act_t* kv, long kv_ld, int k_off, int v_off,
                        const int* qidx, const int* kidx, const int* owner, const int* tiles, const int* blocks,
                        int nblocks, const float* qn_w, const float* qn_b, const float* kn_w, const float* kn_b,
                        const lotus_act_t* out, const lotus_act_t* dout, long out_ld, const float* lse, lotus_act_t* dq, long dq_ld,
                        int dq_off, lotus_act_t* dkv, long dkv_ld, int dk_off, int dv_off, long dkv_part_stride,
                        int atomic_out, const int* kext, const int* ext_pos, int n_extra, lotus_act_t* dkv_extra, float* dqn_w, float* dqn_b, float* dkn_w, float* dkn_b, int accumulate,
                        int H, int d, float scale, float eps, float drop_p, unsigned long long drop_seed,
                        int precision, int k_max, void* workspace, size_t workspace_bytes, void* stream);

/* ---- composite entry points (csrc/blocks.cpp): ONE call enqueues the whole forward / backward of a sub-block by chaining
 * the entry points above in the order the per-launch host path uses (bit-identical results).  The caller passes three flat
 * fp32 buffers whose sizes come from the *_floats() queries — `saved` (activations kept for backward), `grads` (all
 * parameter gradients of the sub-block, dw | db contiguous per layer) and `tmp` (scratch that must stay alive until the
 * enqueued work is done) — plus one workspace / counter buffer per stream.  Weight gradients are enqueued on `side`
 * (side == NULL: everything on `stream`), ordered after the launch that produces their operand by an event of `link` bound
 * to that launch as its completion event (no marker packet in `stream`), or after lotus_streamlink_wait(link, stream, side)
 * when the operand comes from the caller; join != 0 orders `stream` after `side` at the end.  MLP: y = x + drop(fc2(drop(GELU(fc1(LN(x)))))), PointTransformerV3/model.py:577-583,669-673.
 *   saved [n M*C | hpre M*Hd | a M*Hd | mean M | rstd M], grads [dg C | db C | dw1 Hd*C + db1 Hd | dw2 C*Hd + db2 C], every
 *   slice rounded up to 4 floats. */
size_t lotus_ffn_saved_floats(int M, int C, int Hd);
size_t lotus_ffn_grads_floats(int C, int Hd);
size_t lotus_ffn_tmp_floats(int M, int C, int Hd);
size_t lotus_ffn_ws_main_bytes(int M, int C, int Hd);
size_t lotus_ffn_ws_side_bytes(int M, int C, int Hd);
int lotus_ffn_fwd(const lotus_act_t* x, const float* g, const float* b, const float* w1, const float* b1, const float* w2,
                  const float* b2, lotus_act_t* y, float* saved, int M, int C, int Hd, float drop_p, unsigned long long seed1,
                  unsigned long long seed2, int precision, void* ws, size_t ws_bytes, void* counters, void* stream);
/* dz_in (optional): dy times the fc2 dropout mask, handed over by the next sub-block — i.e. the dz_out of a lotus_*_bwd
 * call issued earlier with the same (stream, side) pair; `side` is already ordered after that launch and is NOT ordered
 * after `stream` again for it.  dz_out (optional, dz_out_p > 0): dx times the dropout mask (dz_out_p, dz_out_seed) of the
 * previous sub-block (see lotus_layernorm_bwd). */
int lotus_ffn_bwd(const lotus_act_t* dy, const lotus_act_t* dz_in, const lotus_act_t* x, const float* g, const float* w1, const float* w2,
                  const float* saved, lotus_act_t* dx, lotus_act_t* dz_out, float dz_out_p, unsigned long long dz_out_seed, float* grads,
                  float* tmp, int M, int C, int Hd, float drop_p, unsigned long long seed1, unsigned long long seed2,
                  int precision, void* ws_main, size_t ws_main_bytes, void* ws_side, size_t ws_side_bytes, void* counters_main,
                  void* counters_side, unsigned long long link, int join, void* stream, void* side);

/* Patch self-attention sub-block, y = x + drop(proj(PatchAttention(qkv(LN(x))))), model.py:468-557,664-667.
 *   saved [n M*C | qkv M*3C | att M*C | lse npad*H | mean M | rstd M]
 *   grads [dg C | db C | dwqkv 3C*C + dbqkv 3C | gq d | bq d | gk d | bk d | dwp C*C + dbp C]   (d = C / H)
 * gidx / owner / tiles / blocks / kext / ext_pos: the level's patch tables (lotus_fe_patch + host tile lists). */
size_t lotus_selfattn_saved_floats(int M, int C, int H, int npad);
size_t lotus_selfattn_grads_floats(int C, int H);
size_t lotus_selfattn_tmp_floats(int M, int C, int n_extra);
size_t lotus_selfattn_ws_main_bytes(int M, int C, int H, int nblocks);
size_t lotus_selfattn_ws_side_bytes(int M, int C);
int lotus_selfattn_fwd(const lotus_act_t* x, const float* g, const float* b, const float* wqkv, const float* bqkv, const float* qnw,
                       const float* qnb, const float* knw, const float* knb, const float* wp, const float* bp, lotus_act_t* y,
                       float* saved, const int* gidx, const int* owner, const int* tiles, int ntiles, int npad, int M, int C,
                       int H, float scale, float drop_p, unsigned long long seed, float attn_p, unsigned long long attn_seed,
                       int precision, void* ws, size_t ws_bytes, void* counters, void* stream);
int lotus_selfattn_bwd(const lotus_act_t* dy, const lotus_act_t* dz_in, const lotus_act_t* x, const float* g, const float* wqkv, const float* qnw,
                       const float* qnb, const float* knw, const float* knb, const float* wp, const float* saved, lotus_act_t* dx,
                       float* grads, float* tmp, const int* gidx, const int* owner, const int* tiles, const int* blocks, int nblocks,
                       const int* kext, const int* ext_pos, int n_extra, int npad, int M, int C, int H, float scale, float drop_p,
                       unsigned long long seed, float attn_p, unsigned long long attn_seed, int precision, void* ws_main,
                       size_t ws_main_bytes, void* ws_side, size_t ws_side_bytes, void* counters_main, void* counters_side,
                       unsigned long long link, int join, void* stream, void* side);
/* Cross-attention sub-block, y = x + drop(proj(CrossAttention(q(LN(x)), kv(context)))), model_ca.py:46-101,135-140.
 *   saved [n M*C | q M*C | kv L*2C | att M*C | lse M*H | mean M | rstd M]
 *   grads [dg C | db C | dwq C*C + dbq C | dwkv 2C*Cc + dbkv 2C | gq d | bq d | gk d | bk d | dwp C*C + dbp C]
 * L = context rows, Cc = context channels, G = key-side partial slots of the backward tile lists. */
size_t lotus_crossattn_saved_floats(int M, int C, int H, int L);
size_t lotus_crossattn_grads_floats(int C, int H, int Cc);
size_t lotus_crossattn_tmp_floats(int M, int C, int L, int G);
size_t lotus_crossattn_ws_main_bytes(int M, int C, int H, int L, int Cc, int nblocks);
size_t lotus_crossattn_ws_side_bytes(int M, int C, int L, int Cc);
int lotus_crossattn_fwd(const lotus_act_t* x, const lotus_act_t* context, const float* g, const float* b, const float* wq, const float* bq,
                        const float* wkv, const float* bkv, const float* qnw, const float* qnb, const float* knw, const float* knb,
                        const float* wp, const float* bp, lotus_act_t* y, float* saved, const int* tiles, int ntiles, int M, int C, int H,
                        int L, int Cc, float scale, float drop_p, unsigned long long seed, float attn_p, unsigned long long attn_seed,
                        int precision, int k_max, void* ws, size_t ws_bytes, void* counters, void* stream);
int lotus_crossattn_bwd(const lotus_act_t* dy, const lotus_act_t* dz_in, const lotus_act_t* x, const lotus_act_t* context, const float* g, const float* wq,
                        const float* wkv, const float* qnw, const float* qnb, const float* knw, const float* knb, const float* wp,
                        const float* saved, lotus_act_t* dx, lotus_act_t* dctx, lotus_act_t* dz_out, float dz_out_p, unsigned long long dz_out_seed,
                        float* grads, float* tmp, const int* tiles, const int* blocks, int nblocks, int G, int M, int C, int H, int L,
                        int Cc, float scale, float drop_p, unsigned long long seed, float attn_p, unsigned long long attn_seed,
                        int precision, int k_max, void* ws_main, size_t ws_main_bytes, void* ws_side, size_t ws_side_bytes, void* counters_main,
                        void* counters_side, unsigned long long link, int join, void* stream, void* side);
/* Conditional positional encoding, y = x + LN(Linear(SubMConv3d_3(xs))), model.py:615-625,660-662 (xs == x in the encoder,
 * the stale skip branch in the decoder).  cw_packed: lotus_conv_weight_transpose output for `precision`.
 *   saved [c n*C | l n*C | mean n | rstd n], grads [dg C | db C | dlw C*C + dlb C | dcw C*27*C + dcb C].
 * bwd: dx_conv = conv input gradient (+ dy when add_dy); n_dup != 0 selects the duplicate-voxel passes (lotus_conv_dup_*). */
size_t lotus_cpe_saved_floats(int n, int C);
size_t lotus_cpe_grads_floats(int C);
size_t lotus_cpe_tmp_floats(int n, int C);
size_t lotus_cpe_ws_main_bytes(int n, int C);
size_t lotus_cpe_ws_conv_bytes(int n, int C);
size_t lotus_cpe_ws_side_bytes(int n, int C);
int lotus_cpe_fwd(const lotus_act_t* x, const lotus_act_t* xs, const float* cw, const float* cw_packed, const float* cb, const float* lw,
                  const float* lb, const float* g, const float* b, lotus_act_t* y, float* saved, const int* nbr27, const int* order0,
                  const int* tap_plan, int n, int C, int precision, void* ws, size_t ws_bytes, void* ws_conv, size_t ws_conv_bytes, void* counters, void* stream);
int lotus_cpe_bwd(const lotus_act_t* dy, const lotus_act_t* xs, const float* cw, const float* cw_packed, const float* lw, const float* g,
                  const float* saved, lotus_act_t* dx_conv, int add_dy, float* grads, float* tmp, const int* nbr27, const int* order0,
                  const int* tap_plan, const long long* code0, int n_dup, int n, int C, int precision, void* ws_main, size_t ws_main_bytes, void* ws_conv,
                  size_t ws_conv_bytes, void* ws_side, size_t ws_side_bytes, void* counters_main, void* counters_side,
                  unsigned long long link, int join, void* stream, void* side);
/* One (Block, CABlock) pair of a stage per call (round 4): cpe -> self-attention -> mlp -> cross-attention (kv from the
 * shared slab) -> mlp, model_ca.py:270-310, forward or backward, with the hand-overs of the pre-masked gradients wired
 * inside — exactly the launches of the five composite calls above, bit-identical results, one host transition.  Arguments
 * travel in three HOST arrays: P = device pointers (enum PairPtr in csrc/blocks.cpp, lotus_pair_nptr() entries), I =
 * integers incl. seeds, strides and byte sizes (enum PairInt, lotus_pair_nint()), F = {drop_p, attn_p, scale}.
 *   acts  [x1 | x2 | x3 | x4]           (outputs of the first four sub-blocks, kept for backward)
 *   saved / grads / tmp = the five composites' buffers back to back (cpe, self, mlp, cross_kv, mlp; tmp + 4 M*C) */
size_t lotus_pair_acts_floats(int M, int C);
size_t lotus_pair_saved_floats(int M, int C, int H, int Hd, int npad);
size_t lotus_pair_grads_floats(int C, int H, int Hd);
size_t lotus_pair_tmp_floats(int M, int C, int Hd, int n_extra, int L, int G);
size_t lotus_pair_ws_main_bytes(int M, int C, int H, int Hd, int nblocks_self, int nblocks_ca);
size_t lotus_pair_ws_side_bytes(int M, int C, int Hd);
size_t lotus_pair_ws_conv_bytes(int M, int C);
int lotus_pair_nptr(void);
int lotus_pair_nint(void);
int lotus_pair_fwd(const void* const* P, const long long* I, const double* F);
int lotus_pair_bwd(const void* const* P, const long long* I, const double* F);
/* out[e] = sum_z part[z * stride + e] in fixed order (e.g. the key-side partial slots of the cross-attention backward) */
int lotus_sum_slabs(const lotus_act_t* part, lotus_act_t* out, long n, long stride, int nz, void* stream);
/* out[r][c] (row stride out_ld) = sum_z part[z * stride + r * cols + c]: the same sum written into a column slice of a wider slab */
int lotus_sum_slabs_ld(const lotus_act_t* part, lotus_act_t* out, int rows, int cols, long out_ld, long stride, int nz, void* stream);
/* Cross-attention sub-block with PRECOMPUTED keys / values (round 4).  Every CABlock projects the same context
 * (model_ca.py:46-67), so the model computes kv for all of them with one product [L, Cc] x [Cc, sum 2C] and each block reads
 * its column slice `kv` (row stride kv_ld); backward writes d kv into the block's slice `dkv` (row stride dkv_ld) of the
 * shared gradient slab.  Otherwise identical to lotus_crossattn_fwd / _bwd.
 *   saved [n M*C | q M*C | att M*C | lse M*H | mean M | rstd M]
 *   grads [dg C | db C | dwq C*C + dbq C | gq d | bq d | gk d | bk d | dwp C*C + dbp C] */
size_t lotus_crossattn_kv_saved_floats(int M, int C, int H);
size_t lotus_crossattn_kv_grads_floats(int C, int H);
size_t lotus_crossattn_kv_tmp_floats(int M, int C, int L, int G);
size_t lotus_crossattn_kv_ws_main_bytes(int M, int C, int H, int nblocks);
size_t lotus_crossattn_kv_ws_side_bytes(int M, int C);
int lotus_crossattn_kv_fwd(const lotus_act_t* x, const lotus_act_t* kv, long kv_ld, const float* g, const float* b, const float* wq,
                           const float* bq, const float* qnw, const float* qnb, const float* knw, const float* knb, const float* wp,
                           const float* bp, lotus_act_t* y, float* saved, const int* tiles, int ntiles, int M, int C, int H, float scale,
                           float drop_p, unsigned long long seed, float attn_p, unsigned long long attn_seed, int precision, int k_max,
                           void* ws, size_t ws_bytes, void* counters, void* stream);
int lotus_crossattn_kv_bwd(const lotus_act_t* dy, const lotus_act_t* dz_in, const lotus_act_t* x, const lotus_act_t* kv, long kv_ld,
                           const float* g, const float* wq, const float* qnw, const float* qnb, const float* knw, const float* knb,
                           const float* wp, const float* saved, lotus_act_t* dx, lotus_act_t* dkv, long dkv_ld, lotus_act_t* dz_out,
                           float dz_out_p, unsigned long long dz_out_seed, float* grads, float* tmp, const int* tiles, const int* blocks,
                           int nblocks, int G, int M, int C, int H, int L, float scale, float drop_p, unsigned long long seed,
                           float attn_p, unsigned long long attn_seed, int precision, int k_max, void* ws_main, size_t ws_main_bytes,
                           void* ws_side, size_t ws_side_bytes, void* counters_main, void* counters_side, unsigned long long link,
                           int join, void* stream, void* side);

/* ---- pooling, head, losses ---------------------------------------------------------------- */
/* torch_scatter.segment_csr(reduce="max") and its arg-max backward, model.py:760-762 */
int lotus_pool_max_fwd(const lotus_act_t* x, const int* members, const int* seg, int nc, int C, lotus_act_t* y, int* arg,
                       void* stream);
int lotus_pool_max_bwd(const lotus_act_t* dy, const int* arg, const int* cluster, int n, int C, lotus_act_t* dx, void* stream);
/* SerializedUnpooling: parent.feat + point.feat[inverse], model.py:824 */
int lotus_unpool_fwd(const lotus_act_t* skip, const lotus_act_t* up, const int* cluster, int n, int C, lotus_act_t* x, void* stream);
int lotus_unpool_bwd(const lotus_act_t* dx, const int* members, const int* seg, int nc, int C, lotus_act_t* dup, void* stream);
/* ActionHead reduce == 'max': per-cloud torch.max(x, 0), simple_policy_ptv3.py:117-119 */
size_t lotus_cloud_max_workspace(int B, int C);
int lotus_cloud_max_fwd(const lotus_act_t* x, const int* off, int B, int C, lotus_act_t* y, int* arg, void* workspace, size_t workspace_bytes,
                        void* stream);
int lotus_cloud_max_bwd(const lotus_act_t* dy, const int* arg, const int* batch, int n, int C, const lotus_act_t* add, lotus_act_t* dx,
                        void* stream);
/* compute_loss (heatmap_disc / euler_disc), simple_policy_ptv3.py:322-373: losses[4] = pos, rot,
 * open, total.  xt [n][3*nb]; ae [B][nrot*3+1]; tgt = concatenated disc_pos_probs; gt [B][ga].
 * pos_stats: lotus_loss_stats_floats(B) floats (statistics [B*3][4] followed by slice partials). */
size_t lotus_loss_stats_floats(int B);
int lotus_loss_fwd(const lotus_act_t* xt, const lotus_act_t* ae, const float* tgt, const float* gt, const int* off, int B, int nb,
                   int nrot, int ga, float pos_w, float rot_w, float* losses, float* pos_stats, float* dae,
                   void* stream);
int lotus_loss_bwd(const lotus_act_t* xt, const float* tgt, const int* off, const int* batch, const float* pos_stats,
                   const float* dae_saved, const float* gl, float pos_w, float rot_w, int B, int n, int nb, int nrot,
                   lotus_act_t* dxt, lotus_act_t* dae_out, void* stream);
/* heatmap cross entropy alone, for the trajectory head (one call per trajectory step; per-cloud step masks enter as
 * the upstream gradients g[B*3]), genrobo3d/models/motion_planner_ptv3.py:327-336.  pos_stats as for lotus_loss_fwd;
 * pos_stats[(b*3+c)*4] = CE of cloud b, axis c. */
int lotus_pos_ce_fwd(const lotus_act_t* xt, const float* tgt, const int* off, int B, int nb, float* pos_stats, void* stream);
int lotus_pos_ce_bwd(const lotus_act_t* xt, const float* tgt, const int* off, const int* batch, const float* pos_stats,
                     const float* g, int B, int n, int nb, lotus_act_t* dxt, void* stream);
/* Trajectory losses of the motion planner on its [B, T]-sized tensors, genrobo3d/models/motion_planner_ptv3.py:327-397
 * (heatmap_disc / euler_disc), one launch.  Row r = b * T + t: ae [B*T][nrot*3 + 2] = rotation logits (bin, axis) at
 * bin*3 + axis | openness logit | stop logit; gt [B*T][ga] = gt_trajs (rotation bins at 3..5, openness at ga-1);
 * stop [B*T] = gt_trajs_stop; mask [B*T] = traj_masks; ce [B*T][3] = heatmap cross entropy per (cloud, step, axis)
 * (lotus_pos_ce_fwd).  losses[5] = pos, rot, open, stop, total (= pos_w pos + rot_w rot + open + stop).  dae / dce keep the
 * partial derivatives for lotus_mp_loss_bwd, which scales them by the upstream gradient g[5] (device). */
int lotus_mp_loss_fwd(const lotus_act_t* ae, const float* gt, const float* stop, const float* mask, const float* ce, int B, int T,
                      int nrot, int ga, float pos_w, float rot_w, float* losses, float* dae, float* dce, void* stream);
int lotus_mp_loss_bwd(const float* dae, const float* dce, const float* g, float pos_w, float rot_w, int B, int T, int nrot,
                      lotus_act_t* dae_out, float* dce_out, void* stream);
/* Trajectory head of the motion planner, genrobo3d/models/motion_planner_ptv3.py:88-97,113-114: hidden layer of step t =
 * dropout(act(base + bias_t)) with base [M][C] shared by the steps and bias_t [C] = step-embedding part of the first
 * Linear.  bwd: dpre = dh * act'(base + bias_t) * mask; dbase (+)= dpre (accumulate over the steps), dbias_t = colsum. */
int lotus_step_act_fwd(const lotus_act_t* base, const float* bias, lotus_act_t* out, int M, int C, int act, float drop_p,
                       unsigned long long drop_seed, void* stream);
size_t lotus_step_act_bwd_workspace(int M, int C);
int lotus_step_act_bwd(const lotus_act_t* dh, const lotus_act_t* base, const float* bias, lotus_act_t* dbase, float* dbias, int M, int C, int act,
                       float drop_p, unsigned long long drop_seed, int accumulate, void* workspace, size_t workspace_bytes,
                       void* stream);
/* elementwise plumbing */
int lotus_add(const lotus_act_t* a, const lotus_act_t* b, lotus_act_t* y, long n, void* stream);
/* nn.Dropout with a stateless counter-based mask (same (seed, index) -> same mask in backward) */
int lotus_dropout(const lotus_act_t* x, lotus_act_t* y, long n, float p, unsigned long long seed, void* stream);
/* DropPath / stochastic depth (timm.DropPath on the attention and MLP branches of a Block, model.py:655-657,666,672): one
 * Bernoulli draw per row, kept rows scaled by 1 / (1 - p); y = x + s_row * branch, x optional (null: y = s_row * branch, which
 * is also the backward map).  Stateless: keep iff hash(seed, row) >= p * 2^32.  C % 4 == 0. */
int lotus_drop_path(const lotus_act_t* branch, const lotus_act_t* x, lotus_act_t* y, int M, int C, float p, unsigned long long seed,
                    void* stream);

/* ---- soft position targets / arg-max position decode: get_disc_gt_pos_prob and
 * get_best_pos_from_disc_pos(best='max'), genrobo3d/utils/action_position_utils.py:7-46, :48-64.  pc = point rows whose
 * first three columns are xyz (row stride ld); kind 0 'plain' / 1 'dist'; tgt has the layout lotus_loss_fwd consumes. */
size_t lotus_pos_workspace(int B);
int lotus_pos_targets(const float* pc, long ld, const int* off, const int* batch, const float* gt, int ga,
                      const unsigned char* robot, int B, int n, int nb, double bin_size, int kind, float* tgt, void* workspace,
                      size_t workspace_bytes, void* stream);
int lotus_pos_decode_max(const lotus_act_t* xt, const float* pc, long ld, const int* off, int B, int nb, double bin_size,
                         double* best_pos, void* workspace, size_t workspace_bytes, void* stream);

/* ---- optimiser step: genrobo3d/train/optim/adamw.py:53-112 (HF AdamW: eps outside the bias
 * correction, decoupled decay after the update) and torch.nn.utils.clip_grad_norm_ (train_simple_policy.py:237-241) as
 * two multi-tensor launches.  All tables are device memory built by the caller: pointer arrays [T] (a null gradient
 * skips the tensor), numel [T], chunks [nchunks][2] = (tensor, chunk index) with lotus_mt_chunk() elements per chunk. */
int lotus_mt_chunk(void);
int lotus_grad_norm(const void* g_ptrs, const long* numel, const int* chunks, int nchunks, double* partial, float* norm_out,
                    float max_norm, void* stream);
/* shadow_ptrs (optional, device array [T] of bf16 pointers, null entries allowed): the updated parameter is also stored
 * rounded to bf16 — the weight operand of the bf16-storage products (`precision` 5 of lotus_b16_linear_fwd / _dgrad),
 * refreshed by the very step that changes the fp32 master (BASELINE configs[4]: bf16 weights, fp32 master weights). */
/* used / used_idx (optional, device int32): usage mask of a data-parallel step — tensor t is skipped when
 * used[used_idx ? used_idx[t] : t] == 0, i.e. when NO rank produced a gradient for it (the MAX all-reduce of the ranks'
 * usage flags, parallel.GradReducer.used_mask): what DistributedDataParallel(find_unused_parameters=True) does with its host
 * bitmap (genrobo3d/train/utils/distributed.py:196-205), decided on the device in the step the usage changes. */
int lotus_adamw_step(const void* p_ptrs, const void* g_ptrs, const void* m_ptrs, const void* v_ptrs, const long* numel,
                     const float* step_size, const float* decay, const int* chunks, int nchunks, double beta1, double beta2,
                     double eps, const float* clip_coef, const void* shadow_ptrs, const int* used, const int* used_idx,
                     void* stream);
/* dst[t][i] = bf16(src[t][i]): creation / refresh of the weight shadows outside an optimiser step (same tables as above) */
int lotus_shadow_cast(const void* src_ptrs, const void* dst_ptrs, const long* numel, const int* chunks, int nchunks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOTUS_HIP_H */
