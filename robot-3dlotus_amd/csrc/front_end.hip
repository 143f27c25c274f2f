// lotus-hip: integer front-end of 3D-LOTUS on device — voxelisation, space-filling-curve codes,
// stable radix sort, grid-pooling clusters, patch tables and the submanifold neighbour tables.
// Every output is bit-exact against the oracle (oracle/front_end.py) and therefore against the
// reference's Point.serialization / SerializedPooling / get_padding_and_inverse
// (PointTransformerV3/model.py:83-138, :713-772, :410-466) and serialization/{z_order,hilbert}.py.
//
// All kernels read their point count from device memory (n_ptr) so that the whole multi-level
// pipeline runs without a host synchronisation; the host sizes grids by an upper bound (n_max).
// Everything here is HBM/latency-bound integer work: coalesced SoA accesses, wave64 ballots for
// the stable ranks of the radix scatter, one LDS histogram per wave.
#include "common.h"

// device-side status words (fe_state int32[8]): [0] error flags, [1] depth of level 0
#define FE_ERR_DEPTH 1

__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// ---------------------------------------------------------------- grid coordinates (model.py:96-98)
__global__ void fe_min_kernel(const float* __restrict__ coord, long ld, int n, unsigned* __restrict__ mn) {
  unsigned m[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) m[a] = min(m[a], f2ord(coord[(long)i * ld + a]));
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int o = 32; o > 0; o >>= 1) m[a] = min(m[a], (unsigned)__shfl_xor((int)m[a], o, 64));
    if ((threadIdx.x & 63) == 0) atomicMin(&mn[a], m[a]);
  }
}

__global__ void fe_grid_kernel(const float* __restrict__ coord, long ld, int n, const unsigned* __restrict__ mn,
                               float grid_size, int* __restrict__ grid, int* __restrict__ gmax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int mx = 0;
  if (i < n) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float d = coord[(long)i * ld + a] - ord2f(mn[a]);
      const int g = (int)truncf(__fdiv_rn(d, grid_size));  // IEEE divide, SURVEY.md Appendix C.6
      grid[(long)i * 3 + a] = g;
      mx = max(mx, g);
    }
  }
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(gmax, mx);
}

// ---------------------------------------------------------------- curve codes
__device__ __forceinline__ long long z_code(int x, int y, int z, int depth) {  // z_order.py:40-49
  long long key = 0;
  for (int i = 0; i < depth; ++i) {
    const long long m = 1ll << i;
    key |= (((long long)x & m) << (2 * i + 2)) | (((long long)y & m) << (2 * i + 1)) | (((long long)z & m) << (2 * i));
  }
  return key;
}
__device__ __forceinline__ long long hilbert_code(int x0, int x1, int x2, int depth) {  // hilbert.py:91-198
  unsigned X[3] = {(unsigned)x0, (unsigned)x1, (unsigned)x2};
  for (int pos = depth - 1; pos >= 0; --pos) {
    const unsigned Q = 1u << pos, P = Q - 1u;
#pragma unroll
    for (int dim = 0; dim < 3; ++dim) {
      if (X[dim] & Q) {
        X[0] ^= P;
      } else {
        const unsigned t = (X[0] ^ X[dim]) & P;
        X[0] ^= t;
        X[dim] ^= t;
      }
    }
  }
  unsigned long long g = 0;
  for (int pos = 0; pos < depth; ++pos) {
#pragma unroll
    for (int dim = 0; dim < 3; ++dim) g |= (unsigned long long)((X[dim] >> pos) & 1u) << (3 * pos + (2 - dim));
  }
  for (int s = 1; s < 3 * depth; s <<= 1) g ^= g >> s;
  return (long long)g;
}

// code[slot][i] for slot j = curve perm[j] of (z, z-trans, hilbert, hilbert-trans); default.py:9-24
__global__ void fe_encode_kernel(const int* __restrict__ grid, const int* __restrict__ batch, int n,
                                 const int* __restrict__ gmax, int4 perm, int depth_bound, int* __restrict__ state,
                                 long long* __restrict__ code, long slot_stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int depth = 32 - __clz(*gmax);  // int(grid.max()).bit_length(), model.py:102
  if (i == 0) {
    state[1] = depth;
    if (depth > depth_bound || depth > 16) atomicOr(&state[0], FE_ERR_DEPTH);
  }
  if (i >= n) return;
  const int x = grid[(long)i * 3], y = grid[(long)i * 3 + 1], z = grid[(long)i * 3 + 2];
  const long long b = (long long)batch[i] << (3 * depth);
  const int pm[4] = {perm.x, perm.y, perm.z, perm.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    long long c;
    switch (pm[j]) {
      case 0: c = z_code(x, y, z, depth); break;
      case 1: c = z_code(y, x, z, depth); break;
      case 2: c = hilbert_code(x, y, z, depth); break;
      default: c = hilbert_code(y, x, z, depth); break;
    }
    code[(long)j * slot_stride + i] = b | c;
  }
}

// ---------------------------------------------------------------- stable LSD radix sort (8-bit digits)
// Tile = 1024 keys owned by ONE wave (16 rounds of 64 in order), so the stable rank of a key is
// base[digit] + (number of lower lanes with the same digit in this round): a wave64 match built
// from 8 ballots, no cross-wave bookkeeping.
#define RS_TILE 1024

__device__ __forceinline__ unsigned long long match_digit(int dig) {
  unsigned long long m = ~0ull;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const unsigned long long bal = __ballot((dig >> b) & 1);
    m &= ((dig >> b) & 1) ? bal : ~bal;
  }
  return m;
}

__global__ __launch_bounds__(256) void rs_hist_kernel(const long long* __restrict__ keys, long slot_stride,
                                                      const int* __restrict__ n_ptr, int shift, int ntiles_max,
                                                      int* __restrict__ hist) {
  __shared__ int h[4][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, slot = blockIdx.y;
  const int n = *n_ptr;
  const int tile = blockIdx.x * 4 + wave;
  for (int i = lane; i < 256; i += 64) h[wave][i] = 0;
  __syncthreads();
  const long base = (long)tile * RS_TILE;
  for (int r = 0; r < RS_TILE / 64; ++r) {
    const long i = base + r * 64 + lane;
    if (i < n) atomicAdd(&h[wave][(int)((keys[slot * slot_stride + i] >> shift) & 255)], 1);
  }
  __syncthreads();
  if (tile < ntiles_max)
    for (int dgt = lane; dgt < 256; dgt += 64) hist[((long)slot * 256 + dgt) * ntiles_max + tile] = h[wave][dgt];
}

// exclusive scan of hist[slot][digit][tile] (digit-major) by one block per slot.  Segments of 16384 counters
// are staged through LDS with coalesced loads / stores; every thread scans its 16 consecutive counters in LDS
// (row stride 17 -> conflict-free), one block-wide scan of the thread totals per segment.
#define RS_SCAN_SEG 16384
__global__ __launch_bounds__(1024) void rs_scan_kernel(int* __restrict__ hist, int ntiles_max) {
  __shared__ int buf[1024 * 17];
  __shared__ int wsum[16];
  int* h = hist + (long)blockIdx.x * 256 * ntiles_max;
  const int total = 256 * ntiles_max;
  int carry = 0;
  for (int base = 0; base < total; base += RS_SCAN_SEG) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int e = k * 1024 + threadIdx.x;  // element within the segment
      buf[(e >> 4) * 17 + (e & 15)] = base + e < total ? h[base + e] : 0;
    }
    __syncthreads();
    int* mine = buf + threadIdx.x * 17;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += mine[k];
    int x = s;
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if ((threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    int run = carry + x - s, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int v = wsum[w];
      if (w < (int)(threadIdx.x >> 6)) run += v;
      tot += v;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int v = mine[k];
      mine[k] = run;
      run += v;
    }
    carry += tot;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int e = k * 1024 + threadIdx.x;
      if (base + e < total) h[base + e] = buf[(e >> 4) * 17 + (e & 15)];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void rs_scatter_kernel(const long long* __restrict__ kin, const int* __restrict__ vin,
                                                        long long* __restrict__ kout, int* __restrict__ vout,
                                                        long slot_stride, const int* __restrict__ n_ptr, int shift,
                                                        int ntiles_max, const int* __restrict__ hist, int first_pass) {
  __shared__ int base_s[256];
  const int lane = threadIdx.x, slot = blockIdx.y, tile = blockIdx.x;
  const int n = *n_ptr;
  for (int dgt = lane; dgt < 256; dgt += 64) base_s[dgt] = hist[((long)slot * 256 + dgt) * ntiles_max + tile];
  __syncthreads();
  const long so = (long)slot * slot_stride;
  const long t0 = (long)tile * RS_TILE;
  if (t0 >= n) return;
  long long key[RS_TILE / 64];
  int val[RS_TILE / 64];
#pragma unroll
  for (int r = 0; r < RS_TILE / 64; ++r) {
    const long i = t0 + r * 64 + lane;
    key[r] = i < n ? kin[so + i] : 0;
    val[r] = i < n ? (first_pass ? (int)i : vin[so + i]) : 0;
  }
#pragma unroll
  for (int r = 0; r < RS_TILE / 64; ++r) {
    const long i = t0 + r * 64 + lane;
    const bool live = i < n;
    const int dig = live ? (int)((key[r] >> shift) & 255) : 256 + 0;  // dead lanes never match live digits
    unsigned long long m = match_digit(dig & 255);
    const unsigned long long livemask = __ballot(live);
    m &= live ? livemask : ~livemask;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    int pos = 0;
    if (live) pos = base_s[dig] + rank;
    __builtin_amdgcn_wave_barrier();
    if (live && rank == 0) base_s[dig] += __popcll(m);  // leader (lowest lane of the group) advances the base
    __builtin_amdgcn_wave_barrier();
    if (live) {
      kout[so + pos] = key[r];
      vout[so + pos] = val[r];
    }
  }
}

__global__ void fe_inverse_kernel(const int* __restrict__ order, int* __restrict__ inverse, long slot_stride,
                                  const int* __restrict__ n_ptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *n_ptr) return;
  const long so = (long)blockIdx.y * slot_stride;
  inverse[so + order[so + i]] = i;
}

// ---------------------------------------------------------------- grid pooling (model.py:726-772)
// Parent points sorted by code[0] (sorted keys `skey`, permutation `order0`).  Cluster id of a
// parent = rank of (code[0] >> 3) among the distinct values; children are therefore stored in
// ascending parent-cell order, exactly like torch.unique(sorted=True).
// cluster heads (first element of every run of equal parent keys) counted per 1024-element block
// n_dup (optional): += number of points that share their voxel with a lower-indexed point (equal full keys)
__global__ __launch_bounds__(1024) void fe_pool_count_kernel(const long long* __restrict__ skey,
                                                             const int* __restrict__ n_ptr, int* __restrict__ blk,
                                                             int* __restrict__ n_dup) {
  __shared__ int wsum[16], dsum[16];
  const int n = *n_ptr;
  const int i = blockIdx.x * 1024 + threadIdx.x;
  int head = 0, dup = 0;
  if (i < n) {
    head = (i == 0) || ((skey[i] >> 3) != (skey[i - 1] >> 3));
    dup = (i > 0) && (skey[i] == skey[i - 1]);
  }
  const int c = __popcll(__ballot(head)), d = __popcll(__ballot(dup));
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = c; dsum[threadIdx.x >> 6] = d; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0, td = 0;
    for (int w = 0; w < 16; ++w) { t += wsum[w]; td += dsum[w]; }
    blk[blockIdx.x] = t;
    if (n_dup && td) atomicAdd(n_dup, td);
  }
}

__global__ __launch_bounds__(1024) void fe_pool_scan_kernel(const long long* __restrict__ skey,
                                                            const int* __restrict__ order0,
                                                            const int* __restrict__ n_ptr, const int* __restrict__ blk,
                                                            int* __restrict__ cluster, int* __restrict__ seg_start,
                                                            int* __restrict__ n_child) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int n = *n_ptr;
  if (threadIdx.x < 64) {  // heads in all preceding blocks (gridDim.x <= a few hundred)
    int t = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 64) t += blk[b];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (threadIdx.x == 0) carry_s = t;
  }
  const int i = blockIdx.x * 1024 + threadIdx.x;
  int head = 0;
  if (i < n) head = (i == 0) || ((skey[i] >> 3) != (skey[i - 1] >> 3));
  int x = head;
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if ((threadIdx.x & 63) >= o) x += y;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
  const int id = carry_s + woff + x - 1;  // inclusive scan - 1
  if (i < n) {
    cluster[order0[i]] = id;
    if (head) seg_start[id] = i;
    if (i == n - 1) {
      *n_child = id + 1;
      seg_start[id + 1] = n;
    }
  }
  if (n == 0 && i == 0) {
    *n_child = 0;
    seg_start[0] = 0;
  }
}

// child tables from the head (first in sorted order) of every cluster
__global__ void fe_pool_child_kernel(const long long* __restrict__ pcode, long pslot_stride,
                                     const int* __restrict__ order0, const int* __restrict__ seg_start,
                                     const int* __restrict__ pgrid, const int* __restrict__ pbatch,
                                     const int* __restrict__ n_child, int4 perm, long long* __restrict__ ccode,
                                     long cslot_stride, int* __restrict__ cgrid, int* __restrict__ cbatch,
                                     int* __restrict__ ccounts, int pooling_depth) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= *n_child) return;
  const int head = order0[seg_start[c]];
  const int pm[4] = {perm.x, perm.y, perm.z, perm.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) ccode[(long)j * cslot_stride + c] = pcode[(long)pm[j] * pslot_stride + head] >> (3 * pooling_depth);
#pragma unroll
  for (int a = 0; a < 3; ++a) cgrid[(long)c * 3 + a] = pgrid[(long)head * 3 + a] >> pooling_depth;
  const int b = pbatch[head];
  cbatch[c] = b;
  // children are sorted by cloud, so a wave nearly always holds one cloud: one atomic per wave
  const int b0 = __shfl(b, __ffsll((long long)__ballot(1)) - 1, 64);
  const unsigned long long same = __ballot(b == b0);
  if (same == __ballot(1)) {
    if ((int)(threadIdx.x & 63) == __ffsll((long long)same) - 1) atomicAdd(&ccounts[b0], __popcll(same));
  } else {
    atomicAdd(&ccounts[b], 1);
  }
}

// ---------------------------------------------------------------- patch tables (model.py:410-466)
// gidx[i] = order[pad[i]] for padded position i; owner[i] = 1 iff unpad[inverse[gidx[i]]] == i.
// off / offp: exclusive prefix sums of the per-cloud counts / padded counts (B+1 entries, device).
__global__ void fe_patch_kernel(const int* __restrict__ order, const int* __restrict__ off,
                                const int* __restrict__ offp, int B, int K, int npad, int* __restrict__ gidx,
                                int* __restrict__ owner, int* __restrict__ kext, int* __restrict__ ext_pos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  int lo = 0, hi = B;  // find cloud c with offp[c] <= i < offp[c+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (offp[mid] <= i) lo = mid; else hi = mid;
  }
  const int c = lo;
  const int ncnt = off[c + 1] - off[c];
  const int l = i - offp[c];
  const int src = l < ncnt ? l : l - K;  // the tail patch borrows the K - r points preceding it
  gidx[i] = order[off[c] + src];
  owner[i] = l < ncnt;
  if (kext) {
    const int e = l < ncnt ? -1 : (offp[c] - off[c]) + (l - ncnt);  // compact index of the borrowed copy
    kext[i] = e;
    if (e >= 0) ext_pos[e] = i;
  }
}

// ---------------------------------------------------------------- neighbour tables (spconv semantics)
#define HT_EMPTY 0xffffffffffffffffull
__device__ __forceinline__ unsigned long long vox_key(int b, int x, int y, int z) {
  return ((unsigned long long)b << 48) | ((unsigned long long)x << 32) | ((unsigned long long)y << 16) | (unsigned long long)z;
}
__device__ __forceinline__ unsigned ht_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (unsigned)k;
}
// One 16-byte slot per entry: {key (8 bytes), lowest point index (4), pad} — a probe is ONE 16-byte load instead of a key
// load and a dependent value load from a second array (the kernel sits at the L2 random-access rate: halving the accesses
// halves its time).  An empty slot is all ones: the whole table is initialised by a single memset, and the index is
// merged with an UNSIGNED atomicMin.
struct __attribute__((aligned(16))) HtSlot { unsigned long long key; unsigned val; unsigned pad; };
__global__ void fe_hash_build_kernel(const int* __restrict__ grid, const int* __restrict__ batch, int n, HtSlot* __restrict__ ht,
                                     unsigned mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = vox_key(batch[i], grid[(long)i * 3], grid[(long)i * 3 + 1], grid[(long)i * 3 + 2]);
  unsigned slot = ht_hash(key) & mask;
  while (true) {
    const unsigned long long prev = atomicCAS(&ht[slot].key, HT_EMPTY, key);
    if (prev == HT_EMPTY || prev == key) {
      atomicMin(&ht[slot].val, (unsigned)i);  // duplicate voxels -> lowest index wins (SURVEY.md Trap 5)
      break;
    }
    slot = (slot + 1) & mask;
  }
}
// nbr[t][i], t = ((dx+r)*k + (dy+r))*k + (dz+r).  One thread per (point, tap) and a fully scattering hash: ~19 M random
// probes per level-0 5^3 table, i.e. the kernel sits at the L2 random-access rate.  Measured and rejected: a z-local home
// slot (the 8 cells of a z octet in one 64-byte line) with one thread probing the k cells of a (dx, dy) column — surface
// clouds fill whole octets, probe chains grow and the per-thread chains serialise: 22 -> 65 us per launch, hash build
// 11 -> 28 us.
__global__ void fe_neighbour_kernel(const int* __restrict__ grid, const int* __restrict__ batch, int n, int ksize,
                                    const HtSlot* __restrict__ ht, unsigned mask, int* __restrict__ nbr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (i >= n) return;
  const int r = ksize / 2;
  const int dx = t / (ksize * ksize) - r, dy = (t / ksize) % ksize - r, dz = t % ksize - r;
  const int x = grid[(long)i * 3] + dx, y = grid[(long)i * 3 + 1] + dy, z = grid[(long)i * 3 + 2] + dz;
  int res = -1;
  if (x >= 0 && y >= 0 && z >= 0 && x < 65536 && y < 65536 && z < 65536) {
    const unsigned long long key = vox_key(batch[i], x, y, z);
    unsigned slot = ht_hash(key) & mask;
    while (true) {
      const uint4 s4 = *reinterpret_cast<const uint4*>(&ht[slot]);  // key and index in one access
      const unsigned long long k = (unsigned long long)s4.x | ((unsigned long long)s4.y << 32);
      if (k == key) { res = (int)s4.z; break; }
      if (k == HT_EMPTY) break;
      slot = (slot + 1) & mask;
    }
  }
  nbr[(long)t * n + i] = res;
}

// segment mean of coordinates (SerializedPooling coord, model.py:763-765)
__global__ void fe_pool_coord_kernel(const float* __restrict__ pcoord, const int* __restrict__ order0,
                                     const int* __restrict__ seg_start, int n_child, float* __restrict__ ccoord) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_child) return;
  float s[3] = {0.f, 0.f, 0.f};
  const int a = seg_start[c], b = seg_start[c + 1];
  for (int i = a; i < b; ++i) {
    const int p = order0[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] += pcoord[(long)p * 3 + k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) ccoord[(long)c * 3 + k] = s[k] / (float)(b - a);
}

extern "C" {

// grid i32[n][3] = trunc((coord - min(coord)) / grid_size); gmax (device int) = max grid coordinate.
// scratch: 4 x uint32 (device).  coord rows have stride ld floats (xyz are the first 3 of pc_fts).
int lotus_fe_grid(const float* coord, long ld, int n, float grid_size, int* grid, int* gmax, unsigned* scratch,
                  void* stream) {
  LOTUS_CHECK_ARG(coord && grid && gmax && scratch && n > 0, "lotus_fe_grid: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(scratch, 0xff, 4 * sizeof(unsigned), st);
  (void)hipMemsetAsync(gmax, 0, sizeof(int), st);
  int g = cdiv(n, 256);
  LOTUS_LAUNCH(fe_min_kernel, dim3(g > 1024 ? 1024 : g), dim3(256), 0, st, coord, ld, n, scratch);
  LOTUS_LAUNCH(fe_grid_kernel, dim3(g), dim3(256), 0, st, coord, ld, n, scratch, grid_size, grid, gmax);
  LOTUS_LAUNCH_CHECK("lotus_fe_grid");
  return LOTUS_OK;
}

// code i64[4][slot_stride]: slot j holds curve perm[j] (0 z, 1 z-trans, 2 hilbert, 3 hilbert-trans).
// state int32[8] (device): [0] |= error flags, [1] = depth.
int lotus_fe_encode(const int* grid, const int* batch, int n, const int* gmax, const int* perm4, int depth_bound,
                    int* state, long long* code, long slot_stride, void* stream) {
  LOTUS_CHECK_ARG(grid && batch && gmax && perm4 && state && code && n > 0, "lotus_fe_encode: bad arguments");
  int4 pm = make_int4(perm4[0], perm4[1], perm4[2], perm4[3]);
  LOTUS_LAUNCH(fe_encode_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, grid, batch, n, gmax, pm,
                     depth_bound, state, code, slot_stride);
  LOTUS_LAUNCH_CHECK("lotus_fe_encode");
  return LOTUS_OK;
}

size_t lotus_fe_sort_workspace(int n_max) {
  const int ntiles = cdiv(n_max, RS_TILE);
  const int ntiles4 = cdiv(ntiles, 4) * 4;
  return (size_t)4 * n_max * (sizeof(long long) + sizeof(int)) + (size_t)4 * 256 * ntiles4 * sizeof(int);
}

// Stable argsort of 4 key rows: order[slot][i] = index of the i-th smallest key (ties by index);
// skeys receives the sorted keys.  n is read from n_ptr (device); key_bits bounds the key width.
int lotus_fe_sort(const long long* code, long slot_stride, const int* n_ptr, int n_max, int key_bits,
                  long long* skeys, int* order, int* inverse, void* workspace, size_t workspace_bytes, void* stream) {
  LOTUS_CHECK_ARG(code && n_ptr && skeys && order && n_max > 0 && key_bits > 0 && key_bits <= 64,
                  "lotus_fe_sort: bad arguments");
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= lotus_fe_sort_workspace(n_max), "lotus_fe_sort: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = cdiv(n_max, RS_TILE);
  const int ntiles4 = cdiv(ntiles, 4) * 4;
  long long* kb = (long long*)workspace;
  int* vb = (int*)(kb + (size_t)4 * n_max);
  int* hist = vb + (size_t)4 * n_max;
  const int passes = cdiv(key_bits, 8);
  // ping-pong so that the last pass lands in (skeys, order)
  const long long* kin = code;
  const int* vin = nullptr;
  for (int ps = 0; ps < passes; ++ps) {
    const bool to_final = ((passes - 1 - ps) % 2) == 0;
    long long* kout = to_final ? skeys : kb;
    int* vout = to_final ? order : vb;
    // both buffers use the same row stride when slot_stride == n_max; enforce it
    LOTUS_CHECK_ARG(slot_stride == (long)n_max, "lotus_fe_sort: slot_stride must equal n_max");
    LOTUS_LAUNCH(rs_hist_kernel, dim3(ntiles4 / 4, 4), dim3(256), 0, st, kin, (long)n_max, n_ptr, ps * 8, ntiles4, hist);
    LOTUS_LAUNCH(rs_scan_kernel, dim3(4), dim3(1024), 0, st, hist, ntiles4);
    LOTUS_LAUNCH(rs_scatter_kernel, dim3(ntiles, 4), dim3(64), 0, st, kin, vin, kout, vout, (long)n_max, n_ptr,
                       ps * 8, ntiles4, hist, ps == 0 ? 1 : 0);
    kin = kout;
    vin = vout;
  }
  if (inverse)
    LOTUS_LAUNCH(fe_inverse_kernel, dim3(cdiv(n_max, 256), 4), dim3(256), 0, st, order, inverse, (long)n_max, n_ptr);
  LOTUS_LAUNCH_CHECK("lotus_fe_sort");
  return LOTUS_OK;
}

// Grid pooling tables from the parent level's slot-0 sort.  Outputs (all sized by n_max):
// cluster[n] (child id of every parent), seg_start[n_child + 1] (CSR into order0), n_child (device
// int), child code/grid/batch and per-cloud child counts.
int lotus_fe_pool(const long long* pcode, const long long* skey0, const int* order0, const int* pgrid,
                  const int* pbatch, const int* n_ptr, int n_max, const int* perm4, int nbatch, int* cluster,
                  int* seg_start, int* n_child, long long* ccode, int* cgrid, int* cbatch, int* ccounts,
                  int* n_dup, void* stream) {
  LOTUS_CHECK_ARG(pcode && skey0 && order0 && pgrid && pbatch && n_ptr && perm4 && cluster && seg_start && n_child &&
                      ccode && cgrid && cbatch && ccounts,
                  "lotus_fe_pool: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int4 pm = make_int4(perm4[0], perm4[1], perm4[2], perm4[3]);
  (void)hipMemsetAsync(ccounts, 0, (size_t)nbatch * sizeof(int), st);
  // cbatch doubles as the per-block head-count scratch until fe_pool_child_kernel overwrites it
  const int nblk = cdiv(n_max, 1024);
  LOTUS_CHECK_ARG(nblk <= n_max, "lotus_fe_pool: n_max too small");
  LOTUS_LAUNCH(fe_pool_count_kernel, dim3(nblk), dim3(1024), 0, st, skey0, n_ptr, cbatch, n_dup);
  LOTUS_LAUNCH(fe_pool_scan_kernel, dim3(nblk), dim3(1024), 0, st, skey0, order0, n_ptr, (const int*)cbatch, cluster,
                     seg_start, n_child);
  LOTUS_LAUNCH(fe_pool_child_kernel, dim3(cdiv(n_max, 256)), dim3(256), 0, st, pcode, (long)n_max, order0,
                     seg_start, pgrid, pbatch, n_child, pm, ccode, (long)n_max, cgrid, cbatch, ccounts, 1);
  LOTUS_LAUNCH_CHECK("lotus_fe_pool");
  return LOTUS_OK;
}

int lotus_fe_pool_coord(const float* pcoord, const int* order0, const int* seg_start, int n_child, float* ccoord,
                        void* stream) {
  if (n_child == 0) return LOTUS_OK;
  LOTUS_LAUNCH(fe_pool_coord_kernel, dim3(cdiv(n_child, 256)), dim3(256), 0, (hipStream_t)stream, pcoord, order0,
                     seg_start, n_child, ccoord);
  LOTUS_LAUNCH_CHECK("lotus_fe_pool_coord");
  return LOTUS_OK;
}

// Patch gather table for SerializedAttention: gidx[npad], owner[npad].  off/offp: int32 [B+1] device.
int lotus_fe_patch(const int* order, const int* off, const int* offp, int B, int K, int npad, int* gidx, int* owner,
                   int* kext, int* ext_pos, void* stream) {
  LOTUS_CHECK_ARG(order && off && offp && gidx && owner && B > 0 && K > 0, "lotus_fe_patch: bad arguments");
  if (npad == 0) return LOTUS_OK;
  LOTUS_LAUNCH(fe_patch_kernel, dim3(cdiv(npad, 256)), dim3(256), 0, (hipStream_t)stream, order, off, offp, B, K,
                     npad, gidx, owner, kext, ext_pos);
  LOTUS_LAUNCH_CHECK("lotus_fe_patch");
  return LOTUS_OK;
}

size_t lotus_fe_neighbours_workspace(int n) {
  size_t cap = 1;
  while (cap < (size_t)2 * (n > 0 ? n : 1)) cap <<= 1;
  return cap * sizeof(HtSlot);
}

// Tap plan of a level (lotus_fe_tap_plan): plan = [cnt: 32 ints | in: 27 x n64 | pos: 27 x n], n64 = n rounded up to 64.
// Block t compacts the rows that have a neighbour at tap t, walking them in processing order `rowidx` (curve order: the
// gathered rows of consecutive pairs are close in memory): in[t * n64 + q] = neighbour row of the q-th such row,
// pos[t * n + i] = t * n64 + q (or -1), cnt[t] = their number; in[] is padded with row 0 up to the next multiple of 64 so that
// a 64-row tile of the grouped product only ever gathers valid rows.  Fixed order -> deterministic.
static __global__ __launch_bounds__(1024) void fe_tap_plan_kernel(const int* __restrict__ nbr, const int* __restrict__ rowidx, int n,
                                                           int n64, int* __restrict__ plan) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int t = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int* cnt = plan;
  int* in = plan + 32 + (long)t * n64;
  int* pos = plan + 32 + 27L * n64 + (long)t * n;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int r0 = 0; r0 < n; r0 += 1024) {
    const int r = r0 + tid;
    const int i = r < n ? (rowidx ? rowidx[r] : r) : -1;
    const int j = i >= 0 ? nbr[(long)t * n + i] : -1;
    const unsigned long long b = __ballot(j >= 0);
    if (lane == 0) wsum[wave] = __popcll(b);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (j >= 0) {
      const int q = off + __popcll(b & ((1ull << lane) - 1ull));
      in[q] = j;
      pos[i] = t * n64 + q;
    } else if (i >= 0) {
      pos[i] = -1;
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      base_s += tot;
    }
    __syncthreads();
  }
  const int c = base_s;
  if (tid == 0) cnt[t] = c;
  if (c + tid < ((c + 63) & ~63)) in[c + tid] = 0;
}

// nbr int32 [ksize^3][n] (tap-major), -1 = no active site.
int lotus_fe_neighbours(const int* grid, const int* batch, int n, int ksize, int* nbr, void* workspace,
                        size_t workspace_bytes, void* stream) {
  LOTUS_CHECK_ARG(grid && batch && nbr && n >= 0 && (ksize == 3 || ksize == 5), "lotus_fe_neighbours: bad arguments");
  if (n == 0) return LOTUS_OK;
  size_t cap = 1;
  while (cap < (size_t)2 * n) cap <<= 1;
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= cap * sizeof(HtSlot) && ((uintptr_t)workspace) % 16 == 0,
                  "lotus_fe_neighbours: workspace too small or not 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  HtSlot* ht = (HtSlot*)workspace;
  (void)hipMemsetAsync(ht, 0xff, cap * sizeof(HtSlot), st);
  LOTUS_LAUNCH(fe_hash_build_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, grid, batch, n, ht, (unsigned)(cap - 1));
  LOTUS_LAUNCH(fe_neighbour_kernel, dim3(cdiv(n, 256), ksize * ksize * ksize), dim3(256), 0, st, grid, batch, n,
                     ksize, ht, (unsigned)(cap - 1), nbr);
  LOTUS_LAUNCH_CHECK("lotus_fe_neighbours");
  return LOTUS_OK;
}

size_t lotus_fe_tap_plan_ints(int n) { return 32 + 27 * (size_t)((n + 63) / 64 * 64) + 27 * (size_t)n; }

int lotus_fe_tap_plan(const int* nbr27, const int* rowidx, int n, int* plan, void* stream) {
  LOTUS_CHECK_ARG(nbr27 && plan && n >= 0, "lotus_fe_tap_plan: bad arguments");
  if (n == 0) return LOTUS_OK;
  LOTUS_LAUNCH(fe_tap_plan_kernel, dim3(27), dim3(1024), 0, (hipStream_t)stream, nbr27, rowidx, n, (n + 63) / 64 * 64, plan);
  LOTUS_LAUNCH_CHECK("lotus_fe_tap_plan");
  return LOTUS_OK;
}

}  // extern "C"
