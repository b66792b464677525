// Level scheduler: dense [B,N,N] fp32 adjacency -> per-level node rows, CSR/CSC.
// Integer work only; every output is a deterministic function of the input
// (rows are ordered by (level, graph, node index)), so it is bit-exact against
// oracle/levels.py.  One 64-lane wave per graph; adjacency rows live in LDS as
// 128-bit masks.
#include "sched.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int SCAN_TILE = 2048;  // ints per 256-thread block

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Carver {
  char* base; size_t off;
  template <typename T> T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off = align_up(off + n * sizeof(T));
    return p;
  }
};

SchedView carve(void* base, int B, int N, size_t* total) {
  SchedView v;
  v.B = B; v.N = N;
  v.RM = B * N;
  v.EM = (int)((long long)B * N * (N - 1) / 2);
  if (v.EM < 1) v.EM = 1;
  Carver c{(char*)base, 0};
  v.rowbits = c.take<uint32_t>((size_t)B * N * DVAE_BITW);
  v.colbits = c.take<uint32_t>((size_t)B * N * DVAE_BITW);
  v.err = c.take<int>(64);
  v.scan_tmp = c.take<int>((size_t)cdiv(v.RM + N + 2, SCAN_TILE) + 64);
  v.info = c.take<int>(DVAE_SCHED_INFO_INTS(N));
  v.row_type = c.take<int>(v.RM);
  v.a_pos = c.take<int>((size_t)v.EM + 1);
  v.a_rows = c.take<int>(v.EM);
  v.a_skip = c.take<int>(v.EM);
  for (int d = 0; d < 2; ++d) {
    SchedDir& s = v.dir[d];
    s.level = c.take<int>(v.RM);
    s.node2row = c.take<int>(v.RM);
    s.pos = c.take<int>((size_t)N * B + 1);
    s.row_graph = c.take<int>(v.RM);
    s.row_node = c.take<int>(v.RM);
    s.csr_ptr = c.take<int>(v.RM + 1);
    s.csc_ptr = c.take<int>(v.RM + 1);
    s.csr_idx = c.take<int>(v.EM);
    s.csc_idx = c.take<int>(v.EM);
    if (d == 0) {
      s.csc_prow = c.take<int>(v.EM);
      s.csc_cnt = c.take<int>(v.EM);
      s.prow_node = c.take<int>((size_t)v.RM + v.EM);
    } else {
      s.csc_prow = nullptr; s.csc_cnt = nullptr; s.prow_node = nullptr;
    }
  }
  {
    SchedDir& s = v.enc;
    s.level = v.dir[0].level;      // a value's level is the level of each of its nodes
    s.node2row = c.take<int>(v.RM);
    s.pos = c.take<int>((size_t)N * B + 1);
    s.row_graph = c.take<int>(v.RM);
    s.row_node = c.take<int>(v.RM);
    s.csr_ptr = c.take<int>(v.RM + 1);
    s.csc_ptr = c.take<int>(v.RM + 1);
    s.csr_idx = c.take<int>(v.EM);
    s.csc_idx = c.take<int>(v.EM);
    s.csc_prow = nullptr; s.csc_cnt = nullptr; s.prow_node = nullptr;
    const size_t m = (size_t)std::max(v.RM, v.EM);
    v.vh = c.take<unsigned long long>((size_t)2 * v.RM);
    v.skey[0] = c.take<unsigned long long>(v.RM);
    v.skey[1] = c.take<unsigned long long>(v.RM);
    v.sval[0] = c.take<int>(m);
    v.sval[1] = c.take<int>(m);
    v.pkey[0] = c.take<int>(v.EM);
    v.pkey[1] = c.take<int>(v.EM);
    v.rep = c.take<int>(v.RM);
    v.gidx = c.take<int>((size_t)v.RM + 1);
    v.gfirst = c.take<int>(v.RM);
    // the radix sorts' scratch: a second copy of keys and values plus digit counters; dvae_sched_types asks the
    // library what it needs and leaves the rows unshared if this does not cover it
    v.sort_tmp_bytes = 16 * m + ((size_t)4 << 20);
    v.sort_tmp = c.take<char>(v.sort_tmp_bytes);
  }
  if (total) *total = c.off;
  return v;
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ bool bit_of(const uint32_t* w, int u) { return (w[u >> 5] >> (u & 31)) & 1u; }

// ---- K1: bit matrices, levels in both directions, per-(level,graph) counts --
__global__ __launch_bounds__(64) void sched_levels_kernel(const float* __restrict__ g,
                                                          const int* __restrict__ sizes, SchedView s,
                                                          int want_rev) {
  __shared__ uint32_t rowb[DVAE_MAX_N][DVAE_BITW];
  __shared__ uint32_t colb[DVAE_MAX_N][DVAE_BITW];
  __shared__ int lf[DVAE_MAX_N], lr[DVAE_MAX_N];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int N = s.N, B = s.B;
  int n = sizes ? sizes[b] : N;
  bool bad = false;
  if (n < 0 || n > N) { bad = true; n = max(0, min(n, N)); }
  for (int i = lane; i < DVAE_MAX_N * DVAE_BITW; i += 64) {
    (&rowb[0][0])[i] = 0u;
    (&colb[0][0])[i] = 0u;
  }
  __syncthreads();
  const float* gb = g + (size_t)b * N * N;
  const int total = N * N;
  for (int e = lane; e < total; e += 64) {
    const float v = gb[e];
    if (v != 0.f) {
      const int i = e / N, u = e - i * N;
      if (v != 1.f || u >= i || i >= n) bad = true;
      else {
        atomicOr(&rowb[i][u >> 5], 1u << (u & 31));
        atomicOr(&colb[u][i >> 5], 1u << (i & 31));
      }
    }
  }
  __syncthreads();
  // forward levels: ascending index order
  for (int i = 0; i < n; ++i) {
    int v = 0;
    if (lane < i && bit_of(rowb[i], lane)) v = lf[lane] + 1;
    if (lane + 64 < i && bit_of(rowb[i], lane + 64)) v = max(v, lf[lane + 64] + 1);
    v = wave_max(v);
    if (lane == 0) lf[i] = v;
    __syncthreads();
  }
  // reverse levels: descending index order, successors act as predecessors
  if (want_rev) {
    for (int i = n - 1; i >= 0; --i) {
      int v = 0;
      if (lane > i && lane < n && bit_of(colb[i], lane)) v = lr[lane] + 1;
      if (lane + 64 > i && lane + 64 < n && bit_of(colb[i], lane + 64)) v = max(v, lr[lane + 64] + 1);
      v = wave_max(v);
      if (lane == 0) lr[i] = v;
      __syncthreads();
    }
  }
  for (int i = lane; i < N; i += 64) {
    s.dir[0].level[b * N + i] = i < n ? lf[i] : -1;
    if (want_rev) s.dir[1].level[b * N + i] = i < n ? lr[i] : -1;
#pragma unroll
    for (int w = 0; w < DVAE_BITW; ++w) {
      s.rowbits[((size_t)b * N + i) * DVAE_BITW + w] = rowb[i][w];
      s.colbits[((size_t)b * N + i) * DVAE_BITW + w] = colb[i][w];
    }
  }
  // histogram: cnt[L*B + b] = nodes of this graph at level L
  for (int L = lane; L < N; L += 64) {
    int cf = 0, cr = 0;
    for (int i = 0; i < n; ++i) {
      cf += (lf[i] == L);
      if (want_rev) cr += (lr[i] == L);
    }
    s.dir[0].pos[(size_t)L * B + b] = cf;
    if (want_rev) s.dir[1].pos[(size_t)L * B + b] = cr;
  }
  // a single lane may have seen the bad entry
  if (__any(bad) && lane == 0) atomicOr(s.err, 1);
}

// ---- exclusive scan of n ints (out may alias in), out[n] = total ------------
__global__ __launch_bounds__(256) void scan_tile_kernel(const int* in, int* out, int n, int* tile_sums) {
  __shared__ int wsum[4];
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
  int v[8];
  int t = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] = (base + k) < n ? in[base + k] : 0;
    t += v[k];
  }
  // inclusive scan of t across the block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = t;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(inc, o);
    if (lane >= o) inc += y;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  int run = woff + inc - t;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if ((base + k) < n) out[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 255) tile_sums[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(int* tile_sums, int ntiles, int* total_out) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += 1024) {
    const int i = base + threadIdx.x;
    const int t = i < ntiles ? tile_sums[i] : 0;
    int inc = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(inc, o);
      if (lane >= o) inc += y;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = carry_s;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    if (i < ntiles) tile_sums[i] = woff + inc - t;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ __launch_bounds__(256) void scan_add_kernel(int* out, int n, const int* tile_sums) {
  const int off = tile_sums[blockIdx.x];
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if ((base + k) < n) out[base + k] += off;
}

int exclusive_scan(const int* in, int* out, int n, int* tmp, hipStream_t st) {
  const int ntiles = cdiv(n, SCAN_TILE);
  hipLaunchKernelGGL(scan_tile_kernel, dim3(ntiles), dim3(256), 0, st, in, out, n, tmp);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, tmp, ntiles, out + n);
  hipLaunchKernelGGL(scan_add_kernel, dim3(ntiles), dim3(256), 0, st, out, n, tmp);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

// ---- K3: node rows ------------------------------------------------------------
__global__ __launch_bounds__(64) void sched_assign_kernel(SchedView s, int d) {
  __shared__ int lv[DVAE_MAX_N];
  const int b = blockIdx.x, lane = threadIdx.x, N = s.N, B = s.B;
  SchedDir& S = s.dir[d];
  for (int i = lane; i < N; i += 64) lv[i] = S.level[b * N + i];
  __syncthreads();
  const uint32_t* predb = d == 0 ? s.rowbits : s.colbits;
  const uint32_t* succb = d == 0 ? s.colbits : s.rowbits;
  for (int i = lane; i < N; i += 64) {
    const int L = lv[i];
    int row = -1;
    if (L >= 0) {
      int rank = 0;
      for (int j = 0; j < i; ++j) rank += (lv[j] == L);
      row = S.pos[(size_t)L * B + b] + rank;
      S.row_graph[row] = b;
      S.row_node[row] = i;
      int deg = 0, out = 0;
#pragma unroll
      for (int w = 0; w < DVAE_BITW; ++w) {
        deg += __popc(predb[((size_t)b * N + i) * DVAE_BITW + w]);
        out += __popc(succb[((size_t)b * N + i) * DVAE_BITW + w]);
      }
      S.csr_ptr[row] = deg;   // scanned in place afterwards
      S.csc_ptr[row] = out;
      if (d == 0 && L == 0 && i != 0) atomicOr(s.err + 3, 1);   // a source other than node 0
    }
    S.node2row[b * N + i] = row;
  }
}

// ---- K5: CSR / CSC contents ------------------------------------------------------
__global__ __launch_bounds__(64) void sched_fill_kernel(SchedView s, int d) {
  const int b = blockIdx.x, lane = threadIdx.x, N = s.N;
  SchedDir& S = s.dir[d];
  const uint32_t* predb = d == 0 ? s.rowbits : s.colbits;
  const uint32_t* succb = d == 0 ? s.colbits : s.rowbits;
  for (int i = lane; i < N; i += 64) {
    const int row = S.node2row[b * N + i];
    if (row < 0) continue;
    const uint32_t* pm = predb + ((size_t)b * N + i) * DVAE_BITW;
    int p = S.csr_ptr[row];
    const int deg = S.csr_ptr[row + 1] - p;
    for (int w = DVAE_BITW - 1; w >= 0; --w) {
      uint32_t m = pm[w];
      while (m) {
        const int bit = 31 - __clz(m);
        m &= ~(1u << bit);
        S.csr_idx[p++] = S.node2row[b * N + w * 32 + bit];
      }
    }
    if (d == 0) {
      // decoder prefix rows: k = 0 is the node row itself, k >= 1 is R + edge index
      const int R = S.pos[(size_t)s.N * s.B];
      S.prow_node[row] = row;
      for (int k = 0; k < deg; ++k) S.prow_node[R + S.csr_ptr[row] + k] = row;
      // add_edge scores pair (i, j) on the prefix that holds the predecessors above j
      // (dvae_model.py:210-220): the full prefix is scored only when node 0 is not a predecessor
      for (int k = 0; k < deg; ++k) s.a_pos[S.csr_ptr[row] + k] = (k == deg - 1 && (pm[0] & 1u)) ? 0 : 1;
    }
    // successors of node i, ascending index
    const uint32_t* sm = succb + ((size_t)b * N + i) * DVAE_BITW;
    int q = S.csc_ptr[row];
    for (int w = 0; w < DVAE_BITW; ++w) {
      uint32_t m = sm[w];
      while (m) {
        const int bit = __ffs(m) - 1;
        m &= ~(1u << bit);
        const int j = w * 32 + bit;           // successor node index
        const int rj = S.node2row[b * N + j];
        S.csc_idx[q] = rj;
        if (d == 0) {
          // t = 1 + number of predecessors of j with index above i
          const uint32_t* pj = predb + ((size_t)b * N + j) * DVAE_BITW;
          int above = 0;
          for (int ww = 0; ww < DVAE_BITW; ++ww) {
            uint32_t mm = pj[ww];
            if (ww < (i >> 5)) mm = 0;
            else if (ww == (i >> 5)) mm &= ((i & 31) == 31) ? 0u : (~0u << ((i & 31) + 1));
            above += __popc(mm);
          }
          const int degj = S.csr_ptr[rj + 1] - S.csr_ptr[rj];
          const int t = above + 1;
          S.csc_prow[q] = S.pos[(size_t)s.N * s.B] + S.csr_ptr[rj] + t - 1;
          S.csc_cnt[q] = degj - t + 1;
        }
        ++q;
      }
    }
  }
}

// scored edge rows in ascending order -> a_rows, the others -> a_skip (a_pos: exclusive scan of the flags)
__global__ void sched_scored_kernel(SchedView s) {
  const int R = s.dir[0].pos[(size_t)s.N * s.B];
  const int E = s.dir[0].csr_ptr[s.RM];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    const int q = s.a_pos[e];
    if (s.a_pos[e + 1] != q) s.a_rows[q] = R + e;
    else s.a_skip[e - q] = R + e;
  }
}

__global__ void sched_info_kernel(SchedView s, int want_rev) {
  // single thread: tiny
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int N = s.N, B = s.B;
  int* info = s.info;
  const int R = s.dir[0].pos[(size_t)N * B];
  info[0] = *s.err;
  info[1] = R;
  info[2] = s.dir[0].csr_ptr[s.RM];
  for (int d = 0; d < 2; ++d) {
    int* ls = info + 8 + d * (N + 1);
    int nlev = 0;
    if (d == 0 || want_rev) {
      for (int L = 0; L < N; ++L) {
        const int st = s.dir[d].pos[(size_t)L * B];
        ls[L] = st;
        if (st < R) nlev = L + 1;
      }
    } else {
      for (int L = 0; L < N; ++L) ls[L] = R;
    }
    ls[N] = R;
    info[3 + d] = nlev;
  }
  // forward level starts in EDGE space: csr_ptr at each level's first node row, so
  // that decoder prefix rows of level L are [ls[L]+es[L], ls[L+1]+es[L+1])
  int* es = info + 8 + 2 * (N + 1);
  for (int L = 0; L <= N; ++L) es[L] = s.dir[0].csr_ptr[info[8 + L]];
  // the same starts counted in scored edge rows only (a_rows)
  int* as = info + 8 + 3 * (N + 1);
  for (int L = 0; L <= N; ++L) as[L] = s.a_pos[es[L]];
  info[5] = info[6] = 0;
  info[7] = s.err[3] == 0 ? 1 : 0;
  info[DVAE_SCHED_MERGE_AT(N)] = 0;   // shared encoder rows: dvae_sched_types
}

// node type of every forward node row; err[1] counts rows whose encoding is not one-hot,
// err[2] rows for which "sits in the last level" and "is the last node of its graph" disagree
__global__ void sched_types_kernel(SchedView s, const float* __restrict__ x, int C) {
  const SchedDir& S = s.dir[0];
  const int R = S.pos[(size_t)s.N * s.B];
  const int last_level = s.info[3] - 1;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
    const int b = S.row_graph[r], i = S.row_node[r];
    const float* xr = x + ((size_t)b * s.N + i) * C;
    int t = -1, ones = 0, other = 0;
    for (int c = 0; c < C; ++c) {
      const float v = xr[c];
      if (v == 1.f) { t = c; ++ones; }
      else if (v != 0.f) ++other;
    }
    if (ones != 1 || other != 0) { t = -1; atomicAdd(s.err + 1, 1); }
    s.row_type[r] = t;
    const bool last_node = i == s.N - 1 || S.node2row[b * s.N + i + 1] < 0;
    if (last_node != (S.level[b * s.N + i] == last_level)) atomicAdd(s.err + 2, 1);
  }
}

__global__ void sched_types_info_kernel(SchedView s) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    s.info[5] = s.err[1] == 0 ? 1 : 0;
    s.info[6] = s.err[2] == 0 ? 1 : 0;
  }
}


// ===========================================================================
// shared encoder rows (value numbering of the forward encoder sweep)
// ===========================================================================
// The state of node i (dvae_model.py:63-87) is a function of its type, its index (position one-hot of the
// gate/map input, :71) and the states of its predecessors, summed in descending index order -- nothing else.
// Nodes of different graphs that agree in all three hold the same state, and at ENAS sizes most nodes of the
// first levels do (8 types: 8 / 64 / ~1000 distinct states among the 4096 nodes with index 0 / 1 / 2).  The
// sweep therefore runs on one row per distinct VALUE; gradients reach a shared row from all its consumers, so
// every sum the backward pass forms is the sum over the nodes the row stands for.
//   1. a 128-bit hash of (index, type, hashes of the predecessors) per node, graph by graph;
//   2. nodes sorted by hash (stable radix sort: equal hashes keep ascending node id) -> groups; the
//      representative of a group is its first = lowest node id;
//   3. every node is CHECKED against its representative: same index, type, predecessor index set, and the
//      predecessors' representatives equal one by one -- by induction over the index the states are equal.
//      A failed check (a hash collision) clears the valid flag: the batch runs unshared;
//   4. value rows ordered by (level, graph, index) of the representative, CSR through node2row, and the
//      successor lists by a stable sort of the (predecessor row, row) pairs.
// The last node of a graph keeps a row of its own: it receives that graph's gradient (take_last).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
    v += ((unsigned long long)hi << 32) | lo;
  }
  return v;
}

__global__ __launch_bounds__(64) void merge_hash_kernel(SchedView s) {
  __shared__ unsigned long long h0[DVAE_MAX_N], h1[DVAE_MAX_N];
  const int b = blockIdx.x, lane = threadIdx.x, N = s.N;
  const SchedDir& S = s.dir[0];
  const uint32_t* rb = s.rowbits + (size_t)b * N * DVAE_BITW;
  int n = 0;
  for (int i = 0; i < N; ++i) n += S.level[b * N + i] >= 0;   // nodes [0, n) are present (sched_levels_kernel)
  for (int i = 0; i < n; ++i) {
    unsigned long long a0 = 0, a1 = 0;
    for (int u = lane; u < i; u += 64)
      if (bit_of(rb + i * DVAE_BITW, u)) {
        a0 += mix64(h0[u] ^ 0x6a09e667f3bcc908ull);
        a1 += mix64(h1[u] + 0xbb67ae8584caa73bull);
      }
    a0 = wave_sum_u64(a0);
    a1 = wave_sum_u64(a1);
    if (lane == 0) {
      const int t = s.row_type[S.node2row[b * N + i]];
      const unsigned long long base = ((unsigned long long)(unsigned)i << 32) | (unsigned)(t + 1);
      unsigned long long v0 = mix64(mix64(base) ^ a0), v1 = mix64(mix64(base + 0x3c6ef372fe94f82bull) + a1);
      if (i == n - 1) {   // never shared
        v0 = mix64(v0 ^ mix64(0xa54ff53a5f1d36f1ull + (unsigned)b));
        v1 = mix64(v1 + mix64(0x510e527fade682d1ull ^ (unsigned)b));
      }
      h0[i] = v0;
      h1[i] = v1;
    }
    __syncthreads();
  }
  for (int i = lane; i < N; i += 64) {
    const int id = b * N + i;
    s.vh[2 * (size_t)id] = i < n ? h0[i] : ~0ull;
    s.vh[2 * (size_t)id + 1] = i < n ? h1[i] : ~0ull;
    s.skey[0][id] = i < n ? h0[i] : ~0ull;   // absent nodes sort to the end
    s.sval[0][id] = id;
  }
}

__device__ __forceinline__ bool merge_boundary(const SchedView& s, int k) {
  const int id = s.sval[1][k];
  if (s.dir[0].level[id] < 0) return false;
  if (k == 0) return true;
  const int pv = s.sval[1][k - 1];
  return s.vh[2 * (size_t)id] != s.vh[2 * (size_t)pv] || s.vh[2 * (size_t)id + 1] != s.vh[2 * (size_t)pv + 1] ||
         s.dir[0].level[pv] < 0;
}

__global__ void merge_flags_kernel(SchedView s) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < s.RM; k += gridDim.x * blockDim.x)
    s.gidx[k] = merge_boundary(s, k) ? 1 : 0;
}

// after the exclusive scan: a group's first position holds its index, the others index + 1
__global__ void merge_first_kernel(SchedView s) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < s.RM; k += gridDim.x * blockDim.x)
    if (merge_boundary(s, k)) s.gfirst[s.gidx[k]] = s.sval[1][k];
}

__global__ void merge_rep_kernel(SchedView s) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < s.RM; k += gridDim.x * blockDim.x) {
    const int id = s.sval[1][k];
    if (s.dir[0].level[id] < 0) { s.rep[id] = -1; continue; }
    s.rep[id] = s.gfirst[merge_boundary(s, k) ? s.gidx[k] : s.gidx[k] - 1];
  }
}

// the check of step 3; err[4] |= 1 on any disagreement
__global__ void merge_check_kernel(SchedView s) {
  const int N = s.N;
  const SchedDir& S = s.dir[0];
  for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < s.RM; id += gridDim.x * blockDim.x) {
    if (S.level[id] < 0) continue;
    const int r = s.rep[id];
    const int b = id / N, i = id - b * N;
    const bool last = i == N - 1 || S.level[id + 1] < 0;
    bool ok = r >= 0 && r <= id && S.level[r] >= 0 && s.rep[r] == r;
    if (ok && r != id) {
      const int rb = r / N, ri = r - rb * N;
      const bool rlast = ri == N - 1 || S.level[r + 1] < 0;
      ok = !last && !rlast && ri == i && s.row_type[S.node2row[id]] == s.row_type[S.node2row[r]] &&
           S.level[id] == S.level[r];
      const uint32_t* pa = s.rowbits + (size_t)id * DVAE_BITW;
      const uint32_t* pb = s.rowbits + (size_t)r * DVAE_BITW;
      for (int w = 0; ok && w < DVAE_BITW; ++w) {
        uint32_t m = pa[w];
        ok = m == pb[w];
        while (ok && m) {
          const int u = w * 32 + __ffs(m) - 1;
          m &= m - 1;
          ok = s.rep[b * N + u] == s.rep[rb * N + u];
        }
      }
    }
    if (!ok) atomicOr(s.err + 4, 1);
  }
}

// representatives per (level, graph), as sched_levels_kernel counts nodes
__global__ __launch_bounds__(64) void merge_count_kernel(SchedView s) {
  const int b = blockIdx.x, lane = threadIdx.x, N = s.N, B = s.B;
  for (int L = lane; L < N; L += 64) {
    int c = 0;
    for (int i = 0; i < N; ++i) {
      const int id = b * N + i;
      c += (s.dir[0].level[id] == L && s.rep[id] == id);
    }
    s.enc.pos[(size_t)L * B + b] = c;
  }
}

__global__ __launch_bounds__(64) void merge_assign_kernel(SchedView s) {
  const int b = blockIdx.x, lane = threadIdx.x, N = s.N, B = s.B;
  SchedDir& M = s.enc;
  for (int i = lane; i < N; i += 64) {
    const int id = b * N + i, L = s.dir[0].level[id];
    int row = -1;
    if (L >= 0 && s.rep[id] == id) {
      int rank = 0;
      for (int j = 0; j < i; ++j) rank += (s.dir[0].level[b * N + j] == L && s.rep[b * N + j] == b * N + j);
      row = M.pos[(size_t)L * B + b] + rank;
      M.row_graph[row] = b;
      M.row_node[row] = i;
      int deg = 0;
#pragma unroll
      for (int w = 0; w < DVAE_BITW; ++w) deg += __popc(s.rowbits[(size_t)id * DVAE_BITW + w]);
      M.csr_ptr[row] = deg;   // scanned in place afterwards
    }
    M.node2row[id] = row;
  }
}

__global__ void merge_map_kernel(SchedView s) {
  for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < s.RM; id += gridDim.x * blockDim.x) {
    const int r = s.rep[id];
    if (r >= 0 && r != id) s.enc.node2row[id] = s.enc.node2row[r];
  }
}

// predecessor rows of every value row (descending node index, as dir[0]) and the (predecessor row, row) pairs
__global__ void merge_fill_kernel(SchedView s) {
  const int N = s.N;
  SchedDir& M = s.enc;
  for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < s.RM; id += gridDim.x * blockDim.x) {
    if (s.rep[id] != id) continue;
    const int b = id / N, row = M.node2row[id];
    int p = M.csr_ptr[row];
    const uint32_t* pm = s.rowbits + (size_t)id * DVAE_BITW;
    for (int w = DVAE_BITW - 1; w >= 0; --w) {
      uint32_t m = pm[w];
      while (m) {
        const int bit = 31 - __clz(m);
        m &= ~(1u << bit);
        const int u = M.node2row[b * N + w * 32 + bit];
        M.csr_idx[p] = u;
        s.pkey[0][p] = u;
        s.sval[0][p] = row;
        ++p;
      }
    }
  }
}

// csc_ptr[u] = first sorted pair whose predecessor row is >= u
__global__ void merge_csc_ptr_kernel(SchedView s) {
  const int* key = s.pkey[1];
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u <= s.RM; u += gridDim.x * blockDim.x) {
    int lo = 0, hi = s.EM;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (key[mid] < u) lo = mid + 1; else hi = mid;
    }
    s.enc.csc_ptr[u] = lo;
  }
}

__global__ void merge_info_kernel(SchedView s, int ran) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int N = s.N, B = s.B;
  int* mi = s.info + DVAE_SCHED_MERGE_AT(N);
  if (!ran) { mi[0] = 0; return; }
  mi[0] = (s.err[1] == 0 && s.err[4] == 0) ? 1 : 0;
  mi[1] = s.enc.pos[(size_t)N * B];
  mi[2] = s.enc.csr_ptr[s.RM];
  mi[3] = 0;
  for (int L = 0; L < N; ++L) mi[4 + L] = s.enc.pos[(size_t)L * B];
  mi[4 + N] = mi[1];
  for (int L = 0; L <= N; ++L) mi[5 + N + L] = s.enc.csc_ptr[mi[4 + L]];   // successor links before each level
}

int g_merge_min_slots = getenv("DVAE_NO_MERGE") ? -1 : getenv("DVAE_MERGE_MIN") ? atoi(getenv("DVAE_MERGE_MIN")) : 8192;

// *ran = 1 when the value rows were built (0: the sorts' scratch would not fit -- rows stay unshared)
int merge_rows(SchedView& s, hipStream_t st, int* ran) {
  *ran = 0;
  const int B = s.B, N = s.N;
  const int grid_r = std::min(cdiv(s.RM, 256), 2048);
  // the sorts' scratch must fit what the schedule reserved for them
  size_t need1 = 0, need2 = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, need1, s.skey[0], s.skey[1], s.sval[0], s.sval[1], s.RM, 0, 64, st) != hipSuccess ||
      hipcub::DeviceRadixSort::SortPairs(nullptr, need2, s.pkey[0], s.pkey[1], s.sval[0], s.enc.csc_idx, s.EM, 0, 31, st) != hipSuccess ||
      std::max(need1, need2) > s.sort_tmp_bytes)
    return DVAE_OK;
  DVAE_CHECK_HIP(hipMemsetAsync(s.err + 4, 0, sizeof(int), st));
  hipLaunchKernelGGL(merge_hash_kernel, dim3(B), dim3(64), 0, st, s);
  DVAE_LAUNCH_CHECK();
  size_t tb = s.sort_tmp_bytes;
  DVAE_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(s.sort_tmp, tb, s.skey[0], s.skey[1], s.sval[0], s.sval[1], s.RM, 0, 64, st));
  hipLaunchKernelGGL(merge_flags_kernel, dim3(grid_r), dim3(256), 0, st, s);
  DVAE_LAUNCH_CHECK();
  DVAE_TRY(exclusive_scan(s.gidx, s.gidx, s.RM, s.scan_tmp, st));
  hipLaunchKernelGGL(merge_first_kernel, dim3(grid_r), dim3(256), 0, st, s);
  hipLaunchKernelGGL(merge_rep_kernel, dim3(grid_r), dim3(256), 0, st, s);
  hipLaunchKernelGGL(merge_check_kernel, dim3(grid_r), dim3(256), 0, st, s);
  hipLaunchKernelGGL(merge_count_kernel, dim3(B), dim3(64), 0, st, s);
  DVAE_LAUNCH_CHECK();
  DVAE_TRY(exclusive_scan(s.enc.pos, s.enc.pos, N * B, s.scan_tmp, st));
  DVAE_CHECK_HIP(hipMemsetAsync(s.enc.csr_ptr, 0, sizeof(int) * (s.RM + 1), st));
  DVAE_CHECK_HIP(hipMemsetAsync(s.enc.row_graph, 0, sizeof(int) * s.RM, st));   // rows beyond the last value row name node (0,0)
  DVAE_CHECK_HIP(hipMemsetAsync(s.enc.row_node, 0, sizeof(int) * s.RM, st));
  hipLaunchKernelGGL(merge_assign_kernel, dim3(B), dim3(64), 0, st, s);
  hipLaunchKernelGGL(merge_map_kernel, dim3(grid_r), dim3(256), 0, st, s);
  DVAE_LAUNCH_CHECK();
  DVAE_TRY(exclusive_scan(s.enc.csr_ptr, s.enc.csr_ptr, s.RM, s.scan_tmp, st));
  DVAE_CHECK_HIP(hipMemsetAsync(s.pkey[0], 0x7f, sizeof(int) * (size_t)s.EM, st));   // unused slots sort last
  hipLaunchKernelGGL(merge_fill_kernel, dim3(grid_r), dim3(256), 0, st, s);
  DVAE_LAUNCH_CHECK();
  tb = s.sort_tmp_bytes;
  DVAE_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(s.sort_tmp, tb, s.pkey[0], s.pkey[1], s.sval[0], s.enc.csc_idx, s.EM, 0, 31, st));
  hipLaunchKernelGGL(merge_csc_ptr_kernel, dim3(std::min(cdiv(s.RM + 1, 256), 2048)), dim3(256), 0, st, s);
  DVAE_LAUNCH_CHECK();
  *ran = 1;
  return DVAE_OK;
}

}  // namespace

SchedView sched_view(void* base, int B, int N) { return carve(base, B, N, nullptr); }

extern "C" int dvae_sched_types(const float* x, int B, int N, int C, void* sched, int* info_host, void* stream) {
  DVAE_REQUIRE(x && sched && info_host, "sched_types: null pointer");
  DVAE_REQUIRE(B > 0 && N > 0 && N <= DVAE_MAX_N && C > 0 && C < DVAE_MAX_C, "sched_types: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  SchedView s = sched_view(sched, B, N);
  DVAE_CHECK_HIP(hipMemsetAsync(s.err + 1, 0, 2 * sizeof(int), st));
  const int grid = std::min(cdiv(B * N, 256), 2048);
  hipLaunchKernelGGL(sched_types_kernel, dim3(grid), dim3(256), 0, st, s, x, C);
  hipLaunchKernelGGL(sched_types_info_kernel, dim3(1), dim3(64), 0, st, s);
  DVAE_LAUNCH_CHECK();
  DVAE_CHECK_HIP(hipMemcpyAsync(info_host + 5, s.info + 5, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  // shared encoder rows: worth their build from a few thousand nodes up (dvae_enc_merge_tune)
  int ran = 0;
  if (g_merge_min_slots >= 0 && s.RM >= g_merge_min_slots) DVAE_TRY(merge_rows(s, st, &ran));
  hipLaunchKernelGGL(merge_info_kernel, dim3(1), dim3(64), 0, st, s, ran);
  DVAE_LAUNCH_CHECK();
  DVAE_CHECK_HIP(hipMemcpyAsync(info_host + DVAE_SCHED_MERGE_AT(N), s.info + DVAE_SCHED_MERGE_AT(N), (2 * N + 6) * sizeof(int),
                                hipMemcpyDeviceToHost, st));
  return DVAE_OK;
}

extern "C" int dvae_enc_merge_tune(int min_node_slots) {
  g_merge_min_slots = min_node_slots;
  return DVAE_OK;
}

// every row / edge index of a schedule is an int32: node rows B*N, candidate pairs B*N*(N-1)/2 and the
// decoder's prefix rows (their sum) must all stay below 2^31
bool sched_sizes_ok(int B, int N) {
  if (B <= 0 || N <= 0 || N > DVAE_MAX_N) return false;
  const long long rows = (long long)B * N, pairs = rows * (N - 1) / 2;
  return rows < (1ll << 30) && rows + pairs < (1ll << 31) - 64;
}

extern "C" size_t dvae_sched_bytes(int B, int N) {
  if (!sched_sizes_ok(B, N)) return 0;   // dvae_sched_build reports the reason
  size_t total = 0;
  carve(nullptr, B, N, &total);
  return total;
}

extern "C" int dvae_sched_build(const float* g, const int* sizes, int B, int N, int want_reverse,
                                void* sched, int* info_host, void* stream) {
  DVAE_REQUIRE(B > 0 && N > 0 && N <= DVAE_MAX_N, "sched: need B>0 and 0<N<=%d (got B=%d N=%d)",
               DVAE_MAX_N, B, N);
  DVAE_REQUIRE(sched_sizes_ok(B, N), "sched: batch too large for int32 row indices (B*N < 2^30 and B*N*(N+1)/2 < 2^31; got B=%d N=%d)", B, N);
  DVAE_REQUIRE(g && sched && info_host, "sched: null pointer");
  hipStream_t st = (hipStream_t)stream;
  SchedView s = sched_view(sched, B, N);
  DVAE_CHECK_HIP(hipMemsetAsync(s.err, 0, sizeof(int), st));
  DVAE_CHECK_HIP(hipMemsetAsync(s.err + 3, 0, sizeof(int), st));
  DVAE_CHECK_HIP(hipMemsetAsync(s.a_pos, 0, sizeof(int) * ((size_t)s.EM + 1), st));
  hipLaunchKernelGGL(sched_levels_kernel, dim3(B), dim3(64), 0, st, g, sizes, s, want_reverse);
  DVAE_LAUNCH_CHECK();
  for (int d = 0; d < (want_reverse ? 2 : 1); ++d) {
    SchedDir& S = s.dir[d];
    DVAE_TRY(exclusive_scan(S.pos, S.pos, N * B, s.scan_tmp, st));
    DVAE_CHECK_HIP(hipMemsetAsync(S.csr_ptr, 0, sizeof(int) * (s.RM + 1), st));
    DVAE_CHECK_HIP(hipMemsetAsync(S.csc_ptr, 0, sizeof(int) * (s.RM + 1), st));
    hipLaunchKernelGGL(sched_assign_kernel, dim3(B), dim3(64), 0, st, s, d);
    DVAE_LAUNCH_CHECK();
    // scan over RM ints; slot [RM] receives the total, so scan the first RM only
    DVAE_TRY(exclusive_scan(S.csr_ptr, S.csr_ptr, s.RM, s.scan_tmp, st));
    DVAE_TRY(exclusive_scan(S.csc_ptr, S.csc_ptr, s.RM, s.scan_tmp, st));
    hipLaunchKernelGGL(sched_fill_kernel, dim3(B), dim3(64), 0, st, s, d);
    DVAE_LAUNCH_CHECK();
    if (d == 0) {
      DVAE_TRY(exclusive_scan(s.a_pos, s.a_pos, s.EM, s.scan_tmp, st));
      hipLaunchKernelGGL(sched_scored_kernel, dim3(std::min(cdiv(s.EM, 256), 1024)), dim3(256), 0, st, s);
      DVAE_LAUNCH_CHECK();
    }
  }
  hipLaunchKernelGGL(sched_info_kernel, dim3(1), dim3(64), 0, st, s, want_reverse);
  DVAE_LAUNCH_CHECK();
  DVAE_CHECK_HIP(hipMemcpyAsync(info_host, s.info, sizeof(int) * DVAE_SCHED_INFO_INTS(N),
                                hipMemcpyDeviceToHost, st));
  return DVAE_OK;
}

extern "C" int dvae_sched_export(const void* sched, int B, int N, int which, int* out,
                                 size_t max_ints, void* stream) {
  SchedView s = sched_view(const_cast<void*>(sched), B, N);
  const int* src = nullptr;
  size_t n = 0;
  switch (which) {
    case 0: src = s.dir[0].level; n = s.RM; break;
    case 1: src = s.dir[1].level; n = s.RM; break;
    case 2: src = s.dir[0].node2row; n = s.RM; break;
    case 3: src = s.dir[1].node2row; n = s.RM; break;
    case 4: src = s.dir[0].csr_ptr; n = (size_t)s.RM + 1; break;
    case 5: src = s.dir[0].csr_idx; n = s.EM; break;
    case 6: src = s.dir[0].row_graph; n = s.RM; break;
    case 7: src = s.dir[0].row_node; n = s.RM; break;
    case 8: src = s.a_rows; n = s.EM; break;
    case 9: src = s.a_skip; n = s.EM; break;
    case 10: src = s.enc.node2row; n = s.RM; break;
    case 11: src = s.enc.row_graph; n = s.RM; break;
    case 12: src = s.enc.row_node; n = s.RM; break;
    case 13: src = s.enc.csr_ptr; n = (size_t)s.RM + 1; break;
    case 14: src = s.enc.csr_idx; n = s.EM; break;
    case 15: src = s.enc.csc_ptr; n = (size_t)s.RM + 1; break;
    case 16: src = s.enc.csc_idx; n = s.EM; break;
    case 17: src = s.rep; n = s.RM; break;
    default: dvae_set_error("sched_export: unknown array %d", which); return DVAE_ERR_ARG;
  }
  if (n > max_ints) n = max_ints;
  DVAE_CHECK_HIP(hipMemcpyAsync(out, src, n * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return DVAE_OK;
}
