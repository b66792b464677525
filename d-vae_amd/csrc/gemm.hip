// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   C[M][N] (+)= sum_k A(m,k) * B(n,k)  (+ bias[n], relu)
//
// Every dense contraction of the D-VAE path goes through this one kernel
// template: the GRU hidden GEMM, the gate/map GEMM, the add_vertex / add_edge
// MLP halves, and all of their backward products.  The three shapes differ only
// in which operand has its reduction dimension contiguous in memory:
//   NT  (forward,  Y = X W^T)      A[M][K]  B[N][K]    a_kmajor=0 b_kmajor=0
//   NN  (dX = dY W)                A[M][K]  B[K][N]    a_kmajor=0 b_kmajor=1
//   TN  (dW = dY^T X, K = rows)    A[K][M]  B[K][N]    a_kmajor=1 b_kmajor=1
// A "k-major" operand is staged into LDS exactly as it lies in memory ([k][m])
// and read back with ds_read_b32 (32 consecutive floats per half-wave, no
// conflicts); an "m-major" operand is staged as [m][k] with a 36-float row
// stride and read with ds_read_b128, each half-wave taking 4 consecutive k.
// The fp32 MFMA retires 32x32x2 in 64 cycles per SIMD, so LDS read width is
// nowhere near the limit; what matters is that no transposed copy of any
// weight or activation is ever written to HBM.
//
// The k order inside an 8-wide chunk is permuted (lane half h owns k = 4h..4h+3)
// consistently for A and B, which only reorders the fp32 summation.
#include <algorithm>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"
#include "x6.h"

static int g_splitk_blocks = 768;   // tuning knob: workgroups a split-K launch aims for
static int g_force_prio = -1;       // tuning knob: -1 as requested by the caller, 0/1 override
static int g_force_splits = 0;      // lab knob: split-K factor for every launch (0 = automatic)
static int g_mid_split = getenv("DVAE_NO_MID_SPLIT") ? 0 : 1;   // A-B knob for the mid-sized split-K rule below
extern "C" int dvae_gemm_force_splits(int v) { g_force_splits = v; return 0; }
static int g_dbg = 0;
extern "C" int dvae_gemm_dbg(int v) { g_dbg = v; return 0; }
static int g_variant = 0;           // lab knob (fp32-MFMA core): 0 default (k-tile 16), 3 k-tile 32
// x6 core, prefetch form of the 64x64 / 64x128 / 128x128 kernels: -1 by shape (pf_of below), 0 / 1 / 2 forced (A-B tests)
static int g_pf = getenv("DVAE_GEMM_PF") ? atoi(getenv("DVAE_GEMM_PF")) : -1;
extern "C" int dvae_gemm_variant(int v) { g_variant = v; return 0; }
extern "C" int dvae_gemm_pf(int v) { g_pf = (v >= 0 && v <= 2) ? v : -1; return 0; }
// tools/bench_gemm_pf.py, same box: the compiler weaves the next tile's split arithmetic into this tile's MFMA stream and
// waits for the loads it has only just issued after the FIRST product.  Holding the split behind the products
// (sched_barrier) gains 1.5-3 % on the large NT / NN shapes; two k-tiles in flight gain 11-16 % on the 4096-row NN products
// (64x64 tiles, latency-bound: 119 -> 132 TF at K = 1536) and 5 % on the NT ones; the weight-gradient (TN) shapes lose 3-5 %
// with either and keep the plain form.
static inline int pf_of(int a_kmajor, int b_kmajor, bool quarter) {
  if (g_pf >= 0 && g_pf <= 2) return g_pf;
  if (a_kmajor) return 0;
  (void)b_kmajor;
  return quarter ? 2 : 1;
}

namespace {

template <int KM, int ROWS, int BK>
struct Tile {
  static constexpr int LDM = BK + 4;              // row stride of an m-major LDS tile, floats
  static constexpr int KQ = BK / 4;               // float4 per tile row
  static constexpr int NV = ROWS * BK / 4 / 256;  // float4 per thread
  static_assert(NV >= 1, "tile too small for 256 threads");
  static constexpr int LDS_FLOATS = KM ? BK * ROWS : ROWS * LDM;
  float4 v[NV];
  size_t ro[NV];   // m-major: float offset of this thread's row(s), fixed over the k loop

  // m-major operand: where this thread's rows start; `map` (optional) sends row r of the
  // product to row map[r] of the operand (row gather).  Rows beyond `rows` clamp to the last one.
  __device__ __forceinline__ void rows_init(int ld, int r0, int rows, const int* __restrict__ map, int tid) {
    if (KM == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int r = min(r0 + tid / KQ + i * (256 / KQ), rows - 1);
        ro[i] = (size_t)(map ? map[r] : r) * ld;
      }
    }
  }

  // k-major operand whose reduction rows are gathered through `kmap` (reduction index k reads
  // memory row kmap[k]): kq holds the memory rows of the NEXT tile this thread loads, fetched one
  // tile ahead so that the dependent index load is off the critical path
  int kq[NV];
  __device__ __forceinline__ void kq_fetch(const int* __restrict__ kmap, int k0, int kend, int tid) {
    constexpr int RQ = ROWS / 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) kq[i] = kmap[min(k0 + tid / RQ + i * (256 / RQ), kend - 1)];
  }
  // full tile, no predicates (as load<false>); rows of the reduction come from kq
  __device__ __forceinline__ void load_gathered(const float* __restrict__ P, int ld, int r0, int rows, int tid) {
    constexpr int RQ = ROWS / 4;
    const int r = min(r0 + (tid % RQ) * 4, rows - 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(P + (size_t)kq[i] * ld + r);
  }
  // last (possibly partial) tiles: predicated, index loaded in place
  __device__ __forceinline__ void load_gathered_checked(const float* __restrict__ P, int ld, int r0, int rows, int k0,
                                                        int kend, const int* __restrict__ kmap, int tid) {
    constexpr int RQ = ROWS / 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      const int k = k0 + idx / RQ, r = r0 + (idx % RQ) * 4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < kend && r < rows) val = *reinterpret_cast<const float4*>(P + (size_t)kmap[k] * ld + r);
      v[i] = val;
    }
  }

  // rows: extent of the non-reduction dim; kend: end of this block's k range.
  // CHECK = false: the caller guarantees that the k range [k0, k0+BK) lies inside the
  // reduction; row indices beyond `rows` are clamped to the last valid row(s), whose
  // products land in output rows the epilogue never stores.  No predicates, no branches.
  template <bool CHECK>
  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int r0, int rows, int k0,
                                       int kend, int tid) {
    if (!CHECK) {
      if (KM == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(P + ro[i] + k0 + (tid % KQ) * 4);
      } else {
        constexpr int RQ = ROWS / 4;
        const int r = min(r0 + (tid % RQ) * 4, rows - 4);
        const float* base = P + (size_t)(k0 + tid / RQ) * ld + r;
        const size_t step = (size_t)(256 / RQ) * ld;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(base + i * step);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KM == 0) {
        const int row = idx / KQ, kq = idx % KQ;
        const int r = r0 + row, k = k0 + kq * 4;
        if (r < rows && k < kend) val = *reinterpret_cast<const float4*>(P + ro[i] + k);
      } else {
        constexpr int RQ = ROWS / 4;
        const int kr = idx / RQ, rq = idx % RQ;
        const int k = k0 + kr, r = r0 + rq * 4;
        if (k < kend && r < rows) val = *reinterpret_cast<const float4*>(P + (size_t)k * ld + r);
      }
      v[i] = val;
    }
  }

  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      if (KM == 0) {
        const int row = idx / KQ, kq = idx % KQ;
        *reinterpret_cast<float4*>(lds + row * LDM + kq * 4) = v[i];
      } else {
        constexpr int RQ = ROWS / 4;
        const int kr = idx / RQ, rq = idx % RQ;
        *reinterpret_cast<float4*>(lds + kr * ROWS + rq * 4) = v[i];
      }
    }
  }

  // fragment for one 32-row sub-tile at rbase, k-chunk kc (8 wide): 4 values,
  // value t belongs to k = kc*8 + 4*(lane>>5) + t.
  static __device__ __forceinline__ float4 frag(const float* lds, int rbase, int kc, int lane) {
    if (KM == 0) {
      return *reinterpret_cast<const float4*>(lds + (rbase + (lane & 31)) * LDM + kc * 8 +
                                              (lane >> 5) * 4);
    } else {
      const float* p = lds + (kc * 8 + (lane >> 5) * 4) * ROWS + rbase + (lane & 31);
      return make_float4(p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]);
    }
  }
};


struct GemmParams {
  const float* A; const float* B; float* C; const float* bias;
  int lda, ldb, ldc;
  int M, N, K;
  int relu, accumulate;
  int k_chunk;            // k range per blockIdx.y
  long long c_split;      // C offset (floats) per blockIdx.y
  int tiles_n, nblk;
  int dbg;                // lab only
  int prio;               // wave priority (s_setprio): the per-level chain outranks the bulk side-stream products
  const int* row_map;     // optional: row m of the product reads A row row_map[m] and writes C row row_map[m]
  const int* k_map;       // optional (both operands k-major, KG kernels): reduction index k reads row k_map[k]
};

// bijective XCD-aware remap: blocks that share blockIdx%8 (one XCD under
// round-robin placement) get a contiguous run of tiles, so an XCD's L2 keeps
// its own A panels.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int x = bid & 7, j = bid >> 3;
  const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + j;
}

// Epilogue shared by the GEMM kernels.  D map of the 32x32 MFMA with the n-operand first (C^T in the accumulator):
// m = lane&31, n = (reg&3) + 8*(reg>>2) + 4*(lane>>5)  ->  reg group g holds n = 8g + 4*(lane>>5) + 0..3
template <int TM, int TN>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[TM][TN], const GemmParams& p, float* C, int mbase, int nbase, int lane) {
  const bool vec = ((p.ldc | p.N) & 3) == 0 && ((uintptr_t)C & 15) == 0 &&
                   (!p.bias || ((uintptr_t)p.bias & 15) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = mbase + i * 32 + (lane & 31);
    if (row >= p.M) continue;
    float* crow = C + (size_t)(p.row_map ? p.row_map[row] : row) * p.ldc;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = nbase + j * 32 + 8 * g + 4 * (lane >> 5);
        if (col >= p.N) continue;
        float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        if (vec) {
          if (p.bias) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
          }
          float4* dst = reinterpret_cast<float4*>(crow + col);
          if (p.accumulate) {
            const float4 o = *dst;
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          if (p.relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          }
          *dst = v;
        } else {
          const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (col + t >= p.N) break;
            float x = e[t] + (p.bias ? p.bias[col + t] : 0.f);
            if (p.accumulate) x += crow[col + t];
            if (p.relu) x = fmaxf(x, 0.f);
            crow[col + t] = x;
          }
        }
      }
  }
}

// the six piece products of one 32-deep k-tile on split operands (x6.h), fragments read from LDS
template <int TM, int TN, class TA, class TB>
__device__ __forceinline__ void x6_mma_tile(f32x16 (&acc)[TM][TN], const float* As, const float* Bs, int arow, int brow, int lane) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) af[i][q] = TA::frag(As, q, arow + i * 32, ks, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) bf[j][q] = TB::frag(Bs, q, brow + j * 32, ks, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0], af[i][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][2], af[i][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][1], af[i][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0], af[i][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][1], af[i][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0], af[i][0], acc[i][j], 0, 0, 0);
      }
  }
}

// ---------------------------------------------------------------------------
// In-workgroup split-K ("wsk") form of the x6 kernel for the 4096-row per-level products (round 3).  Their output
// (4096 x 512) is 512 tiles of 64 x 64 -- two small workgroups per CU, each walking the whole reduction with one
// fragment read per MFMA and 8 operand bytes per 64 x 64 x 1 MACs -- while larger tiles leave CUs empty.  Here a
// 512-thread workgroup owns a 128 x 64 tile (256 workgroups, one per CU) and its two groups of four waves take
// alternate 32-deep k-tiles, each group with its own staging area: per CU the same eight waves, but 0.75 x the operand
// bytes and 18 fragment reads per 24 MFMAs instead of 12 per 12.  The two partial tiles meet in LDS (group 1 parks,
// group 0 adds: fixed order) -- no slabs, no combine launch.  A m-major; K % 64 == 0.  tools/bench_level_shapes.py:
// NN 4096 x 512 x 1536 50.6 -> 45.3 us, x 1024 36.8 -> 31.9 us.
// ---------------------------------------------------------------------------
template <int BKM>
__global__ __launch_bounds__(512, 1) void gemm_wsk_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 64, WM = 2, WN = 2, TM = 2, TN = 1;
  using TA = XTile<0, BM>;
  using TB = XTile<BKM, BN>;
  constexpr int GROUP = TA::LDS_FLOATS + TB::LDS_FLOATS;
  constexpr int RED = BM * BN;
  __shared__ __attribute__((aligned(16))) float lds[2 * GROUP > RED ? 2 * GROUP : RED];
  if (p.prio) __builtin_amdgcn_s_setprio(2);
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, p.nblk);
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  float* As = lds + grp * GROUP;
  float* Bs = As + TA::LDS_FLOATS;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  TA ta;
  TB tb;
  ta.rows_init(p.lda, m0, p.M, p.row_map, tid);
  tb.rows_init(p.ldb, n0, p.N, nullptr, tid);
  const int nt = p.K / 64;   // pairs of k-tiles
  ta.template load<true>(p.A, p.lda, m0, p.M, grp * 32, p.K, tid);
  tb.template load<true>(p.B, p.ldb, n0, p.N, grp * 32, p.K, tid);
  ta.store(As, tid);
  tb.store(Bs, tid);
  __syncthreads();
  for (int t = 0; t + 1 < nt; ++t) {
    ta.template load<false>(p.A, p.lda, m0, p.M, (t + 1) * 64 + grp * 32, p.K, tid);
    tb.template load<false>(p.B, p.ldb, n0, p.N, (t + 1) * 64 + grp * 32, p.K, tid);
    x6_mma_tile<TM, TN, TA, TB>(acc, As, Bs, wm * (BM / WM), wn * (BN / WN), lane);
    __syncthreads();
    ta.store(As, tid);
    tb.store(Bs, tid);
    __syncthreads();
  }
  x6_mma_tile<TM, TN, TA, TB>(acc, As, Bs, wm * (BM / WM), wn * (BN / WN), lane);
  __syncthreads();
  float* red = lds;   // [wave][i][reg][lane]
  if (grp == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * TM + i) * 16 + r) * 64 + lane] = acc[i][0][r];
  }
  __syncthreads();
  if (grp == 1) return;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] += red[((wave * TM + i) * 16 + r) * 64 + lane];
  store_tile<TM, TN>(acc, p, p.C, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

template <int AKM, int BKM, int BM, int BN, int WM, int WN, int BK, int ST, int MINB = 2, bool KG = false, bool X6 = false, int PF = 0>
__global__ __launch_bounds__(256, MINB) void gemm_f32_kernel(GemmParams p) {
  static_assert(PF == 0 || ST == 1, "prefetch forms: one LDS stage");
  static_assert(PF != 2 || !KG, "two k-tiles in flight: plain operands only");
  static_assert(!KG || (AKM == 1 && BKM == 1), "a gathered reduction needs two k-major operands");
  static_assert(!X6 || BK == 32, "the x6 core stages 32-wide k-tiles");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(WM * WN == 4, "four waves per workgroup");
  using TA = typename std::conditional<X6, XTile<AKM, BM>, Tile<AKM, BM, BK>>::type;
  using TB = typename std::conditional<X6, XTile<BKM, BN>, Tile<BKM, BN, BK>>::type;
  // ST = 1: one LDS stage, two barriers per k-tile.  ST = 2: two stages, one barrier per
  // k-tile -- a wave refills the other stage as soon as its own MFMAs of this tile are issued.
  constexpr int STAGE = TA::LDS_FLOATS + TB::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[ST * STAGE];

  // issue arbitration between co-resident waves goes by priority, then age: without this a
  // long-running bulk workgroup (older) starves the short chain launches that share its CU
  if (p.prio) __builtin_amdgcn_s_setprio(2);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int tile = xcd_remap(blockIdx.x, p.nblk);
  const int m0 = (tile / p.tiles_n) * BM;
  const int n0 = (tile % p.tiles_n) * BN;
  const int kb = blockIdx.y * p.k_chunk;
  const int ke = min(p.K, kb + p.k_chunk);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TA ta;
  TB tb;
  ta.rows_init(p.lda, m0, p.M, AKM == 0 ? p.row_map : nullptr, tid);
  tb.rows_init(p.ldb, n0, p.N, nullptr, tid);
  if (KG) {
    ta.load_gathered_checked(p.A, p.lda, m0, p.M, kb, ke, p.k_map, tid);
    tb.load_gathered_checked(p.B, p.ldb, n0, p.N, kb, ke, p.k_map, tid);
    ta.kq_fetch(p.k_map, kb + BK, ke, tid);
    tb.kq_fetch(p.k_map, kb + BK, ke, tid);
  } else {
    ta.template load<true>(p.A, p.lda, m0, p.M, kb, ke, tid);
    tb.template load<true>(p.B, p.ldb, n0, p.N, kb, ke, tid);
  }
  ta.store(lds, tid);
  tb.store(lds + TA::LDS_FLOATS, tid);
  __syncthreads();

  int cur = 0;
  auto compute = [&]() {
    const float* As = lds + cur * STAGE;
    const float* Bs = As + TA::LDS_FLOATS;
    if constexpr (X6) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q) af[i][q] = TA::frag(As, q, wm * (BM / WM) + i * 32, ks, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q) bf[j][q] = TB::frag(Bs, q, wn * (BN / WN) + j * 32, ks, lane);
        // the six piece products with i + j <= 4, small ones first; n-operand first as below (C^T in the accumulator)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0], af[i][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][2], af[i][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][1], af[i][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0], af[i][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][1], af[i][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0], af[i][0], acc[i][j], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
    for (int kc = 0; kc < BK / 8; ++kc) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = TA::frag(As, wm * (BM / WM) + i * 32, kc, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = TB::frag(Bs, wn * (BN / WN) + j * 32, kc, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // the n-operand goes first: the accumulator then holds C^T, i.e. each lane owns
          // runs of 4 consecutive n for one m, and the epilogue stores float4
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j].x, af[i].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j].y, af[i].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j].z, af[i].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j].w, af[i].w, acc[i][j], 0, 0, 0);
        }
    }
    }
  };
  auto refill_from = [&](const TA& a, const TB& b) {   // registers -> LDS for the next k-tile
    if (ST == 1) {
      __syncthreads();
      a.store(lds, tid);
      b.store(lds + TA::LDS_FLOATS, tid);
      __syncthreads();
    } else {
      float* nx = lds + (cur ^ 1) * STAGE;
      a.store(nx, tid);
      b.store(nx + TA::LDS_FLOATS, tid);
      __syncthreads();
      cur ^= 1;
    }
  };
  auto refill = [&]() { refill_from(ta, tb); };

  int k0 = kb;
  if constexpr (PF == 2) {
    // two k-tiles in flight: the tile staged after this one's products was loaded a whole iteration ago, so the split
    // arithmetic the compiler weaves into the MFMA stream never waits for memory
    if (k0 + 2 * BK <= ke) {
      TA ta2 = ta;
      TB tb2 = tb;
      ta.template load<false>(p.A, p.lda, m0, p.M, k0 + BK, ke, tid);
      tb.template load<false>(p.B, p.ldb, n0, p.N, k0 + BK, ke, tid);
      for (; k0 + 4 * BK <= ke; k0 += 2 * BK) {
        ta2.template load<false>(p.A, p.lda, m0, p.M, k0 + 2 * BK, ke, tid);
        tb2.template load<false>(p.B, p.ldb, n0, p.N, k0 + 2 * BK, ke, tid);
        compute();
        refill_from(ta, tb);
        ta.template load<false>(p.A, p.lda, m0, p.M, k0 + 3 * BK, ke, tid);
        tb.template load<false>(p.B, p.ldb, n0, p.N, k0 + 3 * BK, ke, tid);
        compute();
        refill_from(ta2, tb2);
      }
      compute();
      refill_from(ta, tb);
      k0 += BK;
    }
  }
  // hot loop: the next k-tile is a full one, loaded without predicates or branches (a loop
  // body with both load forms makes the compiler wait for the loads where the forms merge)
  for (; k0 + 2 * BK <= ke; k0 += BK) {
    if (KG) {
      ta.load_gathered(p.A, p.lda, m0, p.M, tid);
      tb.load_gathered(p.B, p.ldb, n0, p.N, tid);
      ta.kq_fetch(p.k_map, k0 + 2 * BK, ke, tid);   // rows of the tile after this one
      tb.kq_fetch(p.k_map, k0 + 2 * BK, ke, tid);
      // gathered rows, forms 1 and 3: left alone the compiler sinks these loads (they wait for the row indices) to the middle
      // of the tile's products and then waits for them almost at once; the barrier keeps them at the top of the iteration
      if constexpr (PF == 1 || PF == 3) __builtin_amdgcn_sched_barrier(0);
    } else if (!(p.dbg & 2)) {
      ta.template load<false>(p.A, p.lda, m0, p.M, k0 + BK, ke, tid);
      tb.template load<false>(p.B, p.ldb, n0, p.N, k0 + BK, ke, tid);
    }
    compute();
    if constexpr (PF == 1) __builtin_amdgcn_sched_barrier(0);   // the next tile's split stays behind this tile's products
    refill();
  }
  // the last one or two k-tiles (a ragged reduction ends in a partial tile)
  for (; k0 < ke; k0 += BK) {
    const bool more = (k0 + BK) < ke;
    if (more) {
      if (KG) {
        ta.load_gathered_checked(p.A, p.lda, m0, p.M, k0 + BK, ke, p.k_map, tid);
        tb.load_gathered_checked(p.B, p.ldb, n0, p.N, k0 + BK, ke, p.k_map, tid);
      } else {
        ta.template load<true>(p.A, p.lda, m0, p.M, k0 + BK, ke, tid);
        tb.template load<true>(p.B, p.ldb, n0, p.N, k0 + BK, ke, tid);
      }
    }
    compute();
    if (more) refill();
  }
  if (ST == 1) __syncthreads();

  // epilogue
  float* C = p.C + (size_t)blockIdx.y * p.c_split;
  if ((p.dbg & 1) && acc[0][0][0] != 12345.f) return;
  store_tile<TM, TN>(acc, p, C, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

// out[m][n] (+)= sum_s slab[s][m][n] + bias[n]
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int splits, long long stride,
                                     float* __restrict__ C, int ldc, int M, int N,
                                     const float* __restrict__ bias, int relu, int accumulate) {
  const int nq = N >> 2;
  const long long total = (long long)M * nq;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i / nq), q = (int)(i % nq);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(slab + k * stride + (long long)m * N + q * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + q * 4);
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    float4* dst = reinterpret_cast<float4*>(C + (size_t)m * ldc + q * 4);
    if (accumulate) {
      const float4 o = *dst;
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    if (relu) {
      s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
    }
    *dst = s;
  }
}

// ---- optional timing of every GEMM launch with HIP events on its own stream ----
// (bench.py's roofline figures; off by default, no cost when off)
struct GemmProf { hipEvent_t a, b; int variant; double flops; };
struct ProfState {
  bool on = false;
  std::deque<GemmProf> recs;   // references stay valid across push_back
  size_t used = 0;
};
ProfState g_prof;

std::mutex g_prof_mu;   // callers may launch from several host threads

GemmProf* prof_begin(int variant, double flops, hipStream_t st) {
  if (!g_prof.on) return nullptr;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (g_prof.used == g_prof.recs.size()) {
    GemmProf r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return nullptr;
    g_prof.recs.push_back(r);
  }
  GemmProf* r = &g_prof.recs[g_prof.used++];
  r->variant = variant;
  r->flops = flops;
  (void)hipEventRecord(r->a, st);
  return r;
}
void prof_end(GemmProf* r, hipStream_t st) {
  if (r) (void)hipEventRecord(r->b, st);
}
}  // namespace

// the fused level-step kernels (level.hip) run the same MFMA main loop and are counted
// with the NT launches
void* gemm_prof_begin(double flops, hipStream_t st) { return prof_begin(0, flops, st); }
// the HBM-bound kernels bench.py reports beside the GEMM (tags 4..7, `flops` carries their algorithmic BYTES)
void* kernel_prof_begin(int tag, double bytes, hipStream_t st) { return prof_begin(tag, bytes, st); }
void gemm_prof_end(void* h, hipStream_t st) { prof_end(static_cast<GemmProf*>(h), st); }

namespace {

template <int AKM, int BKM, int BM, int BN, int WM, int WN, int BK = 32, int ST = 1, int MINB = 2, bool KG = false, bool X6 = false, int PF = 0>
int launch(const GemmDesc& d, float* ws, size_t ws_bytes, hipStream_t stream) {
  GemmParams p;
  p.A = d.A; p.B = d.B; p.C = d.C; p.bias = d.bias;
  p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
  p.M = d.M; p.N = d.N; p.K = d.K;
  p.relu = d.relu; p.accumulate = d.accumulate;
  p.prio = g_force_prio >= 0 ? g_force_prio : d.prio;
  p.dbg = g_dbg;
  p.row_map = d.row_map;
  p.k_map = d.k_map;
  const int tiles_m = cdiv(d.M, BM);
  p.tiles_n = cdiv(d.N, BN);
  p.nblk = tiles_m * p.tiles_n;

  // split the reduction when the output alone cannot fill the chip (the
  // weight-gradient products: small [out][in] result, reduction over all rows)
  int splits = 1;
  if (p.nblk < 512 && d.K >= 2048 && ws) {
    static const int bulk_blocks = getenv("DVAE_BULK_BLOCKS") ? atoi(getenv("DVAE_BULK_BLOCKS")) : 768;   // tuning
    splits = min(cdiv(d.bulk ? bulk_blocks : g_splitk_blocks, p.nblk), d.K / 512);
  } else if (p.nblk <= 128 && d.K >= 512 && ws) {
    // small-batch per-level products: a handful of tiles, each walking the whole K
    // loop at one workgroup per CU, is latency-bound -- spread K over the idle CUs
    splits = min(cdiv(256, p.nblk), d.K / 128);
  } else if (p.nblk < 448 && d.K >= 1024 && ws && !d.bulk && g_mid_split) {
    // mid-sized per-level products with a long reduction (the backward products of 1.5k .. 3.5k-row levels: K = 2HP,
    // 3HP): a few hundred 64x64 tiles that each walk 32 .. 48 k-tiles alone leave the chip latency-bound; three or
    // four K slices per tile fill it (tools/bench_level_shapes.py: 2048 x 512 x 1536 49 -> 30 us incl. the combine)
    splits = min(cdiv(768, p.nblk), d.K / 256);
  }
  if (g_force_splits > 0 && ws) splits = g_force_splits;
  if (d.row_map) splits = 1;   // the slab reduction does not scatter rows
  if (splits > 1) {
    const size_t slab = (size_t)d.M * d.N * sizeof(float);
    if (slab == 0) return DVAE_OK;
    splits = (int)min((size_t)splits, ws_bytes / slab);
    if (splits < 2 || (d.N & 3)) splits = 1;
  }
  p.k_chunk = round_up(cdiv(d.K, splits), BK);
  splits = cdiv(d.K, p.k_chunk);
  if (splits > 1) {
    p.C = ws; p.ldc = d.N; p.c_split = (long long)d.M * d.N;
    p.bias = nullptr; p.relu = 0; p.accumulate = 0;
  } else {
    p.c_split = 0;
  }
  GemmProf* pr = prof_begin(AKM * 2 + BKM, 2.0 * d.M * d.N * d.K, stream);
  hipLaunchKernelGGL((gemm_f32_kernel<AKM, BKM, BM, BN, WM, WN, BK, ST, MINB, KG, X6, PF>), dim3(p.nblk, splits), dim3(256),
                     0, stream, p);
  prof_end(pr, stream);
  DVAE_LAUNCH_CHECK();
  if (d.partials) {
    d.partials->slab = splits > 1 ? ws : nullptr;
    d.partials->splits = splits;
    d.partials->stride = (long long)d.M * d.N;
  }
  if (splits > 1 && !d.partials) {
    const long long total = (long long)d.M * (d.N / 4);
    const int grid = (int)min((long long)2048, (total + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, stream, ws, splits,
                       (long long)d.M * d.N, d.C, d.ldc, d.M, d.N, d.bias, d.relu, d.accumulate);
    DVAE_LAUNCH_CHECK();
  }
  return DVAE_OK;
}

template <int AKM, int BKM, int BM, int BN, int WM, int WN>
int launch_pf(int pf, const GemmDesc& d, float* ws, size_t ws_bytes, hipStream_t stream) {
  if (pf == 1) return launch<AKM, BKM, BM, BN, WM, WN, 32, 1, 2, false, true, 1>(d, ws, ws_bytes, stream);
  return launch<AKM, BKM, BM, BN, WM, WN, 32, 1, 2, false, true, 2>(d, ws, ws_bytes, stream);
}

}  // namespace

extern "C" int dvae_gemm_tune(int splitk_blocks, int force_prio) {
  if (splitk_blocks > 0) g_splitk_blocks = splitk_blocks;
  g_force_prio = force_prio;
  return DVAE_OK;
}
// 1 (default): fp32 products as six bf16 MFMAs on split operands (the x6 core above); 0: v_mfma_f32_32x32x2_f32
static int g_x6 = getenv("DVAE_GEMM_X6") ? atoi(getenv("DVAE_GEMM_X6")) : 1;
extern "C" int dvae_gemm_x6(int on) { g_x6 = on; return DVAE_OK; }
extern "C" int dvae_gemm_core(void) { return g_x6; }
int gemm_x6_enabled() { return g_x6; }
static int g_force_tile = 0;   // tuning knob (tests/bench_gemm.py): 0 auto, 1 128x128, 2 64x128, 3 64x64, 4 128x96 (x6 NT, N % 96 == 0)
extern "C" int dvae_gemm_force_tile(int mode) { g_force_tile = mode; return DVAE_OK; }

namespace {
// the in-workgroup split-K kernel: one 128 x 64 tile per workgroup, no slabs
int launch_wsk(const GemmDesc& d, hipStream_t stream) {
  GemmParams p;
  p.A = d.A; p.B = d.B; p.C = d.C; p.bias = d.bias;
  p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
  p.M = d.M; p.N = d.N; p.K = d.K;
  p.relu = d.relu; p.accumulate = d.accumulate;
  p.prio = g_force_prio >= 0 ? g_force_prio : d.prio;
  p.dbg = 0; p.row_map = d.row_map; p.k_map = nullptr;
  p.k_chunk = d.K; p.c_split = 0;
  p.tiles_n = cdiv(d.N, 64);
  p.nblk = cdiv(d.M, 128) * p.tiles_n;
  GemmProf* pr = prof_begin(d.b_kmajor ? 1 : 0, 2.0 * d.M * d.N * d.K, stream);
  if (d.b_kmajor) hipLaunchKernelGGL(gemm_wsk_kernel<1>, dim3(p.nblk), dim3(512), 0, stream, p);
  else hipLaunchKernelGGL(gemm_wsk_kernel<0>, dim3(p.nblk), dim3(512), 0, stream, p);
  prof_end(pr, stream);
  DVAE_LAUNCH_CHECK();
  if (d.partials) { d.partials->slab = nullptr; d.partials->splits = 1; d.partials->stride = 0; }
  return DVAE_OK;
}
// opt-in (DVAE_WSK=1): alone it beats the 64 x 64 tiles by 10-13 % on the 4096-row backward products, inside the step it
// LOSES (headline 9.38 vs 9.32 ms, same box): its one 75 KB workgroup per CU cannot share a CU with two of the side
// stream's 50 KB GEMM workgroups, so it queues behind them where the small tiles slip in (profiles/r03_level_gemm.txt)
int g_wsk = getenv("DVAE_WSK") ? 1 : 0;
}  // namespace

int gemm_f32(const GemmDesc& d, float* ws, size_t ws_bytes, hipStream_t stream) {
  if (d.M <= 0 || d.N <= 0) return DVAE_OK;
  DVAE_REQUIRE(d.K > 0, "gemm: K must be positive");
  DVAE_REQUIRE(!d.partials || (!d.bias && !d.relu), "gemm: partial sums carry no bias / relu");
  DVAE_REQUIRE(!d.row_map || !d.a_kmajor, "gemm: a row map needs an m-major A");
  if (d.k_map) {
    // gathered reduction rows: the weight-gradient shape only (TN, 128x128 tiles, split K)
    DVAE_REQUIRE(d.a_kmajor && d.b_kmajor && (d.M & 3) == 0 && (d.N & 3) == 0 && d.N > 32,
                 "gemm: a reduction map needs two k-major operands and N > 32");
    if (g_x6) {
      // gathered rows: the loads held at the top of the iteration (form 3; form 1 also keeps the split behind the products:
      // 4 % slower here, the k-major staging likes its split woven into the MFMA stream).  tools/bench_gemm_kmap.py, 80 % of
      // 57344 rows kept: plain loop 720 us, form 1 632, form 3 606 (plain TN of the same K: 520).  DVAE_GEMM_PF=0 | 1: A-B tests
      if (g_pf == 1) return launch<1, 1, 128, 128, 2, 2, 32, 1, 2, true, true, 1>(d, ws, ws_bytes, stream);
      if (g_pf != 0) return launch<1, 1, 128, 128, 2, 2, 32, 1, 2, true, true, 3>(d, ws, ws_bytes, stream);
      return launch<1, 1, 128, 128, 2, 2, 32, 1, 2, true, true>(d, ws, ws_bytes, stream);
    }
    return launch<1, 1, 128, 128, 2, 2, 16, 1, 2, true>(d, ws, ws_bytes, stream);
  }
  DVAE_REQUIRE(((uintptr_t)d.A & 15) == 0 && ((uintptr_t)d.B & 15) == 0, "gemm: operands must be 16-byte aligned");
  DVAE_REQUIRE((d.lda & 3) == 0 && (d.ldb & 3) == 0, "gemm: leading dimensions must be multiples of 4");
  // the contiguous extent of every operand is loaded as float4
  if (!d.a_kmajor) DVAE_REQUIRE((d.K & 3) == 0, "gemm: K must be a multiple of 4 for an m-major A");
  else DVAE_REQUIRE((d.M & 3) == 0, "gemm: M must be a multiple of 4 for a k-major A");
  if (!d.b_kmajor) DVAE_REQUIRE((d.K & 3) == 0, "gemm: K must be a multiple of 4 for an n-major B");
  else DVAE_REQUIRE((d.N & 3) == 0, "gemm: N must be a multiple of 4 for a k-major B");
  // tile choice: narrow outputs take 128x32.  Otherwise the largest tile that still gives
  // every CU two or more workgroups (measured: below that a launch is latency-bound and the
  // next smaller tile wins), and between 128x128 and 64x128 the one that needs fewer
  // "rounds" of 256 CUs.  Long reductions into a small output keep 128x128 and split K.
  const bool narrow = d.N <= 32;
  // at most 32 output rows (add_vertex.2's weight gradient: 32 x 1024 summed over all node rows;
  // the per-type tables): a 32x128 tile instead of wasting three quarters of a 128-row one
  const bool flat = !narrow && d.M <= 32 && !d.k_map && !d.row_map && g_force_tile == 0;
  bool half = false, quarter = false;
  if (g_force_tile == 1) half = false;
  else if (g_force_tile == 2) half = true;
  else if (g_force_tile == 3) quarter = true;
  else if (!narrow && !flat) {
    const long long t128 = (long long)cdiv(d.M, 128) * cdiv(d.N, 128);
    const long long t64 = (long long)cdiv(d.M, 64) * cdiv(d.N, 128);
    const bool splitk = d.K >= 8192 && t128 < 256 && ws;   // weight gradients: reduction over all rows
    if (!splitk) {
      const long long c128 = ((t128 + 255) / 256) * 2, c64 = (t64 + 255) / 256;
      if (t128 >= 512) half = c64 < c128 && t128 < 2048;
      else if (t64 >= 512) half = true;
      else quarter = true;
    }
  }
  // 4096-row per-level products that would take 64 x 64 tiles: the in-workgroup split-K kernel (gemm_wsk_kernel)
  if (g_x6 && g_wsk && quarter && g_force_tile == 0 && g_force_splits == 0 && !d.a_kmajor && !d.k_map && (d.K & 63) == 0 &&
      d.K >= 512 && (d.N & 3) == 0) {
    const long long t = (long long)cdiv(d.M, 128) * cdiv(d.N, 64);
    if (t >= 224 && t <= 512) return launch_wsk(d, stream);
  }
  // 128 x 96 tiles (x6 core, NT products): an output of ~4096 x 1536 is 768 tiles of 64 x 128 -- one and a half rounds of the
  // 512 workgroup slots -- or 384 of 128 x 128; 128 x 96 makes it exactly 512 (one round, two workgroups on every CU)
  static const bool t96_off = getenv("DVAE_NO_TILE96") != nullptr;   // A-B tests
  if (g_x6 && !d.a_kmajor && !d.b_kmajor && !narrow && !flat && d.N % 96 == 0 && (g_force_tile == 4 || (g_force_tile == 0 && !t96_off && half))) {
    const long long t96 = (long long)cdiv(d.M, 128) * (d.N / 96);
    if (g_force_tile == 4 || (t96 > 384 && t96 <= 512)) {
      const int pf = pf_of(0, 0, false);
      if (pf) return launch_pf<0, 0, 128, 96, 4, 1>(pf, d, ws, ws_bytes, stream);
      return launch<0, 0, 128, 96, 4, 1, 32, 1, 2, false, true>(d, ws, ws_bytes, stream);
    }
  }
#define GEMM_PICK(AK, BKK)                                                                   \
  if (g_x6) {                                                                                \
    if (narrow) return launch<AK, BKK, 128, 32, 4, 1, 32, 1, 2, false, true>(d, ws, ws_bytes, stream);   \
    if (flat) return launch<AK, BKK, 32, 128, 1, 4, 32, 1, 2, false, true>(d, ws, ws_bytes, stream);     \
    /* prefetch form (pf_of): 1 = the next tile's split held behind this tile's products, 2 = two k-tiles in flight */ \
    const int pf = pf_of(AK, BKK, quarter);                                                  \
    if (pf) {                                                                                \
      if (quarter) return launch_pf<AK, BKK, 64, 64, 2, 2>(pf, d, ws, ws_bytes, stream);     \
      if (half) return launch_pf<AK, BKK, 64, 128, 1, 4>(pf, d, ws, ws_bytes, stream);       \
      return launch_pf<AK, BKK, 128, 128, 2, 2>(pf, d, ws, ws_bytes, stream);                \
    }                                                                                        \
    if (quarter) return launch<AK, BKK, 64, 64, 2, 2, 32, 1, 2, false, true>(d, ws, ws_bytes, stream);   \
    return half ? launch<AK, BKK, 64, 128, 1, 4, 32, 1, 2, false, true>(d, ws, ws_bytes, stream)         \
                : launch<AK, BKK, 128, 128, 2, 2, 32, 1, 2, false, true>(d, ws, ws_bytes, stream);       \
  }                                                                                          \
  if (narrow) return launch<AK, BKK, 128, 32, 4, 1>(d, ws, ws_bytes, stream);                \
  if (flat) return launch<AK, BKK, 32, 128, 1, 4>(d, ws, ws_bytes, stream);                  \
  if (quarter) return launch<AK, BKK, 64, 64, 2, 2>(d, ws, ws_bytes, stream);                \
  if (g_variant == 3)                                                                        \
    return half ? launch<AK, BKK, 64, 128, 1, 4, 32, 1>(d, ws, ws_bytes, stream)             \
                : launch<AK, BKK, 128, 128, 2, 2, 32, 1>(d, ws, ws_bytes, stream);           \
  return half ? launch<AK, BKK, 64, 128, 1, 4, 16, 1>(d, ws, ws_bytes, stream)               \
              : launch<AK, BKK, 128, 128, 2, 2, 16, 1>(d, ws, ws_bytes, stream)
  if (!d.a_kmajor && !d.b_kmajor) { GEMM_PICK(0, 0); }
  if (!d.a_kmajor && d.b_kmajor) { GEMM_PICK(0, 1); }
  if (d.a_kmajor && d.b_kmajor) { GEMM_PICK(1, 1); }
#undef GEMM_PICK
  dvae_set_error("gemm: layout a_kmajor=1,b_kmajor=0 is not used by this library");
  return DVAE_ERR_ARG;
}

extern "C" int dvae_gemm_f32(const float* A, int lda, int a_kmajor, const float* B, int ldb,
                             int b_kmajor, float* C, int ldc, int M, int N, int K, const float* bias,
                             int relu, int accumulate, float* ws, size_t ws_bytes, void* stream) {
  GemmDesc d{};
  d.A = A; d.lda = lda; d.a_kmajor = a_kmajor;
  d.B = B; d.ldb = ldb; d.b_kmajor = b_kmajor;
  d.C = C; d.ldc = ldc; d.M = M; d.N = N; d.K = K;
  d.bias = bias; d.relu = relu; d.accumulate = accumulate;
  d.prio = 0; d.bulk = 0; d.row_map = nullptr; d.k_map = nullptr;
  return gemm_f32(d, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int dvae_gemm_f32_mapped(const float* A, int lda, int a_kmajor, const float* B, int ldb,
                                    int b_kmajor, float* C, int ldc, int M, int N, int K, int accumulate,
                                    const int* row_map, const int* k_map, float* ws, size_t ws_bytes,
                                    void* stream) {
  DVAE_REQUIRE(!(row_map && k_map), "gemm: one map at a time");
  GemmDesc d{};
  d.A = A; d.lda = lda; d.a_kmajor = a_kmajor;
  d.B = B; d.ldb = ldb; d.b_kmajor = b_kmajor;
  d.C = C; d.ldc = ldc; d.M = M; d.N = N; d.K = K;
  d.bias = nullptr; d.relu = 0; d.accumulate = accumulate;
  d.prio = 0; d.bulk = 0; d.row_map = row_map; d.k_map = k_map;
  return gemm_f32(d, ws, ws_bytes, (hipStream_t)stream);
}

// variant: 0 NT, 1 NN, 3 TN.  out3[v] = {launches, milliseconds, executed flops}
// enable: 1 start a new log, 2 resume the current one, 0 pause (the log is kept until it is read)
extern "C" int dvae_gemm_profile(int enable) {
  g_prof.on = enable != 0;
  if (enable == 1) g_prof.used = 0;
  return DVAE_OK;
}

// the HBM-bound kernels of the same log: out12 = 4 x {launches, ms, algorithmic bytes} for
// 0 pair scorer forward, 1 pair scorer backward, 2 aggregation forward, 3 aggregation backward.
// Does not clear the log: call it BEFORE dvae_gemm_profile_read.
extern "C" int dvae_hbm_profile_read(double* out12) {
  for (int i = 0; i < 12; ++i) out12[i] = 0.0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    GemmProf& r = g_prof.recs[i];
    if (r.variant < 4 || r.variant > 7) continue;
    DVAE_CHECK_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    DVAE_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    double* o = out12 + 3 * (r.variant - 4);
    o[0] += 1.0; o[1] += ms; o[2] += r.flops;
  }
  return DVAE_OK;
}

extern "C" int dvae_gemm_profile_read(double* out12) {
  for (int i = 0; i < 12; ++i) out12[i] = 0.0;
  // slot 2 (layout index 2 is unused): out12[6] = wall time during which at least one GEMM
  // launch was running -- launches on the two streams overlap, so the per-launch sum
  // counts such time twice
  std::vector<std::pair<double, double>> iv;
  for (size_t i = 0; i < g_prof.used; ++i) {
    GemmProf& r = g_prof.recs[i];
    DVAE_CHECK_HIP(hipEventSynchronize(r.b));
    float ms = 0.f, t0 = 0.f;
    DVAE_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    if (r.variant > 3) continue;   // an HBM-kernel record (dvae_hbm_profile_read)
    double* o = out12 + 3 * (r.variant & 3);
    o[0] += 1.0; o[1] += ms; o[2] += r.flops;
    if (hipEventElapsedTime(&t0, g_prof.recs[0].a, r.a) == hipSuccess) iv.push_back({t0, (double)t0 + ms});
  }
  std::sort(iv.begin(), iv.end());
  double busy = 0.0, cur_lo = 0.0, cur_hi = -1.0;
  for (auto& p : iv) {
    if (p.first > cur_hi) {
      if (cur_hi > cur_lo) busy += cur_hi - cur_lo;
      cur_lo = p.first; cur_hi = p.second;
    } else if (p.second > cur_hi) cur_hi = p.second;
  }
  if (cur_hi > cur_lo) busy += cur_hi - cur_lo;
  out12[6] = busy;
  g_prof.used = 0;
  return DVAE_OK;
}
