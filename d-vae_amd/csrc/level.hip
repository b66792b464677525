// Fused level-step kernels (forward): one launch does the dense contraction of a
// level on the fp32 matrix cores AND the gate arithmetic that consumes it.
//
//   level_gru:  GH = h_in . Whh^T  ->  GRUCell gates r,z,n and the new state
//               (dvae_model.py:83,244), replacing "hidden GEMM + gru_fwd_kernel"
//   level_gm :  GMP = h . [Wg;Wm][:, :H]^T  ->  sigmoid(g) * m with the position
//               columns and biases folded in as posb (dvae_model.py:71-79),
//               replacing "gate/map GEMM + gm_fwd_kernel"
//
// The weight matrices keep their layout (gates stacked at multiples of HP rows).
// A workgroup takes 64 hidden units and loads the matching 64 rows of EVERY gate
// into its LDS tile, so that after the k loop each lane holds all G pre-activations
// of the same (row, hidden unit) and the epilogue needs no second pass over HBM.
// Same MFMA (v_mfma_f32_32x32x2_f32) and the same gate formulas as gemm.hip +
// kernels.hip; the fp32 summation order differs only where the plain GEMM splits K.
#include "kernels.h"
#include <mutex>

#include "x6.h"

namespace {

constexpr int BK = 16;          // k-tile (measured best for these tiles, as in gemm.hip)
constexpr int LDM = BK + 4;
constexpr int KQ = BK / 4;      // float4 per tile row
constexpr int RS = 256 / KQ;    // tile rows covered by one load step of the workgroup
constexpr int BNG = 64;   // hidden units per workgroup

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int x = bid & 7, j = bid >> 3;
  const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + j;
}

struct LevelCore {
  const float* A; int lda; int M;   // [M][K] rows of this launch
  const float* W; int ldw;          // [G*HP][K]
  int K, HP;
  int tiles_n, nblk;
  int prio;
  // GATHER kernels: the A rows do not exist yet -- row m is the gated sum over the predecessors of node row
  // r_base + m (dvae_model.py:77-80), summed by the tile loader from GM through the CSR lists
  const float* GM; const int* csr_ptr; const int* csr_idx; int r_base;
  float* AGG;                       // the summed rows are also written here (the backward pass reads them)
};

struct GruEpi {
  SchedDir S; int R, prefix, p_base;
  const float* x; int N, C;
  GruW w;
  const float* AGG; float* GH; float* GHN; float* HV; float* Hn;
};

struct GmEpi {
  SchedDir S; int r_base;
  const float* posb; float* GMP; float* GM;
};

// per-row state of the GRU epilogue: the node encoding is almost always one-hot, in which
// case the input product x.W_ih^T is ONE row of WihT (same fmaf as the general loop, so the
// same bits) and the epilogue has no divergent loop
struct GruRow { int p, r; const float* xr; int hot; float xv; bool fin; };

// row m of the launch, hidden units [col, col+4): v[g] = the G pre-activations
// gather (GATHER kernels): h_in is summed again from GM here -- the AGG row is being written by another
// workgroup (the first column tile) and may not be complete yet
__device__ __forceinline__ void epilogue(const GruEpi& e, int HP, const GruRow& rw, int col, const float4* v,
                                         const LevelCore* gather = nullptr) {
  float4 gi[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) gi[g] = ld4(e.w.bih + g * HP + col);
  if (rw.hot >= 0) {
#pragma unroll
    for (int g = 0; g < 3; ++g) gi[g] = fma4(rw.xv, ld4(e.w.WihT + (size_t)rw.hot * 3 * HP + g * HP + col), gi[g]);
  } else if (rw.hot == -2) {
    for (int c = 0; c < e.C; ++c) {
      const float xv = rw.xr[c];
      if (xv != 0.f) {
#pragma unroll
        for (int g = 0; g < 3; ++g) gi[g] = fma4(xv, ld4(e.w.WihT + (size_t)c * 3 * HP + g * HP + col), gi[g]);
      }
    }
  }
  const float4 ghr = add4(ld4(e.w.bhh + col), v[0]);
  const float4 ghz = add4(ld4(e.w.bhh + HP + col), v[1]);
  const float4 ghn = add4(ld4(e.w.bhh + 2 * HP + col), v[2]);
  float4 a;
  if (gather) {
    const int c0 = gather->csr_ptr[rw.p], c1 = gather->csr_ptr[rw.p + 1];
    a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = c0; c < c1; ++c) a = add4(a, ld4(gather->GM + (size_t)gather->csr_idx[c] * HP + col));
  } else {
    a = ld4(e.AGG + (size_t)rw.p * HP + col);
  }
  float4 rr, zz, nn, hh;
#define GRU_LANE(f)                                   \
  rr.f = sigm(gi[0].f + ghr.f);                       \
  zz.f = sigm(gi[1].f + ghz.f);                       \
  nn.f = tanhf(gi[2].f + rr.f * ghn.f);               \
  hh.f = (1.f - zz.f) * nn.f + zz.f * a.f;
  GRU_LANE(x) GRU_LANE(y) GRU_LANE(z) GRU_LANE(w)
#undef GRU_LANE
  float* gh = e.GH + (size_t)rw.p * 3 * HP;
  st4(gh + col, rr);
  st4(gh + HP + col, zz);
  st4(gh + 2 * HP + col, nn);
  st4(e.GHN + (size_t)rw.p * HP + col, ghn);
  st4(e.HV + (size_t)rw.p * HP + col, hh);
  if (rw.fin) st4(e.Hn + (size_t)rw.r * HP + col, hh);
}

struct GmRow { int r; const float* pb; };

__device__ __forceinline__ void epilogue(const GmEpi& e, int HP, const GmRow& rw, int col, const float4* v,
                                         const LevelCore* = nullptr) {
  const float4 gpre = add4(v[0], ld4(rw.pb + col));
  const float4 mm = add4(v[1], ld4(rw.pb + HP + col));
  const float4 g = make_float4(sigm(gpre.x), sigm(gpre.y), sigm(gpre.z), sigm(gpre.w));
  float* gp = e.GMP + (size_t)rw.r * 2 * HP;
  st4(gp + col, g);
  st4(gp + HP + col, mm);
  st4(e.GM + (size_t)rw.r * HP + col, make_float4(g.x * mm.x, g.y * mm.y, g.z * mm.z, g.w * mm.w));
}

__device__ __forceinline__ GruRow row_meta(const GruEpi& e, int HP, int m) {
  GruRow rw;
  rw.p = e.p_base + m;
  rw.r = e.prefix ? e.S.prow_node[rw.p] : rw.p;
  rw.xr = e.x + ((size_t)e.S.row_graph[rw.r] * e.N + e.S.row_node[rw.r]) * e.C;
  rw.fin = e.prefix && rw.p == final_prow(e.S, e.R, rw.r);
  // hot = c if x has exactly one nonzero (at c), -1 if x is all zero, -2 otherwise
  rw.hot = -1; rw.xv = 0.f;
  for (int c = 0; c < e.C; ++c) {
    const float xv = rw.xr[c];
    if (xv != 0.f) { rw.hot = rw.hot == -1 ? c : -2; rw.xv = xv; }
  }
  return rw;
}
__device__ __forceinline__ GmRow row_meta(const GmEpi& e, int HP, int m) {
  GmRow rw;
  rw.r = e.r_base + m;
  rw.pb = e.posb + (size_t)e.S.row_node[rw.r] * 2 * HP;
  return rw;
}

// predecessors kept in registers per loader row; longer lists continue from the CSR array in memory
constexpr int GATHER_REG = 8;

// G gates, TM row sub-tiles per wave: workgroup tile = (64*TM) rows x (G x 64) columns,
// four waves as 2 (rows) x 2 (hidden units), each wave TM x G MFMA tiles of 32x32.
// GATHER: "gated-sum predecessor aggregation + GRUCell gate GEMM" in ONE launch: the A-tile loader sums the
// predecessors' GM rows (CSR slice of the tile's rows held in registers) instead of reading a finished AGG row.
template <int G, int TM, class Epi, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void level_kernel(LevelCore p, Epi e) {
  constexpr int BM = 64 * TM;
  constexpr int A_FLOATS = BM * LDM, B_FLOATS = G * BNG * LDM;
  constexpr int NVA = BM * BK / 4 / 256, NVB = G * BNG * BK / 4 / 256;   // float4 per thread
  __shared__ __attribute__((aligned(16))) float lds[A_FLOATS + B_FLOATS];
  float* As = lds;
  float* Bs = lds + A_FLOATS;

  if (p.prio) __builtin_amdgcn_s_setprio(2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, p.nblk);
  const int m0 = (tile / p.tiles_n) * BM;
  const int n0 = (tile % p.tiles_n) * BNG;

  f32x16 acc[TM][G];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][g][r] = 0.f;

  // one tile row = KQ float4; thread t covers row t/KQ (+RS per step), quad t%KQ.  Rows beyond
  // M are clamped to the last row (their results are never stored); K and HP are multiples
  // of the tile sizes, so no load needs a predicate.
  const int lr = tid / KQ, kq = (tid % KQ) * 4;
  const float* pa[NVA];
  const float* pb[NVB];
  int gdeg[NVA], gc0[NVA], gidx[NVA][GATHER_REG];
  (void)gdeg; (void)gc0; (void)gidx;
#pragma unroll
  for (int i = 0; i < NVA; ++i) {
    const int m = min(m0 + lr + RS * i, p.M - 1);
    pa[i] = (GATHER ? p.AGG + (size_t)(p.r_base + m) * p.HP : p.A + (size_t)m * p.lda) + kq;
    if (GATHER) {
      const int r = p.r_base + m;
      gc0[i] = p.csr_ptr[r];
      gdeg[i] = p.csr_ptr[r + 1] - gc0[i];
#pragma unroll
      for (int j = 0; j < GATHER_REG; ++j) gidx[i][j] = j < gdeg[i] ? p.csr_idx[gc0[i] + j] : 0;
    }
  }
  // the first column tile of every row tile writes the summed rows out (rows beyond M are clamped copies)
  const bool agg_out = GATHER && n0 == 0;
#pragma unroll
  for (int i = 0; i < NVB; ++i) {
    const int row = lr + RS * i;   // LDS row = gate * 64 + hidden unit
    pb[i] = p.W + ((size_t)(row / BNG) * p.HP + n0 + row % BNG) * p.ldw + kq;
  }
  float4 va[NVA], vb[NVB];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      if (!GATHER) {
        va[i] = ld4(pa[i] + k0);
      } else {
        // same order as agg_fwd_kernel (CSR order), so the sums carry the same bits
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < GATHER_REG; ++j)
          if (j < gdeg[i]) acc = add4(acc, ld4(p.GM + (size_t)gidx[i][j] * p.HP + kq + k0));
        for (int j = GATHER_REG; j < gdeg[i]; ++j)
          acc = add4(acc, ld4(p.GM + (size_t)p.csr_idx[gc0[i] + j] * p.HP + kq + k0));
        va[i] = acc;
        if (agg_out) st4(const_cast<float*>(pa[i]) + k0, acc);
      }
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) vb[i] = ld4(pb[i] + k0);
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NVA; ++i) st4(As + (lr + RS * i) * LDM + kq, va[i]);
#pragma unroll
    for (int i = 0; i < NVB; ++i) st4(Bs + (lr + RS * i) * LDM + kq, vb[i]);
  };
  auto compute = [&]() {
#pragma unroll
    for (int kc = 0; kc < BK / 8; ++kc) {
      float4 af[TM], bf[G];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = ld4(As + (wm * (BM / 2) + i * 32 + (lane & 31)) * LDM + kc * 8 + (lane >> 5) * 4);
#pragma unroll
      for (int g = 0; g < G; ++g)
        bf[g] = ld4(Bs + (g * BNG + wn * 32 + (lane & 31)) * LDM + kc * 8 + (lane >> 5) * 4);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          // n-operand first: the accumulator holds C^T, each lane owns runs of 4 hidden units
          acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g].x, af[i].x, acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g].y, af[i].y, acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g].z, af[i].z, acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g].w, af[i].w, acc[i][g], 0, 0, 0);
        }
    }
  };

  load(0);
  store();
  __syncthreads();
  int k0 = 0;
  for (; k0 + BK < p.K; k0 += BK) {
    load(k0 + BK);
    compute();
    __syncthreads();
    store();
    __syncthreads();
  }
  compute();

  // epilogue: lane owns row m = lane&31 of each row sub-tile and, per register group q,
  // hidden units n0 + 32*wn + 8q + 4*(lane>>5) + 0..3 of every gate
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / 2) + i * 32 + (lane & 31);
    if (m >= p.M) continue;
    const auto rw = row_meta(e, p.HP, m);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = n0 + wn * 32 + 8 * q + 4 * (lane >> 5);
      float4 v[G];
#pragma unroll
      for (int g = 0; g < G; ++g)
        v[g] = make_float4(acc[i][g][4 * q], acc[i][g][4 * q + 1], acc[i][g][4 * q + 2], acc[i][g][4 * q + 3]);
      epilogue(e, p.HP, rw, col, v, GATHER ? &p : nullptr);
    }
  }
}

// The same fused level step on the x6 arithmetic core (x6.h: operands split exactly into three bf16 pieces while staged,
// six v_mfma_f32_32x32x16_bf16 per fp32 product): k-tile 32, both operands m-major, the weight tile's LDS row is
// gate * 64 + hidden unit as above, so the epilogues are shared.
// PF: prefetch form of the main loop as in gemm.hip (1 = the next tile's split held behind this tile's products,
// 2 = two k-tiles in flight).
template <int G, int TM, class Epi, int PF = 0>
__global__ __launch_bounds__(256, 2) void level_x6_kernel(LevelCore p, Epi e) {
  constexpr int BM = 64 * TM;
  using TA = XTile<0, BM>;
  using TB = XTile<0, G * BNG>;
  __shared__ __attribute__((aligned(16))) float lds[TA::LDS_FLOATS + TB::LDS_FLOATS];
  float* As = lds;
  float* Bs = lds + TA::LDS_FLOATS;

  if (p.prio) __builtin_amdgcn_s_setprio(2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, p.nblk);
  const int m0 = (tile / p.tiles_n) * BM;
  const int n0 = (tile % p.tiles_n) * BNG;

  f32x16 acc[TM][G];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][g][r] = 0.f;

  TA ta;
  TB tb;
  ta.rows_init(p.lda, m0, p.M, nullptr, tid);   // rows beyond M clamp to the last one (never stored)
#pragma unroll
  for (int i = 0; i < TB::NV; ++i) {
    const int row = tid / 8 + i * 32;            // LDS row = gate * 64 + hidden unit
    tb.ro[i] = ((size_t)(row / BNG) * p.HP + n0 + row % BNG) * p.ldw;
  }
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[TM][3], bf[G][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) af[i][q] = TA::frag(As, q, wm * (BM / 2) + i * 32, ks, lane);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int q = 0; q < 3; ++q) bf[g][q] = TB::frag(Bs, q, g * BNG + wn * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          // the six piece products with i + j <= 4, small ones first; n-operand first (C^T in the accumulator)
          acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[g][0], af[i][2], acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[g][2], af[i][0], acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[g][1], af[i][1], acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[g][0], af[i][1], acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[g][1], af[i][0], acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[g][0], af[i][0], acc[i][g], 0, 0, 0);
        }
    }
  };
  // K is a multiple of 32 (HP is a multiple of 64): no load needs a predicate
  ta.template load<false>(p.A, p.lda, m0, p.M, 0, p.K, tid);
  tb.template load<false>(p.W, p.ldw, 0, 0, 0, p.K, tid);
  ta.store(As, tid);
  tb.store(Bs, tid);
  __syncthreads();
  auto refill_from = [&](const TA& a, const TB& b) {
    __syncthreads();
    a.store(As, tid);
    b.store(Bs, tid);
    __syncthreads();
  };
  int k0 = 0;
  if constexpr (PF == 2) {
    if (k0 + 64 <= p.K) {
      TA ta2 = ta;
      TB tb2 = tb;
      ta.template load<false>(p.A, p.lda, m0, p.M, k0 + 32, p.K, tid);
      tb.template load<false>(p.W, p.ldw, 0, 0, k0 + 32, p.K, tid);
      for (; k0 + 128 <= p.K; k0 += 64) {
        ta2.template load<false>(p.A, p.lda, m0, p.M, k0 + 64, p.K, tid);
        tb2.template load<false>(p.W, p.ldw, 0, 0, k0 + 64, p.K, tid);
        compute();
        refill_from(ta, tb);
        ta.template load<false>(p.A, p.lda, m0, p.M, k0 + 96, p.K, tid);
        tb.template load<false>(p.W, p.ldw, 0, 0, k0 + 96, p.K, tid);
        compute();
        refill_from(ta2, tb2);
      }
      compute();
      refill_from(ta, tb);
      k0 += 32;
    }
  }
  for (; k0 + 32 < p.K; k0 += 32) {
    ta.template load<false>(p.A, p.lda, m0, p.M, k0 + 32, p.K, tid);
    tb.template load<false>(p.W, p.ldw, 0, 0, k0 + 32, p.K, tid);
    compute();
    if constexpr (PF == 1) __builtin_amdgcn_sched_barrier(0);
    refill_from(ta, tb);
  }
  compute();

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / 2) + i * 32 + (lane & 31);
    if (m >= p.M) continue;
    const auto rw = row_meta(e, p.HP, m);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = n0 + wn * 32 + 8 * q + 4 * (lane >> 5);
      float4 v[G];
#pragma unroll
      for (int g = 0; g < G; ++g)
        v[g] = make_float4(acc[i][g][4 * q], acc[i][g][4 * q + 1], acc[i][g][4 * q + 2], acc[i][g][4 * q + 3]);
      epilogue(e, p.HP, rw, col, v, nullptr);
    }
  }
}


// ---------------------------------------------------------------------------
// Small levels (<= ~1024 rows: the per-GPU shard of a strong-scaled batch, config 1, the thin deep levels of
// large graphs).  There a 64-row tile grid leaves most CUs empty and every workgroup walks the whole reduction
// alone (one 64x64x512 tile is 6.8 us of MFMA issue on its CU; splitting K over workgroups costs slabs and a
// combine).  This kernel takes a 32-row x (G x 32-column) output tile per workgroup -- 256 workgroups for
// 512 rows x 512 units -- and splits K over its FOUR WAVES: wave w stages and multiplies the k-tiles
// w, w+4, .. on its own (wave-private LDS staging, no workgroup barrier in the k loop, so the waves drift apart
// and hide each other's load latency), then the four partial accumulators meet in LDS in a fixed order
// (deterministic) and wave q finishes register group q of every gate: the same per-lane layout as
// level_kernel's epilogue, so the GRU / gate-map epilogues are shared.  BKM = 1: the weight operand is
// k-major ([K][N], the backward products dX = dY . W).
// ---------------------------------------------------------------------------
constexpr int SBK = 16;            // k-tile of one wave
constexpr int SLD = SBK + 4;       // row stride of an m-major staging tile
constexpr int SKN = 40;            // row stride of a k-major staging tile ([k][32 n] padded: half-waves on disjoint banks)

struct PlainEpi {   // C[row][g*HP + col] (+)= v[g] (+ bias)
  float* C; int ldc; int accumulate; const float* bias;
};
struct PlainRow { int r; };
__device__ __forceinline__ PlainRow row_meta(const PlainEpi&, int, int m) { return PlainRow{m}; }
template <int G>
__device__ __forceinline__ void epilogue_plain(const PlainEpi& e, int HP, const PlainRow& rw, int col, const float4* v) {
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float* dst = e.C + (size_t)rw.r * e.ldc + g * HP + col;
    float4 x = v[g];
    if (e.bias) x = add4(x, ld4(e.bias + g * HP + col));
    if (e.accumulate) x = add4(x, ld4(dst));
    st4(dst, x);
  }
}

// lin01 chain of the bidirectional encoder (dvae_model.py:119): hv_{t+1} = W01a.hv_t + (W01b.hb + b01) for the graphs that
// still have a reverse-sweep step t, hv_t for the others.  Q holds W01b.hb + b01 per reverse-sweep node row (one GEMM).
struct ChainFwdEpi {
  const int* sizes; int N, t; const int* node2row;
  const float* hv_t; const float* Q; float* hv_next;
};
struct ChainFwdRow { int b; };
__device__ __forceinline__ ChainFwdRow row_meta(const ChainFwdEpi&, int, int m) { return ChainFwdRow{m}; }
// backward of one chain step: d_t = d_next.W01a for the active graphs (their d_next row also goes to DQ, the gradient of
// Q's row, and to DA, the masked copy the weight gradient sums over), d_next for the finished ones
struct ChainBwdEpi {
  const int* sizes; int N, t; const int* node2row;
  const float* d_next; float* d_t; float* DQ; float* DA;
};
struct ChainBwdRow { int b; int qrow; };         // qrow < 0: finished
__device__ __forceinline__ ChainBwdRow row_meta(const ChainBwdEpi& e, int, int m) {
  const int n = e.sizes ? e.sizes[m] : e.N;
  return ChainBwdRow{m, e.t < n ? e.node2row[m * e.N + n - 1 - e.t] : -1};
}

template <int G> __device__ __forceinline__ void small_epilogue(const ChainFwdEpi& e, int HP, const ChainFwdRow& rw, int col, const float4* v) {
  const int n = e.sizes ? e.sizes[rw.b] : e.N;
  float4 x;
  if (e.t < n) x = add4(v[0], ld4(e.Q + (size_t)e.node2row[rw.b * e.N + n - 1 - e.t] * HP + col));
  else x = ld4(e.hv_t + (size_t)rw.b * HP + col);
  st4(e.hv_next + (size_t)rw.b * HP + col, x);
}
template <int G> __device__ __forceinline__ void small_epilogue(const ChainBwdEpi& e, int HP, const ChainBwdRow& rw, int col, const float4* v) {
  const float4 dn = ld4(e.d_next + (size_t)rw.b * HP + col);
  if (rw.qrow >= 0) {
    st4(e.d_t + (size_t)rw.b * HP + col, v[0]);
    st4(e.DQ + (size_t)rw.qrow * HP + col, dn);
    st4(e.DA + (size_t)rw.b * HP + col, dn);
  } else {
    st4(e.d_t + (size_t)rw.b * HP + col, dn);
    st4(e.DA + (size_t)rw.b * HP + col, make_float4(0.f, 0.f, 0.f, 0.f));
  }
}
// prefix sweeps, backward: dL/dGM[u] = (direct part, gathered into DGM) + dPG[u].Whh, then the gate/map backward
// (kernels.hip gm_bwd_apply_kernel's formulas) on the product's own tile: GMP (g, m) -> (dg_pre, dm)
struct GmBwdEpi { int r_base; const float* DGM; float* GMP; };
struct GmBwdRow { int r; };
__device__ __forceinline__ GmBwdRow row_meta(const GmBwdEpi& e, int, int m) { return GmBwdRow{e.r_base + m}; }
template <int G> __device__ __forceinline__ void small_epilogue(const GmBwdEpi& e, int HP, const GmBwdRow& rw, int col, const float4* v) {
  const float4 d = add4(v[0], ld4(e.DGM + (size_t)rw.r * HP + col));
  float* gp = e.GMP + (size_t)rw.r * 2 * HP;
  const float4 g = ld4(gp + col), m = ld4(gp + HP + col);
  st4(gp + col, make_float4(d.x * m.x * g.x * (1.f - g.x), d.y * m.y * g.y * (1.f - g.y),
                            d.z * m.z * g.z * (1.f - g.z), d.w * m.w * g.w * (1.f - g.w)));
  st4(gp + HP + col, make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w));
}
// node sweeps, backward: dL/dh of node rows = what DH holds + dGMP.[Wg;Wm] (this product), then the GRUCell backward
// (kernels.hip gru_bwd_kernel's formulas) on the product's own tile
struct GruBwdEpi { int r_base; float* GH; float* GHN; const float* AGG; float* DH; };
struct GruBwdRow { int p; };
__device__ __forceinline__ GruBwdRow row_meta(const GruBwdEpi& e, int, int m) { return GruBwdRow{e.r_base + m}; }
template <int G> __device__ __forceinline__ void small_epilogue(const GruBwdEpi& e, int HP, const GruBwdRow& rw, int col, const float4* v) {
  const float4 dh = add4(v[0], ld4(e.DH + (size_t)rw.p * HP + col));
  float* gh = e.GH + (size_t)rw.p * 3 * HP;
  const float4 rr = ld4(gh + col), zz = ld4(gh + HP + col), nn = ld4(gh + 2 * HP + col);
  const float4 ghn = ld4(e.GHN + (size_t)rw.p * HP + col);
  const float4 a = ld4(e.AGG + (size_t)rw.p * HP + col);
  float4 drp, dzp, dnp, dnr, da;
#define BWD_LANE(f)                                              \
  dnp.f = dh.f * (1.f - zz.f) * (1.f - nn.f * nn.f);              \
  dzp.f = dh.f * (a.f - nn.f) * zz.f * (1.f - zz.f);              \
  drp.f = dnp.f * ghn.f * rr.f * (1.f - rr.f);                    \
  dnr.f = dnp.f * rr.f;                                           \
  da.f = dh.f * zz.f;
  BWD_LANE(x) BWD_LANE(y) BWD_LANE(z) BWD_LANE(w)
#undef BWD_LANE
  st4(gh + col, drp);
  st4(gh + HP + col, dzp);
  st4(gh + 2 * HP + col, dnr);
  st4(e.GHN + (size_t)rw.p * HP + col, dnp);
  st4(e.DH + (size_t)rw.p * HP + col, da);
}
template <int G> __device__ __forceinline__ void small_epilogue(const GruEpi& e, int HP, const GruRow& rw, int col, const float4* v) { epilogue(e, HP, rw, col, v); }
template <int G> __device__ __forceinline__ void small_epilogue(const GmEpi& e, int HP, const GmRow& rw, int col, const float4* v) { epilogue(e, HP, rw, col, v); }
template <int G> __device__ __forceinline__ void small_epilogue(const PlainEpi& e, int HP, const PlainRow& rw, int col, const float4* v) { epilogue_plain<G>(e, HP, rw, col, v); }

// NW: waves per workgroup = ways the reduction is split (4, or 8: half as many sequential k-tiles per wave -- the kernel
// runs one workgroup per CU at these sizes, i.e. one or two waves per SIMD, and is bound by that chain's latency)
template <int G, class Epi, int BKM, int NW>
__global__ __launch_bounds__(NW * 64) void small_kernel(LevelCore p, Epi e) {
  constexpr int A_FL = 32 * SLD;
  constexpr int B_FL = BKM ? G * SBK * SKN : G * 32 * SLD;
  constexpr int STAGE = A_FL + B_FL;                 // per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.prio) __builtin_amdgcn_s_setprio(2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x;
  const int m0 = (tile / p.tiles_n) * 32;
  const int n0 = (tile % p.tiles_n) * 32;
  float* As = lds + wave * STAGE;
  float* Bs = As + A_FL;

  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  // this lane's share of a k-tile: A 32 x 16 floats = 2 float4, B G x (32 x 16) floats = 2G float4
  const float* pa[2];
  const float* pb[2 * G];
  int sa[2], sb[2 * G];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = lane + 64 * i, row = idx >> 2, kq = (idx & 3) * 4;
    pa[i] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + kq;
    sa[i] = row * SLD + kq;
  }
#pragma unroll
  for (int i = 0; i < 2 * G; ++i) {
    const int idx = lane + 64 * i;
    if (BKM == 0) {
      const int row = idx >> 2, kq = (idx & 3) * 4;          // row = gate * 32 + unit
      pb[i] = p.W + ((size_t)(row >> 5) * p.HP + n0 + (row & 31)) * p.ldw + kq;
      sb[i] = row * SLD + kq;
    } else {
      const int g = idx >> 7, kr = (idx & 127) >> 3, nq = (idx & 7) * 4;   // per gate: 16 k-rows x 8 float4
      pb[i] = p.W + (size_t)kr * p.ldw + g * p.HP + n0 + nq;
      sb[i] = g * SBK * SKN + kr * SKN + nq;
    }
  }
  const int ntiles = p.K / SBK;
  float4 va[2], vb[2 * G];
  auto load = [&](int t) {
    const size_t ka = (size_t)t * SBK;
#pragma unroll
    for (int i = 0; i < 2; ++i) va[i] = ld4(pa[i] + ka);
#pragma unroll
    for (int i = 0; i < 2 * G; ++i) vb[i] = ld4(pb[i] + (BKM ? ka * p.ldw : ka));
  };
  int t = wave;
  if (t < ntiles) load(t);
  for (; t < ntiles; t += NW) {
    // registers -> this wave's staging tiles (the previous tile's fragments are consumed: same wave, in order)
#pragma unroll
    for (int i = 0; i < 2; ++i) st4(As + sa[i], va[i]);
#pragma unroll
    for (int i = 0; i < 2 * G; ++i) st4(Bs + sb[i], vb[i]);
    if (t + NW < ntiles) load(t + NW);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int kc = 0; kc < SBK / 8; ++kc) {
      const float4 af = ld4(As + (lane & 31) * SLD + kc * 8 + (lane >> 5) * 4);
      float4 bf[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (BKM == 0) {
          bf[g] = ld4(Bs + (g * 32 + (lane & 31)) * SLD + kc * 8 + (lane >> 5) * 4);
        } else {
          const float* q = Bs + g * SBK * SKN + (kc * 8 + (lane >> 5) * 4) * SKN + (lane & 31);
          bf[g] = make_float4(q[0], q[SKN], q[2 * SKN], q[3 * SKN]);
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g].x, af.x, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g].y, af.y, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g].z, af.z, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g].w, af.w, acc[g], 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // the NW partial accumulators meet in LDS (over the staging area): red[wave][reg][lane]
  __syncthreads();
  float* red = lds;
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * G * 16 + g * 16 + r) * 64 + lane] = acc[g][r];
  __syncthreads();
  // wave q finishes register group q of every gate: row m0 + (lane & 31), columns n0 + 8q + 4*(lane>>5) + 0..3
  const int q = wave;
  const int m = m0 + (lane & 31);
  if (m >= p.M || q >= 4) return;     // the first four waves finish one register group each
  float4 v[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = g * 16 + 4 * q + c;
      float acc_s = red[r * 64 + lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) acc_s += red[(w * G * 16 + r) * 64 + lane];   // fixed order: wave 0, 1, ..
      s[c] = acc_s;
    }
    v[g] = make_float4(s[0], s[1], s[2], s[3]);
  }
  const auto rw = row_meta(e, p.HP, m);
  const int col = n0 + 8 * q + 4 * (lane >> 5);
  small_epilogue<G>(e, p.HP, rw, col, v);
}

template <int G, class Epi, bool GATHER = false>
int launch_level(LevelCore c, const Epi& e, hipStream_t st) {
  c.tiles_n = c.HP / BNG;
  const bool x6 = !GATHER && gemm_x6_enabled() && c.K % 32 == 0;
  // 128-row tiles once they still give every CU two workgroups (the three-gate x6 tile of 128 rows would spill: 64 rows)
  const bool big = (long long)cdiv(c.M, 128) * c.tiles_n >= 512 && !(x6 && G == 3);
  void* pr = gemm_prof_begin(2.0 * c.M * G * c.HP * (double)c.K, st);
  c.nblk = cdiv(c.M, big ? 128 : 64) * c.tiles_n;
  if constexpr (!GATHER) {
    // prefetch form (as gemm.hip's pf_of): two k-tiles in flight by default -- same box headline -0.4 %, c3 -1.1 %, c2 neutral
    static const int pf = getenv("DVAE_LEVEL_PF") ? atoi(getenv("DVAE_LEVEL_PF")) : 2;   // A-B knob
    if constexpr (G != 3) {
      if (x6 && big) {
        if (pf == 1) hipLaunchKernelGGL((level_x6_kernel<G, 2, Epi, 1>), dim3(c.nblk), dim3(256), 0, st, c, e);
        else if (pf == 2) hipLaunchKernelGGL((level_x6_kernel<G, 2, Epi, 2>), dim3(c.nblk), dim3(256), 0, st, c, e);
        else hipLaunchKernelGGL((level_x6_kernel<G, 2, Epi>), dim3(c.nblk), dim3(256), 0, st, c, e);
      }
    }
    if (x6 && !big) {
      if (pf == 1) hipLaunchKernelGGL((level_x6_kernel<G, 1, Epi, 1>), dim3(c.nblk), dim3(256), 0, st, c, e);
      else if (pf == 2) hipLaunchKernelGGL((level_x6_kernel<G, 1, Epi, 2>), dim3(c.nblk), dim3(256), 0, st, c, e);
      else hipLaunchKernelGGL((level_x6_kernel<G, 1, Epi>), dim3(c.nblk), dim3(256), 0, st, c, e);
    }
  }
  if (!x6 && big) hipLaunchKernelGGL((level_kernel<G, 2, Epi, GATHER>), dim3(c.nblk), dim3(256), 0, st, c, e);
  if (!x6 && !big) hipLaunchKernelGGL((level_kernel<G, 1, Epi, GATHER>), dim3(c.nblk), dim3(256), 0, st, c, e);
  gemm_prof_end(pr, st);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

constexpr int small_lds_bytes(int G, int BKM, int NW) {
  const int stage = 32 * SLD + (BKM ? G * SBK * SKN : G * 32 * SLD), red = G * 16 * 64;
  return (stage > red ? stage : red) * NW * 4;
}

// waves per workgroup of the small-level kernels: 4 (default) or 8 (DVAE_SMALL_WAVES=8: half as many k-tiles per wave;
// measured neutral -- B=512 2.33 vs 2.32 ms, B=32 1.62 vs 1.61, same box -- the step is not bound by these kernels' inner loops)
static int g_small_waves = getenv("DVAE_SMALL_WAVES") ? atoi(getenv("DVAE_SMALL_WAVES")) : 4;

template <int G, class Epi, int BKM, int NW>
int launch_small_nw(const LevelCore& c, const Epi& e, hipStream_t st) {
  constexpr int LDSB = small_lds_bytes(G, BKM, NW);
  if (LDSB > 64 * 1024) {   // more than 64 KB of LDS is opt-in, once per device
    static std::once_flag big_lds[16];
    int dev = 0;
    DVAE_CHECK_HIP(hipGetDevice(&dev));
    hipError_t err = hipSuccess;
    std::call_once(big_lds[dev & 15], [&] {
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(small_kernel<G, Epi, BKM, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    });
    DVAE_CHECK_HIP(err);
  }
  hipLaunchKernelGGL((small_kernel<G, Epi, BKM, NW>), dim3(c.nblk), dim3(NW * 64), LDSB, st, c, e);
  return DVAE_OK;
}

template <int G, class Epi, int BKM>
int launch_small(LevelCore c, const Epi& e, hipStream_t st) {
  c.tiles_n = c.HP / 32;
  c.nblk = cdiv(c.M, 32) * c.tiles_n;
  void* pr = gemm_prof_begin(2.0 * c.M * G * c.HP * (double)c.K, st);
  // eight waves only where every wave still gets two or more k-tiles
  const int rc = (g_small_waves == 8 && c.K / SBK >= 16) ? launch_small_nw<G, Epi, BKM, 8>(c, e, st)
                                                          : launch_small_nw<G, Epi, BKM, 4>(c, e, st);
  gemm_prof_end(pr, st);
  DVAE_TRY(rc);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

}  // namespace

// plain products on small row counts: C[M][G*HP] (+)= A[M][K] . W^T, W = [G*HP][K] (w_kmajor = 0) or
// C[M][HP] (+)= A[M][K] . W, W = [K][HP] (w_kmajor = 1, G = 1)
int k_small_gemm(const float* A, int lda, const float* W, int ldw, int w_kmajor, float* C, int ldc, int M, int G, int HP,
                 int K, int accumulate, int prio, hipStream_t st) {
  if (M <= 0) return DVAE_OK;
  DVAE_REQUIRE(HP % 32 == 0 && K % SBK == 0 && G >= 1 && G <= 3 && (!w_kmajor || G == 1), "small_gemm: unsupported shape");
  LevelCore c{};
  c.A = A; c.lda = lda; c.M = M; c.W = W; c.ldw = ldw; c.K = K; c.HP = HP; c.prio = prio;
  PlainEpi e{C, ldc, accumulate, nullptr};
  if (w_kmajor) return launch_small<1, PlainEpi, 1>(c, e, st);
  if (G == 1) return launch_small<1, PlainEpi, 0>(c, e, st);
  if (G == 2) return launch_small<2, PlainEpi, 0>(c, e, st);
  return launch_small<3, PlainEpi, 0>(c, e, st);
}

// one step of the lin01 chain, forward: hv_next[b] = hv_t[b].W01a^T + Q[row(b, t)] (graphs with a step t) | hv_t[b]
int k_chain_fwd_step(const SchedDir& Srev, const int* sizes, int B, int N, int t, const float* hv_t, const float* W01, int ldw,
                     const float* Q, float* hv_next, int HP, int prio, hipStream_t st) {
  if (B <= 0) return DVAE_OK;
  DVAE_REQUIRE(HP % 32 == 0, "chain_fwd_step: HP must be a multiple of 32");
  LevelCore c{};
  c.A = hv_t; c.lda = HP; c.M = B; c.W = W01; c.ldw = ldw; c.K = HP; c.HP = HP; c.prio = prio;
  ChainFwdEpi e{sizes, N, t, Srev.node2row, hv_t, Q, hv_next};
  return launch_small<1, ChainFwdEpi, 0>(c, e, st);
}

// backward: d_t[b] = d_next[b].W01a (graphs with a step t; DQ[row(b, t)] = DA[b] = d_next[b]) | d_next[b] (DA[b] = 0)
int k_chain_bwd_step(const SchedDir& Srev, const int* sizes, int B, int N, int t, const float* d_next, const float* W01, int ldw,
                     float* d_t, float* DQ, float* DA, int HP, int prio, hipStream_t st) {
  if (B <= 0) return DVAE_OK;
  DVAE_REQUIRE(HP % 32 == 0, "chain_bwd_step: HP must be a multiple of 32");
  LevelCore c{};
  c.A = d_next; c.lda = HP; c.M = B; c.W = W01; c.ldw = ldw; c.K = HP; c.HP = HP; c.prio = prio;
  ChainBwdEpi e{sizes, N, t, Srev.node2row, d_next, d_t, DQ, DA};
  return launch_small<1, ChainBwdEpi, 1>(c, e, st);
}

// prefix sweeps, backward, node rows [ns,ne): hidden part of dL/dGM (dPG.Whh) + gate/map backward, one launch
int k_level_gm_bwd_small(int ns, int ne, const SweepBufs& b, const float* Whh, int HP, int prio, hipStream_t st) {
  if (ne <= ns) return DVAE_OK;
  DVAE_REQUIRE(HP % 32 == 0, "level_gm_bwd_small: HP must be a multiple of 32");
  LevelCore c{};
  c.A = b.PG + (size_t)ns * 3 * HP; c.lda = 3 * HP; c.M = ne - ns;
  c.W = Whh; c.ldw = HP; c.K = 3 * HP; c.HP = HP; c.prio = prio;
  GmBwdEpi e{ns, b.DGM, b.GMP};
  return launch_small<1, GmBwdEpi, 1>(c, e, st);
}

// node sweeps, backward, node rows [ns,ne): gate/map input gradient (dGMP.[Wg;Wm]) into dL/dh + GRUCell backward, one launch
int k_level_gru_bwd_small(int ns, int ne, const SweepBufs& b, const float* Wgm, int HP, int prio, hipStream_t st) {
  if (ne <= ns) return DVAE_OK;
  DVAE_REQUIRE(HP % 32 == 0 && !b.prefix, "level_gru_bwd_small: node sweeps, HP a multiple of 32");
  LevelCore c{};
  c.A = b.GMP + (size_t)ns * 2 * HP; c.lda = 2 * HP; c.M = ne - ns;
  c.W = Wgm; c.ldw = HP; c.K = 2 * HP; c.HP = HP; c.prio = prio;
  GruBwdEpi e{ns, b.GH, b.GHN, b.AGG, b.DH};
  return launch_small<1, GruBwdEpi, 1>(c, e, st);
}

int k_level_gru_fwd_small(const SchedDir& S, int p0, int p1, const SweepBufs& b, const float* x, int N, int C,
                          const GruW& w, int HP, int prio, hipStream_t st) {
  if (p1 <= p0) return DVAE_OK;
  DVAE_REQUIRE(HP % 32 == 0, "level_gru_small: HP must be a multiple of 32");
  LevelCore c{};
  c.A = b.AGG + (size_t)p0 * HP; c.lda = HP; c.M = p1 - p0;
  c.W = w.Whh; c.ldw = HP; c.K = HP; c.HP = HP; c.prio = prio;
  GruEpi e{S, b.R, b.prefix, p0, x, N, C, w, b.AGG, b.GH, b.GHN, b.HV, b.Hn};
  return launch_small<3, GruEpi, 0>(c, e, st);
}

int k_level_gm_fwd_small(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* Wgm, const float* posb,
                         int HP, int prio, hipStream_t st) {
  if (ne <= ns) return DVAE_OK;
  DVAE_REQUIRE(HP % 32 == 0, "level_gm_small: HP must be a multiple of 32");
  LevelCore c{};
  c.A = b.Hn + (size_t)ns * HP; c.lda = HP; c.M = ne - ns;
  c.W = Wgm; c.ldw = HP; c.K = HP; c.HP = HP; c.prio = prio;
  GmEpi e{S, ns, posb, b.GMP, b.GM};
  return launch_small<2, GmEpi, 0>(c, e, st);
}

int k_level_gru_fwd(const SchedDir& S, int p0, int p1, const SweepBufs& b, const float* x, int N, int C,
                    const GruW& w, int HP, int prio, hipStream_t st) {
  if (p1 <= p0) return DVAE_OK;
  DVAE_REQUIRE(HP % BNG == 0, "level_gru: HP must be a multiple of 64");
  LevelCore c{};
  c.A = b.AGG + (size_t)p0 * HP; c.lda = HP; c.M = p1 - p0;
  c.W = w.Whh; c.ldw = HP; c.K = HP; c.HP = HP; c.prio = prio;
  GruEpi e{S, b.R, b.prefix, p0, x, N, C, w, b.AGG, b.GH, b.GHN, b.HV, b.Hn};
  return launch_level<3>(c, e, st);
}

// node sweeps (encoder, both directions), levels >= 1: aggregation + hidden GEMM + GRU gates of node rows
// [ns,ne) in ONE launch; also writes the AGG rows the backward pass needs
int k_level_agg_gru_fwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* x, int N, int C,
                        const GruW& w, int HP, int prio, hipStream_t st) {
  if (ne <= ns) return DVAE_OK;
  DVAE_REQUIRE(HP % BNG == 0, "level_agg_gru: HP must be a multiple of 64");
  DVAE_REQUIRE(!b.prefix, "level_agg_gru: node sweeps only");
  LevelCore c{};
  c.A = nullptr; c.lda = HP; c.M = ne - ns;
  c.W = w.Whh; c.ldw = HP; c.K = HP; c.HP = HP; c.prio = prio;
  c.GM = b.GM; c.csr_ptr = S.csr_ptr; c.csr_idx = S.csr_idx; c.r_base = ns; c.AGG = b.AGG;
  GruEpi e{S, b.R, 0, ns, x, N, C, w, b.AGG, b.GH, b.GHN, b.HV, b.Hn};
  return launch_level<3, GruEpi, true>(c, e, st);
}

int k_level_gm_fwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* Wgm, const float* posb,
                   int HP, int prio, hipStream_t st) {
  if (ne <= ns) return DVAE_OK;
  DVAE_REQUIRE(HP % BNG == 0, "level_gm: HP must be a multiple of 64");
  LevelCore c{};
  c.A = b.Hn + (size_t)ns * HP; c.lda = HP; c.M = ne - ns;
  c.W = Wgm; c.ldw = HP; c.K = HP; c.HP = HP; c.prio = prio;
  GmEpi e{S, ns, posb, b.GMP, b.GM};
  return launch_level<2>(c, e, st);
}

// unit-test surface of small_kernel's plain form (include/dvae_hip.h)
extern "C" int dvae_small_gemm(const float* A, int lda, const float* W, int ldw, int w_kmajor, float* C, int ldc,
                               int M, int G, int HP, int K, int accumulate, void* stream) {
  DVAE_REQUIRE(A && W && C, "small_gemm: null pointer");
  DVAE_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0 && (lda & 3) == 0 &&
                   (ldw & 3) == 0 && (ldc & 3) == 0, "small_gemm: operands must be 16-byte aligned, strides multiples of 4");
  return k_small_gemm(A, lda, W, ldw, w_kmajor, C, ldc, M, G, HP, K, accumulate, 0, (hipStream_t)stream);
}
