// Bandwidth-bound kernels of the D-VAE path: CSR gated-sum aggregation, GRU
// gate epilogues, gate/map epilogues, their backward forms, the deterministic
// weight-gradient column reductions, the add_edge pair scorer and glue copies.
// No atomics anywhere: every reduction is "partials in a fixed order, then a
// sum", so results are bitwise reproducible run to run.
#include <mutex>

#include "kernels.h"

#include <algorithm>
#include <cstdlib>

namespace {

inline dim3 grid_for(long long items, int cap = 8192) {
  long long g = (items + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return dim3((unsigned)g);
}

#define ITEM_LOOP(total) \
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < (total); it += (long long)gridDim.x * blockDim.x)

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// sum over the 64 lanes of a wave, the same value in every lane.  DPP moves inside the VALU (quad permutes, row
// rotations, row broadcasts; the total lands in lane 63) instead of six ds_bpermute round trips through the LDS
// crossbar: the pair scorers reduce once per scored pair.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xb1, 0xf>(v);    // quad_perm:[1,0,3,2]
  v = dpp_add<0x4e, 0xf>(v);    // quad_perm:[2,3,0,1]
  v = dpp_add<0x124, 0xf>(v);   // row_ror:4
  v = dpp_add<0x128, 0xf>(v);   // row_ror:8      every lane of a row of 16 holds its row's sum
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15   rows 1 and 3 add the sum of the row before
  v = dpp_add<0x143, 0xc>(v);   // row_bcast:31   rows 2 and 3 add lane 31: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---------------------------------------------------------------------------
// aggregation: h_in = sum of gated states of the predecessors (dvae_model.py:77-80)
// prefix mode writes every running sum (decoder suffix states, :223-241).
// ---------------------------------------------------------------------------
__global__ void agg_fwd_kernel(SchedDir S, int R, int ns, int ne, int prefix, const float* __restrict__ GM,
                               float* __restrict__ AGG, const float* __restrict__ h0, int HP,
                               const float* __restrict__ PG, float* __restrict__ GH) {
  const int HP4 = HP >> 2;
  const long long total = (long long)(ne - ns) * HP4;
  ITEM_LOOP(total) {
    const int r = ns + (int)(it / HP4), q = (int)(it % HP4);
    const int p = S.csr_ptr[r], deg = S.csr_ptr[r + 1] - p;
    float4 acc = zero4();
    if (prefix && PG) {
      // running sums of the gated states AND of their hidden products (three gate columns each)
      if (h0 && S.row_node[r] == 0) acc = ld4(h0 + (size_t)S.row_graph[r] * HP + q * 4);
      st4(AGG + (size_t)r * HP + q * 4, acc);
      float4 g0 = zero4(), g1 = zero4(), g2 = zero4();
      for (int t = 0; t < deg; ++t) {
        const size_t u = S.csr_idx[p + t];
        acc = add4(acc, ld4(GM + u * HP + q * 4));
        const float* pu = PG + u * 3 * HP + q * 4;
        g0 = add4(g0, ld4(pu));
        g1 = add4(g1, ld4(pu + HP));
        g2 = add4(g2, ld4(pu + 2 * HP));
        const size_t row = (size_t)R + p + t;
        st4(AGG + row * HP + q * 4, acc);
        float* gh = GH + row * 3 * HP + q * 4;
        st4(gh, g0);
        st4(gh + HP, g1);
        st4(gh + 2 * HP, g2);
      }
    } else if (!prefix) {
      for (int t = 0; t < deg; ++t) acc = add4(acc, ld4(GM + (size_t)S.csr_idx[p + t] * HP + q * 4));
      st4(AGG + (size_t)r * HP + q * 4, acc);
    } else {
      // row r itself holds the k = 0 state (h0 for node 0, else 0); edge rows the running sums
      if (h0 && S.row_node[r] == 0) acc = ld4(h0 + (size_t)S.row_graph[r] * HP + q * 4);
      st4(AGG + (size_t)r * HP + q * 4, acc);
      for (int t = 0; t < deg; ++t) {
        acc = add4(acc, ld4(GM + (size_t)S.csr_idx[p + t] * HP + q * 4));
        st4(AGG + ((size_t)R + p + t) * HP + q * 4, acc);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// GRUCell epilogue (gate order r,z,n; dvae_model.py:83,244).  GH holds agg.Whh^T.
// ---------------------------------------------------------------------------
__global__ void gru_fwd_kernel(SchedDir S, int R, int p0, int p1, int prefix, int no_gh, float* __restrict__ GH,
                               const float* __restrict__ AGG, float* __restrict__ GHN,
                               float* __restrict__ HV, float* __restrict__ Hn,
                               const float* __restrict__ x, int N, int C, GruW w, int HP, GemmPartials parts) {
  const int HP4 = HP >> 2;
  const long long total = (long long)(p1 - p0) * HP4;
  ITEM_LOOP(total) {
    const int p = p0 + (int)(it / HP4), q = (int)(it % HP4);
    const int r = prefix ? S.prow_node[p] : p;
    const int b = S.row_graph[r], i = S.row_node[r];
    const float* xr = x + ((size_t)b * N + i) * C;
    const int j = q * 4;
    float4 gi[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) gi[g] = ld4(w.bih + g * HP + j);
    for (int c = 0; c < C; ++c) {
      const float xv = xr[c];
      if (xv != 0.f) {
#pragma unroll
        for (int g = 0; g < 3; ++g) gi[g] = fma4(xv, ld4(w.WihT + (size_t)c * 3 * HP + g * HP + j), gi[g]);
      }
    }
    float* gh = GH + (size_t)p * 3 * HP;
    // no_gh: h_in = 0 for these rows, so the hidden product is just the bias
    float4 ghr = ld4(w.bhh + j), ghz = ld4(w.bhh + HP + j), ghn = ld4(w.bhh + 2 * HP + j);
    float4 a = zero4();
    if (!no_gh) {
      if (parts.slab) {   // the hidden product arrives as split-K slabs: add them up in slab order
        const float* sp = parts.slab + (size_t)(p - p0) * 3 * HP + j;
        float4 sr = zero4(), sz = zero4(), sn = zero4();
        for (int k = 0; k < parts.splits; ++k) {
          const float* sk = sp + (size_t)k * parts.stride;
          sr = add4(sr, ld4(sk));
          sz = add4(sz, ld4(sk + HP));
          sn = add4(sn, ld4(sk + 2 * HP));
        }
        ghr = add4(ghr, sr); ghz = add4(ghz, sz); ghn = add4(ghn, sn);
      } else {
        ghr = add4(ghr, ld4(gh + j));
        ghz = add4(ghz, ld4(gh + HP + j));
        ghn = add4(ghn, ld4(gh + 2 * HP + j));
      }
      a = ld4(AGG + (size_t)p * HP + j);
    }
    float4 rr, zz, nn, hh;
#define GRU_LANE(f)                                   \
    rr.f = sigm(gi[0].f + ghr.f);                     \
    zz.f = sigm(gi[1].f + ghz.f);                     \
    nn.f = tanhf(gi[2].f + rr.f * ghn.f);             \
    hh.f = (1.f - zz.f) * nn.f + zz.f * a.f;
    GRU_LANE(x) GRU_LANE(y) GRU_LANE(z) GRU_LANE(w)
#undef GRU_LANE
    st4(gh + j, rr);
    st4(gh + HP + j, zz);
    st4(gh + 2 * HP + j, nn);
    st4(GHN + (size_t)p * HP + j, ghn);
    st4(HV + (size_t)p * HP + j, hh);
    if (prefix && p == final_prow(S, R, r)) st4(Hn + (size_t)r * HP + j, hh);
  }
}

// Prefix sweeps with one hidden product per node (SweepBufs::PG): aggregation and GRUCell epilogue of
// every edge prefix row of node rows [ns, ne) in one pass.  Row k of a node adds predecessor k's gated
// state to the running h_in and its hidden product PG to the running W_hh.h_in, then applies the
// gates; the input part W_ih.x + b_ih is the node's and is computed once.  Nothing of the running
// sums goes through memory except the h_in rows the backward pass reads (AGG).
// SAVE = false: the backward pass recomputes the gates from GM / PG (prefix_gru_bwd_kernel), so only the states are written
template <bool SAVE>
__global__ void prefix_gru_fwd_kernel(SchedDir S, int R, int ns, int ne, const float* __restrict__ GM,
                                      const float* __restrict__ PG, float* __restrict__ AGG,
                                      float* __restrict__ GH, float* __restrict__ GHN, float* __restrict__ HV,
                                      float* __restrict__ Hn, const float* __restrict__ x, int N, int C, GruW w,
                                      int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)(ne - ns) * HP4;
  ITEM_LOOP(total) {
    const int r = ns + (int)(it / HP4), j = (int)(it % HP4) * 4;
    const int p = S.csr_ptr[r], deg = S.csr_ptr[r + 1] - p;
    st4(AGG + (size_t)r * HP + j, zero4());     // the node's own (k = 0) row: h_in = 0 at levels >= 1 (its backward reads it)
    if (deg == 0) continue;
    const float* xr = x + ((size_t)S.row_graph[r] * N + S.row_node[r]) * C;
    float4 gi[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) gi[g] = ld4(w.bih + g * HP + j);
    for (int c = 0; c < C; ++c) {
      const float xv = xr[c];
      if (xv != 0.f) {
#pragma unroll
        for (int g = 0; g < 3; ++g) gi[g] = fma4(xv, ld4(w.WihT + (size_t)c * 3 * HP + g * HP + j), gi[g]);
      }
    }
    const float4 br = ld4(w.bhh + j), bz = ld4(w.bhh + HP + j), bn = ld4(w.bhh + 2 * HP + j);
    float4 a = zero4(), g0 = zero4(), g1 = zero4(), g2 = zero4();
    for (int t = 0; t < deg; ++t) {
      const size_t u = S.csr_idx[p + t];
      a = add4(a, ld4(GM + u * HP + j));
      const float* pu = PG + u * 3 * HP + j;
      g0 = add4(g0, ld4(pu));
      g1 = add4(g1, ld4(pu + HP));
      g2 = add4(g2, ld4(pu + 2 * HP));
      const float4 ghr = add4(br, g0), ghz = add4(bz, g1), ghn = add4(bn, g2);
      float4 rr, zz, nn, hh;
#define PG_LANE(f)                                    \
      rr.f = sigm(gi[0].f + ghr.f);                     \
      zz.f = sigm(gi[1].f + ghz.f);                     \
      nn.f = tanhf(gi[2].f + rr.f * ghn.f);             \
      hh.f = (1.f - zz.f) * nn.f + zz.f * a.f;
      PG_LANE(x) PG_LANE(y) PG_LANE(z) PG_LANE(w)
#undef PG_LANE
      const size_t row = (size_t)R + p + t;
      if (SAVE) {
        st4(AGG + row * HP + j, a);
        float* gh = GH + row * 3 * HP + j;
        st4(gh, rr);
        st4(gh + HP, zz);
        st4(gh + 2 * HP, nn);
        st4(GHN + row * HP + j, ghn);
      }
      st4(HV + row * HP + j, hh);
      if (t == deg - 1) st4(Hn + (size_t)r * HP + j, hh);
    }
  }
}

// The backward of prefix_gru_fwd_kernel<false>: the same walk over a node's predecessors rebuilds every prefix row's h_in,
// hidden product and gates from GM / PG (the same operations in the same order: the same bits as the forward pass had),
// reads the row's dL/dHV and writes what gru_bwd_kernel writes -- dgh into GH, dn_pre into GHN, dL/dh_in into DH.  The five
// activation tables per prefix row are never stored: the forward pass writes one table instead of six, this pass reads
// GM / PG rows that its neighbours share through L2 instead of five private rows.  PG of the predecessors still holds the
// forward products here: dgm_gather overwrites a level's PG rows only after the levels above it are through.
__global__ void prefix_gru_bwd_kernel(SchedDir S, int R, int ns, int ne, const float* __restrict__ GM,
                                      const float* __restrict__ PG, float* __restrict__ GH, float* __restrict__ GHN,
                                      float* __restrict__ DH, const float* __restrict__ DHn, const float* __restrict__ x,
                                      int N, int C, GruW w, int HP, GemmPartials parts, int parts_row0,
                                      const int* __restrict__ apos) {
  const int HP4 = HP >> 2;
  const long long total = (long long)(ne - ns) * HP4;
  ITEM_LOOP(total) {
    const int r = ns + (int)(it / HP4), j = (int)(it % HP4) * 4;
    const int p = S.csr_ptr[r], deg = S.csr_ptr[r + 1] - p;
    if (deg == 0) continue;
    const float* xr = x + ((size_t)S.row_graph[r] * N + S.row_node[r]) * C;
    float4 gi[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) gi[g] = ld4(w.bih + g * HP + j);
    for (int c = 0; c < C; ++c) {
      const float xv = xr[c];
      if (xv != 0.f) {
#pragma unroll
        for (int g = 0; g < 3; ++g) gi[g] = fma4(xv, ld4(w.WihT + (size_t)c * 3 * HP + g * HP + j), gi[g]);
      }
    }
    const float4 br = ld4(w.bhh + j), bz = ld4(w.bhh + HP + j), bn = ld4(w.bhh + 2 * HP + j);
    float4 a = zero4(), g0 = zero4(), g1 = zero4(), g2 = zero4();
    for (int t = 0; t < deg; ++t) {
      const size_t u = S.csr_idx[p + t];
      a = add4(a, ld4(GM + u * HP + j));
      const float* pu = PG + u * 3 * HP + j;
      g0 = add4(g0, ld4(pu));
      g1 = add4(g1, ld4(pu + HP));
      g2 = add4(g2, ld4(pu + 2 * HP));
      const float4 ghr = add4(br, g0), ghz = add4(bz, g1), ghn = add4(bn, g2);
      const size_t row = (size_t)R + p + t;
      // an unscored row (only a node's full prefix can be one) has no gradient of its own: nothing was written there
      float4 dh = (!apos || t < deg - 1 || apos[p + t + 1] != apos[p + t]) ? ld4(DH + row * HP + j) : zero4();
      if (t == deg - 1) {   // the node's final state also takes the gradient of everything that reads the node (gru_bwd_kernel)
        float4 e = ld4(DHn + (size_t)r * HP + j);
        if (parts.slab) {
          const float* sp = parts.slab + (size_t)(r - parts_row0) * HP + j;
          float4 sum = zero4();
          for (int k = 0; k < parts.splits; ++k) sum = add4(sum, ld4(sp + (size_t)k * parts.stride));
          e = add4(sum, e);
        }
        dh = add4(dh, e);
      }
      float4 rr, zz, nn, drp, dzp, dnp, dnr, da;
#define PB_LANE(f)                                              \
      rr.f = sigm(gi[0].f + ghr.f);                               \
      zz.f = sigm(gi[1].f + ghz.f);                               \
      nn.f = tanhf(gi[2].f + rr.f * ghn.f);                       \
      dnp.f = dh.f * (1.f - zz.f) * (1.f - nn.f * nn.f);          \
      dzp.f = dh.f * (a.f - nn.f) * zz.f * (1.f - zz.f);          \
      drp.f = dnp.f * ghn.f * rr.f * (1.f - rr.f);                \
      dnr.f = dnp.f * rr.f;                                       \
      da.f = dh.f * zz.f;
      PB_LANE(x) PB_LANE(y) PB_LANE(z) PB_LANE(w)
#undef PB_LANE
      float* gh = GH + row * 3 * HP + j;
      st4(gh, drp);
      st4(gh + HP, dzp);
      st4(gh + 2 * HP, dnr);
      st4(GHN + row * HP + j, dnp);
      st4(DH + row * HP + j, da);
    }
  }
}

// GRU state of a node of type c whose h_in is 0: TYPEV[c] for c < C, zero rows up to 32.
// Same arithmetic as gru_fwd_kernel's no_gh path on a one-hot encoding.
__global__ void type_states_kernel(int C, GruW w, float* __restrict__ TYPEV, int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)32 * HP4;
  ITEM_LOOP(total) {
    const int c = (int)(it / HP4), j = (int)(it % HP4) * 4;
    float4 hh = zero4();
    if (c < C) {
      float4 gi[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) gi[g] = fma4(1.f, ld4(w.WihT + (size_t)c * 3 * HP + g * HP + j), ld4(w.bih + g * HP + j));
      const float4 ghr = ld4(w.bhh + j), ghz = ld4(w.bhh + HP + j), ghn = ld4(w.bhh + 2 * HP + j);
#define TS_LANE(f)                                                  \
      {                                                             \
        const float rr = sigm(gi[0].f + ghr.f), zz = sigm(gi[1].f + ghz.f); \
        const float nn = tanhf(gi[2].f + rr * ghn.f);               \
        hh.f = (1.f - zz) * nn + zz * 0.f;                          \
      }
      TS_LANE(x) TS_LANE(y) TS_LANE(z) TS_LANE(w)
#undef TS_LANE
    }
    st4(TYPEV + (size_t)c * HP + j, hh);
  }
}

// GRU backward of the type-shared rows.  All rows of a type have the same gates and h_in = 0, so
// the per-row backward is one linear map per type and only the per-type SUM of dL/dhv matters
// (T[c]).  Output: one gradient row per TYPE -- TG [32][3HP] (r, z, n*r gates), TGN [32][HP]
// (n pre-activation) -- and XPT [32][32], the (type one-hot | 1) rows the weight-gradient product
// pairs them with.  Gates are recomputed from the weights (as type_states_kernel).
__global__ void type_gru_bwd_kernel(int C, GruW w, const float* __restrict__ T, float* __restrict__ TG,
                                    float* __restrict__ TGN, float* __restrict__ XPT, int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)32 * HP4;
  ITEM_LOOP(total) {
    const int c = (int)(it / HP4), j = (int)(it % HP4) * 4;
    if (j < 32) {
      const int q = j >> 2;   // columns 4q .. 4q+3 of XPT row c
      float4 o;
      o.x = (c < C && (4 * q == c || 4 * q == C)) ? 1.f : 0.f;
      o.y = (c < C && (4 * q + 1 == c || 4 * q + 1 == C)) ? 1.f : 0.f;
      o.z = (c < C && (4 * q + 2 == c || 4 * q + 2 == C)) ? 1.f : 0.f;
      o.w = (c < C && (4 * q + 3 == c || 4 * q + 3 == C)) ? 1.f : 0.f;
      st4(XPT + c * 32 + j, o);
    }
    float4 drp = zero4(), dzp = zero4(), dnp = zero4(), dnr = zero4();
    if (c < C) {
      float4 gi[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) gi[g] = fma4(1.f, ld4(w.WihT + (size_t)c * 3 * HP + g * HP + j), ld4(w.bih + g * HP + j));
      const float4 ghr = ld4(w.bhh + j), ghz = ld4(w.bhh + HP + j), ghn = ld4(w.bhh + 2 * HP + j);
      const float4 dh = ld4(T + (size_t)c * HP + j);
#define TB_LANE(f)                                                          \
      {                                                                       \
        const float rr = sigm(gi[0].f + ghr.f), zz = sigm(gi[1].f + ghz.f);   \
        const float nn = tanhf(gi[2].f + rr * ghn.f);                         \
        dnp.f = dh.f * (1.f - zz) * (1.f - nn * nn);                          \
        dzp.f = dh.f * (0.f - nn) * zz * (1.f - zz);                          \
        drp.f = dnp.f * ghn.f * rr * (1.f - rr);                              \
        dnr.f = dnp.f * rr;                                                   \
      }
      TB_LANE(x) TB_LANE(y) TB_LANE(z) TB_LANE(w)
#undef TB_LANE
    }
    float* gh = TG + (size_t)c * 3 * HP;
    st4(gh + j, drp);
    st4(gh + HP + j, dzp);
    st4(gh + 2 * HP + j, dnr);
    st4(TGN + (size_t)c * HP + j, dnp);
  }
}

// gate/map epilogue, mod None: GMP holds h.[Wg;Wm][:, :H]^T; the position one-hot
// columns and the biases come in as posb[i] (dvae_model.py:71-79)
__global__ void gm_fwd_kernel(SchedDir S, int ns, int ne, float* __restrict__ GMP,
                              const float* __restrict__ posb, float* __restrict__ GM, int HP,
                              GemmPartials parts) {
  const int HP4 = HP >> 2;
  const long long total = (long long)(ne - ns) * HP4;
  ITEM_LOOP(total) {
    const int r = ns + (int)(it / HP4), j = (int)(it % HP4) * 4;
    const int i = S.row_node[r];
    float* gp = GMP + (size_t)r * 2 * HP;
    float4 pg, pm;
    if (parts.slab) {   // the product arrives as split-K slabs: add them up in slab order
      const float* sp = parts.slab + (size_t)(r - ns) * 2 * HP + j;
      pg = zero4(); pm = zero4();
      for (int k = 0; k < parts.splits; ++k) {
        pg = add4(pg, ld4(sp + (size_t)k * parts.stride));
        pm = add4(pm, ld4(sp + (size_t)k * parts.stride + HP));
      }
    } else {
      pg = ld4(gp + j);
      pm = ld4(gp + HP + j);
    }
    const float4 gpre = add4(pg, ld4(posb + (size_t)i * 2 * HP + j));
    const float4 m = add4(pm, ld4(posb + (size_t)i * 2 * HP + HP + j));
    const float4 g = make_float4(sigm(gpre.x), sigm(gpre.y), sigm(gpre.z), sigm(gpre.w));
    st4(gp + j, g);
    st4(gp + HP + j, m);
    st4(GM + (size_t)r * HP + j, make_float4(g.x * m.x, g.y * m.y, g.z * m.z, g.w * m.w));
  }
}

// mod 'bn': the gate/map input is the predecessor's node encoding only (:73-74)
__global__ void gm_bn_fwd_kernel(SchedDir S, int R, float* __restrict__ GMP, float* __restrict__ GM,
                                 const float* __restrict__ x, int N, int C, const float* __restrict__ WgmT,
                                 const float* __restrict__ bgm, int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)R * HP4;
  ITEM_LOOP(total) {
    const int r = (int)(it / HP4), j = (int)(it % HP4) * 4;
    const float* xr = x + ((size_t)S.row_graph[r] * N + S.row_node[r]) * C;
    float4 gpre = ld4(bgm + j), m = ld4(bgm + HP + j);
    for (int c = 0; c < C; ++c) {
      const float xv = xr[c];
      if (xv != 0.f) {
        gpre = fma4(xv, ld4(WgmT + (size_t)c * 2 * HP + j), gpre);
        m = fma4(xv, ld4(WgmT + (size_t)c * 2 * HP + HP + j), m);
      }
    }
    const float4 g = make_float4(sigm(gpre.x), sigm(gpre.y), sigm(gpre.z), sigm(gpre.w));
    float* gp = GMP + (size_t)r * 2 * HP;
    st4(gp + j, g);
    st4(gp + HP + j, m);
    st4(GM + (size_t)r * HP + j, make_float4(g.x * m.x, g.y * m.y, g.z * m.z, g.w * m.w));
  }
}

// dGM of a node = sum over its consumers of dAGG (transpose CSR gather), then
// through g*m:  GMP <- (dGM*m*g*(1-g), dGM*g)
__global__ void dgm_bwd_kernel(SchedDir S, int ns, int ne, int prefix, const float* __restrict__ DAGG,
                               float* __restrict__ GMP, int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)(ne - ns) * HP4;
  ITEM_LOOP(total) {
    const int r = ns + (int)(it / HP4), j = (int)(it % HP4) * 4;
    float4 d = zero4();
    const int e0 = S.csc_ptr[r], e1 = S.csc_ptr[r + 1];
    for (int e = e0; e < e1; ++e) {
      if (prefix) {
        const size_t pr = S.csc_prow[e];
        const int cnt = S.csc_cnt[e];
        for (int t = 0; t < cnt; ++t) d = add4(d, ld4(DAGG + (pr + t) * HP + j));
      } else {
        d = add4(d, ld4(DAGG + (size_t)S.csc_idx[e] * HP + j));
      }
    }
    float* gp = GMP + (size_t)r * 2 * HP;
    const float4 g = ld4(gp + j), m = ld4(gp + HP + j);
    st4(gp + j, make_float4(d.x * m.x * g.x * (1.f - g.x), d.y * m.y * g.y * (1.f - g.y),
                            d.z * m.z * g.z * (1.f - g.z), d.w * m.w * g.w * (1.f - g.w)));
    st4(gp + HP + j, make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w));
  }
}

// The same for rows with LONG successor lists -- shared encoder rows of the first levels (sched.h
// SchedView::enc: a value row of level 0 feeds thousands of rows): SL lanes walk a row's list side by side
// (lane s takes links s, s + SL, ...), their partial sums are folded through LDS in lane order.  One block =
// one row x 256/SL float4 columns.
template <int SL>
__global__ __launch_bounds__(256) void dgm_bwd_wide_kernel(SchedDir S, int ns, int ne, const float* __restrict__ DAGG,
                                                           float* __restrict__ GMP, int HP) {
  constexpr int CL = 256 / SL;
  __shared__ float4 sh[SL][CL];
  const int cl = threadIdx.x % CL, sl = threadIdx.x / CL;
  const int chunks = HP / (4 * CL);
  const int r = ns + blockIdx.x / chunks, j = ((blockIdx.x % chunks) * CL + cl) * 4;
  float4 d = zero4();
  const int e1 = S.csc_ptr[r + 1];
  for (int e = S.csc_ptr[r] + sl; e < e1; e += SL) d = add4(d, ld4(DAGG + (size_t)S.csc_idx[e] * HP + j));
  sh[sl][cl] = d;
  __syncthreads();
  if (sl != 0) return;
  for (int k = 1; k < SL; ++k) d = add4(d, sh[k][cl]);
  float* gp = GMP + (size_t)r * 2 * HP;
  const float4 g = ld4(gp + j), m = ld4(gp + HP + j);
  st4(gp + j, make_float4(d.x * m.x * g.x * (1.f - g.x), d.y * m.y * g.y * (1.f - g.y),
                          d.z * m.z * g.z * (1.f - g.z), d.w * m.w * g.w * (1.f - g.w)));
  st4(gp + HP + j, make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w));
}

// Prefix sweeps with one hidden product per node: for node u, over every consumer prefix row that
// holds u, DGM[u] = sum of the direct gradients (DH after gru_bwd) and PG[u] = sum of dgh (GH after
// gru_bwd).  dL/dGM[u] = DGM[u] + PG[u].Whh follows by a GEMM, then gm_bwd_apply.
__global__ void dgm_gather_kernel(SchedDir S, int ns, int ne, const float* __restrict__ DAGG,
                                  const float* __restrict__ GH, float* __restrict__ DGM, float* __restrict__ PG,
                                  int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)(ne - ns) * HP4;
  ITEM_LOOP(total) {
    const int r = ns + (int)(it / HP4), j = (int)(it % HP4) * 4;
    float4 d = zero4(), g0 = zero4(), g1 = zero4(), g2 = zero4();
    const int e0 = S.csc_ptr[r], e1 = S.csc_ptr[r + 1];
    for (int e = e0; e < e1; ++e) {
      const size_t pr = S.csc_prow[e];
      const int cnt = S.csc_cnt[e];
      for (int t = 0; t < cnt; ++t) {
        d = add4(d, ld4(DAGG + (pr + t) * HP + j));
        const float* gh = GH + (pr + t) * 3 * HP + j;
        g0 = add4(g0, ld4(gh));
        g1 = add4(g1, ld4(gh + HP));
        g2 = add4(g2, ld4(gh + 2 * HP));
      }
    }
    st4(DGM + (size_t)r * HP + j, d);
    float* pg = PG + (size_t)r * 3 * HP + j;
    st4(pg, g0);
    st4(pg + HP, g1);
    st4(pg + 2 * HP, g2);
  }
}

__global__ void gm_bwd_apply_kernel(int ns, int ne, const float* __restrict__ DGM, float* __restrict__ GMP, int HP,
                                    GemmPartials parts) {
  const int HP4 = HP >> 2;
  const long long total = (long long)(ne - ns) * HP4;
  ITEM_LOOP(total) {
    const int r = ns + (int)(it / HP4), j = (int)(it % HP4) * 4;
    float4 d = ld4(DGM + (size_t)r * HP + j);
    if (parts.slab) {   // the hidden part arrives as split-K slabs: (their sum, in slab order) + the direct part
      const float* sp = parts.slab + (size_t)(r - ns) * HP + j;
      float4 s = zero4();
      for (int k = 0; k < parts.splits; ++k) s = add4(s, ld4(sp + (size_t)k * parts.stride));
      d = add4(s, d);
    }
    float* gp = GMP + (size_t)r * 2 * HP;
    const float4 g = ld4(gp + j), m = ld4(gp + HP + j);
    st4(gp + j, make_float4(d.x * m.x * g.x * (1.f - g.x), d.y * m.y * g.y * (1.f - g.y),
                            d.z * m.z * g.z * (1.f - g.z), d.w * m.w * g.w * (1.f - g.w)));
    st4(gp + HP + j, make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w));
  }
}

// GRUCell backward epilogue: GH (r,z,n) -> (dr_pre, dz_pre, dn_pre*r) = d(gh);
// GHN -> dn_pre (so d(gi) = (GH_r, GH_z, GHN)); DH -> dh*z (direct path to h_in)
__global__ void gru_bwd_kernel(SchedDir S, int R, int p0, int p1, int prefix, float* __restrict__ GH,
                               float* __restrict__ GHN, const float* __restrict__ AGG,
                               float* __restrict__ DH, const float* __restrict__ DHn, int HP,
                               GemmPartials parts, int parts_row0, const int* __restrict__ apos) {
  const int HP4 = HP >> 2;
  const long long total = (long long)(p1 - p0) * HP4;
  ITEM_LOOP(total) {
    const int p = p0 + (int)(it / HP4), j = (int)(it % HP4) * 4;
    // apos (prefix sweeps): an unscored edge row has no gradient of its own, nothing was written there
    float4 dh = (!apos || p < R || apos[p - R + 1] != apos[p - R]) ? ld4(DH + (size_t)p * HP + j) : zero4();
    // parts: the gate/map input gradient of node rows [parts_row0, ..) still as split-K slabs; it belongs to
    // the table the product accumulates into (DH of the node row, or DHn for prefix sweeps) and is added
    // first, as the combine launch would have
    if (prefix && DHn) {
      const int r = S.prow_node[p];
      if (p == final_prow(S, R, r)) {
        float4 t = ld4(DHn + (size_t)r * HP + j);
        if (parts.slab) {
          const float* sp = parts.slab + (size_t)(r - parts_row0) * HP + j;
          float4 s = zero4();
          for (int k = 0; k < parts.splits; ++k) s = add4(s, ld4(sp + (size_t)k * parts.stride));
          t = add4(s, t);
        }
        dh = add4(dh, t);
      }
    } else if (!prefix && parts.slab) {
      const float* sp = parts.slab + (size_t)(p - parts_row0) * HP + j;
      float4 s = zero4();
      for (int k = 0; k < parts.splits; ++k) s = add4(s, ld4(sp + (size_t)k * parts.stride));
      dh = add4(s, dh);
    }
    float* gh = GH + (size_t)p * 3 * HP;
    const float4 rr = ld4(gh + j), zz = ld4(gh + HP + j), nn = ld4(gh + 2 * HP + j);
    const float4 ghn = ld4(GHN + (size_t)p * HP + j);
    const float4 a = ld4(AGG + (size_t)p * HP + j);
    float4 drp, dzp, dnp, dnr, da;
#define BWD_LANE(f)                                              \
    dnp.f = dh.f * (1.f - zz.f) * (1.f - nn.f * nn.f);            \
    dzp.f = dh.f * (a.f - nn.f) * zz.f * (1.f - zz.f);            \
    drp.f = dnp.f * ghn.f * rr.f * (1.f - rr.f);                  \
    dnr.f = dnp.f * rr.f;                                         \
    da.f = dh.f * zz.f;
    BWD_LANE(x) BWD_LANE(y) BWD_LANE(z) BWD_LANE(w)
#undef BWD_LANE
    st4(gh + j, drp);
    st4(gh + HP + j, dzp);
    st4(gh + 2 * HP + j, dnr);
    st4(GHN + (size_t)p * HP + j, dnp);
    st4(DH + (size_t)p * HP + j, da);
  }
}

// ---------------------------------------------------------------------------
// deterministic column reductions for the skinny weight gradients
// ---------------------------------------------------------------------------
// out[k] = sum_c part[c*stride + k]; 64 columns x 4 interleaved chunk groups per
// block, combined in a fixed order
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ part, int nparts,
                                                           size_t stride, float* __restrict__ out, size_t len) {
  __shared__ float sh[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (size_t base = (size_t)blockIdx.x * 64; base < len; base += (size_t)gridDim.x * 64) {
    const size_t k = base + col;
    float s = 0.f;
    if (k < len)
      for (int c = grp; c < nparts; c += 4) s += part[c * stride + k];
    sh[grp][col] = s;
    __syncthreads();
    if (grp == 0 && k < len) out[k] = (sh[0][col] + sh[1][col]) + (sh[2][col] + sh[3][col]);
    __syncthreads();
  }
}

// tmp[g*len + k] = sum of partials [g*fold, (g+1)*fold) at k; same 4-way interleave as above
__global__ __launch_bounds__(256) void reduce_groups_kernel(const float* __restrict__ part, int nparts, size_t stride,
                                                            int fold, float* __restrict__ tmp, size_t len) {
  __shared__ float sh[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const size_t k = (size_t)blockIdx.x * 64 + col;
  const int c0 = blockIdx.y * fold, c1 = min(nparts, c0 + fold);
  float s = 0.f;
  if (k < len)
    for (int c = c0 + grp; c < c1; c += 4) s += part[c * stride + k];
  sh[grp][col] = s;
  __syncthreads();
  if (grp == 0 && k < len) tmp[(size_t)blockIdx.y * len + k] = (sh[0][col] + sh[1][col]) + (sh[2][col] + sh[3][col]);
}

int reduce_parts(const float* part, int nparts, size_t stride, float* out, size_t len, hipStream_t st) {
  const int grid = (int)std::min<size_t>((len + 63) / 64, 4096);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(grid), dim3(256), 0, st, part, nparts, stride, out, len);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

// Same sum in two fixed-order stages for a SHORT vector with MANY partials (pair_bwd_a's
// dw2: 2052 floats x 2048 waves): stage 1 folds groups of `fold` consecutive partials into
// the head of the partial buffer's own tail-free scratch `tmp`, stage 2 sums the groups.
int reduce_parts_2stage(const float* part, int nparts, size_t stride, float* tmp, size_t tmp_floats, float* out,
                        size_t len, hipStream_t st) {
  const int fold = 32;
  const int groups = cdiv(nparts, fold);
  if (nparts <= 64 || (size_t)groups * len > tmp_floats) return reduce_parts(part, nparts, stride, out, len, st);
  const int gx = (int)((len + 63) / 64);
  hipLaunchKernelGGL(reduce_groups_kernel, dim3(gx, groups), dim3(256), 0, st, part, nparts, stride, fold, tmp, len);
  DVAE_LAUNCH_CHECK();
  return reduce_parts(tmp, groups, len, out, len, st);
}

// one-hot side tables for the skinny weight gradients (TN GEMMs):
//   XP [rows][32]: node encoding of the row's node in cols [0,C), 1 in col C
//   PX [R][PP]   : one-hot of the node's position in cols [0,npos), 1 in col npos
__global__ void build_xp_kernel(SchedDir S, int prefix, int rows, const float* __restrict__ x, int N, int C,
                                float* __restrict__ XP) {
  const long long total = (long long)rows * 32;
  ITEM_LOOP(total) {
    const int p = (int)(it >> 5), c = (int)(it & 31);
    const int r = prefix ? S.prow_node[p] : p;
    float v = 0.f;
    if (c < C) v = x[((size_t)S.row_graph[r] * N + S.row_node[r]) * C + c];
    else if (c == C) v = 1.f;
    XP[it] = v;
  }
}

__global__ void build_xt_kernel(SchedDir S, int prefix, int rows, const float* __restrict__ x, int N, int C,
                                int* __restrict__ XT) {
  ITEM_LOOP(rows) {
    const int r = prefix ? S.prow_node[it] : (int)it;
    const float* xr = x + ((size_t)S.row_graph[r] * N + S.row_node[r]) * C;
    int t = 0;
    for (int c = 0; c < C; ++c)
      if (xr[c] != 0.f) t = c;
    XT[it] = t;
  }
}

__global__ void build_px_kernel(SchedDir S, int R, int npos, int PP, float* __restrict__ PX) {
  const long long total = (long long)R * PP;
  ITEM_LOOP(total) {
    const int r = (int)(it / PP), c = (int)(it % PP);
    PX[it] = (c == S.row_node[r] || c == npos) ? 1.f : 0.f;
  }
}

// part[chunk][w] = sum over the chunk's rows of X[row][w]: a block owns 64 columns x one
// row chunk; 16 row lanes x 16 float4 columns, rows strided by 16, lanes folded in fixed order.
// MASK: X *= (Y > 0) first, written back (ReLU backward and the bias gradient in one pass).
template <bool MASK>
__global__ __launch_bounds__(256) void colsum_kernel(float* __restrict__ X, const float* __restrict__ Y, int ld,
                                                     int rows, int rpc, int W, float* __restrict__ part) {
  __shared__ float4 sh[16][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + cx * 4;
  const int r0 = blockIdx.y * rpc, r1 = min(rows, r0 + rpc);
  float4 s = zero4();
  if (col < W)
    for (int r = r0 + ry; r < r1; r += 16) {
      float4 v = ld4(X + (size_t)r * ld + col);
      if (MASK) {
        const float4 y = ld4(Y + (size_t)r * ld + col);
        v.x = y.x > 0.f ? v.x : 0.f;
        v.y = y.y > 0.f ? v.y : 0.f;
        v.z = y.z > 0.f ? v.z : 0.f;
        v.w = y.w > 0.f ? v.w : 0.f;
        st4(X + (size_t)r * ld + col, v);
      }
      s = add4(s, v);
    }
  sh[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < W) {
    float4 t = sh[0][cx];
#pragma unroll
    for (int k = 1; k < 16; ++k) t = add4(t, sh[k][cx]);
    st4(part + (size_t)blockIdx.y * W + col, t);
  }
}

// Column sums per node TYPE: part[chunk][t][w] = sum over the chunk's rows r with row_type[r] == t of X[r][w].
// This is the skinny product X^T . onehot(type) (the decoder's per-type gradient sums DTT = DA^T . XP, sched.h TypeShare;
// the input-weight and bias gradients of the GRU cells; the position columns of the gate/map gradients) done as what it
// is -- one streaming pass over X -- instead of a [W x 32] GEMM tile grid that walks X in 16 KB steps.  A WAVE takes whole
// rows, 1 KB (256 columns) at a time, four rows in flight; the row's type is uniform over the wave, so the accumulator of
// each type is a REGISTER float4 picked by a scalar branch -- no LDS, no atomics.  A block = four waves on rows r, r+1,
// r+2, r+3 (mod 4) of one row chunk x one group of 256 columns; the waves' sums meet in LDS in wave order.  Rows whose type
// is outside [0, C) must not occur (the callers check info[5]).  CT = register accumulators (8 / 16 / 32 >= C).
template <int CT>
__device__ __forceinline__ void typed_add(float4 (&acc)[CT], int t, const float4& v) {
#pragma unroll
  for (int k = 0; k < CT; ++k)
    if (t == k) acc[k] = add4(acc[k], v);
}

template <int CT>
__global__ __launch_bounds__(256) void typed_colsum_kernel(const float* __restrict__ X, int ld, int rows, int rpc, int W,
                                                           const int* __restrict__ row_type, int C, float* __restrict__ part) {
  __shared__ float4 sh[3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + lane * 4;
  const bool on = col < W;
  const int r0 = blockIdx.y * rpc, r1 = min(rows, r0 + rpc);
  float4 acc[CT], tot = zero4();   // tot: all rows (the column of ones)
#pragma unroll
  for (int k = 0; k < CT; ++k) acc[k] = zero4();
  const float* xc = X + (on ? col : 0);
  int r = r0 + wave;
  for (; r + 12 < r1; r += 16) {
    const int t0 = __builtin_amdgcn_readfirstlane(row_type[r]), t1 = __builtin_amdgcn_readfirstlane(row_type[r + 4]);
    const int t2 = __builtin_amdgcn_readfirstlane(row_type[r + 8]), t3 = __builtin_amdgcn_readfirstlane(row_type[r + 12]);
    float4 v0 = zero4(), v1 = zero4(), v2 = zero4(), v3 = zero4();
    if (on) {
      v0 = ld4(xc + (size_t)r * ld);
      v1 = ld4(xc + (size_t)(r + 4) * ld);
      v2 = ld4(xc + (size_t)(r + 8) * ld);
      v3 = ld4(xc + (size_t)(r + 12) * ld);
    }
    typed_add<CT>(acc, t0, v0);
    typed_add<CT>(acc, t1, v1);
    typed_add<CT>(acc, t2, v2);
    typed_add<CT>(acc, t3, v3);
    tot = add4(add4(tot, v0), add4(add4(v1, v2), v3));
  }
  for (; r < r1; r += 4) {
    const int t0 = __builtin_amdgcn_readfirstlane(row_type[r]);
    const float4 v0 = on ? ld4(xc + (size_t)r * ld) : zero4();
    typed_add<CT>(acc, t0, v0);
    tot = add4(tot, v0);
  }
  // part[chunk][t][w], t = C: the total
  float* pc = part + (size_t)blockIdx.y * (C + 1) * W + col;
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    if (t < C) {   // uniform
      if (wave) sh[wave - 1][lane] = acc[t];
      __syncthreads();
      if (wave == 0 && on) st4(pc + (size_t)t * W, add4(add4(acc[t], sh[0][lane]), add4(sh[1][lane], sh[2][lane])));
      __syncthreads();
    }
  }
  if (wave) sh[wave - 1][lane] = tot;
  __syncthreads();
  if (wave == 0 && on) st4(pc + (size_t)C * W, add4(add4(tot, sh[0][lane]), add4(sh[1][lane], sh[2][lane])));
}

// out[w][c] (+)= sum over chunks of part[chunk][c][w] for c <= C (row C of a chunk: all rows, the column of ones of the
// one-hot table); the other columns of the ldo are zero (set, not added: a launch that accumulates has no blocks for them).
// Block (x, c): 16 float4 of w for output column c, sixteen lanes per sum over the chunks (four loads in flight each),
// folded in lane order.
__global__ __launch_bounds__(256) void typed_colsum_final_kernel(const float* __restrict__ part, int nch, int C, int W,
                                                                 float* __restrict__ out, int ldo, int accumulate) {
  __shared__ float4 sh[16][16];
  const int wq = threadIdx.x & 15, kl = threadIdx.x >> 4;
  const int c = blockIdx.y, w = (blockIdx.x * 16 + wq) * 4;
  const bool on = w < W;
  float4 s = zero4();
  if (on && c <= C) {
    const size_t kstep = (size_t)(C + 1) * W;
    const float* q = part + (size_t)c * W + w;
    int k = kl;
    for (; k + 48 < nch; k += 64) {
      const float4 a0 = ld4(q + k * kstep), a1 = ld4(q + (k + 16) * kstep), a2 = ld4(q + (k + 32) * kstep),
                   a3 = ld4(q + (k + 48) * kstep);
      s = add4(add4(s, a0), add4(add4(a1, a2), a3));
    }
    for (; k < nch; k += 16) s = add4(s, ld4(q + k * kstep));
  }
  sh[kl][wq] = s;
  __syncthreads();
  if (kl == 0 && on) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s = add4(s, sh[k][wq]);
    float* o = out + (size_t)w * ldo + c;
    if (accumulate && c <= C) {
      o[0] += s.x; o[ldo] += s.y; o[2 * (size_t)ldo] += s.z; o[3 * (size_t)ldo] += s.w;
    } else if (!accumulate) {
      o[0] = s.x; o[ldo] = s.y; o[2 * (size_t)ldo] = s.z; o[3 * (size_t)ldo] = s.w;
    }
  }
}

// ---------------------------------------------------------------------------
// glue
// ---------------------------------------------------------------------------
__global__ void rows_to_dense_kernel(SchedDir S, int R, const float* __restrict__ T, int ld,
                                     float* __restrict__ out, int N, int W) {
  const long long total = (long long)R * W;
  ITEM_LOOP(total) {
    const int r = (int)(it / W), c = (int)(it % W);
    out[((size_t)S.row_graph[r] * N + S.row_node[r]) * W + c] = T[(size_t)r * ld + c];
  }
}

__global__ void nodes_to_dense_kernel(SchedDir S, int B, int N, const float* __restrict__ T, int ld,
                                      float* __restrict__ out, int W) {
  const long long total = (long long)B * N * W;
  ITEM_LOOP(total) {
    const int id = (int)(it / W), c = (int)(it % W);
    const int r = S.node2row[id];
    out[it] = r >= 0 ? T[(size_t)r * ld + c] : 0.f;
  }
}

__global__ void dense_to_rows_kernel(SchedDir S, int R, const float* __restrict__ in, int N, int W,
                                     float* __restrict__ T, int ld) {
  const long long total = (long long)R * ld;
  ITEM_LOOP(total) {
    const int r = (int)(it / ld), c = (int)(it % ld);
    T[(size_t)r * ld + c] = c < W ? in[((size_t)S.row_graph[r] * N + S.row_node[r]) * W + c] : 0.f;
  }
}

__device__ __forceinline__ int graph_size(const int* sizes, int b, int N) { return sizes ? sizes[b] : N; }

__global__ void take_last_kernel(SchedDir S, const int* __restrict__ sizes, int B, int N,
                                 const float* __restrict__ src, float* __restrict__ dst, int HP, int bwd) {
  const int HP4 = HP >> 2;
  const long long total = (long long)B * HP4;
  ITEM_LOOP(total) {
    const int b = (int)(it / HP4), j = (int)(it % HP4) * 4;
    const int n = graph_size(sizes, b, N);
    if (n <= 0) {
      if (!bwd) st4(dst + (size_t)b * HP + j, zero4());
      continue;
    }
    const int row = S.node2row[b * N + n - 1];
    if (!bwd) st4(dst + (size_t)b * HP + j, ld4(src + (size_t)row * HP + j));
    else st4(dst + (size_t)row * HP + j, add4(ld4(dst + (size_t)row * HP + j), ld4(src + (size_t)b * HP + j)));
  }
}

// graph_state of node i: h0 (i=0) | state of node i-1 (None, :257) | sum of states < i (bn, :251-255)
__global__ void graph_state_fwd_kernel(SchedDir S, int R, int N, int mod, const float* __restrict__ Hn,
                                       const float* __restrict__ h0, float* __restrict__ GS, int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)R * HP4;
  ITEM_LOOP(total) {
    const int r = (int)(it / HP4), j = (int)(it % HP4) * 4;
    const int b = S.row_graph[r], i = S.row_node[r];
    float4 v;
    if (i == 0) v = ld4(h0 + (size_t)b * HP + j);
    else if (mod == 0) v = ld4(Hn + (size_t)S.node2row[b * N + i - 1] * HP + j);
    else {
      v = zero4();
      for (int u = 0; u < i; ++u) v = add4(v, ld4(Hn + (size_t)S.node2row[b * N + u] * HP + j));
    }
    st4(GS + (size_t)r * HP + j, v);
  }
}

// DHn[(b,u)] = dGS of the consumers of node u's state; dh0[b] = dGS[(b,0)]
__global__ void graph_state_bwd_kernel(SchedDir S, int R, int N, int mod, const int* __restrict__ sizes,
                                       const float* __restrict__ DGS, float* __restrict__ DHn,
                                       float* __restrict__ dh0, int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)R * HP4;
  ITEM_LOOP(total) {
    const int r = (int)(it / HP4), j = (int)(it % HP4) * 4;
    const int b = S.row_graph[r], u = S.row_node[r];
    const int n = graph_size(sizes, b, N);
    float4 v = zero4();
    if (mod == 0) {
      if (u + 1 < n) v = ld4(DGS + (size_t)S.node2row[b * N + u + 1] * HP + j);
    } else {
      for (int i = u + 1; i < n; ++i) v = add4(v, ld4(DGS + (size_t)S.node2row[b * N + i] * HP + j));
    }
    st4(DHn + (size_t)r * HP + j, v);
    if (u == 0) st4(dh0 + (size_t)b * HP + j, ld4(DGS + (size_t)r * HP + j));
  }
}

__global__ void graph_sum_kernel(SchedDir S, const int* __restrict__ sizes, int B, int N,
                                 const float* __restrict__ T, int W, float* __restrict__ out) {
  const int W4 = W >> 2;
  const long long total = (long long)B * W4;
  ITEM_LOOP(total) {
    const int b = (int)(it / W4), j = (int)(it % W4) * 4;
    const int n = graph_size(sizes, b, N);
    float4 v = zero4();
    for (int i = 0; i < n; ++i) v = add4(v, ld4(T + (size_t)S.node2row[b * N + i] * W + j));
    st4(out + (size_t)b * W + j, v);
  }
}

__global__ void add_dh0_kernel(SchedDir S, int B, int N, const float* __restrict__ DAGG,
                               float* __restrict__ dh0, int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)B * HP4;
  ITEM_LOOP(total) {
    const int b = (int)(it / HP4), j = (int)(it % HP4) * 4;
    const int r = S.node2row[b * N];
    if (r < 0) continue;
    st4(dh0 + (size_t)b * HP + j, add4(ld4(dh0 + (size_t)b * HP + j), ld4(DAGG + (size_t)r * HP + j)));
  }
}

__global__ void pad_rows_kernel(const float* __restrict__ in, int rows, int W, float* __restrict__ out, int ld,
                                int ncols) {
  const long long total = (long long)rows * ncols;
  ITEM_LOOP(total) {
    const int r = (int)(it / ncols), c = (int)(it % ncols);
    out[(size_t)r * ld + c] = (in && c < W) ? in[(size_t)r * W + c] : 0.f;
  }
}

// two tables side by side in one padded row: out[r][0:ncols] <- in0[r][0:W] | 0, out[r][ncols:2 ncols] <- in1[r][0:W] | 0
__global__ void pad_rows2_kernel(const float* __restrict__ in0, const float* __restrict__ in1, int rows, int W,
                                 float* __restrict__ out, int ld, int ncols) {
  const long long total = (long long)rows * 2 * ncols;
  ITEM_LOOP(total) {
    const int r = (int)(it / (2 * ncols)), cc = (int)(it % (2 * ncols));
    const int h = cc >= ncols, c = cc - h * ncols;
    const float* in = h ? in1 : in0;
    out[(size_t)r * ld + cc] = (in && c < W) ? in[(size_t)r * W + c] : 0.f;   // a missing table reads as zeros
  }
}
// and back: out0[r][0:W] <- in[r][0:W], out1[r][0:W] <- in[r][off:off+W]
__global__ void unpad_rows2_kernel(const float* __restrict__ in, int ld, int off, int rows, int W, float* __restrict__ out0,
                                   float* __restrict__ out1) {
  const long long total = (long long)rows * 2 * W;
  ITEM_LOOP(total) {
    const int h = it >= (long long)rows * W;
    const long long k = it - (h ? (long long)rows * W : 0);
    const int r = (int)(k / W), c = (int)(k % W);
    (h ? out1 : out0)[k] = in[(size_t)r * ld + h * off + c];
  }
}

__global__ void relu_bwd_kernel(float* __restrict__ d, const float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (!(y[i] > 0.f)) d[i] = 0.f;
}

__global__ void zero_kernel(float4* __restrict__ p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// X[rows[i]][0..W) = 0 (W a multiple of 4)
__global__ void zero_rows_kernel(float* __restrict__ X, int ld, int W, const int* __restrict__ rows, int n) {
  const int wq = W >> 2;
  const long long total = (long long)n * wq;
  ITEM_LOOP(total) {
    const int i = (int)(it / wq), q = (int)(it % wq);
    st4(X + (size_t)rows[i] * ld + q * 4, zero4());
  }
}

__global__ void unpad_rows_kernel(const float* __restrict__ in, int ld, int rows, int W, float* __restrict__ out) {
  const long long total = (long long)rows * W;
  ITEM_LOOP(total) {
    const int r = (int)(it / W), c = (int)(it % W);
    out[it] = in[(size_t)r * ld + c];
  }
}

__global__ void add_rows_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds_,
                                int rows, int W) {
  const long long total = (long long)rows * W;
  ITEM_LOOP(total) {
    const int r = (int)(it / W), c = (int)(it % W);
    dst[(size_t)r * ldd + c] += src[(size_t)r * lds_ + c];
  }
}

// ---------------------------------------------------------------------------
// add_edge pair scorer (dvae_model.py:211-219):
//   logit(i,j) = w2 . relu(A[prow(i,k)] + Bq[j] (+ Cq[b])) + b2,
//   k = number of predecessors of i with index above j.  One wave per row.
// ---------------------------------------------------------------------------
// all pairs (i, j < i) of node row r, by one wave
template <int VPL>
__device__ __forceinline__ void pair_fwd_row(const SchedView& sv, const TypeShare& ts, int R, int r, int lane,
                                             const float4 (&wv)[VPL], float bias2, const float* __restrict__ A,
                                             const float* __restrict__ Bq, const float* __restrict__ Cq,
                                             float* __restrict__ edge_logits) {
  constexpr int EHP = VPL * 256;
  const SchedDir& S = sv.dir[0];
  const int N = sv.N;
  const int i = S.row_node[r];
  if (i == 0) return;
  const int b = S.row_graph[r];
  const int p = S.csr_ptr[r], deg = S.csr_ptr[r + 1] - p;
  int t = 0, cur = -1;
  int next_pred = deg > 0 ? S.row_node[S.csr_idx[p]] : -1;
  float4 a[VPL], c[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) c[v] = Cq ? ld4(Cq + (size_t)b * EHP + (lane + 64 * v) * 4) : zero4();
  for (int j = i - 1; j >= 0; --j) {
    while (t < deg && next_pred > j) {
      ++t;
      next_pred = t < deg ? S.row_node[S.csr_idx[p + t]] : -1;
    }
    if (t != cur) {
      const float* ar = a_row(sv, ts, A, EHP, R, r, t);
#pragma unroll
      for (int v = 0; v < VPL; ++v)
        a[v] = add4(ld4(ar + (lane + 64 * v) * 4), c[v]);
      cur = t;
    }
    const size_t rj = S.node2row[b * N + j];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const float4 q = add4(a[v], ld4(Bq + rj * EHP + (lane + 64 * v) * 4));
      s = fmaf(wv[v].x, fmaxf(q.x, 0.f), s);
      s = fmaf(wv[v].y, fmaxf(q.y, 0.f), s);
      s = fmaf(wv[v].z, fmaxf(q.z, 0.f), s);
      s = fmaf(wv[v].w, fmaxf(q.w, 0.f), s);
    }
    s = wave_sum(s);
    if (lane == 0) edge_logits[((size_t)b * N + i) * N + j] = s + bias2;
  }
}

// One workgroup per graph (grid-stride); its four waves share the graph's node rows so that the Bq
// rows, which every later node of the graph reads again, stay in this CU's L1 / this XCD's L2.
// Wave w takes the node pairs {n-1-k, k}, k = w, w+4, ..: n-1 pair evaluations each, balanced.
template <int VPL>
__global__ __launch_bounds__(256) void pair_fwd_kernel(SchedView sv, TypeShare ts, int R, const int* __restrict__ sizes,
                                                       const float* __restrict__ A,
                                                       const float* __restrict__ Bq,
                                                       const float* __restrict__ Cq,
                                                       const float* __restrict__ w2,
                                                       const float* __restrict__ b2,
                                                       float* __restrict__ edge_logits) {
  const SchedDir& S = sv.dir[0];
  const int N = sv.N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 wv[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) wv[v] = ld4(w2 + (lane + 64 * v) * 4);
  const float bias2 = b2[0];
  for (int b = blockIdx.x; b < sv.B; b += gridDim.x) {
    const int n = graph_size(sizes, b, N);
    for (int k = wave; 2 * k <= n - 1; k += 4) {
      const int hi = n - 1 - k;
      pair_fwd_row<VPL>(sv, ts, R, S.node2row[b * N + hi], lane, wv, bias2, A, Bq, Cq, edge_logits);
      if (k != hi && k > 0)
        pair_fwd_row<VPL>(sv, ts, R, S.node2row[b * N + k], lane, wv, bias2, A, Bq, Cq, edge_logits);
    }
  }
}

// The pair scorer for LARGER graphs (N > 16): pair_fwd_kernel reads an 8-12 KB Bq row per PAIR, and a 64-node graph's
// Bq rows (512 KB) fit no L1 and, with ~100 graphs in flight per XCD, no L2 either (config 3: 53 GB per step, the
// algorithmic 13).  Here a 1024-thread workgroup takes one graph and walks the add_edge width in slices of 512 columns:
// the slice of the Bq rows of a window of CW candidate nodes is staged in LDS (CW x 2 KB), every wave takes node rows of
// the graph and scores their pairs against it (A's slice of a prefix row is read once per slice, Bq comes from LDS), and
// the partial logits of the slices meet in an LDS table (row i, candidate j: always the same wave, the same lane -- no
// atomics, a fixed order), written out once per window.
constexpr int PF_SL = 512;   // columns per slice
__global__ __launch_bounds__(1024) void pair_fwd_slice_kernel(SchedView sv, TypeShare ts, int R, int EHP, int CW,
                                                              const int* __restrict__ sizes,
                                                              const float* __restrict__ A,
                                                              const float* __restrict__ Bq,
                                                              const float* __restrict__ Cq,
                                                              const float* __restrict__ w2,
                                                              const float* __restrict__ b2,
                                                              float* __restrict__ edge_logits) {
  extern __shared__ __attribute__((aligned(16))) float pf_lds[];
  float* bql = pf_lds;                         // [CW][PF_SL]
  float* lg = pf_lds + (size_t)CW * PF_SL;     // [N][CW] partial logits of (row i, candidate jb + jj)
  const SchedDir& S = sv.dir[0];
  const int N = sv.N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int ns = EHP / PF_SL;
  const float bias2 = b2[0];
  for (int b = blockIdx.x; b < sv.B; b += gridDim.x) {
    const int n = graph_size(sizes, b, N);
    for (int jb = 0; jb + 1 < n; jb += CW) {
      const int cw = min(CW, n - jb);          // candidates jb .. jb + cw - 1 (the last node is never one: harmless)
      for (int sl = 0; sl < ns; ++sl) {
        const int c0 = sl * PF_SL;
        __syncthreads();                       // the previous slice / window is done with bql (and lg, when sl == 0)
        for (int q = threadIdx.x; q < cw * (PF_SL / 4); q += blockDim.x) {
          const int jj = q / (PF_SL / 4), cq = (q % (PF_SL / 4)) * 4;
          st4(bql + jj * PF_SL + cq, ld4(Bq + (size_t)S.node2row[b * N + jb + jj] * EHP + c0 + cq));
        }
        __syncthreads();
        const float4 w0 = ld4(w2 + c0 + lane * 4), w1 = ld4(w2 + c0 + 256 + lane * 4);
        const float4 cc0 = Cq ? ld4(Cq + (size_t)b * EHP + c0 + lane * 4) : zero4();
        const float4 cc1 = Cq ? ld4(Cq + (size_t)b * EHP + c0 + 256 + lane * 4) : zero4();
        for (int i = jb + 1 + wave; i < n; i += nwaves) {
          const int r = S.node2row[b * N + i];
          const int p = S.csr_ptr[r], deg = S.csr_ptr[r + 1] - p;
          const int jtop = min(jb + cw, i) - 1;   // the highest candidate of this window that node i is paired with
          int t = 0;
          while (t < deg && S.row_node[S.csr_idx[p + t]] > jtop) ++t;
          int next_pred = t < deg ? S.row_node[S.csr_idx[p + t]] : -1;
          int cur = -1;
          float4 a0 = zero4(), a1 = zero4();
          for (int j = jtop; j >= jb; --j) {
            while (t < deg && next_pred > j) {
              ++t;
              next_pred = t < deg ? S.row_node[S.csr_idx[p + t]] : -1;
            }
            if (t != cur) {
              const float* ar = a_row(sv, ts, A, EHP, R, r, t) + c0;
              a0 = add4(ld4(ar + lane * 4), cc0);
              a1 = add4(ld4(ar + 256 + lane * 4), cc1);
              cur = t;
            }
            const float* bj = bql + (j - jb) * PF_SL;
            const float4 q0 = add4(a0, ld4(bj + lane * 4)), q1 = add4(a1, ld4(bj + 256 + lane * 4));
            float sum = 0.f;
            sum = fmaf(w0.x, fmaxf(q0.x, 0.f), sum);
            sum = fmaf(w0.y, fmaxf(q0.y, 0.f), sum);
            sum = fmaf(w0.z, fmaxf(q0.z, 0.f), sum);
            sum = fmaf(w0.w, fmaxf(q0.w, 0.f), sum);
            sum = fmaf(w1.x, fmaxf(q1.x, 0.f), sum);
            sum = fmaf(w1.y, fmaxf(q1.y, 0.f), sum);
            sum = fmaf(w1.z, fmaxf(q1.z, 0.f), sum);
            sum = fmaf(w1.w, fmaxf(q1.w, 0.f), sum);
            sum = wave_sum(sum);
            if (lane == 0) {
              float* dst = lg + i * CW + (j - jb);
              *dst = sl == 0 ? sum : *dst + sum;
            }
          }
        }
      }
      __syncthreads();
      for (int q = threadIdx.x; q < (n - jb - 1) * cw; q += blockDim.x) {
        const int i = jb + 1 + q / cw, jj = q % cw;
        if (jb + jj < i) edge_logits[((size_t)b * N + i) * N + jb + jj] = lg[i * CW + jj] + bias2;
      }
    }
  }
}

// the same with one wave per node row in row order (rows of one level side by side)
template <int VPL>
__global__ __launch_bounds__(256) void pair_fwd_rows_kernel(SchedView sv, TypeShare ts, int R,
                                                            const float* __restrict__ A,
                                                            const float* __restrict__ Bq,
                                                            const float* __restrict__ Cq,
                                                            const float* __restrict__ w2,
                                                            const float* __restrict__ b2,
                                                            float* __restrict__ edge_logits) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  float4 wv[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) wv[v] = ld4(w2 + (lane + 64 * v) * 4);
  const float bias2 = b2[0];
  for (int r = wid; r < R; r += nw) pair_fwd_row<VPL>(sv, ts, R, r, lane, wv, bias2, A, Bq, Cq, edge_logits);
}

// one wave per prefix row p = (i,k): pairs j in [u_{k+1}, u_k - 1]
template <int VPL>
__global__ __launch_bounds__(256) void pair_bwd_a_kernel(SchedView sv, TypeShare ts, int R, int RH, const float* __restrict__ A,
                                                         const float* __restrict__ Bq,
                                                         const float* __restrict__ Cq,
                                                         const float* __restrict__ w2,
                                                         const float* __restrict__ d_edge,
                                                         float* __restrict__ DA, float* __restrict__ part) {
  constexpr int EHP = VPL * 256;
  const SchedDir& S = sv.dir[0];
  const int N = sv.N;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  float4 wv[VPL], dw[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    wv[v] = ld4(w2 + (lane + 64 * v) * 4);
    dw[v] = zero4();
  }
  float db = 0.f;
  for (int p = wid; p < RH; p += nw) {
    const int r = S.prow_node[p];
    const int i = S.row_node[r], b = S.row_graph[r];
    const int e0 = S.csr_ptr[r], deg = S.csr_ptr[r + 1] - e0;
    const int k = p < R ? 0 : p - R - e0 + 1;
    const int hi = (k == 0 ? i : S.row_node[S.csr_idx[e0 + k - 1]]) - 1;
    const int lo = k < deg ? S.row_node[S.csr_idx[e0 + k]] : 0;
    float4 a[VPL], da[VPL];
    const float* ar = p < R ? a_row(sv, ts, A, EHP, R, p, 0) : A + (size_t)p * EHP;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      da[v] = zero4();
      a[v] = ld4(ar + (lane + 64 * v) * 4);
      if (Cq) a[v] = add4(a[v], ld4(Cq + (size_t)b * EHP + (lane + 64 * v) * 4));
    }
    for (int j = lo; j <= hi; ++j) {
      const float ds = d_edge[((size_t)b * N + i) * N + j];
      const size_t rj = S.node2row[b * N + j];
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const float4 q = add4(a[v], ld4(Bq + rj * EHP + (lane + 64 * v) * 4));
#define PB_LANE(f)                                   \
        if (q.f > 0.f) {                             \
          da[v].f = fmaf(ds, wv[v].f, da[v].f);      \
          dw[v].f = fmaf(ds, q.f, dw[v].f);          \
        }
        PB_LANE(x) PB_LANE(y) PB_LANE(z) PB_LANE(w)
#undef PB_LANE
      }
      db += ds;
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) st4(DA + (size_t)p * EHP + (lane + 64 * v) * 4, da[v]);
  }
  float* out = part + (size_t)wid * (EHP + 4);
#pragma unroll
  for (int v = 0; v < VPL; ++v) st4(out + (lane + 64 * v) * 4, dw[v]);
  if (lane == 0) st4(out + EHP, make_float4(db, 0.f, 0.f, 0.f));
}

// one wave per node j: pairs (i,j) for every later node i of the same graph
template <int VPL>
__global__ __launch_bounds__(256) void pair_bwd_b_kernel(SchedView sv, TypeShare ts, int R, const int* __restrict__ sizes,
                                                         const float* __restrict__ A,
                                                         const float* __restrict__ Bq,
                                                         const float* __restrict__ Cq,
                                                         const float* __restrict__ w2,
                                                         const float* __restrict__ d_edge,
                                                         float* __restrict__ DBq) {
  constexpr int EHP = VPL * 256;
  const SchedDir& S = sv.dir[0];
  const int N = sv.N;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  float4 wv[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) wv[v] = ld4(w2 + (lane + 64 * v) * 4);
  for (int rj = wid; rj < R; rj += nw) {
    const int j = S.row_node[rj], b = S.row_graph[rj];
    const int n = graph_size(sizes, b, N);
    float4 bq[VPL], dbq[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      dbq[v] = zero4();
      bq[v] = ld4(Bq + (size_t)rj * EHP + (lane + 64 * v) * 4);
      if (Cq) bq[v] = add4(bq[v], ld4(Cq + (size_t)b * EHP + (lane + 64 * v) * 4));
    }
    for (int i = j + 1; i < n; ++i) {
      const int ri = S.node2row[b * N + i];
      const uint32_t* pm = sv.rowbits + ((size_t)b * N + i) * DVAE_BITW;
      int k = 0;
#pragma unroll
      for (int w = 0; w < DVAE_BITW; ++w) {
        uint32_t m = pm[w];
        if (w < (j >> 5)) m = 0;
        else if (w == (j >> 5)) m &= ((j & 31) == 31) ? 0u : (~0u << ((j & 31) + 1));
        k += __popc(m);
      }
      const float* ar = a_row(sv, ts, A, EHP, R, ri, k);
      const float ds = d_edge[((size_t)b * N + i) * N + j];
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const float4 q = add4(bq[v], ld4(ar + (lane + 64 * v) * 4));
        if (q.x > 0.f) dbq[v].x = fmaf(ds, wv[v].x, dbq[v].x);
        if (q.y > 0.f) dbq[v].y = fmaf(ds, wv[v].y, dbq[v].y);
        if (q.z > 0.f) dbq[v].z = fmaf(ds, wv[v].z, dbq[v].z);
        if (q.w > 0.f) dbq[v].w = fmaf(ds, wv[v].w, dbq[v].w);
      }
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) st4(DBq + (size_t)rj * EHP + (lane + 64 * v) * 4, dbq[v]);
  }
}

// Fused backward of the pair scorer for small graphs: one workgroup per graph walks the graph's
// pairs once, in prefix-row order, and produces BOTH input gradients -- DA per prefix row (kept in
// registers while the row's pairs pass) and DBq per candidate node (NREG > 0: in registers, for
// graphs of up to NREG nodes; NREG = 0: in LDS, [N][EHP]; either way each thread owns its columns,
// so no barrier is needed) -- plus the w2/b2 partials.  pair_bwd_a + pair_bwd_b read every A and
// Bq row twice from HBM/L2 and recompute the ReLU mask twice.
template <int VPL, int NREG>
__global__ __launch_bounds__(256) void pair_bwd_fused_kernel(SchedView sv, TypeShare ts, int R,
                                                             const int* __restrict__ sizes,
                                                             const float* __restrict__ A,
                                                             const float* __restrict__ Bq,
                                                             const float* __restrict__ Cq,
                                                             const float* __restrict__ w2,
                                                             const float* __restrict__ d_edge,
                                                             float* __restrict__ DA, float* __restrict__ DBq,
                                                             float* __restrict__ part) {
  constexpr int EHP = VPL * 256, Q = VPL / 4;
  static_assert(VPL % 4 == 0, "one float4 per thread per 1024 columns");
  extern __shared__ __attribute__((aligned(16))) float dbq[];   // [N][EHP]
  const SchedDir& S = sv.dir[0];
  const int N = sv.N, B = sv.B;
  const int col0 = threadIdx.x * 4;
  float4 wv[Q], dw[Q], dbs[Q];   // dbs: column sums of every DBq row = the gradient of add_edge.0's bias
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    wv[q] = ld4(w2 + col0 + 1024 * q);
    dw[q] = zero4();
    dbs[q] = zero4();
  }
  float db = 0.f;
  constexpr int NR = NREG > 0 ? NREG : 1;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int n = graph_size(sizes, b, N);
    float4 dreg[NR][Q];
    if (NREG > 0) {
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) dreg[j][q] = zero4();
    } else {
      for (int j = 0; j < n; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) st4(dbq + (size_t)j * EHP + col0 + 1024 * q, zero4());
    }
    float4 c[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) c[q] = Cq ? ld4(Cq + (size_t)b * EHP + col0 + 1024 * q) : zero4();
    for (int i = 0; i < n; ++i) {
      const int r = S.node2row[b * N + i];
      const int p = S.csr_ptr[r], deg = S.csr_ptr[r + 1] - p;
      int t = 0, cur = -1;
      int next_pred = deg > 0 ? S.row_node[S.csr_idx[p]] : -1;
      float4 a[Q], da[Q];
      // one pair (i,j): prefix row switch, ReLU mask, the three gradient accumulations
      auto pair = [&](int j, float4* dj) {
        while (t < deg && next_pred > j) {
          ++t;
          next_pred = t < deg ? S.row_node[S.csr_idx[p + t]] : -1;
        }
        if (t != cur) {
          if (cur >= 0) {
            float* dst = DA + (size_t)prow_of(S, R, r, cur) * EHP;
#pragma unroll
            for (int q = 0; q < Q; ++q) st4(dst + col0 + 1024 * q, da[q]);
          }
          const float* ar = a_row(sv, ts, A, EHP, R, r, t);
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            a[q] = add4(ld4(ar + col0 + 1024 * q), c[q]);
            da[q] = zero4();
          }
          cur = t;
        }
        const float ds = d_edge[((size_t)b * N + i) * N + j];
        const float* bj = Bq + (size_t)S.node2row[b * N + j] * EHP;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const float4 x = add4(a[q], ld4(bj + col0 + 1024 * q));
#define PF_LANE(f)                                   \
          if (x.f > 0.f) {                           \
            const float u = ds * wv[q].f;            \
            da[q].f += u;                            \
            dj[q].f += u;                            \
            dw[q].f = fmaf(ds, x.f, dw[q].f);        \
          }
          PF_LANE(x) PF_LANE(y) PF_LANE(z) PF_LANE(w)
#undef PF_LANE
        }
        db += ds;
      };
      if (NREG > 0) {
#pragma unroll
        for (int j = NR - 2; j >= 0; --j)
          if (j < i) pair(j, dreg[j]);
      } else {
        for (int j = i - 1; j >= 0; --j) {
          float4 g[Q];
#pragma unroll
          for (int q = 0; q < Q; ++q) g[q] = ld4(dbq + (size_t)j * EHP + col0 + 1024 * q);
          pair(j, g);
#pragma unroll
          for (int q = 0; q < Q; ++q) st4(dbq + (size_t)j * EHP + col0 + 1024 * q, g[q]);
        }
      }
      // flush the last visited prefix row; rows without any pair (node 0's row, a final row whose
      // lowest predecessor is node 0) get zeros
      for (int k = 0; k <= deg; ++k) {
        if (k < cur) continue;
        float* dst = DA + (size_t)prow_of(S, R, r, k) * EHP;
#pragma unroll
        for (int q = 0; q < Q; ++q) st4(dst + col0 + 1024 * q, k == cur ? da[q] : zero4());
      }
    }
    if (NREG > 0) {
#pragma unroll
      for (int j = 0; j < NR; ++j)
        if (j < n) {
          float* dst = DBq + (size_t)S.node2row[b * N + j] * EHP;
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            st4(dst + col0 + 1024 * q, dreg[j][q]);
            dbs[q] = add4(dbs[q], dreg[j][q]);
          }
        }
    } else {
      for (int j = 0; j < n; ++j) {
        float* dst = DBq + (size_t)S.node2row[b * N + j] * EHP;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const float4 v = ld4(dbq + (size_t)j * EHP + col0 + 1024 * q);
          st4(dst + col0 + 1024 * q, v);
          dbs[q] = add4(dbs[q], v);
        }
      }
    }
  }
  // one partial row per block, laid out as the packed gradient has them: [db1 | dw2 | db2]
  float* out = part + (size_t)blockIdx.x * (2 * EHP + 4);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    st4(out + col0 + 1024 * q, dbs[q]);
    st4(out + EHP + col0 + 1024 * q, dw[q]);
  }
  if (threadIdx.x == 0) st4(out + 2 * EHP, make_float4(db, 0.f, 0.f, 0.f));
}

// Fused backward of the pair scorer for LARGER graphs (N > 8: the candidates' accumulators no longer fit
// in registers or LDS at once): the candidate nodes are walked in windows of WIN.  A work item is
// (graph, group of 1024 columns); its 256 threads own one float4 of columns each, so -- as in
// pair_bwd_fused_kernel -- no thread ever needs another thread's data and nothing is reduced across lanes.
// Per window the thread keeps Bq[j] and dL/dBq[j] of the WIN candidates in registers and walks every
// later node's prefix rows that meet the window: an A row is read once per window it meets (once per
// pair in pair_bwd_a / pair_bwd_b, which also read a Bq row per pair), and dL/dA of a row that spans
// several windows is accumulated in place (same thread, same columns: program order, no atomics).
template <int WIN>
__global__ __launch_bounds__(256) void pair_bwd_win_kernel(SchedView sv, TypeShare ts, int R, int EHP, int G,
                                                           const int* __restrict__ sizes,
                                                           const float* __restrict__ A,
                                                           const float* __restrict__ Bq,
                                                           const float* __restrict__ Cq,
                                                           const float* __restrict__ w2,
                                                           const float* __restrict__ d_edge,
                                                           float* __restrict__ DA, float* __restrict__ DBq,
                                                           float* __restrict__ part) {
  const SchedDir& S = sv.dir[0];
  const int N = sv.N, B = sv.B;
  // gridDim.x is a multiple of G: a block keeps its column group over its grid-stride loop
  const int cg = blockIdx.x % G;
  const int col = cg * 1024 + threadIdx.x * 4;
  const float4 wv = ld4(w2 + col);
  float4 dw = zero4(), dbs = zero4();
  float db = 0.f;
  for (long long w = blockIdx.x; w < (long long)B * G; w += gridDim.x) {
    const int b = (int)(w / G);
    const int n = graph_size(sizes, b, N);
    const float4 c = Cq ? ld4(Cq + (size_t)b * EHP + col) : zero4();
    // rows that no pair is scored on: node 0's row, and a full prefix whose lowest predecessor is node 0
    for (int i = 0; i < n; ++i) {
      const int r = S.node2row[b * N + i];
      const int p = S.csr_ptr[r], deg = S.csr_ptr[r + 1] - p;
      if (i == 0) st4(DA + (size_t)prow_of(S, R, r, 0) * EHP + col, zero4());
      else if (deg > 0 && S.row_node[S.csr_idx[p + deg - 1]] == 0) st4(DA + (size_t)prow_of(S, R, r, deg) * EHP + col, zero4());
    }
    for (int jb = 0; jb + 1 < n; jb += WIN) {
      float4 bq[WIN], dj[WIN];
#pragma unroll
      for (int jj = 0; jj < WIN; ++jj) {
        dj[jj] = zero4();
        const int j = jb + jj;
        bq[jj] = j < n ? ld4(Bq + (size_t)S.node2row[b * N + j] * EHP + col) : zero4();
      }
      for (int i = jb + 1; i < n; ++i) {
        const int r = S.node2row[b * N + i];
        const int p = S.csr_ptr[r], deg = S.csr_ptr[r + 1] - p;
        const float* de = d_edge + ((size_t)b * N + i) * N;
        // prefix index of the highest candidate of this window that node i is paired with
        const int jtop = min(jb + WIN, i) - 1;
        int t = 0;
        while (t < deg && S.row_node[S.csr_idx[p + t]] > jtop) ++t;
        int next_pred = t < deg ? S.row_node[S.csr_idx[p + t]] : -1;
        int cur = -1;
        float4 a = zero4(), da = zero4();
        auto flush = [&]() {   // row (i, cur) leaves the window: its first window sets, later ones add
          if (cur < 0) return;
          float* dst = DA + (size_t)prow_of(S, R, r, cur) * EHP + col;
          const int lo = cur < deg ? S.row_node[S.csr_idx[p + cur]] : 0;   // lowest candidate of the row
          st4(dst, lo >= jb ? da : add4(ld4(dst), da));
        };
#pragma unroll
        for (int jj = WIN - 1; jj >= 0; --jj) {
          const int j = jb + jj;
          if (j > jtop) continue;
          while (t < deg && next_pred > j) {
            ++t;
            next_pred = t < deg ? S.row_node[S.csr_idx[p + t]] : -1;
          }
          if (t != cur) {
            flush();
            a = add4(ld4(a_row(sv, ts, A, EHP, R, r, t) + col), c);
            da = zero4();
            cur = t;
          }
          const float ds = de[j];
          const float4 x = add4(a, bq[jj]);
#define PW_LANE(f)                                   \
          if (x.f > 0.f) {                           \
            const float u = ds * wv.f;               \
            da.f += u;                               \
            dj[jj].f += u;                           \
            dw.f = fmaf(ds, x.f, dw.f);              \
          }
          PW_LANE(x) PW_LANE(y) PW_LANE(z) PW_LANE(w)
#undef PW_LANE
          if (cg == 0) db += ds;
        }
        flush();
      }
#pragma unroll
      for (int jj = 0; jj < WIN; ++jj) {
        const int j = jb + jj;
        if (j < n) {
          st4(DBq + (size_t)S.node2row[b * N + j] * EHP + col, dj[jj]);
          dbs = add4(dbs, dj[jj]);
        }
      }
    }
    if (n >= 1 && ((n - 1) % WIN) == 0) {
      // the last node opens no window of its own when n-1 is a multiple of WIN: it is never a candidate
      // (dvae_model.py:210), its dL/dBq row is zero
      st4(DBq + (size_t)S.node2row[b * N + n - 1] * EHP + col, zero4());
    }
  }
  // one partial row per block, [db1 | dw2 | db2] as the packed gradient has them; zeros in the columns of
  // the other column groups
  float* out = part + (size_t)blockIdx.x * (2 * EHP + 4);
  for (int g = 0; g < EHP / 1024; ++g) {
    st4(out + g * 1024 + threadIdx.x * 4, g == cg ? dbs : zero4());
    st4(out + EHP + g * 1024 + threadIdx.x * 4, g == cg ? dw : zero4());
  }
  if (threadIdx.x == 0) st4(out + 2 * EHP, make_float4(db, 0.f, 0.f, 0.f));
}

// predecessor-mask helpers (SchedView::rowbits: bit u of a node's 128-bit mask set <=> u is a predecessor)
struct PredMask { uint32_t w[DVAE_BITW]; };
__device__ __forceinline__ PredMask pred_mask(const SchedView& sv, int b, int N, int i) {
  PredMask m;
  const uint4 v = *reinterpret_cast<const uint4*>(sv.rowbits + ((size_t)b * N + i) * DVAE_BITW);
  m.w[0] = v.x; m.w[1] = v.y; m.w[2] = v.z; m.w[3] = v.w;
  return m;
}
// number of predecessors with index above x (x >= -1)
__device__ __forceinline__ int preds_above(const PredMask& m, int x) {
  int c = 0;
#pragma unroll
  for (int w = 0; w < DVAE_BITW; ++w) {
    const int lo = x + 1 - 32 * w;                      // first bit of word w that counts
    const uint32_t keep = lo <= 0 ? ~0u : lo >= 32 ? 0u : (~0u << lo);
    c += __popc(m.w[w] & keep);
  }
  return c;
}
// highest predecessor index <= x, or -1
__device__ __forceinline__ int pred_at_or_below(const PredMask& m, int x) {
  int best = -1;
#pragma unroll
  for (int w = 0; w < DVAE_BITW; ++w) {
    const int hi = x - 32 * w;                          // last bit of word w that counts
    const uint32_t keep = hi < 0 ? 0u : hi >= 31 ? ~0u : ((2u << hi) - 1u);
    const uint32_t v = m.w[w] & keep;
    if (v) best = 32 * w + 31 - __clz(v);
  }
  return best;
}

// pair_bwd_win_kernel with the walk driven by the predecessor MASK instead of the CSR lists (no dependent index loads
// between a node's rows) and with every load issued one step ahead: the next visited row's A (of this node, or the
// first row the next node meets in this window), the old dL/dA of a row that an earlier window has started, the next
// node's row index / CSR offset / mask.  Row k of node i carries the candidates [u_{k+1}, u_k - 1] (u_0 = i, u_1 > u_2
// > .. the predecessors, none left: 0); within the window [jb, jb + WIN) that is [max(u_{k+1}, jb), min(u_k, jb + WIN) - 1].
template <int WIN>
__global__ __launch_bounds__(256) void pair_bwd_win2_kernel(SchedView sv, TypeShare ts, int R, int EHP, int G,
                                                            const int* __restrict__ sizes, const float* __restrict__ A,
                                                            const float* __restrict__ Bq, const float* __restrict__ Cq,
                                                            const float* __restrict__ w2, const float* __restrict__ d_edge,
                                                            float* __restrict__ DA, float* __restrict__ DBq,
                                                            float* __restrict__ part) {
  const SchedDir& S = sv.dir[0];
  const int N = sv.N, B = sv.B;
  const int cg = blockIdx.x % G;
  const int col = cg * 1024 + threadIdx.x * 4;
  const float4 wv = ld4(w2 + col);
  float4 dw = zero4(), dbs = zero4();
  float db = 0.f;
  for (long long w = blockIdx.x; w < (long long)B * G; w += gridDim.x) {
    const int b = (int)(w / G);
    const int n = graph_size(sizes, b, N);
    const float4 c = Cq ? ld4(Cq + (size_t)b * EHP + col) : zero4();
    // rows that no pair is scored on: node 0's row, and a full prefix whose lowest predecessor is node 0
    for (int i = 0; i < n; ++i) {
      const int r = S.node2row[b * N + i];
      const PredMask m = pred_mask(sv, b, N, i);
      if (i == 0) st4(DA + (size_t)prow_of(S, R, r, 0) * EHP + col, zero4());
      else if (m.w[0] & 1u) st4(DA + (size_t)prow_of(S, R, r, preds_above(m, -1)) * EHP + col, zero4());
    }
    for (int jb = 0; jb + 1 < n; jb += WIN) {
      float4 bq[WIN], dj[WIN];
#pragma unroll
      for (int jj = 0; jj < WIN; ++jj) {
        dj[jj] = zero4();
        const int j = jb + jj;
        bq[jj] = j < n ? ld4(Bq + (size_t)S.node2row[b * N + j] * EHP + col) : zero4();
      }
      // node jb + 1: its row index, CSR offset, mask, and the first row it meets in this window
      int r = S.node2row[b * N + jb + 1];
      int p = S.csr_ptr[r];
      PredMask m = pred_mask(sv, b, N, jb + 1);
      int k = preds_above(m, jb);                       // jtop = min(jb + WIN, jb + 1) - 1 = jb
      float4 a_nxt = ld4(a_row(sv, ts, A, EHP, R, r, k) + col);
      for (int i = jb + 1; i < n; ++i) {
        const float* de = d_edge + ((size_t)b * N + i) * N;
        // the next node's bookkeeping is on the way while this node's rows are processed
        int r2 = 0, p2 = 0;
        PredMask m2 = m;
        if (i + 1 < n) {
          r2 = S.node2row[b * N + i + 1];
          m2 = pred_mask(sv, b, N, i + 1);
        }
        int hi = min(jb + WIN, i) - 1;
        for (;;) {
          const int u = pred_at_or_below(m, hi);        // lowest candidate of the row overall (none: node 0)
          const int lo_full = u < 0 ? 0 : u;
          const int lo = max(lo_full, jb);
          const bool first = lo_full >= jb;              // no earlier (lower) window has met this row
          const bool more = u > jb;                      // the node's next row meets this window too (u >= 0 then)
          float* dst = DA + (size_t)prow_of(S, R, r, k) * EHP + col;
          const float4 a = add4(a_nxt, c);
          float4 old = zero4();
          if (!first) old = ld4(dst);
          if (more) {
            a_nxt = ld4(A + ((size_t)R + p + k) * EHP + col);                    // prow(r, k + 1)
          } else if (i + 1 < n) {
            p2 = S.csr_ptr[r2];
            a_nxt = ld4(a_row(sv, ts, A, EHP, R, r2, preds_above(m2, min(jb + WIN, i + 1) - 1)) + col);
          }
          float4 da = zero4();
#pragma unroll
          for (int jj = WIN - 1; jj >= 0; --jj) {
            const int j = jb + jj;
            if (j > hi || j < lo) continue;
            const float ds = de[j];
            const float4 x = add4(a, bq[jj]);
#define PW2_LANE(f)                                  \
            if (x.f > 0.f) {                           \
              const float uu = ds * wv.f;              \
              da.f += uu;                              \
              dj[jj].f += uu;                          \
              dw.f = fmaf(ds, x.f, dw.f);              \
            }
            PW2_LANE(x) PW2_LANE(y) PW2_LANE(z) PW2_LANE(w)
#undef PW2_LANE
            if (cg == 0) db += ds;
          }
          st4(dst, first ? da : add4(old, da));
          if (!more) break;
          hi = u - 1;
          ++k;
        }
        if (i + 1 < n) {
          r = r2; p = p2; m = m2;
          k = preds_above(m, min(jb + WIN, i + 1) - 1);
        }
      }
#pragma unroll
      for (int jj = 0; jj < WIN; ++jj) {
        const int j = jb + jj;
        if (j < n) {
          st4(DBq + (size_t)S.node2row[b * N + j] * EHP + col, dj[jj]);
          dbs = add4(dbs, dj[jj]);
        }
      }
    }
    if (n >= 1 && ((n - 1) % WIN) == 0) {
      // the last node opens no window of its own when n-1 is a multiple of WIN: it is never a candidate
      // (dvae_model.py:210), its dL/dBq row is zero
      st4(DBq + (size_t)S.node2row[b * N + n - 1] * EHP + col, zero4());
    }
  }
  float* out = part + (size_t)blockIdx.x * (2 * EHP + 4);
  for (int g = 0; g < EHP / 1024; ++g) {
    st4(out + g * 1024 + threadIdx.x * 4, g == cg ? dbs : zero4());
    st4(out + EHP + g * 1024 + threadIdx.x * 4, g == cg ? dw : zero4());
  }
  if (threadIdx.x == 0) st4(out + 2 * EHP, make_float4(db, 0.f, 0.f, 0.f));
}

// The same work items for graphs of at most 8 nodes: ONE window holds every candidate (Bq[j] and dL/dBq[j] of j = 0..6 in
// registers), so a node's prefix rows are walked once, in order -- row k of node i carries the pairs (i, j) with
// u_{k+1} <= j < u_k (u_0 = i, u_1 > u_2 > .. the predecessors, u_{deg+1} = 0) -- and the NEXT row that will be visited
// is known before the current one is processed: its A row is loaded one row ahead (the one-row-at-a-time walk left one
// 16-byte load in flight per thread; the kernel was bound by that latency, 3.2 TB/s).  Every dL/dA row is stored exactly
// once (zeros for the rows no pair is scored on).
__global__ __launch_bounds__(256) void pair_bwd_n8_kernel(SchedView sv, TypeShare ts, int R, int EHP, int G,
                                                          const int* __restrict__ sizes, const float* __restrict__ A,
                                                          const float* __restrict__ Bq, const float* __restrict__ Cq,
                                                          const float* __restrict__ w2, const float* __restrict__ d_edge,
                                                          float* __restrict__ DA, float* __restrict__ DBq,
                                                          float* __restrict__ part) {
  constexpr int W = 7;   // candidates 0 .. 6 (the last node of a graph is never one)
  const SchedDir& S = sv.dir[0];
  const int N = sv.N, B = sv.B;
  const int cg = blockIdx.x % G;
  const int col = cg * 1024 + threadIdx.x * 4;
  const float4 wv = ld4(w2 + col);
  float4 dw = zero4(), dbs = zero4();
  float db = 0.f;
  for (long long w = blockIdx.x; w < (long long)B * G; w += gridDim.x) {
    const int b = (int)(w / G);
    const int n = graph_size(sizes, b, N);
    if (n <= 0) continue;
    const float4 c = Cq ? ld4(Cq + (size_t)b * EHP + col) : zero4();
    float4 bq[W], dj[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
      dj[j] = zero4();
      bq[j] = j + 1 < n ? ld4(Bq + (size_t)S.node2row[b * N + j] * EHP + col) : zero4();
    }
    // row (0, 0) has no pair; the walk starts at node 1 with its row 0 already on the way
    st4(DA + (size_t)prow_of(S, R, S.node2row[b * N], 0) * EHP + col, zero4());   // (node 0 has no predecessor: no edge rows)
    int r = n > 1 ? S.node2row[b * N + 1] : 0;
    float4 a_nxt = n > 1 ? ld4(a_row(sv, ts, A, EHP, R, r, 0) + col) : zero4();
    for (int i = 1; i < n; ++i) {
      const int p = S.csr_ptr[r], deg = S.csr_ptr[r + 1] - p;
      const int r_next = i + 1 < n ? S.node2row[b * N + i + 1] : 0;
      const float* de = d_edge + ((size_t)b * N + i) * N;
      int hi = i - 1;
      int lo = deg > 0 ? S.row_node[S.csr_idx[p]] : 0;          // lowest candidate of row 0
      bool have = true;                                          // row 0 of a node i >= 1 always carries pair (i, i-1)
      for (int k = 0; k <= deg; ++k) {
        const float4 a = add4(a_nxt, c);
        // the next row that some pair is scored on: (i, k+1) unless its range is empty, else (i+1, 0)
        const int lo2 = k + 1 < deg ? S.row_node[S.csr_idx[p + k + 1]] : 0;
        const bool have2 = k < deg && lo - 1 >= lo2;
        if (have2) a_nxt = ld4(A + ((size_t)R + p + k) * EHP + col);            // prow(r, k+1) = R + csr_ptr[r] + k
        else if (k == deg && i + 1 < n) a_nxt = ld4(a_row(sv, ts, A, EHP, R, r_next, 0) + col);
        float4 da = zero4();
        if (have) {
#pragma unroll
          for (int j = W - 1; j >= 0; --j) {
            if (j > hi || j < lo) continue;
            const float ds = de[j];
            const float4 x = add4(a, bq[j]);
#define PN_LANE(f)                                   \
            if (x.f > 0.f) {                           \
              const float u = ds * wv.f;               \
              da.f += u;                               \
              dj[j].f += u;                            \
              dw.f = fmaf(ds, x.f, dw.f);              \
            }
            PN_LANE(x) PN_LANE(y) PN_LANE(z) PN_LANE(w)
#undef PN_LANE
            if (cg == 0) db += ds;
          }
        }
        st4(DA + (size_t)prow_of(S, R, r, k) * EHP + col, da);
        hi = lo - 1;
        lo = lo2;
        have = have2;
      }
      r = r_next;
    }
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (j + 1 < n) {
        st4(DBq + (size_t)S.node2row[b * N + j] * EHP + col, dj[j]);
        dbs = add4(dbs, dj[j]);
      }
    st4(DBq + (size_t)S.node2row[b * N + n - 1] * EHP + col, zero4());   // never a candidate (dvae_model.py:210)
  }
  float* out = part + (size_t)blockIdx.x * (2 * EHP + 4);
  for (int g = 0; g < EHP / 1024; ++g) {
    st4(out + g * 1024 + threadIdx.x * 4, g == cg ? dbs : zero4());
    st4(out + EHP + g * 1024 + threadIdx.x * 4, g == cg ? dw : zero4());
  }
  if (threadIdx.x == 0) st4(out + 2 * EHP, make_float4(db, 0.f, 0.f, 0.f));
}

#define VPL_DISPATCH(EHP, CALL)                                                   \
  switch ((EHP) / 256) {                                                          \
    case 1: { constexpr int V = 1; CALL; } break;                                 \
    case 2: { constexpr int V = 2; CALL; } break;                                 \
    case 3: { constexpr int V = 3; CALL; } break;                                 \
    case 4: { constexpr int V = 4; CALL; } break;                                 \
    case 6: { constexpr int V = 6; CALL; } break;                                 \
    case 8: { constexpr int V = 8; CALL; } break;                                 \
    case 12: { constexpr int V = 12; CALL; } break;                               \
    default:                                                                      \
      dvae_set_error("pair kernels: add_edge width %d (padded) is not instantiated", (EHP)); \
      return DVAE_ERR_ARG;                                                        \
  }

}  // namespace

// ===========================================================================
// launchers
// ===========================================================================
int k_agg_fwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* h0, int HP, hipStream_t st,
              int with_pg) {
  if (ne <= ns) return DVAE_OK;
  hipLaunchKernelGGL(agg_fwd_kernel, grid_for((long long)(ne - ns) * (HP / 4)), dim3(256), 0, st, S, b.R, ns, ne,
                     b.prefix, b.GM, b.AGG, h0, HP, with_pg ? b.PG : nullptr, b.GH);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_prefix_gru_fwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* x, int N, int C,
                     const GruW& w, int HP, hipStream_t st, int save) {
  if (ne <= ns) return DVAE_OK;
  hipLaunchKernelGGL(save ? prefix_gru_fwd_kernel<true> : prefix_gru_fwd_kernel<false>,
                     grid_for((long long)(ne - ns) * (HP / 4)), dim3(256), 0, st, S, b.R, ns, ne, b.GM, b.PG, b.AGG, b.GH, b.GHN,
                     b.HV, b.Hn, x, N, C, w, HP);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_prefix_gru_bwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* x, int N, int C, const GruW& w, int HP,
                     hipStream_t st, const GemmPartials* parts, int parts_row0) {
  if (ne <= ns) return DVAE_OK;
  hipLaunchKernelGGL(prefix_gru_bwd_kernel, grid_for((long long)(ne - ns) * (HP / 4)), dim3(256), 0, st, S, b.R, ns, ne, b.GM,
                     b.PG, b.GH, b.GHN, b.DH, b.DHn, x, N, C, w, HP, parts ? *parts : GemmPartials{nullptr, 1, 0}, parts_row0, b.apos);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_dgm_gather(const SchedDir& S, int ns, int ne, const SweepBufs& b, int HP, hipStream_t st) {
  if (ne <= ns) return DVAE_OK;
  hipLaunchKernelGGL(dgm_gather_kernel, grid_for((long long)(ne - ns) * (HP / 4)), dim3(256), 0, st, S, ns, ne, b.DH,
                     b.GH, b.DGM, b.PG, HP);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_gm_bwd_apply(int ns, int ne, const SweepBufs& b, int HP, hipStream_t st, const GemmPartials* parts) {
  if (ne <= ns) return DVAE_OK;
  hipLaunchKernelGGL(gm_bwd_apply_kernel, grid_for((long long)(ne - ns) * (HP / 4)), dim3(256), 0, st, ns, ne, b.DGM,
                     b.GMP, HP,
                     parts ? *parts : GemmPartials{nullptr, 1, 0});
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_gru_fwd(const SchedDir& S, int p0, int p1, int no_gh, const SweepBufs& b, const float* x, int N, int C,
              const GruW& w, int HP, hipStream_t st, const GemmPartials* parts) {
  if (p1 <= p0) return DVAE_OK;
  hipLaunchKernelGGL(gru_fwd_kernel, grid_for((long long)(p1 - p0) * (HP / 4)), dim3(256), 0, st, S, b.R, p0, p1,
                     b.prefix, no_gh, b.GH, b.AGG, b.GHN, b.HV, b.Hn, x, N, C, w, HP,
                     parts ? *parts : GemmPartials{nullptr, 1, 0});
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_type_states(int C, const GruW& w, float* TYPEV, int HP, hipStream_t st) {
  hipLaunchKernelGGL(type_states_kernel, grid_for((long long)32 * (HP / 4)), dim3(256), 0, st, C, w, TYPEV, HP);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_type_gru_bwd(int C, const GruW& w, const float* T, float* TG, float* TGN, float* XPT, int HP, hipStream_t st) {
  hipLaunchKernelGGL(type_gru_bwd_kernel, grid_for((long long)32 * (HP / 4)), dim3(256), 0, st, C, w, T, TG, TGN,
                     XPT, HP);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_gm_fwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* posb, int HP, hipStream_t st,
             const GemmPartials* parts) {
  if (ne <= ns) return DVAE_OK;
  hipLaunchKernelGGL(gm_fwd_kernel, grid_for((long long)(ne - ns) * (HP / 4)), dim3(256), 0, st, S, ns, ne,
                     b.GMP, posb, b.GM, HP, parts ? *parts : GemmPartials{nullptr, 1, 0});
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_gm_bn_fwd(const SchedDir& S, int R, const SweepBufs& b, const float* x, int N, int C,
                const float* WgmT, const float* bgm, int HP, hipStream_t st) {
  if (R <= 0) return DVAE_OK;
  hipLaunchKernelGGL(gm_bn_fwd_kernel, grid_for((long long)R * (HP / 4)), dim3(256), 0, st, S, R, b.GMP, b.GM,
                     x, N, C, WgmT, bgm, HP);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_dgm_bwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, int HP, hipStream_t st, double links_per_row) {
  if (ne <= ns) return DVAE_OK;
  // long successor lists (shared encoder rows): lanes side by side on a row's list; HP is a multiple of 64
  if (!b.prefix && links_per_row > 6.0 && (long long)(ne - ns) * (HP / 16) < (1ll << 31)) {
    if (links_per_row > 48.0)
      hipLaunchKernelGGL(dgm_bwd_wide_kernel<64>, dim3((unsigned)((ne - ns) * (HP / 16))), dim3(256), 0, st, S, ns, ne, b.DH,
                         b.GMP, HP);
    else
      hipLaunchKernelGGL(dgm_bwd_wide_kernel<16>, dim3((unsigned)((ne - ns) * (HP / 64))), dim3(256), 0, st, S, ns, ne, b.DH,
                         b.GMP, HP);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
  }
  hipLaunchKernelGGL(dgm_bwd_kernel, grid_for((long long)(ne - ns) * (HP / 4)), dim3(256), 0, st, S, ns, ne,
                     b.prefix, b.DH, b.GMP, HP);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_gru_bwd(const SchedDir& S, int p0, int p1, const SweepBufs& b, int HP, hipStream_t st, const GemmPartials* parts,
              int parts_row0) {
  if (p1 <= p0) return DVAE_OK;
  hipLaunchKernelGGL(gru_bwd_kernel, grid_for((long long)(p1 - p0) * (HP / 4)), dim3(256), 0, st, S, b.R, p0, p1,
                     b.prefix, b.GH, b.GHN, b.AGG, b.DH, b.DHn, HP, parts ? *parts : GemmPartials{nullptr, 1, 0},
                     parts_row0, b.prefix ? b.apos : nullptr);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_build_xp(const SchedDir& S, int prefix, int rows, const float* x, int N, int C, float* XP, hipStream_t st) {
  if (rows <= 0) return DVAE_OK;
  hipLaunchKernelGGL(build_xp_kernel, grid_for((long long)rows * 32), dim3(256), 0, st, S, prefix, rows, x, N, C, XP);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_build_xt(const SchedDir& S, int prefix, int rows, const float* x, int N, int C, int* XT, hipStream_t st) {
  if (rows <= 0) return DVAE_OK;
  hipLaunchKernelGGL(build_xt_kernel, grid_for(rows), dim3(256), 0, st, S, prefix, rows, x, N, C, XT);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_build_px(const SchedDir& S, int R, int npos, int PP, float* PX, hipStream_t st) {
  if (R <= 0) return DVAE_OK;
  hipLaunchKernelGGL(build_px_kernel, grid_for((long long)R * PP), dim3(256), 0, st, S, R, npos, PP, PX);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

namespace {
int colsum_launch(float* X, const float* Y, int ld, int rows, int W, float* out, float* part, size_t part_floats,
                  hipStream_t st) {
  DVAE_REQUIRE((W & 3) == 0 && (ld & 3) == 0, "colsum: widths must be multiples of 4");
  DVAE_REQUIRE(part_floats >= (size_t)W, "colsum: scratch too small");
  // enough chunks to fill the chip (column groups x chunks >= ~1024 blocks), at least 64 rows each
  const int gx = cdiv(W, 64);
  int nch = std::max(1, std::min(cdiv(1024, gx), cdiv(rows, 64)));
  nch = (int)std::max<size_t>(1, std::min<size_t>(nch, part_floats / (size_t)W));
  const int rpc = std::max(1, cdiv(rows, nch));
  nch = std::max(1, cdiv(rows, rpc));
  if (Y) hipLaunchKernelGGL(colsum_kernel<true>, dim3(gx, nch), dim3(256), 0, st, X, Y, ld, rows, rpc, W, part);
  else hipLaunchKernelGGL(colsum_kernel<false>, dim3(gx, nch), dim3(256), 0, st, X, Y, ld, rows, rpc, W, part);
  DVAE_LAUNCH_CHECK();
  return reduce_parts(part, nch, (size_t)W, out, (size_t)W, st);
}
}  // namespace

int k_colsum(const float* X, int ld, int rows, int W, float* out, float* part, size_t part_floats,
             hipStream_t st) {
  return colsum_launch(const_cast<float*>(X), nullptr, ld, rows, W, out, part, part_floats, st);
}

// D *= (Y > 0) in place and out[w] = sum over rows of the masked D: one pass over both tables
int k_relu_bwd_colsum(float* D, const float* Y, int ld, int rows, int W, float* out, float* part,
                      size_t part_floats, hipStream_t st) {
  return colsum_launch(D, Y, ld, rows, W, out, part, part_floats, st);
}

// out[W][ldo] (+)= X[rows][W]^T . [onehot(row_type) | 1 | 0..]  (the layout of the TN product it replaces); C < ldo types
int k_typed_colsum(const float* X, int ld, int rows, int W, const int* row_type, int C, float* out, float* part,
                   size_t part_floats, hipStream_t st, int ldo, int accumulate) {
  DVAE_REQUIRE((W & 3) == 0 && (ld & 3) == 0 && C >= 1 && C < 32 && C < ldo, "typed_colsum: widths must be multiples of 4, 1 <= C < 32");
  if (rows <= 0) {
    if (!accumulate) DVAE_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)W * ldo * 4, st));
    return DVAE_OK;
  }
  DVAE_REQUIRE(part_floats >= (size_t)W * (C + 1), "typed_colsum: scratch too small");
  const int gx = cdiv(W, 256);
  int nch = std::max(1, std::min(cdiv(768, gx), cdiv(rows, 64)));
  nch = (int)std::max<size_t>(1, std::min<size_t>(nch, part_floats / ((size_t)W * (C + 1))));
  const int rpc = std::max(1, cdiv(rows, nch));
  nch = std::max(1, cdiv(rows, rpc));
  if (C <= 8) hipLaunchKernelGGL(typed_colsum_kernel<8>, dim3(gx, nch), dim3(256), 0, st, X, ld, rows, rpc, W, row_type, C, part);
  else if (C <= 16) hipLaunchKernelGGL(typed_colsum_kernel<16>, dim3(gx, nch), dim3(256), 0, st, X, ld, rows, rpc, W, row_type, C, part);
  else hipLaunchKernelGGL(typed_colsum_kernel<32>, dim3(gx, nch), dim3(256), 0, st, X, ld, rows, rpc, W, row_type, C, part);
  DVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(typed_colsum_final_kernel, dim3(cdiv(W, 64), accumulate ? C + 1 : ldo), dim3(256), 0, st, part, nch, C, W, out, ldo, accumulate);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

// unit-test surface of k_typed_colsum (include/dvae_hip.h)
extern "C" int dvae_typed_colsum(const float* X, int ld, int rows, int W, const int* row_type, int C, float* out, int ldo,
                                 int accumulate, float* part, size_t part_floats, void* stream) {
  DVAE_REQUIRE(X && row_type && out && part, "typed_colsum: null pointer");
  DVAE_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)part & 15) == 0, "typed_colsum: operands must be 16-byte aligned");
  return k_typed_colsum(X, ld, rows, W, row_type, C, out, part, part_floats, (hipStream_t)stream, ldo, accumulate);
}

int k_nodes_to_dense(const SchedDir& S, int B, int N, const float* T, int ld, float* out, int W, hipStream_t st) {
  if (B <= 0 || N <= 0) return DVAE_OK;
  hipLaunchKernelGGL(nodes_to_dense_kernel, grid_for((long long)B * N * W), dim3(256), 0, st, S, B, N, T, ld, out, W);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_rows_to_dense(const SchedDir& S, int R, const float* T, int ld, float* out, int N, int W, hipStream_t st) {
  if (R <= 0) return DVAE_OK;
  hipLaunchKernelGGL(rows_to_dense_kernel, grid_for((long long)R * W), dim3(256), 0, st, S, R, T, ld, out, N, W);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_dense_to_rows(const SchedDir& S, int R, const float* in, int N, int W, float* T, int ld, hipStream_t st) {
  if (R <= 0) return DVAE_OK;
  hipLaunchKernelGGL(dense_to_rows_kernel, grid_for((long long)R * ld), dim3(256), 0, st, S, R, in, N, W, T, ld);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_take_last_fwd(const SchedDir& S, const int* sizes, int B, int N, const float* Hn, float* out, int HP,
                    hipStream_t st) {
  hipLaunchKernelGGL(take_last_kernel, grid_for((long long)B * (HP / 4)), dim3(256), 0, st, S, sizes, B, N, Hn,
                     out, HP, 0);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_take_last_bwd(const SchedDir& S, const int* sizes, int B, int N, const float* d_out, float* DH, int HP,
                    hipStream_t st) {
  hipLaunchKernelGGL(take_last_kernel, grid_for((long long)B * (HP / 4)), dim3(256), 0, st, S, sizes, B, N,
                     d_out, DH, HP, 1);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_graph_state_fwd(const SchedDir& S, int R, int N, int mod, const float* Hn, const float* h0, float* GS,
                      int HP, hipStream_t st) {
  if (R <= 0) return DVAE_OK;
  hipLaunchKernelGGL(graph_state_fwd_kernel, grid_for((long long)R * (HP / 4)), dim3(256), 0, st, S, R, N, mod,
                     Hn, h0, GS, HP);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_graph_state_bwd(const SchedDir& S, int R, int B, int N, int mod, const int* sizes, const float* DGS,
                      float* DHn, float* dh0, int HP, hipStream_t st) {
  (void)B;
  if (R <= 0) return DVAE_OK;
  hipLaunchKernelGGL(graph_state_bwd_kernel, grid_for((long long)R * (HP / 4)), dim3(256), 0, st, S, R, N, mod,
                     sizes, DGS, DHn, dh0, HP);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

// shapes the LDS-sliced forward scorer covers, and the candidates per Bq window it takes there
static bool pair_fwd_slice_shape(int N, int EHP) { return N > 16 && N <= 128 && EHP % PF_SL == 0; }
static int pair_fwd_slice_window(int N) { return N <= 64 ? N : 56; }

int k_pair_fwd(const SchedView& sv, const TypeShare& ts, int R, const int* sizes, const float* A, const float* Bq,
               const float* Cq, const float* w2, const float* b2, float* edge_logits, int EHP, hipStream_t st) {
  if (R <= 0) return DVAE_OK;
  static const bool by_rows = getenv("DVAE_PAIR_FWD_ROWS") != nullptr;   // lab: the row-order mapping
  static const int cap = getenv("DVAE_PAIR_FWD_GRID") ? atoi(getenv("DVAE_PAIR_FWD_GRID")) : 4096;   // lab
  // graphs of more than 16 nodes: Bq windows in LDS, the add_edge width in slices (DVAE_NO_PAIR_FWD_SLICE=1: A-B tests)
  static const bool no_slice = getenv("DVAE_NO_PAIR_FWD_SLICE") != nullptr;
  if (!by_rows && !no_slice && pair_fwd_slice_shape(sv.N, EHP)) {
    const int CW = pair_fwd_slice_window(sv.N);
    const size_t lds_bytes = ((size_t)CW * PF_SL + (size_t)sv.N * CW) * sizeof(float);   // <= 143 KB
    static std::once_flag big_lds[16];   // per device: more than 64 KB of LDS is opt-in
    int dev = 0;
    DVAE_CHECK_HIP(hipGetDevice(&dev));
    hipError_t attr_err = hipSuccess;
    std::call_once(big_lds[dev & 15], [&] {
      attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(pair_fwd_slice_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    DVAE_CHECK_HIP(attr_err);
    hipLaunchKernelGGL(pair_fwd_slice_kernel, dim3(std::min(sv.B, 1024)), dim3(1024), lds_bytes, st, sv, ts, R, EHP, CW,
                       sizes, A, Bq, Cq, w2, b2, edge_logits);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
  }
  if (by_rows) {
    const int grid = std::min(cdiv(R, 4), cap);
    VPL_DISPATCH(EHP, hipLaunchKernelGGL((pair_fwd_rows_kernel<V>), dim3(grid), dim3(256), 0, st, sv, ts, R, A, Bq, Cq,
                                         w2, b2, edge_logits));
  } else {
    const int grid = std::min(sv.B, cap);
    VPL_DISPATCH(EHP, hipLaunchKernelGGL((pair_fwd_kernel<V>), dim3(grid), dim3(256), 0, st, sv, ts, R, sizes, A, Bq,
                                         Cq, w2, b2, edge_logits));
  }
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_pair_bwd_a(const SchedView& sv, const TypeShare& ts, int R, int RH, const float* A, const float* Bq, const float* Cq,
                 const float* w2, const float* d_edge, float* DA, float* out, float* part,
                 size_t part_floats, int EHP, hipStream_t st) {
  if (RH <= 0) return DVAE_OK;
  const size_t per = (size_t)EHP + 4;
  int grid = std::min(cdiv(RH, 4), 512);
  grid = (int)std::min<size_t>(grid, part_floats / (per * 4));
  DVAE_REQUIRE(grid >= 1, "pair_bwd_a: scratch too small");
  VPL_DISPATCH(EHP, hipLaunchKernelGGL((pair_bwd_a_kernel<V>), dim3(grid), dim3(256), 0, st, sv, ts, R, RH, A, Bq, Cq,
                                       w2, d_edge, DA, part));
  DVAE_LAUNCH_CHECK();
  // the scratch behind the per-wave partials holds the group sums of the two-stage reduction
  const size_t used = (size_t)grid * 4 * per;
  return reduce_parts_2stage(part, grid * 4, per, part + used, part_floats - used, out, per, st);
}

bool pair_bwd_win_ok(int EHP);
// true when k_pair_bwd_fused covers this shape (the DBq accumulators of a graph fit in 64 KB of LDS)
bool pair_bwd_fused_ok(int N, int EHP) {
  static const bool off = getenv("DVAE_NO_PAIR_FUSE") != nullptr;   // A-B tests
  // graphs of up to 8 nodes at whole groups of 1024 columns take pair_bwd_n8_kernel (k_pair_bwd_win) instead
  static const bool n8_off = getenv("DVAE_PAIR_FUSED_N8") != nullptr;   // A-B tests: the per-graph kernel there, too
  if (N <= 8 && EHP % 1024 == 0 && !n8_off && pair_bwd_win_ok(EHP)) return false;
  return !off && (EHP == 2048 || EHP == 1024 || EHP == 3072) && (N <= 8 || (size_t)N * EHP * sizeof(float) <= 65536);
}

int k_pair_bwd_fused(const SchedView& sv, const TypeShare& ts, int R, const int* sizes, const float* A,
                     const float* Bq, const float* Cq, const float* w2, const float* d_edge, float* DA, float* DBq,
                     float* out, float* part, size_t part_floats, int EHP, hipStream_t st) {
  if (R <= 0) return DVAE_OK;
  const size_t per = (size_t)2 * EHP + 4;   // [db1 | dw2 | db2], contiguous in the packed gradient too
  int grid = std::min(sv.B, 512);
  grid = (int)std::min<size_t>(grid, part_floats / per);
  DVAE_REQUIRE(grid >= 1, "pair_bwd_fused: scratch too small");
  // up to 8 nodes: DBq accumulators in registers (no LDS, more workgroups per CU); else in LDS
  static const bool force_lds = getenv("DVAE_PAIR_LDS") != nullptr;   // lab: same-box A-B
  const bool reg = sv.N <= 8 && !force_lds;
  const size_t lds = reg ? 0 : (size_t)sv.N * EHP * sizeof(float);
  static const int cap = getenv("DVAE_PAIR_BWD_GRID") ? atoi(getenv("DVAE_PAIR_BWD_GRID")) : 1024;   // lab
  grid = std::min(sv.B, reg ? cap : 512);
  grid = (int)std::min<size_t>(grid, part_floats / per * 32 / 33);   // room for the group sums of the 2-stage reduction
  grid = std::max(grid, 1);
#define PAIR_FUSED(V)                                                                                          \
  if (reg)                                                                                                     \
    hipLaunchKernelGGL((pair_bwd_fused_kernel<V, 8>), dim3(grid), dim3(256), 0, st, sv, ts, R, sizes, A, Bq, Cq, \
                       w2, d_edge, DA, DBq, part);                                                             \
  else                                                                                                         \
    hipLaunchKernelGGL((pair_bwd_fused_kernel<V, 0>), dim3(grid), dim3(256), lds, st, sv, ts, R, sizes, A, Bq, \
                       Cq, w2, d_edge, DA, DBq, part)
  switch (EHP / 256) {
    case 4: PAIR_FUSED(4); break;
    case 8: PAIR_FUSED(8); break;
    case 12: PAIR_FUSED(12); break;
    default: dvae_set_error("pair_bwd_fused: width %d not instantiated", EHP); return DVAE_ERR_ARG;
  }
#undef PAIR_FUSED
  DVAE_LAUNCH_CHECK();
  const size_t used = (size_t)grid * per;
  return reduce_parts_2stage(part, grid, per, part + used, part_floats - used, out, per, st);
}

// true when k_pair_bwd_win covers this width (whole groups of 1024 columns)
bool pair_bwd_win_ok(int EHP) {
  static const bool off = getenv("DVAE_NO_PAIR_WIN") != nullptr;   // A-B tests: the two-kernel backward instead
  return !off && EHP >= 1024 && EHP % 1024 == 0;
}

int k_pair_bwd_win(const SchedView& sv, const TypeShare& ts, int R, const int* sizes, const float* A,
                   const float* Bq, const float* Cq, const float* w2, const float* d_edge, float* DA, float* DBq,
                   float* out, float* part, size_t part_floats, int EHP, hipStream_t st) {
  if (R <= 0) return DVAE_OK;
  const int G = EHP / 1024;
  const size_t per = (size_t)2 * EHP + 4;   // [db1 | dw2 | db2], contiguous in the packed gradient too
  static const int cap = getenv("DVAE_PAIR_WIN_GRID") ? atoi(getenv("DVAE_PAIR_WIN_GRID")) : 2048;   // lab
  long long grid = std::min<long long>((long long)sv.B * G, (sv.N <= 8 && !getenv("DVAE_PAIR_WIN_GRID")) ? 1024 : cap);   // n8: 768 .. 1536 measure alike
  grid = std::min<long long>(grid, (long long)(part_floats / per * 32 / 33));   // room for the 2-stage reduction's group sums
  grid = grid / G * G;                       // a block keeps one column group
  DVAE_REQUIRE(grid >= G, "pair_bwd_win: scratch too small");
  static const int win = getenv("DVAE_PAIR_WIN") ? atoi(getenv("DVAE_PAIR_WIN")) : 8;   // lab: candidates per window (8: 119 VGPRs, 16: 185; same-box c3 129.6 vs 131.5 ms)
  static const bool no_n8 = getenv("DVAE_NO_PAIR_N8") != nullptr;   // A-B tests
  if (sv.N <= 8 && !no_n8)
    hipLaunchKernelGGL(pair_bwd_n8_kernel, dim3((unsigned)grid), dim3(256), 0, st, sv, ts, R, EHP, G, sizes, A, Bq, Cq, w2,
                       d_edge, DA, DBq, part);
  else if (win == 8 && !getenv("DVAE_PAIR_WIN_OLD"))   // (DVAE_PAIR_WIN_OLD: the CSR-driven walk, A-B tests)
    hipLaunchKernelGGL((pair_bwd_win2_kernel<8>), dim3((unsigned)grid), dim3(256), 0, st, sv, ts, R, EHP, G, sizes, A, Bq, Cq,
                       w2, d_edge, DA, DBq, part);
  else if (win == 8)
    hipLaunchKernelGGL((pair_bwd_win_kernel<8>), dim3((unsigned)grid), dim3(256), 0, st, sv, ts, R, EHP, G, sizes, A, Bq, Cq,
                       w2, d_edge, DA, DBq, part);
  else
    hipLaunchKernelGGL((pair_bwd_win_kernel<16>), dim3((unsigned)grid), dim3(256), 0, st, sv, ts, R, EHP, G, sizes, A, Bq,
                       Cq, w2, d_edge, DA, DBq, part);
  DVAE_LAUNCH_CHECK();
  const size_t used = (size_t)grid * per;
  return reduce_parts_2stage(part, (int)grid, per, part + used, part_floats - used, out, per, st);
}

// test surface: which pair-scorer kernels a batch of N-node graphs with add_edge width EHP runs on (the same
// predicates the launchers above use; environment overrides for A-B runs aside)
//   1 forward = pair_fwd_slice_kernel   2 ... with more than one Bq window   4 backward = pair_bwd_fused_kernel
//   8 backward = pair_bwd_win_kernel    (no backward bit: the two-kernel pair_bwd_a + pair_bwd_b)
extern "C" int dvae_pair_path(int N, int EHP) {
  int bits = 0;
  if (pair_fwd_slice_shape(N, EHP)) bits |= 1 | (pair_fwd_slice_window(N) < N ? 2 : 0);
  if (pair_bwd_fused_ok(N, EHP)) bits |= 4;
  else if (pair_bwd_win_ok(EHP)) bits |= 8;
  return bits;
}

int k_pair_bwd_b(const SchedView& sv, const TypeShare& ts, int R, const int* sizes, const float* A, const float* Bq,
                 const float* Cq, const float* w2, const float* d_edge, float* DBq, int EHP, hipStream_t st) {
  if (R <= 0) return DVAE_OK;
  const int grid = std::min(cdiv(R, 4), 4096);
  VPL_DISPATCH(EHP, hipLaunchKernelGGL((pair_bwd_b_kernel<V>), dim3(grid), dim3(256), 0, st, sv, ts, R, sizes, A, Bq,
                                       Cq, w2, d_edge, DBq));
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_graph_sum(const SchedDir& S, const int* sizes, int B, int N, const float* T, int W, float* out,
                hipStream_t st) {
  hipLaunchKernelGGL(graph_sum_kernel, grid_for((long long)B * (W / 4)), dim3(256), 0, st, S, sizes, B, N, T, W,
                     out);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_add_dh0(const SchedDir& S, int B, int N, const float* DAGG, float* dh0, int HP, hipStream_t st) {
  hipLaunchKernelGGL(add_dh0_kernel, grid_for((long long)B * (HP / 4)), dim3(256), 0, st, S, B, N, DAGG, dh0, HP);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_pad_rows(const float* in, int rows, int W, float* out, int ld, int ncols, hipStream_t st) {
  if (rows <= 0) return DVAE_OK;
  hipLaunchKernelGGL(pad_rows_kernel, grid_for((long long)rows * ncols), dim3(256), 0, st, in, rows, W, out, ld,
                     ncols);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_pad_rows2(const float* in0, const float* in1, int rows, int W, float* out, int ld, int ncols, hipStream_t st) {
  if (rows <= 0) return DVAE_OK;
  hipLaunchKernelGGL(pad_rows2_kernel, grid_for((long long)rows * 2 * ncols), dim3(256), 0, st, in0, in1, rows, W, out, ld,
                     ncols);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_unpad_rows2(const float* in, int ld, int off, int rows, int W, float* out0, float* out1, hipStream_t st) {
  if (rows <= 0) return DVAE_OK;
  hipLaunchKernelGGL(unpad_rows2_kernel, grid_for((long long)rows * 2 * W), dim3(256), 0, st, in, ld, off, rows, W, out0, out1);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_relu_bwd(float* d, const float* y, size_t n, hipStream_t st) {
  if (n == 0) return DVAE_OK;
  hipLaunchKernelGGL(relu_bwd_kernel, grid_for((long long)n), dim3(256), 0, st, d, y, n);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

// large buffers: the runtime's fill kernel runs 256 workgroups (about 0.6 TB/s measured here)
int k_zero(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return DVAE_OK;
  if (bytes < ((size_t)1 << 20) || (((uintptr_t)p | bytes) & 15)) {
    DVAE_CHECK_HIP(hipMemsetAsync(p, 0, bytes, st));
    return DVAE_OK;
  }
  const size_t n4 = bytes / 16;
  const int grid = (int)std::min<size_t>((n4 + 255) / 256, 4096);
  hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<float4*>(p), n4);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_zero_rows(float* X, int ld, int W, const int* rows, int n, hipStream_t st) {
  if (n <= 0) return DVAE_OK;
  hipLaunchKernelGGL(zero_rows_kernel, grid_for((long long)n * (W / 4)), dim3(256), 0, st, X, ld, W, rows, n);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_unpad_rows(const float* in, int ld, int rows, int W, float* out, hipStream_t st) {
  if (rows <= 0) return DVAE_OK;
  hipLaunchKernelGGL(unpad_rows_kernel, grid_for((long long)rows * W), dim3(256), 0, st, in, ld, rows, W, out);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

int k_add_rows(float* dst, int ldd, const float* src, int lds_, int rows, int W, hipStream_t st) {
  if (rows <= 0) return DVAE_OK;
  hipLaunchKernelGGL(add_rows_kernel, grid_for((long long)rows * W), dim3(256), 0, st, dst, ldd, src, lds_, rows,
                     W);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}
