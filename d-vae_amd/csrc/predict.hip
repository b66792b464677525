// Autoregressive decode (Decoder.predict, dvae_model.py:263-362).
// All M latents advance in lockstep over (node i, candidate predecessor j), so
// every step is a dense [M,*] GEMM plus a small decision kernel.  The gated
// state GM_j and the W1b half of add_edge are computed once per finished node
// (the reference recomputes both for every (i,j), :323-336).
#include <algorithm>

#include "layout.h"

namespace {

#define ITEM_LOOP(total) \
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < (total); it += (long long)gridDim.x * blockDim.x)

inline dim3 grid_for(long long items, int cap = 8192) {
  long long g = (items + 255) / 256;
  return dim3((unsigned)std::max<long long>(1, std::min<long long>(g, cap)));
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// node type: argmax (:287) or inverse-CDF draw with an injected uniform (:285)
__global__ void pick_type_kernel(int M, int N, int C, int i, const float* __restrict__ TL, int ldt,
                                 const float* __restrict__ u_type, int* __restrict__ type,
                                 float* __restrict__ onehot, float* __restrict__ marg) {
  ITEM_LOOP(M) {
    const int m = (int)it;
    const float* t = TL + (size_t)m * ldt;
    int best = 0;
    float b1 = t[0], b2 = -INFINITY;
    for (int c = 1; c < C; ++c) {
      const float v = t[c];
      if (v > b1) { b2 = b1; b1 = v; best = c; }
      else if (v > b2) b2 = v;
    }
    int pick = best;
    if (u_type) {
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += expf(t[c] - b1);
      const float u = u_type[(size_t)m * N + i];
      float cdf = 0.f;
      int cnt = 0;
      for (int c = 0; c < C; ++c) {
        cdf += expf(t[c] - b1) / s;
        cnt += (cdf <= u);
      }
      pick = min(cnt, C - 1);
    } else if (C > 1) {
      marg[m] = fminf(marg[m], b1 - b2);
    }
    type[m] = pick;
    float* oh = onehot + ((size_t)m * N + i) * C;
    for (int c = 0; c < C; ++c) oh[c] = c == pick ? 1.f : 0.f;
  }
}

__global__ void gru_pred_kernel(int M, const float* __restrict__ GH, const float* __restrict__ S,
                                const int* __restrict__ type, GruW w, float* __restrict__ HV, int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)M * HP4;
  ITEM_LOOP(total) {
    const int m = (int)(it / HP4), j = (int)(it % HP4) * 4;
    const int c = type[m];
    const float* gh = GH + (size_t)m * 3 * HP;
    const float4 gir = add4(ld4(w.bih + j), ld4(w.WihT + (size_t)c * 3 * HP + j));
    const float4 giz = add4(ld4(w.bih + HP + j), ld4(w.WihT + (size_t)c * 3 * HP + HP + j));
    const float4 gin = add4(ld4(w.bih + 2 * HP + j), ld4(w.WihT + (size_t)c * 3 * HP + 2 * HP + j));
    const float4 ghr = add4(ld4(gh + j), ld4(w.bhh + j));
    const float4 ghz = add4(ld4(gh + HP + j), ld4(w.bhh + HP + j));
    const float4 ghn = add4(ld4(gh + 2 * HP + j), ld4(w.bhh + 2 * HP + j));
    const float4 a = ld4(S + (size_t)m * HP + j);
    float4 h;
#define GL(f)                                              \
    {                                                      \
      const float r = sigm(gir.f + ghr.f);                 \
      const float z = sigm(giz.f + ghz.f);                 \
      const float n = tanhf(gin.f + r * ghn.f);            \
      h.f = (1.f - z) * n + z * a.f;                       \
    }
    GL(x) GL(y) GL(z) GL(w)
#undef GL
    st4(HV + (size_t)m * HP + j, h);
  }
}

// one wave per latent: logit, decision, and S += e * GM_j
template <int VPL>
__global__ __launch_bounds__(256) void edge_decide_kernel(int M, int N, int i, int j,
                                                          const float* __restrict__ APRE,
                                                          const float* __restrict__ BQj,
                                                          const float* __restrict__ Cq,
                                                          const float* __restrict__ w2,
                                                          const float* __restrict__ b2,
                                                          const float* __restrict__ u_edge,
                                                          const float* __restrict__ GMj,
                                                          float* __restrict__ S, float* __restrict__ edges,
                                                          float* __restrict__ marg, int HP) {
  constexpr int EHP = VPL * 256;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  float4 wv[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) wv[v] = ld4(w2 + (lane + 64 * v) * 4);
  const float bias2 = b2[0];
  for (int m = wid; m < M; m += nw) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const size_t o = (size_t)m * EHP + (lane + 64 * v) * 4;
      float4 q = add4(ld4(APRE + o), ld4(BQj + o));
      if (Cq) q = add4(q, ld4(Cq + o));
      s = fmaf(wv[v].x, fmaxf(q.x, 0.f), s);
      s = fmaf(wv[v].y, fmaxf(q.y, 0.f), s);
      s = fmaf(wv[v].z, fmaxf(q.z, 0.f), s);
      s = fmaf(wv[v].w, fmaxf(q.w, 0.f), s);
    }
    const float logit = wave_sum(s) + bias2;
    const float pr = 1.f / (1.f + expf(-logit));
    bool e;
    if (u_edge) e = u_edge[((size_t)m * N + i) * N + j] < pr;   // torch.bernoulli (:314)
    else e = pr >= 0.5f;                                         // :316-317
    if (lane == 0) {
      edges[((size_t)m * N + i) * N + j] = e ? 1.f : 0.f;
      if (!u_edge) marg[m] = fminf(marg[m], fabsf(logit));
    }
    if (e) {
      for (int c = lane * 4; c < HP; c += 256) {
        float* sp = S + (size_t)m * HP + c;
        st4(sp, add4(ld4(sp), ld4(GMj + (size_t)m * HP + c)));
      }
    }
  }
}

__global__ void gm_pred_kernel(int M, int i, const float* __restrict__ GMP, const float* __restrict__ posb,
                               float* __restrict__ GM, int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)M * HP4;
  ITEM_LOOP(total) {
    const int m = (int)(it / HP4), j = (int)(it % HP4) * 4;
    const float* gp = GMP + (size_t)m * 2 * HP;
    const float4 g = add4(ld4(gp + j), ld4(posb + (size_t)i * 2 * HP + j));
    const float4 mm = add4(ld4(gp + HP + j), ld4(posb + (size_t)i * 2 * HP + HP + j));
    st4(GM + (size_t)m * HP + j, make_float4(sigm(g.x) * mm.x, sigm(g.y) * mm.y, sigm(g.z) * mm.z, sigm(g.w) * mm.w));
  }
}

__global__ void gm_pred_bn_kernel(int M, const int* __restrict__ type, const float* __restrict__ WgmT,
                                  const float* __restrict__ bgm, float* __restrict__ GM, int HP) {
  const int HP4 = HP >> 2;
  const long long total = (long long)M * HP4;
  ITEM_LOOP(total) {
    const int m = (int)(it / HP4), j = (int)(it % HP4) * 4;
    const int c = type[m];
    const float4 g = add4(ld4(bgm + j), ld4(WgmT + (size_t)c * 2 * HP + j));
    const float4 mm = add4(ld4(bgm + HP + j), ld4(WgmT + (size_t)c * 2 * HP + HP + j));
    st4(GM + (size_t)m * HP + j, make_float4(sigm(g.x) * mm.x, sigm(g.y) * mm.y, sigm(g.z) * mm.z, sigm(g.w) * mm.w));
  }
}

__global__ void fill_kernel(float* p, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

__global__ __launch_bounds__(256) void min_reduce_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float sh[4];
  float m = INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) m = fminf(m, v[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fminf(fminf(sh[0], sh[1]), fminf(sh[2], sh[3]));
}

struct Carver {
  char* base; size_t off;
  float* f(size_t n) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off = (off + n * sizeof(float) + 255) & ~(size_t)255;
    return p;
  }
};

constexpr size_t PRED_GEMM_WS = (size_t)16 << 20;

struct PredWs {
  float *ZPAD, *H0, *GST, *VH, *TL, *S, *GH, *APRE, *GMP, *Cq, *MARG;
  float *HD, *GMT, *BQ;   // [N][M][*]
  int* type;
  float* gemm_ws;
  size_t bytes;
};

PredWs pred_ws(const dvae_dims& d, void* base) {
  PredWs w;
  Carver c{(char*)base, 0};
  const size_t M = d.B, N = d.N;
  w.ZPAD = c.f(M * d.ZP);
  w.H0 = c.f(M * d.HP);
  w.GST = c.f(M * d.HP);
  w.VH = c.f(M * d.VHP);
  w.TL = c.f(M * d.CP);
  w.S = c.f(M * d.HP);
  w.GH = c.f(M * 3 * d.HP);
  w.APRE = c.f(M * d.EHP);
  w.GMP = c.f(M * 2 * d.HP);
  w.Cq = d.mod ? c.f(M * d.EHP) : nullptr;
  w.MARG = c.f(M);
  w.HD = c.f(N * M * d.HP);
  w.GMT = c.f(N * M * d.HP);
  w.BQ = c.f(N * M * d.EHP);
  w.type = reinterpret_cast<int*>(c.f(M));
  w.gemm_ws = c.f(PRED_GEMM_WS / 4);
  w.bytes = c.off;
  return w;
}

int nt(const float* X, int ldx, const float* W, int ldw, float* Y, int ldy, int M, int N, int K,
       const float* bias, int relu, float* ws, hipStream_t st) {
  GemmDesc d{X, ldx, 0, W, ldw, 0, Y, ldy, M, N, K, bias, relu, 0};
  return gemm_f32(d, ws, PRED_GEMM_WS, st);
}

}  // namespace

extern "C" size_t dvae_predict_ws_bytes(const dvae_dims* d) { return pred_ws(*d, nullptr).bytes; }

extern "C" int dvae_decoder_predict(const dvae_dims* dp, const float* z, const float* P, void* ws,
                                    const float* u_type, const float* u_edge, float* edges, float* onehot,
                                    float* min_margin, void* stream) {
  DVAE_REQUIRE(dp && z && P && ws && edges && onehot, "predict: null pointer");
  DVAE_REQUIRE((u_type == nullptr) == (u_edge == nullptr), "predict: u_type and u_edge go together");
  const dvae_dims& d = *dp;
  hipStream_t st = (hipStream_t)stream;
  PredWs w = pred_ws(d, ws);
  const DecLayout L = dec_layout(d);
  const int M = d.B, N = d.N, HP = d.HP, EHP = d.EHP;
  const GruW gw = gru_at(P + L.gru, HP, d.C);
  const float* gm = P + L.gm;
  const size_t MH = (size_t)M * HP;

  hipLaunchKernelGGL(fill_kernel, grid_for(M), dim3(256), 0, st, w.MARG, INFINITY, (size_t)M);
  DVAE_CHECK_HIP(hipMemsetAsync(edges, 0, (size_t)M * N * N * 4, st));
  DVAE_TRY(k_pad_rows(z, M, d.Z, w.ZPAD, d.ZP, d.ZP, st));
  DVAE_TRY(nt(w.ZPAD, d.ZP, P + L.Wl1, d.ZP, w.H0, HP, M, HP, d.ZP, P + L.bl1, 0, w.gemm_ws, st));
  if (d.mod) DVAE_TRY(nt(w.H0, HP, P + L.W1c, HP, w.Cq, EHP, M, EHP, HP, nullptr, 0, w.gemm_ws, st));
  const float* gstate = w.H0;
  for (int i = 0; i < N; ++i) {
    // node type (:282-291)
    DVAE_TRY(nt(gstate, HP, P + L.Wv1, HP, w.VH, d.VHP, M, d.VHP, HP, P + L.bv1, 1, w.gemm_ws, st));
    DVAE_TRY(nt(w.VH, d.VHP, P + L.Wv2, d.VHP, w.TL, d.CP, M, d.CP, d.VHP, P + L.bv2, 0, w.gemm_ws, st));
    hipLaunchKernelGGL(pick_type_kernel, grid_for(M), dim3(256), 0, st, M, N, d.C, i, w.TL, d.CP, u_type, w.type,
                       onehot, w.MARG);
    DVAE_LAUNCH_CHECK();
    // initial state (:294-298)
    float* HV = w.HD + (size_t)i * MH;
    if (i == 0) {
      DVAE_CHECK_HIP(hipMemcpyAsync(w.S, w.H0, MH * 4, hipMemcpyDeviceToDevice, st));
      DVAE_TRY(nt(w.S, HP, gw.Whh, HP, w.GH, 3 * HP, M, 3 * HP, HP, nullptr, 0, w.gemm_ws, st));
    } else {
      DVAE_CHECK_HIP(hipMemsetAsync(w.S, 0, MH * 4, st));
      DVAE_CHECK_HIP(hipMemsetAsync(w.GH, 0, MH * 3 * 4, st));
    }
    hipLaunchKernelGGL(gru_pred_kernel, grid_for((long long)M * (HP / 4)), dim3(256), 0, st, M, w.GH, w.S, w.type,
                       gw, HV, HP);
    DVAE_LAUNCH_CHECK();
    // edges, latest candidate first (:303-339)
    for (int j = i - 1; j >= 0; --j) {
      DVAE_TRY(nt(HV, HP, P + L.W1a, HP, w.APRE, EHP, M, EHP, HP, nullptr, 0, w.gemm_ws, st));
      const int grid = std::min(cdiv(M, 4), 4096);
      const float* BQj = w.BQ + (size_t)j * M * EHP;
      const float* GMj = w.GMT + (size_t)j * MH;
#define EDGE_CALL(V)                                                                                     \
      hipLaunchKernelGGL((edge_decide_kernel<V>), dim3(grid), dim3(256), 0, st, M, N, i, j, w.APRE, BQj, w.Cq,  \
                         P + L.w2, P + L.b2, u_edge, GMj, w.S, edges, w.MARG, HP)
      switch (EHP / 256) {
        case 1: EDGE_CALL(1); break;
        case 2: EDGE_CALL(2); break;
        case 3: EDGE_CALL(3); break;
        case 4: EDGE_CALL(4); break;
        case 6: EDGE_CALL(6); break;
        case 8: EDGE_CALL(8); break;
        case 12: EDGE_CALL(12); break;
        default: dvae_set_error("predict: add_edge width %d not instantiated", EHP); return DVAE_ERR_ARG;
      }
#undef EDGE_CALL
      DVAE_LAUNCH_CHECK();
      DVAE_TRY(nt(w.S, HP, gw.Whh, HP, w.GH, 3 * HP, M, 3 * HP, HP, nullptr, 0, w.gemm_ws, st));
      hipLaunchKernelGGL(gru_pred_kernel, grid_for((long long)M * (HP / 4)), dim3(256), 0, st, M, w.GH, w.S,
                         w.type, gw, HV, HP);
      DVAE_LAUNCH_CHECK();
    }
    // per finished node: gated state and the W1b half of add_edge, once
    if (i + 1 < N) {
      if (d.mod == 0) {
        DVAE_TRY(nt(HV, HP, gm, HP, w.GMP, 2 * HP, M, 2 * HP, HP, nullptr, 0, w.gemm_ws, st));
        hipLaunchKernelGGL(gm_pred_kernel, grid_for((long long)M * (HP / 4)), dim3(256), 0, st, M, i, w.GMP,
                           gm + (size_t)2 * HP * HP, w.GMT + (size_t)i * MH, HP);
      } else {
        hipLaunchKernelGGL(gm_pred_bn_kernel, grid_for((long long)M * (HP / 4)), dim3(256), 0, st, M, w.type, gm,
                           gm + (size_t)d.C * 2 * HP, w.GMT + (size_t)i * MH, HP);
      }
      DVAE_LAUNCH_CHECK();
      DVAE_TRY(nt(HV, HP, P + L.W1b, HP, w.BQ + (size_t)i * M * EHP, EHP, M, EHP, HP, P + L.b1, 0, w.gemm_ws, st));
    }
    // graph state for the next node (:346-354)
    if (d.mod == 0 || i == 0) {
      gstate = HV;
      if (d.mod) {
        DVAE_CHECK_HIP(hipMemcpyAsync(w.GST, HV, MH * 4, hipMemcpyDeviceToDevice, st));
        gstate = w.GST;
      }
    } else {
      hipLaunchKernelGGL(add_inplace_kernel, grid_for((long long)MH), dim3(256), 0, st, w.GST, HV, MH);
      DVAE_LAUNCH_CHECK();
      gstate = w.GST;
    }
  }
  if (min_margin) {
    hipLaunchKernelGGL(min_reduce_kernel, dim3(1), dim3(256), 0, st, w.MARG, M, min_margin);
    DVAE_LAUNCH_CHECK();
  }
  return DVAE_OK;
}
