// Parameter packing: reference-shaped tensors (state_dict names/shapes of
// dvae_model.py:22-51,143-175) <-> the zero-padded buffers the kernels read.
#include <stdarg.h>
#include <stdio.h>

#include "layout.h"
#include "svae_layout.h"

static thread_local char g_err[512] = "";
void dvae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dvae_last_error(void) { return g_err; }
extern "C" int dvae_abi_version(void) { return 1; }

extern "C" int dvae_dims_init(dvae_dims* d) {
  DVAE_REQUIRE(d, "dims: null");
  DVAE_REQUIRE(d->B > 0, "dims: B must be positive");
  DVAE_REQUIRE(d->N > 0 && d->N <= DVAE_MAX_N, "dims: N must be in [1,%d]", DVAE_MAX_N);
  DVAE_REQUIRE(sched_sizes_ok(d->B, d->N), "dims: batch too large for int32 row indices (B*N < 2^30 and B*N*(N+1)/2 < 2^31; got B=%d N=%d)", d->B, d->N);
  DVAE_REQUIRE(d->C > 0 && d->C < DVAE_MAX_C, "dims: C must be in [1,%d]", DVAE_MAX_C);
  DVAE_REQUIRE(d->H > 0 && d->H <= 2048 && d->Z > 0 && d->Z <= 1024, "dims: H/Z out of range");
  DVAE_REQUIRE(d->mod == 0 || d->mod == 1, "dims: mod must be 0 (None) or 1 ('bn')");
  if (d->mod == 0) DVAE_REQUIRE(d->npos >= d->N && d->npos <= DVAE_MAX_N, "dims: npos (max_seq) must be >= N");
  else d->npos = 0;
  d->HP = round_up(d->H, 64);
  d->ZP = round_up(d->Z, 64);
  d->EH = (d->mod ? 6 : 4) * d->H;
  d->EHP = round_up((d->mod ? 6 : 4) * d->HP, 256);
  d->VHP = 2 * d->HP;
  d->CP = 32;
  switch (d->EHP / 256) {
    case 1: case 2: case 3: case 4: case 6: case 8: case 12: break;
    default: dvae_set_error("dims: hidden size %d gives an add_edge width without a kernel instance", d->H);
      return DVAE_ERR_ARG;
  }
  return DVAE_OK;
}

namespace {

struct TableBuilder {
  dvae_pack_entry* t; int n, cap; bool overflow;
  const uint64_t* raw; const int64_t* grad;
  void add(int kind, int idx, size_t raw_off_floats, int rows, int cols, int raw_ld, int raw_col0,
           size_t dst_off, int dst_ld, int aux_idx = -1) {
    if (n >= cap) { overflow = true; return; }
    dvae_pack_entry& e = t[n++];
    e.raw = raw[idx] + raw_off_floats * 4;
    e.grad = (grad && grad[idx] >= 0) ? grad[idx] + (int64_t)raw_off_floats : -1;
    e.aux_raw = aux_idx >= 0 ? raw[aux_idx] : 0;
    e.aux_grad = (aux_idx >= 0 && grad) ? grad[aux_idx] : -1;
    e.dst_off = (int64_t)dst_off;
    e.rows = rows; e.cols = cols; e.raw_ld = raw_ld; e.raw_col0 = raw_col0; e.dst_ld = dst_ld; e.kind = kind;
    e.g_off = (int64_t)dst_off; e.g_ld = dst_ld; e.g_kind = (kind == 0 || kind == 1) ? kind : -1;
  }
  // packed-side only (its gradient arrives through add_g entries)
  void add_p(int kind, int idx, size_t raw_off_floats, int rows, int cols, int raw_ld, int raw_col0,
             size_t dst_off, int dst_ld, int aux_idx = -1) {
    add(kind, idx, raw_off_floats, rows, cols, raw_ld, raw_col0, dst_off, dst_ld, aux_idx);
    if (!overflow) t[n - 1].g_kind = -1;
  }
  // gradient-side only: grad[r*raw_ld + raw_col0 + c] = G[g_off + r*g_ld + c]
  void add_g(int idx, size_t raw_off_floats, int rows, int cols, int raw_ld, int raw_col0, size_t g_off, int g_ld) {
    add(0, idx, raw_off_floats, rows, cols, raw_ld, raw_col0, 0, 0);
    if (overflow) return;
    dvae_pack_entry& e = t[n - 1];
    e.kind = -1; e.g_off = (int64_t)g_off; e.g_ld = g_ld; e.g_kind = 0;
  }
  // GRUCell: raw indices (w_ih, w_hh, b_ih, b_hh)
  void gru(const dvae_dims& d, size_t base, size_t xg, int wih, int whh, int bih, int bhh) {
    const int H = d.H, HP = d.HP, C = d.C;
    const size_t oWihT = base + (size_t)3 * HP * HP, obih = oWihT + (size_t)C * 3 * HP, obhh = obih + 3 * HP;
    const size_t x1 = xg, x2 = xg + (size_t)3 * HP * 32;   // [3HP][32] = dgh^T.XP, [HP][32] = dn_pre^T.XP
    for (int g = 0; g < 3; ++g) {
      add(0, whh, (size_t)g * H * H, H, H, H, 0, base + (size_t)g * HP * HP, HP);
      add_p(1, wih, (size_t)g * H * C, H, C, C, 0, oWihT + (size_t)g * HP, 3 * HP);
      add_p(0, bih, (size_t)g * H, 1, H, H, 0, obih + (size_t)g * HP, HP);
      add_p(0, bhh, (size_t)g * H, 1, H, H, 0, obhh + (size_t)g * HP, HP);
      const size_t gi = g < 2 ? x1 + (size_t)g * HP * 32 : x2;
      add_g(wih, (size_t)g * H * C, H, C, C, 0, gi, 32);
      add_g(bih, (size_t)g * H, H, 1, 1, 0, gi + C, 32);
      add_g(bhh, (size_t)g * H, H, 1, 1, 0, x1 + (size_t)g * HP * 32 + C, 32);
    }
  }
  // gate/map pair: raw indices (gate w, gate b, map w, map b)
  void gm(const dvae_dims& d, size_t base, size_t xg, int gw, int gb, int mw, int mb) {
    const int H = d.H, HP = d.HP, C = d.C, PP = pos_cols(d);
    const size_t xm = xg + (size_t)HP * PP;   // gradient rows of the map half
    if (d.mod == 0) {
      const int ld = H + d.npos;
      const size_t oposb = base + (size_t)2 * HP * HP;
      add(0, gw, 0, H, H, ld, 0, base, HP);
      add(0, mw, 0, H, H, ld, 0, base + (size_t)HP * HP, HP);
      add_p(2, gw, 0, H, d.npos, ld, H, oposb, 2 * HP, gb);
      add_p(2, mw, 0, H, d.npos, ld, H, oposb + HP, 2 * HP, mb);
      add_g(gw, 0, H, d.npos, ld, H, xg, PP);
      add_g(mw, 0, H, d.npos, ld, H, xm, PP);
      add_g(gb, 0, H, 1, 1, 0, xg + d.npos, PP);
      add_g(mb, 0, H, 1, 1, 0, xm + d.npos, PP);
    } else {
      const size_t ob = base + (size_t)C * 2 * HP;
      add_p(1, gw, 0, H, C, C, 0, base, 2 * HP);
      add_p(1, mw, 0, H, C, C, 0, base + HP, 2 * HP);
      add_p(0, gb, 0, 1, H, H, 0, ob, HP);
      add_p(0, mb, 0, 1, H, H, 0, ob + HP, HP);
      add_g(gw, 0, H, C, C, 0, xg, 32);
      add_g(mw, 0, H, C, C, 0, xm, 32);
      add_g(gb, 0, H, 1, 1, 0, xg + C, 32);
      add_g(mb, 0, H, 1, 1, 0, xm + C, 32);
    }
  }
  // S-VAE sequence GRU (nn.GRU layout: gates r|z|n stacked on rows; K = raw input width)
  void seq_gru(const dvae_svae_dims& d, const SvaeGru& w, int K, int KP, int wih, int whh, int bih, int bhh) {
    const int H = d.H, HP = d.HP;
    for (int g = 0; g < 3; ++g) {
      add(0, wih, (size_t)g * H * K, H, K, K, 0, w.Wih + (size_t)g * HP * KP, KP);
      add(0, whh, (size_t)g * H * H, H, H, H, 0, w.Whh + (size_t)g * HP * HP, HP);
      add(0, bih, (size_t)g * H, 1, H, H, 0, w.bih + (size_t)g * HP, HP);
      add(0, bhh, (size_t)g * H, 1, H, H, 0, w.bhh + (size_t)g * HP, HP);
    }
  }
};

__global__ void pack_kernel(const dvae_pack_entry* __restrict__ table, float* __restrict__ P) {
  const dvae_pack_entry e = table[blockIdx.x];
  const float* raw = reinterpret_cast<const float*>(e.raw);
  if (e.kind < 0) return;
  const float* aux = reinterpret_cast<const float*>(e.aux_raw);
  float* dst = P + e.dst_off;
  const long long total = (long long)e.rows * e.cols;
  for (long long it = (long long)blockIdx.y * blockDim.x + threadIdx.x; it < total;
       it += (long long)gridDim.y * blockDim.x) {
    if (e.kind == 0) {
      const int r = (int)(it / e.cols), c = (int)(it % e.cols);
      dst[(size_t)r * e.dst_ld + c] = raw[(size_t)r * e.raw_ld + e.raw_col0 + c];
    } else {
      const int c = (int)(it / e.rows), r = (int)(it % e.rows);
      float v = raw[(size_t)r * e.raw_ld + e.raw_col0 + c];
      if (e.kind == 2) v += aux[r];
      dst[(size_t)c * e.dst_ld + r] = v;
    }
  }
}

__global__ void unpack_kernel(const dvae_pack_entry* __restrict__ table, const float* __restrict__ G,
                              float* __restrict__ flat) {
  const dvae_pack_entry e = table[blockIdx.x];
  if (e.grad < 0 || e.g_kind < 0) return;
  float* grad = flat + e.grad;
  const float* src = G + e.g_off;
  const long long total = (long long)e.rows * e.cols;
  for (long long it = (long long)blockIdx.y * blockDim.x + threadIdx.x; it < total;
       it += (long long)gridDim.y * blockDim.x) {
    if (e.g_kind == 0) {
      const int r = (int)(it / e.cols), c = (int)(it % e.cols);
      grad[(size_t)r * e.raw_ld + e.raw_col0 + c] = src[(size_t)r * e.g_ld + c];
    } else {
      const int c = (int)(it / e.rows), r = (int)(it % e.rows);
      grad[(size_t)r * e.raw_ld + e.raw_col0 + c] = src[(size_t)c * e.g_ld + r];
    }
  }
}

}  // namespace

extern "C" size_t dvae_enc_param_floats(const dvae_dims* d) { return enc_layout(*d).total; }
extern "C" size_t dvae_dec_param_floats(const dvae_dims* d) { return dec_layout(*d).total; }
extern "C" size_t dvae_enc_grad_floats(const dvae_dims* d) { return enc_layout(*d).gtotal; }
extern "C" size_t dvae_dec_grad_floats(const dvae_dims* d) { return dec_layout(*d).gtotal; }
extern "C" int dvae_enc_param_count(const dvae_dims* d) { return d->bidir ? 22 : 12; }
extern "C" int dvae_dec_param_count(const dvae_dims*) { return 18; }

extern "C" int dvae_enc_pack_table(const dvae_dims* dp, const uint64_t* raw, const int64_t* grad,
                                   dvae_pack_entry* table, int max_entries) {
  const dvae_dims& d = *dp;
  const EncLayout L = enc_layout(d);
  TableBuilder tb{table, 0, max_entries, false, raw, grad};
  // canonical order: 0 gating.0.weight 1 gating.0.bias 2 mapping.weight 3 mapping.bias
  // 4 gru.weight_ih 5 gru.weight_hh 6 gru.bias_ih 7 gru.bias_hh 8 lin11.weight 9 lin11.bias
  // 10 lin12.weight 11 lin12.bias | bidir: 12-15 back_gru.{weight_ih,weight_hh,bias_ih,bias_hh}
  // 16 back_mapping.weight 17 back_mapping.bias 18 back_gating.0.weight 19 back_gating.0.bias
  // 20 lin01.weight 21 lin01.bias
  tb.gru(d, L.gru_f, L.xg_gru_f, 4, 5, 6, 7);
  tb.gm(d, L.gm_f, L.xg_gm_f, 0, 1, 2, 3);
  tb.add(0, 8, 0, d.Z, d.H, d.H, 0, L.Wml, d.HP);
  tb.add(0, 10, 0, d.Z, d.H, d.H, 0, L.Wml + (size_t)d.ZP * d.HP, d.HP);
  tb.add(0, 9, 0, 1, d.Z, d.Z, 0, L.bml, d.ZP);
  tb.add(0, 11, 0, 1, d.Z, d.Z, 0, L.bml + d.ZP, d.ZP);
  if (d.bidir) {
    tb.gru(d, L.gru_b, L.xg_gru_b, 12, 13, 14, 15);
    tb.gm(d, L.gm_b, L.xg_gm_b, 18, 19, 16, 17);
    tb.add(0, 20, 0, d.H, d.H, 2 * d.H, 0, L.W01, 2 * d.HP);
    tb.add(0, 20, 0, d.H, d.H, 2 * d.H, d.H, L.W01 + d.HP, 2 * d.HP);
    tb.add(0, 21, 0, 1, d.H, d.H, 0, L.b01, d.HP);
  }
  DVAE_REQUIRE(!tb.overflow, "enc_pack_table: table too small");
  return tb.n;
}

extern "C" int dvae_dec_pack_table(const dvae_dims* dp, const uint64_t* raw, const int64_t* grad,
                                   dvae_pack_entry* table, int max_entries) {
  const dvae_dims& d = *dp;
  const DecLayout L = dec_layout(d);
  TableBuilder tb{table, 0, max_entries, false, raw, grad};
  // canonical order: 0 lin1.weight 1 lin1.bias 2 add_vertex.0.weight 3 add_vertex.0.bias
  // 4 add_vertex.2.weight 5 add_vertex.2.bias 6 add_edge.0.weight 7 add_edge.0.bias
  // 8 add_edge.2.weight 9 add_edge.2.bias 10 gating.0.weight 11 gating.0.bias
  // 12 mapping.weight 13 mapping.bias 14 gru.weight_ih 15 gru.weight_hh 16 gru.bias_ih 17 gru.bias_hh
  tb.add(0, 0, 0, d.H, d.Z, d.Z, 0, L.Wl1, d.ZP);
  tb.add(0, 1, 0, 1, d.H, d.H, 0, L.bl1, d.HP);
  tb.gru(d, L.gru, L.xg_gru, 14, 15, 16, 17);
  tb.gm(d, L.gm, L.xg_gm, 10, 11, 12, 13);
  tb.add(0, 2, 0, 2 * d.H, d.H, d.H, 0, L.Wv1, d.HP);
  tb.add(0, 3, 0, 1, 2 * d.H, 2 * d.H, 0, L.bv1, d.VHP);
  tb.add(0, 4, 0, d.C, 2 * d.H, 2 * d.H, 0, L.Wv2, d.VHP);
  tb.add(0, 5, 0, 1, d.C, d.C, 0, L.bv2, d.CP);
  const int nin = d.mod ? 3 : 2;
  tb.add(0, 6, 0, d.EH, d.H, nin * d.H, 0, L.W1a, d.HP);
  tb.add(0, 6, 0, d.EH, d.H, nin * d.H, d.H, L.W1b, d.HP);
  if (d.mod) tb.add(0, 6, 0, d.EH, d.H, nin * d.H, 2 * d.H, L.W1c, d.HP);
  tb.add(0, 7, 0, 1, d.EH, d.EH, 0, L.b1, d.EHP);
  tb.add(0, 8, 0, 1, d.EH, d.EH, 0, L.w2, d.EHP);
  tb.add(0, 9, 0, 1, 1, 1, 0, L.b2, 4);
  DVAE_REQUIRE(!tb.overflow, "dec_pack_table: table too small");
  return tb.n;
}

// ---- S-VAE baseline (svae_model.py:13-79) ---------------------------------------
extern "C" int dvae_svae_dims_init(dvae_svae_dims* d) {
  DVAE_REQUIRE(d, "svae dims: null");
  DVAE_REQUIRE(d->B > 0 && d->T > 0 && d->T <= 1024, "svae dims: B and T must be positive (T <= 1024)");
  DVAE_REQUIRE(d->C > 0 && d->C <= 32 && d->N > 0 && d->N <= 32, "svae dims: node_type and max_seq_len must be in [1,32]");
  DVAE_REQUIRE(d->H > 0 && d->H <= 2048 && d->Z > 0 && d->Z <= 1024, "svae dims: H/Z out of range");
  d->I = d->C + d->N;
  d->IP = round_up(d->I, 32);
  d->HP = round_up(d->H, 64);
  d->ZP = round_up(d->Z, 64);
  return DVAE_OK;
}
extern "C" size_t dvae_svae_enc_param_floats(const dvae_svae_dims* d) { return svae_enc_layout(*d).total; }
extern "C" size_t dvae_svae_dec_param_floats(const dvae_svae_dims* d) { return svae_dec_layout(*d).total; }
extern "C" int dvae_svae_enc_param_count(const dvae_svae_dims* d) { return d->bidir ? 14 : 8; }
extern "C" int dvae_svae_dec_param_count(const dvae_svae_dims*) { return 14; }

extern "C" int dvae_svae_enc_pack_table(const dvae_svae_dims* dp, const uint64_t* raw, const int64_t* grad,
                                        dvae_pack_entry* table, int max_entries) {
  const dvae_svae_dims& d = *dp;
  const SvaeEncLayout L = svae_enc_layout(d);
  TableBuilder tb{table, 0, max_entries, false, raw, grad};
  // canonical order: 0-3 gru_layer.{weight_ih,weight_hh,bias_ih,bias_hh}_l0  4 fc_mu.weight 5 fc_mu.bias
  // 6 fc_var.weight 7 fc_var.bias | bidir: 8-11 gru_layer.*_l0_reverse  12 hg_unify.weight 13 hg_unify.bias
  tb.seq_gru(d, L.gru[0], d.I, d.IP, 0, 1, 2, 3);
  tb.add(0, 4, 0, d.Z, d.H, d.H, 0, L.Wml, d.HP);
  tb.add(0, 6, 0, d.Z, d.H, d.H, 0, L.Wml + (size_t)d.ZP * d.HP, d.HP);
  tb.add(0, 5, 0, 1, d.Z, d.Z, 0, L.bml, d.ZP);
  tb.add(0, 7, 0, 1, d.Z, d.Z, 0, L.bml + d.ZP, d.ZP);
  if (d.bidir) {
    tb.seq_gru(d, L.gru[1], d.I, d.IP, 8, 9, 10, 11);
    tb.add(0, 12, 0, d.H, d.H, 2 * d.H, 0, L.Wun, 2 * d.HP);
    tb.add(0, 12, 0, d.H, d.H, 2 * d.H, d.H, L.Wun + d.HP, 2 * d.HP);
    tb.add(0, 13, 0, 1, d.H, d.H, 0, L.bun, d.HP);
  }
  DVAE_REQUIRE(!tb.overflow, "svae_enc_pack_table: table too small");
  return tb.n;
}

extern "C" int dvae_svae_dec_pack_table(const dvae_svae_dims* dp, const uint64_t* raw, const int64_t* grad,
                                        dvae_pack_entry* table, int max_entries) {
  const dvae_svae_dims& d = *dp;
  const SvaeDecLayout L = svae_dec_layout(d);
  TableBuilder tb{table, 0, max_entries, false, raw, grad};
  // canonical order: 0-3 gru_layer.{weight_ih,weight_hh,bias_ih,bias_hh}_l0  4 fc_h0.weight 5 fc_h0.bias
  // 6 add_vertex.0.weight 7 add_vertex.0.bias 8 add_vertex.2.weight 9 add_vertex.2.bias
  // 10 add_edges.0.weight 11 add_edges.0.bias 12 add_edges.2.weight 13 add_edges.2.bias
  tb.seq_gru(d, L.gru, d.H, d.HP, 0, 1, 2, 3);
  tb.add(0, 4, 0, d.H, d.Z, d.Z, 0, L.Wh0, d.ZP);
  tb.add(0, 5, 0, 1, d.H, d.H, 0, L.bh0, d.HP);
  tb.add(0, 6, 0, d.H, d.H, d.H, 0, L.W1, d.HP);
  tb.add(0, 10, 0, d.H, d.H, d.H, 0, L.W1 + (size_t)d.HP * d.HP, d.HP);
  tb.add(0, 7, 0, 1, d.H, d.H, 0, L.b1, d.HP);
  tb.add(0, 11, 0, 1, d.H, d.H, 0, L.b1 + d.HP, d.HP);
  tb.add(0, 8, 0, d.C, d.H, d.H, 0, L.Wv2, d.HP);
  tb.add(0, 9, 0, 1, d.C, d.C, 0, L.bv2, 32);
  tb.add(0, 12, 0, d.N, d.H, d.H, 0, L.We2, d.HP);
  tb.add(0, 13, 0, 1, d.N, d.N, 0, L.be2, 32);
  DVAE_REQUIRE(!tb.overflow, "svae_dec_pack_table: table too small");
  return tb.n;
}

extern "C" int dvae_pack_params(const dvae_pack_entry* table, int n, float* packed, void* stream) {
  if (n <= 0) return DVAE_OK;
  hipLaunchKernelGGL(pack_kernel, dim3(n, 64), dim3(256), 0, (hipStream_t)stream, table, packed);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

extern "C" int dvae_unpack_grads(const dvae_pack_entry* table, int n, const float* packed_grad,
                                 float* grad_flat, void* stream) {
  if (n <= 0) return DVAE_OK;
  hipLaunchKernelGGL(unpack_kernel, dim3(n, 64), dim3(256), 0, (hipStream_t)stream, table, packed_grad,
                     grad_flat);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}
