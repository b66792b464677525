// Packed (zero-padded) parameter layouts of the encoder and the decoder.
// The same offsets address the packed gradient buffers.
#pragma once
#include "kernels.h"

struct EncLayout {
  size_t gru_f, gm_f, Wml, bml;          // forward sweep, heads (lin11 rows [0,Z), lin12 rows [ZP,ZP+Z))
  size_t gru_b, gm_b, W01, b01;          // bidirectional only
  size_t total;
  // gradient-only regions (packed gradient buffer = [0,total) + these)
  size_t xg_gru_f, xg_gm_f, xg_gru_b, xg_gm_b, gtotal;
};
struct DecLayout {
  size_t Wl1, bl1;                        // lin1 [HP][ZP]
  size_t gru, gm;
  size_t Wv1, bv1, Wv2, bv2;              // add_vertex [VHP][HP], [CP][VHP]
  size_t W1a, W1b, W1c, b1, w2, b2;       // add_edge.0 split by input block, add_edge.2
  size_t total;
  size_t xg_gru, xg_gm, gtotal;           // gradient-only regions
};

// Skinny weight gradients come out of TN GEMMs against a 32-column one-hot table
// (node type | 1) or a PP-column one (node position | 1):
//   GRU : [3HP][32] = dgh^T.XP (cols <C: dW_ih r,z gates; col C: db_ih r,z and db_hh)
//         [HP][32]  = dn_pre^T.XP (dW_ih n gate, db_ih n gate)
//   gate/map: [2HP][PP] (None) or [2HP][32] (bn)
static inline int pos_cols(const dvae_dims& d) { return d.mod == 0 ? (d.npos + 1 + 31) / 32 * 32 : 32; }
static inline size_t xg_gru_floats(const dvae_dims& d) { return (size_t)4 * d.HP * 32; }
static inline size_t xg_gm_floats(const dvae_dims& d) { return (size_t)2 * d.HP * pos_cols(d); }

static inline EncLayout enc_layout(const dvae_dims& d) {
  EncLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) / 64 * 64; return r; };
  L.gru_f = take(gru_floats(d.HP, d.C));
  L.gm_f = take(gm_floats(d.HP, d.C, d.npos, d.mod));
  L.Wml = take((size_t)2 * d.ZP * d.HP);
  L.bml = take((size_t)2 * d.ZP);
  L.gru_b = L.gm_b = L.W01 = L.b01 = 0;
  if (d.bidir) {
    L.gru_b = take(gru_floats(d.HP, d.C));
    L.gm_b = take(gm_floats(d.HP, d.C, d.npos, d.mod));
    L.W01 = take((size_t)d.HP * 2 * d.HP);
    L.b01 = take((size_t)d.HP);
  }
  L.total = o;
  L.xg_gru_f = take(xg_gru_floats(d));
  L.xg_gm_f = take(xg_gm_floats(d));
  L.xg_gru_b = L.xg_gm_b = 0;
  if (d.bidir) {
    L.xg_gru_b = take(xg_gru_floats(d));
    L.xg_gm_b = take(xg_gm_floats(d));
  }
  L.gtotal = o;
  return L;
}

static inline DecLayout dec_layout(const dvae_dims& d) {
  DecLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) / 64 * 64; return r; };
  L.Wl1 = take((size_t)d.HP * d.ZP);
  L.bl1 = take((size_t)d.HP);
  L.gru = take(gru_floats(d.HP, d.C));
  L.gm = take(gm_floats(d.HP, d.C, d.npos, d.mod));
  L.Wv1 = take((size_t)d.VHP * d.HP);
  L.bv1 = take((size_t)d.VHP);
  L.Wv2 = take((size_t)d.CP * d.VHP);
  L.bv2 = take((size_t)d.CP);
  L.W1a = take((size_t)d.EHP * d.HP);
  L.W1b = take((size_t)d.EHP * d.HP);
  L.W1c = d.mod ? take((size_t)d.EHP * d.HP) : 0;
  L.b1 = take((size_t)d.EHP);
  L.w2 = take((size_t)d.EHP);
  L.b2 = take(64);
  L.total = o;
  L.xg_gru = take(xg_gru_floats(d));
  L.xg_gm = take(xg_gm_floats(d));
  L.gtotal = o;
  return L;
}
