// Packed parameter layouts of the S-VAE baseline (svae_model.py:13-79).  Every weight keeps the
// reference's [out][in] orientation, zero-padded (H -> HP, gates at multiples of HP); the packed
// GRADIENT buffer has the same layout, so a weight-gradient product writes where its weight sits.
#pragma once
#include "common.h"

// one GRU: Wih [3HP][KP] | Whh [3HP][HP] | bih [3HP] | bhh [3HP]
struct SvaeGru { size_t Wih, Whh, bih, bhh; };
static inline SvaeGru svae_gru_at(size_t base, int HP, int KP) {
  SvaeGru w;
  w.Wih = base; w.Whh = w.Wih + (size_t)3 * HP * KP; w.bih = w.Whh + (size_t)3 * HP * HP; w.bhh = w.bih + 3 * HP;
  return w;
}
static inline size_t svae_gru_floats(int HP, int KP) { return (size_t)3 * HP * KP + (size_t)3 * HP * HP + 6 * HP; }

struct SvaeEncLayout {
  SvaeGru gru[2];    // forward, reverse (bidir)
  size_t Wml, bml;   // fc_mu rows [0,Z), fc_var rows [ZP,ZP+Z): [2ZP][HP] | [2ZP]
  size_t Wun, bun;   // hg_unify [HP][2HP] (column halves at 0 and HP) | [HP]
  size_t total;
};
static inline SvaeEncLayout svae_enc_layout(const dvae_svae_dims& d) {
  SvaeEncLayout L;
  size_t o = 0;
  L.gru[0] = svae_gru_at(o, d.HP, d.IP); o += svae_gru_floats(d.HP, d.IP);
  L.gru[1] = svae_gru_at(o, d.HP, d.IP); if (d.bidir) o += svae_gru_floats(d.HP, d.IP);
  L.Wml = o; o += (size_t)2 * d.ZP * d.HP;
  L.bml = o; o += 2 * d.ZP;
  L.Wun = o; if (d.bidir) o += (size_t)d.HP * 2 * d.HP;
  L.bun = o; if (d.bidir) o += d.HP;
  L.total = o;
  return L;
}

struct SvaeDecLayout {
  size_t Wh0, bh0;   // fc_h0 [HP][ZP] | [HP]
  SvaeGru gru;       // input width HP
  size_t W1, b1;     // add_vertex.0 rows [0,H), add_edges.0 rows [HP,HP+H): [2HP][HP] | [2HP]
  size_t Wv2, bv2;   // add_vertex.2 [32][HP] | [32]
  size_t We2, be2;   // add_edges.2  [32][HP] | [32]
  size_t total;
};
static inline SvaeDecLayout svae_dec_layout(const dvae_svae_dims& d) {
  SvaeDecLayout L;
  size_t o = 0;
  L.Wh0 = o; o += (size_t)d.HP * d.ZP;
  L.bh0 = o; o += d.HP;
  L.gru = svae_gru_at(o, d.HP, d.HP); o += svae_gru_floats(d.HP, d.HP);
  L.W1 = o; o += (size_t)2 * d.HP * d.HP;
  L.b1 = o; o += 2 * d.HP;
  L.Wv2 = o; o += (size_t)32 * d.HP;
  L.bv2 = o; o += 32;
  L.We2 = o; o += (size_t)32 * d.HP;
  L.be2 = o; o += 32;
  L.total = o;
  return L;
}
