// Launchers of the bandwidth-bound kernels of the D-VAE path (kernels.hip).
// Tables are row-major with a row stride of HP floats (HP % 64 == 0); a work item
// is one float4 of one row.
#pragma once
#include "common.h"
#include "sched.h"

// packed weights of one GRUCell: Whh [3HP][HP] | WihT [C][3HP] | bih [3HP] | bhh [3HP]
struct GruW {
  const float* Whh; const float* WihT; const float* bih; const float* bhh;
};
static inline size_t gru_floats(int HP, int C) { return (size_t)3 * HP * HP + (size_t)C * 3 * HP + 6 * HP; }
static inline GruW gru_at(const float* p, int HP, int C) {
  GruW w;
  w.Whh = p; w.WihT = p + (size_t)3 * HP * HP; w.bih = w.WihT + (size_t)C * 3 * HP; w.bhh = w.bih + 3 * HP;
  return w;
}
// gate/map pair, mod None: Wgm [2HP][HP] | posb [npos][2HP]
//                mod bn  : WgmT [C][2HP] | bgm [2HP]
static inline size_t gm_floats(int HP, int C, int npos, int mod) {
  return mod == 0 ? (size_t)2 * HP * HP + (size_t)npos * 2 * HP : (size_t)C * 2 * HP + 2 * HP;
}

// One level sweep's tables (encoder forward, encoder reverse, decoder).
struct SweepBufs {
  int prefix;       // 1: decoder (one row per (node, prefix length k)); 0: one row per node
  int R;            // node rows
  int RH;           // rows of the h-tables: R (+E when prefix)
  float* AGG;       // [RH][HP]  h_in of every row
  float* GH;        // [RH][3HP] hidden GEMM -> (r,z,n) after gru_fwd -> dgh after gru_bwd
  float* GHN;       // [RH][HP]  W_hn h + b_hn -> d n_pre after gru_bwd
  float* HV;        // [RH][HP]  GRU outputs (== Hn when !prefix)
  float* Hn;        // [R][HP]   final node states
  float* GMP;       // [R][2HP]  gate/map pre-activations -> (g,m) -> their grads
  float* GM;        // [R][HP]   g*m
  float* DH;        // [RH][HP]  dL/dHV -> dL/dAGG after gru_bwd
  float* DHn;       // [R][HP]   prefix only: extra gradient of the final states
  float* XP;        // [RH][32]  node encoding | 1 of every row (weight-gradient GEMMs)
  float* PX;        // [R][PP]   None: position one-hot | 1; bn: node encoding | 1 per node
  int* XT;          // [RH]      node type of every row when every encoding is one-hot (XP / PX as index lists: the skinny
                    //           weight gradients are then column sums per type, kernels.hip typed_colsum_kernel)
  // prefix sweeps, mod None ("hidden product per node"): the GRU's hidden product of a prefix row is
  // linear in its h_in = sum of the predecessors' gated states, so it is summed from ONE product per
  // node, PG[u] = Whh.GM[u], instead of being computed per prefix row (E + N rows -> N rows)
  float* PG;        // [R][3HP]  Whh.GM[u] -> after the backward gather: dL/dPG[u] = sum of dgh over the rows that hold u
  float* DGM;       // [R][HP]   backward: dL/dGM[u] (direct part gathered, hidden part added by a GEMM)
  // prefix sweeps, backward: when set (SchedView::a_pos), only the SCORED edge rows of DH were written on entry -- edge row e
  // is scored iff apos[e + 1] != apos[e] -- and the GRU backward takes dL/dHV of the others as zero instead of reading it
  // (they used to be zero-filled, kernels.hip zero_rows_kernel)
  const int* apos = nullptr;
};

// with_pg (prefix sweeps): also sums the nodes' hidden products PG into the GH rows of the prefix rows
int k_agg_fwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* h0, int HP, hipStream_t st,
              int with_pg = 0);
// prefix sweeps with PG, levels >= 1: aggregation (running h_in and running Whh.h_in from PG) and the
// GRUCell epilogue of every edge prefix row of node rows [ns,ne) in one pass
int k_prefix_gru_fwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* x, int N, int C,
                     const GruW& w, int HP, hipStream_t st, int save = 1);
// its backward with the gates REBUILT from GM / PG (for a forward pass run with save = 0): what k_gru_bwd does on the
// edge rows of node rows [ns,ne), parts as there
int k_prefix_gru_bwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* x, int N, int C, const GruW& w, int HP,
                     hipStream_t st, const GemmPartials* parts = nullptr, int parts_row0 = 0);
// prefix sweeps with PG: DGM[u] = sum of DH over the consumer rows that hold u, PG[u] = the same sum of dgh (GH)
int k_dgm_gather(const SchedDir& S, int ns, int ne, const SweepBufs& b, int HP, hipStream_t st);
// ... and, once DGM is complete, through g*m:  GMP <- (dGM*m*g*(1-g), dGM*g)
// parts: the hidden part of DGM as split-K slabs (GemmDesc::partials) still to be added to the direct part
int k_gm_bwd_apply(int ns, int ne, const SweepBufs& b, int HP, hipStream_t st, const GemmPartials* parts = nullptr);
// parts: the hidden product of rows [p0,p1) as split-K slabs (GemmDesc::partials) instead of whole in GH
int k_gru_fwd(const SchedDir& S, int p0, int p1, int no_gh, const SweepBufs& b, const float* x, int N, int C,
              const GruW& w, int HP, hipStream_t st, const GemmPartials* parts = nullptr);
// parts: the gate/map product as split-K slabs (GemmDesc::partials) instead of whole in GMP
int k_gm_fwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* posb, int HP, hipStream_t st,
             const GemmPartials* parts = nullptr);
int k_gm_bn_fwd(const SchedDir& S, int R, const SweepBufs& b, const float* x, int N, int C,
                const float* WgmT, const float* bgm, int HP, hipStream_t st);
// fused level steps (level.hip): hidden GEMM + GRU gates for prefix/node rows [p0,p1) whose
// h_in is in AGG; gate/map GEMM + sigmoid(g)*m for node rows [ns,ne)
int k_level_gru_fwd(const SchedDir& S, int p0, int p1, const SweepBufs& b, const float* x, int N, int C,
                    const GruW& w, int HP, int prio, hipStream_t st);
int k_level_gm_fwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* Wgm, const float* posb,
                   int HP, int prio, hipStream_t st);
// the same two fused steps and plain products for SMALL row counts (level.hip small_kernel: 32-row tiles, the
// reduction split over the four waves of a workgroup)
int k_level_gru_fwd_small(const SchedDir& S, int p0, int p1, const SweepBufs& b, const float* x, int N, int C,
                          const GruW& w, int HP, int prio, hipStream_t st);
int k_level_gm_fwd_small(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* Wgm, const float* posb,
                         int HP, int prio, hipStream_t st);
// the bidirectional encoder's lin01 chain, one launch per step (level.hip; dvae_model.py:119)
int k_chain_fwd_step(const SchedDir& Srev, const int* sizes, int B, int N, int t, const float* hv_t, const float* W01, int ldw,
                     const float* Q, float* hv_next, int HP, int prio, hipStream_t st);
int k_chain_bwd_step(const SchedDir& Srev, const int* sizes, int B, int N, int t, const float* d_next, const float* W01, int ldw,
                     float* d_t, float* DQ, float* DA, int HP, int prio, hipStream_t st);
int k_level_gm_bwd_small(int ns, int ne, const SweepBufs& b, const float* Whh, int HP, int prio, hipStream_t st);
int k_level_gru_bwd_small(int ns, int ne, const SweepBufs& b, const float* Wgm, int HP, int prio, hipStream_t st);
int k_small_gemm(const float* A, int lda, const float* W, int ldw, int w_kmajor, float* C, int ldc, int M, int G, int HP,
                 int K, int accumulate, int prio, hipStream_t st);
// node sweeps, levels >= 1: CSR gated-sum aggregation (in the A-tile loader) + hidden GEMM + GRU gates, one launch
int k_level_agg_gru_fwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, const float* x, int N, int C,
                        const GruW& w, int HP, int prio, hipStream_t st);
// with_gm_grad=0 (mod bn): only gathers dGM into GMP[:, :HP] scaled for the weight grads
// links_per_row: average successor-list length of the rows, when the caller knows it (long lists take a kernel that
// splits each list over lanes); 0 = unknown
int k_dgm_bwd(const SchedDir& S, int ns, int ne, const SweepBufs& b, int HP, hipStream_t st, double links_per_row = 0.0);
// parts: the gate/map input gradient of node rows [parts_row0, ..) as split-K slabs (GemmDesc::partials),
// still to be added to the table that product accumulates into (DH, or DHn for prefix sweeps)
int k_gru_bwd(const SchedDir& S, int p0, int p1, const SweepBufs& b, int HP, hipStream_t st,
              const GemmPartials* parts = nullptr, int parts_row0 = 0);
// one-hot side tables for the skinny weight gradients (see layout.h)
int k_build_xp(const SchedDir& S, int prefix, int rows, const float* x, int N, int C, float* XP, hipStream_t st);
// XT[p] = the column of the 1 in the (one-hot) encoding of row p's node
int k_build_xt(const SchedDir& S, int prefix, int rows, const float* x, int N, int C, int* XT, hipStream_t st);
int k_build_px(const SchedDir& S, int R, int npos, int PP, float* PX, hipStream_t st);
// out[w] = sum_rows X[row][w]   (W % 4 == 0)
int k_colsum(const float* X, int ld, int rows, int W, float* out, float* part, size_t part_floats,
             hipStream_t st);
int k_relu_bwd_colsum(float* D, const float* Y, int ld, int rows, int W, float* out, float* part,
                      size_t part_floats, hipStream_t st);   // D *= (Y > 0), out = column sums of D
// out[W][ldo] (+)= X[rows][W]^T . [onehot(row_type[row]) | 1 | 0 ..]: column sums per node type (C < 32 types, every row
// typed), one streaming pass; the layout of the skinny TN product it replaces
int k_typed_colsum(const float* X, int ld, int rows, int W, const int* row_type, int C, float* out, float* part,
                   size_t part_floats, hipStream_t st, int ldo = 32, int accumulate = 0);

// [B][N][W] API tensors <-> node-row tables
// two [rows][W] tables into the halves of one padded row / back out of them (the mu | logvar heads): one launch each
int k_pad_rows2(const float* in0, const float* in1, int rows, int W, float* out, int ld, int ncols, hipStream_t st);
int k_unpad_rows2(const float* in, int ld, int off, int rows, int W, float* out0, float* out1, hipStream_t st);
int k_rows_to_dense(const SchedDir& S, int R, const float* T, int ld, float* out, int N, int W, hipStream_t st);
// the same through node2row: out[b][i][:W] = T[row of node (b,i)] (0 for absent nodes); for schedules whose rows
// stand for several nodes (SchedView::enc)
int k_nodes_to_dense(const SchedDir& S, int B, int N, const float* T, int ld, float* out, int W, hipStream_t st);
int k_dense_to_rows(const SchedDir& S, int R, const float* in, int N, int W, float* T, int ld, hipStream_t st);

// encoder glue: state of each graph's last node (dvae_model.py:121 uses the loop's final hv)
int k_take_last_fwd(const SchedDir& S, const int* sizes, int B, int N, const float* Hn, float* out, int HP,
                    hipStream_t st);
int k_take_last_bwd(const SchedDir& S, const int* sizes, int B, int N, const float* d_out, float* DH, int HP,
                    hipStream_t st);

// type-shared prefix rows (sched.h TypeShare): TYPEV [32][HP] = GRU(one-hot c, h_in = 0);
// backward: from the per-type sums T[c] = dL/dhv(type), one gate-gradient row per type
// (TG [32][3HP], TGN [32][HP]) and the (type one-hot | 1) table XPT [32][32] they pair with
int k_type_states(int C, const GruW& w, float* TYPEV, int HP, hipStream_t st);
int k_type_gru_bwd(int C, const GruW& w, const float* T, float* TG, float* TGN, float* XPT, int HP, hipStream_t st);

// decoder glue -----------------------------------------------------------------
int k_graph_state_fwd(const SchedDir& S, int R, int N, int mod, const float* Hn, const float* h0, float* GS,
                      int HP, hipStream_t st);
int k_graph_state_bwd(const SchedDir& S, int R, int B, int N, int mod, const int* sizes, const float* DGS,
                      float* DHn, float* dh0, int HP, hipStream_t st);
int k_pair_fwd(const SchedView& sv, const TypeShare& ts, int R, const int* sizes, const float* A, const float* Bq,
               const float* Cq, const float* w2, const float* b2, float* edge_logits, int EHP, hipStream_t st);
// DA [RH][EHP]; out = dw2 [EHP] | db2 [4]
int k_pair_bwd_a(const SchedView& sv, const TypeShare& ts, int R, int RH, const float* A, const float* Bq, const float* Cq,
                 const float* w2, const float* d_edge, float* DA, float* out, float* part,
                 size_t part_floats, int EHP, hipStream_t st);
int k_pair_bwd_b(const SchedView& sv, const TypeShare& ts, int R, const int* sizes, const float* A, const float* Bq,
                 const float* Cq, const float* w2, const float* d_edge, float* DBq, int EHP, hipStream_t st);
// both of the above in one pass, one workgroup per graph (small N only: pair_bwd_fused_ok)
bool pair_bwd_fused_ok(int N, int EHP);
int k_pair_bwd_fused(const SchedView& sv, const TypeShare& ts, int R, const int* sizes, const float* A,
                     const float* Bq, const float* Cq, const float* w2, const float* d_edge, float* DA, float* DBq,
                     float* out, float* part, size_t part_floats, int EHP, hipStream_t st);
// the same for graphs of any size: candidates walked in register-resident windows (pair_bwd_win_ok: widths
// that are whole groups of 1024 columns)
bool pair_bwd_win_ok(int EHP);
int k_pair_bwd_win(const SchedView& sv, const TypeShare& ts, int R, const int* sizes, const float* A,
                   const float* Bq, const float* Cq, const float* w2, const float* d_edge, float* DA, float* DBq,
                   float* out, float* part, size_t part_floats, int EHP, hipStream_t st);
// out[b] = sum over the graph's nodes of T[row]   (W % 4 == 0)
int k_graph_sum(const SchedDir& S, const int* sizes, int B, int N, const float* T, int W, float* out,
                hipStream_t st);
// dh0[b] += DAGG[prefix row 0 of node 0 of graph b]
int k_add_dh0(const SchedDir& S, int B, int N, const float* DAGG, float* dh0, int HP, hipStream_t st);

// misc
// out[r][c] = c < W ? in[r][c] : 0 for c < ncols (in == NULL: zeros)
int k_pad_rows(const float* in, int rows, int W, float* out, int ld, int ncols, hipStream_t st);
int k_relu_bwd(float* d, const float* y, size_t n, hipStream_t st);   // d *= (y > 0)
int k_zero(void* p, size_t bytes, hipStream_t st);   // memset 0, wide kernel for large buffers
int k_zero_rows(float* X, int ld, int W, const int* rows, int n, hipStream_t st);   // X[rows[i]][0..W) = 0
int k_unpad_rows(const float* in, int ld, int rows, int W, float* out, hipStream_t st);
int k_add_rows(float* dst, int ldd, const float* src, int lds_, int rows, int W, hipStream_t st);
