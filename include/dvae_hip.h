/* C ABI of libdvae_hip.so: the gfx950 (MI355X) backend of the D-VAE hot path.
 *
 * The reference (eeishaan/d-vae) has no FFI: its hot path is the Python class
 * API of dvae/models/dvae_model.py.  Each entry point below names the reference
 * lines whose arithmetic it replaces; the Python side that mirrors the class
 * API lives in d-vae_amd/model.py and calls these through ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless the
 *     name ends in _host; all floating point is fp32; all indices are int32.
 *   - `stream` is a hipStream_t passed as void*.
 *   - every function returns DVAE_OK (0) or a negative DVAE_ERR_* code and never
 *     throws or aborts; dvae_last_error() gives the text of the calling thread's
 *     last failure.
 *   - no entry point allocates or frees device memory: the caller passes
 *     workspaces whose sizes come from the *_bytes queries.
 */
#ifndef DVAE_HIP_H
#define DVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  DVAE_OK = 0,
  DVAE_ERR_ARG = -1,   /* bad argument / unsupported size */
  DVAE_ERR_HIP = -2,   /* a HIP runtime call failed */
  DVAE_ERR_GRAPH = -3  /* adjacency is not a 0/1 strictly lower-triangular DAG */
};

const char* dvae_last_error(void);
int dvae_abi_version(void);

/* ------------------------------------------------------------------------
 * Problem description.  Fill the first block and call dvae_dims_init().
 * ---------------------------------------------------------------------- */
typedef struct dvae_dims {
  int B;      /* graphs in the batch                                          */
  int N;      /* max nodes per graph = seq_len (dvae_model.py:55)             */
  int C;      /* node types = num_classes                                     */
  int H;      /* hidden_state_size (501, dvae_model.py:380)                   */
  int Z;      /* latent_space_dim (56, dvae_model.py:16)                      */
  int mod;    /* 0 = None, 1 = 'bn' (dvae_model.py:23,59,73)                  */
  int bidir;  /* is_bidirectional (dvae_model.py:43,89)                       */
  int npos;   /* position columns of the gate/map Linear = max_seq (mod None) */
  /* derived */
  int HP;     /* H rounded up to 64: row stride of every hidden-state table   */
  int ZP;     /* Z rounded up to 64                                           */
  int EH;     /* add_edge hidden width: 4H (None) / 6H (bn) (dvae_model.py:149-160) */
  int EHP;    /* 4*HP / 6*HP                                                  */
  int VHP;    /* 2*HP (add_vertex hidden, dvae_model.py:144-148)              */
  int CP;     /* 32: padded class count                                       */
} dvae_dims;

int dvae_dims_init(dvae_dims* d);

/* ------------------------------------------------------------------------
 * fp32 MFMA GEMM (unit-test surface of the kernel behind every contraction):
 * C[M][N] (+)= sum_k A(m,k) B(n,k) (+bias[n]) (relu).  a_kmajor/b_kmajor say
 * whether the reduction index is the slow (row) index of that operand.
 * Replaces the nn.Linear / nn.GRUCell matmuls of dvae_model.py:77-83,199-244.
 * ---------------------------------------------------------------------- */
int dvae_gemm_f32(const float* A, int lda, int a_kmajor, const float* B, int ldb, int b_kmajor,
                  float* C, int ldc, int M, int N, int K, const float* bias, int relu,
                  int accumulate, float* ws, size_t ws_bytes, void* stream);

/* The same product over a scattered subset of rows (device int32 index arrays), as the
 * decoder uses it for add_edge's first layer, which dvae_model.py:210-220 evaluates on the
 * prefix states some pair is scored on only:
 *   row_map (A m-major):   row m of the product reads A row row_map[m], writes C row row_map[m]
 *   k_map (A, B k-major):  reduction index k reads row k_map[k] of A and of B
 * (at most one of the two; no bias / relu). */
int dvae_gemm_f32_mapped(const float* A, int lda, int a_kmajor, const float* B, int ldb, int b_kmajor,
                         float* C, int ldc, int M, int N, int K, int accumulate, const int* row_map,
                         const int* k_map, float* ws, size_t ws_bytes, void* stream);

/* The product kernel of SMALL levels (<= ~1024 rows; unit-test surface): 32-row x (G x 32-column) tiles, the
 * reduction split over the four waves of a workgroup and combined in LDS in a fixed order.
 *   w_kmajor = 0:  C[M][G*HP] (+)= A[M][K] . W^T,  W = [G*HP][K]   (forward products, G = 1..3 gate blocks)
 *   w_kmajor = 1:  C[M][HP]   (+)= A[M][K] . W,    W = [K][HP]     (backward products, G = 1)
 * HP % 32 == 0, K % 16 == 0.  Same replacements as dvae_gemm_f32 (dvae_model.py:77-83,199-244). */
int dvae_small_gemm(const float* A, int lda, const float* W, int ldw, int w_kmajor, float* C, int ldc,
                    int M, int G, int HP, int K, int accumulate, void* stream);

/* Instrumentation for bench.py (no counterpart in the reference): while enabled
 * (dvae_gemm_profile: 1 = start a new log, 2 = resume the current log, 0 = pause),
 * every GEMM launch is bracketed by HIP events on its own stream.
 * dvae_gemm_profile_read waits for them and fills out12 = 4 x {launches, ms,
 * executed flops} indexed by layout (0 NT, 1 NN, 3 TN), then clears the log;
 * out12[6] (the unused layout slot) = wall milliseconds during which at least
 * one GEMM launch was running (launches on the two streams overlap). */
int dvae_gemm_force_tile(int mode);   /* tuning knob: 0 auto, 1 128x128, 2 64x128, 3 64x64 */
int dvae_gemm_force_splits(int v);     /* lab knob: split-K factor of every launch (0 = automatic) */
int dvae_gemm_variant(int v);          /* lab knob: force a tile shape (0 = automatic) */
int dvae_gemm_pf(int v);               /* A-B knob, x6 core: prefetch form of the GEMM main loop (-1 by shape, 0 plain, 1 split held behind the
                                          products, 2 two k-tiles in flight); results are bit-identical across the forms */
/* arithmetic core of dvae_gemm_f32 and of every product the entry points below launch through it:
 * 0 = v_mfma_f32_32x32x2_f32; 1 = each fp32 operand split exactly into three bf16 pieces while it is staged,
 * the product formed from the six piece products above 2^-24 |a b| on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation (fp32-accurate, 16/6 of the fp32 MFMA rate).  Default 1; environment: DVAE_GEMM_X6.  Same replacements
 * as dvae_gemm_f32. */
int dvae_gemm_x6(int on);
int dvae_gemm_core(void);              /* the core in use: 1 = bf16 x6 (default), 0 = fp32 MFMA */
int dvae_gemm_dbg(int bits);           /* lab knob: 1 skip the epilogue stores, 2 skip reloads (timing ablations) */
int dvae_gemm_tune(int splitk_blocks, int force_prio);   /* tuning knobs: workgroups a split-K launch aims for
                                                           (0 = keep), wave priority override (-1 = caller's) */
int dvae_gemm_profile(int enable);
int dvae_gemm_profile_read(double* out12_host);
/* The same log also brackets the HBM-bound kernels north_star names (gated-sum aggregation, dvae_model.py:77-80, and
 * edge scoring, :211-219): out12 = 4 x {launches, ms, ALGORITHMIC bytes} for 0 pair scorer forward,
 * 1 pair scorer backward, 2 aggregation forward (agg_fwd, prefix_gru_fwd), 3 aggregation backward (dgm_bwd, dgm_gather).
 * Does not clear the log: call it before dvae_gemm_profile_read. */
int dvae_hbm_profile_read(double* out12_host);

/* ------------------------------------------------------------------------
 * Level scheduler.  The reference walks nodes in index order
 * (dvae_model.py:63,93,196); this groups the nodes of every graph by
 * topological depth, level(i) = 0 if pred(i) empty else 1 + max level(pred).
 *
 * g      [B][N][N] fp32 0/1, g[b][i][j]=1 <=> edge j->i, strictly lower-tri
 *        (dataloader.py:75-86)
 * sizes  [B] node counts (ragged batches) or NULL (every graph has N nodes)
 * sched  workspace of dvae_sched_bytes(B,N) bytes, filled by the call
 * info_host  pinned HOST int32[DVAE_SCHED_INFO_INTS(N)], valid after the
 *        stream has been synchronised:
 *          [0] error flag (0 ok)   [1] R = valid nodes   [2] E = edges
 *          [3] forward level count [4] reverse level count
 *          [5] 1 after dvae_sched_types found every node encoding one-hot, else 0
 *          [6] 1 after dvae_sched_types found that the last level holds exactly the last node of
 *              every graph (true for chain-backed ENAS graphs), else 0
 *          [7] 1 when level 0 holds node 0 of every graph and nothing else (no other source node)
 *          [8 .. 8+N]        forward level starts (node rows), [8+nlev] = R
 *          [8+N+1 .. 8+2N+1] reverse level starts
 *          [8+2N+2 .. 8+3N+2] forward level starts in edge space (csr_ptr there)
 *          [8+3N+3 .. 8+4N+3] the same starts counted in SCORED edge rows only: the decoder's
 *              prefix rows that add_edge reads (dvae_model.py:210-220 scores pair (i,j) on the
 *              prefix holding the predecessors above j, so a node's full prefix is read only
 *              when node 0 is not among its predecessors)
 *          [8+4N+4 ..] shared encoder rows (dvae_sched_types): [+0] 1 when valid, [+1] value rows, [+2] links
 *              between them, [+3] spare, [+4 .. +4+N] level starts in value rows, [+5+N .. +5+2N] successor
 *              links before each level start
 * ---------------------------------------------------------------------- */
#define DVAE_SCHED_INFO_INTS(N) (8 + 6 * ((N) + 1) + 4)
size_t dvae_sched_bytes(int B, int N);
int dvae_sched_build(const float* g, const int* sizes, int B, int N, int want_reverse, void* sched,
                     int* info_host, void* stream);
/* Optional second step of the schedule (same stream, before waiting for info_host): looks at
 * the node encodings x [B][N][C] (dataloader.py:62-73 builds them one-hot) and sets
 * info_host[5] = 1 when every valid node's row is exactly one-hot.  The decoder then serves
 * the prefix rows whose GRU input is (type, h_in = 0) from a C-row table instead of
 * computing them per row (about a third of add_edge's first-layer work at ENAS sizes);
 * without this call, or with other encodings, every row is computed.  It also sets
 * info_host[6] (see above): the last node of a graph is never a candidate predecessor, so its
 * W1b product is skipped when those nodes form the (contiguous) last level.
 * Shared encoder rows: with one-hot encodings and at least dvae_enc_merge_tune's row count (default 8192
 * node slots B*N), the call also numbers the VALUES of the forward encoder sweep -- nodes with the same index,
 * type and predecessor values have the same state (dvae_model.py:63-87 reads nothing else) -- and builds the
 * sweep's schedule over one row per value; dvae_encoder_fwd / _bwd then run the forward-direction sweep on those
 * rows (every graph's outputs are unchanged: the shared row is the row each of its nodes would have computed).
 * The grouping is by a 128-bit hash and then CHECKED field by field; a mismatch turns sharing off for the batch. */
int dvae_sched_types(const float* x, int B, int N, int C, void* sched, int* info_host, void* stream);
/* copies one int32 array of the schedule to `out` (device); which:
 * 0 level_fwd[B*N] 1 level_rev[B*N] 2 node2row_fwd[B*N] 3 node2row_rev[B*N]
 * 4 csr_ptr_fwd[R+1] 5 csr_idx_fwd[E] 6 row_graph_fwd[R] 7 row_node_fwd[R]
 * 8 scored edge prefix rows (R + edge index, ascending) 9 the unscored ones
 * shared encoder rows: 10 node2row[B*N] 11 row_graph 12 row_node 13 csr_ptr[B*N+1] 14 csr_idx 15 csc_ptr[B*N+1]
 * 16 csc_idx 17 representative node id of every node [B*N] */
int dvae_sched_export(const void* sched, int B, int N, int which, int* out, size_t max_ints,
                      void* stream);
/* tuning / test knob: dvae_sched_types shares encoder rows from this many node slots (B*N) up; 0 = always,
 * negative = never.  No counterpart in the reference (it walks every node of every graph). */
int dvae_enc_merge_tune(int min_node_slots);

/* ------------------------------------------------------------------------
 * Parameter packing.  Weights keep the reference's state_dict names and
 * shapes on the torch side; the kernels read a zero-padded
 * copy (H -> HP, gates at multiples of HP).  `raw_host` is a HOST array of
 * DEVICE pointers in the canonical order documented in INTEGRATION.md;
 * `grad_off_host` gives, in the same order, the float offset of each
 * tensor's gradient inside one flat gradient buffer (-1: no gradient).
 * ---------------------------------------------------------------------- */
typedef struct dvae_pack_entry {
  uint64_t raw;      /* parameter storage (device address)                  */
  int64_t grad;      /* float offset of its gradient in the flat gradient
                        buffer given to dvae_unpack_grads, -1 = none        */
  uint64_t aux_raw;  /* kind 2: bias added to every position column         */
  int64_t aux_grad;  /* float offset of that bias's gradient, -1 = none     */
  int64_t dst_off;   /* float offset in the packed buffer                   */
  int rows, cols;    /* block extent in the raw tensor                      */
  int raw_ld, raw_col0;
  int dst_ld;
  int kind;          /* packed side: 0 copy, 1 transposed, 2 transposed +
                        aux[r], -1 = entry is gradient-only                 */
  int64_t g_off;     /* gradient side: float offset in the packed GRADIENT
                        buffer (its layout differs where a weight gradient
                        comes out of a [features][32] GEMM)                 */
  int g_ld;
  int g_kind;        /* 0 copy, 1 transposed, -1 = no gradient from here    */
} dvae_pack_entry;

size_t dvae_enc_param_floats(const dvae_dims* d);
size_t dvae_dec_param_floats(const dvae_dims* d);
size_t dvae_enc_grad_floats(const dvae_dims* d);   /* packed gradient buffer sizes */
size_t dvae_dec_grad_floats(const dvae_dims* d);
int dvae_enc_param_count(const dvae_dims* d);   /* raw tensors expected */
int dvae_dec_param_count(const dvae_dims* d);
int dvae_enc_pack_table(const dvae_dims* d, const uint64_t* raw_host, const int64_t* grad_off_host,
                        dvae_pack_entry* table_host, int max_entries);
int dvae_dec_pack_table(const dvae_dims* d, const uint64_t* raw_host, const int64_t* grad_off_host,
                        dvae_pack_entry* table_host, int max_entries);
/* table: DEVICE copy of the host table */
int dvae_pack_params(const dvae_pack_entry* table, int n, float* packed, void* stream);
int dvae_unpack_grads(const dvae_pack_entry* table, int n, const float* packed_grad, float* grad_flat,
                      void* stream);

/* ------------------------------------------------------------------------
 * Encoder (dvae_model.py:53-123): level-synchronous gated-sum aggregation +
 * GRUCell, optional reverse sweep with the lin01 chain, mu/logvar heads.
 *   x      [B][N][C] node encodings            P      packed encoder params
 *   ws     dvae_enc_ws_bytes(d,R,E) bytes, kept by the caller until backward
 *   hv,mu,logvar  outputs [B][H],[B][Z],[B][Z]
 *   states optional [B][N][H] per-node forward states (parity tests), or NULL
 * ---------------------------------------------------------------------- */
size_t dvae_enc_ws_bytes(const dvae_dims* d, int R, int E);
int dvae_encoder_fwd(const dvae_dims* d, const float* x, const int* sizes, const void* sched,
                     const int* info_host, const float* P, void* ws, float* hv, float* mu,
                     float* logvar, float* states, float* states_rev, void* stream);
/* d_hv/d_mu/d_logvar: upstream gradients (any may be NULL); G: packed grads out.
 * side_stream (hipStream_t or NULL): when given, the products that are off the
 * per-level chain (add_edge first layer in decoder_fwd, every weight-gradient
 * product in the *_bwd calls) are issued there and ordered with events; on return
 * `side_stream` may still hold work that writes G / edge_logits' inputs, so the
 * caller orders its next use of G after it (hipStreamWaitEvent).  decoder_fwd
 * joins the side stream itself before it returns. */
int dvae_encoder_bwd(const dvae_dims* d, const float* x, const int* sizes, const void* sched,
                     const int* info_host, const float* P, void* ws, const float* d_hv,
                     const float* d_mu, const float* d_logvar, float* G, void* stream,
                     void* side_stream);

/* ------------------------------------------------------------------------
 * Teacher-forced decoder (dvae_model.py:178-261).
 *   z [B][Z];  edge_logits [B][N][N] (strict lower triangle written, rest 0),
 *   type_logits [B][N][C];  states optional [B][N][H] final node states.
 * ---------------------------------------------------------------------- */
size_t dvae_dec_ws_bytes(const dvae_dims* d, int R, int E);
int dvae_decoder_fwd(const dvae_dims* d, const float* z, const float* x, const int* sizes,
                     const void* sched, const int* info_host, const float* P, void* ws,
                     float* edge_logits, float* type_logits, float* states, void* stream,
                     void* side_stream);
int dvae_decoder_bwd(const dvae_dims* d, const float* z, const float* x, const int* sizes,
                     const void* sched, const int* info_host, const float* P, void* ws,
                     const float* d_edge_logits, const float* d_type_logits, float* G, float* d_z,
                     void* stream, void* side_stream);

/* ------------------------------------------------------------------------
 * One fused train step in two calls: Dvae.forward + training_step's loss (dvae_model.py:415-432) and
 * the backward that Lightning runs after it, without a host language in between.  Same kernels in
 * the same order as the entry points above (results are bit-identical to calling them one by one);
 * this only removes the per-call host work, which is what a small batch is bound by.
 *   part 1: zero Ge/Gd, pack both parameter sets (the decoder's on the side stream), encoder fwd,
 *           reparameterise with the injected eps, decoder fwd, ELBO + gradients, decoder bwd,
 *           reparameterise bwd.  On return Gd is complete once the SIDE stream has drained.
 *   part 2: encoder bwd.  On return Ge is complete once the side stream has drained.
 * The caller unpacks / all-reduces Gd between the two calls (on the side stream), Ge after part 2.
 * ws_step: dvae_step_ws_bytes(d_dec) bytes (mu, logvar, z, logits and their gradients, ELBO partials);
 * ws_enc / ws_dec: as for dvae_encoder_fwd / dvae_decoder_fwd; all kept until part 2 has run.
 * ---------------------------------------------------------------------- */
typedef struct dvae_step {
  const dvae_dims* d_enc; const dvae_dims* d_dec;     /* same B, N, C, H, Z; bidir only in d_enc           */
  const float* g; const float* x; const int* sizes;   /* the batch (sizes may be NULL)                      */
  const void* sched; const int* info_host;            /* dvae_sched_build (+ dvae_sched_types) results      */
  const dvae_pack_entry* enc_table; int enc_n;        /* DEVICE pack tables                                 */
  const dvae_pack_entry* dec_table; int dec_n;
  float* Pe; float* Pd;                               /* packed parameter buffers (written by part 1)       */
  void* ws_enc; void* ws_dec; void* ws_step;
  const float* eps; float beta;                       /* eps [B][Z]; loss = edge + vertex + beta * kl       */
  float* Ge; float* Gd;                               /* packed gradient buffers (zeroed by part 1)         */
  float* out4;                                        /* {edge_loss, vertex_loss, kl_loss, loss}            */
} dvae_step;
size_t dvae_step_ws_bytes(const dvae_dims* d);
int dvae_step_part1(const dvae_step* s, void* stream, void* side_stream);
int dvae_step_part2(const dvae_step* s, void* stream, void* side_stream);

/* Tuning / test knob of the forward level sweep (no counterpart in the reference): node rows
 * per level from which the gate/map step runs as ONE fused launch (GEMM + sigmoid*map
 * epilogue; -1 never, 0 always, -2 = default: 2048; the fused kernels run on the arithmetic core dvae_gemm_x6 selects)
 * and whether the GRU step runs fused (hidden
 * GEMM + gates in one launch; default 0); type_share: whether the decoder serves its type-only
 * prefix rows from the per-type table when dvae_sched_types allows it (default 1).  Every
 * combination computes the same formulas (fp32 summation order may differ). */
int dvae_level_tune(int fused_gm_min_rows, int fused_gru, int type_share);
/* tuning / test knob: per-level products of at most `rows` rows run on the small-level kernels (32-row tiles, the
 * reduction split over the waves of a workgroup, gates fused into the forward products; default 768, environment
 * DVAE_SMALL_ROWS); 0 = never.  Same formulas either way (fp32 summation order differs). */
int dvae_small_rows_tune(int rows);
/* tuning / test knob: decoder sweep levels (mod None) of at least min_rows node rows do not save the GRU gates of their prefix
 * rows in the forward pass; the backward pass rebuilds them from the gated states and hidden products it already reads (same
 * operations in the same order: the same bits).  Default 1024 (DVAE_RECOMPUTE_MIN / DVAE_NO_RECOMPUTE in the environment),
 * negative = never.  Change it between steps only: a backward pass must see the setting its forward pass ran with. */
int dvae_recompute_tune(int min_rows);
/* Per-type column sums (unit-test surface): out32[w][c] = sum over the rows of type c of X[row][w] for c < C, out32[w][C] =
 * the sum over all rows, the other entries of the 32 zero -- the product X^T . [onehot(type) | 1] of the decoder's per-type
 * gradient sums (the backward of Decoder.forward's k = 0 states, dvae_model.py:203-207, shared by node type) as one
 * streaming pass.  The same kernel forms the GRU cells' input-weight / bias gradients (X = dgh, types = node types,
 * dvae_model.py:83,244) and the position columns of the gate/map gradients (types = node index, :71) when every encoding
 * of the batch is one-hot.  row_type[rows] in [0, C), C < 32, C < ldo; W, ld multiples of 4; out has ldo columns per w;
 * accumulate != 0 adds to out instead of setting it; part: scratch of at least W * (C + 1) floats (more scratch = more row
 * chunks in flight). */
int dvae_typed_colsum(const float* X, int ld, int rows, int W, const int* row_type, int C, float* out, int ldo,
                      int accumulate, float* part, size_t part_floats, void* stream);
/* The optimizer of the train step: torch.optim.Adam's update (dvae_model.py:471-475 configure_optimizers, Adam(lr=1e-4);
 * no amsgrad, weight_decay folded into the gradient as torch does) over a FLAT fp32 buffer of n parameters in one
 * streaming pass: g' = g + wd p; m += (1 - beta1)(g' - m); v = beta2 v + (1 - beta2) g'^2;
 * p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps).  step counts from 1.  All four buffers
 * 16-byte aligned, device memory; grad is not modified.  (dvae_amd.optim.FlatAdam keeps a model's parameters, both
 * moments and the flat gradients of Dvae.loss_and_grad in such buffers.) */
int dvae_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                   double beta2, double eps, double weight_decay, long long step, void* stream);
/* lab (no counterpart in the reference): with DVAE_PHASES=1 in the environment the fused step records a timing event on
 * the caller's stream at a dozen phase boundaries (0 start, 1 encoder forward done, 2 decoder sweep done, 3 side stream's
 * add_edge products joined, 4 pairs scored, 5 add_vertex joined, 6 ELBO done, 7 pair backward done, 8 add_edge input
 * gradients done, 9 decoder backward sweep done, 10 end of part 1, 11 encoder backward's chain done, 12 side stream
 * joined).  dvae_phase_read waits for the device, returns the marks' times in ms since the first one and their tags
 * (at most max), and forgets them.  tools/lab/phases.py prints the per-phase means. */
int dvae_phase_read(double* out_ms, int* tags, int max);
/* test surface (no counterpart in the reference): which kernels score add_edge's pairs (dvae_model.py:210-220) for
 * graphs of up to N nodes at add_edge width EHP (dvae_dims.EHP).  Bits: 1 forward on the LDS-sliced scorer, 2 with more
 * than one window of candidate rows, 4 backward fused per graph (accumulators on chip), 8 backward windowed one-pass;
 * no backward bit = the two-kernel backward.  The parity tests assert that a case runs the kernels it is meant for. */
int dvae_pair_path(int N, int EHP);
/* tuning / test knob: share of a sweep's rows whose weight-gradient products are issued on the
 * side stream while the backward sweep is still running (default 0.35; 0 = none), and the row
 * count below which a sweep has no early chunk (default 16384). */
int dvae_wgrad_split_tune(double frac, int min_rows);

/* Side stream for the *_fwd / *_bwd calls above (no counterpart in the reference).
 * It is created with the LOWEST scheduling priority the device offers, so that the
 * per-level chain on the caller's stream wins the CUs whenever both have work and
 * the bulk products on the side stream fill what the chain leaves idle.
 * priority_out (optional, host) receives the priority actually used. */
int dvae_side_stream_create(void** stream_out, int* priority_out);
int dvae_side_stream_destroy(void* stream);

/* ------------------------------------------------------------------------
 * Reparameterise (dvae_model.py:410-413) with injected eps, and its backward.
 * ---------------------------------------------------------------------- */
int dvae_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long long n,
                     void* stream);
int dvae_reparam_bwd(const float* d_z, const float* logvar, const float* eps, float* d_mu,
                     float* d_logvar, long long n, void* stream);

/* ------------------------------------------------------------------------
 * ELBO (dvae_model.py:425-432, 447-454): BCE-with-logits summed over the full
 * NxN block of each graph (n_b x n_b for ragged graphs), -sum log_softmax*x,
 * KL.  out4 = {edge_loss, vertex_loss, kl_loss, loss} (device, 4 floats).
 * When d_edge/d_type/d_mu/d_logvar are non-NULL the gradients of `loss` are
 * written in the same pass.
 * partial: workspace of dvae_elbo_ws_bytes(B) bytes.
 * ---------------------------------------------------------------------- */
size_t dvae_elbo_ws_bytes(int B);
int dvae_elbo(const dvae_dims* d, const float* edge_logits, const float* type_logits,
              const float* g, const float* x, const int* sizes, const float* mu,
              const float* logvar, float beta, float* out4, void* partial, float* d_edge,
              float* d_type, float* d_mu, float* d_logvar, void* stream);

/* dvae_elbo for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving / evaluation loops, harnesses that forbid hipMalloc).  NO counterpart in the reference.
 * Same arguments as dvae_elbo minus `partial`: the per-graph partial sums live in a file-scope
 * device array inside the library (1.5 MiB per GPU, held until the process ends), whose address is
 * resolved with hipGetSymbolAddress once per device: by dvae_side_stream_create() on that device (the
 * library's set-up call) or else by the first dvae_elbo_static call there.  That look-up (hipGetDevice +
 * hipGetSymbolAddress, which may load the code object) must NOT happen inside a stream capture: a caller that
 * captures graphs either creates the side stream first or makes one warm-up call outside the capture.  The
 * call itself performs no hipMalloc / hipFree.  Limit: d->B <= 65536 graphs (DVAE_ERR_ARG beyond that).  All calls on one
 * GPU share the array: calls that overlap on one GPU (two streams, two threads) race; calls ordered
 * on one stream are safe.  Results are bit-identical to dvae_elbo's. */
#define DVAE_ELBO_STATIC_MAX_B 65536
int dvae_elbo_static(const dvae_dims* d, const float* edge_logits, const float* type_logits,
                     const float* g, const float* x, const int* sizes, const float* mu,
                     const float* logvar, float beta, float* out4, float* d_edge, float* d_type,
                     float* d_mu, float* d_logvar, void* stream);

/* ------------------------------------------------------------------------
 * Autoregressive decode (dvae_model.py:263-362).  u_type [M][N] / u_edge
 * [M][N][N] are injected uniforms for the stochastic mode, NULL for greedy.
 * edges [M][N][N] 0/1, onehot [M][N][C].
 * ---------------------------------------------------------------------- */
size_t dvae_predict_ws_bytes(const dvae_dims* d);   /* d->B = M */
int dvae_decoder_predict(const dvae_dims* d, const float* z, const float* P, void* ws,
                         const float* u_type, const float* u_edge, float* edges, float* onehot,
                         float* min_margin, void* stream);

/* ------------------------------------------------------------------------
 * S-VAE baseline (dvae/models/svae_model.py:13-301): the comparison
 * model of the reference's experiments on the same boundary.  A structure is a string of
 * T = max_seq_len - 1 rows [one-hot type | parent flags] (dataloader.py:96-103); the
 * encoder is a (bi)directional GRU over it, the decoder a GRU fed relu(fc_h0 z) at
 * every step with two MLP heads.  Tables are time-major inside the library (one step's
 * rows contiguous), the arguments keep the reference's batch-first layouts.
 * ---------------------------------------------------------------------- */
typedef struct dvae_svae_dims {
  int B;      /* strings in the batch                                          */
  int T;      /* steps = max_seq_len - 1 (svae_model.py:75)                    */
  int C;      /* node_type  (svae_model.py:15)                                 */
  int N;      /* max_seq_len: parent-flag columns (svae_model.py:22)           */
  int H;      /* hidden_size (501)                                             */
  int Z;      /* latent_dim (56)                                               */
  int bidir;  /* bidirectional encoder GRU + hg_unify (svae_model.py:24-33)    */
  /* derived */
  int I;      /* C + N: input width of the encoder GRU                         */
  int IP;     /* I rounded up to 32                                            */
  int HP;     /* H rounded up to 64                                            */
  int ZP;     /* Z rounded up to 64                                            */
} dvae_svae_dims;
int dvae_svae_dims_init(dvae_svae_dims* d);

/* packing: as dvae_enc_pack_table / dvae_dec_pack_table above (canonical tensor order in
 * INTEGRATION.md); the packed gradient buffer has the size and layout of the packed
 * parameters.  Pack / unpack with dvae_pack_params / dvae_unpack_grads. */
size_t dvae_svae_enc_param_floats(const dvae_svae_dims* d);
size_t dvae_svae_dec_param_floats(const dvae_svae_dims* d);
int dvae_svae_enc_param_count(const dvae_svae_dims* d);
int dvae_svae_dec_param_count(const dvae_svae_dims* d);
int dvae_svae_enc_pack_table(const dvae_svae_dims* d, const uint64_t* raw_host, const int64_t* grad_off_host,
                             dvae_pack_entry* table_host, int max_entries);
int dvae_svae_dec_pack_table(const dvae_svae_dims* d, const uint64_t* raw_host, const int64_t* grad_off_host,
                             dvae_pack_entry* table_host, int max_entries);

/* Encoder.forward (svae_model.py:35-44): x [B][T][I] -> mu, logvar [B][Z].
 * ws: dvae_svae_enc_ws_bytes(d) bytes, kept by the caller until the backward call. */
size_t dvae_svae_enc_ws_bytes(const dvae_svae_dims* d);
int dvae_svae_enc_fwd(const dvae_svae_dims* d, const float* x, const float* P, void* ws, float* mu,
                      float* logvar, void* stream);
/* d_mu / d_logvar [B][Z] (either may be NULL); G: packed gradients out (every entry written).
 * side_stream (hipStream_t or NULL), here and in dvae_svae_dec_bwd: when given, the weight-gradient products
 * are issued there beside the through-time chain, ordered with events; on return it may still hold work
 * that writes G, so the caller orders its next use of G after it (as for dvae_encoder_bwd). */
int dvae_svae_enc_bwd(const dvae_svae_dims* d, const float* P, void* ws, const float* d_mu,
                      const float* d_logvar, float* G, void* stream, void* side_stream);

/* Decoder.forward (svae_model.py:73-79): z [B][Z] -> type_scores [B][T][C], edge_scores [B][T][N] */
size_t dvae_svae_dec_ws_bytes(const dvae_svae_dims* d);
int dvae_svae_dec_fwd(const dvae_svae_dims* d, const float* z, const float* P, void* ws,
                      float* type_scores, float* edge_scores, void* stream);
int dvae_svae_dec_bwd(const dvae_svae_dims* d, const float* P, void* ws, const float* d_type_scores,
                      const float* d_edge_scores, float* G, float* d_z, void* stream, void* side_stream);

/* Losses of Svae.training_step / validation_step (svae_model.py:171-182, 210-221): BCE-with-logits
 * (sum) of edge_scores against the parent flags x[..., C:], cross entropy (sum) of type_scores
 * against argmax x[..., :C], KL.  out4 = {edge_loss, vertex_loss, kl_loss, loss}; gradients of
 * `loss` are written when the d_* pointers are non-NULL.  partial: dvae_elbo_ws_bytes(B) bytes. */
int dvae_svae_elbo(const dvae_svae_dims* d, const float* type_scores, const float* edge_scores,
                   const float* x, const float* mu, const float* logvar, float beta, float* out4,
                   void* partial, float* d_type, float* d_edge, float* d_mu, float* d_logvar,
                   void* stream);

/* Svae._predict_from_logits (svae_model.py:112-131) over rows = B*T score rows: greedy (argmax,
 * sigmoid > 0.5) when u_type / u_edge are NULL, otherwise inverse-CDF draws from the injected
 * uniforms u_type [rows], u_edge [rows][N] (the reference draws with torch.multinomial /
 * torch.bernoulli).  onehot [rows][C], flags [rows][N], both exact 0/1. */
int dvae_svae_predict(const dvae_svae_dims* d, long long rows, const float* type_scores,
                      const float* edge_scores, const float* u_type, const float* u_edge,
                      float* onehot, float* flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif
