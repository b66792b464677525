// Host orchestration of the level-synchronous encoder / teacher-forced decoder
// (forward and backward).  Everything here is launches on the caller's streams;
// no allocation, no synchronisation with the host.
//
// Two streams: the per-level chain (aggregate -> hidden GEMM -> GRU epilogue ->
// gate/map GEMM) is a sequence of mid-sized launches that cannot fill 256 CUs on
// their own, so everything that does not sit on that chain -- the add_edge
// first-layer products in the forward pass, every weight-gradient product in the
// backward pass -- is issued on `side` and ordered against the chain with events.
// side == NULL (or == stream) runs everything in order on one stream.
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <vector>

#include "layout.h"

namespace {


// lab: DVAE_PHASES=1 records a timing event on the chain at a dozen phase boundaries of the fused step (what the chain
// waits for, and where); dvae_phase_read() returns the times since the first mark.  Off: nothing is created or recorded.
struct Phases {
  bool on = getenv("DVAE_PHASES") != nullptr;
  std::vector<hipEvent_t> ev;
  std::vector<int> tag;
};
Phases g_phases;
void phase_mark(int tag, hipStream_t st) {
  if (!g_phases.on) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess || hipEventRecord(e, st) != hipSuccess) return;
  g_phases.ev.push_back(e);
  g_phases.tag.push_back(tag);
}

struct LevelCfg {
  int fused_gm_min_rows; int fused_gru; int type_share;
  int node_hidden = getenv("DVAE_NO_NODE_HIDDEN") ? 0 : 1;   // prefix sweeps: one hidden product per node (SweepBufs::PG)
  // per-level products of at most this many rows run as level.hip small_kernel launches (32-row tiles, K split
  // over the waves of a workgroup, gates fused in the forward pass); 0 = never.  DVAE_SMALL_ROWS / dvae_small_rows_tune
  // (768 since the x6 GEMM's 64x64 tiles run with two k-tiles in flight: 1024-row levels now take the GEMM path, B = 1024 3.128 ->
  // 3.084 ms same box; 512-row levels keep the small kernels, B = 512 2.05 vs 2.16 without them)
  int small_rows = getenv("DVAE_SMALL_ROWS") ? atoi(getenv("DVAE_SMALL_ROWS")) : 768;
  LevelCfg() {
    type_share = getenv("DVAE_NO_TYPE_SHARE") ? 0 : 1;
    // -2 = automatic: from 2048 rows up (the fused kernel runs on the arithmetic core the GEMM uses, level.hip)
    fused_gm_min_rows = getenv("DVAE_UNFUSED") ? -1 : getenv("DVAE_FUSED_GM_MIN") ? atoi(getenv("DVAE_FUSED_GM_MIN")) : -2;
    // 0: hidden GEMM + gate kernel; 1: fused hidden GEMM + gates; 2: node sweeps run aggregation + hidden GEMM +
    // gates as ONE launch (level.hip GATHER kernels), everything else as 0
    fused_gru = (!getenv("DVAE_UNFUSED") && getenv("DVAE_FUSED_GRU")) ? std::max(1, atoi(getenv("DVAE_FUSED_GRU"))) : 0;
  }
};
LevelCfg g_level_cfg;

struct Carver {
  char* base; size_t off;
  float* f(size_t n) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off = (off + n * sizeof(float) + 255) & ~(size_t)255;
    return p;
  }
};

constexpr size_t GEMM_WS_BYTES = (size_t)96 << 20;

size_t part_floats_for(const dvae_dims& d) {
  size_t m = (size_t)(d.EHP + 4) * (2048 + 64);   // pair_bwd_a: one partial per wave + the group sums of its 2-stage reduction
  m = std::max(m, (size_t)(2 * d.EHP + 4) * (2048 + 80));   // pair_bwd_fused / pair_bwd_win: [db1 | dw2 | db2] per block
  m = std::max(m, (size_t)d.EHP * 512);
  return m;
}

void carve_sweep(Carver& c, SweepBufs& s, int prefix, int R, int E, int HP, int PP) {
  s.prefix = prefix;
  s.R = R;
  s.RH = prefix ? R + E : R;
  const size_t rh = (size_t)std::max(s.RH, 1), r = (size_t)std::max(R, 1);
  s.AGG = c.f(rh * HP);
  s.GH = c.f(rh * 3 * HP);
  s.GHN = c.f(rh * HP);
  s.DH = c.f(rh * HP);
  s.GMP = c.f(r * 2 * HP);
  s.GM = c.f(r * HP);
  s.XP = c.f(rh * 32);
  s.PX = c.f(r * PP);
  s.XT = reinterpret_cast<int*>(c.f(rh));
  if (prefix) {
    s.HV = c.f(rh * HP);
    s.Hn = c.f(r * HP);
    s.DHn = c.f(r * HP);
    s.PG = c.f(r * 3 * HP);
    s.DGM = c.f(r * HP);
  } else {
    s.Hn = c.f(r * HP);
    s.HV = s.Hn;
    s.DHn = nullptr;
    s.PG = s.DGM = nullptr;
  }
}

// scratch that a stream owns while it runs GEMMs / column reductions
struct Scratch { float* gemm_ws; float* part; size_t part_floats; };

struct EncWs {
  SweepBufs f, r;
  float *HVF, *ML, *DML, *DHVF;
  float *HVC, *DHVC;                // bidirectional chain: hv_t / dL/dhv_t, [(N+1)][B][HP]
  float *Q, *DQ, *DA;               // W01b.hb + b01 per reverse-sweep node row and its gradient; masked copy of dL/dhv_{t+1}
  Scratch sc[2];
  size_t bytes;
};

void carve_scratch(Carver& c, const dvae_dims& d, Scratch* sc) {
  for (int i = 0; i < 2; ++i) {
    sc[i].gemm_ws = c.f(GEMM_WS_BYTES / 4);
    sc[i].part_floats = part_floats_for(d);
    sc[i].part = c.f(sc[i].part_floats);
  }
}

EncWs enc_ws(const dvae_dims& d, int R, int E, void* base) {
  EncWs w;
  Carver c{(char*)base, 0};
  carve_sweep(c, w.f, 0, R, E, d.HP, pos_cols(d));
  const size_t B = d.B;
  w.ML = c.f(B * 2 * d.ZP);
  w.DML = c.f(B * 2 * d.ZP);
  w.DHVF = c.f(B * d.HP);
  if (d.bidir) {
    carve_sweep(c, w.r, 0, R, E, d.HP, pos_cols(d));
    w.HVC = c.f((size_t)(d.N + 1) * B * d.HP);
    w.DHVC = c.f((size_t)(d.N + 1) * B * d.HP);
    w.Q = c.f((size_t)std::max(R, 1) * d.HP);
    w.DQ = c.f((size_t)std::max(R, 1) * d.HP);
    w.DA = c.f((size_t)d.N * B * d.HP);
    w.HVF = w.HVC + (size_t)d.N * B * d.HP;
  } else {
    w.r = SweepBufs{};
    w.HVC = w.DHVC = w.Q = w.DQ = w.DA = nullptr;
    w.HVF = c.f(B * d.HP);
  }
  carve_scratch(c, d, w.sc);
  w.bytes = c.off;
  return w;
}

struct DecWs {
  SweepBufs s;
  float *ZPAD, *H0, *DH0, *DZP;
  float *A, *DA, *Bq, *DBq, *Cq, *DCq;
  float *GS, *DGS, *VH, *DVH, *TL, *DTL;
  float *TYPEV, *AT, *DTT, *DTV, *TG, *TGN, *XPT;   // type-shared prefix rows (sched.h TypeShare)
  Scratch sc[2];
  size_t bytes;
};

DecWs dec_ws(const dvae_dims& d, int R, int E, void* base) {
  DecWs w;
  Carver c{(char*)base, 0};
  carve_sweep(c, w.s, 1, R, E, d.HP, pos_cols(d));
  const size_t B = d.B, r = std::max(R, 1), rh = std::max(R + E, 1);
  w.ZPAD = c.f(B * d.ZP);
  w.H0 = c.f(B * d.HP);
  w.DH0 = c.f(B * d.HP);
  w.DZP = c.f(B * d.ZP);
  w.A = c.f(rh * d.EHP);
  w.DA = c.f(rh * d.EHP);
  w.Bq = c.f(r * d.EHP);
  w.DBq = c.f(r * d.EHP);
  w.Cq = d.mod ? c.f(B * d.EHP) : nullptr;
  w.DCq = d.mod ? c.f(B * d.EHP) : nullptr;
  w.GS = c.f(r * d.HP);
  w.DGS = c.f(r * d.HP);
  w.VH = c.f(r * d.VHP);
  w.DVH = c.f(r * d.VHP);
  w.TL = c.f(r * d.CP);
  w.DTL = c.f(r * d.CP);
  w.TYPEV = c.f((size_t)32 * d.HP);
  w.AT = c.f((size_t)32 * d.EHP);
  w.DTT = c.f((size_t)d.EHP * 32);
  w.DTV = c.f((size_t)32 * d.HP);
  w.TG = c.f((size_t)32 * 3 * d.HP);
  w.TGN = c.f((size_t)32 * d.HP);
  w.XPT = c.f(32 * 32);
  carve_scratch(c, d, w.sc);
  w.bytes = c.off;
  return w;
}

// ---- stream ordering --------------------------------------------------------
// Event pool for cross-stream edges.  hipStreamWaitEvent captures the event's
// state when it is enqueued, so an event may be re-recorded once every wait on
// it has been enqueued; the pool is deep enough that this always holds here.
struct Events {
  std::vector<hipEvent_t> ev;
  size_t next = 0;
  int take(hipEvent_t* out) {   // next event of the ring
    if (ev.size() < 256) {
      hipEvent_t e;
      DVAE_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ev.push_back(e);
      next = ev.size() - 1;
    } else {
      next = (next + 1) % ev.size();
    }
    *out = ev[next];
    return DVAE_OK;
  }
  int edge(hipStream_t from, hipStream_t to) {   // `to` continues after what `from` has queued so far
    if (from == to) return DVAE_OK;
    hipEvent_t e;
    DVAE_TRY(take(&e));
    DVAE_CHECK_HIP(hipEventRecord(e, from));
    DVAE_CHECK_HIP(hipStreamWaitEvent(to, e, 0));
    return DVAE_OK;
  }
};
thread_local Events g_events;

// ---- GEMM shorthands ---------------------------------------------------------
struct G {
  float* ws; hipStream_t st; int prio = 0; int bulk = 0;
  float* part = nullptr; size_t part_floats = 0;   // the stream's scratch for column sums (Scratch::part), when the caller has one
  // Y[M][N] (+)= X[M][K] W[N][K]^T + bias
  int nt(const float* X, int ldx, const float* W, int ldw, float* Y, int ldy, int M, int N, int K,
         const float* bias = nullptr, int relu = 0, int acc = 0, const int* rows = nullptr,
         GemmPartials* parts = nullptr) const {
    GemmDesc d{X, ldx, 0, W, ldw, 0, Y, ldy, M, N, K, bias, relu, acc, prio, bulk, rows, nullptr, parts};
    return gemm_f32(d, ws, GEMM_WS_BYTES, st);
  }
  // dX[M][N] (+)= dY[M][K] W[K][N]
  int nn(const float* dY, int ldy, const float* W, int ldw, float* dX, int ldx, int M, int N, int K,
         int acc = 0, const int* rows = nullptr, GemmPartials* parts = nullptr) const {
    GemmDesc d{dY, ldy, 0, W, ldw, 1, dX, ldx, M, N, K, nullptr, 0, acc, prio, bulk, rows, nullptr, parts};
    return gemm_f32(d, ws, GEMM_WS_BYTES, st);
  }
  // dW[M][N] (+)= dY[K][M]^T X[K][N]   (K = rows)
  // krows: the K summed rows are rows krows[0..K) of dY and X
  int tn(const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, int M, int N, int K,
         int acc = 0, const int* krows = nullptr) const {
    if (K <= 0) {
      if (acc) return DVAE_OK;
      if (hipMemset2DAsync(dW, (size_t)ldw * 4, 0, (size_t)N * 4, M, st) != hipSuccess) {
        dvae_set_error("memset2d failed");
        return DVAE_ERR_HIP;
      }
      return DVAE_OK;
    }
    GemmDesc d{dY, ldy, 1, X, ldx, 1, dW, ldw, M, N, K, nullptr, 0, acc, prio, bulk, nullptr, krows};
    return gemm_f32(d, ws, GEMM_WS_BYTES, st);
  }
};

struct Levels {
  int R, E, nlev, dir;
  const int* ls;   // node-row level starts [nlev+1]
  const int* es;   // edge-space level starts (forward only)
  const int* as;   // the same in scored edge rows (SchedView::a_rows)
  const int* cs = nullptr;   // shared encoder rows only: successor links before each level start
  int N;
  // scored edge rows before edge-space position e, which must be a level start
  // predecessor links of level L's nodes (exact in the forward direction, pro rata in the reverse one, whose
  // edge-space starts the schedule summary does not carry): bench.py's byte counts only
  double level_edges(int L) const {
    if (dir == 0) return es[L + 1] - es[L];
    return R > 0 ? (double)E * (ls[L + 1] - ls[L]) / R : 0.0;
  }
  int scored_before(int e) const {
    for (int L = 0; L <= N; ++L)
      if (es[L] == e) return as[L];
    return -1;
  }
};

Levels levels_of(const dvae_dims& d, const int* info, int dir) {
  Levels l;
  l.R = info[1]; l.E = info[2]; l.nlev = info[3 + dir]; l.dir = dir;
  l.ls = info + 8 + dir * (d.N + 1);
  l.es = info + 8 + 2 * (d.N + 1);
  l.as = info + 8 + 3 * (d.N + 1);
  l.N = d.N;
  return l;
}

// shared encoder rows (sched.h SchedView::enc): the encoder's forward-direction sweep on one row per distinct value
bool enc_shared(const dvae_dims& d, const int* info) { return info[DVAE_SCHED_MERGE_AT(d.N)] == 1; }
Levels levels_shared(const dvae_dims& d, const int* info, const Levels& lf) {
  const int* mi = info + DVAE_SCHED_MERGE_AT(d.N);
  Levels l = lf;          // same level count; es / as belong to prefix sweeps and are not read by a node sweep
  l.R = mi[1]; l.E = mi[2]; l.ls = mi + 4; l.dir = 2;
  l.cs = mi + 5 + d.N;
  return l;
}

int check_info(const dvae_dims& d, const int* info) {
  DVAE_REQUIRE(info, "null schedule info");
  if (info[0] != 0) {
    dvae_set_error("adjacency must be 0/1 and strictly lower-triangular within each graph's size");
    return DVAE_ERR_GRAPH;
  }
  DVAE_REQUIRE(info[1] >= 0 && info[1] <= d.B * d.N, "schedule info is stale or corrupt");
  return DVAE_OK;
}

// every node encoding is one-hot (dvae_sched_types) and there are levels >= 1: the decoder's
// k = 0 prefix rows of those levels are shared by node type (sched.h TypeShare)
bool type_shared(const dvae_dims& d, const int* info, const Levels& lv) {
  (void)d;
  return g_level_cfg.type_share && info[5] == 1 && lv.nlev > 1 && lv.R > lv.ls[1];
}

// node rows that need a W1b product (Bq = W1b.hd + b1): all of them, minus the last level when it
// holds exactly the last node of every graph (info[6]) -- a last node is never a candidate
// predecessor (dvae_model.py:210 walks v_j < index only), so its Bq row is never read
int bq_rows(const int* info, const Levels& lv) {
  if (g_level_cfg.type_share && info[6] == 1 && lv.nlev > 0) return lv.ls[lv.nlev - 1];
  return lv.R;
}

// add_edge's first layer A = W1a.hv is read on the prefixes some pair is scored on only
// (SchedView::a_rows): the other edge rows -- and node 0's row, which has no pair at all, when
// level 0 holds nothing else (info[7]) -- take no W1a product, forward or backward.
// DVAE_NO_ROW_SKIP=1: every row (A-B tests).
bool scored_rows_only() {
  static const bool off = getenv("DVAE_NO_ROW_SKIP") != nullptr;
  return g_level_cfg.type_share && !off;
}
bool skip_level0(const int* info) { return scored_rows_only() && info[7] == 1; }

// prefix sweeps in mod None sum the GRU's hidden product from one product per node (kernels.h SweepBufs::PG)
bool node_hidden(const dvae_dims& d, const SweepBufs& b) {
  return g_level_cfg.node_hidden && b.prefix && d.mod == 0 && g_level_cfg.fused_gru != 1;
}
// ... and then the backward pass rebuilds the gates of every prefix row from GM / PG instead of reading five saved tables
// per row (kernels.hip prefix_gru_bwd_kernel); the forward pass writes the states only.  DVAE_NO_RECOMPUTE=1: A-B tests
// Per level, from 1024 node rows up: below that the row-parallel gru_bwd_kernel fills the card better than a walk per node
// (B = 512: 2.069 ms against 2.090 with recomputation everywhere; headline 8.28 -> 8.20, c3 101.0 -> 98.3, same box).
int g_recompute_min_rows = getenv("DVAE_NO_RECOMPUTE") ? -1 : getenv("DVAE_RECOMPUTE_MIN") ? atoi(getenv("DVAE_RECOMPUTE_MIN")) : 1024;
bool gates_recomputed(const dvae_dims& d, const SweepBufs& b, int level_rows) {
  static const bool off = getenv("DVAE_PG_SPLIT") != nullptr;
  return !off && g_recompute_min_rows >= 0 && node_hidden(d, b) && level_rows >= g_recompute_min_rows;
}

// SURVEY.md 8(d), pair scoring forward: every A row (E + N prefix rows) and every Bq row (N) read once, 16 H bytes each
// (24 H in bn mode, whose add_edge is 6H wide), one logit written per pair; backward = 2 x.  Pairs counted as for
// equal-size graphs (R / B nodes each), a lower bound on ragged batches.
double pair_bytes(const dvae_dims& d, const Levels& lv) {
  const double n = d.B > 0 ? (double)lv.R / d.B : 0.0;
  return 4.0 * d.EH * (lv.E + 2.0 * lv.R) + 4.0 * lv.R * (n - 1.0) / 2.0;
}

struct SweepW {
  GruW gru; const float* gm;   // gm: Wgm|posb (None) or WgmT|bgm (bn)
};

// called after the GRU states of rows [p0,p1) and of nodes [ns,ne) are final
using LevelHook = std::function<int(int p0, int p1, int ns, int ne)>;

// level sweep forward (dvae_model.py:63-87 / :93-116 / the final-state part of :196-249).
// Row classes of a prefix (decoder) sweep, see sched.h: node rows [ls[L], ls[L+1]) carry
// the k = 0 states (h_in = 0, or h0 for node 0 -> only level 0 needs the hidden GEMM);
// edge rows R + [es[L], es[L+1]) carry the running predecessor sums.
int sweep_fwd(const dvae_dims& d, const SchedDir& S, const Levels& lv, SweepBufs& b, const SweepW& w,
              const float* x, const float* h0, const G& g, const LevelHook& hook = nullptr,
              bool k0_shared = false) {
  const int HP = d.HP, R = lv.R;
  hipStream_t st = g.st;
  if (R <= 0) return DVAE_OK;
  const int z1 = lv.nlev > 0 ? lv.ls[1] : R;   // end of level 0
  // Fused level steps (level.hip).  The gate/map step is fused by default from 2048 rows up
  // (one launch, 25 % less time than GEMM + epilogue kernel; below that the fused tile grid
  // is too small to fill the chip).  The fused GRU step is parity-green but measured slower
  // than "64x128-tile GEMM + gru_fwd_kernel" on MI355X (a workgroup must hold all three
  // gates of its hidden units, which leaves 2 instead of 3 waves per SIMD): opt-in through
  // dvae_level_tune / DVAE_FUSED_GRU=1.
  const bool fused_gru = g_level_cfg.fused_gru == 1;
  const bool gather_gru = g_level_cfg.fused_gru == 2 && !b.prefix && d.mod == 0;
  const int fused_gm_min = g_level_cfg.fused_gm_min_rows != -2 ? g_level_cfg.fused_gm_min_rows : 2048;
  static const bool fold_partials = getenv("DVAE_NO_FOLD") == nullptr;   // A-B tests: combine launch instead
  const int small_rows = g_level_cfg.small_rows;
  auto gru_rows = [&](int p0, int p1) -> int {
    if (fused_gru) return k_level_gru_fwd(S, p0, p1, b, x, d.N, d.C, w.gru, HP, g.prio, st);
    if (p1 - p0 <= small_rows) return k_level_gru_fwd_small(S, p0, p1, b, x, d.N, d.C, w.gru, HP, g.prio, st);
    GemmPartials parts{nullptr, 1, 0};   // a small level's split-K slabs go to the gate kernel as they are
    DVAE_TRY(g.nt(b.AGG + (size_t)p0 * HP, HP, w.gru.Whh, HP, b.GH + (size_t)p0 * 3 * HP, 3 * HP, p1 - p0, 3 * HP, HP,
                  nullptr, 0, 0, nullptr, fold_partials ? &parts : nullptr));
    return k_gru_fwd(S, p0, p1, 0, b, x, d.N, d.C, w.gru, HP, st, &parts);
  };
  if (d.mod == 1) {
    // gate/map depends on node types only: no inter-node dependency, one "level"
    DVAE_TRY(k_gm_bn_fwd(S, R, b, x, d.N, d.C, w.gm, w.gm + (size_t)d.C * 2 * HP, HP, st));
    {
      KernelProf kp(6, 4.0 * d.H * (lv.E + 2.0 * R), st);
      DVAE_TRY(k_agg_fwd(S, 0, R, b, h0, HP, st));
    }
    if (!b.prefix) {
      DVAE_TRY(gru_rows(0, R));
    } else {
      DVAE_TRY(gru_rows(0, z1));
      if (!k0_shared) DVAE_TRY(k_gru_fwd(S, z1, R, 1, b, x, d.N, d.C, w.gru, HP, st));
      DVAE_TRY(gru_rows(R, b.RH));
    }
    if (hook) DVAE_TRY(hook(0, b.RH, 0, R));
    return DVAE_OK;
  }
  // rows whose h_in is 0 depend on nothing but the node type: all levels in one launch
  // (k0_shared: those rows are served per type, sched.h TypeShare -- nothing to compute per row)
  if (b.prefix && !k0_shared) {
    DVAE_TRY(k_gru_fwd(S, z1, R, 1, b, x, d.N, d.C, w.gru, HP, st));
    if (hook) DVAE_TRY(hook(z1, R, 0, 0));
  }
  const bool pg = node_hidden(d, b);
  for (int L = 0; L < lv.nlev; ++L) {
    const int ns = lv.ls[L], ne = lv.ls[L + 1];
    int p0 = ns, p1 = ne;
    static const bool split_pg = getenv("DVAE_PG_SPLIT") != nullptr;   // lab: aggregation and gates as two launches
    if (gather_gru && L > 0) {
      // aggregation + hidden GEMM + GRU gates of this level's node rows in one launch
      DVAE_TRY(k_level_agg_gru_fwd(S, ns, ne, b, x, d.N, d.C, w.gru, HP, g.prio, st));
    } else if (!(pg && L > 0 && !split_pg)) {
      KernelProf kp(6, 4.0 * d.H * (lv.level_edges(L) + 2.0 * (ne - ns)), st);
      DVAE_TRY(k_agg_fwd(S, ns, ne, b, h0, HP, st, pg ? 1 : 0));
    }
    if (gather_gru && L > 0) {
      // (done above)
    } else if (L == 0) {
      if (h0) DVAE_TRY(gru_rows(ns, ne));
      else DVAE_TRY(k_gru_fwd(S, ns, ne, 1, b, x, d.N, d.C, w.gru, HP, st));
    } else {
      if (b.prefix) { p0 = R + lv.es[L]; p1 = R + lv.es[L + 1]; }
      if (pg && !split_pg) {                                                                     // aggregation + gates
        KernelProf kp(6, 4.0 * d.H * (lv.level_edges(L) + 2.0 * (p1 - p0)), st);
        DVAE_TRY(k_prefix_gru_fwd(S, ns, ne, b, x, d.N, d.C, w.gru, HP, st, gates_recomputed(d, b, ne - ns) ? 0 : 1));
      }
      else if (pg) DVAE_TRY(k_gru_fwd(S, p0, p1, 0, b, x, d.N, d.C, w.gru, HP, st));              // GH rows summed by k_agg_fwd
      else DVAE_TRY(gru_rows(p0, p1));
    }
    if (hook) DVAE_TRY(hook(p0, p1, ns, ne));
    if (L + 1 < lv.nlev) {   // nodes of the last level have no successors
      if (fused_gm_min >= 0 && ne - ns >= fused_gm_min) {
        DVAE_TRY(k_level_gm_fwd(S, ns, ne, b, w.gm, w.gm + (size_t)2 * HP * HP, HP, g.prio, st));
      } else if (ne - ns <= small_rows) {
        DVAE_TRY(k_level_gm_fwd_small(S, ns, ne, b, w.gm, w.gm + (size_t)2 * HP * HP, HP, g.prio, st));
      } else {
        // a small level's product is split over K: its slabs go to the gate/map kernel as they are
        GemmPartials parts{nullptr, 1, 0};
        DVAE_TRY(g.nt(b.Hn + (size_t)ns * HP, HP, w.gm, HP, b.GMP + (size_t)ns * 2 * HP, 2 * HP, ne - ns, 2 * HP,
                      HP, nullptr, 0, 0, nullptr, fold_partials ? &parts : nullptr));
        DVAE_TRY(k_gm_fwd(S, ns, ne, b, w.gm + (size_t)2 * HP * HP, HP, st, &parts));
      }
      // the hidden product of this level's gated states, once per node
      if (pg && ne - ns <= small_rows)
        DVAE_TRY(k_small_gemm(b.GM + (size_t)ns * HP, HP, w.gru.Whh, HP, 0, b.PG + (size_t)ns * 3 * HP, 3 * HP, ne - ns, 3, HP,
                              HP, 0, g.prio, st));
      else if (pg)
        DVAE_TRY(g.nt(b.GM + (size_t)ns * HP, HP, w.gru.Whh, HP, b.PG + (size_t)ns * 3 * HP, 3 * HP, ne - ns, 3 * HP, HP));
    }
  }
  return DVAE_OK;
}

// reverse-level sweep: DH holds dL/dHV on entry, dL/dAGG on exit
int sweep_bwd(const dvae_dims& d, const SchedDir& S, const Levels& lv, SweepBufs& b, const SweepW& w, const float* x,
              bool seeded, const G& g, bool k0_shared = false,
              const std::function<int(int)>& after_level = nullptr) {
  const int HP = d.HP, R = lv.R;
  hipStream_t st = g.st;
  if (R <= 0) return DVAE_OK;
  const int z1 = lv.nlev > 0 ? lv.ls[1] : R;
  const int small_rows = g_level_cfg.small_rows;
  // dX[M][HP] += dY[M][K] . W[K][HP] on a small level: one launch, no split-K slabs
  auto small_nn = [&](const float* dY, int K, const float* W, float* dX, int M) {
    return k_small_gemm(dY, K, W, HP, 1, dX, HP, M, 1, HP, K, 1, g.prio, st);
  };
  auto dagg_gemm = [&](int p0, int p1) {   // dAGG += dgh . Whh
    if (p1 - p0 <= small_rows) return small_nn(b.GH + (size_t)p0 * 3 * HP, 3 * HP, w.gru.Whh, b.DH + (size_t)p0 * HP, p1 - p0);
    return g.nn(b.GH + (size_t)p0 * 3 * HP, 3 * HP, w.gru.Whh, HP, b.DH + (size_t)p0 * HP, HP, p1 - p0, HP, 3 * HP, 1);
  };
  if (d.mod == 1) {
    if (b.prefix && k0_shared) {   // the k = 0 rows [z1,R) are handled per type by the caller
      DVAE_TRY(k_gru_bwd(S, 0, z1, b, HP, st));
      DVAE_TRY(k_gru_bwd(S, R, b.RH, b, HP, st));
    } else {
      DVAE_TRY(k_gru_bwd(S, 0, b.RH, b, HP, st));
    }
    if (!b.prefix) {
      DVAE_TRY(dagg_gemm(0, R));
    } else {
      if (seeded) DVAE_TRY(dagg_gemm(0, z1));
      DVAE_TRY(dagg_gemm(R, b.RH));
    }
    {
      KernelProf kp(7, 8.0 * d.H * (lv.E + 2.0 * R), st);
      if (lv.cs) {   // shared encoder rows: level by level, the first levels' long successor lists split over lanes
        for (int L = 0; L < lv.nlev; ++L) {
          const int ns = lv.ls[L], ne = lv.ls[L + 1];
          DVAE_TRY(k_dgm_bwd(S, ns, ne, b, HP, st, ne > ns ? (double)(lv.cs[L + 1] - lv.cs[L]) / (ne - ns) : 0.0));
        }
      } else {
        DVAE_TRY(k_dgm_bwd(S, 0, R, b, HP, st));
      }
    }
    return DVAE_OK;
  }
  float* dtarget = b.prefix ? b.DHn : b.DH;
  const bool pg = node_hidden(d, b);
  static const bool fold_partials_bwd = getenv("DVAE_NO_FOLD") == nullptr;   // A-B tests: combine launch instead
  for (int L = lv.nlev - 1; L >= 0; --L) {
    const int ns = lv.ls[L], ne = lv.ls[L + 1];
    GemmPartials gparts{nullptr, 1, 0};   // gate/map input gradient of this level, when left as split-K slabs
    bool gru_done = false;                // this level's GRUCell backward ran as the epilogue of that product
    if (L + 1 < lv.nlev) {
      if (pg) {
        // dL/dGM[u] = (direct gradients of the rows that hold u) + (their dgh, summed).Whh: the hidden
        // product's input gradient once per node instead of once per prefix row
        {
          // the consumers of level L's nodes are the predecessor links of the levels above
          KernelProf kp(7, 8.0 * d.H * (lv.level_edges(L + 1) + 2.0 * (ne - ns)), st);
          DVAE_TRY(k_dgm_gather(S, ns, ne, b, HP, st));
        }
        if (ne - ns <= small_rows) {
          // hidden part of dL/dGM and the gate/map backward in one launch (level.hip small_kernel)
          DVAE_TRY(k_level_gm_bwd_small(ns, ne, b, w.gru.Whh, HP, g.prio, st));
        } else {
          GemmPartials parts{nullptr, 1, 0};   // a mid-sized level's split-K slabs are added by the consumer
          DVAE_TRY(g.nn(b.PG + (size_t)ns * 3 * HP, 3 * HP, w.gru.Whh, HP, b.DGM + (size_t)ns * HP, HP, ne - ns, HP,
                        3 * HP, 1, nullptr, fold_partials_bwd ? &parts : nullptr));
          DVAE_TRY(k_gm_bwd_apply(ns, ne, b, HP, st, &parts));
        }
      } else {
        KernelProf kp(7, 8.0 * d.H * (lv.level_edges(L + 1) + 2.0 * (ne - ns)), st);
        DVAE_TRY(k_dgm_bwd(S, ns, ne, b, HP, st, lv.cs && ne > ns ? (double)(lv.cs[L + 1] - lv.cs[L]) / (ne - ns) : 0.0));
      }
      if (ne - ns <= small_rows && !b.prefix) {
        // node sweeps: the gate/map input gradient lands on the rows whose GRU step comes next -- product and
        // GRUCell backward in one launch (level.hip small_kernel)
        DVAE_TRY(k_level_gru_bwd_small(ns, ne, b, w.gm, HP, g.prio, st));
        gru_done = true;
      } else if (ne - ns <= small_rows) {
        DVAE_TRY(small_nn(b.GMP + (size_t)ns * 2 * HP, 2 * HP, w.gm, dtarget + (size_t)ns * HP, ne - ns));
      } else {
        DVAE_TRY(g.nn(b.GMP + (size_t)ns * 2 * HP, 2 * HP, w.gm, HP, dtarget + (size_t)ns * HP, HP, ne - ns, HP,
                      2 * HP, 1, nullptr, fold_partials_bwd ? &gparts : nullptr));
      }
    }
    if (L == 0) {
      if (!gru_done) DVAE_TRY(k_gru_bwd(S, ns, ne, b, HP, st, &gparts, ns));
      if (seeded) DVAE_TRY(dagg_gemm(ns, ne));      // only node 0's h_in = h0 needs it
    } else {
      const int p0 = b.prefix ? R + lv.es[L] : ns, p1 = b.prefix ? R + lv.es[L + 1] : ne;
      if (gates_recomputed(d, b, ne - ns)) DVAE_TRY(k_prefix_gru_bwd(S, ns, ne, b, x, d.N, d.C, w.gru, HP, st, &gparts, ns));
      else if (!gru_done) DVAE_TRY(k_gru_bwd(S, p0, p1, b, HP, st, &gparts, ns));
      if (!pg) DVAE_TRY(dagg_gemm(p0, p1));
    }
    if (after_level) DVAE_TRY(after_level(L));   // levels >= L are final from here on
  }
  // k = 0 rows of the later levels feed nothing backwards except the weight gradients
  if (b.prefix && !k0_shared) DVAE_TRY(k_gru_bwd(S, z1, R, b, HP, st));
  return DVAE_OK;
}

// weight gradients of one sweep, reduced over the rows of levels [La, Lb).
// Ggru -> dWhh [3HP][HP]; Xgru / Xgm -> the gradient-only regions of layout.h
// first: this call sets the outputs (and builds the one-hot tables), later calls accumulate.
// The chunk with La = 0 also takes what belongs to no level: the k = 0 rows of a prefix sweep.
// tg: those rows, [z1,R), are shared by type -- their GRU gradients come as one row per type
// (k_type_gru_bwd) instead of per-row GH/GHN entries
struct TypeGrad { const float *TG, *TGN, *XPT; };
struct WgradOut { float *Ggru, *Ggm, *Xgru, *Xgm; };
// onehot: every node encoding of the batch is one-hot (info[5]): the products against the one-hot tables XP / PX are
// column sums per type / per position (k_typed_colsum: one streaming pass at 3.5-5 TB/s instead of a [3HP x 32] tile grid
// that walks the table at 1.3) and the tables themselves are never built
int sweep_wgrad(const dvae_dims& d, const SchedDir& S, const Levels& lv, SweepBufs& b, const float* x,
                const WgradOut& o, bool seeded, const G& g, const TypeGrad* tg, int La, int Lb, bool first, bool onehot,
                int which = 3) {
  // which: 1 = the GRU cell's gradients, 2 = the gate/map network's (mod None: the two halves share no table, so the tail
  // of a step can run them on two streams), 3 = both
  const int HP = d.HP, PP = pos_cols(d), R = lv.R;
  hipStream_t st = g.st;
  const int RH = R > 0 ? b.RH : 0;
  const int z1 = (R > 0 && lv.nlev > 0) ? lv.ls[1] : 0;
  const int Rg = (d.mod == 0 && lv.nlev > 0) ? lv.ls[lv.nlev - 1] : 0;   // nodes that have a gate/map output
  const int acc = first ? 0 : 1;
  const bool low = La == 0;
  if (R <= 0) { La = 0; Lb = 0; }
  static const bool typed_off = getenv("DVAE_NO_TYPED_WGRAD") != nullptr;   // A-B tests: the TN products
  const bool typed = onehot && !typed_off && g.part != nullptr;
  const bool typed_pos = typed && d.mod == 0 && d.npos < 32 && d.npos < PP;   // position one-hot: at most 31 positions
  if (first && (which & 1)) {
    if (typed) DVAE_TRY(k_build_xt(S, b.prefix, RH, x, d.N, d.C, b.XT, st));
    else DVAE_TRY(k_build_xp(S, b.prefix, RH, x, d.N, d.C, b.XP, st));
  }
  if (first && (which & 2)) {
    if (d.mod == 0 && !typed_pos) DVAE_TRY(k_build_px(S, Rg, d.npos, PP, b.PX, st));
    else if (d.mod != 0 && !typed) DVAE_TRY(k_build_xp(S, 0, R, x, d.N, d.C, b.PX, st));
  }
  float* Xn = o.Xgru + (size_t)3 * HP * 32;
  auto gru_rows = [&](int r0, int r1, int a) -> int {   // dW_ih, db_ih, db_hh = (dgh | dn_pre)^T . (x | 1)
    if (r1 <= r0 && a) return DVAE_OK;
    if (typed) {
      DVAE_TRY(k_typed_colsum(b.GH + (size_t)r0 * 3 * HP, 3 * HP, r1 - r0, 3 * HP, b.XT + r0, d.C, o.Xgru, g.part, g.part_floats,
                              st, 32, a));
      return k_typed_colsum(b.GHN + (size_t)r0 * HP, HP, r1 - r0, HP, b.XT + r0, d.C, Xn, g.part, g.part_floats, st, 32, a);
    }
    DVAE_TRY(g.tn(b.GH + (size_t)r0 * 3 * HP, 3 * HP, b.XP + (size_t)r0 * 32, 32, o.Xgru, 32, 3 * HP, 32, r1 - r0, a));
    return g.tn(b.GHN + (size_t)r0 * HP, HP, b.XP + (size_t)r0 * 32, 32, Xn, 32, HP, 32, r1 - r0, a);
  };
  const int n0 = R > 0 ? lv.ls[La] : 0, n1 = R > 0 ? lv.ls[Lb] : 0;   // node rows of the chunk
  if (!(which & 1)) {
    // (the GRU half runs elsewhere)
  } else if (!b.prefix) {
    // dWhh = dgh^T . h_in over the rows whose h_in is not identically zero (level 0 of a node sweep has h_in = 0)
    const int r0 = d.mod == 0 ? std::max(n0, z1) : n0;
    DVAE_TRY(g.tn(b.GH + (size_t)r0 * 3 * HP, 3 * HP, b.AGG + (size_t)r0 * HP, HP, o.Ggru, HP, 3 * HP, HP, n1 - r0, acc));
    DVAE_TRY(gru_rows(n0, n1, acc));
  } else {
    const int e0 = R > 0 ? R + lv.es[La] : 0, e1 = R > 0 ? R + lv.es[Lb] : 0;   // edge rows of the chunk
    if (node_hidden(d, b)) {
      // dWhh = sum over nodes of (dgh summed over the rows that hold the node)^T . GM[node]
      const int r1 = std::min(n1, Rg);
      DVAE_TRY(g.tn(b.PG + (size_t)n0 * 3 * HP, 3 * HP, b.GM + (size_t)n0 * HP, HP, o.Ggru, HP, 3 * HP, HP, r1 - n0, acc));
    } else {
      DVAE_TRY(g.tn(b.GH + (size_t)e0 * 3 * HP, 3 * HP, b.AGG + (size_t)e0 * HP, HP, o.Ggru, HP, 3 * HP, HP, e1 - e0, acc));
    }
    DVAE_TRY(gru_rows(e0, e1, acc));
    if (low) {
      if (seeded && z1 > 0) DVAE_TRY(g.tn(b.GH, 3 * HP, b.AGG, HP, o.Ggru, HP, 3 * HP, HP, z1, 1));
      if (tg) {
        DVAE_TRY(gru_rows(0, z1, 1));
        DVAE_TRY(g.tn(tg->TG, 3 * HP, tg->XPT, 32, o.Xgru, 32, 3 * HP, 32, 32, 1));
        DVAE_TRY(g.tn(tg->TGN, HP, tg->XPT, 32, Xn, 32, HP, 32, 32, 1));
      } else {
        DVAE_TRY(gru_rows(0, R, 1));
      }
    }
  }
  if (!(which & 2)) return DVAE_OK;
  if (d.mod == 0) {
    const int r1 = std::min(n1, Rg);
    DVAE_TRY(g.tn(b.GMP + (size_t)n0 * 2 * HP, 2 * HP, b.Hn + (size_t)n0 * HP, HP, o.Ggm, HP, 2 * HP, HP, r1 - n0, acc));
    if (typed_pos)   // the rows' positions are the schedule's row_node
      DVAE_TRY(k_typed_colsum(b.GMP + (size_t)n0 * 2 * HP, 2 * HP, r1 - n0, 2 * HP, S.row_node + n0, d.npos, o.Xgm, g.part,
                              g.part_floats, st, PP, acc));
    else
      DVAE_TRY(g.tn(b.GMP + (size_t)n0 * 2 * HP, 2 * HP, b.PX + (size_t)n0 * PP, PP, o.Xgm, PP, 2 * HP, PP, r1 - n0, acc));
  } else if (typed) {
    DVAE_TRY(k_typed_colsum(b.GMP, 2 * HP, R, 2 * HP, b.XT, d.C, o.Xgm, g.part, g.part_floats, st, 32, acc));
  } else {
    DVAE_TRY(g.tn(b.GMP, 2 * HP, b.PX, 32, o.Xgm, 32, 2 * HP, 32, R, acc));
  }
  return DVAE_OK;
}

// The level from which the weight-gradient products of a sweep are issued early: once the
// backward sweep has passed level L0, the rows of levels >= L0 are final and their products run
// on the side stream beside the rest of the sweep (which alone leaves the matrix cores idle
// between its per-level launches); the lower levels follow after the sweep.  One early chunk
// only: per-level products were measured slower (small launches).  0 = no early chunk.
// DVAE_WGRAD_SPLIT=<fraction of rows in the early chunk, 0 = off> (default 0.35).
struct WgradSplitCfg {
  double frac = getenv("DVAE_WGRAD_SPLIT") ? atof(getenv("DVAE_WGRAD_SPLIT")) : 0.35;
  int min_rows = 16384;   // small batches: the split only adds launches
};
WgradSplitCfg g_wgrad_split;

int wgrad_split_level(const dvae_dims& d, const Levels& lv, bool prefix, bool two_streams) {
  const double frac = g_wgrad_split.frac;
  if (!two_streams || frac <= 0.0 || d.mod != 0 || lv.nlev < 3) return 0;
  const int* cum = prefix ? lv.es : lv.ls;
  const int T = cum[lv.nlev];
  if (T < g_wgrad_split.min_rows) return 0;
  int L0 = 0;
  for (int L = 1; L < lv.nlev; ++L)
    if ((double)(T - cum[L]) >= frac * T) L0 = L;
  return L0;
}

// backward sweep on the chain + its weight-gradient products on the side stream
int sweep_bwd_wgrad(const dvae_dims& d, const SchedDir& S, const Levels& lv, SweepBufs& b, const SweepW& w,
                    const float* x, const WgradOut& o, bool seeded, const G& g, const G& gs, Events& ev, bool onehot,
                    bool k0_shared = false, const TypeGrad* tg = nullptr, const G* tail_chain = nullptr) {
  const int L0 = wgrad_split_level(d, lv, b.prefix, g.st != gs.st);
  bool early = false;
  std::function<int(int)> after = [&](int L) -> int {
    if (L != L0 || L0 <= 0) return DVAE_OK;
    DVAE_TRY(ev.edge(g.st, gs.st));
    early = true;
    return sweep_wgrad(d, S, lv, b, x, o, seeded, gs, tg, L0, lv.nlev, true, onehot);
  };
  DVAE_TRY(sweep_bwd(d, S, lv, b, w, x, seeded, g, k0_shared, after));
  DVAE_TRY(ev.edge(g.st, gs.st));
  // tail_chain (the LAST sweep of a step, node sweeps, mod None): nothing is left for the chain but to wait for these
  // products, so it takes the gate/map half itself; the side stream then waits for it (G is complete when the side
  // stream has drained, as documented).  For small batches, whose products cannot fill the card one at a time: B = 512 -1.4 %.
  static const bool split_tail = getenv("DVAE_NO_TAIL_SPLIT") == nullptr;
  if (tail_chain && split_tail && tail_chain->st != gs.st && d.mod == 0 && !b.prefix) {
    const int Lend = early ? L0 : lv.nlev;
    // the chain's half ACCUMULATES into what the early chunk set on the side stream: it must see those writes (the event
    // is recorded before the side stream's own half is queued, so the chain waits for nothing newer)
    if (early) DVAE_TRY(ev.edge(gs.st, tail_chain->st));
    DVAE_TRY(sweep_wgrad(d, S, lv, b, x, o, seeded, gs, tg, 0, Lend, !early, onehot, 1));
    DVAE_TRY(sweep_wgrad(d, S, lv, b, x, o, seeded, *tail_chain, tg, 0, Lend, !early, onehot, 2));
    return ev.edge(tail_chain->st, gs.st);
  }
  return sweep_wgrad(d, S, lv, b, x, o, seeded, gs, tg, 0, early ? L0 : lv.nlev, !early, onehot);
}

}  // namespace

// ===========================================================================
// encoder
// ===========================================================================
extern "C" size_t dvae_enc_ws_bytes(const dvae_dims* d, int R, int E) { return enc_ws(*d, R, E, nullptr).bytes; }

extern "C" int dvae_encoder_fwd(const dvae_dims* dp, const float* x, const int* sizes, const void* sched,
                                const int* info, const float* P, void* ws, float* hv, float* mu,
                                float* logvar, float* states, float* states_rev, void* stream) {
  DVAE_REQUIRE(dp && x && sched && P && ws, "encoder_fwd: null pointer");
  const dvae_dims& d = *dp;
  DVAE_TRY(check_info(d, info));
  hipStream_t st = (hipStream_t)stream;
  const SchedView sv = sched_view(const_cast<void*>(sched), d.B, d.N);
  const Levels lf = levels_of(d, info, 0);
  EncWs w = enc_ws(d, lf.R, lf.E, ws);
  const EncLayout L = enc_layout(d);
  const G g{w.sc[0].gemm_ws, st, 1};
  const int HP = d.HP, B = d.B;

  // the forward-direction sweep runs on one row per distinct node value when the schedule has them
  const bool shared = enc_shared(d, info);
  const SchedDir& Sf = shared ? sv.enc : sv.dir[0];
  const Levels lfs = shared ? levels_shared(d, info, lf) : lf;
  if (shared) w.f.R = w.f.RH = lfs.R;   // the tables were carved for one row per node; the sweep's rows end here
  SweepW wf{gru_at(P + L.gru_f, HP, d.C), P + L.gm_f};
  DVAE_TRY(sweep_fwd(d, Sf, lfs, w.f, wf, x, nullptr, g));
  if (d.bidir) {
    const Levels lr = levels_of(d, info, 1);
    SweepW wr{gru_at(P + L.gru_b, HP, d.C), P + L.gm_b};
    DVAE_TRY(sweep_fwd(d, sv.dir[1], lr, w.r, wr, x, nullptr, g));
    // lin01 chain, applied once per reverse-sweep step (dvae_model.py:119): hv_{t+1} = W01.[hv_t ; hb] + b01.  The hb
    // half does not depend on the chain: Q = W01b.hb + b01 for every reverse-sweep node row in ONE product, and a step is
    // one launch (hv_t.W01a^T + Q's row for the graphs that still have a step t, hv_t for the others)
    DVAE_TRY(k_take_last_fwd(Sf, sizes, B, d.N, w.f.Hn, w.HVC, HP, st));
    DVAE_TRY(g.nt(w.r.Hn, HP, P + L.W01 + HP, 2 * HP, w.Q, HP, lr.R, HP, HP, P + L.b01));
    for (int t = 0; t < d.N; ++t) {
      float* hv_t = w.HVC + (size_t)t * B * HP;
      DVAE_TRY(k_chain_fwd_step(sv.dir[1], sizes, B, d.N, t, hv_t, P + L.W01, 2 * HP, w.Q, hv_t + (size_t)B * HP, HP, g.prio, st));
    }
  } else {
    DVAE_TRY(k_take_last_fwd(Sf, sizes, B, d.N, w.f.Hn, w.HVF, HP, st));
  }
  // mu | logvar heads (dvae_model.py:121-122)
  DVAE_TRY(g.nt(w.HVF, HP, P + L.Wml, HP, w.ML, 2 * d.ZP, B, 2 * d.ZP, HP, P + L.bml));
  if (hv) DVAE_TRY(k_unpad_rows(w.HVF, HP, B, d.H, hv, st));
  if (mu && logvar) {
    DVAE_TRY(k_unpad_rows2(w.ML, 2 * d.ZP, d.ZP, B, d.Z, mu, logvar, st));
  } else {
    if (mu) DVAE_TRY(k_unpad_rows(w.ML, 2 * d.ZP, B, d.Z, mu, st));
    if (logvar) DVAE_TRY(k_unpad_rows(w.ML + d.ZP, 2 * d.ZP, B, d.Z, logvar, st));
  }
  if (states && shared) {
    DVAE_TRY(k_nodes_to_dense(Sf, B, d.N, w.f.Hn, HP, states, d.H, st));
  } else if (states) {
    DVAE_CHECK_HIP(hipMemsetAsync(states, 0, (size_t)B * d.N * d.H * 4, st));
    DVAE_TRY(k_rows_to_dense(sv.dir[0], lf.R, w.f.Hn, HP, states, d.N, d.H, st));
  }
  if (states_rev && d.bidir) {
    DVAE_CHECK_HIP(hipMemsetAsync(states_rev, 0, (size_t)B * d.N * d.H * 4, st));
    DVAE_TRY(k_rows_to_dense(sv.dir[1], lf.R, w.r.Hn, HP, states_rev, d.N, d.H, st));
  }
  return DVAE_OK;
}

extern "C" int dvae_encoder_bwd(const dvae_dims* dp, const float* x, const int* sizes, const void* sched,
                                const int* info, const float* P, void* ws, const float* d_hv,
                                const float* d_mu, const float* d_logvar, float* Gp, void* stream,
                                void* side_stream) {
  DVAE_REQUIRE(dp && x && sched && P && ws && Gp, "encoder_bwd: null pointer");
  const dvae_dims& d = *dp;
  DVAE_TRY(check_info(d, info));
  hipStream_t st = (hipStream_t)stream;
  hipStream_t sd = side_stream ? (hipStream_t)side_stream : st;
  const SchedView sv = sched_view(const_cast<void*>(sched), d.B, d.N);
  const Levels lf = levels_of(d, info, 0);
  EncWs w = enc_ws(d, lf.R, lf.E, ws);
  const EncLayout L = enc_layout(d);
  const G g{w.sc[0].gemm_ws, st, sd != st ? 1 : 0};
  const G gs{w.sc[1].gemm_ws, sd, 0, sd != st ? 1 : 0, w.sc[1].part, w.sc[1].part_floats};       // weight-gradient products
  const Scratch& ssc = w.sc[1];
  Events& ev = g_events;
  const int HP = d.HP, B = d.B, ZP = d.ZP;

  // heads
  DVAE_TRY(k_pad_rows2(d_mu, d_logvar, B, d.Z, w.DML, 2 * ZP, ZP, st));
  DVAE_TRY(ev.edge(st, sd));
  DVAE_TRY(gs.tn(w.DML, 2 * ZP, w.HVF, HP, Gp + L.Wml, HP, 2 * ZP, HP, B));
  DVAE_TRY(k_colsum(w.DML, 2 * ZP, B, 2 * ZP, Gp + L.bml, ssc.part, ssc.part_floats, sd));
  DVAE_TRY(g.nn(w.DML, 2 * ZP, P + L.Wml, HP, w.DHVF, HP, B, HP, 2 * ZP));
  if (d_hv) DVAE_TRY(k_add_rows(w.DHVF, HP, d_hv, d.H, B, d.H, st));

  DVAE_TRY(k_zero(w.f.DH, (size_t)std::max(lf.R, 1) * HP * 4, st));
  SweepW wf{gru_at(P + L.gru_f, HP, d.C), P + L.gm_f};
  // shared encoder rows: a row's gradient is the sum over its consumers, i.e. over every node it stands for
  const bool shared = enc_shared(d, info);
  const SchedDir& Sf = shared ? sv.enc : sv.dir[0];
  const Levels lfs = shared ? levels_shared(d, info, lf) : lf;
  if (shared) w.f.R = w.f.RH = lfs.R;
  if (d.bidir) {
    const Levels lr = levels_of(d, info, 1);
    SweepW wr{gru_at(P + L.gru_b, HP, d.C), P + L.gm_b};
    DVAE_TRY(k_zero(w.r.DH, (size_t)std::max(lf.R, 1) * HP * 4, st));
    // chain backward; DHVC[t] = dL/d hv_t, DHVC[N] seeded by the heads.  One launch per step: d_t = d_next.W01a for the
    // graphs with a step t (their d_next row is also the gradient of Q's row, DQ, and goes to DA, the copy with the
    // finished graphs' rows zeroed that the weight gradient sums over), d_next itself for the finished ones
    DVAE_CHECK_HIP(hipMemcpyAsync(w.DHVC + (size_t)d.N * B * HP, w.DHVF, (size_t)B * HP * 4,
                                  hipMemcpyDeviceToDevice, st));
    for (int t = d.N - 1; t >= 0; --t) {
      float* d_t = w.DHVC + (size_t)t * B * HP;
      DVAE_TRY(k_chain_bwd_step(sv.dir[1], sizes, B, d.N, t, d_t + (size_t)B * HP, P + L.W01, 2 * HP, d_t, w.DQ,
                                w.DA + (size_t)t * B * HP, HP, g.prio, st));
    }
    DVAE_TRY(k_take_last_bwd(Sf, sizes, B, d.N, w.DHVC, w.f.DH, HP, st));
    // dL/dhb = DQ.W01b into the reverse sweep's DH (zero so far); weight gradients on the side stream:
    // dW01a = DA^T.hv_t over all N*B chain rows, dW01b = DQ^T.hb over the reverse-sweep rows, db01 = column sums of DQ
    DVAE_TRY(g.nn(w.DQ, HP, P + L.W01 + HP, 2 * HP, w.r.DH, HP, lr.R, HP, HP, 1));
    DVAE_TRY(ev.edge(st, sd));
    DVAE_TRY(gs.tn(w.DA, HP, w.HVC, HP, Gp + L.W01, 2 * HP, HP, HP, d.N * B));
    DVAE_TRY(gs.tn(w.DQ, HP, w.r.Hn, HP, Gp + L.W01 + HP, 2 * HP, HP, HP, lr.R));
    DVAE_TRY(k_colsum(w.DQ, HP, lr.R, HP, Gp + L.b01, ssc.part, ssc.part_floats, sd));
    DVAE_TRY(sweep_bwd_wgrad(d, sv.dir[1], lr, w.r, wr, x,
                             WgradOut{Gp + L.gru_b, Gp + L.gm_b, Gp + L.xg_gru_b, Gp + L.xg_gm_b}, false, g, gs, ev, info[5] == 1));
  } else {
    DVAE_TRY(k_take_last_bwd(Sf, sizes, B, d.N, w.DHVF, w.f.DH, HP, st));
  }
  const G gtail{w.sc[0].gemm_ws, st, 0, 0, w.sc[0].part, w.sc[0].part_floats};   // the chain's half of the last products
  DVAE_TRY(sweep_bwd_wgrad(d, Sf, lfs, w.f, wf, x,
                           WgradOut{Gp + L.gru_f, Gp + L.gm_f, Gp + L.xg_gru_f, Gp + L.xg_gm_f}, false, g, gs, ev, info[5] == 1,
                           false, nullptr, &gtail));
  return DVAE_OK;
}

// ===========================================================================
// teacher-forced decoder
// ===========================================================================
extern "C" size_t dvae_dec_ws_bytes(const dvae_dims* d, int R, int E) { return dec_ws(*d, R, E, nullptr).bytes; }

extern "C" int dvae_decoder_fwd(const dvae_dims* dp, const float* z, const float* x, const int* sizes,
                                const void* sched, const int* info, const float* P, void* ws,
                                float* edge_logits, float* type_logits, float* states, void* stream,
                                void* side_stream) {
  DVAE_REQUIRE(dp && z && x && sched && P && ws && edge_logits && type_logits, "decoder_fwd: null pointer");
  const dvae_dims& d = *dp;
  DVAE_TRY(check_info(d, info));
  hipStream_t st = (hipStream_t)stream;
  hipStream_t sd = side_stream ? (hipStream_t)side_stream : st;
  const SchedView sv = sched_view(const_cast<void*>(sched), d.B, d.N);
  const SchedDir& S = sv.dir[0];
  const Levels lv = levels_of(d, info, 0);
  DecWs w = dec_ws(d, lv.R, lv.E, ws);
  const DecLayout L = dec_layout(d);
  const G g{w.sc[0].gemm_ws, st, sd != st ? 1 : 0};
  const G gs{w.sc[1].gemm_ws, sd, 0, sd != st ? 1 : 0};
  Events& ev = g_events;
  const int HP = d.HP, B = d.B, R = lv.R, EHP = d.EHP;

  // h0 = lin1(z) (dvae_model.py:194)
  DVAE_TRY(k_pad_rows(z, B, d.Z, w.ZPAD, d.ZP, d.ZP, st));
  DVAE_TRY(g.nt(w.ZPAD, d.ZP, P + L.Wl1, d.ZP, w.H0, HP, B, HP, d.ZP, P + L.bl1));
  if (d.mod) {
    DVAE_TRY(ev.edge(st, sd));
    DVAE_TRY(gs.nt(w.H0, HP, P + L.W1c, HP, w.Cq, EHP, B, EHP, HP));
  }
  // add_edge's first layer, split by input block (dvae_model.py:211-218), follows the
  // sweep level by level on the side stream
  SweepW sw{gru_at(P + L.gru, HP, d.C), P + L.gm};
  const bool typed = type_shared(d, info, lv);
  const int z1 = lv.nlev > 0 ? lv.ls[1] : R;
  TypeShare ts{nullptr, z1};
  if (typed) {
    // rows with (type, h_in = 0): one table row per type instead of one GEMM row per node
    DVAE_TRY(ev.edge(st, sd));
    DVAE_TRY(k_type_states(d.C, sw.gru, w.TYPEV, HP, sd));
    DVAE_TRY(gs.nt(w.TYPEV, HP, P + L.W1a, HP, w.AT, EHP, 32, EHP, HP));
    ts.AT = w.AT;
  }
  const int Rb = bq_rows(info, lv);
  const bool scored = scored_rows_only(), skip0 = skip_level0(info);
  LevelHook hook = [&](int p0, int p1, int ns, int ne) -> int {
    DVAE_TRY(ev.edge(st, sd));
    // k = 0 rows, then edge rows
    // (type-shared k = 0 rows [z1,R) come from the per-type table, not from a product)
    const int n0 = (skip0 && p0 == 0) ? std::min(z1, p1) : p0, n1 = std::min(p1, typed ? z1 : R);
    if (n1 > n0)
      DVAE_TRY(gs.nt(w.s.HV + (size_t)n0 * HP, HP, P + L.W1a, HP, w.A + (size_t)n0 * EHP, EHP, n1 - n0, EHP, HP));
    const int e0 = std::max(p0, R), e1 = p1;
    if (e1 > e0) {
      const int u0 = scored ? lv.scored_before(e0 - R) : -1, u1 = scored ? lv.scored_before(e1 - R) : -1;
      if (u0 >= 0 && u1 >= u0) {
        DVAE_TRY(gs.nt(w.s.HV, HP, P + L.W1a, HP, w.A, EHP, u1 - u0, EHP, HP, nullptr, 0, 0, sv.a_rows + u0));
      } else {
        DVAE_TRY(gs.nt(w.s.HV + (size_t)e0 * HP, HP, P + L.W1a, HP, w.A + (size_t)e0 * EHP, EHP, e1 - e0, EHP, HP));
      }
    }
    DVAE_TRY(gs.nt(w.s.Hn + (size_t)ns * HP, HP, P + L.W1b, HP, w.Bq + (size_t)ns * EHP, EHP,
                   std::min(ne, Rb) - ns, EHP, HP, P + L.b1));
    return DVAE_OK;
  };
  DVAE_TRY(sweep_fwd(d, S, lv, w.s, sw, x, w.H0, g, hook, typed));
  phase_mark(2, st);

  // pair scores need every A / Bq row: the chain waits for the side stream's products here
  DVAE_CHECK_HIP(hipMemsetAsync(edge_logits, 0, (size_t)B * d.N * d.N * 4, st));
  DVAE_TRY(ev.edge(sd, st));
  phase_mark(3, st);
  // add_vertex on the graph state seen before each node (dvae_model.py:199-200,251-257): MFMA
  // work on the side stream, beside the bandwidth-bound pair scorer on the chain
  // (DVAE_AV_MAIN=1 keeps it on the chain: A-B tests)
  static const bool av_main = getenv("DVAE_AV_MAIN") != nullptr;
  hipStream_t av = av_main ? st : sd;
  const G& gav = av_main ? g : gs;
  DVAE_TRY(ev.edge(st, av));
  DVAE_TRY(k_graph_state_fwd(S, R, d.N, d.mod, w.s.Hn, w.H0, w.GS, HP, av));
  DVAE_TRY(gav.nt(w.GS, HP, P + L.Wv1, HP, w.VH, d.VHP, R, d.VHP, HP, P + L.bv1, 1));
  DVAE_TRY(gav.nt(w.VH, d.VHP, P + L.Wv2, d.VHP, w.TL, d.CP, R, d.CP, d.VHP, P + L.bv2));
  DVAE_CHECK_HIP(hipMemsetAsync(type_logits, 0, (size_t)B * d.N * d.C * 4, av));
  DVAE_TRY(k_rows_to_dense(S, R, w.TL, d.CP, type_logits, d.N, d.C, av));
  if (states) {
    DVAE_CHECK_HIP(hipMemsetAsync(states, 0, (size_t)B * d.N * d.H * 4, st));
    DVAE_TRY(k_rows_to_dense(S, R, w.s.Hn, HP, states, d.N, d.H, st));
  }
  {
    KernelProf kp(4, pair_bytes(d, lv), st);
    DVAE_TRY(k_pair_fwd(sv, ts, R, sizes, w.A, w.Bq, w.Cq, P + L.w2, P + L.b2, edge_logits, EHP, st));
  }
  phase_mark(4, st);
  DVAE_TRY(ev.edge(sd, st));
  phase_mark(5, st);
  return DVAE_OK;
}

extern "C" int dvae_decoder_bwd(const dvae_dims* dp, const float* z, const float* x, const int* sizes,
                                const void* sched, const int* info, const float* P, void* ws,
                                const float* d_edge, const float* d_type, float* Gp, float* d_z,
                                void* stream, void* side_stream) {
  (void)z;
  DVAE_REQUIRE(dp && x && sched && P && ws && d_edge && d_type && Gp, "decoder_bwd: null pointer");
  const dvae_dims& d = *dp;
  DVAE_TRY(check_info(d, info));
  hipStream_t st = (hipStream_t)stream;
  hipStream_t sd = side_stream ? (hipStream_t)side_stream : st;
  const SchedView sv = sched_view(const_cast<void*>(sched), d.B, d.N);
  const SchedDir& S = sv.dir[0];
  const Levels lv = levels_of(d, info, 0);
  DecWs w = dec_ws(d, lv.R, lv.E, ws);
  const DecLayout L = dec_layout(d);
  const G g{w.sc[0].gemm_ws, st, sd != st ? 1 : 0};
  const G gs{w.sc[1].gemm_ws, sd, 0, sd != st ? 1 : 0, w.sc[1].part, w.sc[1].part_floats};       // weight-gradient products
  const Scratch& msc = w.sc[0];
  const Scratch& ssc = w.sc[1];
  Events& ev = g_events;
  const int HP = d.HP, B = d.B, R = lv.R, RH = lv.R + lv.E, EHP = d.EHP;

  // ---- add_vertex backward (MFMA-bound) runs on the side stream beside the bandwidth-bound
  // pair-scorer backward on the chain; then every weight-gradient product follows there
  // (DVAE_AV_MAIN=1 keeps the input-gradient part on the chain: A-B tests) ----
  static const bool av_main = getenv("DVAE_AV_MAIN") != nullptr;
  hipStream_t av = av_main ? st : sd;
  const G& gav = av_main ? g : gs;
  DVAE_TRY(ev.edge(st, sd));
  DVAE_TRY(k_dense_to_rows(S, R, d_type, d.N, d.C, w.DTL, d.CP, av));
  DVAE_TRY(gav.nn(w.DTL, d.CP, P + L.Wv2, d.VHP, w.DVH, d.VHP, R, d.VHP, d.CP));
  // ReLU backward and add_vertex.0's bias gradient in one pass over DVH
  DVAE_TRY(k_relu_bwd_colsum(w.DVH, w.VH, d.VHP, R, d.VHP, Gp + L.bv1, av_main ? msc.part : ssc.part,
                             av_main ? msc.part_floats : ssc.part_floats, av));
  // the graph state seen by add_vertex at node i is the state of node i-1 (bn: all earlier
  // nodes), which may sit in any level: this product cannot follow the sweep level by level
  DVAE_TRY(gav.nn(w.DVH, d.VHP, P + L.Wv1, HP, w.DGS, HP, R, HP, d.VHP));
  hipEvent_t dgs_done = nullptr;
  if (av != st) {
    DVAE_TRY(ev.take(&dgs_done));
    DVAE_CHECK_HIP(hipEventRecord(dgs_done, sd));
  } else {
    DVAE_TRY(ev.edge(st, sd));
  }
  DVAE_TRY(gs.tn(w.DTL, d.CP, w.VH, d.VHP, Gp + L.Wv2, d.VHP, d.CP, d.VHP, R));
  DVAE_TRY(k_colsum(w.DTL, d.CP, R, d.CP, Gp + L.bv2, ssc.part, ssc.part_floats, sd));
  DVAE_TRY(gs.tn(w.DVH, d.VHP, w.GS, HP, Gp + L.Wv1, HP, d.VHP, HP, R));

  SweepW sw{gru_at(P + L.gru, HP, d.C), P + L.gm};
  const bool typed = type_shared(d, info, lv);
  const int z1 = lv.nlev > 0 ? lv.ls[1] : R;
  const TypeShare ts{typed ? w.AT : nullptr, z1};
  const bool scored = scored_rows_only(), skip0 = skip_level0(info);
  const int U = (scored && lv.as[d.N] <= lv.E) ? lv.as[d.N] : -1;   // scored edge rows (sv.a_rows), -1: all rows
  // one-pass backward of the pair scorer: small graphs keep every candidate's accumulator on chip, larger
  // ones walk the candidates in register-resident windows; both also sum db1 (the column sums of DBq)
  const bool pair_one_pass = pair_bwd_fused_ok(d.N, EHP) || pair_bwd_win_ok(EHP);
  {
  KernelProf pair_prof(5, 2.0 * pair_bytes(d, lv), st);   // scope: the scorer's backward launches
  if (pair_one_pass) {
    // its partial sums carry db1, dw2 and db2, adjacent in the packed layout
    DVAE_REQUIRE(L.w2 == L.b1 + (size_t)EHP && L.b2 == L.w2 + (size_t)EHP, "decoder layout: b1 | w2 | b2 must be adjacent");
    if (pair_bwd_fused_ok(d.N, EHP))
      DVAE_TRY(k_pair_bwd_fused(sv, ts, R, sizes, w.A, w.Bq, w.Cq, P + L.w2, d_edge, w.DA, w.DBq, Gp + L.b1, msc.part,
                                msc.part_floats, EHP, st));
    else
      DVAE_TRY(k_pair_bwd_win(sv, ts, R, sizes, w.A, w.Bq, w.Cq, P + L.w2, d_edge, w.DA, w.DBq, Gp + L.b1, msc.part,
                              msc.part_floats, EHP, st));
  } else {
    DVAE_TRY(k_pair_bwd_a(sv, ts, R, RH, w.A, w.Bq, w.Cq, P + L.w2, d_edge, w.DA, Gp + L.w2, msc.part,
                          msc.part_floats, EHP, st));
    DVAE_TRY(k_pair_bwd_b(sv, ts, R, sizes, w.A, w.Bq, w.Cq, P + L.w2, d_edge, w.DBq, EHP, st));
  }
  }
  if (d.mod) DVAE_TRY(k_graph_sum(S, sizes, B, d.N, w.DBq, EHP, w.DCq, st));
  phase_mark(7, st);
  // the add_edge weight-gradient products (side stream).  Lab switch: issue them AFTER the chain's own
  // add_edge input-gradient products, so that those two -- the largest launches on the chain (B=512: 219 +
  // 80 us beside the side stream's products, 111 + 40 us alone) -- run alone.  Measured neutral (B=512: 2.613
  // vs 2.618 ms, B=1024: 3.96 vs 3.92, same box): what the chain gains there the sweep behind it loses to
  // the deferred products.  Off by default; DVAE_DEFER_SIDE_ROWS=<node rows below which they are deferred>.
  static const int defer_rows = getenv("DVAE_DEFER_SIDE_ROWS") ? atoi(getenv("DVAE_DEFER_SIDE_ROWS")) : 0;
  const bool defer_side = sd != st && R < defer_rows;
  const int Rb = bq_rows(info, lv);   // DBq rows beyond Rb are zero (no pair has them as candidate)
  auto side_products = [&]() -> int {
    DVAE_TRY(ev.edge(st, sd));
    if (typed) {
      // type-shared rows [z1,R): their gradients meet in the per-type sums DTT = DA^T . onehot(type)
      //   d hv(type) = DTT^T . W1a  -> GRU backward once per type (k_type_gru_bwd)
      //   dW1a      += DTT . hv(type)
      DVAE_TRY(k_build_xp(S, 1, RH, x, d.N, d.C, w.s.XP, sd));
      // DTT = DA^T . onehot(type) over the k = 0 rows of levels >= 1: a per-type column sum, one streaming pass over DA
      // (DVAE_NO_TYPED_COLSUM=1: the [EHP x 32] TN product instead; A-B tests)
      static const bool dtt_gemm = getenv("DVAE_NO_TYPED_COLSUM") != nullptr;
      if (dtt_gemm)
        DVAE_TRY(gs.tn(w.DA + (size_t)z1 * EHP, EHP, w.s.XP + (size_t)z1 * 32, 32, w.DTT, 32, EHP, 32, R - z1));
      else
        DVAE_TRY(k_typed_colsum(w.DA + (size_t)z1 * EHP, EHP, R - z1, EHP, sv.row_type + z1, d.C, w.DTT, ssc.part,
                                ssc.part_floats, sd));
      DVAE_TRY(gs.tn(w.DTT, 32, P + L.W1a, HP, w.DTV, HP, 32, HP, EHP));
      DVAE_TRY(k_type_gru_bwd(d.C, sw.gru, w.DTV, w.TG, w.TGN, w.XPT, HP, sd));
      // dW1a over the scored edge rows: gathered through the row list, or (DVAE_W1A_TN_ALL=1, lab) over every edge row --
      // the pair scorer's backward writes zeros into the DA rows no pair is scored on, so both sums are the same
      static const bool w1a_all = getenv("DVAE_W1A_TN_ALL") != nullptr;
      if (U >= 0 && !w1a_all) DVAE_TRY(gs.tn(w.DA, EHP, w.s.HV, HP, Gp + L.W1a, HP, EHP, HP, U, 0, sv.a_rows));
      else DVAE_TRY(gs.tn(w.DA + (size_t)R * EHP, EHP, w.s.HV + (size_t)R * HP, HP, Gp + L.W1a, HP, EHP, HP, RH - R));
      if (!skip0) DVAE_TRY(gs.tn(w.DA, EHP, w.s.HV, HP, Gp + L.W1a, HP, EHP, HP, z1, 1));
      DVAE_TRY(gs.nn(w.DTT, 32, w.TYPEV, HP, Gp + L.W1a, HP, EHP, HP, 32, 1));
    } else {
      DVAE_TRY(gs.tn(w.DA, EHP, w.s.HV, HP, Gp + L.W1a, HP, EHP, HP, RH));
    }
    DVAE_TRY(gs.tn(w.DBq, EHP, w.s.Hn, HP, Gp + L.W1b, HP, EHP, HP, Rb));
    if (!pair_one_pass)   // (the one-pass pair backward has summed db1 already)
      DVAE_TRY(k_colsum(w.DBq, EHP, R, EHP, Gp + L.b1, ssc.part, ssc.part_floats, sd));
    if (d.mod) DVAE_TRY(gs.tn(w.DCq, EHP, w.H0, HP, Gp + L.W1c, HP, EHP, HP, B));
    return DVAE_OK;
  };
  if (!defer_side) DVAE_TRY(side_products());

  // ---- add_edge input gradients, then the level sweep ----
  // (measured and dropped: producing dA.W1a / dBq.W1b level by level just ahead of the sweep and
  // issuing the sweep's weight-gradient products level by level behind it -- the smaller launches
  // cost more than the earlier start and the shorter tail gave: +0.5 ms at B=4096)
  {
    // k = 0 rows: level 0 (unless it holds node 0 only, which no pair is scored on), and the later
    // levels when they are not shared by type
    const int n0 = skip0 ? z1 : 0, n1 = typed ? z1 : R;
    if (n0 > 0) DVAE_TRY(k_zero(w.s.DH, (size_t)n0 * HP * 4, st));
    if (n1 > n0)
      DVAE_TRY(g.nn(w.DA + (size_t)n0 * EHP, EHP, P + L.W1a, HP, w.s.DH + (size_t)n0 * HP, HP, n1 - n0, HP, EHP));
    // edge rows: the scored ones, zeros in the others
    if (U >= 0) {
      // the unscored rows are not written at all: the GRU backward takes their dL/dHV as zero (SweepBufs::apos);
      // DVAE_ZERO_SKIPPED=1 zero-fills them as before (A-B tests)
      static const bool zero_skipped = getenv("DVAE_ZERO_SKIPPED") != nullptr;
      if (zero_skipped) DVAE_TRY(k_zero_rows(w.s.DH, HP, HP, sv.a_skip, lv.E - U, st));
      else w.s.apos = sv.a_pos;
      DVAE_TRY(g.nn(w.DA, EHP, P + L.W1a, HP, w.s.DH, HP, U, HP, EHP, 0, sv.a_rows));
    } else {
      DVAE_TRY(g.nn(w.DA + (size_t)R * EHP, EHP, P + L.W1a, HP, w.s.DH + (size_t)R * HP, HP, RH - R, HP, EHP));
    }
  }
  // the add_vertex input gradient (side stream) is needed from here on only: the product above
  // runs while the side stream still computes it
  if (dgs_done) DVAE_CHECK_HIP(hipStreamWaitEvent(st, dgs_done, 0));
  DVAE_TRY(k_zero(w.DH0, (size_t)B * HP * 4, st));
  DVAE_TRY(k_graph_state_bwd(S, R, B, d.N, d.mod, sizes, w.DGS, w.s.DHn, w.DH0, HP, st));
  DVAE_TRY(g.nn(w.DBq, EHP, P + L.W1b, HP, w.s.DHn, HP, Rb, HP, EHP, 1));
  if (defer_side) DVAE_TRY(side_products());
  phase_mark(8, st);
  const TypeGrad tgrad{w.TG, w.TGN, w.XPT};
  DVAE_TRY(sweep_bwd_wgrad(d, S, lv, w.s, sw, x, WgradOut{Gp + L.gru, Gp + L.gm, Gp + L.xg_gru, Gp + L.xg_gm}, true,
                           g, gs, ev, info[5] == 1, typed, typed ? &tgrad : nullptr));
  phase_mark(9, st);
  DVAE_TRY(k_add_dh0(S, B, d.N, w.s.DH, w.DH0, HP, st));
  if (d.mod) DVAE_TRY(g.nn(w.DCq, EHP, P + L.W1c, HP, w.DH0, HP, B, HP, EHP, 1));

  // ---- lin1 ----
  // its weight gradients go to the side stream like every other write to G: the caller orders its use of G
  // after the SIDE stream only (include/dvae_hip.h), so a write to G queued on the chain would race with the
  // caller's unpack whenever the side stream drains first (small batches)
  DVAE_TRY(ev.edge(st, sd));
  DVAE_TRY(gs.tn(w.DH0, HP, w.ZPAD, d.ZP, Gp + L.Wl1, d.ZP, HP, d.ZP, B));
  DVAE_TRY(k_colsum(w.DH0, HP, B, HP, Gp + L.bl1, ssc.part, ssc.part_floats, sd));
  if (d_z) {
    DVAE_TRY(g.nn(w.DH0, HP, P + L.Wl1, d.ZP, w.DZP, d.ZP, B, d.ZP, HP));
    DVAE_TRY(k_unpad_rows(w.DZP, d.ZP, B, d.Z, d_z, st));
  }
  return DVAE_OK;
}

// ===========================================================================
// side stream
// ===========================================================================
extern "C" int dvae_side_stream_create(void** stream_out, int* priority_out) {
  DVAE_REQUIRE(stream_out, "side_stream_create: null pointer");
  int least = 0, greatest = 0;
  DVAE_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  int prio = least;
  if (const char* e = getenv("DVAE_SIDE_PRIORITY")) prio = std::max(std::min(atoi(e), least), greatest);   // tuning
  hipStream_t s;
  DVAE_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
  *stream_out = (void*)s;
  if (priority_out) *priority_out = prio;
  DVAE_TRY(elbo_static_warm());   // dvae_elbo_static's per-device symbol look-up, done at set-up (loss.hip)
  return DVAE_OK;
}

extern "C" int dvae_side_stream_destroy(void* stream) {
  if (stream) DVAE_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
  return DVAE_OK;
}

// tuning / test knob: share of a sweep's rows whose weight-gradient products start before the
// sweep has finished (0 = none), and the row count below which there is no early chunk
extern "C" int dvae_wgrad_split_tune(double frac, int min_rows) {
  g_wgrad_split.frac = frac;
  g_wgrad_split.min_rows = min_rows;
  return DVAE_OK;
}

// tuning / test knob: decoder levels of at least `min_rows` node rows rebuild their GRU gates in the backward pass instead of
// saving them in the forward pass (negative = never).  Set it between steps, not between a forward and its backward.
extern "C" int dvae_recompute_tune(int min_rows) {
  g_recompute_min_rows = min_rows;
  return DVAE_OK;
}

// tuning / test knob: per-level products of at most `rows` rows take level.hip's small-level kernels (0 = never)
extern "C" int dvae_small_rows_tune(int rows) {
  g_level_cfg.small_rows = rows;
  return DVAE_OK;
}

// tuning / test knob: rows from which the gate/map step runs fused (-1 never, 0 always),
// whether the GRU step runs fused, whether the decoder shares its type-only prefix rows
extern "C" int dvae_level_tune(int fused_gm_min_rows, int fused_gru, int type_share) {
  g_level_cfg.fused_gm_min_rows = fused_gm_min_rows;
  g_level_cfg.fused_gru = fused_gru;
  g_level_cfg.type_share = type_share;
  return DVAE_OK;
}

// ===========================================================================
// one fused train step behind two calls (the host glue of Dvae.loss_and_grad in C)
// ===========================================================================
namespace {
struct StepWs {
  float *mu, *logvar, *z, *edge_logits, *type_logits, *d_edge, *d_type, *d_mu, *d_logvar, *d_z, *dmu2, *dlv2;
  void* elbo_part;
  size_t bytes;
};
StepWs step_ws(const dvae_dims& d, void* base) {
  StepWs w;
  Carver c{(char*)base, 0};
  const size_t B = d.B, N = d.N;
  w.mu = c.f(B * d.Z); w.logvar = c.f(B * d.Z); w.z = c.f(B * d.Z);
  w.edge_logits = c.f(B * N * N); w.type_logits = c.f(B * N * d.C);
  w.d_edge = c.f(B * N * N); w.d_type = c.f(B * N * d.C);
  w.d_mu = c.f(B * d.Z); w.d_logvar = c.f(B * d.Z); w.d_z = c.f(B * d.Z);
  w.dmu2 = c.f(B * d.Z); w.dlv2 = c.f(B * d.Z);
  w.elbo_part = c.f((dvae_elbo_ws_bytes(d.B) + 3) / 4);
  w.bytes = c.off;
  return w;
}
}  // namespace

extern "C" size_t dvae_step_ws_bytes(const dvae_dims* d) {
  if (!d || d->HP <= 0) {
    dvae_set_error("step_ws_bytes: null or uninitialised dims (call dvae_dims_init)");
    return 0;
  }
  return step_ws(*d, nullptr).bytes;
}

// the two dims of a step describe ONE batch: ws_step is carved from d_dec while the encoder writes mu / logvar sized
// by d_enc, so a mismatched pair would overrun it
static int step_dims_ok(const dvae_dims& de, const dvae_dims& dd) {
  DVAE_REQUIRE(de.HP > 0 && dd.HP > 0, "step: dims not initialised (call dvae_dims_init)");
  DVAE_REQUIRE(de.B == dd.B && de.N == dd.N && de.C == dd.C && de.H == dd.H && de.HP == dd.HP && de.Z == dd.Z &&
                   de.mod == dd.mod,
               "step: d_enc and d_dec disagree (B %d/%d N %d/%d C %d/%d H %d/%d Z %d/%d mod %d/%d)", de.B, dd.B, de.N, dd.N,
               de.C, dd.C, de.H, dd.H, de.Z, dd.Z, de.mod, dd.mod);
  return DVAE_OK;
}

extern "C" int dvae_step_part1(const dvae_step* s, void* stream, void* side_stream) {
  DVAE_REQUIRE(s && s->d_enc && s->d_dec && s->g && s->x && s->sched && s->info_host && s->Pe && s->Pd && s->ws_enc &&
                   s->ws_dec && s->ws_step && s->eps && s->Ge && s->Gd && s->out4 && s->enc_table && s->dec_table,
               "step_part1: null pointer");
  const dvae_dims& de = *s->d_enc;
  const dvae_dims& dd = *s->d_dec;
  DVAE_TRY(step_dims_ok(de, dd));
  hipStream_t st = (hipStream_t)stream;
  hipStream_t sd = side_stream ? (hipStream_t)side_stream : st;
  Events& ev = g_events;
  const StepWs w = step_ws(dd, s->ws_step);
  phase_mark(0, st);
  const long long nz = (long long)dd.B * dd.Z;
  // packed gradient buffers start at zero; the decoder's weights are packed on the side stream, they are
  // not needed before the encoder is through
  // (on the side stream: every write to Ge / Gd but the pair scorer's bias sums is queued there, and the chain meets the
  // side stream again before the decoder starts)
  DVAE_TRY(dvae_pack_params(s->enc_table, s->enc_n, s->Pe, st));
  DVAE_TRY(ev.edge(st, sd));
  DVAE_TRY(k_zero(s->Ge, dvae_enc_grad_floats(&de) * sizeof(float), sd));
  DVAE_TRY(k_zero(s->Gd, dvae_dec_grad_floats(&dd) * sizeof(float), sd));
  DVAE_TRY(dvae_pack_params(s->dec_table, s->dec_n, s->Pd, sd));
  DVAE_TRY(dvae_encoder_fwd(&de, s->x, s->sizes, s->sched, s->info_host, s->Pe, s->ws_enc, nullptr, w.mu, w.logvar,
                            nullptr, nullptr, st));
  DVAE_TRY(dvae_reparam_fwd(w.mu, w.logvar, s->eps, w.z, nz, st));
  phase_mark(1, st);
  DVAE_TRY(ev.edge(sd, st));
  DVAE_TRY(dvae_decoder_fwd(&dd, w.z, s->x, s->sizes, s->sched, s->info_host, s->Pd, s->ws_dec, w.edge_logits,
                            w.type_logits, nullptr, st, side_stream));
  DVAE_TRY(dvae_elbo(&dd, w.edge_logits, w.type_logits, s->g, s->x, s->sizes, w.mu, w.logvar, s->beta, s->out4,
                     w.elbo_part, w.d_edge, w.d_type, w.d_mu, w.d_logvar, st));
  phase_mark(6, st);
  DVAE_TRY(dvae_decoder_bwd(&dd, w.z, s->x, s->sizes, s->sched, s->info_host, s->Pd, s->ws_dec, w.d_edge, w.d_type,
                            s->Gd, w.d_z, st, side_stream));
  DVAE_TRY(k_reparam_bwd_acc(w.d_z, w.logvar, s->eps, w.d_mu, w.d_logvar, nz, st));
  phase_mark(10, st);
  return DVAE_OK;
}

extern "C" int dvae_step_part2(const dvae_step* s, void* stream, void* side_stream) {
  DVAE_REQUIRE(s && s->d_enc && s->d_dec && s->x && s->sched && s->info_host && s->Pe && s->ws_enc && s->ws_step && s->Ge,
               "step_part2: null pointer");
  DVAE_TRY(step_dims_ok(*s->d_enc, *s->d_dec));
  const StepWs w = step_ws(*s->d_dec, s->ws_step);
  const int rc = dvae_encoder_bwd(s->d_enc, s->x, s->sizes, s->sched, s->info_host, s->Pe, s->ws_enc, nullptr, w.d_mu,
                                  w.d_logvar, s->Ge, stream, side_stream);
  phase_mark(11, (hipStream_t)stream);
  if (g_phases.on && side_stream) {   // ... and when the side stream is through with the step's last weight gradient
    g_events.edge((hipStream_t)side_stream, (hipStream_t)stream);
    phase_mark(12, (hipStream_t)stream);
  }
  return rc;
}

// lab (DVAE_PHASES=1): waits for the device, then out_ms[i] = time of mark i since the first mark, tags[i] = its tag;
// returns the number of marks (at most max) and forgets them
extern "C" int dvae_phase_read(double* out_ms, int* tags, int max) {
  DVAE_CHECK_HIP(hipDeviceSynchronize());
  const int n = (int)std::min<size_t>(g_phases.ev.size(), (size_t)std::max(max, 0));
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    DVAE_CHECK_HIP(hipEventElapsedTime(&ms, g_phases.ev[0], g_phases.ev[i]));
    out_ms[i] = ms;
    tags[i] = g_phases.tag[i];
  }
  for (hipEvent_t e : g_phases.ev) (void)hipEventDestroy(e);
  g_phases.ev.clear();
  g_phases.tag.clear();
  return n;
}
