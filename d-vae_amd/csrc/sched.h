// Device-side view of the level schedule built by dvae_sched_build().
#pragma once
#include "common.h"

// One sweep direction.  "Predecessor" means the nodes whose gated states are
// summed into a node: columns of its adjacency row for the forward sweep
// (dvae_model.py:67), rows of its adjacency column for the reverse sweep (:90,97).
struct SchedDir {
  int* level;      // [B*N]   level per node, -1 for nodes beyond sizes[b]
  int* node2row;   // [B*N]   node row (nodes sorted by (level, graph, index)), -1 if absent
  int* pos;        // [N*B+1] first node row of (level L, graph b) at [L*B+b]; [N*B] = R
  int* row_graph;  // [RM]
  int* row_node;   // [RM]
  int* csr_ptr;    // [RM+1]  predecessor list bounds (lists in DESCENDING node index)
  int* csc_ptr;    // [RM+1]  successor list bounds
  int* csr_idx;    // [EM]    node row of each predecessor
  int* csc_idx;    // [EM]    node row of each successor
  int* csc_prow;   // [EM]    forward only: first decoder prefix row of the successor that
                   //         includes this node, = prow(succ, t)
  int* csc_cnt;    // [EM]    forward only: how many prefix rows of the successor include it
  int* prow_node;  // [RM+EM] forward only: node row of each decoder prefix row
};

struct SchedView {
  int B, N, RM, EM;
  uint32_t* rowbits;  // [B*N*4] bit u of row i set <=> edge u->i
  uint32_t* colbits;  // [B*N*4] bit u of row i set <=> edge i->u
  int* err;           // [64] 0: invalid adjacency, 1-2: dvae_sched_types, 3: level 0 holds a node other than node 0
  int* scan_tmp;      // tile sums for the scan
  int* info;          // [DVAE_SCHED_INFO_INTS(N)] device copy of info_host
  int* row_type;      // [RM] forward node row -> node type if its encoding is exactly one-hot,
                      //      else -1 (filled by dvae_sched_types; info[5] = 1 when every row has one)
  int* a_pos;         // [EM+1] scored edge rows before edge e (exclusive scan)
  int* a_rows;        // [EM] the decoder's edge prefix rows that add_edge scores, ascending: every
                      //      edge row except a node's FULL prefix when node 0 is among its predecessors
                      //      (the last pair of such a node, (i,0), is scored before edge 0 is added)
  int* a_skip;        // [EM] the other edge rows, ascending
  SchedDir dir[2];    // 0 forward, 1 reverse
  // ---- encoder value numbering (dvae_sched_types, sched.hip "shared encoder rows") ----
  // Two nodes of the batch with the same index, the same type and predecessor lists that hold the same
  // VALUES have the same encoder state (dvae_model.py:63-87 computes it from nothing else), so the forward
  // sweep of the encoder runs on one row per distinct value.  enc is dir[0] with those rows: node2row maps
  // every node to its value's row, row_graph/row_node name the representative (lowest graph index), csr/csc
  // are the links between value rows (csc lists ascending).  The last node of a graph is never shared (its row
  // takes the graph's own gradient).  info[DVAE_SCHED_MERGE_AT(N)] = 1 when enc is valid.
  SchedDir enc;
  unsigned long long* vh;       // [2*RM] 128-bit value hash per node
  unsigned long long* skey[2];  // [RM] sort keys in / out
  int* sval[2];                 // [max(RM,EM)] sort values in / out
  int* pkey[2];                 // [EM] (predecessor row, row) pairs: keys in / out
  int* rep;                     // [RM] node id (b*N+i) of the representative of each node's value
  int* gidx;                    // [RM+1] value group of each sorted position
  int* gfirst;                  // [RM] first node id of each group
  void* sort_tmp; size_t sort_tmp_bytes;
};
// first int of the shared-row summary in the schedule info: [0] valid, [1] rows, [2] links, [3] spare, [4..4+N] level
// starts, [5+N..5+2N] successor links before each level start
#define DVAE_SCHED_MERGE_AT(N) (8 + 4 * ((N) + 1))

// Decoder prefix rows of node row r (k = 0..deg(r), k predecessors summed):
//   prow(r,0) = r                       -- h_in = 0 (or h0 for node 0): no hidden GEMM
//   prow(r,k) = R + csr_ptr[r] + k - 1  -- one row per edge
// so both classes are contiguous per level: [ls[L], ls[L+1]) and R + [es[L], es[L+1]).
__host__ __device__ static inline int prow_of(const SchedDir& S, int R, int r, int k) {
  return k == 0 ? r : R + S.csr_ptr[r] + k - 1;
}
__host__ __device__ static inline int final_prow(const SchedDir& S, int R, int r) {
  const int e1 = S.csr_ptr[r + 1];
  return e1 == S.csr_ptr[r] ? r : R + e1 - 1;
}
SchedView sched_view(void* base, int B, int N);
bool sched_sizes_ok(int B, int N);   // every int32 row / edge index of a [B][N][N] batch fits

// Decoder k = 0 prefix rows of levels >= 1 have h_in = 0, so their GRU state -- and with it
// their add_edge first-layer row A = W1a.hv -- depends on the node TYPE only.  When every
// node encoding of the batch is exactly one-hot (info[5]), those rows are served from a
// C-row table instead of being computed per row.  AT == NULL: off.
struct TypeShare {
  const float* AT;   // [32][EHP] W1a . hv(type)
  int z1;            // node rows >= z1 (levels >= 1) share by type
};
__host__ __device__ static inline const float* a_row(const SchedView& sv, const TypeShare& ts, const float* A,
                                                     int EHP, int R, int r, int k) {
  if (k == 0 && ts.AT && r >= ts.z1) return ts.AT + (size_t)sv.row_type[r] * EHP;
  return A + (size_t)prow_of(sv.dir[0], R, r, k) * EHP;
}
