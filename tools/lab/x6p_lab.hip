// Lab (round 3): x6 GEMM on PRE-SPLIT operands.  Is it worth taking the fp32 -> 3 x bf16 split out of the k loop?
//   split_kernel   fp32 [R][K] -> "P3" rows: per row, per k-tile of 32: [piece1: 32 bf16][piece2][piece3] (192 B)
//   x6p_v1         register staged (global_load_dwordx4 -> ds_write_b128), one LDS stage, 256 threads, 2 workgroups / CU
//   x6p_v2         LDS-DMA staged (global_load_lds_dwordx4), two LDS stages, 512 threads, 256 x 128 tile, 1 workgroup / CU
// NT only: C[M][N] = sum_k A[m][k] B[n][k].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../d-vae_amd/csrc/x6.h"

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void split_kernel(const float* __restrict__ X, int ld, int R, int K, uint4* __restrict__ out) {
  // one thread per (row, k-tile, octet): 8 floats -> three 16-byte chunks
  const long long total = (long long)R * (K / 8);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / (K / 8)), oc = (int)(i % (K / 8));
    const float4 a = *reinterpret_cast<const float4*>(X + (size_t)r * ld + oc * 8);
    const float4 b = *reinterpret_cast<const float4*>(X + (size_t)r * ld + oc * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    Split3 s[8];
    for (int j = 0; j < 8; ++j) s[j] = split3(v[j]);
    uint4* dst = out + ((size_t)r * (K / 32) + oc / 4) * 12 + (oc & 3);
    dst[0] = make_uint4(pack_hi(s[0].p1, s[1].p1), pack_hi(s[2].p1, s[3].p1), pack_hi(s[4].p1, s[5].p1), pack_hi(s[6].p1, s[7].p1));
    dst[4] = make_uint4(pack_hi(s[0].p2, s[1].p2), pack_hi(s[2].p2, s[3].p2), pack_hi(s[4].p2, s[5].p2), pack_hi(s[6].p2, s[7].p2));
    dst[8] = make_uint4(pack_hi(s[0].p3, s[1].p3), pack_hi(s[2].p3, s[3].p3), pack_hi(s[4].p3, s[5].p3), pack_hi(s[6].p3, s[7].p3));
  }
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int x = bid & 7, j = bid >> 3;
  const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + j;
}

struct P {
  const uint4* A; const uint4* B; float* C;
  int M, N, K, ldc, tiles_n, nblk;
};

// ---------------- v1: register staged ----------------
template <int ROWS>
struct PTile {
  static constexpr int NCH = ROWS * 12 / 256;   // 16-byte chunks per thread
  static constexpr int ROWSP = ROWS + 1;        // odd: the 8 lanes of a ds_write_b128 group hit 8 different bank quads
  static constexpr int PLANE = 4 * ROWSP * 16;
  static constexpr int LDS_BYTES = 3 * PLANE;
  u32x4 v[NCH];
  const u32x4* src[NCH];
  __device__ __forceinline__ void init(const uint4* P3, int K, int r0, int rows, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * 256;
      const int r = min(r0 + id / 12, rows - 1);
      src[i] = reinterpret_cast<const u32x4*>(P3) + (size_t)r * (K / 32) * 12 + id % 12;
    }
  }
  __device__ __forceinline__ void load(int t) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = src[i][(size_t)t * 12];
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * 256;
      const int row = id / 12, c = id % 12;
      *reinterpret_cast<u32x4*>(lds + (c * ROWSP + row) * 16) = v[i];
    }
  }
  static __device__ __forceinline__ bf16x8 frag(const char* lds, int p, int rbase, int ks, int lane) {
    const char* q = lds + (((p * 4) + ks * 2 + (lane >> 5)) * ROWSP + rbase + (lane & 31)) * 16;
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q));
  }
};

template <int TM, int TN, class TA, class TB>
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[TM][TN], const char* As, const char* Bs, int arow, int brow, int lane) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) af[i][q] = TA::frag(As, q, arow + i * 32, ks, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) bf[j][q] = TB::frag(Bs, q, brow + j * 32, ks, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0], af[i][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][2], af[i][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][1], af[i][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0], af[i][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][1], af[i][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0], af[i][0], acc[i][j], 0, 0, 0);
      }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[TM][TN], const P& p, int mbase, int nbase, int lane) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = mbase + i * 32 + (lane & 31);
    if (row >= p.M) continue;
    float* crow = p.C + (size_t)row * p.ldc;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = nbase + j * 32 + 8 * g + 4 * (lane >> 5);
        if (col >= p.N) continue;
        *reinterpret_cast<float4*>(crow + col) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
  }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void x6p_v1(P p) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using TA = PTile<BM>;
  using TB = PTile<BN>;
  __shared__ __attribute__((aligned(16))) char lds[TA::LDS_BYTES + TB::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, p.nblk);
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  TA ta; TB tb;
  ta.init(p.A, p.K, m0, p.M, tid);
  tb.init(p.B, p.K, n0, p.N, tid);
  const int nt = p.K / 32;
  ta.load(0); tb.load(0);
  ta.store(lds, tid); tb.store(lds + TA::LDS_BYTES, tid);
  __syncthreads();
  for (int t = 0; t + 1 < nt; ++t) {
    ta.load(t + 1); tb.load(t + 1);
    mma_tile<TM, TN, TA, TB>(acc, lds, lds + TA::LDS_BYTES, wm * (BM / WM), wn * (BN / WN), lane);
    __syncthreads();
    ta.store(lds, tid); tb.store(lds + TA::LDS_BYTES, tid);
    __syncthreads();
  }
  mma_tile<TM, TN, TA, TB>(acc, lds, lds + TA::LDS_BYTES, wm * (BM / WM), wn * (BN / WN), lane);
  epilogue<TM, TN>(acc, p, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

// ---------------- v2: LDS-DMA staged, two stages ----------------
// image per piece plane: [row][64 B], the four 16-byte octet chunks of a row XOR-swizzled with (row >> 2) & 3
template <int ROWS>
struct DTile {
  static constexpr int PLANE = ROWS * 64;
  static constexpr int LDS_BYTES = 3 * PLANE;
  static __device__ __forceinline__ bf16x8 frag(const char* lds, int p, int rbase, int ks, int lane) {
    const int row = rbase + (lane & 31);
    const int o = (ks * 2 + (lane >> 5)) ^ ((row >> 2) & 3);
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + p * PLANE + row * 64 + o * 16));
  }
};

template <int BM, int BN, int WM, int WN, int NTHR>
__global__ __launch_bounds__(NTHR, 1) void x6p_v2(P p) {
  constexpr int NW = NTHR / 64;
  static_assert(WM * WN == NW, "waves");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using TA = DTile<BM>;
  using TB = DTile<BN>;
  constexpr int STAGE = TA::LDS_BYTES + TB::LDS_BYTES;
  constexpr int NINS = STAGE / 1024;            // wave-instructions per stage
  constexpr int PER_WAVE = NINS / NW;
  static_assert(NINS % NW == 0, "instructions per wave");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, p.nblk);
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // instruction q of this wave fills LDS bytes [ins * 1024, +1024) of a stage, ins = wave * PER_WAVE + q:
  // a 16-row slab of one piece plane of A (ins < 3 * BM / 16) or of B
  const char* src[PER_WAVE];
#pragma unroll
  for (int q = 0; q < PER_WAVE; ++q) {
    const int ins = wave * PER_WAVE + q;
    const bool isA = ins < 3 * BM / 16;
    const int rel = isA ? ins : ins - 3 * BM / 16;
    const int rows16 = (isA ? BM : BN) / 16;
    const int plane = rel / rows16, row = (rel % rows16) * 16 + (lane >> 2);
    const int o = (lane & 3) ^ ((row >> 2) & 3);
    const int gr = isA ? min(m0 + row, p.M - 1) : min(n0 + row, p.N - 1);
    src[q] = reinterpret_cast<const char*>(isA ? p.A : p.B) + ((size_t)gr * (p.K / 32) * 12 + plane * 4 + o) * 16;
  }
  auto stage = [&](int buf, int t) {
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
      const int ins = wave * PER_WAVE + q;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + (size_t)t * 192),
                                       (__attribute__((address_space(3))) void*)(lds + buf * STAGE + ins * 1024), 16, 0, 0);
    }
  };
  const int nt = p.K / 32;
  stage(0, 0);
  __syncthreads();
  int cur = 0;
  for (int t = 0; t + 1 < nt; ++t) {
    stage(cur ^ 1, t + 1);
    const char* As = lds + cur * STAGE;
    mma_tile<TM, TN, TA, TB>(acc, As, As + TA::LDS_BYTES, wm * (BM / WM), wn * (BN / WN), lane);
    __syncthreads();
    cur ^= 1;
  }
  {
    const char* As = lds + cur * STAGE;
    mma_tile<TM, TN, TA, TB>(acc, As, As + TA::LDS_BYTES, wm * (BM / WM), wn * (BN / WN), lane);
  }
  epilogue<TM, TN>(acc, p, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

// ---------------- v3: hybrid -- A split in the k loop (the in-tree XTile), B (the weights) pre-split ----------------
template <int ROWS>
struct XA : XTile<0, ROWS> {
  static constexpr int LDS_BYTES = XTile<0, ROWS>::LDS_FLOATS * 4;
  static __device__ __forceinline__ bf16x8 frag(const char* lds, int p, int rbase, int ks, int lane) {
    return XTile<0, ROWS>::frag(reinterpret_cast<const float*>(lds), p, rbase, ks, lane);
  }
};
template <int BKM, int ROWS>
struct XB : XTile<BKM, ROWS> {
  static constexpr int LDS_BYTES = XTile<BKM, ROWS>::LDS_FLOATS * 4;
  static __device__ __forceinline__ bf16x8 frag(const char* lds, int p, int rbase, int ks, int lane) {
    return XTile<BKM, ROWS>::frag(reinterpret_cast<const float*>(lds), p, rbase, ks, lane);
  }
};
struct PH { const float* A; int lda; };
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void x6p_v3(P p, PH h) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using TA = XA<BM>;
  using TB = PTile<BN>;
  __shared__ __attribute__((aligned(16))) char lds[TA::LDS_BYTES + TB::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, p.nblk);
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  TA ta; TB tb;
  ta.rows_init(h.lda, m0, p.M, nullptr, tid);
  tb.init(p.B, p.K, n0, p.N, tid);
  const int nt = p.K / 32;
  ta.template load<false>(h.A, h.lda, m0, p.M, 0, p.K, tid); tb.load(0);
  ta.store(reinterpret_cast<float*>(lds), tid); tb.store(lds + TA::LDS_BYTES, tid);
  __syncthreads();
  for (int t = 0; t + 1 < nt; ++t) {
    ta.template load<false>(h.A, h.lda, m0, p.M, (t + 1) * 32, p.K, tid); tb.load(t + 1);
    mma_tile<TM, TN, TA, TB>(acc, lds, lds + TA::LDS_BYTES, wm * (BM / WM), wn * (BN / WN), lane);
    __syncthreads();
    ta.store(reinterpret_cast<float*>(lds), tid); tb.store(lds + TA::LDS_BYTES, tid);
    __syncthreads();
  }
  mma_tile<TM, TN, TA, TB>(acc, lds, lds + TA::LDS_BYTES, wm * (BM / WM), wn * (BN / WN), lane);
  epilogue<TM, TN>(acc, p, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

// ---------------- v4: in-workgroup split-K ("wsk"): fp32 operands split in the loop as in-tree; BM x BN tile, 512 threads =
// two groups of four waves, group g takes the k-tiles g, g+2, .. with its own staging area; the two partial tiles meet in LDS ----
template <int BM, int BN, int WM, int WN, int BKM>
__global__ __launch_bounds__(512, 1) void x6_wsk(P p, PH h, const float* Bf, int ldb) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(WM * WN == 4, "four waves per group");
  using TA = XA<BM>;
  using TB = XB<BKM, BN>;
  constexpr int GROUP = TA::LDS_BYTES + TB::LDS_BYTES;
  constexpr int REDB = BM * BN * 4;
  constexpr int LDSB = 2 * GROUP > REDB ? 2 * GROUP : REDB;
  __shared__ __attribute__((aligned(16))) char lds[LDSB];
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, p.nblk);
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  char* As = lds + grp * GROUP;
  char* Bs = As + TA::LDS_BYTES;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  TA ta; TB tb;
  ta.rows_init(h.lda, m0, p.M, nullptr, tid);
  tb.rows_init(ldb, n0, p.N, nullptr, tid);
  const int nt = p.K / 64;     // k-tile pairs (K % 64 == 0 in the lab)
  ta.template load<false>(h.A, h.lda, m0, p.M, grp * 32, p.K, tid);
  tb.template load<false>(Bf, ldb, n0, p.N, grp * 32, p.K, tid);
  ta.store(reinterpret_cast<float*>(As), tid); tb.store(reinterpret_cast<float*>(Bs), tid);
  __syncthreads();
  for (int t = 0; t + 1 < nt; ++t) {
    ta.template load<false>(h.A, h.lda, m0, p.M, (t + 1) * 64 + grp * 32, p.K, tid);
    tb.template load<false>(Bf, ldb, n0, p.N, (t + 1) * 64 + grp * 32, p.K, tid);
    mma_tile<TM, TN, TA, TB>(acc, As, Bs, wm * (BM / WM), wn * (BN / WN), lane);
    __syncthreads();
    ta.store(reinterpret_cast<float*>(As), tid); tb.store(reinterpret_cast<float*>(Bs), tid);
    __syncthreads();
  }
  mma_tile<TM, TN, TA, TB>(acc, As, Bs, wm * (BM / WM), wn * (BN / WN), lane);
  __syncthreads();
  // group 1 parks its partial tile in LDS ([wave][i][j][reg][lane]), group 0 adds it and stores
  float* red = reinterpret_cast<float*>(lds);
  if (grp == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((wave * TM + i) * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
  }
  __syncthreads();
  if (grp == 1) return;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] += red[(((wave * TM + i) * TN + j) * 16 + r) * 64 + lane];
  epilogue<TM, TN>(acc, p, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

}  // namespace

extern "C" int x6_wsk_gemm(int variant, const float* A, int lda, const float* B, int ldb, int b_kmajor, float* C, int M, int N, int K, void* stream) {
  P p; p.A = nullptr; p.B = nullptr; p.C = C; p.M = M; p.N = N; p.K = K; p.ldc = N;
  PH h; h.A = A; h.lda = lda;
  hipStream_t st = (hipStream_t)stream;
  auto grid = [&](int BM, int BN) { p.tiles_n = (N + BN - 1) / BN; p.nblk = ((M + BM - 1) / BM) * p.tiles_n; return dim3(p.nblk); };
  if (variant == 0 && !b_kmajor) { dim3 g = grid(128, 64); hipLaunchKernelGGL((x6_wsk<128, 64, 2, 2, 0>), g, dim3(512), 0, st, p, h, B, ldb); }
  else if (variant == 0) { dim3 g = grid(128, 64); hipLaunchKernelGGL((x6_wsk<128, 64, 2, 2, 1>), g, dim3(512), 0, st, p, h, B, ldb); }
  else if (variant == 1 && !b_kmajor) { dim3 g = grid(64, 128); hipLaunchKernelGGL((x6_wsk<64, 128, 1, 4, 0>), g, dim3(512), 0, st, p, h, B, ldb); }
  else if (variant == 1) { dim3 g = grid(64, 128); hipLaunchKernelGGL((x6_wsk<64, 128, 1, 4, 1>), g, dim3(512), 0, st, p, h, B, ldb); }
  else if (variant == 2 && !b_kmajor) { dim3 g = grid(128, 128); hipLaunchKernelGGL((x6_wsk<128, 128, 2, 2, 0>), g, dim3(512), 0, st, p, h, B, ldb); }
  else if (variant == 2) { dim3 g = grid(128, 128); hipLaunchKernelGGL((x6_wsk<128, 128, 2, 2, 1>), g, dim3(512), 0, st, p, h, B, ldb); }
  else return -1;
  return (int)hipGetLastError();
}

extern "C" int x6p_gemm_hybrid(int variant, const float* A, int lda, const void* B3, float* C, int M, int N, int K, void* stream) {
  P p; p.A = nullptr; p.B = (const uint4*)B3; p.C = C; p.M = M; p.N = N; p.K = K; p.ldc = N;
  PH h; h.A = A; h.lda = lda;
  hipStream_t st = (hipStream_t)stream;
  auto grid = [&](int BM, int BN) { p.tiles_n = (N + BN - 1) / BN; p.nblk = ((M + BM - 1) / BM) * p.tiles_n; return dim3(p.nblk); };
  switch (variant) {
    case 0: { dim3 g = grid(128, 128); hipLaunchKernelGGL((x6p_v3<128, 128, 2, 2>), g, dim3(256), 0, st, p, h); break; }
    case 1: { dim3 g = grid(64, 128); hipLaunchKernelGGL((x6p_v3<64, 128, 1, 4>), g, dim3(256), 0, st, p, h); break; }
    case 2: { dim3 g = grid(64, 64); hipLaunchKernelGGL((x6p_v3<64, 64, 2, 2>), g, dim3(256), 0, st, p, h); break; }
    default: return -1;
  }
  return (int)hipGetLastError();
}

extern "C" int x6p_split(const float* X, int ld, int R, int K, void* out, void* stream) {
  hipLaunchKernelGGL(split_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, X, ld, R, K, (uint4*)out);
  return (int)hipGetLastError();
}

extern "C" int x6p_gemm(int variant, const void* A3, const void* B3, float* C, int M, int N, int K, void* stream) {
  P p; p.A = (const uint4*)A3; p.B = (const uint4*)B3; p.C = C; p.M = M; p.N = N; p.K = K; p.ldc = N;
  hipStream_t st = (hipStream_t)stream;
  auto grid = [&](int BM, int BN) { p.tiles_n = (N + BN - 1) / BN; p.nblk = ((M + BM - 1) / BM) * p.tiles_n; return dim3(p.nblk); };
  switch (variant) {
    case 0: { dim3 g = grid(128, 128); hipLaunchKernelGGL((x6p_v1<128, 128, 2, 2>), g, dim3(256), 0, st, p); break; }
    case 1: { dim3 g = grid(64, 128); hipLaunchKernelGGL((x6p_v1<64, 128, 1, 4>), g, dim3(256), 0, st, p); break; }
    case 2: { dim3 g = grid(64, 64); hipLaunchKernelGGL((x6p_v1<64, 64, 2, 2>), g, dim3(256), 0, st, p); break; }
    case 3: {
      dim3 g = grid(256, 128);
      constexpr int LDSB = 2 * 384 * 192;
      static bool once = false;
      if (!once) { (void)hipFuncSetAttribute((const void*)x6p_v2<256, 128, 4, 2, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); once = true; }
      hipLaunchKernelGGL((x6p_v2<256, 128, 4, 2, 512>), g, dim3(512), LDSB, st, p);
      break;
    }
    case 4: {
      dim3 g = grid(128, 128);
      constexpr int LDSB = 2 * 256 * 192;
      static bool once = false;
      if (!once) { (void)hipFuncSetAttribute((const void*)x6p_v2<128, 128, 2, 2, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); once = true; }
      hipLaunchKernelGGL((x6p_v2<128, 128, 2, 2, 256>), g, dim3(256), LDSB, st, p);
      break;
    }
    default: return -1;
  }
  return (int)hipGetLastError();
}
