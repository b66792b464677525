// Shared declarations for the D-VAE gfx950 library (host + device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dvae_hip.h"

#define DVAE_MAX_N 128      // max nodes per graph (adjacency rows kept as 128-bit masks)
#define DVAE_MAX_C 32       // max node types
#define DVAE_BITW 4         // 32-bit words per adjacency bit-row

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int64_t round_up64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// last error text (thread-local), read through dvae_last_error()
void dvae_set_error(const char* fmt, ...);

#define DVAE_CHECK_HIP(expr)                                                          \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      dvae_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                     __LINE__);                                                       \
      return DVAE_ERR_HIP;                                                            \
    }                                                                                 \
  } while (0)

#define DVAE_REQUIRE(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      dvae_set_error(__VA_ARGS__);   \
      return DVAE_ERR_ARG;           \
    }                                \
  } while (0)

#define DVAE_TRY(expr)              \
  do {                              \
    int s_ = (expr);                \
    if (s_ != DVAE_OK) return s_;   \
  } while (0)

#define DVAE_LAUNCH_CHECK() DVAE_CHECK_HIP(hipGetLastError())

// ---------------------------------------------------------------------------
// internal launchers shared between translation units
// ---------------------------------------------------------------------------
// split-K partial sums left for the consumer kernel to add up (GemmDesc::partials): slab s of the product is
// slab + s * stride, [M][N] with leading dimension N; slab == NULL: the product was written to C whole
struct GemmPartials { const float* slab; int splits; long long stride; };

struct GemmDesc {
  const float* A; int lda; int a_kmajor;   // a_kmajor=0: A[M][K]; 1: A[K][M]
  const float* B; int ldb; int b_kmajor;   // b_kmajor=0: B[N][K]; 1: B[K][N]
  float* C; int ldc;
  int M, N, K;
  const float* bias;   // per column n, or null
  int relu;            // apply max(0,.) after bias
  int accumulate;      // C += result
  int prio;            // 1: waves run at raised priority (launches on the per-level chain)
  int bulk;            // 1: side-stream product: a split-K launch takes ONE workgroup per CU, so
                       //    that chain launches arriving later still find free slots on every CU
  const int* row_map;  // optional (m-major A only): row m of the product reads A row row_map[m] and
                       // writes C row row_map[m] -- a product over a scattered subset of rows
  const int* k_map;    // optional (both operands k-major): reduction index k reads memory row k_map[k]
                       // of A and of B -- a weight gradient summed over a scattered subset of rows
  GemmPartials* partials;   // optional out (no bias / relu): when the product is split over K, skip the combine
                            // launch and describe the slabs here; the consumer sums them in slab order (the
                            // combine's order), then adds what C holds if `accumulate` -- C is left untouched --
                            // before the next product on this stream reuses ws
};
// C[M][N] (+)= sum_k A(m,k) * B(n,k).  ws/ws_bytes: scratch for split-K (may be null).
int gemm_f32(const GemmDesc& d, float* ws, size_t ws_bytes, hipStream_t stream);

// arithmetic core in use (gemm.hip): 1 = fp32 products as six bf16 MFMAs on split operands, 0 = fp32 MFMA
int gemm_x6_enabled();

// bench.py's per-launch timing hooks (gemm.hip); begin returns NULL when profiling is off
void* gemm_prof_begin(double flops, hipStream_t st);
void gemm_prof_end(void* h, hipStream_t st);
// the same log for the HBM-bound kernels (dvae_hbm_profile_read): tag 4 pair scorer forward, 5 pair scorer backward,
// 6 aggregation forward (agg_fwd, prefix_gru_fwd), 7 aggregation backward (dgm_bwd, dgm_gather); bytes = the launch's
// ALGORITHMIC bytes.  Ends with gemm_prof_end.
void* kernel_prof_begin(int tag, double bytes, hipStream_t st);
struct KernelProf {
  void* h; hipStream_t st;
  KernelProf(int tag, double bytes, hipStream_t s) : h(kernel_prof_begin(tag, bytes, s)), st(s) {}
  ~KernelProf() { gemm_prof_end(h, st); }
  KernelProf(const KernelProf&) = delete;
};

// loss.hip: resolve dvae_elbo_static's per-device array address on the current device (set-up time)
int elbo_static_warm();
// d_mu += d_z, d_logvar += d_z . eps . 0.5 exp(0.5 logvar): dvae_reparam_bwd and the two additions behind it in one launch
int k_reparam_bwd_acc(const float* d_z, const float* logvar, const float* eps, float* d_mu, float* d_logvar, long long n,
                      hipStream_t st);
