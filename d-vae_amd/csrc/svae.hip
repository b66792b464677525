// S-VAE baseline (dvae/models/svae_model.py:13-301) on the library's fp32 MFMA GEMM: sequence GRUs
// with time-major tables, the two MLP heads, the losses and the decision rule.
//
// Row order of every [T*B][.] table: row = step * B + b, so one step's rows are contiguous and the
// step's hidden product is one GEMM over B rows; whole-sequence products (input projection, heads,
// every weight gradient) are single GEMMs over T*B rows.
#include <algorithm>
#include <vector>

#include "kernels.h"
#include "svae_layout.h"

namespace {

constexpr size_t SVAE_GEMM_WS = (size_t)64 << 20;    // split-K scratch
constexpr size_t SVAE_PART_FLOATS = (size_t)1 << 20;  // column-sum partials

struct Carver {
  char* p; size_t off = 0;
  float* take(size_t floats) {
    float* r = p ? reinterpret_cast<float*>(p + off) : nullptr;
    off += round_up64((int64_t)floats * 4, 256);
    return r;
  }
};

struct GruTables {
  float* GI;     // [TB][3HP] input projection (time-major) -> d(input projection) after backward
  float* HALL;   // [(T+1)B][HP] state before each step; slab T = final state
  float* GATES;  // [TB][3HP] (step-major) hidden product -> (r,z,n) -> d(hidden product)
  float* GHN;    // [TB][HP]  W_hn h + b_hn
};

struct EncWs {
  float* XT;     // [TB][IP] input rows, time-major, zero-padded
  GruTables g[2];
  float* CAT;    // [B][2HP] final states of both directions (bidir)
  float* HN;     // [B][HP]  h_n after hg_unify (bidir) / alias of the final state
  float* ML;     // [B][2ZP] mu | logvar -> their gradients
  float* DHN;    // [B][HP]
  float* DCAT;   // [B][2HP]
  float* CARRY;  // [B][HP]
  float* gemm_ws; float* part;
  float* gemm_ws2; float* part2;   // the side stream's scratch
  size_t bytes;
};

void carve_gru(Carver& c, const dvae_svae_dims& d, GruTables& g, bool own_gi, size_t gi_rows) {
  const size_t TB = (size_t)d.T * d.B;
  g.GI = own_gi ? c.take(gi_rows * 3 * d.HP) : nullptr;
  g.HALL = c.take((TB + d.B) * d.HP);
  g.GATES = c.take(TB * 3 * d.HP);
  g.GHN = c.take(TB * d.HP);
}

EncWs enc_ws(const dvae_svae_dims& d, void* base) {
  EncWs w;
  Carver c{static_cast<char*>(base)};
  const size_t TB = (size_t)d.T * d.B;
  w.XT = c.take(TB * d.IP);
  for (int dir = 0; dir < (d.bidir ? 2 : 1); ++dir) carve_gru(c, d, w.g[dir], true, TB);
  if (!d.bidir) w.g[1] = GruTables{nullptr, nullptr, nullptr, nullptr};
  w.CAT = c.take((size_t)d.B * 2 * d.HP);
  w.HN = c.take((size_t)d.B * d.HP);
  w.ML = c.take((size_t)d.B * 2 * d.ZP);
  w.DHN = c.take((size_t)d.B * d.HP);
  w.DCAT = c.take((size_t)d.B * 2 * d.HP);
  w.CARRY = c.take((size_t)d.B * d.HP);
  w.gemm_ws = c.take(SVAE_GEMM_WS / 4);
  w.part = c.take(SVAE_PART_FLOATS);
  w.gemm_ws2 = c.take(SVAE_GEMM_WS / 4);
  w.part2 = c.take(SVAE_PART_FLOATS);
  w.bytes = c.off;
  return w;
}

struct DecWs {
  float* ZT;     // [B][ZP]
  float* H0;     // [B][HP] relu(fc_h0 z): the GRU's input at every step
  float* GI0;    // [B][3HP] its input projection (shared by all steps) -> summed gradient
  GruTables g;   // g.GI = [TB][3HP] per-step d(input projection), backward only
  float* HID;    // [TB][2HP] relu(add_vertex.0 h) | relu(add_edges.0 h)
  float* DHID;   // [TB][2HP]
  float* TS;     // [TB][32] type scores -> their gradients
  float* ES;     // [TB][32] edge scores -> their gradients
  float* DHOUT;  // [TB][HP]
  float* CARRY;  // [B][HP]
  float* DH0;    // [B][HP]
  float* DZT;    // [B][ZP]
  float* gemm_ws; float* part;
  float* gemm_ws2; float* part2;   // the side stream's scratch
  size_t bytes;
};

DecWs dec_ws(const dvae_svae_dims& d, void* base) {
  DecWs w;
  Carver c{static_cast<char*>(base)};
  const size_t TB = (size_t)d.T * d.B;
  w.ZT = c.take((size_t)d.B * d.ZP);
  w.H0 = c.take((size_t)d.B * d.HP);
  w.GI0 = c.take((size_t)d.B * 3 * d.HP);
  carve_gru(c, d, w.g, true, TB);
  w.HID = c.take(TB * 2 * d.HP);
  w.DHID = c.take(TB * 2 * d.HP);
  w.TS = c.take(TB * 32);
  w.ES = c.take(TB * 32);
  w.DHOUT = c.take(TB * d.HP);
  w.CARRY = c.take((size_t)d.B * d.HP);
  w.DH0 = c.take((size_t)d.B * d.HP);
  w.DZT = c.take((size_t)d.B * d.ZP);
  w.gemm_ws = c.take(SVAE_GEMM_WS / 4);
  w.part = c.take(SVAE_PART_FLOATS);
  w.gemm_ws2 = c.take(SVAE_GEMM_WS / 4);
  w.part2 = c.take(SVAE_PART_FLOATS);
  w.bytes = c.off;
  return w;
}

// `to` continues after what `from` has queued so far (events from a small per-thread ring)
struct EventRing {
  std::vector<hipEvent_t> ev;
  size_t next = 0;
  int edge(hipStream_t from, hipStream_t to) {
    if (from == to) return DVAE_OK;
    if (ev.size() < 64) {
      hipEvent_t e;
      DVAE_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ev.push_back(e);
      next = ev.size() - 1;
    } else {
      next = (next + 1) % ev.size();
    }
    DVAE_CHECK_HIP(hipEventRecord(ev[next], from));
    DVAE_CHECK_HIP(hipStreamWaitEvent(to, ev[next], 0));
    return DVAE_OK;
  }
};
thread_local EventRing g_ring;

struct G {
  float* ws; hipStream_t st;
  int nt(const float* X, int ldx, const float* W, int ldw, float* Y, int ldy, int M, int N, int K,
         const float* bias = nullptr, int relu = 0) const {
    GemmDesc d{X, ldx, 0, W, ldw, 0, Y, ldy, M, N, K, bias, relu, 0, 0, 0, nullptr, nullptr};
    return gemm_f32(d, ws, SVAE_GEMM_WS, st);
  }
  int nn(const float* dY, int ldy, const float* W, int ldw, float* dX, int ldx, int M, int N, int K, int acc = 0) const {
    GemmDesc d{dY, ldy, 0, W, ldw, 1, dX, ldx, M, N, K, nullptr, 0, acc, 0, 0, nullptr, nullptr};
    return gemm_f32(d, ws, SVAE_GEMM_WS, st);
  }
  int tn(const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, int M, int N, int K) const {
    GemmDesc d{dY, ldy, 1, X, ldx, 1, dW, ldw, M, N, K, nullptr, 0, 0, 0, 0, nullptr, nullptr};
    return gemm_f32(d, ws, SVAE_GEMM_WS, st);
  }
};

inline dim3 grid_of(long long items) { return dim3((unsigned)std::min<long long>((items + 255) / 256, 1 << 20)); }

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

// [B][T][W] (batch-first argument) <-> [T*B][ld] time-major table, zero-padded to `cols`
__global__ void bt_to_tm_kernel(const float* __restrict__ in, int B, int T, int W, int col0, int win,
                                float* __restrict__ out, int ld, int cols) {
  const long long total = (long long)T * B * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    const long long row = i / cols;
    const int b = (int)(row % B), t = (int)(row / B);
    out[row * ld + c] = (in && c < W) ? in[((size_t)b * T + t) * win + col0 + c] : 0.f;
  }
}
__global__ void tm_to_bt_kernel(const float* __restrict__ in, int ld, int B, int T, int W, float* __restrict__ out) {
  const long long total = (long long)T * B * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % W);
    const long long r = i / W;          // b * T + t
    const int t = (int)(r % T), b = (int)(r / T);
    out[i] = in[((size_t)t * B + b) * ld + c];
  }
}

// GRU step (torch.nn.GRU, svae_model.py:24-26,58): r,z = sigmoid(gi + gh), n = tanh(gi_n + r * gh_n),
// h' = (1 - z) n + z h.  GH holds the hidden product on entry and (r,z,n) on return.
__global__ void seq_gru_fwd_kernel(int B, int HP, const float* __restrict__ GI, float* __restrict__ GH,
                                   const float* __restrict__ Hp, float* __restrict__ Hn_, float* __restrict__ GHN) {
  const int q = HP / 4;
  const long long total = (long long)B * q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / q), j = (int)(i % q) * 4;
    const float* gi = GI + (size_t)b * 3 * HP + j;
    float* gh = GH + (size_t)b * 3 * HP + j;
    const float4 ir = *reinterpret_cast<const float4*>(gi), iz = *reinterpret_cast<const float4*>(gi + HP),
                 in_ = *reinterpret_cast<const float4*>(gi + 2 * HP);
    const float4 hr = *reinterpret_cast<const float4*>(gh), hz = *reinterpret_cast<const float4*>(gh + HP),
                 hn = *reinterpret_cast<const float4*>(gh + 2 * HP);
    const float4 hp = *reinterpret_cast<const float4*>(Hp + (size_t)b * HP + j);
    float4 r, z, n, h;
#define SVAE_LANE(c)                                  \
    r.c = sigm(ir.c + hr.c); z.c = sigm(iz.c + hz.c); \
    n.c = tanhf(in_.c + r.c * hn.c);                  \
    h.c = (1.f - z.c) * n.c + z.c * hp.c;
    SVAE_LANE(x) SVAE_LANE(y) SVAE_LANE(z) SVAE_LANE(w)
#undef SVAE_LANE
    *reinterpret_cast<float4*>(gh) = r;
    *reinterpret_cast<float4*>(gh + HP) = z;
    *reinterpret_cast<float4*>(gh + 2 * HP) = n;
    *reinterpret_cast<float4*>(GHN + (size_t)b * HP + j) = hn;
    *reinterpret_cast<float4*>(Hn_ + (size_t)b * HP + j) = h;
  }
}

// its backward: dh = DHout (optional) + CARRY; writes d(input projection) to DGI, d(hidden product)
// over (r,z,n) in GATES, and the direct part dh*z of the previous state's gradient to CARRY
__global__ void seq_gru_bwd_kernel(int B, int HP, const float* __restrict__ DHout, float* __restrict__ CARRY,
                                   float* __restrict__ GATES, const float* __restrict__ GHN,
                                   const float* __restrict__ Hp, float* __restrict__ DGI) {
  const int q = HP / 4;
  const long long total = (long long)B * q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / q), j = (int)(i % q) * 4;
    float* ga = GATES + (size_t)b * 3 * HP + j;
    float* dgi = DGI + (size_t)b * 3 * HP + j;
    const float4 r = *reinterpret_cast<const float4*>(ga), z = *reinterpret_cast<const float4*>(ga + HP),
                 n = *reinterpret_cast<const float4*>(ga + 2 * HP);
    const float4 hn = *reinterpret_cast<const float4*>(GHN + (size_t)b * HP + j);
    const float4 hp = *reinterpret_cast<const float4*>(Hp + (size_t)b * HP + j);
    float4 dh = *reinterpret_cast<const float4*>(CARRY + (size_t)b * HP + j);
    if (DHout) {
      const float4 o = *reinterpret_cast<const float4*>(DHout + (size_t)b * HP + j);
      dh.x += o.x; dh.y += o.y; dh.z += o.z; dh.w += o.w;
    }
    float4 dr, dz, dn, dnr, c;
#define SVAE_LANE(k)                                        \
    dn.k = dh.k * (1.f - z.k) * (1.f - n.k * n.k);          \
    dz.k = dh.k * (hp.k - n.k) * z.k * (1.f - z.k);         \
    dr.k = dn.k * hn.k * r.k * (1.f - r.k);                 \
    dnr.k = dn.k * r.k;                                     \
    c.k = dh.k * z.k;
    SVAE_LANE(x) SVAE_LANE(y) SVAE_LANE(z) SVAE_LANE(w)
#undef SVAE_LANE
    *reinterpret_cast<float4*>(dgi) = dr;
    *reinterpret_cast<float4*>(dgi + HP) = dz;
    *reinterpret_cast<float4*>(dgi + 2 * HP) = dn;
    *reinterpret_cast<float4*>(ga) = dr;
    *reinterpret_cast<float4*>(ga + HP) = dz;
    *reinterpret_cast<float4*>(ga + 2 * HP) = dnr;
    *reinterpret_cast<float4*>(CARRY + (size_t)b * HP + j) = c;
  }
}

// out[i] = sum over s of X[s * slab + i]   (fixed order)
__global__ void sum_slabs_kernel(const float* __restrict__ X, size_t slab, int S, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slab; i += (size_t)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += X[(size_t)s * slab + i];
    out[i] = a;
  }
}

// ML [B][2ZP] <- d_mu | d_logvar (NULL = 0), zero-padded;  and back out of the forward table
__global__ void ml_pack_kernel(const float* __restrict__ a, const float* __restrict__ b, int B, int Z, int ZP,
                               float* __restrict__ ML) {
  const long long total = (long long)B * 2 * ZP;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % (2 * ZP)), r = (int)(i / (2 * ZP));
    const float* src = c < ZP ? a : b;
    const int k = c < ZP ? c : c - ZP;
    ML[i] = (src && k < Z) ? src[(size_t)r * Z + k] : 0.f;
  }
}
__global__ void ml_unpack_kernel(const float* __restrict__ ML, int B, int Z, int ZP, float* __restrict__ mu,
                                 float* __restrict__ lv) {
  const long long total = (long long)B * Z;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Z), r = (int)(i / Z);
    mu[i] = ML[(size_t)r * 2 * ZP + k];
    lv[i] = ML[(size_t)r * 2 * ZP + ZP + k];
  }
}

int gru_seq_fwd(const dvae_svae_dims& d, const G& g, const float* P, const SvaeGru& w, const GruTables& t,
                const float* GI, int gi_per_step, bool reverse) {
  const int B = d.B, HP = d.HP;
  DVAE_TRY(k_zero(t.HALL, (size_t)B * HP * 4, g.st));
  for (int s = 0; s < d.T; ++s) {
    const int tt = reverse ? d.T - 1 - s : s;
    float* gates = t.GATES + (size_t)s * B * 3 * HP;
    const float* hp = t.HALL + (size_t)s * B * HP;
    DVAE_TRY(g.nt(hp, HP, P + w.Whh, HP, gates, 3 * HP, B, 3 * HP, HP, P + w.bhh));
    hipLaunchKernelGGL(seq_gru_fwd_kernel, grid_of((long long)B * HP / 4), dim3(256), 0, g.st, B, HP,
                       GI + (gi_per_step ? (size_t)tt * B * 3 * HP : 0), gates, hp,
                       t.HALL + (size_t)(s + 1) * B * HP, t.GHN + (size_t)s * B * HP);
    DVAE_LAUNCH_CHECK();
  }
  return DVAE_OK;
}

// CARRY holds dL/d(final state) on entry; DHout (optional, step-major) adds a gradient per step
int gru_seq_bwd(const dvae_svae_dims& d, const G& g, const float* P, const SvaeGru& w, const GruTables& t,
                const float* DHout, float* CARRY, bool reverse) {
  const int B = d.B, HP = d.HP;
  for (int s = d.T - 1; s >= 0; --s) {
    const int tt = reverse ? d.T - 1 - s : s;
    float* gates = t.GATES + (size_t)s * B * 3 * HP;
    hipLaunchKernelGGL(seq_gru_bwd_kernel, grid_of((long long)B * HP / 4), dim3(256), 0, g.st, B, HP,
                       DHout ? DHout + (size_t)s * B * HP : nullptr, CARRY, gates, t.GHN + (size_t)s * B * HP,
                       t.HALL + (size_t)s * B * HP, t.GI + (size_t)tt * B * 3 * HP);
    DVAE_LAUNCH_CHECK();
    if (s > 0) DVAE_TRY(g.nn(gates, 3 * HP, P + w.Whh, HP, CARRY, HP, B, HP, 3 * HP, 1));
  }
  return DVAE_OK;
}

// weight gradients of one GRU: the hidden half (states x d(hidden product), both step-major) ...
int gru_wgrad_hh(const dvae_svae_dims& d, const G& g, const SvaeGru& w, const GruTables& t, float* Gp, float* part) {
  const int HP = d.HP, TB = d.T * d.B;
  DVAE_TRY(g.tn(t.GATES, 3 * HP, t.HALL, HP, Gp + w.Whh, HP, 3 * HP, HP, TB));
  DVAE_TRY(k_colsum(t.GATES, 3 * HP, TB, 3 * HP, Gp + w.bhh, part, SVAE_PART_FLOATS, g.st));
  return DVAE_OK;
}
// ... and the input half: input rows X [rows_x][KP] paired with DGI
int gru_wgrad_ih(const dvae_svae_dims& d, const G& g, const SvaeGru& w, const float* DGI, const float* X, int KP,
                 int rows_x, float* Gp, float* part) {
  const int HP = d.HP;
  DVAE_TRY(g.tn(DGI, 3 * HP, X, KP, Gp + w.Wih, KP, 3 * HP, KP, rows_x));
  DVAE_TRY(k_colsum(DGI, 3 * HP, rows_x, 3 * HP, Gp + w.bih, part, SVAE_PART_FLOATS, g.st));
  return DVAE_OK;
}

int check_dims(const dvae_svae_dims* d) {
  DVAE_REQUIRE(d, "svae: null dims");
  DVAE_REQUIRE(d->HP >= d->H && d->HP % 64 == 0 && d->ZP >= d->Z && d->ZP % 64 == 0 && d->IP >= d->I &&
               d->IP % 32 == 0 && d->I == d->C + d->N && d->B > 0 && d->T > 0 && d->C <= 32 && d->N <= 32,
               "svae: dims not initialised (dvae_svae_dims_init)");
  return DVAE_OK;
}

}  // namespace

extern "C" size_t dvae_svae_enc_ws_bytes(const dvae_svae_dims* d) { return enc_ws(*d, nullptr).bytes; }
extern "C" size_t dvae_svae_dec_ws_bytes(const dvae_svae_dims* d) { return dec_ws(*d, nullptr).bytes; }

extern "C" int dvae_svae_enc_fwd(const dvae_svae_dims* dp, const float* x, const float* P, void* ws, float* mu,
                                 float* logvar, void* stream) {
  DVAE_TRY(check_dims(dp));
  DVAE_REQUIRE(x && P && ws && mu && logvar, "svae_enc_fwd: null pointer");
  const dvae_svae_dims& d = *dp;
  hipStream_t st = (hipStream_t)stream;
  const EncWs w = enc_ws(d, ws);
  const SvaeEncLayout L = svae_enc_layout(d);
  const G g{w.gemm_ws, st};
  const int B = d.B, HP = d.HP, TB = d.T * d.B;
  hipLaunchKernelGGL(bt_to_tm_kernel, grid_of((long long)TB * d.IP), dim3(256), 0, st, x, B, d.T, d.I, 0, d.I, w.XT,
                     d.IP, d.IP);
  DVAE_LAUNCH_CHECK();
  for (int dir = 0; dir < (d.bidir ? 2 : 1); ++dir) {
    const SvaeGru& gw = L.gru[dir];
    DVAE_TRY(g.nt(w.XT, d.IP, P + gw.Wih, d.IP, w.g[dir].GI, 3 * HP, TB, 3 * HP, d.IP, P + gw.bih));
    DVAE_TRY(gru_seq_fwd(d, g, P, gw, w.g[dir], w.g[dir].GI, 1, dir == 1));
  }
  const float* hn = w.g[0].HALL + (size_t)TB * HP;
  if (d.bidir) {   // h_n = hg_unify([h_fwd ; h_bwd]) (svae_model.py:38-41)
    DVAE_CHECK_HIP(hipMemcpy2DAsync(w.CAT, (size_t)2 * HP * 4, hn, (size_t)HP * 4, (size_t)HP * 4, B,
                                    hipMemcpyDeviceToDevice, st));
    DVAE_CHECK_HIP(hipMemcpy2DAsync(w.CAT + HP, (size_t)2 * HP * 4, w.g[1].HALL + (size_t)TB * HP, (size_t)HP * 4,
                                    (size_t)HP * 4, B, hipMemcpyDeviceToDevice, st));
    DVAE_TRY(g.nt(w.CAT, 2 * HP, P + L.Wun, 2 * HP, w.HN, HP, B, HP, 2 * HP, P + L.bun));
    hn = w.HN;
  }
  DVAE_TRY(g.nt(hn, HP, P + L.Wml, HP, w.ML, 2 * d.ZP, B, 2 * d.ZP, HP, P + L.bml));
  hipLaunchKernelGGL(ml_unpack_kernel, grid_of((long long)B * d.Z), dim3(256), 0, st, w.ML, B, d.Z, d.ZP, mu, logvar);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

extern "C" int dvae_svae_enc_bwd(const dvae_svae_dims* dp, const float* P, void* ws, const float* d_mu,
                                 const float* d_logvar, float* Gp, void* stream, void* side_stream) {
  DVAE_TRY(check_dims(dp));
  DVAE_REQUIRE(P && ws && Gp, "svae_enc_bwd: null pointer");
  const dvae_svae_dims& d = *dp;
  hipStream_t st = (hipStream_t)stream;
  hipStream_t sd = side_stream ? (hipStream_t)side_stream : st;   // weight gradients run beside the chain
  const EncWs w = enc_ws(d, ws);
  const SvaeEncLayout L = svae_enc_layout(d);
  const G g{w.gemm_ws, st};
  const G gs{sd == st ? w.gemm_ws : w.gemm_ws2, sd};
  float* spart = sd == st ? w.part : w.part2;
  const int B = d.B, HP = d.HP, ZP = d.ZP, TB = d.T * d.B;
  const float* hn = d.bidir ? w.HN : w.g[0].HALL + (size_t)TB * HP;
  hipLaunchKernelGGL(ml_pack_kernel, grid_of((long long)B * 2 * ZP), dim3(256), 0, st, d_mu, d_logvar, B, d.Z, ZP, w.ML);
  DVAE_LAUNCH_CHECK();
  DVAE_TRY(g_ring.edge(st, sd));
  DVAE_TRY(gs.tn(w.ML, 2 * ZP, hn, HP, Gp + L.Wml, HP, 2 * ZP, HP, B));
  DVAE_TRY(k_colsum(w.ML, 2 * ZP, B, 2 * ZP, Gp + L.bml, spart, SVAE_PART_FLOATS, sd));
  DVAE_TRY(g.nn(w.ML, 2 * ZP, P + L.Wml, HP, w.DHN, HP, B, HP, 2 * ZP));
  if (d.bidir) {
    DVAE_TRY(g_ring.edge(st, sd));
    DVAE_TRY(gs.tn(w.DHN, HP, w.CAT, 2 * HP, Gp + L.Wun, 2 * HP, HP, 2 * HP, B));
    DVAE_TRY(k_colsum(w.DHN, HP, B, HP, Gp + L.bun, spart, SVAE_PART_FLOATS, sd));
    DVAE_TRY(g.nn(w.DHN, HP, P + L.Wun, 2 * HP, w.DCAT, 2 * HP, B, 2 * HP, HP));
  }
  for (int dir = 0; dir < (d.bidir ? 2 : 1); ++dir) {
    const float* src = d.bidir ? w.DCAT + (size_t)dir * HP : w.DHN;
    DVAE_CHECK_HIP(hipMemcpy2DAsync(w.CARRY, (size_t)HP * 4, src, (size_t)(d.bidir ? 2 : 1) * HP * 4, (size_t)HP * 4,
                                    B, hipMemcpyDeviceToDevice, st));
    DVAE_TRY(gru_seq_bwd(d, g, P, L.gru[dir], w.g[dir], nullptr, w.CARRY, dir == 1));
    DVAE_TRY(g_ring.edge(st, sd));
    DVAE_TRY(gru_wgrad_hh(d, gs, L.gru[dir], w.g[dir], Gp, spart));
    DVAE_TRY(gru_wgrad_ih(d, gs, L.gru[dir], w.g[dir].GI, w.XT, d.IP, TB, Gp, spart));
  }
  return DVAE_OK;
}

extern "C" int dvae_svae_dec_fwd(const dvae_svae_dims* dp, const float* z, const float* P, void* ws,
                                 float* type_scores, float* edge_scores, void* stream) {
  DVAE_TRY(check_dims(dp));
  DVAE_REQUIRE(z && P && ws && type_scores && edge_scores, "svae_dec_fwd: null pointer");
  const dvae_svae_dims& d = *dp;
  hipStream_t st = (hipStream_t)stream;
  const DecWs w = dec_ws(d, ws);
  const SvaeDecLayout L = svae_dec_layout(d);
  const G g{w.gemm_ws, st};
  const int B = d.B, HP = d.HP, TB = d.T * d.B;
  DVAE_TRY(k_pad_rows(z, B, d.Z, w.ZT, d.ZP, d.ZP, st));
  DVAE_TRY(g.nt(w.ZT, d.ZP, P + L.Wh0, d.ZP, w.H0, HP, B, HP, d.ZP, P + L.bh0, 1));          // :74
  DVAE_TRY(g.nt(w.H0, HP, P + L.gru.Wih, HP, w.GI0, 3 * HP, B, 3 * HP, HP, P + L.gru.bih));   // same input every step (:75)
  DVAE_TRY(gru_seq_fwd(d, g, P, L.gru, w.g, w.GI0, 0, false));
  const float* hout = w.g.HALL + (size_t)B * HP;
  DVAE_TRY(g.nt(hout, HP, P + L.W1, HP, w.HID, 2 * HP, TB, 2 * HP, HP, P + L.b1, 1));
  DVAE_TRY(g.nt(w.HID, 2 * HP, P + L.Wv2, HP, w.TS, 32, TB, 32, HP, P + L.bv2));
  DVAE_TRY(g.nt(w.HID + HP, 2 * HP, P + L.We2, HP, w.ES, 32, TB, 32, HP, P + L.be2));
  hipLaunchKernelGGL(tm_to_bt_kernel, grid_of((long long)TB * d.C), dim3(256), 0, st, w.TS, 32, B, d.T, d.C, type_scores);
  DVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(tm_to_bt_kernel, grid_of((long long)TB * d.N), dim3(256), 0, st, w.ES, 32, B, d.T, d.N, edge_scores);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

extern "C" int dvae_svae_dec_bwd(const dvae_svae_dims* dp, const float* P, void* ws, const float* d_type,
                                 const float* d_edge, float* Gp, float* d_z, void* stream, void* side_stream) {
  DVAE_TRY(check_dims(dp));
  DVAE_REQUIRE(P && ws && Gp, "svae_dec_bwd: null pointer");
  const dvae_svae_dims& d = *dp;
  hipStream_t st = (hipStream_t)stream;
  hipStream_t sd = side_stream ? (hipStream_t)side_stream : st;   // weight gradients run beside the chain
  const DecWs w = dec_ws(d, ws);
  const SvaeDecLayout L = svae_dec_layout(d);
  const G g{w.gemm_ws, st};
  const G gs{sd == st ? w.gemm_ws : w.gemm_ws2, sd};
  float* spart = sd == st ? w.part : w.part2;
  const int B = d.B, HP = d.HP, TB = d.T * d.B;
  const float* hout = w.g.HALL + (size_t)B * HP;
  hipLaunchKernelGGL(bt_to_tm_kernel, grid_of((long long)TB * 32), dim3(256), 0, st, d_type, B, d.T, d.C, 0, d.C, w.TS, 32, 32);
  DVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(bt_to_tm_kernel, grid_of((long long)TB * 32), dim3(256), 0, st, d_edge, B, d.T, d.N, 0, d.N, w.ES, 32, 32);
  DVAE_LAUNCH_CHECK();
  // heads (weight gradients on the side stream, the input-gradient chain on the caller's)
  DVAE_TRY(g_ring.edge(st, sd));
  DVAE_TRY(gs.tn(w.TS, 32, w.HID, 2 * HP, Gp + L.Wv2, HP, 32, HP, TB));
  DVAE_TRY(gs.tn(w.ES, 32, w.HID + HP, 2 * HP, Gp + L.We2, HP, 32, HP, TB));
  DVAE_TRY(k_colsum(w.TS, 32, TB, 32, Gp + L.bv2, spart, SVAE_PART_FLOATS, sd));
  DVAE_TRY(k_colsum(w.ES, 32, TB, 32, Gp + L.be2, spart, SVAE_PART_FLOATS, sd));
  DVAE_TRY(g.nn(w.TS, 32, P + L.Wv2, HP, w.DHID, 2 * HP, TB, HP, 32));
  DVAE_TRY(g.nn(w.ES, 32, P + L.We2, HP, w.DHID + HP, 2 * HP, TB, HP, 32));
  DVAE_TRY(k_relu_bwd_colsum(w.DHID, w.HID, 2 * HP, TB, 2 * HP, Gp + L.b1, w.part, SVAE_PART_FLOATS, st));
  DVAE_TRY(g_ring.edge(st, sd));
  DVAE_TRY(gs.tn(w.DHID, 2 * HP, hout, HP, Gp + L.W1, HP, 2 * HP, HP, TB));
  DVAE_TRY(g.nn(w.DHID, 2 * HP, P + L.W1, HP, w.DHOUT, HP, TB, HP, 2 * HP));
  // GRU through time
  DVAE_TRY(k_zero(w.CARRY, (size_t)B * HP * 4, st));
  DVAE_TRY(gru_seq_bwd(d, g, P, L.gru, w.g, w.DHOUT, w.CARRY, false));
  DVAE_TRY(g_ring.edge(st, sd));
  DVAE_TRY(gru_wgrad_hh(d, gs, L.gru, w.g, Gp, spart));
  hipLaunchKernelGGL(sum_slabs_kernel, grid_of((long long)B * 3 * HP), dim3(256), 0, st, w.g.GI, (size_t)B * 3 * HP,
                     d.T, w.GI0);
  DVAE_LAUNCH_CHECK();
  DVAE_TRY(g_ring.edge(st, sd));
  DVAE_TRY(gru_wgrad_ih(d, gs, L.gru, w.GI0, w.H0, HP, B, Gp, spart));
  // fc_h0
  DVAE_TRY(g.nn(w.GI0, 3 * HP, P + L.gru.Wih, HP, w.DH0, HP, B, HP, 3 * HP));
  DVAE_TRY(k_relu_bwd_colsum(w.DH0, w.H0, HP, B, HP, Gp + L.bh0, w.part, SVAE_PART_FLOATS, st));
  DVAE_TRY(g_ring.edge(st, sd));
  DVAE_TRY(gs.tn(w.DH0, HP, w.ZT, d.ZP, Gp + L.Wh0, d.ZP, HP, d.ZP, B));
  if (d_z) {
    DVAE_TRY(g.nn(w.DH0, HP, P + L.Wh0, d.ZP, w.DZT, d.ZP, B, d.ZP, HP));
    DVAE_TRY(k_unpad_rows(w.DZT, d.ZP, B, d.Z, d_z, st));
  }
  return DVAE_OK;
}

// ---- losses and decisions -------------------------------------------------------
namespace {

__device__ __forceinline__ double wave_sum_dd(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per string; partial[b] = {edge, vertex, kl}
__global__ __launch_bounds__(64) void svae_elbo_kernel(dvae_svae_dims d, const float* __restrict__ ts,
                                                       const float* __restrict__ es, const float* __restrict__ x,
                                                       const float* __restrict__ mu, const float* __restrict__ lv,
                                                       float beta, double* __restrict__ partial,
                                                       float* __restrict__ d_type, float* __restrict__ d_edge,
                                                       float* __restrict__ d_mu, float* __restrict__ d_lv) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int T = d.T, C = d.C, N = d.N, I = d.I, Z = d.Z;
  double se = 0.0, sv = 0.0, sk = 0.0;
  // BCE-with-logits (sum) against the parent flags (svae_model.py:167,171)
  for (int e = lane; e < T * N; e += 64) {
    const int t = e / N, j = e - t * N;
    const float l = es[(size_t)b * T * N + e], y = x[((size_t)b * T + t) * I + C + j];
    se += (double)(fmaxf(l, 0.f) - l * y + log1pf(expf(-fabsf(l))));
    if (d_edge) d_edge[(size_t)b * T * N + e] = 1.f / (1.f + expf(-l)) - y;
  }
  // cross entropy (sum) against argmax of the type columns (first maximum, svae_model.py:173-178)
  for (int t = lane; t < T; t += 64) {
    const float* s = ts + ((size_t)b * T + t) * C;
    const float* xr = x + ((size_t)b * T + t) * I;
    int target = 0;
    for (int c = 1; c < C; ++c) if (xr[c] > xr[target]) target = c;
    float m = s[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, s[c]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(s[c] - m);
    const float lse = m + logf(sum);
    sv += (double)(lse - s[target]);
    if (d_type)
      for (int c = 0; c < C; ++c) d_type[((size_t)b * T + t) * C + c] = expf(s[c] - lse) - (c == target ? 1.f : 0.f);
  }
  for (int k = lane; k < Z; k += 64) {   // :180
    const float m = mu[(size_t)b * Z + k], l = lv[(size_t)b * Z + k];
    const float ex = expf(l);
    sk += (double)(-0.5f * (1.f + l - m * m - ex));
    if (d_mu) {
      d_mu[(size_t)b * Z + k] = beta * m;
      d_lv[(size_t)b * Z + k] = beta * 0.5f * (ex - 1.f);
    }
  }
  se = wave_sum_dd(se); sv = wave_sum_dd(sv); sk = wave_sum_dd(sk);
  if (lane == 0) {
    partial[(size_t)b * 3 + 0] = se;
    partial[(size_t)b * 3 + 1] = sv;
    partial[(size_t)b * 3 + 2] = sk;
  }
}

__global__ __launch_bounds__(256) void svae_elbo_final_kernel(const double* __restrict__ partial, int B, float beta,
                                                              float* __restrict__ out4) {
  __shared__ double sh[3][4];
  double s[3] = {0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < B; b += 256)
    for (int k = 0; k < 3; ++k) s[k] += partial[(size_t)b * 3 + k];
  for (int k = 0; k < 3; ++k) s[k] = wave_sum_dd(s[k]);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 3; ++k) sh[k][threadIdx.x >> 6] = s[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[3];
    for (int k = 0; k < 3; ++k) t[k] = sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3];
    out4[0] = (float)t[0];
    out4[1] = (float)t[1];
    out4[2] = (float)t[2];
    out4[3] = (float)(t[0] + t[1] + (double)beta * t[2]);
  }
}

__global__ void svae_predict_kernel(long long rows, int C, int N, const float* __restrict__ ts,
                                    const float* __restrict__ es, const float* __restrict__ ut,
                                    const float* __restrict__ ue, float* __restrict__ onehot,
                                    float* __restrict__ flags) {
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
    const float* s = ts + r * C;
    int pick = 0;
    float m = s[0];
    for (int c = 1; c < C; ++c) if (s[c] > m) { m = s[c]; pick = c; }
    if (ut) {   // inverse CDF of softmax(s) at the injected uniform
      float sum = 0.f;
      for (int c = 0; c < C; ++c) sum += expf(s[c] - m);
      const float u = ut[r] * sum;
      float acc = 0.f;
      pick = C - 1;
      for (int c = 0; c < C; ++c) {
        acc += expf(s[c] - m);
        if (u < acc) { pick = c; break; }
      }
    }
    for (int c = 0; c < C; ++c) onehot[r * C + c] = c == pick ? 1.f : 0.f;
    for (int j = 0; j < N; ++j) {
      const float p = 1.f / (1.f + expf(-es[r * N + j]));
      flags[r * N + j] = (ue ? ue[r * N + j] < p : p > 0.5f) ? 1.f : 0.f;
    }
  }
}

}  // namespace

extern "C" int dvae_svae_elbo(const dvae_svae_dims* dp, const float* type_scores, const float* edge_scores,
                              const float* x, const float* mu, const float* logvar, float beta, float* out4,
                              void* partial, float* d_type, float* d_edge, float* d_mu, float* d_logvar,
                              void* stream) {
  DVAE_TRY(check_dims(dp));
  DVAE_REQUIRE(type_scores && edge_scores && x && mu && logvar && out4 && partial, "svae_elbo: null pointer");
  DVAE_REQUIRE((d_mu == nullptr) == (d_logvar == nullptr), "svae_elbo: d_mu and d_logvar go together");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(svae_elbo_kernel, dim3(dp->B), dim3(64), 0, st, *dp, type_scores, edge_scores, x, mu, logvar,
                     beta, (double*)partial, d_type, d_edge, d_mu, d_logvar);
  DVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(svae_elbo_final_kernel, dim3(1), dim3(256), 0, st, (const double*)partial, dp->B, beta, out4);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}

extern "C" int dvae_svae_predict(const dvae_svae_dims* dp, long long rows, const float* type_scores,
                                 const float* edge_scores, const float* u_type, const float* u_edge,
                                 float* onehot, float* flags, void* stream) {
  DVAE_TRY(check_dims(dp));
  DVAE_REQUIRE(type_scores && edge_scores && onehot && flags, "svae_predict: null pointer");
  DVAE_REQUIRE((u_type == nullptr) == (u_edge == nullptr), "svae_predict: u_type and u_edge go together");
  if (rows <= 0) return DVAE_OK;
  hipLaunchKernelGGL(svae_predict_kernel, grid_of(rows), dim3(256), 0, (hipStream_t)stream, rows, dp->C, dp->N,
                     type_scores, edge_scores, u_type, u_edge, onehot, flags);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}
