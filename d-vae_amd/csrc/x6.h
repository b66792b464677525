// The x6 arithmetic core shared by gemm.hip and level.hip: fp32 operands split exactly into three bf16 pieces while
// they are staged into LDS, fragments for v_mfma_f32_32x32x16_bf16.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

// ---------------------------------------------------------------------------
// fp32 products on the bf16 matrix pipe ("x6" core).  CDNA4 retires v_mfma_f32_32x32x16_bf16 in 32 cycles and
// v_mfma_f32_32x32x2_f32 in 64: sixteen times the multiply-adds per clock.  An fp32 value splits EXACTLY into
// three bf16 pieces (x = x1 + x2 + x3: truncation keeps 8 significand bits per piece, 24 in all), a bf16 x bf16
// product is exact in the fp32 accumulator, and of the nine piece products the six with i + j <= 4
//     a1b1, a1b2, a2b1, a1b3, a2b2, a3b1
// carry everything above 2^-24 |a b|, which is fp32's own rounding step (the dropped a2b3 + a3b2 + a3b3 are
// ~2^-25 |a b| on average).  So six bf16 MFMAs give the fp32 product at 16 / 6 = 2.7 x the fp32 MFMA rate, with
// fp32 accumulation as before.
// WORST CASE (not the average): truncation gives |x2| < 2^-7 |x| and |x3| < 2^-15 |x|, so the dropped products of one
// a*b are below (2^-22 + 2^-22 + 2^-30) |a b| < 2^-21 |a b|, and they all carry the sign of a*b (truncation rounds
// every piece towards zero): a dot product is off by less than 2^-21 sum_k |a_k||b_k| towards zero, on top of the
// fp32 accumulation error, whatever K is and whatever the signs are -- the bias does not average out on same-sign
// operands, but it cannot exceed this bound either (tests/test_gpu_kernels.py: K = 57344 same-sign operands,
// all-ones significands, denormals).  The maximum is reached by significands that are all ones (2^-23 |a b|).
// Non-finite operands: split3(+-inf) = (inf, inf - inf = NaN, NaN), so an inf operand gives NaN where the fp32 MFMA
// gives inf; NaN stays NaN.  Non-finite in, non-finite out -- but inf is not preserved as inf.  Denormal pieces may be
// flushed by the matrix pipe; they are below 2^-126 and only matter for dot products that are denormal throughout.  Operands are split while they are staged into LDS (5.5 VALU operations per
// element, once per k-tile) and kept as three planes of bf16 in the layout the 32x32x16 fragment read wants:
//     plane[p][k-octet o][row r] = 8 bf16 (16 bytes) = pieces p of k = 8o .. 8o+7 of row r
// so a lane's fragment is one ds_read_b128 and 16 lanes of a read group touch 16 consecutive 16-byte slots.
// A k-major operand ([k][m] in memory) is transposed in registers on the way: a thread loads NV consecutive
// k rows of 4 consecutive m and writes, per m, NV consecutive k.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct Split3 { unsigned p1, p2, p3; };   // bf16 pieces in the HIGH halves
__device__ __forceinline__ Split3 split3(float x) {
  Split3 s;
  s.p1 = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(s.p1);
  s.p2 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(s.p2);
  s.p3 = __float_as_uint(r2);
  return s;
}
// (high half of lo) | (high half of hi) << 16
__device__ __forceinline__ unsigned pack_hi(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

template <int KM, int ROWS>
struct XTile {
  static constexpr int BK = 32;
  static constexpr int NV = ROWS / 32;              // float4 per thread
  static constexpr int KGN = 32 / (KM ? NV : 1);              // k-major: groups of NV consecutive k
  static constexpr int RQ = ROWS / 4;
  static constexpr int ROWSP = ROWS + (KM ? 1 : 2); // 16-byte slots per k-octet (the pad spreads the staging stores over the banks)
  static constexpr int PLANE = 4 * ROWSP * 16;      // bytes
  static constexpr int LDS_FLOATS = 3 * PLANE / 4;
  static_assert(NV >= 1 && NV <= 8 && (KM == 0 || NV == 1 || NV == 2 || NV == 4), "x6 tiles: m-major 32..256 rows, k-major 32 / 64 / 128");
  float4 v[NV];
  size_t ro[NV];
  int kq[NV];

  __device__ __forceinline__ void rows_init(int ld, int r0, int rows, const int* __restrict__ map, int tid) {
    if (KM == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int r = min(r0 + tid / 8 + i * 32, rows - 1);
        ro[i] = (size_t)(map ? map[r] : r) * ld;
      }
    }
  }
  __device__ __forceinline__ void kq_fetch(const int* __restrict__ kmap, int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) kq[i] = kmap[min(k0 + (tid % KGN) * NV + i, kend - 1)];
  }
  __device__ __forceinline__ void load_gathered(const float* __restrict__ P, int ld, int r0, int rows, int tid) {
    const int r = min(r0 + (tid / KGN) * 4, rows - 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(P + (size_t)kq[i] * ld + r);
  }
  __device__ __forceinline__ void load_gathered_checked(const float* __restrict__ P, int ld, int r0, int rows, int k0,
                                                        int kend, const int* __restrict__ kmap, int tid) {
    const int r = r0 + (tid / KGN) * 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int k = k0 + (tid % KGN) * NV + i;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < kend && r < rows) val = *reinterpret_cast<const float4*>(P + (size_t)kmap[k] * ld + r);
      v[i] = val;
    }
  }
  template <bool CHECK>
  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int r0, int rows, int k0, int kend, int tid) {
    if (KM == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int k = k0 + (tid % 8) * 4;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!CHECK || (r0 + tid / 8 + i * 32 < rows && k < kend)) val = *reinterpret_cast<const float4*>(P + ro[i] + k);
        v[i] = val;
      }
    } else {
      const int rr = r0 + (tid / KGN) * 4;
      const int r = CHECK ? rr : min(rr, rows - 4);
      const float* base = P + (size_t)(k0 + (tid % KGN) * NV) * ld + r;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!CHECK || (k0 + (tid % KGN) * NV + i < kend && rr < rows)) val = *reinterpret_cast<const float4*>(base + (size_t)i * ld);
        v[i] = val;
      }
    }
  }

  // split into bf16 pieces and stage: lds -> three planes
  __device__ __forceinline__ void store(float* lds, int tid) const {
    char* base = reinterpret_cast<char*>(lds);
    if (KM == 0) {
      const int kq4 = tid % 8;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int row = tid / 8 + i * 32;
        const Split3 a = split3(v[i].x), b = split3(v[i].y), c = split3(v[i].z), d = split3(v[i].w);
        char* q = base + ((kq4 >> 1) * ROWSP + row) * 16 + (kq4 & 1) * 8;
        *reinterpret_cast<uint2*>(q) = make_uint2(pack_hi(a.p1, b.p1), pack_hi(c.p1, d.p1));
        *reinterpret_cast<uint2*>(q + PLANE) = make_uint2(pack_hi(a.p2, b.p2), pack_hi(c.p2, d.p2));
        *reinterpret_cast<uint2*>(q + 2 * PLANE) = make_uint2(pack_hi(a.p3, b.p3), pack_hi(c.p3, d.p3));
      }
    } else {
      const int kg = tid % KGN, mq = tid / KGN;
      const int kk = kg * NV;   // first of this thread's NV consecutive k
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        Split3 s[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) s[i] = split3(c == 0 ? v[i].x : c == 1 ? v[i].y : c == 2 ? v[i].z : v[i].w);
        char* q = base + ((kk >> 3) * ROWSP + mq * 4 + c) * 16 + (kk & 7) * 2;
        if (NV == 4) {
          *reinterpret_cast<uint2*>(q) = make_uint2(pack_hi(s[0].p1, s[1].p1), pack_hi(s[2].p1, s[3].p1));
          *reinterpret_cast<uint2*>(q + PLANE) = make_uint2(pack_hi(s[0].p2, s[1].p2), pack_hi(s[2].p2, s[3].p2));
          *reinterpret_cast<uint2*>(q + 2 * PLANE) = make_uint2(pack_hi(s[0].p3, s[1].p3), pack_hi(s[2].p3, s[3].p3));
        } else if (NV == 2) {
          *reinterpret_cast<unsigned*>(q) = pack_hi(s[0].p1, s[NV - 1].p1);
          *reinterpret_cast<unsigned*>(q + PLANE) = pack_hi(s[0].p2, s[NV - 1].p2);
          *reinterpret_cast<unsigned*>(q + 2 * PLANE) = pack_hi(s[0].p3, s[NV - 1].p3);
        } else {
          *reinterpret_cast<unsigned short*>(q) = (unsigned short)(s[0].p1 >> 16);
          *reinterpret_cast<unsigned short*>(q + PLANE) = (unsigned short)(s[0].p2 >> 16);
          *reinterpret_cast<unsigned short*>(q + 2 * PLANE) = (unsigned short)(s[0].p3 >> 16);
        }
      }
    }
  }

  // piece p of the 32-row sub-tile at rbase, k-step ks (16 wide): lane holds k = 16 ks + 8 (lane>>5) + 0..7 of row lane&31
  static __device__ __forceinline__ bf16x8 frag(const float* lds, int p, int rbase, int ks, int lane) {
    const char* q = reinterpret_cast<const char*>(lds) + p * PLANE + ((ks * 2 + (lane >> 5)) * ROWSP + rbase + (lane & 31)) * 16;
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q));
  }
};
