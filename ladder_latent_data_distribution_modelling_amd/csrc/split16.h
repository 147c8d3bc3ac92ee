// Operand splitting helpers shared by the split-precision contraction kernels (convsplit.hip, igemm.hip): precision formats,
// fp32 -> 16-bit plane splitting, the power-of-two tensor scale of the fp16 format and the 16-bit MFMA wrapper.  See the header
// comment of convsplit.hip for the numerics.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// Precision format of a split contraction: number of planes and the 16-bit element type.
template <int PREC> struct Fmt;
template <> struct Fmt<LADDER_PREC_BF16X6> { static constexpr int NS = 3; static constexpr bool F16 = false; };
template <> struct Fmt<LADDER_PREC_BF16X3> { static constexpr int NS = 2; static constexpr bool F16 = false; };
template <> struct Fmt<LADDER_PREC_F16X3>  { static constexpr int NS = 2; static constexpr bool F16 = true; };
inline bool prec_ok(int prec) { return prec == LADDER_PREC_BF16X6 || prec == LADDER_PREC_BF16X3 || prec == LADDER_PREC_F16X3; }
inline int prec_planes(int prec) { return prec == LADDER_PREC_BF16X6 ? 3 : 2; }

// packed pair of 16-bit floats, round-to-nearest-even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32): low half = a, high half = b;
// `back` = the pair converted back to fp32 (exact)
template <bool F16>
__device__ __forceinline__ uint32_t pk16(float a, float b, float& back_a, float& back_b) {
  const f32x2 v = {a, b};
  if (F16) {
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 f = __builtin_convertvector(h, f32x2);
    back_a = f.x;
    back_b = f.y;
    return __builtin_bit_cast(uint32_t, h);
  } else {
    const uint32_t p = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    back_a = __builtin_bit_cast(float, p << 16);
    back_b = __builtin_bit_cast(float, p & 0xffff0000u);
    return p;
  }
}

// Splits 4 floats into NS planes of 4 packed 16-bit values each (plane p -> out[p] = {elements 0,1 | elements 2,3}).
template <int NS, bool F16>
__device__ __forceinline__ void split4(const float4 v, uint2 (&out)[NS]) {
  float a = v.x, b = v.y, c = v.z, d = v.w;
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    float fa, fb, fc, fd;
    const uint32_t q0 = pk16<F16>(a, b, fa, fb), q1 = pk16<F16>(c, d, fc, fd);
    out[p] = make_uint2(q0, q1);
    if (p + 1 < NS) {
      a -= fa;
      b -= fb;
      c -= fc;
      d -= fd;
    }
  }
}

// Power-of-two scale c with |x| c < 2^14 for every |x| <= amax (fp16 tops out at 65504; the planes of the largest elements then keep
// 2 binades of headroom); 1 for an all-zero or non-finite tensor.
__device__ __forceinline__ float scale_from_absmax(float amax) {
  const uint32_t bits = __builtin_bit_cast(uint32_t, amax);
  const int e = (int)((bits >> 23) & 0xff);               // biased exponent: amax < 2^(e - 126)
  if (e == 0 || e == 255) return 1.f;
  int se = 127 + 14 - (e - 126);                          // biased exponent of 2^(14 - (e - 126))
  se = se < 1 ? 1 : (se > 254 ? 254 : se);
  return __builtin_bit_cast(float, (uint32_t)se << 23);
}

// a wave-uniform float back into a scalar register (float arithmetic runs on the vector unit even on uniform operands)
__device__ __forceinline__ float uniform_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

// Scale of sample n (see the record layouts in common.h): its own power-of-two scale when the record carries per-sample bounds, capped at
// 2^AMAX_PS_CAP above the tensor-wide scale ct = scale_from_absmax(tmax) (an all-zero sample takes ct), else ct.
__device__ __forceinline__ float scale_for_sample(const float* rec, int n, bool per_sample, float tmax) {
  const float ct = scale_from_absmax(tmax);
  if (!per_sample) return ct;
  const float bn = rec[amax_ps_index(n)];
  return bn > 0.f ? fminf(scale_from_absmax(bn), ct * (float)(1 << AMAX_PS_CAP)) : ct;
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const uint4 a, const uint4 b, const f32x16 c) {
  if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Bytes of the split filter image [tap][Cin/16][ceil(Cout/128)][plane][2][128][8 x 16 bit] (convsplit.hip: filter_pack_kernel); the
// filter's absolute-maximum record (f16x3 scale; LADDER_ABSMAX_FLOATS floats) is stored behind it.
inline size_t pack_payload_bytes(int ntaps, int Cin, int Cout, int prec) {
  return (size_t)ntaps * (Cin / 16) * ((Cout + 127) / 128) * prec_planes(prec) * 4096;
}
__device__ __forceinline__ size_t pack_payload_bytes_dev(int ntaps, int Cin, int Cout, int planes) {
  return (size_t)ntaps * (Cin / 16) * ((Cout + 127) / 128) * planes * 4096;
}

}  // namespace
