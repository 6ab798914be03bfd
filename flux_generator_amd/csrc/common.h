// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libfluxhip.
// Wavefront = 64 lanes everywhere in this tree; nothing here is portable to
// 32-wide hardware and nothing tries to be.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits in memory

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define FLUXHIP_OK 0
#define FLUXHIP_EINVAL (-1)
#define FLUXHIP_ELAUNCH (-2)

#define DEVINL __device__ __forceinline__

DEVINL float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
DEVINL float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
DEVINL float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// round-to-nearest-even f32 -> bf16 (v_cvt_pk_bf16_f32 on gfx950)
DEVINL uint32_t pack_bf16x2(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<uint32_t*>(&r);
}
DEVINL bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}
// value of f after a round trip through bf16 storage (MLX op-boundary rounding)
DEVINL float rbf(float f) { return bf2f(f2bf(f)); }

// ---- 16-bit storage element, selected at compile time: bfloat16 (H = false) or IEEE float16 (H = true).
// The stable_diffusion/ path runs in float16 when the caller says float16=True (the reference's flux_app.py setting): same
// kernels, same fp32 accumulation / norms / softmax, the element conversions and the MFMA operand type are the only difference.
typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;   // MFMA A/B fragment of v_mfma_f32_16x16x32_f16 (4 VGPRs)
DEVINL float h2f(uint16_t h) { return (float)__builtin_bit_cast(half_t, h); }
DEVINL float h_lo(uint32_t w) { return (float)__builtin_bit_cast(half_t, (uint16_t)(w & 0xffffu)); }
DEVINL float h_hi(uint32_t w) { return (float)__builtin_bit_cast(half_t, (uint16_t)(w >> 16)); }
// round-to-nearest-even f32 -> f16 (values beyond 65504 become inf, as in the reference's float16 arithmetic)
DEVINL uint32_t pack_f16x2(float lo, float hi) {
  f32x2 v = {lo, hi};
  f16x2_t r = __builtin_convertvector(v, f16x2_t);
  return __builtin_bit_cast(uint32_t, r);
}
DEVINL uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (half_t)f); }
template <bool H> DEVINL float e2f(uint16_t h) { if constexpr (H) return h2f(h); else return bf2f(h); }
template <bool H> DEVINL float e_lo(uint32_t w) { if constexpr (H) return h_lo(w); else return bf_lo(w); }
template <bool H> DEVINL float e_hi(uint32_t w) { if constexpr (H) return h_hi(w); else return bf_hi(w); }
template <bool H> DEVINL uint32_t e_pack(float lo, float hi) { if constexpr (H) return pack_f16x2(lo, hi); else return pack_bf16x2(lo, hi); }
template <bool H> DEVINL uint16_t f2e(float f) { if constexpr (H) return f2h(f); else return f2bf(f); }
// value of f after a round trip through the 16-bit storage type (op-boundary rounding of the reference's arrays)
template <bool H> DEVINL float e_rnd(float f) { return e2f<H>(f2e<H>(f)); }

// Activations of the 16-bit paths, written for the VALU count: the result is rounded to bf16 / float16 by the caller, so one
// v_exp_f32 + one v_rcp_f32 (1 ulp each) replace the IEEE division sequence (div_scale / rcp / 4 fma / div_fmas / div_fixup:
// 23 VALU instructions per GELU, 7 now).  A 256 x 256 GELU epilogue is 128 elements per lane on two waves per SIMD, i.e.
// the division form was ~12 us of VALU issue per tile round.
DEVINL float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// GELU, tanh approximation: 0.5x(1+tanh(u)) = x * sigmoid(2u), u = sqrt(2/pi)(x+0.044715x^3); exp(-2u) = exp2(x (c0 + c1 x^2)).
// The sigmoid form has no 1 - 2/(e+1) cancellation for negative x (the tanh form is off by up to 20 % relative near
// x = -5, 4 % of bf16 results differ from a float64 evaluation; this form: 2e-6 relative); +-inf saturate to x and -0.
DEVINL float gelu_tanh_f(float x) {
  constexpr float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, c1, c0));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// exact (erf) GELU, used by the SD UNet GEGLU and the OpenCLIP towers: x * Phi(x), Phi(x) = 1 - erfc(|x| / sqrt 2) / 2 for x >= 0 and
// erfc(|x| / sqrt 2) / 2 below (no 1 + erf cancellation on the negative side).  erfc(z) = t (a1 + t (a2 + ... a5 t)) exp(-z^2),
// t = 1 / (1 + p z) (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute): 14 VALU instructions incl. one v_rcp_f32 and one
// v_exp_f32 where libm's erff is ~40 with branches - the GEGLU epilogue of the SDXL UNet is 64 gate elements per lane per tile.
// The bound is ABSOLUTE on erfc: the result x * erfc / 2 is right to |x| * 0.75e-7, i.e. below half a float16 / bfloat16 ulp
// wherever |gelu(x)| >= ~3e-4 and a few ulps of the (tiny) results in the negative tail beyond x ~ -3.7 - relative 2e-3 at x = -4
// where gelu = -1.3e-4 (tests/test_sd_f16_gpu.py::test_gelu_erf_epilogue_negative_tail_vs_float64 sweeps [-6, 0] against
// float64).  The reference's float16 evaluation x (1 + erf(x / sqrt 2)) / 2 has no significant bit left there (1 + erf rounds
// to 0 or 4.9e-4 below x = -3.3), so the tail is closer to the exact function than to anything the reference computes.
DEVINL float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.7071067811865476f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float q = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  q = __builtin_fmaf(t, q, 1.421413741f);
  q = __builtin_fmaf(t, q, -0.284496736f);
  q = __builtin_fmaf(t, q, 0.254829592f);
  const float h = 0.5f * (t * q) * __builtin_amdgcn_exp2f(-1.4426950408889634f * (z * z));     // erfc(z) / 2
  return x * (x >= 0.f ? 1.0f - h : h);
}

DEVINL float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over each aligned group of 16 lanes, result in all 16 (DPP: quad swaps, half-row mirror, row mirror; no LDS crossbar)
DEVINL float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}
DEVINL float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Exchange between the two 32-lane halves of a wave without the LDS crossbar (v_permlane32_swap_b32, gfx950):
// with both operands = v the result pair is (v of lane l&31, v of lane (l&31)+32) in EVERY lane, so the
// max / sum over the lane pair (l, l^32) needs no select.  (__shfl_xor(v, 32) is a ds_bpermute: LDS latency plus an
// lgkmcnt wait that also drains every outstanding ds_read of the wave.)
DEVINL float pair32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
DEVINL float pair32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// 16-byte async copy global -> LDS. LDS destination = wave-uniform base + lane*16.
DEVINL void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

DEVINL void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E)
#include <type_traits>
#include <utility>
template <int B, int... Is, class F>
DEVINL void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, B + Is>{}), ...);
}
template <int B, int E, class F>
DEVINL void static_for(F&& f) {
  static_for_impl<B>(static_cast<F&&>(f), std::make_integer_sequence<int, E - B>{});
}
