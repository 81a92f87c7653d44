// Device side of the numerics contract (DESIGN.md "Numerics contract").
//
// Everything here is written so that gfx950 produces the same bits as an IEEE-754 host evaluating the
// same expression tree: no FMA contraction (-ffp-contract=off), correctly rounded divide/sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt), RTNE fp16 conversions (v_cvt_f16_f32), native fp16
// add/mul, minNum/maxNum (v_min_f32 / v_max_f32) and small polynomial sin/cos/acos/atan forms instead of
// library calls whose rounding differs between vendors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

// ---- shader constants, shaders/framebuffer.comp:27-42
#define PT_EPSILON 0.0001f
#define PT_FLOAT_COMPARE 0.01f
#define PT_PI 3.14159265f
#define PT_TWO_PI 6.28318530f
#define PT_TMIN 0.001f
#define PT_TMAX 100.0f
#define PT_NORMAL_OFFSET 0.005f

typedef _Float16 h16;

// fp32 -> fp16 of a value that was first rounded to fp32.  The empty asm pins the fp32 value in a VGPR:
// without it the gfx950 backend folds "fptrunc(fmul a, b)" into v_fma_mixlo_f16, which rounds the exact
// product ONCE to fp16 -- a different result from the contract's two roundings in ~1 case per 10^4.
PT_DEV h16 to_h16(float x) { asm("" : "+v"(x)); return (h16) x; }
PT_DEV float q16(float x) { return (float) to_h16(x); }               // float(float16_t(x))
PT_DEV uint32_t h_bits(h16 h) { return (uint32_t) __builtin_bit_cast(uint16_t, h); }
PT_DEV h16 h_from_bits(uint32_t b) { return __builtin_bit_cast(h16, (uint16_t) b); }
PT_DEV float fmin_(float a, float b) { return __builtin_fminf(a, b); }
PT_DEV float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }

struct v3 { float x, y, z; };
PT_DEV v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
PT_DEV v3 operator*(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
PT_DEV v3 operator/(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
PT_DEV v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
PT_DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV v3 cross(v3 a, v3 b) { return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
PT_DEV v3 normalize(v3 a) { float inv = 1.0f / __builtin_sqrtf(dot(a, a)); return a * inv; }
PT_DEV v3 q16(v3 a) { return V(q16(a.x), q16(a.y), q16(a.z)); }
// 1.0f / x for an x that came out of q16(): v_rcp_f32 + one Newton step is bit-equal to the IEEE division for every normal
// |x| < 2^126 (tools/verify_rcp.hip: all of them checked), which covers every finite non-zero fp16 value (2^-24 ... 65504);
// v_div_fixup_f32 supplies the IEEE answers for the rest of fp16 (+-0 -> +-inf, +-inf -> +-0, NaN).  5 instructions for the
// 12 of the general division; test_gpu_math.py::test_rcp_of_fp16_values checks all 65,536 inputs against the division.
PT_DEV float rcp_of_q16(float x)
{
	const float r = __builtin_amdgcn_rcpf(x);
	return __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r), x, 1.0f);
}

struct m3 { v3 r0, r1, r2; }; // rows
PT_DEV v3 mul(const m3 &m, v3 v) { return V(dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)); }
PT_DEV m3 inverse(const m3 &m)
{
	float a = m.r0.x, b = m.r0.y, c = m.r0.z;
	float d = m.r1.x, e = m.r1.y, f = m.r1.z;
	float g = m.r2.x, h = m.r2.y, i = m.r2.z;
	float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
	float det = a * c00 + b * c01 + c * c02;
	float id = 1.0f / det;
	m3 r;
	r.r0 = V(c00 * id, (c * h - b * i) * id, (b * f - c * e) * id);
	r.r1 = V(c01 * id, (a * i - c * g) * id, (c * d - a * f) * id);
	r.r2 = V(c02 * id, (b * g - a * h) * id, (a * e - b * d) * id);
	return r;
}

// ---- transcendental forms of the contract
PT_DEV void ptm_sincos(float x, float *s, float *c)
{
	float q = __builtin_rintf(x * 0.63661977236758134f);
	int n = (int) q;
	float r = ((x - q * 1.5703125f) - q * 4.837512969970703125e-4f) - q * 7.54978995489188216e-8f;
	float r2 = r * r;
	float sp = ((-1.9515295891e-4f * r2 + 8.3321608736e-3f) * r2 - 1.6666654611e-1f) * r2 * r + r;
	float cp = ((2.443315711809948e-5f * r2 - 1.388731625493765e-3f) * r2 + 4.166664568298827e-2f) * r2 * r2
	           - 0.5f * r2 + 1.0f;
	bool swap = n & 1;
	float ss = swap ? cp : sp, cc = swap ? sp : cp;
	*s = (n & 2) ? -ss : ss;
	*c = ((n + 1) & 2) ? -cc : cc;
}
PT_DEV float ptm_sin(float x) { float s, c; ptm_sincos(x, &s, &c); return s; }
PT_DEV float ptm_cos(float x) { float s, c; ptm_sincos(x, &s, &c); return c; }
PT_DEV float ptm_tan(float x) { float s, c; ptm_sincos(x, &s, &c); return s / c; }
PT_DEV float ptm_asin_core(float x)
{
	float z = x * x;
	return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
	        + 1.6666752422e-1f) * z * x + x;
}
PT_DEV float ptm_acos(float x)
{
	x = fmin_(1.0f, fmax_(-1.0f, x));
	if (x < -0.5f) return 3.14159274101257324f - 2.0f * ptm_asin_core(__builtin_sqrtf(0.5f * (1.0f + x)));
	if (x > 0.5f) return 2.0f * ptm_asin_core(__builtin_sqrtf(0.5f * (1.0f - x)));
	return 1.57079637050628662f - ptm_asin_core(x);
}
PT_DEV float ptm_atan(float x)
{
	float ax = __builtin_fabsf(x), y;
	if (ax > 2.414213562373095f) { y = 1.57079637050628662f; ax = -(1.0f / ax); }
	else if (ax > 0.4142135623730950f) { y = 0.785398185253143311f; ax = (ax - 1.0f) / (ax + 1.0f); }
	else y = 0.0f;
	float z = ax * ax;
	y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * ax
	         + ax);
	return x < 0.0f ? -y : y;
}
PT_DEV float ptm_atan2(float y, float x)
{
	if (x == 0.0f) return y > 0.0f ? 1.57079637050628662f : (y < 0.0f ? -1.57079637050628662f : 0.0f);
	if (y == 0.0f) return x < 0.0f ? 3.14159274101257324f : 0.0f;
	float w = x < 0.0f ? (y < 0.0f ? -3.14159274101257324f : 3.14159274101257324f) : 0.0f;
	return w + ptm_atan(y / x);
}

// ---- RNG, shaders/framebuffer.comp:111-139
PT_DEV uint32_t seed3(uint32_t x, uint32_t y, uint32_t z) { return 19u * x + 47u * y + 101u * z + 131u; }
PT_DEV uint32_t pcg(uint32_t &st)
{
	uint32_t state = st;
	st = st * 747796405u + 2891336453u;
	uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
	return (word >> 22u) ^ word;
}
// float(hash) / float(0xfffffff0u): the divisor rounds to 2^32 in fp32, so this is an exact scaling
PT_DEV float rnd(uint32_t &st)
{
	float f = (float) pcg(st) * 2.3283064365386963e-10f;
	return fmax_(PT_EPSILON, fmin_(1.0f - PT_EPSILON, f));
}
