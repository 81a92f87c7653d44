// TEST INFRASTRUCTURE ONLY -- see oracle.h.
//
// CPU restatement of /root/reference/shaders/framebuffer.comp (the render hot
// path).  Every function cites the shader lines it follows ("fc:NNN").  The
// arithmetic contract (DESIGN.md "Numerics contract") is:
//   * fp32 +,-,*,/,sqrt are IEEE-754 round-to-nearest-even, never contracted to
//     FMA (built with -ffp-contract=off), left-to-right as GLSL writes them;
//   * float16_t values are true binary16: every fp16 result is rounded RTNE
//     (computed here as fp32 op + one rounding, which is exact for +,* because
//     24 >= 2*11+2);
//   * min/max are IEEE-754-2008 minNum/maxNum with -0 < +0 (GLSL leaves NaN
//     operands undefined; this is what gfx950's v_min_f32/v_max_f32 do);
//   * sin/cos/tan/acos/atan have unspecified precision in GLSL.  The contract
//     fixes them to the small deterministic polynomial forms ptm_* below so
//     that a GPU implementation can match bit for bit; ORACLE_FLAG_LIBM swaps
//     in glibc's sinf/cosf/tanf/acosf/atan2f to show the choice is immaterial
//     beyond rounding noise (tests/test_oracle.py: test_transcendental_contract_is_close_to_libm,
//     test_libm_variant_is_statistically_the_same_image);
//   * pow(x,2) and pow(x,5) are repeated multiplication.
#include "oracle.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- bit helpers
inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// binary16 <-> binary32, round-to-nearest-even
uint16_t f2h(float f)
{
	uint32_t x = fbits(f);
	uint32_t sign = (x >> 16) & 0x8000u;
	x &= 0x7fffffffu;
	if (x >= 0x7f800000u) // inf / NaN (NaN keeps its top payload bits, forced quiet)
		return (uint16_t) (sign | 0x7c00u | (x > 0x7f800000u ? (0x200u | ((x >> 13) & 0x3ffu)) : 0u));
	if (x < 0x38800000u) { // below 2^-14: half subnormal = RTNE(|f| * 2^24)
		uint32_t m = (uint32_t) nearbyintf(bitsf(x) * 16777216.0f);
		return (uint16_t) (sign | m);
	}
	uint32_t odd = (x >> 13) & 1u;
	x += 0xfffu + odd;
	x -= 0x38000000u;
	uint32_t h = x >> 13;
	if (h >= 0x7c00u) h = 0x7c00u;
	return (uint16_t) (sign | h);
}

float h2f(uint16_t h)
{
	uint32_t s = ((uint32_t) h & 0x8000u) << 16;
	uint32_t e = (h >> 10) & 31u, m = h & 0x3ffu;
	if (e == 0) {
		float v = (float) m * 5.9604644775390625e-8f; // 2^-24, exact
		return bitsf(fbits(v) | s);
	}
	if (e == 31) return bitsf(s | 0x7f800000u | (m << 13));
	return bitsf(s | ((e + 112u) << 23) | (m << 13));
}

// a value held in a float16_t variable, widened: float(float16_t(x))
inline float q16(float x) { return h2f(f2h(x)); }

inline float fmin_(float a, float b)
{
	if (a != a) return b;
	if (b != b) return a;
	if (a == b) return (fbits(a) & 0x80000000u) ? a : b;
	return a < b ? a : b;
}
inline float fmax_(float a, float b)
{
	if (a != a) return b;
	if (b != b) return a;
	if (a == b) return (fbits(a) & 0x80000000u) ? b : a;
	return a > b ? a : b;
}

// ---------------------------------------------------------------- vec3
struct v3 { float x, y, z; };
inline v3 V(float x, float y, float z) { return v3{x, y, z}; }
inline v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
inline v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
inline v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
inline v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
inline v3 operator*(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
inline v3 operator/(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
inline v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline v3 cross(v3 a, v3 b) { return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
// contract: normalize(v) = v * (1 / sqrt(dot(v,v)))
inline v3 normalize(v3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
inline v3 q16(v3 a) { return V(q16(a.x), q16(a.y), q16(a.z)); }

struct m3 { v3 r0, r1, r2; }; // stored by ROWS: (M*v).i = dot(r_i, v), summed x,y,z
inline v3 mul(const m3 &m, v3 v) { return V(dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)); }

// contract: inverse(mat3) by cofactors / determinant (fc:908 needs a true inverse)
m3 inverse(const m3 &m)
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

// ---------------------------------------------------------------- constants (fc:27-42)
const float EPSILON = 0.0001f;
const float FLOAT_COMPARE = 0.01f;
const float PI = 3.14159265f;
const float TWO_PI = 6.28318530f;
const float TMIN = 0.001f;
const float TMAX = 100.0f;
const float ENVIRONMENT_MAP_LE = 1.0f;
const float NEE_SPECULAR_ROUGHNESS_CUTOFF = 0.0f;
const float NORMAL_OFFSET = 0.005f;

// ---------------------------------------------------------------- transcendental contract

// sin and cos together; Cody-Waite reduction by pi/2 in three exact-product pieces,
// degree-7 / degree-8 minimax kernels on [-pi/4, pi/4].
inline void ptm_sincos(float x, float *s, float *c)
{
	float q = nearbyintf(x * 0.63661977236758134f);
	int n = (int) q;
	float r = ((x - q * 1.5703125f) - q * 4.837512969970703125e-4f) - q * 7.54978995489188216e-8f;
	float r2 = r * r;
	float sp = ((-1.9515295891e-4f * r2 + 8.3321608736e-3f) * r2 - 1.6666654611e-1f) * r2 * r + r;
	float cp = ((2.443315711809948e-5f * r2 - 1.388731625493765e-3f) * r2 + 4.166664568298827e-2f) * r2 * r2
	           - 0.5f * r2 + 1.0f;
	switch (n & 3) {
	case 0: *s = sp; *c = cp; break;
	case 1: *s = cp; *c = -sp; break;
	case 2: *s = -sp; *c = -cp; break;
	default: *s = -cp; *c = sp; break;
	}
}
inline float ptm_asin_core(float x) // |x| <= 0.5
{
	float z = x * x;
	return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
	        + 1.6666752422e-1f) * z * x + x;
}
// argument clamped to [-1,1] (GLSL: undefined outside; normalised vectors overshoot by an ulp)
inline float ptm_acos(float x)
{
	x = fmin_(1.0f, fmax_(-1.0f, x));
	if (x < -0.5f) return 3.14159274101257324f - 2.0f * ptm_asin_core(sqrtf(0.5f * (1.0f + x)));
	if (x > 0.5f) return 2.0f * ptm_asin_core(sqrtf(0.5f * (1.0f - x)));
	return 1.57079637050628662f - ptm_asin_core(x);
}
inline float ptm_atan(float x)
{
	float ax = fabsf(x), y;
	if (ax > 2.414213562373095f) { y = 1.57079637050628662f; ax = -(1.0f / ax); }
	else if (ax > 0.4142135623730950f) { y = 0.785398185253143311f; ax = (ax - 1.0f) / (ax + 1.0f); }
	else y = 0.0f;
	float z = ax * ax;
	y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * ax
	         + ax);
	return x < 0.0f ? -y : y;
}
inline float ptm_atan2(float y, float x)
{
	if (x == 0.0f) return y > 0.0f ? 1.57079637050628662f : (y < 0.0f ? -1.57079637050628662f : 0.0f);
	if (y == 0.0f) return x < 0.0f ? 3.14159274101257324f : 0.0f;
	float w = x < 0.0f ? (y < 0.0f ? -3.14159274101257324f : 3.14159274101257324f) : 0.0f;
	return w + ptm_atan(y / x);
}

struct Fn {
	bool libm;
	void sincos(float x, float *s, float *c) const
	{
		if (libm) { *s = sinf(x); *c = cosf(x); } else ptm_sincos(x, s, c);
	}
	float sin_(float x) const { float s, c; sincos(x, &s, &c); return s; }
	float cos_(float x) const { float s, c; sincos(x, &s, &c); return c; }
	float tan_(float x) const
	{
		if (libm) return tanf(x);
		float s, c; ptm_sincos(x, &s, &c); return s / c;
	}
	float acos_(float x) const { return libm ? acosf(fmin_(1.0f, fmax_(-1.0f, x))) : ptm_acos(x); }
	float atan2_(float y, float x) const { return libm ? atan2f(y, x) : ptm_atan2(y, x); }
};

// ---------------------------------------------------------------- scene views
struct Tri { float d1[4], d2[4], d3[4]; uint32_t d4[4]; };          // fc:46-52
struct Sph { float d[4]; uint32_t mat[4]; };                        // fc:54-58
struct Mat { float d1[4], d2[4], d3[4]; };                          // fc:60-65
struct Node { float d1[4], d2[4]; };                                // fc:67-71
static_assert(sizeof(Tri) == 64 && sizeof(Sph) == 32 && sizeof(Mat) == 48 && sizeof(Node) == 32, "layout");

struct Scene {
	const Sph *spheres; uint32_t n_spheres;
	const Tri *tris; uint32_t n_tris;
	const uint32_t *light_tris; uint32_t n_light_tris;
	const Mat *mats; uint32_t n_mats;
	const Node *nodes; uint32_t n_nodes;
	const uint32_t *light_spheres; uint32_t n_light_spheres;
	const uint8_t *env; uint32_t env_w, env_h;
	const uint8_t *tex; uint32_t tex_layers;
	uint32_t tex_wrap_s, tex_wrap_t, tex_mag_filter;
};
Scene view(const oracle_scene *s)
{
	Scene v;
	v.spheres = (const Sph *) s->spheres; v.n_spheres = s->n_spheres;
	v.tris = (const Tri *) s->tris; v.n_tris = s->n_tris;
	v.light_tris = s->light_tris; v.n_light_tris = s->n_light_tris;
	v.mats = (const Mat *) s->materials; v.n_mats = s->n_materials;
	v.nodes = (const Node *) s->nodes; v.n_nodes = s->n_nodes;
	v.light_spheres = s->light_spheres; v.n_light_spheres = s->n_light_spheres;
	v.env = s->env_rgb8; v.env_w = s->env_w; v.env_h = s->env_h;
	v.tex = s->tex_rgb8; v.tex_layers = s->tex_layers;
	v.tex_wrap_s = s->tex_wrap_s; v.tex_wrap_t = s->tex_wrap_t; v.tex_mag_filter = s->tex_mag_filter;
	return v;
}

struct Counters { uint64_t prim = 0, shadow = 0, bounce = 0, steps = 0, tests = 0, max_stack = 0, paths = 0; };

// ---------------------------------------------------------------- RNG (fc:111-139)
inline uint32_t seed3(uint32_t x, uint32_t y, uint32_t z) { return 19u * x + 47u * y + 101u * z + 131u; }
inline uint32_t pcg(uint32_t &st)
{
	uint32_t state = st;
	st = st * 747796405u + 2891336453u;
	uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
	return (word >> 22u) ^ word;
}
inline float hash_to_float(uint32_t h) { return (float) h / (float) 0xfffffff0u; }
inline float rnd(uint32_t &st) { return fmax_(EPSILON, fmin_(1.0f - EPSILON, hash_to_float(pcg(st)))); }

// ---------------------------------------------------------------- hit record (fc:214-222)
struct Hit {
	uint16_t t;        // float16_t
	uint16_t n[3];     // f16vec3
	uint8_t mat;       // uint8_t
	uint16_t obj16;    // uint16_t object_index (wraps above 65535: SURVEY quirk 2)
	uint8_t type;      // 0 tri, 1 sphere
	uint16_t uv[2];    // f16vec2
	// not part of the reference record: full-width index + barycentrics for the bit-exact trace contract
	uint32_t obj32;
	float u, v;
};

// fc:203-210, all float16_t arithmetic; the z term mixes a float literal so it is fp32 then rounded once
void oct_decode(uint32_t packed, uint16_t out[3])
{
	uint16_t fx = (uint16_t) (packed & 0xffffu), fy = (uint16_t) (packed >> 16);
	float x = h2f(fx), y = h2f(fy);
	float z = q16((1.0f - fabsf(x)) - fabsf(y));
	float t = fmax_(-z, 0.0f); // exact in fp16
	x = q16(x + (x >= 0.0f ? -t : t));
	y = q16(y + (y >= 0.0f ? -t : t));
	// contract: normalize() of an f16vec3 is evaluated in fp32 and rounded once per component
	v3 n = normalize(V(x, y, z));
	out[0] = f2h(n.x); out[1] = f2h(n.y); out[2] = f2h(n.z);
}

// fc:271-316
bool intersect_triangle(const Scene &sc, v3 ro, v3 rd, uint32_t tri_index, Hit &data, float tmax)
{
	const Tri &tri = sc.tris[tri_index];
	v3 v0 = V(tri.d1[0], tri.d1[1], tri.d1[2]);
	v3 edge1 = V(tri.d2[0], tri.d2[1], tri.d2[2]) - v0;
	v3 edge2 = V(tri.d3[0], tri.d3[1], tri.d3[2]) - v0;
	v3 pvec = cross(rd, edge2);
	float dt = dot(edge1, pvec);
	float inv_determinant = 1.0f / dt;
	v3 tvec = ro - v0;
	float u = dot(tvec, pvec) * inv_determinant;
	if (fabsf(dt) < EPSILON || (u < 0.0f) || (u > 1.0f)) return false;
	v3 qvec = cross(tvec, edge1);
	float v = inv_determinant * dot(rd, qvec);
	float t = inv_determinant * dot(edge2, qvec);
	if ((v >= 0.0f) && (u + v <= 1.0f) && t > TMIN && t < tmax) {
		uint16_t n0[3], n1[3], n2[3];
		oct_decode(tri.d4[0], n0);
		oct_decode(tri.d4[1], n1);
		oct_decode(tri.d4[2], n2);
		data.t = f2h(t);
		data.mat = (uint8_t) tri.d4[3];
		data.obj16 = (uint16_t) tri_index;
		data.obj32 = tri_index;
		data.type = 0;
		data.u = u; data.v = v;
		float w = q16((1.0f - u) - v); // float16_t w
		float uh = q16(u), vh = q16(v);
		uint32_t p0 = fbits(tri.d1[3]), p1 = fbits(tri.d2[3]), p2 = fbits(tri.d3[3]);
		for (int k = 0; k < 2; k++) { // f16 mul, f16 add, left to right (fc:304-306)
			float a = h2f((uint16_t) (k ? p0 >> 16 : p0 & 0xffffu));
			float b = h2f((uint16_t) (k ? p1 >> 16 : p1 & 0xffffu));
			float c = h2f((uint16_t) (k ? p2 >> 16 : p2 & 0xffffu));
			data.uv[k] = f2h(q16(q16(w * a) + q16(uh * b)) + q16(vh * c));
		}
		v3 surface_normal = normalize(cross(edge1, edge2));
		// w * n0 is f16*f16vec3 -> f16; u * n1 and v * n2 are float*f16vec3 -> vec3 (fc:309)
		v3 wn0 = V(q16(w * h2f(n0[0])), q16(w * h2f(n0[1])), q16(w * h2f(n0[2])));
		v3 s = (wn0 + u * V(h2f(n1[0]), h2f(n1[1]), h2f(n1[2]))) + v * V(h2f(n2[0]), h2f(n2[1]), h2f(n2[2]));
		v3 nn = normalize(s);
		float sgn = dot(surface_normal, rd) < 0.0f ? 1.0f : -1.0f;
		data.n[0] = f2h(q16(nn.x) * sgn);
		data.n[1] = f2h(q16(nn.y) * sgn);
		data.n[2] = f2h(q16(nn.z) * sgn);
		return true;
	}
	return false;
}

// fc:320-327
inline void intersect_aabb(v3 ro, v3 inv_dir, const Node &n, float t, float *entry, float *exit_)
{
	v3 tmin = (V(n.d1[0], n.d1[1], n.d1[2]) - ro) * inv_dir;
	v3 tmax = (V(n.d2[0], n.d2[1], n.d2[2]) - ro) * inv_dir;
	v3 t0 = V(fmin_(tmin.x, tmax.x), fmin_(tmin.y, tmax.y), fmin_(tmin.z, tmax.z));
	v3 t1 = V(fmax_(tmin.x, tmax.x), fmax_(tmin.y, tmax.y), fmax_(tmin.z, tmax.z));
	*entry = fmax_(t0.x, fmax_(t0.y, fmax_(t0.z, TMIN)));
	*exit_ = fmin_(t1.x, fmin_(t1.y, fmin_(t1.z, t)));
}

const int STACK_CAP = 128; // reference: 16 slots, no overflow check (fc:339,397); builder max_depth 64

// fc:343-421
bool intersect_bvh_stack(const Scene &sc, v3 ro, v3 rd, Hit &data, float &tmax, Counters &cn)
{
	bool hit_anything = false;
	v3 inv_dir = V(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
	uint32_t stack[STACK_CAP];
	int stack_size = 0;
	uint32_t current_index = 1;
	if (sc.n_nodes < 3) return false; // reference reads out of bounds here
	while (true) {
		cn.steps++;
		const Node &node_left = sc.nodes[current_index];
		const Node &node_right = sc.nodes[current_index + 1];
		float le, lx, re, rx;
		intersect_aabb(ro, inv_dir, node_left, tmax, &le, &lx);
		intersect_aabb(ro, inv_dir, node_right, tmax, &re, &rx);
		bool hit_left = le <= lx;
		bool hit_right = re <= rx;
		bool is_left_leaf = node_left.d2[3] > 0;
		bool is_right_leaf = node_right.d2[3] > 0;
		uint32_t num_tris = (uint32_t) (hit_left ? node_left.d2[3] : 0.0f) + (uint32_t) (hit_right ? node_right.d2[3] : 0.0f);
		if (num_tris > 0) {
			uint32_t first_prim = (uint32_t) (hit_left && is_left_leaf ? node_left.d1[3] : node_right.d1[3]);
			for (uint32_t i = 0; i < num_tris; i++) {
				Hit local_data = data;
				cn.tests++;
				bool hit_tri = first_prim + i < sc.n_tris && intersect_triangle(sc, ro, rd, first_prim + i, local_data, tmax);
				if (hit_tri) {
					hit_anything = true;
					tmax = h2f(local_data.t);
					data = local_data;
				}
			}
			hit_left = hit_left && !is_left_leaf;
			hit_right = hit_right && !is_right_leaf;
		}
		if (hit_left) {
			if (hit_right) {
				uint32_t first = (uint32_t) node_left.d1[3];
				uint32_t second = (uint32_t) node_right.d1[3];
				if (le > re) { uint32_t tmp = first; first = second; second = tmp; }
				if (stack_size < STACK_CAP) stack[stack_size] = second;
				stack_size++;
				if ((uint64_t) stack_size > cn.max_stack) cn.max_stack = (uint64_t) stack_size;
				if (stack_size > STACK_CAP) { stack_size = STACK_CAP; } // cannot happen: depth <= 64
				current_index = first;
			} else {
				current_index = (uint32_t) node_left.d1[3];
			}
		} else if (hit_right) {
			current_index = (uint32_t) node_right.d1[3];
		} else {
			if (stack_size == 0) break;
			current_index = stack[--stack_size];
		}
	}
	return hit_anything;
}

// fc:226-268
float intersect_sphere(v3 ro, v3 rd, const Sph &s)
{
	v3 oc = ro - V(s.d[0], s.d[1], s.d[2]);
	float a = dot(rd, rd);
	float b = 2.0f * dot(oc, rd);
	float c = dot(oc, oc) - s.d[3] * s.d[3];
	float discriminant = b * b - 4.0f * a * c;
	if (discriminant >= 0) {
		if (discriminant <= EPSILON) {
			float t = -b / (2.0f * a);
			if (t >= TMIN && t <= TMAX) return t;
		} else {
			float sq = sqrtf(discriminant);
			float t1 = (-b + sq) / (2.0f * a);
			float t2 = (-b - sq) / (2.0f * a);
			if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
			if (t1 >= TMIN && t1 <= TMAX) return t1;
			if (t2 >= TMIN && t2 <= TMAX) return t2;
		}
	}
	return -1.0f;
}

// fc:423-456
bool intersect(const Scene &sc, v3 ro, v3 rd, Hit &result, Counters &cn)
{
	bool hit_anything = false;
	float tmax = TMAX;
	memset(&result, 0, sizeof result); // `out` members a miss leaves undefined: contract says zero
	hit_anything = intersect_bvh_stack(sc, q16(ro), q16(rd), result, tmax, cn); // fc:437: f16vec3(ro), f16vec3(rd)
	for (uint32_t i = 0; i < (sc.n_spheres & 0xffffu); i++) { // uint16_t counter
		const Sph &s = sc.spheres[i];
		uint16_t th = f2h(intersect_sphere(ro, rd, s));
		float current_t = h2f(th);
		if (current_t >= TMIN && current_t <= tmax) {
			tmax = current_t;
			hit_anything = true;
			result.t = th;
			v3 n = ((ro + rd * current_t) - V(s.d[0], s.d[1], s.d[2])) / s.d[3];
			result.n[0] = f2h(n.x); result.n[1] = f2h(n.y); result.n[2] = f2h(n.z);
			result.mat = (uint8_t) s.mat[0];
			result.obj16 = (uint16_t) i;
			result.obj32 = i;
			result.type = 1;
		}
	}
	return hit_anything;
}

// ---------------------------------------------------------------- sampling helpers
struct Ctx {
	const Scene &sc;
	Fn fn;
	Counters cn;
	bool dbg = false;
};

// fc:141-152, returned transposed (fc:907): rows t, n, bt
m3 tnb_rows(v3 n)
{
	float s = n.z < 0.0f ? -1.0f : 1.0f;
	float a = -1.0f / (s + n.z);
	float b = n.x * n.y * a;
	m3 m;
	m.r0 = V(1.0f + s * n.x * n.x * a, s * b, -s * n.x);
	m.r1 = n;
	m.r2 = V(b, s + n.y * n.y * a, -n.y);
	return m;
}

// fc:154-170
v3 map_to_unit_sphere(const Fn &fn, float vx, float vy)
{
	float cos_theta = 2.0f * vx - 1.0f;
	float phi = TWO_PI * vy;
	float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
	float sp, cp;
	fn.sincos(phi, &sp, &cp);
	return V(sin_theta * cp, cos_theta, sin_theta * sp);
}

// fc:180-184
void cartesian_to_spherical(const Fn &fn, v3 vec, float *theta, float *phi)
{
	*theta = fn.acos_(vec.y);
	*phi = fn.atan2_(vec.x, vec.z) + PI;
}

// GL_LINEAR, CLAMP_TO_EDGE lookup into an RGB8 image stored as RGB16F (display.cpp:174-189):
// texel = fp16(byte / 255), bilinear weights in fp32.
v3 texel_rgb8(const uint8_t *img, uint32_t w, uint32_t h, int x, int y)
{
	x = x < 0 ? 0 : (x >= (int) w ? (int) w - 1 : x);
	y = y < 0 ? 0 : (y >= (int) h ? (int) h - 1 : y);
	const uint8_t *p = img + ((size_t) y * w + (size_t) x) * 3;
	return V(q16((float) p[0] / 255.0f), q16((float) p[1] / 255.0f), q16((float) p[2] / 255.0f));
}
v3 sample_bilinear_clamp(const uint8_t *img, uint32_t w, uint32_t h, float s, float t)
{
	float x = s * (float) w - 0.5f, y = t * (float) h - 0.5f;
	float fx = floorf(x), fy = floorf(y);
	float ax = x - fx, ay = y - fy;
	int x0 = (int) fx, y0 = (int) fy;
	v3 c00 = texel_rgb8(img, w, h, x0, y0), c10 = texel_rgb8(img, w, h, x0 + 1, y0);
	v3 c01 = texel_rgb8(img, w, h, x0, y0 + 1), c11 = texel_rgb8(img, w, h, x0 + 1, y0 + 1);
	v3 top = c00 * (1.0f - ax) + c10 * ax;
	v3 bot = c01 * (1.0f - ax) + c11 * ax;
	return top * (1.0f - ay) + bot * ay;
}

// fc:186-201, returns the f16vec3 widened
v3 sky_color(const Ctx &cx, v3 dir)
{
	if (!cx.sc.env || cx.sc.env_w == 0 || cx.sc.env_h == 0) return V(0, 0, 0); // texture without storage
	float theta, phi;
	cartesian_to_spherical(cx.fn, dir, &theta, &phi);
	phi += PI / 8.0f;
	if (phi < 0.0f) phi += TWO_PI;
	if (phi > TWO_PI) phi -= TWO_PI;
	return q16(sample_bilinear_clamp(cx.sc.env, cx.sc.env_w, cx.sc.env_h, phi / TWO_PI, theta / PI));
}

// fc:458-472
v3 oren_nayar_brdf(const Fn &fn, v3 albedo, float roughness, v3 wi, v3 wo)
{
	float theta_i, phi_i, theta_o, phi_o;
	cartesian_to_spherical(fn, wi, &theta_i, &phi_i);
	cartesian_to_spherical(fn, wo, &theta_o, &phi_o);
	float sq = roughness * roughness;
	float A = 1.0f - 0.5f * sq / (sq + 0.33f);
	float B = 0.45f * sq / (sq + 0.09f);
	float alpha = fmax_(theta_i, theta_o);
	float beta = fmin_(theta_i, theta_o);
	float k = A + B * fmax_(0.0f, fn.cos_(phi_i - phi_o)) * fn.sin_(alpha) * fn.tan_(beta);
	return (albedo / PI) * k;
}

inline float chi_plus(float v) { return v > 0.0f ? 1.0f : 0.0f; } // fc:474-477

// fc:505-514
float trowbridge_reitz_NDF(v3 wm, v3 wg, float alpha_ggx)
{
	float d = dot(wg, wm);
	float sa = fmax_(EPSILON * 10.0f, alpha_ggx * alpha_ggx);
	float inner = 1.0f + d * d * (sa - 1.0f);
	float denominator = PI * (inner * inner);
	return chi_plus(d) * sa / denominator;
}
// fc:517-524
float trowbridge_reitz_lambda(v3 dir, float alpha_ggx)
{
	float cos_theta = dir.y;
	float sin_theta_2 = fmax_(0.0f, 1.0f - cos_theta * cos_theta);
	float tan_theta = sqrtf(sin_theta_2) / cos_theta;
	float a = 1.0f / (alpha_ggx * tan_theta);
	return -0.5f + 0.5f * sqrtf(1.0f + 1.0f / (a * a));
}
// fc:527-532
float G2(v3 wi, v3 wo, v3 wm, float alpha_ggx)
{
	float result = chi_plus(dot(wm, wo)) * chi_plus(dot(wm, wi));
	result /= (1.0f + trowbridge_reitz_lambda(wo, alpha_ggx) + trowbridge_reitz_lambda(wi, alpha_ggx));
	return result;
}
// fc:535-538
v3 fresnel(v3 wm, v3 wi, v3 F0)
{
	float x = 1.0f - fmin_(1.0f, fmax_(0.0f, dot(wm, wi)));
	float x2 = x * x;
	float p = x2 * x2 * x;
	return F0 + (V(1.0f, 1.0f, 1.0f) - F0) * p;
}
// fc:540-549
v3 GGX_Smith_BRDF_non_IS(v3 wo, v3 wi, v3 F0, float alpha_ggx)
{
	v3 wg = V(0.0f, 1.0f, 0.0f);
	v3 wm = normalize(wo + wi);
	v3 result = fresnel(wm, wi, F0) * G2(wi, wo, wm, alpha_ggx) * trowbridge_reitz_NDF(wm, wg, alpha_ggx);
	return result / fmax_(EPSILON * 10.0f, 4.0f * fabsf(dot(wg, wi)) * fabsf(dot(wg, wo)));
}
// fc:551-560
v3 GGX_Smith_BRDF(v3 wo, v3 wm, v3 wi, v3 F0, float alpha_ggx)
{
	v3 wg = V(0.0f, 1.0f, 0.0f);
	v3 result = fresnel(wm, wi, F0) * G2(wi, wo, wm, alpha_ggx) * fabsf(dot(wo, wm));
	return result / fmax_(EPSILON * 10.0f, fabsf(dot(wg, wo)) * fabsf(dot(wg, wm)));
}

// fc:563-589 (using_NEE is always false on the MIS path, so type 2 importance-samples GGX)
v3 pick_wi(const Fn &fn, v3 wo, const Mat &mat, bool using_NEE, v3 *wm, float *cos_theta, float *pdf, uint32_t &rng)
{
	bool is_ggx = !using_NEE;
	float ty = mat.d2[3];
	if (ty == -1.0f || ty == 0.0f || ty == 1.0f || (ty == 2.0f && !is_ggx)) {
		*wm = V(0.0f, 1.0f, 0.0f);
		float r1 = rnd(rng), r2 = rnd(rng);
		v3 wi = normalize(map_to_unit_sphere(fn, r1, r2) + *wm);
		*cos_theta = dot(*wm, wi);
		*pdf = *cos_theta / PI;
		return wi;
	}
	if (ty == 2.0f && is_ggx) {
		*cos_theta = 1.0f;
		*pdf = 1.0f;
		float alpha_ggx = mat.d1[3];
		float alpha_squared = alpha_ggx * alpha_ggx; // fc:481
		float e1 = rnd(rng), e2 = rnd(rng);
		float theta = fn.acos_(sqrtf((1.0f - e1) / ((alpha_squared - 1.0f) * e1 + 1.0f)));
		float phi = TWO_PI * e2;
		float st, ct, sp, cp;
		fn.sincos(theta, &st, &ct);
		fn.sincos(phi, &sp, &cp);
		*wm = V(st * cp, ct, st * sp);
		v3 I = -wo; // reflect(I, N) = I - 2 dot(N, I) N
		v3 wi = normalize(I - (2.0f * dot(*wm, I)) * *wm);
		return wi;
	}
	// material not handled (fc:588): out parameters are undefined in GLSL; contract says zero
	*wm = V(0, 0, 0); *cos_theta = 0.0f; *pdf = 0.0f;
	return V(INFINITY, INFINITY, INFINITY);
}

// sampler2DArray lookup, 512x512 RGB8 at mip 0 (texture() in a compute shader: lod 0, magnification filter); wrap modes
// and filter are the array-wide sampler state of loader.cpp:242-251 (defaults GL_REPEAT / GL_LINEAR, loader.cpp:50-58)
v3 sample_texture(const Scene &sc, float u, float v, float layer)
{
	if (!sc.tex || sc.tex_layers == 0) return V(0, 0, 0);
	int l = (int) floorf(layer + 0.5f);
	l = l < 0 ? 0 : (l >= (int) sc.tex_layers ? (int) sc.tex_layers - 1 : l);
	const uint8_t *img = sc.tex + (size_t) l * 512 * 512 * 3;
	// OpenGL 4.6 section 8.14.2, table 8.21: wrap(i) per mode; 0 = the loader's defaults GL_REPEAT / GL_LINEAR
	auto wrap = [](int i, uint32_t mode) {
		if (mode == 33071u) return i < 0 ? 0 : (i > 511 ? 511 : i);               // GL_CLAMP_TO_EDGE
		if (mode == 33648u) {                                                       // GL_MIRRORED_REPEAT
			int a = ((i % 1024) + 1024) % 1024 - 512;
			a = a >= 0 ? a : -(1 + a);
			return 511 - a;
		}
		return ((i % 512) + 512) % 512;                                             // GL_REPEAT
	};
	auto tx = [&](int xx, int yy) {
		const uint8_t *p = img + ((size_t) yy * 512 + (size_t) xx) * 3;
		return V((float) p[0] / 255.0f, (float) p[1] / 255.0f, (float) p[2] / 255.0f);
	};
	if (sc.tex_mag_filter == 9728u) {                                               // GL_NEAREST
		return tx(wrap((int) floorf(u * 512.0f), sc.tex_wrap_s), wrap((int) floorf(v * 512.0f), sc.tex_wrap_t));
	}
	const float x = u * 512.0f - 0.5f, y = v * 512.0f - 0.5f;
	const float fx = floorf(x), fy = floorf(y);
	const float ax = x - fx, ay = y - fy;
	const int x0 = wrap((int) fx, sc.tex_wrap_s), y0 = wrap((int) fy, sc.tex_wrap_t);
	const int x1 = wrap((int) fx + 1, sc.tex_wrap_s), y1 = wrap((int) fy + 1, sc.tex_wrap_t);
	const v3 top = tx(x0, y0) * (1.0f - ax) + tx(x1, y0) * ax;
	const v3 bot = tx(x0, y1) * (1.0f - ax) + tx(x1, y1) * ax;
	return top * (1.0f - ay) + bot * ay;
}

// fc:591-643
v3 calc_BRDF(const Ctx &cx, v3 wo, v3 wm, v3 wi, const Mat &mat, const uint16_t uvs[2], bool using_NEE)
{
	v3 brdf = V(1.0f, 1.0f, 0.0f);
	int mat_type = (int) mat.d2[3];
	v3 albedo = V(mat.d1[0], mat.d1[1], mat.d1[2]);
	v3 F0 = V(mat.d2[0], mat.d2[1], mat.d2[2]);
	v3 diffuse_texture = V(0, 0, 0);
	if (mat.d3[3] > -1.0f) {
		diffuse_texture = sample_texture(cx.sc, h2f(uvs[0]), h2f(uvs[1]), mat.d3[3]);
		albedo = albedo * diffuse_texture;
	}
	if (mat_type == -1) brdf = V(0, 0, 0);
	if (mat_type == 1) brdf = oren_nayar_brdf(cx.fn, albedo, mat.d1[3], wi, wo);
	if (mat_type == 0) brdf = albedo / PI;
	if (mat_type == 2) {
		if (mat.d3[3] > -1.0f) F0 = F0 * diffuse_texture;
		if (!using_NEE) brdf = GGX_Smith_BRDF(wo, wm, wi, F0, mat.d1[3]);
		else brdf = GGX_Smith_BRDF_non_IS(wo, wi, F0, mat.d1[3]);
	}
	return brdf;
}

// fc:695-700 (|e1||e2|/2, not |e1 x e2|/2: SURVEY quirk 7)
float area_triangle(const Tri &t)
{
	v3 v0 = V(t.d1[0], t.d1[1], t.d1[2]);
	v3 e1 = V(t.d2[0], t.d2[1], t.d2[2]) - v0;
	v3 e2 = V(t.d3[0], t.d3[1], t.d3[2]) - v0;
	return sqrtf(dot(e1, e1) * dot(e2, e2)) * 0.5f;
}
inline float area_sphere(float r) { return 4.0f * PI * r * r; } // fc:702-705

// fc:707-723
v3 map_to_triangle(float u, float v, v3 v0, v3 v1, v3 v2)
{
	if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
	v3 p = u * (v1 - v0) + v * (v2 - v0);
	return p + v0;
}

inline float balance_heuristic(float a, float b) { return a / (a + b); } // fc:869-872
inline v3 hn(const uint16_t n[3]) { return V(h2f(n[0]), h2f(n[1]), h2f(n[2])); }
inline v3 le_of(const Mat &m) { return V(m.d3[0], m.d3[1], m.d3[2]); }

// fc:874-1081
v3 estimator_path_tracing_mis(Ctx &cx, v3 ro, v3 rd, uint32_t rng_state, uint32_t frame_bounces)
{
	const Scene &sc = cx.sc;
	v3 color = V(0, 0, 0);
	v3 throughput_term = V(1, 1, 1);

	Hit data;
	cx.cn.prim++;
	bool hit_anything = intersect(sc, ro, rd, data, cx.cn);
	if (!hit_anything) return sky_color(cx, rd) * ENVIRONMENT_MAP_LE;

	v3 y = (ro + rd * h2f(data.t)) + NORMAL_OFFSET * hn(data.n);
	uint16_t normal_y[3] = {data.n[0], data.n[1], data.n[2]};
	Mat mat_y = sc.mats[data.mat];

	color = color + le_of(mat_y);
	if (cx.dbg) printf("[cpu] stage 0 col %a %a %a | ro %a %a %a rd %a %a %a | hit prim %u t %a u %a v %a | n %a %a %a mat %u y %a %a %a\n", color.x, color.y, color.z,
	                   ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, data.obj32, h2f(data.t), data.u, data.v, h2f(data.n[0]), h2f(data.n[1]), h2f(data.n[2]), data.mat, y.x, y.y, y.z);

	uint32_t num_light_sources = sc.n_light_tris + sc.n_light_spheres;
	uint32_t num_bounces = frame_bounces + 1;
	for (uint32_t b = 1; b < num_bounces; b++) {
		v3 x = y;
		uint16_t normal_x[3] = {normal_y[0], normal_y[1], normal_y[2]};
		Mat mat_x = mat_y;
		float alpha_squared = mat_x.d1[3] * mat_x.d1[3];

		m3 tnb = tnb_rows(hn(normal_x));
		m3 inverse_tnb = inverse(tnb);
		v3 wo = normalize(mul(tnb, -rd));

		bool can_use_NEE = num_light_sources > 0;
		can_use_NEE = can_use_NEE && (mat_x.d2[3] == 0.0f || mat_x.d2[3] == 1.0f ||
		                              (mat_x.d2[3] == 2.0f && alpha_squared > NEE_SPECULAR_ROUGHNESS_CUTOFF));
		if (can_use_NEE) {
			v3 direct_illumination = V(0, 0, 0);
			float pdf_pick_light = 1.0f / (float) num_light_sources;
			int picked_light_source = (int) (pcg(rng_state) % num_light_sources);
			float light_area = 0.0f;
			v3 y_nee = V(0, 0, 0);
			Mat light_source_mat;
			memset(&light_source_mat, 0, sizeof light_source_mat);
			if (sc.n_light_tris > 0 && picked_light_source < (int) sc.n_light_tris) {
				const Tri &ls = sc.tris[sc.light_tris[picked_light_source]];
				light_area = area_triangle(ls);
				light_source_mat = sc.mats[ls.d4[3]];
				float r1 = rnd(rng_state), r2 = rnd(rng_state);
				y_nee = map_to_triangle(r1, r2, V(ls.d1[0], ls.d1[1], ls.d1[2]), V(ls.d2[0], ls.d2[1], ls.d2[2]),
				                        V(ls.d3[0], ls.d3[1], ls.d3[2]));
			} else if (sc.n_light_spheres > 0) {
				const Sph &ls = sc.spheres[sc.light_spheres[picked_light_source - (int) sc.n_light_tris]];
				float radius = ls.d[3];
				light_area = area_sphere(radius);
				light_source_mat = sc.mats[ls.mat[0]];
				float r1 = rnd(rng_state), r2 = rnd(rng_state);
				y_nee = map_to_unit_sphere(cx.fn, r1, r2) * radius + V(ls.d[0], ls.d[1], ls.d[2]);
			}
			if (cx.dbg) printf("[cpu]   nee b %u y_nee %08x %08x %08x area %a rng %08x\n", b, fbits(y_nee.x), fbits(y_nee.y), fbits(y_nee.z), light_area, rng_state);
			v3 dist_vec = y_nee - x;
			v3 shadow_rd = normalize(dist_vec);
			v3 shadow_ro = x;
			float squared_dist = dot(dist_vec, dist_vec);
			float cos_theta_x = fmax_(0.0f, dot(hn(normal_x), shadow_rd));
			if (cx.dbg) printf("[cpu]   x %08x %08x %08x srd %08x %08x %08x d2 %08x cosx %08x\n", fbits(x.x), fbits(x.y), fbits(x.z), fbits(shadow_rd.x), fbits(shadow_rd.y), fbits(shadow_rd.z), fbits(squared_dist), fbits(cos_theta_x));

			Hit shadow_data;
			cx.cn.shadow++;
			bool shadow_hit_anything = intersect(sc, shadow_ro, shadow_rd, shadow_data, cx.cn);
			v3 dlt = y_nee - (shadow_ro + shadow_rd * h2f(shadow_data.t));
			if (shadow_hit_anything && dlt.x < FLOAT_COMPARE && dlt.y < FLOAT_COMPARE && dlt.z < FLOAT_COMPARE) {
				v3 wi = normalize(mul(tnb, shadow_rd));
				v3 wm = normalize(mul(tnb, hn(data.n)));
				v3 BRDF = calc_BRDF(cx, wo, wm, wi, mat_x, data.uv, true);
				float cos_theta_y = dot(hn(shadow_data.n), -shadow_rd);
				if (cos_theta_y > 0.0f) {
					float pdf_pick_point_on_light = 1.0f / light_area;
					float pdf_light_area = pdf_pick_light * pdf_pick_point_on_light;
					float pdf_NEE_sa = pdf_light_area * squared_dist / cos_theta_y;
					float pdf_BSDF_sa = cos_theta_x / PI;
					float wNEE = balance_heuristic(pdf_NEE_sa, pdf_BSDF_sa);
					direct_illumination = direct_illumination + le_of(light_source_mat) * BRDF * cos_theta_x * wNEE / pdf_NEE_sa;
				}
			}
			direct_illumination = direct_illumination / 1.0f; // NUM_SHADOW_RAYS = 1
			color = color + throughput_term * direct_illumination;
		}

		ro = x + NORMAL_OFFSET * hn(normal_x);

		float cos_theta_x, pdf_BSDF_sa;
		v3 wm;
		v3 wi = pick_wi(cx.fn, wo, mat_x, false, &wm, &cos_theta_x, &pdf_BSDF_sa, rng_state);
		rd = normalize(mul(inverse_tnb, wi));

		v3 BRDF = calc_BRDF(cx, wo, wm, wi, mat_x, data.uv, false);

		cx.cn.bounce++;
		hit_anything = intersect(sc, ro, rd, data, cx.cn);
		if (!hit_anything) {
			color = color + ENVIRONMENT_MAP_LE * throughput_term * BRDF * cos_theta_x * sky_color(cx, rd) / pdf_BSDF_sa;
			return color;
		}

		float cos_theta_y = dot(hn(data.n), -rd);
		normal_y[0] = data.n[0]; normal_y[1] = data.n[1]; normal_y[2] = data.n[2];
		y = (ro + rd * h2f(data.t)) + NORMAL_OFFSET * hn(normal_y);
		mat_y = sc.mats[data.mat];

		float wBSDF = 1.0f;
		if (can_use_NEE && cos_theta_y > 0.0f) {
			if (mat_y.d2[3] == -1.0f) {
				float pdf_NEE_area = 0.0f;
				if (data.type == 0) {
					pdf_NEE_area = 1.0f / area_triangle(sc.tris[data.obj16 < sc.n_tris ? data.obj16 : 0]);
				} else if (data.type == 1) {
					pdf_NEE_area = 1.0f / area_sphere(sc.spheres[data.obj16 < sc.n_spheres ? data.obj16 : 0].d[3]);
				}
				pdf_NEE_area /= (float) num_light_sources;
				float th = h2f(data.t);
				float pdf_NEE_sa = pdf_NEE_area * th * th / cos_theta_y;
				wBSDF = balance_heuristic(pdf_BSDF_sa, pdf_NEE_sa);
			}
			color = color + throughput_term * BRDF * le_of(mat_y) * cos_theta_y * wBSDF / pdf_BSDF_sa;
		}

		throughput_term = throughput_term * (BRDF * cos_theta_x / pdf_BSDF_sa);

		if (!can_use_NEE) {
			if (mat_x.d2[3] == 2.0f) color = color + throughput_term * le_of(mat_y);
		}
		if (cx.dbg) printf("[cpu] stage %u col %a %a %a thr %a %a %a | ro %a %a %a rd %a %a %a | hit prim %u t %a u %a v %a | n %a %a %a mat %u y %a %a %a | brdf %a %a %a cos %a pdf %a\n", b, color.x, color.y, color.z,
		                   throughput_term.x, throughput_term.y, throughput_term.z, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, data.obj32, h2f(data.t), data.u, data.v,
		                   h2f(data.n[0]), h2f(data.n[1]), h2f(data.n[2]), data.mat, y.x, y.y, y.z, BRDF.x, BRDF.y, BRDF.z, cos_theta_x, pdf_BSDF_sa);
	}
	return color;
}

// Light sampling shared by the NEE and MIS estimators (fc:765-808 == fc:926-956): 1 raw draw + 2 rand draws.
struct LightSample { v3 y; float area; Mat mat; };
LightSample sample_light(Ctx &cx, uint32_t num_light_sources, uint32_t &rng_state)
{
	const Scene &sc = cx.sc;
	LightSample L;
	L.y = V(0, 0, 0); L.area = 0.0f;
	memset(&L.mat, 0, sizeof L.mat);
	int picked = (int) (pcg(rng_state) % num_light_sources);
	if (sc.n_light_tris > 0 && picked < (int) sc.n_light_tris) {
		const Tri &ls = sc.tris[sc.light_tris[picked]];
		L.area = area_triangle(ls);
		L.mat = sc.mats[ls.d4[3]];
		float r1 = rnd(rng_state), r2 = rnd(rng_state);
		L.y = map_to_triangle(r1, r2, V(ls.d1[0], ls.d1[1], ls.d1[2]), V(ls.d2[0], ls.d2[1], ls.d2[2]), V(ls.d3[0], ls.d3[1], ls.d3[2]));
	} else if (sc.n_light_spheres > 0) {
		const Sph &ls = sc.spheres[sc.light_spheres[picked - (int) sc.n_light_tris]];
		float radius = ls.d[3];
		L.area = area_sphere(radius);
		L.mat = sc.mats[ls.mat[0]];
		float r1 = rnd(rng_state), r2 = rnd(rng_state);
		L.y = map_to_unit_sphere(cx.fn, r1, r2) * radius + V(ls.d[0], ls.d[1], ls.d[2]);
	}
	return L;
}

// fc:647-693 -- BSDF sampling only (commented out at fc:1109; ORACLE_FLAG_EST_BRDF selects it)
v3 estimator_path_tracing_BRDF(Ctx &cx, v3 ro, v3 rd, uint32_t rng_state, uint32_t frame_bounces)
{
	const Scene &sc = cx.sc;
	v3 color = V(0, 0, 0);
	v3 throughput_term = V(1, 1, 1);
	for (uint32_t b = 0; b < frame_bounces; b++) {
		Hit data;
		if (b == 0) cx.cn.prim++; else cx.cn.bounce++;
		if (!intersect(sc, ro, rd, data, cx.cn)) {
			color = color + throughput_term * sky_color(cx, rd) * ENVIRONMENT_MAP_LE;
			return color;
		}
		Mat mat = sc.mats[data.mat];
		m3 tnb = tnb_rows(hn(data.n));
		m3 inverse_tnb = inverse(tnb);
		color = color + throughput_term * le_of(mat);
		ro = (ro + (rd * h2f(data.t))) + (NORMAL_OFFSET * hn(data.n));
		float pdf, cos_theta;
		v3 wm;
		v3 wo = normalize(mul(tnb, -rd));
		v3 wi = pick_wi(cx.fn, wo, mat, false, &wm, &cos_theta, &pdf, rng_state);
		rd = normalize(mul(inverse_tnb, wi));
		v3 BRDF = calc_BRDF(cx, wo, wm, wi, mat, data.uv, false);
		throughput_term = throughput_term * ((cos_theta * BRDF) / pdf); // fc:689: cos_theta * BRDF / pdf
	}
	return color;
}

// fc:725-867 -- next-event estimation only (commented out at fc:1110; ORACLE_FLAG_EST_NEE selects it)
v3 estimator_path_tracing_nee(Ctx &cx, v3 ro, v3 rd, uint32_t rng_state, uint32_t frame_bounces)
{
	const Scene &sc = cx.sc;
	v3 color = V(0, 0, 0);
	v3 throughput_term = V(1, 1, 1);
	bool prev_bounce_specular = false;
	for (uint32_t bounce = 0; bounce < frame_bounces; bounce++) {
		Hit data;
		if (bounce == 0) cx.cn.prim++; else cx.cn.bounce++;
		if (!intersect(sc, ro, rd, data, cx.cn)) {
			color = color + throughput_term * sky_color(cx, rd) * ENVIRONMENT_MAP_LE;
			return color;
		}
		m3 tnb = tnb_rows(hn(data.n));
		m3 inverse_tnb = inverse(tnb);
		v3 wo = normalize(mul(tnb, -rd));
		Mat mat = sc.mats[data.mat];
		uint32_t num_light_sources = sc.n_light_tris + sc.n_light_spheres;
		bool can_use_NEE = num_light_sources > 0;
		can_use_NEE = can_use_NEE && (mat.d2[3] == 0.0f || mat.d2[3] == 1.0f ||
		                              (mat.d2[3] == 2.0f && mat.d1[3] * mat.d1[3] > NEE_SPECULAR_ROUGHNESS_CUTOFF));
		if (mat.d2[3] == -1.0f) {
			if (bounce == 0 || (bounce > 0 && prev_bounce_specular)) color = color + throughput_term * le_of(mat);
			return color;
		} else if (can_use_NEE) {
			v3 direct_illumination = V(0, 0, 0);
			float pdf_pick_light = 1.0f / (float) num_light_sources;
			LightSample L = sample_light(cx, num_light_sources, rng_state);
			float pdf_pick_point_on_light = 1.0f / L.area;
			float pdf_light_area = pdf_pick_light * pdf_pick_point_on_light;
			v3 shadow_ro = (ro + rd * h2f(data.t)) + NORMAL_OFFSET * hn(data.n);
			v3 dist_vec = L.y - shadow_ro;
			v3 shadow_rd = normalize(dist_vec);
			Hit shadow_data;
			cx.cn.shadow++;
			bool shadow_hit_anything = intersect(sc, shadow_ro, shadow_rd, shadow_data, cx.cn);
			v3 shadow_ray_hit = shadow_ro + shadow_rd * h2f(shadow_data.t);
			v3 dl = L.y - shadow_ray_hit;
			if (shadow_hit_anything && fabsf(dl.x) <= FLOAT_COMPARE && fabsf(dl.y) <= FLOAT_COMPARE && fabsf(dl.z) <= FLOAT_COMPARE) {
				v3 wi = normalize(mul(tnb, shadow_rd));
				v3 wm = normalize(mul(tnb, hn(data.n)));
				v3 BRDF = calc_BRDF(cx, wo, wm, wi, mat, data.uv, true);
				float cos_theta_x = fmax_(0.0f, dot(hn(data.n), shadow_rd));
				float cos_theta_y = fmax_(0.0f, dot(hn(shadow_data.n), -shadow_rd));
				float squared_dist = dot(dist_vec, dist_vec);
				float G = cos_theta_x * cos_theta_y / squared_dist;
				direct_illumination = direct_illumination + le_of(L.mat) * BRDF * G / pdf_light_area;
			}
			direct_illumination = direct_illumination / 1.0f;
			color = color + throughput_term * direct_illumination;
		}
		ro = ro + (rd * h2f(data.t) + NORMAL_OFFSET * hn(data.n)); // fc:848: ro += rd*t + offset*n
		v3 wm;
		float cos_theta_x, pdf;
		v3 wi = pick_wi(cx.fn, wo, mat, true, &wm, &cos_theta_x, &pdf, rng_state);
		rd = normalize(mul(inverse_tnb, wi));
		v3 BRDF = calc_BRDF(cx, wo, wm, wi, mat, data.uv, true);
		prev_bounce_specular = (mat.d2[3] == 2.0f && mat.d1[3] * mat.d1[3] <= NEE_SPECULAR_ROUGHNESS_CUTOFF);
		throughput_term = throughput_term * (BRDF * cos_theta_x / pdf);
	}
	return color;
}

// fc:1088-1107: the jittered pinhole ray of pixel (px, py); two draws, x first
void primary_ray(const float cam[12], float sw, float sh, int px, int py, uint32_t &rng_state, v3 *ro, v3 *rd)
{
	v3 cam_origin = V(cam[0], cam[1], cam[2]);
	v3 cam_forward = V(cam[4], cam[5], cam[6]);
	v3 cam_right = V(cam[8], cam[9], cam[10]);
	float grid_height = 2.0f;
	float grid_width = grid_height * sw / sh;
	v3 cam_up = normalize(cross(cam_right, cam_forward));
	v3 grid_x = cam_right * grid_width;
	v3 grid_y = cam_up * grid_height;
	v3 grid_origin = (cam_origin - (grid_x * 0.5f)) - (grid_y * 0.5f);
	grid_origin = grid_origin + 2.0f * cam_forward;
	float ox = rnd(rng_state) - 0.5f;
	float oy = rnd(rng_state) - 0.5f;
	float u = ((float) px + ox) / sw;
	float v = ((float) py + oy) / sh;
	v3 point_on_grid = (grid_origin + u * grid_x) + v * grid_y;
	*ro = cam_origin;
	*rd = normalize(point_on_grid - cam_origin);
}

// fc:1088-1112
v3 render_function(Ctx &cx, const float cam[12], float sw, float sh, int px, int py, uint32_t rng_state, uint32_t bounces, uint32_t flags)
{
	v3 cam_origin, ray_direction;
	primary_ray(cam, sw, sh, px, py, rng_state, &cam_origin, &ray_direction);
	// fc:1109-1111: the reference picks the estimator by (un)commenting one of these three lines
	if (flags & ORACLE_FLAG_EST_BRDF) return estimator_path_tracing_BRDF(cx, cam_origin, ray_direction, rng_state, bounces);
	if (flags & ORACLE_FLAG_EST_NEE) return estimator_path_tracing_nee(cx, cam_origin, ray_direction, rng_state, bounces);
	return estimator_path_tracing_mis(cx, cam_origin, ray_direction, rng_state, bounces);
}

} // namespace

// ===================================================================== C API
extern "C" {

int oracle_trace_closest(const oracle_scene *s, const float *ro, const float *rd, uint32_t n, uint32_t *prim,
                         uint16_t *t_h, uint8_t *kind, float *uv, uint32_t *steps, uint32_t *tests)
{
	Scene sc = view(s);
	for (uint32_t i = 0; i < n; i++) {
		Counters cn;
		Hit h;
		bool hit = intersect(sc, V(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), V(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]), h, cn);
		kind[i] = hit ? (uint8_t) (h.type + 1) : 0;
		prim[i] = hit ? h.obj32 : 0xffffffffu;
		t_h[i] = hit ? h.t : 0;
		if (uv) {
			// triangle: the barycentrics; sphere: HitData.uvs as the shader leaves them (fc:447-451 does not write uvs, so they
			// are the last accepted triangle's, or the zeroed record's)
			uv[2 * i] = !hit ? 0.0f : h.type == 0 ? h.u : h2f(h.uv[0]);
			uv[2 * i + 1] = !hit ? 0.0f : h.type == 0 ? h.v : h2f(h.uv[1]);
		}
		if (steps) steps[i] = (uint32_t) cn.steps;
		if (tests) tests[i] = (uint32_t) cn.tests;
	}
	return 0;
}

int oracle_render(const oracle_scene *s, const float cam[12], uint32_t width, uint32_t height, uint32_t first_frame,
                  uint32_t n_frames, const uint32_t *seeds, uint32_t bounces, float *rgba, uint32_t flags,
                  uint32_t n_threads, oracle_counters *out)
{
	Scene sc = view(s);
	if (n_threads == 0) n_threads = std::thread::hardware_concurrency();
	if (n_threads == 0) n_threads = 1;
	std::vector<Counters> per(n_threads);
	const char *dbg_env = getenv("ORACLE_DEBUG_PIXEL");
	const int dbg_px = dbg_env ? atoi(dbg_env) : -1;
	for (uint32_t f = 0; f < n_frames; f++) {
		uint32_t frame = first_frame + f, seed = seeds[f];
		// 8x8 pixel blocks handed out dynamically (one GLSL work group each, fc:8)
		uint32_t bx = (width + 7) / 8, by = (height + 7) / 8;
		std::atomic<uint32_t> next(0);
		auto work = [&](uint32_t tid) {
			Ctx cx{sc, Fn{(flags & ORACLE_FLAG_LIBM) != 0}, Counters()};
			for (;;) {
				uint32_t b = next.fetch_add(1);
				if (b >= bx * by) break;
				uint32_t x0 = (b % bx) * 8, y0 = (b / bx) * 8;
				for (uint32_t py = y0; py < y0 + 8 && py < height; py++)
					for (uint32_t px = x0; px < x0 + 8 && px < width; px++) {
						uint32_t rng_state = seed3(px, py, seed); // fc:1120
						cx.cn.paths++;
						cx.dbg = dbg_px == (int) (py * width + px);
						v3 color = render_function(cx, cam, (float) width, (float) height, (int) px, (int) py, rng_state, bounces, flags);
						float *p = rgba + ((size_t) py * width + px) * 4; // fc:1126-1129
						float fa = (float) frame / (float) (frame + 1), fb = 1.0f / (float) (frame + 1);
						p[0] = fa * p[0] + fb * color.x;
						p[1] = fa * p[1] + fb * color.y;
						p[2] = fa * p[2] + fb * color.z;
						p[3] = 1.0f;
					}
			}
			Counters &d = per[tid];
			d.prim += cx.cn.prim; d.shadow += cx.cn.shadow; d.bounce += cx.cn.bounce;
			d.steps += cx.cn.steps; d.tests += cx.cn.tests; d.paths += cx.cn.paths;
			if (cx.cn.max_stack > d.max_stack) d.max_stack = cx.cn.max_stack;
		};
		if (n_threads == 1) work(0);
		else {
			std::vector<std::thread> th;
			for (uint32_t t = 0; t < n_threads; t++) th.emplace_back(work, t);
			for (auto &t : th) t.join();
		}
	}
	if (out) {
		memset(out, 0, sizeof *out);
		for (auto &c : per) {
			out->paths += c.paths; out->rays_primary += c.prim; out->rays_shadow += c.shadow; out->rays_bounce += c.bounce;
			out->node_steps += c.steps; out->tri_tests += c.tests;
			if (c.max_stack > out->max_stack) out->max_stack = c.max_stack;
		}
	}
	return 0;
}

int oracle_math(int op, const float *a, const float *b, uint32_t n, float *out)
{
	Fn fn{false};
	for (uint32_t i = 0; i < n; i++) {
		float x = a[i], y = b ? b[i] : 0.0f, r;
		switch (op) {
		case ORACLE_OP_MIN: r = fmin_(x, y); break;
		case ORACLE_OP_MAX: r = fmax_(x, y); break;
		case ORACLE_OP_DIV: r = x / y; break;
		case ORACLE_OP_SQRT: r = sqrtf(x); break;
		case ORACLE_OP_F2H2F: r = q16(x); break;
		case ORACLE_OP_SIN: r = fn.sin_(x); break;
		case ORACLE_OP_COS: r = fn.cos_(x); break;
		case ORACLE_OP_TAN: r = fn.tan_(x); break;
		case ORACLE_OP_ACOS: r = fn.acos_(x); break;
		case ORACLE_OP_ATAN2: r = fn.atan2_(x, y); break;
		case ORACLE_OP_HADD: r = q16(q16(x) + q16(y)); break;
		case ORACLE_OP_HMUL: r = q16(q16(x) * q16(y)); break;
		case ORACLE_OP_RAND: { uint32_t st = fbits(x); r = rnd(st); break; }
		case ORACLE_OP_OREN_NAYAR: case ORACLE_OP_GGX_NEE: case ORACLE_OP_GGX_IS: {
			// composite BRDF probes: wo, wi cosine-sampled from the PCG stream seeded with the bits of a[i]
			uint32_t st = fbits(x);
			v3 up = V(0.0f, 1.0f, 0.0f);
			float r1 = rnd(st), r2 = rnd(st);
			v3 wo = normalize(map_to_unit_sphere(fn, r1, r2) + up);
			Mat m; memset(&m, 0, sizeof m);
			m.d1[3] = y; m.d2[0] = 0.944f; m.d2[1] = 0.776f; m.d2[2] = 0.373f; m.d2[3] = 2.0f; m.d3[3] = -1.0f;
			if (op == ORACLE_OP_OREN_NAYAR) {
				r1 = rnd(st); r2 = rnd(st);
				v3 wi = normalize(map_to_unit_sphere(fn, r1, r2) + up);
				r = oren_nayar_brdf(fn, V(1.0f, 0.5f, 0.25f), y, wi, wo).y;
			} else if (op == ORACLE_OP_GGX_NEE) {
				r1 = rnd(st); r2 = rnd(st);
				v3 wi = normalize(map_to_unit_sphere(fn, r1, r2) + up);
				r = GGX_Smith_BRDF_non_IS(wo, wi, V(m.d2[0], m.d2[1], m.d2[2]), y).x;
			} else {
				v3 wm; float c, pdf;
				v3 wi = pick_wi(fn, wo, m, false, &wm, &c, &pdf, st);
				v3 b3 = GGX_Smith_BRDF(wo, wm, wi, V(m.d2[0], m.d2[1], m.d2[2]), y);
				r = b3.x + wi.y;
			}
			break;
		}
		default: return -1;
		}
		out[i] = r;
	}
	return 0;
}

int oracle_probe(int what, const float *in, uint32_t n, float *out, uint32_t flags)
{
	Fn fn{(flags & ORACLE_FLAG_LIBM) != 0};
	auto put = [](float *o, v3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; };
	for (uint32_t i = 0; i < n; i++) {
		switch (what) {
		case ORACLE_PROBE_MAP_TO_UNIT_SPHERE: { const float *a = in + 2 * (size_t) i; put(out + 3 * (size_t) i, map_to_unit_sphere(fn, a[0], a[1])); break; }
		case ORACLE_PROBE_TNB_ROWS: {
			const float *a = in + 3 * (size_t) i;
			m3 m = tnb_rows(V(a[0], a[1], a[2]));
			put(out + 9 * (size_t) i, m.r0); put(out + 9 * (size_t) i + 3, m.r1); put(out + 9 * (size_t) i + 6, m.r2);
			break;
		}
		case ORACLE_PROBE_MAP_TO_TRIANGLE: {
			const float *a = in + 11 * (size_t) i;
			put(out + 3 * (size_t) i, map_to_triangle(a[0], a[1], V(a[2], a[3], a[4]), V(a[5], a[6], a[7]), V(a[8], a[9], a[10])));
			break;
		}
		case ORACLE_PROBE_COSINE_HEMISPHERE: {
			const float *a = in + 5 * (size_t) i;
			put(out + 3 * (size_t) i, normalize(map_to_unit_sphere(fn, a[0], a[1]) + V(a[2], a[3], a[4])));
			break;
		}
		case ORACLE_PROBE_NDF: { const float *a = in + 4 * (size_t) i; out[i] = trowbridge_reitz_NDF(V(a[0], a[1], a[2]), V(0.0f, 1.0f, 0.0f), a[3]); break; }
		case ORACLE_PROBE_PICK_WI: {
			const float *a = in + 6 * (size_t) i;
			uint32_t st = fbits(a[0]);
			Mat m; memset(&m, 0, sizeof m);
			m.d2[3] = a[4]; m.d1[3] = a[5]; m.d3[3] = -1.0f;
			v3 wm; float c, pdf;
			v3 wi = pick_wi(fn, V(a[1], a[2], a[3]), m, false, &wm, &c, &pdf, st);
			float *o = out + 8 * (size_t) i;
			put(o, wi); put(o + 3, wm); o[6] = c; o[7] = pdf;
			break;
		}
		case ORACLE_PROBE_CALC_BRDF: {
			const float *a = in + 18 * (size_t) i;
			static const Scene empty{};
			Ctx cx{empty, fn, Counters()};
			Mat m; memset(&m, 0, sizeof m);
			m.d1[0] = a[9]; m.d1[1] = a[10]; m.d1[2] = a[11]; m.d1[3] = a[16];
			m.d2[0] = a[12]; m.d2[1] = a[13]; m.d2[2] = a[14]; m.d2[3] = a[15]; m.d3[3] = -1.0f;
			const uint16_t uvs[2] = {0, 0};
			put(out + 3 * (size_t) i, calc_BRDF(cx, V(a[0], a[1], a[2]), V(a[3], a[4], a[5]), V(a[6], a[7], a[8]), m, uvs, a[17] != 0.0f));
			break;
		}
		case ORACLE_PROBE_OCT_DECODE: {
			uint16_t h[3];
			oct_decode(fbits(in[i]), h);
			put(out + 3 * (size_t) i, V(h2f(h[0]), h2f(h[1]), h2f(h[2])));
			break;
		}
		case ORACLE_PROBE_INVERSE3: {
			const float *a = in + 9 * (size_t) i;
			m3 m; m.r0 = V(a[0], a[1], a[2]); m.r1 = V(a[3], a[4], a[5]); m.r2 = V(a[6], a[7], a[8]);
			m3 r = inverse(m);
			put(out + 9 * (size_t) i, r.r0); put(out + 9 * (size_t) i + 3, r.r1); put(out + 9 * (size_t) i + 6, r.r2);
			break;
		}
		case ORACLE_PROBE_PRIMARY_RAY: {
			const float *a = in + 17 * (size_t) i;
			uint32_t rng_state = seed3((uint32_t) (int) a[14], (uint32_t) (int) a[15], fbits(a[16])); // fc:1120
			v3 ro, rd;
			primary_ray(a, a[12], a[13], (int) a[14], (int) a[15], rng_state, &ro, &rd);
			float *o = out + 7 * (size_t) i;
			put(o, ro); put(o + 3, rd);
			memcpy(o + 6, &rng_state, 4);
			break;
		}
		default: return -1;
		}
	}
	return 0;
}

int oracle_estimate(const oracle_scene *s, const float ro[3], const float rd[3], uint32_t n_samples, uint32_t bounces,
                    uint32_t flags, uint32_t seed, double out_mean[3])
{
	Scene sc = view(s);
	Ctx cx{sc, Fn{(flags & ORACLE_FLAG_LIBM) != 0}, Counters()};
	double acc[3] = {0, 0, 0};
	const v3 o = V(ro[0], ro[1], ro[2]), d = V(rd[0], rd[1], rd[2]);
	for (uint32_t i = 0; i < n_samples; i++) {
		uint32_t rng_state = seed3(i & 0xffffu, i >> 16, seed);
		v3 c;
		if (flags & ORACLE_FLAG_EST_BRDF) c = estimator_path_tracing_BRDF(cx, o, d, rng_state, bounces);
		else if (flags & ORACLE_FLAG_EST_NEE) c = estimator_path_tracing_nee(cx, o, d, rng_state, bounces);
		else c = estimator_path_tracing_mis(cx, o, d, rng_state, bounces);
		acc[0] += c.x; acc[1] += c.y; acc[2] += c.z;
	}
	for (int k = 0; k < 3; k++) out_mean[k] = n_samples ? acc[k] / n_samples : 0.0;
	return 0;
}

uint16_t oracle_f2h(float f) { return f2h(f); }
float oracle_h2f(uint16_t h) { return h2f(h); }

void oracle_pcg_stream(uint32_t x, uint32_t y, uint32_t seed, uint32_t n, uint32_t *out)
{
	uint32_t st = seed3(x, y, seed);
	for (uint32_t i = 0; i < n; i++) out[i] = pcg(st);
}

} // extern "C"
