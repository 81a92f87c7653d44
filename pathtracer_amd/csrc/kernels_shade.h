// Wavefront stages around k_trace: camera rays, MIS shading (estimator_path_tracing_mis of
// shaders/framebuffer.comp:874-1081 cut at its three intersect() calls), running-mean accumulation.
//
// A path lives in exactly one slot of the current queue; every per-path quantity is a structure-of-
// arrays column indexed by that slot, so all queue traffic is coalesced.  A shading launch reads queue
// `in`; lanes whose path continues are compacted into queue `out`: ballot + mbcnt prefix inside a wave,
// the four wave counts summed through LDS, ONE returning atomic per 256-thread block on the fill counter
// of the block's output region (block b fills region b % PT_SHARDS; see kernels_trace.h).  Paths that
// end blend straight into the RGBA32F image: a pixel carries at most one path per frame, so no atomics
// are needed there.
#pragma once
#include "kernels_trace.h"

#define PT_SHADE_BLOCK 256
#ifndef PT_SHADE_MIN_BLOCKS
#define PT_SHADE_MIN_BLOCKS 7 // 72 VGPRs -> 7 waves/SIMD: 1.7 % faster than the 75 VGPRs / 6 waves the compiler picks alone; 8 (64 VGPRs) spills
#endif
#define PT_EST_MIS 0u   // estimator_path_tracing_mis,  fc:874-1081 (the live one, fc:1111)
#define PT_EST_NEE 1u   // estimator_path_tracing_nee,  fc:725-867
#define PT_EST_BRDF 2u  // estimator_path_tracing_BRDF, fc:647-693

// columns of one path queue; every array has `stride` entries per component
struct PathQ {
	uint32_t *pix;   // path id in the batch: frame_slot * n_paths_frame + path-in-frame (k_raygen order); k_shade's own stages keep it in `meta`
	uint32_t *rng;   // PCG state
	uint32_t *meta;  // bit0: shadow ray present, bit1: can_use_NEE at x, bit2: material at x is SPECULAR_METAL; k_shade: path id << 4 above them
	float *thr;      // 3: throughput_term
	float *col;      // 3: color so far
	float *ray;      // 6: bounce ray ro, rd                                      (trace segment 0)
	float *sray;     // 6: shadow ray origin x; direction only in k_shade_alt's stages (trace segment 1 rebuilds it from x and y_nee)
	float *nee;      // 8: P = Le*BRDF*cos_x (3), A = pdf_light_area*d^2, pdf_BSDF(nee), y_nee (3)
	float *bs;       // 5: BRDF (3), cos_theta_x, pdf_BSDF_sa of the sampled direction
	uint4 *hit;      // closest hit of the bounce ray, one 16-byte record: (prim, t_h|kind<<16, u, v)
	uint4 *shit;     // closest hit of the shadow ray
};

struct ShadeArgs {
	DevScene sc;
	PathQ in, out;
	uint32_t stride;
	uint32_t cap;           // slots per queue region = blocks_per_region * PT_SHADE_BLOCK
	uint32_t blocks_per_region;
	uint32_t launch_bpr;    // blocks launched per region by the queue-driven stages (<= blocks_per_region)
	DevCtl *ctl;
	uint32_t in_q;          // index of `in` in ctl->qcount
	float cam[12];
	uint32_t width, height;
	uint32_t bounces;
	uint32_t stage;         // 0: after the primary trace; k: after the trace of bounce k
	uint32_t estimator;     // PT_EST_*
	uint32_t flags;
	float4 *sample;         // [n_batch][n_paths_frame]: the finished sample of every path of the batch
	float *fb;
	// batch: n_batch consecutive frames first_frame.. with their host-PCG seeds (FrameData.x), traced together
	uint32_t n_batch, first_frame, n_paths_frame, per_region;
	uint32_t seeds[PT_MAX_BATCH];
	// partition (raygen / resolve only)
	uint32_t rank, world, partition;
};

struct HitRec { // HitData, fc:214-222, members widened to float where they are float16_t
	float t; v3 n; uint32_t mat, obj16, type; float uvx, uvy;
};

// fc:203-210
PT_DEV v3 oct_decode(uint32_t packed)
{
	float x = (float) h_from_bits(packed & 0xffffu), y = (float) h_from_bits(packed >> 16);
	float z = q16((1.0f - __builtin_fabsf(x)) - __builtin_fabsf(y));
	float t = fmax_(-z, 0.0f);
	x = q16(x + (x >= 0.0f ? -t : t));
	y = q16(y + (y >= 0.0f ? -t : t));
	return q16(normalize(V(x, y, z)));
}

// the accept block of intersect_triangle (fc:294-310) and the sphere branch of intersect (fc:447-451),
// re-evaluated for the one hit the traversal kept.  ro, rd: the full-precision ray.
PT_DEV HitRec hit_attributes(const DevScene &sc, uint32_t prim, uint32_t tk, float u, float v, v3 ro, v3 rd)
{
	HitRec h;
	h.t = (float) h_from_bits(tk & 0xffffu);
	h.uvx = 0.0f; h.uvy = 0.0f;
	if ((tk >> 16) == 1u) {
		const float4 *tp = sc.tris + 4u * (size_t) prim;
		const float4 d1 = tp[0], d2 = tp[1], d3 = tp[2];
		const uint4 d4 = *(const uint4 *) (tp + 3);
		const v3 v0 = V(d1.x, d1.y, d1.z);
		const v3 e1 = V(d2.x, d2.y, d2.z) - v0;
		const v3 e2 = V(d3.x, d3.y, d3.z) - v0;
		const v3 n0 = oct_decode(d4.x), n1 = oct_decode(d4.y), n2 = oct_decode(d4.z);
		h.mat = d4.w & 0xffu;
		h.obj16 = prim & 0xffffu;
		h.type = 0;
		const float w = q16((1.0f - u) - v), uh = q16(u), vh = q16(v);
		const uint32_t p0 = __float_as_uint(d1.w), p1 = __float_as_uint(d2.w), p2 = __float_as_uint(d3.w);
		h.uvx = q16(q16(q16(w * (float) h_from_bits(p0 & 0xffffu)) + q16(uh * (float) h_from_bits(p1 & 0xffffu))) +
		            q16(vh * (float) h_from_bits(p2 & 0xffffu)));
		h.uvy = q16(q16(q16(w * (float) h_from_bits(p0 >> 16)) + q16(uh * (float) h_from_bits(p1 >> 16))) +
		            q16(vh * (float) h_from_bits(p2 >> 16))); // the same expression as tri_uv_bits (kernels_trace.h)
		const v3 sn = normalize(cross(e1, e2));
		const v3 wn0 = V(q16(w * n0.x), q16(w * n0.y), q16(w * n0.z));
		const v3 nn = normalize((wn0 + u * n1) + v * n2);
		const float sgn = dot(sn, q16(rd)) < 0.0f ? 1.0f : -1.0f;
		h.n = V(q16(q16(nn.x) * sgn), q16(q16(nn.y) * sgn), q16(q16(nn.z) * sgn));
	} else {
		const float4 s = sc.spheres[2u * prim];
		const uint4 sm = *(const uint4 *) (sc.spheres + 2u * prim + 1);
		h.n = q16(((ro + rd * h.t) - V(s.x, s.y, s.z)) / s.w);
		h.mat = sm.x & 0xffu;
		h.obj16 = prim & 0xffffu;
		h.type = 1;
		// fc:447-451 does not write result.uvs: they are still those of the last triangle the BVH pass accepted, which
		// k_trace's flush put into the record's third word (tri_uv_bits; +0 when no triangle was accepted)
		h.uvx = (float) h_from_bits(__float_as_uint(u) & 0xffffu);
		h.uvy = (float) h_from_bits(__float_as_uint(u) >> 16);
	}
	return h;
}

struct MatRec { float4 d1, d2, d3; };
PT_DEV MatRec load_mat(const DevScene &sc, uint32_t i)
{
	MatRec m;
	if (i >= sc.n_mats) i = 0;
	m.d1 = sc.mats[3u * i]; m.d2 = sc.mats[3u * i + 1]; m.d3 = sc.mats[3u * i + 2];
	return m;
}

// fc:141-152 transposed (fc:907)
PT_DEV m3 tnb_rows(v3 n)
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
PT_DEV v3 map_to_unit_sphere(float vx, float vy)
{
	float cos_theta = 2.0f * vx - 1.0f;
	float phi = PT_TWO_PI * vy;
	float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
	float sp, cp;
	ptm_sincos(phi, &sp, &cp);
	return V(sin_theta * cp, cos_theta, sin_theta * sp);
}

// fc:180-184
PT_DEV void cartesian_to_spherical(v3 vec, float *theta, float *phi)
{
	*theta = ptm_acos(vec.y);
	*phi = ptm_atan2(vec.x, vec.z) + PT_PI;
}

PT_DEV v3 texel_rgb8_h(const uint8_t *img, uint32_t w, uint32_t h, int x, int y)
{
	x = x < 0 ? 0 : (x >= (int) w ? (int) w - 1 : x);
	y = y < 0 ? 0 : (y >= (int) h ? (int) h - 1 : y);
	const uint8_t *p = img + ((size_t) y * w + (size_t) x) * 3;
	return V(q16((float) p[0] / 255.0f), q16((float) p[1] / 255.0f), q16((float) p[2] / 255.0f));
}

// fc:186-201 over an RGB8 image stored as RGB16F, GL_LINEAR + CLAMP_TO_EDGE (src/display/display.cpp:174-189)
PT_DEV v3 sky_color(const DevScene &sc, v3 dir)
{
	if (!sc.env || sc.env_w == 0 || sc.env_h == 0) return V(0.0f, 0.0f, 0.0f);
	float theta, phi;
	cartesian_to_spherical(dir, &theta, &phi);
	phi += PT_PI / 8.0f;
	if (phi < 0.0f) phi += PT_TWO_PI;
	if (phi > PT_TWO_PI) phi -= PT_TWO_PI;
	const float s = phi / PT_TWO_PI, t = theta / PT_PI;
	const float x = s * (float) sc.env_w - 0.5f, y = t * (float) sc.env_h - 0.5f;
	const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
	const float ax = x - fx, ay = y - fy;
	const int x0 = (int) fx, y0 = (int) fy;
	const v3 c00 = texel_rgb8_h(sc.env, sc.env_w, sc.env_h, x0, y0), c10 = texel_rgb8_h(sc.env, sc.env_w, sc.env_h, x0 + 1, y0);
	const v3 c01 = texel_rgb8_h(sc.env, sc.env_w, sc.env_h, x0, y0 + 1), c11 = texel_rgb8_h(sc.env, sc.env_w, sc.env_h, x0 + 1, y0 + 1);
	const v3 top = c00 * (1.0f - ax) + c10 * ax;
	const v3 bot = c01 * (1.0f - ax) + c11 * ax;
	return q16(top * (1.0f - ay) + bot * ay);
}

// sampler2DArray, 512x512 RGB8, level 0 (a compute shader's texture() has no derivatives: lod 0, the magnification
// filter applies).  Wrap modes and filter are the array-wide sampler state the loader sets from the glTF sampler
// (src/loader.cpp:242-251; defaults GL_REPEAT / GL_LINEAR, src/loader.cpp:50-58).
PT_DEV v3 sample_texture(const DevScene &sc, float u, float v, float layer)
{
	if (!sc.tex || sc.tex_layers == 0) return V(0.0f, 0.0f, 0.0f);
	int l = (int) __builtin_floorf(layer + 0.5f);
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
		return tx(wrap((int) __builtin_floorf(u * 512.0f), sc.tex_wrap_s), wrap((int) __builtin_floorf(v * 512.0f), sc.tex_wrap_t));
	}
	const float x = u * 512.0f - 0.5f, y = v * 512.0f - 0.5f;
	const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
	const float ax = x - fx, ay = y - fy;
	const int x0 = wrap((int) fx, sc.tex_wrap_s), y0 = wrap((int) fy, sc.tex_wrap_t);
	const int x1 = wrap((int) fx + 1, sc.tex_wrap_s), y1 = wrap((int) fy + 1, sc.tex_wrap_t);
	const v3 top = tx(x0, y0) * (1.0f - ax) + tx(x1, y0) * ax;
	const v3 bot = tx(x0, y1) * (1.0f - ax) + tx(x1, y1) * ax;
	return top * (1.0f - ay) + bot * ay;
}

// fc:458-472
PT_DEV v3 oren_nayar_brdf(v3 albedo, float roughness, v3 wi, v3 wo)
{
	float theta_i, phi_i, theta_o, phi_o;
	cartesian_to_spherical(wi, &theta_i, &phi_i);
	cartesian_to_spherical(wo, &theta_o, &phi_o);
	float sq = roughness * roughness;
	float A = 1.0f - 0.5f * sq / (sq + 0.33f);
	float B = 0.45f * sq / (sq + 0.09f);
	float alpha = fmax_(theta_i, theta_o);
	float beta = fmin_(theta_i, theta_o);
	float k = A + B * fmax_(0.0f, ptm_cos(phi_i - phi_o)) * ptm_sin(alpha) * ptm_tan(beta);
	return (albedo / PT_PI) * k;
}

PT_DEV float chi_plus(float v) { return v > 0.0f ? 1.0f : 0.0f; }
// fc:505-514
PT_DEV float ggx_ndf(v3 wm, v3 wg, float alpha_ggx)
{
	float d = dot(wg, wm);
	float sa = fmax_(PT_EPSILON * 10.0f, alpha_ggx * alpha_ggx);
	float inner = 1.0f + d * d * (sa - 1.0f);
	float denominator = PT_PI * (inner * inner);
	return chi_plus(d) * sa / denominator;
}
// fc:517-524
PT_DEV float ggx_lambda(v3 dir, float alpha_ggx)
{
	float cos_theta = dir.y;
	float sin_theta_2 = fmax_(0.0f, 1.0f - cos_theta * cos_theta);
	float tan_theta = __builtin_sqrtf(sin_theta_2) / cos_theta;
	float a = 1.0f / (alpha_ggx * tan_theta);
	return -0.5f + 0.5f * __builtin_sqrtf(1.0f + 1.0f / (a * a));
}
// fc:527-532
PT_DEV float ggx_g2(v3 wi, v3 wo, v3 wm, float alpha_ggx)
{
	float result = chi_plus(dot(wm, wo)) * chi_plus(dot(wm, wi));
	result /= (1.0f + ggx_lambda(wo, alpha_ggx) + ggx_lambda(wi, alpha_ggx));
	return result;
}
// fc:535-538
PT_DEV v3 fresnel(v3 wm, v3 wi, v3 F0)
{
	float x = 1.0f - fmin_(1.0f, fmax_(0.0f, dot(wm, wi)));
	float x2 = x * x;
	float p = x2 * x2 * x;
	return F0 + (V(1.0f, 1.0f, 1.0f) - F0) * p;
}
// fc:540-549
PT_DEV v3 ggx_brdf_non_is(v3 wo, v3 wi, v3 F0, float alpha_ggx)
{
	v3 wg = V(0.0f, 1.0f, 0.0f);
	v3 wm = normalize(wo + wi);
	v3 result = fresnel(wm, wi, F0) * ggx_g2(wi, wo, wm, alpha_ggx) * ggx_ndf(wm, wg, alpha_ggx);
	return result / fmax_(PT_EPSILON * 10.0f, 4.0f * __builtin_fabsf(dot(wg, wi)) * __builtin_fabsf(dot(wg, wo)));
}
// fc:551-560
PT_DEV v3 ggx_brdf_is(v3 wo, v3 wm, v3 wi, v3 F0, float alpha_ggx)
{
	v3 wg = V(0.0f, 1.0f, 0.0f);
	v3 result = fresnel(wm, wi, F0) * ggx_g2(wi, wo, wm, alpha_ggx) * __builtin_fabsf(dot(wo, wm));
	return result / fmax_(PT_EPSILON * 10.0f, __builtin_fabsf(dot(wg, wo)) * __builtin_fabsf(dot(wg, wm)));
}

// fc:591-643
PT_DEV v3 calc_brdf(const DevScene &sc, v3 wo, v3 wm, v3 wi, const MatRec &mat, float uvx, float uvy, bool using_nee)
{
	v3 brdf = V(1.0f, 1.0f, 0.0f);
	const int mat_type = (int) mat.d2.w;
	v3 albedo = V(mat.d1.x, mat.d1.y, mat.d1.z);
	v3 F0 = V(mat.d2.x, mat.d2.y, mat.d2.z);
	v3 diffuse_texture = V(0.0f, 0.0f, 0.0f);
	if (mat.d3.w > -1.0f) {
		diffuse_texture = sample_texture(sc, uvx, uvy, mat.d3.w);
		albedo = albedo * diffuse_texture;
	}
	if (mat_type == -1) brdf = V(0.0f, 0.0f, 0.0f);
	if (mat_type == 1) brdf = oren_nayar_brdf(albedo, mat.d1.w, wi, wo);
	if (mat_type == 0) brdf = albedo / PT_PI;
	if (mat_type == 2) {
		if (mat.d3.w > -1.0f) F0 = F0 * diffuse_texture;
		if (!using_nee) brdf = ggx_brdf_is(wo, wm, wi, F0, mat.d1.w);
		else brdf = ggx_brdf_non_is(wo, wi, F0, mat.d1.w);
	}
	return brdf;
}

// fc:695-700
PT_DEV float area_triangle(const DevScene &sc, uint32_t i)
{
	if (i >= sc.n_tris) i = 0;
	const float4 *tp = sc.tris + 4u * (size_t) i;
	const float4 d1 = tp[0], d2 = tp[1], d3 = tp[2];
	const v3 v0 = V(d1.x, d1.y, d1.z);
	const v3 e1 = V(d2.x, d2.y, d2.z) - v0, e2 = V(d3.x, d3.y, d3.z) - v0;
	return __builtin_sqrtf(dot(e1, e1) * dot(e2, e2)) * 0.5f;
}
PT_DEV float area_sphere(float r) { return 4.0f * PT_PI * r * r; }

// queue columns: each slot is read once and written once per stage -> streaming (non-temporal) accesses, see kernels_trace.h
#ifndef PT_NT_SHADE
#define PT_NT_SHADE 0
#endif
#if PT_NT_SHADE
#define PT_QLD(p) __builtin_nontemporal_load(p)
#define PT_QST(v, p) __builtin_nontemporal_store(v, p)
#else
#define PT_QLD(p) (*(p))
#define PT_QST(v, p) (*(p) = (v))
#endif
PT_DEV v3 ld3(const float *p, uint32_t i, uint32_t st) { return V(PT_QLD(&p[i]), PT_QLD(&p[st + i]), PT_QLD(&p[2 * st + i])); }
PT_DEV void st3(float *p, uint32_t i, uint32_t st, v3 v) { PT_QST(v.x, &p[i]); PT_QST(v.y, &p[st + i]); PT_QST(v.z, &p[2 * st + i]); }
PT_DEV uint4 ldq4(const uint4 *p) { const pt_u32x4 r = PT_QLD((const pt_u32x4 *) p); return make_uint4(r.x, r.y, r.z, r.w); }

// Light sampling shared by the MIS and NEE estimators (fc:926-956 == fc:765-808): one raw draw picks the light, two
// rand draws the point on it.
struct LightPick { v3 y, le; float area; };
PT_DEV LightPick sample_light(const DevScene &sc, uint32_t num_light_sources, uint32_t &rng)
{
	LightPick L;
	L.y = V(0.0f, 0.0f, 0.0f); L.le = V(0.0f, 0.0f, 0.0f); L.area = 0.0f;
	const int picked = (int) (pcg(rng) % num_light_sources);
	if (sc.n_light_tris > 0u && picked < (int) sc.n_light_tris) {
		uint32_t li = sc.light_tris[picked];
		if (li >= sc.n_tris) li = 0;
		const float4 *tp = sc.tris + 4u * (size_t) li;
		const float4 d1 = tp[0], d2 = tp[1], d3 = tp[2];
		const uint4 d4 = *(const uint4 *) (tp + 3);
		L.area = area_triangle(sc, li);
		const MatRec lm = load_mat(sc, d4.w);
		L.le = V(lm.d3.x, lm.d3.y, lm.d3.z);
		float r1 = rnd(rng), r2 = rnd(rng);
		if (r1 + r2 > 1.0f) { r1 = 1.0f - r1; r2 = 1.0f - r2; } // fc:707-723
		const v3 v0 = V(d1.x, d1.y, d1.z);
		const v3 pnt = r1 * (V(d2.x, d2.y, d2.z) - v0) + r2 * (V(d3.x, d3.y, d3.z) - v0);
		L.y = pnt + v0;
	} else if (sc.n_light_spheres > 0u) {
		uint32_t si = sc.light_spheres[picked - (int) sc.n_light_tris];
		if (si >= sc.n_spheres) si = 0;
		const float4 s = sc.spheres[2u * si];
		const uint4 sm = *(const uint4 *) (sc.spheres + 2u * si + 1);
		L.area = area_sphere(s.w);
		const MatRec lm = load_mat(sc, sm.x);
		L.le = V(lm.d3.x, lm.d3.y, lm.d3.z);
		const float r1 = rnd(rng), r2 = rnd(rng);
		L.y = map_to_unit_sphere(r1, r2) * s.w + V(s.x, s.y, s.z);
	}
	return L;
}

// fc:563-589.  using_NEE (the NEE estimator) turns GGX importance sampling off: metals are cosine-sampled too.
PT_DEV v3 pick_wi(v3 wo, const MatRec &mat, bool using_nee, v3 *wm, float *cos_theta, float *pdf, uint32_t &rng)
{
	const float ty = mat.d2.w;
	const bool is_ggx = !using_nee;
	if (ty == -1.0f || ty == 0.0f || ty == 1.0f || (ty == 2.0f && !is_ggx)) {
		*wm = V(0.0f, 1.0f, 0.0f);
		const float r1 = rnd(rng), r2 = rnd(rng);
		const v3 wi = normalize(map_to_unit_sphere(r1, r2) + *wm);
		*cos_theta = dot(*wm, wi);
		*pdf = *cos_theta / PT_PI;
		return wi;
	}
	if (ty == 2.0f) {
		*cos_theta = 1.0f;
		*pdf = 1.0f;
		const float alpha_ggx = mat.d1.w;
		const float a2 = alpha_ggx * alpha_ggx;
		const float e1 = rnd(rng), e2 = rnd(rng);
		const float theta = ptm_acos(__builtin_sqrtf((1.0f - e1) / ((a2 - 1.0f) * e1 + 1.0f)));
		const float phi = PT_TWO_PI * e2;
		float s_t, c_t, s_p, c_p;
		ptm_sincos(theta, &s_t, &c_t);
		ptm_sincos(phi, &s_p, &c_p);
		*wm = V(s_t * c_p, c_t, s_t * s_p);
		const v3 I = -wo;
		return normalize(I - (2.0f * dot(*wm, I)) * *wm);
	}
	*wm = V(0.0f, 0.0f, 0.0f); *cos_theta = 0.0f; *pdf = 0.0f;
	return V(__builtin_inff(), __builtin_inff(), __builtin_inff());
}

// Compaction into the next queue: ballot + mbcnt inside a wave, the four wave counts through LDS, ONE returning
// atomic per block on the fill counter of the block's output region.  Every thread of the block must call it.
// Returns the slot for a live lane.  n_rays_bounce / shadow: lanes that enqueue a ray of that kind (counted here).
PT_DEV uint32_t claim_slot(const ShadeArgs &a, bool alive, bool bounce_ray, bool shadow_ray)
{
	__shared__ uint32_t s_live[PT_SHADE_BLOCK / 64], s_b[PT_SHADE_BLOCK / 64], s_s[PT_SHADE_BLOCK / 64], s_base;
	const unsigned long long live = __ballot(alive);
	const unsigned long long bm = __ballot(bounce_ray), sm = __ballot(shadow_ray);
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	if (lane == 0) { s_live[wv] = (uint32_t) __popcll(live); s_b[wv] = (uint32_t) __popcll(bm); s_s[wv] = (uint32_t) __popcll(sm); }
	__syncthreads();
	const uint32_t out_region = blockIdx.x % PT_SHARDS;
	if (threadIdx.x == 0) {
		uint32_t tot = 0, nb = 0, ns = 0;
		for (uint32_t w = 0; w < PT_SHADE_BLOCK / 64; w++) { tot += s_live[w]; nb += s_b[w]; ns += s_s[w]; }
		uint32_t base = 0;
		if (tot) base = atomicAdd(&a.ctl->qcount[a.in_q ^ 1u][out_region * PT_SHARD_STRIDE], tot);
		// rays are counted when they are enqueued
		if (nb) atomicAdd(&a.ctl->rays_bounce[out_region * (PT_SHARD_STRIDE / 2)], (unsigned long long) nb);
		if (ns) atomicAdd(&a.ctl->rays_shadow[out_region * (PT_SHARD_STRIDE / 2)], (unsigned long long) ns);
		s_base = base;
	}
	__syncthreads();
	uint32_t o = out_region * a.cap + s_base;
	for (uint32_t w = 0; w < wv; w++) o += s_live[w];
	o += __builtin_amdgcn_mbcnt_hi((uint32_t) (live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) live, 0u));
	return o;
}

// main() + render_function(), fc:1088-1122: seed, pixel jitter, pinhole ray -- for every pixel this rank owns, for
// every frame of the batch.  Path p (frame slot p / n_paths_frame) starts in region p / per_region of queue 0.
__global__ __launch_bounds__(PT_SHADE_BLOCK) void k_raygen(ShadeArgs a)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t n_total = a.n_batch * a.n_paths_frame;
	const bool no_rays = a.estimator != PT_EST_MIS && a.bounces == 0u; // the alternative estimators' loops never run
	if (p == 0) {
		atomicAdd(&a.ctl->paths, (unsigned long long) n_total);
		if (!no_rays) atomicAdd(&a.ctl->rays_primary, (unsigned long long) n_total);
	}
	if (p < PT_SHARDS) {
		const uint32_t lo = p * a.per_region;
		a.ctl->qcount[0][p * PT_SHARD_STRIDE] = (n_total > lo && !no_rays) ? (n_total - lo < a.per_region ? n_total - lo : a.per_region) : 0u;
		a.ctl->shard[p * PT_SHARD_STRIDE] = 0;
	}
	if (p >= n_total) return;
	if (no_rays) { a.sample[p] = make_float4(0.0f, 0.0f, 0.0f, 1.0f); return; }
	v3 cam_origin, rd;
	uint32_t rng;
	primary_ray(a, p, &cam_origin, &rd, &rng);

	const uint32_t st = a.stride;
	const uint32_t region = p / a.per_region;
	const uint32_t o = region * a.cap + (p - region * a.per_region);
	a.out.pix[o] = p;
	a.out.rng[o] = rng;
	a.out.meta[o] = 0;
	st3(a.out.thr, o, st, V(1.0f, 1.0f, 1.0f));
	st3(a.out.col, o, st, V(0.0f, 0.0f, 0.0f));
	st3(a.out.ray, o, st, cam_origin);
	st3(a.out.ray + 3 * (size_t) st, o, st, rd);
}

// A finished path parks its sample; k_resolve blends the batch into the image in frame order.  (Several frames of
// one pixel are in flight at once, and the running mean of fc:1124-1129 is order-dependent in floating point.)
PT_DEV void accumulate(const ShadeArgs &a, uint32_t pid, v3 color)
{
	a.sample[pid] = make_float4(color.x, color.y, color.z, 1.0f);
}

// fc:1124-1129, once per frame of the batch, in frame order: new = n/(n+1) * old + 1/(n+1) * sample, alpha = 1
__global__ __launch_bounds__(PT_SHADE_BLOCK) void k_resolve(ShadeArgs a)
{
	const uint32_t pf = blockIdx.x * blockDim.x + threadIdx.x;
	if (pf >= a.n_paths_frame) return;
	float4 *px = (float4 *) a.fb + path_pixel(pf, a.width, a.partition, a.rank, a.world);
	float4 acc = *px;
	for (uint32_t s = 0; s < a.n_batch; s++) {
		const float4 c = a.sample[(size_t) s * a.n_paths_frame + pf];
		const uint32_t frame = a.first_frame + s;
		const float fa = (float) frame / (float) (frame + 1u), fb = 1.0f / (float) (frame + 1u);
		acc.x = fa * acc.x + fb * c.x;
		acc.y = fa * acc.y + fb * c.y;
		acc.z = fa * acc.z + fb * c.z;
		acc.w = 1.0f;
	}
	*px = acc;
}

// One shading stage.  stage 0 consumes the primary hit (fc:880-896); stage k >= 1 consumes the shadow and
// bounce hits of bounce k (fc:970-1077) and, unless k == bounces, emits bounce k+1 (fc:901-968, 1009-1017).
// PRIMARY = true is stage 0 of the fused primary stage: slot i IS path i of the batch (dense, no regions), its hit was
// written by k_trace<.., GEN>, and ray + RNG state are generated again by primary_ray instead of being read -- no
// k_raygen launch, and neither kernel touches the 52 bytes per path that k_raygen would have written.
template <bool PRIMARY>
__global__ __launch_bounds__(PT_SHADE_BLOCK, PT_SHADE_MIN_BLOCKS) void k_shade(ShadeArgs a)
{
	// launch_bpr blocks are launched per region: the front of each region up to the largest fill this batch can produce
	// (region_block_bound in pt355.hip), not every slot of a queue sized for the largest batch
	const uint32_t in_region = PRIMARY ? 0u : blockIdx.x / a.launch_bpr;
	const uint32_t block_off = (blockIdx.x - in_region * a.launch_bpr) * PT_SHADE_BLOCK; // first slot of this block inside its region
	const uint32_t i = PRIMARY ? blockIdx.x * blockDim.x + threadIdx.x : in_region * a.cap + block_off + threadIdx.x;
	const uint32_t n_in = PRIMARY ? a.n_batch * a.n_paths_frame : a.ctl->qcount[a.in_q][in_region * PT_SHARD_STRIDE];
	if (i < PT_SHARDS) a.ctl->shard[i * PT_SHARD_STRIDE] = 0; // for the trace launch that follows
	if (PRIMARY && i == 0) { // what k_raygen counts
		atomicAdd(&a.ctl->paths, (unsigned long long) n_in);
		atomicAdd(&a.ctl->rays_primary, (unsigned long long) n_in);
	}
	const uint32_t st = a.stride;
	const DevScene &sc = a.sc;
	// a block wholly past its region's fill has nothing to read and nothing to claim (the fill counts live on the device, so
	// the grid covers the bound, not the fill): leave before the block-wide compaction
	if (block_off >= n_in) return;
	bool alive = block_off + threadIdx.x < n_in;

	uint32_t pixel = 0 /* path id in the batch */, rng = 0, meta = 0;
	v3 thr = V(0, 0, 0), col = V(0, 0, 0), ro = V(0, 0, 0), rd = V(0, 0, 0);
	HitRec data;
	data.t = 0; data.n = V(0, 0, 0); data.mat = 0; data.obj16 = 0; data.type = 0; data.uvx = 0; data.uvy = 0;
	v3 y = V(0, 0, 0);
	MatRec mat_y;
	mat_y.d1 = mat_y.d2 = mat_y.d3 = make_float4(0, 0, 0, 0);
	const uint32_t num_light_sources = sc.n_light_tris + sc.n_light_spheres;

	if (alive) {
		if (PRIMARY) {
			pixel = i;
			primary_ray(a, i, &ro, &rd, &rng);
			thr = V(1.0f, 1.0f, 1.0f);
		} else {
			// the stages of this kernel keep path id and flags in ONE word, (id << 4) | flags, in the meta column (bit 0 is what
			// the trace kernel reads); only k_raygen's output (stage 0 of the unfused form) has them apart
			rng = PT_QLD(&a.in.rng[i]); meta = PT_QLD(&a.in.meta[i]);
			if (a.stage == 0u) pixel = PT_QLD(&a.in.pix[i]);
			else { pixel = meta >> 4; meta &= 15u; }
			// stage 0 never touches the throughput (fc:874-897: it is first multiplied at fc:1069), so the stage after it knows
			// it is 1 and stage 0 does not write it: 12 B per path less on either side
			thr = a.stage == 1u ? V(1.0f, 1.0f, 1.0f) : ld3(a.in.thr, i, st);
			col = ld3(a.in.col, i, st);
			ro = ld3(a.in.ray, i, st); rd = ld3(a.in.ray + 3 * (size_t) st, i, st);
		}
		const uint4 hrec = ldq4(&a.in.hit[i]);
		const uint32_t hp = hrec.x, htk = hrec.y;
		const float hu = __uint_as_float(hrec.z), hv = __uint_as_float(hrec.w);
		const bool hit_anything = (htk >> 16) != 0u;

		if (PRIMARY || a.stage == 0) {
			if (!hit_anything) {
				col = sky_color(sc, rd) * 1.0f; // fc:884
				accumulate(a, pixel, col);
				alive = false;
			} else {
				data = hit_attributes(sc, hp, htk, hu, hv, ro, rd);
				y = (ro + rd * data.t) + PT_NORMAL_OFFSET * data.n; // fc:887
				mat_y = load_mat(sc, data.mat);
				col = col + V(mat_y.d3.x, mat_y.d3.y, mat_y.d3.z);   // fc:894
			}
		} else {
			const bool can_use_nee = (meta & 2u) != 0u;
			// ---- NEE of this bounce, resolved now that the shadow ray is back (fc:970-1006)
			if (meta & 1u) {
				v3 direct = V(0.0f, 0.0f, 0.0f);
				const uint4 srec = ldq4(&a.in.shit[i]);
				const uint32_t sp_ = srec.x, stk = srec.y;
				if ((stk >> 16) != 0u && (stk >> 16) != 3u) { // kind 3: an any-hit traversal already found an occluder
					const v3 sro = ld3(a.in.sray, i, st);
					const v3 y_nee = ld3(a.in.nee + 5 * (size_t) st, i, st);
					const v3 srd = normalize(y_nee - sro); // fc:959-960 again, bit for bit: the direction is not kept in the queue
					const float su = __uint_as_float(srec.z), sv = __uint_as_float(srec.w);
					const HitRec sh = hit_attributes(sc, sp_, stk, su, sv, sro, srd);
					const v3 dlt = y_nee - (sro + srd * sh.t);
					if (dlt.x < PT_FLOAT_COMPARE && dlt.y < PT_FLOAT_COMPARE && dlt.z < PT_FLOAT_COMPARE) {
						const float cos_theta_y = dot(sh.n, -srd);
						if (cos_theta_y > 0.0f) {
							const v3 P = ld3(a.in.nee, i, st);
							const float A = PT_QLD(&a.in.nee[3 * (size_t) st + i]), pdf_bsdf = PT_QLD(&a.in.nee[4 * (size_t) st + i]);
							const float pdf_nee_sa = A / cos_theta_y;
							const float w_nee = pdf_nee_sa / (pdf_nee_sa + pdf_bsdf);
							direct = direct + P * w_nee / pdf_nee_sa;
						}
					}
				}
				direct = direct / 1.0f;
				col = col + thr * direct;
			}
			// ---- the bounce ray (fc:1023-1077)
			const v3 brdf = ld3(a.in.bs, i, st);
			const float cos_theta_x = PT_QLD(&a.in.bs[3 * (size_t) st + i]), pdf_bsdf_sa = PT_QLD(&a.in.bs[4 * (size_t) st + i]);
			if (!hit_anything) {
				col = col + 1.0f * thr * brdf * cos_theta_x * sky_color(sc, rd) / pdf_bsdf_sa; // fc:1027
				accumulate(a, pixel, col);
				alive = false;
			} else {
				data = hit_attributes(sc, hp, htk, hu, hv, ro, rd);
				const float cos_theta_y = dot(data.n, -rd);
				y = (ro + rd * data.t) + PT_NORMAL_OFFSET * data.n;
				mat_y = load_mat(sc, data.mat);
				const v3 le_y = V(mat_y.d3.x, mat_y.d3.y, mat_y.d3.z);
				float w_bsdf = 1.0f;
				if (can_use_nee && cos_theta_y > 0.0f) {
					if (mat_y.d2.w == -1.0f) {
						float pdf_nee_area = 0.0f;
						if (data.type == 0u) pdf_nee_area = 1.0f / area_triangle(sc, data.obj16);
						else if (data.type == 1u) pdf_nee_area = 1.0f / area_sphere(sc.spheres[2u * (data.obj16 < sc.n_spheres ? data.obj16 : 0u)].w);
						pdf_nee_area /= (float) num_light_sources;
						const float pdf_nee_sa = pdf_nee_area * data.t * data.t / cos_theta_y;
						w_bsdf = pdf_bsdf_sa / (pdf_bsdf_sa + pdf_nee_sa);
					}
					col = col + thr * brdf * le_y * cos_theta_y * w_bsdf / pdf_bsdf_sa; // fc:1066
				}
				thr = thr * (brdf * cos_theta_x / pdf_bsdf_sa); // fc:1069
				if (!can_use_nee && (meta & 4u)) col = col + thr * le_y; // fc:1071-1077
				const bool dead = thr.x == 0.0f && thr.y == 0.0f && thr.z == 0.0f && !(a.flags & PT_FLAG_TRACE_DEAD_PATHS);
				if (a.stage == a.bounces || dead) {
					accumulate(a, pixel, col);
					alive = false;
				}
			}
		}
	}
	if ((PRIMARY || a.stage == 0) && a.bounces == 0 && alive) { // frame_data.z == 0: the loop at fc:899 never runs
		accumulate(a, pixel, col);
		alive = false;
	}

	// ---- claim a slot of the next queue
	const uint32_t num_ls = num_light_sources;
	bool can_use_nee = false;
	if (alive) { // fc:914-915 for the vertex the next bounce leaves from (x = y, mat_x = mat_y)
		const float a2 = mat_y.d1.w * mat_y.d1.w;
		can_use_nee = num_ls > 0u && (mat_y.d2.w == 0.0f || mat_y.d2.w == 1.0f || (mat_y.d2.w == 2.0f && a2 > 0.0f));
	}
	const uint32_t o = claim_slot(a, alive, alive, can_use_nee);
	if (!alive) return;

	// ---- emit bounce b = stage + 1 from x = y (fc:901-968, 1009-1017)
	const v3 x = y;
	const v3 normal_x = data.n;
	const MatRec mat_x = mat_y;
	const m3 tnb = tnb_rows(normal_x);
	const m3 inverse_tnb = inverse(tnb);
	const v3 wo = normalize(mul(tnb, -rd));
	uint32_t new_meta = (can_use_nee ? 3u : 0u) | (mat_x.d2.w == 2.0f ? 4u : 0u);
	if (can_use_nee) {
		const float pdf_pick_light = 1.0f / (float) num_light_sources;
		const LightPick L = sample_light(sc, num_light_sources, rng);
		const float light_area = L.area;
		const v3 y_nee = L.y, le = L.le;
		const v3 dist_vec = y_nee - x;
		const v3 shadow_rd = normalize(dist_vec);
		const float squared_dist = dot(dist_vec, dist_vec);
		const float cos_theta_x = fmax_(0.0f, dot(normal_x, shadow_rd));
		const v3 wi = normalize(mul(tnb, shadow_rd));
		const v3 wm = normalize(mul(tnb, data.n));
		const v3 brdf = calc_brdf(sc, wo, wm, wi, mat_x, data.uvx, data.uvy, true);
		const v3 P = le * brdf * cos_theta_x;
		const float pdf_light_area = pdf_pick_light * (1.0f / light_area);
		st3(a.out.sray, o, st, x); // the direction is rebuilt from x and y_nee where it is needed (k_trace refill, next stage)
		st3(a.out.nee, o, st, P);
		PT_QST(pdf_light_area * squared_dist, &a.out.nee[3 * (size_t) st + o]);
		PT_QST(cos_theta_x / PT_PI, &a.out.nee[4 * (size_t) st + o]);
		st3(a.out.nee + 5 * (size_t) st, o, st, y_nee);
	}
	const v3 new_ro = x + PT_NORMAL_OFFSET * normal_x; // fc:1009

	v3 wm;
	float cos_theta_x, pdf_bsdf_sa;
	const v3 wi = pick_wi(wo, mat_x, false, &wm, &cos_theta_x, &pdf_bsdf_sa, rng);
	const v3 new_rd = normalize(mul(inverse_tnb, wi));
	const v3 brdf = calc_brdf(sc, wo, wm, wi, mat_x, data.uvx, data.uvy, false);

	PT_QST(rng, &a.out.rng[o]);
	PT_QST((pixel << 4) | new_meta, &a.out.meta[o]); // path ids stay below 2^28: pt_create takes at most 16384 x 16384 pixels, and caps a launch set of several frames at 2^26 paths
	if (!(PRIMARY || a.stage == 0u)) st3(a.out.thr, o, st, thr); // stage 0: still 1, see above
	st3(a.out.col, o, st, col);
	st3(a.out.ray, o, st, new_ro);
	st3(a.out.ray + 3 * (size_t) st, o, st, new_rd);
	st3(a.out.bs, o, st, brdf);
	PT_QST(cos_theta_x, &a.out.bs[3 * (size_t) st + o]);
	PT_QST(pdf_bsdf_sa, &a.out.bs[4 * (size_t) st + o]);
}

// The two estimators the reference keeps commented out at fc:1109-1110, as wavefront stages.  Stage b consumes the
// hit of the loop's b-th intersect() (fc:659 / fc:735).  NEE estimator: the shadow ray emitted at stage b is resolved
// at the head of stage b+1 (nothing else touches `color` in between, so the order of additions is the shader's), and
// a last stage b == bounces only resolves.  Queue columns are reused: nee = Le*BRDF (3), cos_x, d^2, y (3);
// bs = throughput before this bounce (3), pdf_light_area.  meta bit0: shadow ray present, bit3: prev_bounce_specular.
// 6 blocks per CU = 6 waves/SIMD: 80 VGPRs + 36 B of scratch.  Left alone the compiler takes 103 VGPRs (4 waves/SIMD); measured on
// the bench scene, 32 frames, k_shade_alt ms (NEE / BRDF estimator): 4 waves 12.64 / 6.80, 5 waves 11.99 / 6.51, 6 waves 12.13 / 6.38.
#ifndef PT_SHADE_ALT_MIN_BLOCKS
#define PT_SHADE_ALT_MIN_BLOCKS 6
#endif
__global__ __launch_bounds__(PT_SHADE_BLOCK, PT_SHADE_ALT_MIN_BLOCKS) void k_shade_alt(ShadeArgs a)
{
	const uint32_t in_region = blockIdx.x / a.launch_bpr; // as in k_shade
	const uint32_t block_off = (blockIdx.x - in_region * a.launch_bpr) * PT_SHADE_BLOCK;
	const uint32_t i = in_region * a.cap + block_off + threadIdx.x;
	const uint32_t n_in = a.ctl->qcount[a.in_q][in_region * PT_SHARD_STRIDE];
	if (i < PT_SHARDS) a.ctl->shard[i * PT_SHARD_STRIDE] = 0; // for the trace launch that follows
	const uint32_t st = a.stride;
	const DevScene &sc = a.sc;
	const bool nee_est = a.estimator == PT_EST_NEE;
	if (block_off >= n_in) return; // wholly past the region's fill: leave before the block-wide compaction
	bool alive = block_off + threadIdx.x < n_in;
	bool emit_bounce = false, emit_shadow = false;

	uint32_t pid = 0, rng = 0, meta = 0, new_meta = 0;
	v3 thr = V(0, 0, 0), col = V(0, 0, 0), ro = V(0, 0, 0), rd = V(0, 0, 0);
	v3 new_ro = V(0, 0, 0), new_rd = V(0, 0, 0), s_ro = V(0, 0, 0), s_rd = V(0, 0, 0), s_lb = V(0, 0, 0), s_y = V(0, 0, 0), thr_old = V(0, 0, 0);
	float s_cos = 0.0f, s_d2 = 0.0f, s_pdf = 0.0f;

	if (alive) {
		pid = a.in.pix[i]; rng = a.in.rng[i]; meta = a.in.meta[i];
		thr = ld3(a.in.thr, i, st); col = ld3(a.in.col, i, st);
		ro = ld3(a.in.ray, i, st); rd = ld3(a.in.ray + 3 * (size_t) st, i, st);
		// ---- NEE of the previous stage (fc:815-842)
		if (nee_est && (meta & 1u)) {
			v3 direct = V(0.0f, 0.0f, 0.0f);
			const uint4 srec = a.in.shit[i];
			if ((srec.y >> 16) != 0u) {
				const v3 sro = ld3(a.in.sray, i, st), srd = ld3(a.in.sray + 3 * (size_t) st, i, st);
				const HitRec sh = hit_attributes(sc, srec.x, srec.y, __uint_as_float(srec.z), __uint_as_float(srec.w), sro, srd);
				const v3 y_l = ld3(a.in.nee + 5 * (size_t) st, i, st);
				const v3 dl = y_l - (sro + srd * sh.t);
				if (__builtin_fabsf(dl.x) <= PT_FLOAT_COMPARE && __builtin_fabsf(dl.y) <= PT_FLOAT_COMPARE && __builtin_fabsf(dl.z) <= PT_FLOAT_COMPARE) {
					const v3 lb = ld3(a.in.nee, i, st);
					const float cos_x = a.in.nee[3 * (size_t) st + i], d2 = a.in.nee[4 * (size_t) st + i];
					const float pdf_light_area = a.in.bs[3 * (size_t) st + i];
					const float cos_y = fmax_(0.0f, dot(sh.n, -srd));
					const float G = cos_x * cos_y / d2;
					direct = direct + lb * G / pdf_light_area;
				}
			}
			direct = direct / 1.0f;
			col = col + ld3(a.in.bs, i, st) * direct;
		}
		if (nee_est && a.stage == a.bounces) {
			accumulate(a, pid, col);
			alive = false;
		} else {
			const uint4 hrec = a.in.hit[i];
			if ((hrec.y >> 16) == 0u) {
				col = col + thr * sky_color(sc, rd) * 1.0f; // fc:662 / fc:738
				accumulate(a, pid, col);
				alive = false;
			} else {
				const HitRec data = hit_attributes(sc, hrec.x, hrec.y, __uint_as_float(hrec.z), __uint_as_float(hrec.w), ro, rd);
				const MatRec mat = load_mat(sc, data.mat);
				const v3 le = V(mat.d3.x, mat.d3.y, mat.d3.z);
				const m3 tnb = tnb_rows(data.n);
				const m3 inverse_tnb = inverse(tnb);
				const v3 wo = normalize(mul(tnb, -rd));
				const bool last = a.stage + 1u == a.bounces;
				if (!nee_est) {
					// fc:666-689
					col = col + thr * le;
					new_ro = (ro + (rd * data.t)) + (PT_NORMAL_OFFSET * data.n);
					v3 wm; float cos_theta, pdf;
					const v3 wi = pick_wi(wo, mat, false, &wm, &cos_theta, &pdf, rng);
					new_rd = normalize(mul(inverse_tnb, wi));
					const v3 brdf = calc_brdf(sc, wo, wm, wi, mat, data.uvx, data.uvy, false);
					thr = thr * ((cos_theta * brdf) / pdf);
					const bool dead = thr.x == 0.0f && thr.y == 0.0f && thr.z == 0.0f && !(a.flags & PT_FLAG_TRACE_DEAD_PATHS);
					if (last || dead) { accumulate(a, pid, col); alive = false; }
					else emit_bounce = true;
				} else {
					const uint32_t num_ls = sc.n_light_tris + sc.n_light_spheres;
					const float a2 = mat.d1.w * mat.d1.w;
					const bool can_use_nee = num_ls > 0u && (mat.d2.w == 0.0f || mat.d2.w == 1.0f || (mat.d2.w == 2.0f && a2 > 0.0f));
					if (mat.d2.w == -1.0f) {
						if (a.stage == 0u || (meta & 8u)) col = col + thr * le; // fc:758-759
						accumulate(a, pid, col);
						alive = false;
					} else {
						if (can_use_nee) {
							// fc:765-842 up to the shadow ray; the visibility-dependent part runs next stage
							const float pdf_pick_light = 1.0f / (float) num_ls;
							const LightPick L = sample_light(sc, num_ls, rng);
							s_pdf = pdf_pick_light * (1.0f / L.area);
							s_ro = (ro + rd * data.t) + PT_NORMAL_OFFSET * data.n;
							const v3 dist_vec = L.y - s_ro;
							s_rd = normalize(dist_vec);
							const v3 wi = normalize(mul(tnb, s_rd));
							const v3 wm = normalize(mul(tnb, data.n));
							s_lb = L.le * calc_brdf(sc, wo, wm, wi, mat, data.uvx, data.uvy, true);
							s_cos = fmax_(0.0f, dot(data.n, s_rd));
							s_d2 = dot(dist_vec, dist_vec);
							s_y = L.y;
							thr_old = thr;
							emit_shadow = true;
						}
						new_ro = ro + (rd * data.t + PT_NORMAL_OFFSET * data.n); // fc:848
						v3 wm; float cos_theta_x, pdf;
						const v3 wi = pick_wi(wo, mat, true, &wm, &cos_theta_x, &pdf, rng);
						new_rd = normalize(mul(inverse_tnb, wi));
						const v3 brdf = calc_brdf(sc, wo, wm, wi, mat, data.uvx, data.uvy, true);
						new_meta = (emit_shadow ? 1u : 0u) | ((mat.d2.w == 2.0f && a2 <= 0.0f) ? 8u : 0u);
						thr = thr * (brdf * cos_theta_x / pdf);
						emit_bounce = !last;
						if (last && !emit_shadow) { accumulate(a, pid, col); alive = false; }
					}
				}
			}
		}
	}
	if (!alive) { emit_bounce = false; emit_shadow = false; }
	const uint32_t o = claim_slot(a, alive, emit_bounce, emit_shadow);
	if (!alive) return;
	a.out.pix[o] = pid;
	a.out.rng[o] = rng;
	a.out.meta[o] = new_meta;
	st3(a.out.thr, o, st, thr);
	st3(a.out.col, o, st, col);
	st3(a.out.ray, o, st, new_ro);
	st3(a.out.ray + 3 * (size_t) st, o, st, new_rd);
	if (emit_shadow) {
		st3(a.out.sray, o, st, s_ro);
		st3(a.out.sray + 3 * (size_t) st, o, st, s_rd);
		st3(a.out.nee, o, st, s_lb);
		a.out.nee[3 * (size_t) st + o] = s_cos;
		a.out.nee[4 * (size_t) st + o] = s_d2;
		st3(a.out.nee + 5 * (size_t) st, o, st, s_y);
		st3(a.out.bs, o, st, thr_old);
		a.out.bs[3 * (size_t) st + o] = s_pdf;
	}
}

// multi-GPU helpers: the packed order is the path order of k_raygen
__global__ void k_pack_owned(const float4 *fb, float4 *dst, uint32_t n, uint32_t width, uint32_t partition, uint32_t rank, uint32_t world)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p < n) dst[p] = fb[path_pixel(p, width, partition, rank, world)];
}
__global__ void k_unpack_from(float4 *fb, const float4 *src, uint32_t n, uint32_t width, uint32_t partition, uint32_t rank, uint32_t world)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p < n) fb[path_pixel(p, width, partition, rank, world)] = src[p];
}

// numerics self-test: one primitive of the contract per launch (op codes: oracle/oracle.h)
__global__ void k_debug_eval(int op, const float *a, const float *b, uint32_t n, float *out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float x = a[i], y = b ? b[i] : 0.0f;
	float r = 0.0f;
	switch (op) {
	case 0: r = fmin_(x, y); break;
	case 1: r = fmax_(x, y); break;
	case 2: r = x / y; break;
	case 3: r = __builtin_sqrtf(x); break;
	case 4: r = q16(x); break;
	case 5: r = ptm_sin(x); break;
	case 6: r = ptm_cos(x); break;
	case 7: r = ptm_tan(x); break;
	case 8: r = ptm_acos(x); break;
	case 9: r = ptm_atan2(x, y); break;
	case 10: r = (float) ((h16) x + (h16) y); break; // native fp16 add
	case 11: r = (float) ((h16) x * (h16) y); break; // native fp16 mul
	case 12: { uint32_t s = __float_as_uint(x); r = rnd(s); break; }
	case 13: case 14: case 15: {
		uint32_t s = __float_as_uint(x);
		const v3 up = V(0.0f, 1.0f, 0.0f);
		float r1 = rnd(s), r2 = rnd(s);
		const v3 wo = normalize(map_to_unit_sphere(r1, r2) + up);
		const v3 F0 = V(0.944f, 0.776f, 0.373f);
		if (op == 13) {
			r1 = rnd(s); r2 = rnd(s);
			const v3 wi = normalize(map_to_unit_sphere(r1, r2) + up);
			r = oren_nayar_brdf(V(1.0f, 0.5f, 0.25f), y, wi, wo).y;
		} else if (op == 14) {
			r1 = rnd(s); r2 = rnd(s);
			const v3 wi = normalize(map_to_unit_sphere(r1, r2) + up);
			r = ggx_brdf_non_is(wo, wi, F0, y).x;
		} else {
			const float a2 = y * y;
			const float e1 = rnd(s), e2 = rnd(s);
			const float theta = ptm_acos(__builtin_sqrtf((1.0f - e1) / ((a2 - 1.0f) * e1 + 1.0f)));
			const float phi = PT_TWO_PI * e2;
			float s_t, c_t, s_p, c_p;
			ptm_sincos(theta, &s_t, &c_t);
			ptm_sincos(phi, &s_p, &c_p);
			const v3 wm = V(s_t * c_p, c_t, s_t * s_p);
			const v3 I = -wo;
			const v3 wi = normalize(I - (2.0f * dot(wm, I)) * wm);
			r = ggx_brdf_is(wo, wm, wi, F0, y).x + wi.y;
		}
		break;
	}
	case 100: r = rcp_of_q16(x); break; // k_trace's reciprocal of a direction component (device only: no oracle counterpart)
	default: break;
	}
	out[i] = r;
}
