// Scene container + procedural stand-in scenes + small host utilities (include/pt355_host.h).
// Mirrors what src/main.cpp:18-63 assembles on the host before the six SSBO uploads.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>

#include "../../../include/pt355_host.h"
#include "layouts.h"

struct pth_scene {
	// "fat" input triangles (src/scene/triangle.hpp:8-24) until finalize packs them
	std::vector<float> v, n, uv;
	std::vector<uint32_t> mat;
	// the six arrays, binding order 0..5
	std::vector<PtSphere> spheres;
	std::vector<PtTriangle> tris; // BVH leaf order after finalize
	std::vector<uint32_t> light_tris;
	std::vector<PtMaterial> materials;
	std::vector<PtNode> nodes;
	std::vector<uint32_t> light_spheres;
	bool finalized = false;
};

namespace {

struct Lcg { // s = s*1664525 + 1013904223
	uint32_t s;
	uint32_t next() { s = s * 1664525u + 1013904223u; return s; }
	float unit() { return (float) (next() >> 8) * (1.0f / 16777216.0f); }
};

struct Vec { float x, y, z; };
inline Vec vsub(Vec a, Vec b) { return Vec{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec vcross(Vec a, Vec b) { return Vec{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec vnorm(Vec a)
{
	float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
	return l > 0 ? Vec{a.x / l, a.y / l, a.z / l} : Vec{0, 1, 0};
}

void push_tri(pth_scene *s, Vec a, Vec b, Vec c, Vec na, Vec nb, Vec nc, const float uvs[6], uint32_t m)
{
	const float vv[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
	const float nn[9] = {na.x, na.y, na.z, nb.x, nb.y, nb.z, nc.x, nc.y, nc.z};
	pth_scene_add_triangles(s, vv, nn, uvs, &m, 1);
}

// quad a-b-c-d (counter-clockwise seen from the side its normal points to), flat normal
void push_quad(pth_scene *s, Vec a, Vec b, Vec c, Vec d, uint32_t m)
{
	Vec nrm = vnorm(vcross(vsub(b, a), vsub(c, a)));
	const float uv1[6] = {0, 0, 1, 0, 1, 1}, uv2[6] = {0, 0, 1, 1, 0, 1};
	push_tri(s, a, b, c, nrm, nrm, nrm, uv1, m);
	push_tri(s, a, c, d, nrm, nrm, nrm, uv2, m);
}

// axis-aligned box of half extents h, rotated by `angle` about +y, centred at c: 12 triangles
void push_box(pth_scene *s, Vec c, Vec h, float angle, uint32_t m)
{
	const float ca = cosf(angle), sa = sinf(angle);
	Vec p[8];
	for (int i = 0; i < 8; i++) {
		const float x = (i & 1) ? h.x : -h.x, y = (i & 2) ? h.y : -h.y, z = (i & 4) ? h.z : -h.z;
		p[i] = Vec{c.x + ca * x + sa * z, c.y + y, c.z - sa * x + ca * z};
	}
	push_quad(s, p[4], p[5], p[7], p[6], m); // +z
	push_quad(s, p[1], p[0], p[2], p[3], m); // -z
	push_quad(s, p[5], p[1], p[3], p[7], m); // +x
	push_quad(s, p[0], p[4], p[6], p[2], m); // -x
	push_quad(s, p[2], p[6], p[7], p[3], m); // +y
	push_quad(s, p[0], p[1], p[5], p[4], m); // -y
}

int add_mat(pth_scene *s, float dr, float dg, float db, float sr, float sg, float sb, float lr, float lg, float lb,
            float rough, int type)
{
	const float d[3] = {dr, dg, db}, sp[3] = {sr, sg, sb}, le[3] = {lr, lg, lb};
	return pth_scene_add_material(s, d, sp, le, rough, -1, type);
}

struct CornellMats { uint32_t white, red, green, light; };

// the box occupies x,y in [-2,2], z in [-8,-4]; the camera sits at the origin looking down -z
// (src/main.cpp:26-28 places the glb model around (0,0,-6); src/main.cpp:67 is the camera)
CornellMats push_cornell_shell(pth_scene *s)
{
	CornellMats m;
	m.white = (uint32_t) add_mat(s, 0.73f, 0.73f, 0.73f, 0, 0, 0, 0, 0, 0, 0.0f, PTH_MAT_LAMBERTIAN);
	m.red = (uint32_t) add_mat(s, 0.65f, 0.05f, 0.05f, 0, 0, 0, 0, 0, 0, 0.0f, PTH_MAT_LAMBERTIAN);
	m.green = (uint32_t) add_mat(s, 0.12f, 0.45f, 0.15f, 0, 0, 0, 0, 0, 0, 0.0f, PTH_MAT_LAMBERTIAN);
	m.light = (uint32_t) add_mat(s, 0, 0, 0, 0, 0, 0, 17.0f, 12.0f, 4.0f, 0.0f, PTH_MAT_LIGHT);
	const float x0 = -2, x1 = 2, y0 = -2, y1 = 2, z0 = -8, z1 = -4;
	push_quad(s, Vec{x0, y0, z1}, Vec{x1, y0, z1}, Vec{x1, y0, z0}, Vec{x0, y0, z0}, m.white); // floor, +y
	push_quad(s, Vec{x0, y1, z0}, Vec{x1, y1, z0}, Vec{x1, y1, z1}, Vec{x0, y1, z1}, m.white); // ceiling, -y
	push_quad(s, Vec{x0, y0, z0}, Vec{x1, y0, z0}, Vec{x1, y1, z0}, Vec{x0, y1, z0}, m.white); // back, +z
	push_quad(s, Vec{x0, y0, z1}, Vec{x0, y0, z0}, Vec{x0, y1, z0}, Vec{x0, y1, z1}, m.red);   // left, +x
	push_quad(s, Vec{x1, y0, z0}, Vec{x1, y0, z1}, Vec{x1, y1, z1}, Vec{x1, y1, z0}, m.green); // right, -x
	const float ly = 1.98f;
	push_quad(s, Vec{-0.6f, ly, -6.6f}, Vec{0.6f, ly, -6.6f}, Vec{0.6f, ly, -5.4f}, Vec{-0.6f, ly, -5.4f}, m.light);
	return m;
}

void push_gold_spheres(pth_scene *s) // src/main.cpp:44-51
{
	const float rough[4] = {0.0f, 0.1f, 0.15f, 0.2f};
	const float cx[4] = {-1.0f, -0.4f, 0.2f, 0.8f};
	for (int i = 0; i < 4; i++) {
		int m = add_mat(s, 0, 0, 0, 0.944f, 0.776f, 0.373f, 0, 0, 0, rough[i], PTH_MAT_SPECULAR_METAL);
		const float o[3] = {cx[i], 1.0f, -5.0f};
		pth_scene_add_sphere(s, o, 0.3f, (uint32_t) m);
	}
}

// radial displacement of the stand-in mesh: smooth lobes + seeded finer bumps
struct Bumps {
	float ph[6];
	float radius(Vec d) const
	{
		float r = 1.0f;
		r += 0.22f * sinf(2.0f * d.x + ph[0]) * sinf(3.0f * d.y + ph[1]);
		r += 0.15f * sinf(4.0f * d.z + ph[2]) * cosf(2.0f * d.x + ph[3]);
		r += 0.04f * sinf(11.0f * d.x + ph[4]) * sinf(13.0f * d.y + ph[5]) * sinf(9.0f * d.z + ph[0]);
		r += 0.015f * sinf(37.0f * d.y + ph[2]) * sinf(41.0f * d.z + ph[1]);
		return r;
	}
	Vec point(Vec d) const { float r = radius(d); return Vec{d.x * r, d.y * r, d.z * r}; }
	Vec normal(Vec d) const
	{
		Vec t = fabsf(d.y) < 0.9f ? vnorm(vcross(Vec{0, 1, 0}, d)) : vnorm(vcross(Vec{1, 0, 0}, d));
		Vec b = vcross(d, t);
		const float e = 1e-3f;
		Vec pa = point(vnorm(Vec{d.x + e * t.x, d.y + e * t.y, d.z + e * t.z}));
		Vec pb = point(vnorm(Vec{d.x - e * t.x, d.y - e * t.y, d.z - e * t.z}));
		Vec pc = point(vnorm(Vec{d.x + e * b.x, d.y + e * b.y, d.z + e * b.z}));
		Vec pd = point(vnorm(Vec{d.x - e * b.x, d.y - e * b.y, d.z - e * b.z}));
		Vec nn = vnorm(vcross(vsub(pa, pb), vsub(pc, pd)));
		if (nn.x * d.x + nn.y * d.y + nn.z * d.z < 0) nn = Vec{-nn.x, -nn.y, -nn.z};
		return nn;
	}
};

uint64_t fnv1a(uint64_t h, const void *p, size_t n)
{
	const uint8_t *b = (const uint8_t *) p;
	for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ULL; }
	return h;
}

} // namespace

extern "C" {

pth_scene *pth_scene_create(void) { return new pth_scene(); }
void pth_scene_destroy(pth_scene *s) { delete s; }

int pth_scene_add_material(pth_scene *s, const float diffuse[3], const float specular[3], const float le[3],
                           float roughness, int tex_index, int type)
{
	PtMaterial m;
	pth_pack_material(diffuse, specular, le, roughness, tex_index, type, &m);
	s->materials.push_back(m);
	s->finalized = false;
	return (int) s->materials.size() - 1;
}

uint32_t pth_scene_add_packed_materials(pth_scene *s, const void *materials48, uint32_t count)
{
	const uint32_t base = (uint32_t) s->materials.size();
	const PtMaterial *m = (const PtMaterial *) materials48;
	s->materials.insert(s->materials.end(), m, m + count);
	s->finalized = false;
	return base;
}

int pth_scene_add_sphere(pth_scene *s, const float origin[3], float radius, uint32_t mat_index)
{
	PtSphere sp;
	pth_pack_sphere(origin, radius, mat_index, &sp);
	s->spheres.push_back(sp);
	s->finalized = false;
	return (int) s->spheres.size() - 1;
}

int pth_scene_add_triangles(pth_scene *s, const float *v, const float *n, const float *uv, const uint32_t *mat,
                            uint32_t count)
{
	if (!v || !n || !mat) return PT_EINVAL;
	s->v.insert(s->v.end(), v, v + 9 * (size_t) count);
	s->n.insert(s->n.end(), n, n + 9 * (size_t) count);
	if (uv) s->uv.insert(s->uv.end(), uv, uv + 6 * (size_t) count);
	else s->uv.insert(s->uv.end(), 6 * (size_t) count, 0.0f);
	s->mat.insert(s->mat.end(), mat, mat + count);
	s->finalized = false;
	return PT_OK;
}

uint32_t pth_scene_fat_count(const pth_scene *s) { return (uint32_t) s->mat.size(); }
int pth_scene_get_fat(const pth_scene *s, float *v9, float *n9, float *uv6, uint32_t *mat)
{
	const size_t n = s->mat.size();
	if (v9) memcpy(v9, s->v.data(), n * 36);
	if (n9) memcpy(n9, s->n.data(), n * 36);
	if (uv6) memcpy(uv6, s->uv.data(), n * 24);
	if (mat) memcpy(mat, s->mat.data(), n * 4);
	return PT_OK;
}

static uint32_t reference_builder(const void *tris, uint32_t n, void *nodes_out, void *sorted_out, void *)
{
	return pth_build_bvh(tris, n, nodes_out, sorted_out);
}

int pth_scene_finalize(pth_scene *s) { return pth_scene_finalize_with(s, reference_builder, nullptr); }

int pth_scene_finalize_with(pth_scene *s, pth_bvh_builder build, void *user)
{
	if (!s || !build) return PT_EINVAL;
	const size_t n = s->mat.size();
	if (n == 0) return PT_EINVAL;
	for (size_t i = 0; i < n; i++)
		if (s->mat[i] >= s->materials.size()) return PT_EINVAL;
	for (const PtSphere &sp : s->spheres)
		if (sp.mat >= s->materials.size()) return PT_EINVAL;
	std::vector<PtTriangle> unsorted(n);
	for (size_t i = 0; i < n; i++)
		pth_pack_triangle(&s->v[9 * i], &s->uv[6 * i], &s->n[9 * i], s->mat[i], &unsorted[i]);
	// build into temporaries: a failed build must leave an earlier finalization untouched
	std::vector<PtTriangle> sorted(n);
	std::vector<PtNode> nodes(2 * n + 1);
	uint32_t nn = build(unsorted.data(), (uint32_t) n, nodes.data(), sorted.data(), user);
	if (nn == 0) return PT_EINVAL;
	nodes.resize(nn);
	s->tris.swap(sorted);
	s->nodes.swap(nodes);
	s->light_tris.resize(n);
	s->light_tris.resize(pth_find_emissive_tris(s->tris.data(), (uint32_t) n, s->materials.data(),
	                                            (uint32_t) s->materials.size(), s->light_tris.data()));
	s->light_spheres.resize(s->spheres.size());
	s->light_spheres.resize(pth_find_emissive_spheres(s->spheres.data(), (uint32_t) s->spheres.size(),
	                                                  s->materials.data(), (uint32_t) s->materials.size(),
	                                                  s->light_spheres.data()));
	s->finalized = true;
	return PT_OK;
}

int pth_scene_desc(const pth_scene *s, pt_scene_desc *out)
{
	if (!s->finalized) return PT_ESTATE;
	memset(out, 0, sizeof *out);
	out->spheres = s->spheres.empty() ? nullptr : s->spheres.data();
	out->n_spheres = (uint32_t) s->spheres.size();
	out->tris = s->tris.data();
	out->n_tris = (uint32_t) s->tris.size();
	out->light_tris = s->light_tris.empty() ? nullptr : s->light_tris.data();
	out->n_light_tris = (uint32_t) s->light_tris.size();
	out->materials = s->materials.data();
	out->n_materials = (uint32_t) s->materials.size();
	out->nodes = s->nodes.data();
	out->n_nodes = (uint32_t) s->nodes.size();
	out->light_spheres = s->light_spheres.empty() ? nullptr : s->light_spheres.data();
	out->n_light_spheres = (uint32_t) s->light_spheres.size();
	return PT_OK;
}

uint64_t pth_scene_hash(const pth_scene *s)
{
	uint64_t h = 14695981039346656037ULL;
	h = fnv1a(h, s->spheres.data(), s->spheres.size() * sizeof(PtSphere));
	h = fnv1a(h, s->tris.data(), s->tris.size() * sizeof(PtTriangle));
	h = fnv1a(h, s->light_tris.data(), s->light_tris.size() * 4);
	h = fnv1a(h, s->materials.data(), s->materials.size() * sizeof(PtMaterial));
	h = fnv1a(h, s->nodes.data(), s->nodes.size() * sizeof(PtNode));
	h = fnv1a(h, s->light_spheres.data(), s->light_spheres.size() * 4);
	return h;
}

uint32_t pth_find_emissive_tris(const void *tris_in, uint32_t n_tris, const void *mats_in, uint32_t n_mats, uint32_t *out)
{
	const PtTriangle *tris = (const PtTriangle *) tris_in;
	const PtMaterial *mats = (const PtMaterial *) mats_in;
	const float eps = 0.0001f; // EPSILON, src/math/math.hpp
	uint32_t k = 0;
	for (uint32_t i = 0; i < n_tris; i++) {
		if (tris[i].mat >= n_mats) continue;
		const float *le = mats[tris[i].mat].le;
		if (le[0] >= eps || le[1] >= eps || le[2] >= eps) out[k++] = i;
	}
	return k;
}

uint32_t pth_find_emissive_spheres(const void *spheres_in, uint32_t n_spheres, const void *mats_in, uint32_t n_mats,
                                   uint32_t *out)
{
	const PtSphere *sp = (const PtSphere *) spheres_in;
	const PtMaterial *mats = (const PtMaterial *) mats_in;
	const float eps = 0.0001f;
	uint32_t k = 0;
	for (uint32_t i = 0; i < n_spheres; i++) {
		if (sp[i].mat >= n_mats) continue;
		const float *le = mats[sp[i].mat].le;
		if (le[0] > eps && le[1] > eps && le[2] > eps) out[k++] = i;
	}
	return k;
}

void pth_frame_seeds(uint32_t first, uint32_t n, uint32_t *out)
{
	// PCG-XSH-RR 64/32 from the library's static initialiser state (pcg_basic.h PCG32_INITIALIZER)
	uint64_t state = 0x853c49e6748fea9bULL;
	const uint64_t inc = 0xda3e39cb94b95bdbULL;
	for (uint64_t i = 0; i < (uint64_t) first + n; i++) {
		const uint64_t old = state;
		state = old * 6364136223846793005ULL + inc;
		const uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27);
		const uint32_t rot = (uint32_t) (old >> 59);
		const uint32_t r = (xs >> rot) | (xs << ((32 - rot) & 31));
		if (i >= first) out[i - first] = r;
	}
}

int pth_scene_make_cornell(pth_scene *s, uint32_t flags)
{
	CornellMats m = push_cornell_shell(s);
	push_box(s, Vec{0.65f, -1.4f, -5.4f}, Vec{0.6f, 0.6f, 0.6f}, -0.3f, m.white);  // short box
	push_box(s, Vec{-0.7f, -0.8f, -6.6f}, Vec{0.6f, 1.2f, 0.6f}, 0.35f, m.white);  // tall box
	if (flags & 1u) push_gold_spheres(s);
	return pth_scene_finalize(s);
}

int pth_scene_make_bunny(pth_scene *s, uint32_t freq, uint32_t target_tris, uint32_t seed, int mesh_material,
                         float mesh_roughness, uint32_t flags)
{
	if (freq == 0 || freq > 1024) return PT_EINVAL;
	push_cornell_shell(s);
	int mm;
	if (mesh_material == PTH_MAT_SPECULAR_METAL)
		mm = add_mat(s, 0, 0, 0, 0.944f, 0.776f, 0.373f, 0, 0, 0, mesh_roughness, PTH_MAT_SPECULAR_METAL);
	else
		mm = add_mat(s, 0.62f, 0.55f, 0.48f, 0, 0, 0, 0, 0, 0, mesh_roughness, mesh_material);

	Lcg rng{seed * 2654435761u + 12345u};
	Bumps bumps;
	for (float &p : bumps.ph) p = 6.2831853f * rng.unit();

	// class-I geodesic sphere: every icosahedron face cut into freq^2 triangles
	const float t = 1.618033988749895f;
	const Vec iv[12] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t},
	                    {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
	const int fi[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4},
	                       {11, 10, 2}, {10, 7, 6}, {7, 1, 8}, {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8},
	                       {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
	const Vec centre{0.0f, -0.85f, -6.0f};
	const float scale = 0.95f;
	uint32_t emitted = 0;
	const uint32_t limit = target_tris ? target_tris : 0xffffffffu;
	auto corner = [&](const Vec &a, const Vec &b, const Vec &c, uint32_t i, uint32_t j) {
		const float fa = (float) (freq - i - j) / (float) freq, fb = (float) i / (float) freq, fc = (float) j / (float) freq;
		return vnorm(Vec{a.x * fa + b.x * fb + c.x * fc, a.y * fa + b.y * fb + c.y * fc, a.z * fa + b.z * fb + c.z * fc});
	};
	auto emit = [&](Vec d0, Vec d1, Vec d2) {
		if (emitted >= limit) return;
		Vec d[3] = {d0, d1, d2}, p[3], nn[3];
		float uvs[6];
		for (int k = 0; k < 3; k++) {
			Vec q = bumps.point(d[k]);
			p[k] = Vec{centre.x + scale * q.x, centre.y + scale * q.y, centre.z + scale * q.z};
			nn[k] = bumps.normal(d[k]);
			uvs[2 * k] = atan2f(d[k].x, d[k].z) * (0.5f / 3.14159265f) + 0.5f;
			uvs[2 * k + 1] = acosf(d[k].y > 1 ? 1 : (d[k].y < -1 ? -1 : d[k].y)) * (1.0f / 3.14159265f);
		}
		push_tri(s, p[0], p[1], p[2], nn[0], nn[1], nn[2], uvs, (uint32_t) mm);
		emitted++;
	};
	for (int f = 0; f < 20; f++) {
		const Vec a = vnorm(iv[fi[f][0]]), b = vnorm(iv[fi[f][1]]), c = vnorm(iv[fi[f][2]]);
		for (uint32_t i = 0; i < freq; i++)
			for (uint32_t j = 0; j + i < freq; j++) {
				emit(corner(a, b, c, i, j), corner(a, b, c, i + 1, j), corner(a, b, c, i, j + 1));
				if (i + j + 1 < freq)
					emit(corner(a, b, c, i + 1, j), corner(a, b, c, i + 1, j + 1), corner(a, b, c, i, j + 1));
			}
	}
	if (flags & 1u) push_gold_spheres(s);
	return pth_scene_finalize(s);
}

int pth_scene_make_soup(pth_scene *s, uint32_t n_tris, uint32_t seed, float size)
{
	CornellMats m = push_cornell_shell(s);
	int grey = add_mat(s, 0.5f, 0.5f, 0.5f, 0, 0, 0, 0, 0, 0, 0.3f, PTH_MAT_OREN_NAYAR);
	Lcg rng{seed};
	for (uint32_t i = 0; i < n_tris; i++) {
		Vec c{-1.8f + 3.6f * rng.unit(), -1.8f + 3.6f * rng.unit(), -7.8f + 3.6f * rng.unit()};
		Vec p[3];
		for (int k = 0; k < 3; k++)
			p[k] = Vec{c.x + size * (rng.unit() - 0.5f), c.y + size * (rng.unit() - 0.5f), c.z + size * (rng.unit() - 0.5f)};
		Vec nrm = vnorm(vcross(vsub(p[1], p[0]), vsub(p[2], p[0])));
		const float uvs[6] = {0, 0, 1, 0, 0, 1};
		push_tri(s, p[0], p[1], p[2], nrm, nrm, nrm, uvs, (i % 7 == 0) ? m.red : (uint32_t) grey);
	}
	return pth_scene_finalize(s);
}

// ---- "atrium": the C4-class stand-in of SURVEY.md 8(d) -- a hall with a tiled floor, two rows of tessellated columns
// joined by arches, textured walls, several ceiling light panels of different radiance, a few metal objects: ~24
// materials (Lambert / Oren-Nayar / GGX, six of them textured), ~16,000 * detail^2 triangles (detail 4 = 264 k).
namespace {

int add_tex_mat(pth_scene *s, float dr, float dg, float db, float rough, int tex, int type)
{
	const float d[3] = {dr, dg, db}, z[3] = {0, 0, 0};
	return pth_scene_add_material(s, d, z, z, rough, tex, type);
}

// quad with explicit uvs (u along a->b, v along a->d), flat normal
void push_quad_uv(pth_scene *s, Vec a, Vec b, Vec c, Vec d, float u0, float v0, float u1, float v1, uint32_t m)
{
	Vec nrm = vnorm(vcross(vsub(b, a), vsub(c, a)));
	const float uv1[6] = {u0, v0, u1, v0, u1, v1}, uv2[6] = {u0, v0, u1, v1, u0, v1};
	push_tri(s, a, b, c, nrm, nrm, nrm, uv1, m);
	push_tri(s, a, c, d, nrm, nrm, nrm, uv2, m);
}

// surface of revolution about the vertical through (cx, cz): radius r(t), height y(t), t in [0,1]; smooth normals
void push_revolved(pth_scene *s, float cx, float cz, float y0, float y1, const std::function<float(float)> &radius, uint32_t seg, uint32_t rings, uint32_t m)
{
	const float two_pi = 6.2831853f;
	auto point = [&](uint32_t i, uint32_t j) {
		const float t = (float) j / (float) rings, a = two_pi * (float) (i % seg) / (float) seg;
		const float r = radius(t);
		return Vec{cx + r * cosf(a), y0 + (y1 - y0) * t, cz + r * sinf(a)};
	};
	auto normal = [&](uint32_t i, uint32_t j) {
		const float t = (float) j / (float) rings, a = two_pi * (float) (i % seg) / (float) seg;
		const float dr = (radius(t + 1e-3f) - radius(t - 1e-3f)) / 2e-3f / (y1 - y0); // d radius / d height
		return vnorm(Vec{cosf(a), -dr, sinf(a)});
	};
	for (uint32_t j = 0; j < rings; j++)
		for (uint32_t i = 0; i < seg; i++) {
			const Vec a = point(i, j), b = point(i + 1, j), c = point(i + 1, j + 1), d = point(i, j + 1);
			const float u0 = (float) i / (float) seg, u1 = (float) (i + 1) / (float) seg, v0 = (float) j / (float) rings, v1 = (float) (j + 1) / (float) rings;
			const float uv1[6] = {u0 * 4, v0 * 2, u1 * 4, v1 * 2, u1 * 4, v0 * 2}, uv2[6] = {u0 * 4, v0 * 2, u0 * 4, v1 * 2, u1 * 4, v1 * 2};
			push_tri(s, a, c, b, normal(i, j), normal(i + 1, j + 1), normal(i + 1, j), uv1, m);
			push_tri(s, a, d, c, normal(i, j), normal(i, j + 1), normal(i + 1, j + 1), uv2, m);
		}
}

// half torus (an arch) springing at height y from (xa, za) to (xb, zb)
void push_arch(pth_scene *s, float xa, float za, float xb, float zb, float y, float tube, uint32_t arc, uint32_t ring, uint32_t m)
{
	const float pi = 3.14159265f;
	const float cx = 0.5f * (xa + xb), cz = 0.5f * (za + zb);
	const float hx = 0.5f * (xb - xa), hz = 0.5f * (zb - za), R = sqrtf(hx * hx + hz * hz);
	const Vec along{hx / R, 0.0f, hz / R}, side{-along.z, 0.0f, along.x};
	auto frame = [&](uint32_t i, uint32_t j, Vec *p, Vec *n) {
		const float a = pi * (float) i / (float) arc, b = 2.0f * pi * (float) (j % ring) / (float) ring;
		const float ca = cosf(a), sa = sinf(a);
		const Vec centre{cx - R * ca * along.x, y + R * sa, cz - R * ca * along.z};
		const Vec out{-ca * along.x, sa, -ca * along.z};                 // from the arch's axis outwards
		*n = vnorm(Vec{out.x * cosf(b) + side.x * sinf(b), out.y * cosf(b), out.z * cosf(b) + side.z * sinf(b)});
		*p = Vec{centre.x + tube * n->x, centre.y + tube * n->y, centre.z + tube * n->z};
	};
	for (uint32_t i = 0; i < arc; i++)
		for (uint32_t j = 0; j < ring; j++) {
			Vec p[4], n[4];
			frame(i, j, &p[0], &n[0]); frame(i + 1, j, &p[1], &n[1]); frame(i + 1, j + 1, &p[2], &n[2]); frame(i, j + 1, &p[3], &n[3]);
			const float u0 = (float) i / (float) arc * 3, u1 = (float) (i + 1) / (float) arc * 3, v0 = (float) j / (float) ring, v1 = (float) (j + 1) / (float) ring;
			const float uv1[6] = {u0, v0, u1, v0, u1, v1}, uv2[6] = {u0, v0, u1, v1, u0, v1};
			push_tri(s, p[0], p[1], p[2], n[0], n[1], n[2], uv1, m);
			push_tri(s, p[0], p[2], p[3], n[0], n[2], n[3], uv2, m);
		}
}

} // namespace

int pth_scene_make_atrium(pth_scene *s, uint32_t detail, uint32_t seed)
{
	if (detail == 0 || detail > 16) return PT_EINVAL;
	Lcg rng{seed * 747796405u + 2891336453u};
	const float x0 = -4.0f, x1 = 4.0f, y0 = -2.0f, y1 = 2.5f, z0 = -13.0f, z1 = -1.0f;
	// materials: 0..5 textured (layers 0..5), then plain ones
	const int tile_a = add_tex_mat(s, 0.85f, 0.85f, 0.85f, 0.0f, 0, PTH_MAT_LAMBERTIAN);
	const int tile_b = add_tex_mat(s, 0.80f, 0.78f, 0.75f, 0.25f, 1, PTH_MAT_OREN_NAYAR);
	const int wall_l = add_tex_mat(s, 0.9f, 0.9f, 0.9f, 0.2f, 2, PTH_MAT_OREN_NAYAR);
	const int wall_r = add_tex_mat(s, 0.9f, 0.9f, 0.9f, 0.0f, 3, PTH_MAT_LAMBERTIAN);
	const int wall_b = add_tex_mat(s, 1.0f, 1.0f, 1.0f, 0.3f, 4, PTH_MAT_OREN_NAYAR);
	int brass_tex;
	{
		const float z[3] = {0, 0, 0}, f0[3] = {0.95f, 0.80f, 0.45f};
		brass_tex = pth_scene_add_material(s, z, f0, z, 0.25f, 5, PTH_MAT_SPECULAR_METAL);
	}
	const int ceiling = add_mat(s, 0.75f, 0.75f, 0.78f, 0, 0, 0, 0, 0, 0, 0.0f, PTH_MAT_LAMBERTIAN);
	int column_mat[8], arch_mat[4], light_mat[6];
	for (int i = 0; i < 8; i++)
		column_mat[i] = add_mat(s, 0.45f + 0.4f * rng.unit(), 0.45f + 0.4f * rng.unit(), 0.4f + 0.4f * rng.unit(), 0, 0, 0, 0, 0, 0,
		                        0.05f + 0.3f * rng.unit(), PTH_MAT_OREN_NAYAR);
	for (int i = 0; i < 4; i++)
		arch_mat[i] = add_mat(s, 0.3f + 0.5f * rng.unit(), 0.3f + 0.5f * rng.unit(), 0.3f + 0.5f * rng.unit(), 0, 0, 0, 0, 0, 0, 0.0f, PTH_MAT_LAMBERTIAN);
	const float le[6][3] = {{17, 12, 4}, {6, 8, 14}, {12, 12, 12}, {15, 6, 3}, {4, 10, 6}, {9, 9, 16}};
	for (int i = 0; i < 6; i++) light_mat[i] = add_mat(s, 0, 0, 0, 0, 0, 0, le[i][0], le[i][1], le[i][2], 0.0f, PTH_MAT_LIGHT);
	const int gold = add_mat(s, 0, 0, 0, 0.944f, 0.776f, 0.373f, 0, 0, 0, 0.15f, PTH_MAT_SPECULAR_METAL);
	const int steel = add_mat(s, 0, 0, 0, 0.56f, 0.57f, 0.58f, 0, 0, 0, 0.3f, PTH_MAT_SPECULAR_METAL);
	const int copper = add_mat(s, 0, 0, 0, 0.955f, 0.638f, 0.538f, 0, 0, 0, 0.08f, PTH_MAT_SPECULAR_METAL);

	// floor tiles (checker of the two textured tile materials), ceiling, walls
	const uint32_t tx = 8u * detail, tz = 12u * detail;
	for (uint32_t i = 0; i < tx; i++)
		for (uint32_t k = 0; k < tz; k++) {
			const float ax = x0 + (x1 - x0) * (float) i / (float) tx, bx = x0 + (x1 - x0) * (float) (i + 1) / (float) tx;
			const float az = z1 + (z0 - z1) * (float) k / (float) tz, bz = z1 + (z0 - z1) * (float) (k + 1) / (float) tz;
			push_quad_uv(s, Vec{ax, y0, az}, Vec{bx, y0, az}, Vec{bx, y0, bz}, Vec{ax, y0, bz}, 0, 0, 1, 1, (uint32_t) (((i + k) & 1u) ? tile_a : tile_b));
		}
	push_quad(s, Vec{x0, y1, z0}, Vec{x1, y1, z0}, Vec{x1, y1, z1}, Vec{x0, y1, z1}, (uint32_t) ceiling);
	push_quad_uv(s, Vec{x0, y0, z1}, Vec{x0, y0, z0}, Vec{x0, y1, z0}, Vec{x0, y1, z1}, -0.5f, -0.5f, 5.5f, 2.0f, (uint32_t) wall_l);
	push_quad_uv(s, Vec{x1, y0, z0}, Vec{x1, y0, z1}, Vec{x1, y1, z1}, Vec{x1, y1, z0}, 0.0f, 0.0f, 6.0f, 2.25f, (uint32_t) wall_r);
	push_quad_uv(s, Vec{x0, y0, z0}, Vec{x1, y0, z0}, Vec{x1, y1, z0}, Vec{x0, y1, z0}, -1.0f, 0.0f, 3.0f, 2.25f, (uint32_t) wall_b);
	// six light panels just under the ceiling
	for (int i = 0; i < 6; i++) {
		const float cx = (i & 1) ? 1.6f : -1.6f, cz = -3.0f - 3.8f * (float) (i / 2), h = 0.45f, ly = y1 - 0.02f;
		push_quad(s, Vec{cx - h, ly, cz - h}, Vec{cx + h, ly, cz - h}, Vec{cx + h, ly, cz + h}, Vec{cx - h, ly, cz + h}, (uint32_t) light_mat[i]);
	}
	// two rows of eight columns with entasis, base and capital; an arch from each column to the next of its row
	const uint32_t seg = 16u * detail, rings = 24u * detail, arc = 16u * detail, ring = 8u * detail;
	const float spring = 1.2f, pitch = 1.45f;
	for (int row = 0; row < 2; row++) {
		const float cx = row ? 2.4f : -2.4f;
		for (int k = 0; k < 8; k++) {
			const float cz = -2.2f - pitch * (float) k;
			const float r0 = 0.22f + 0.04f * rng.unit();
			auto shaft = [r0](float t) {
				const float base = t < 0.06f ? 1.45f - 7.5f * t : 1.0f, cap = t > 0.92f ? 1.0f + 6.0f * (t - 0.92f) : 1.0f;
				return r0 * base * cap * (1.0f - 0.12f * t * t) * (1.0f + 0.015f * sinf(60.0f * t));
			};
			push_revolved(s, cx, cz, y0, spring, shaft, seg, rings, (uint32_t) column_mat[(row * 3 + k) & 7]);
			if (k < 7) push_arch(s, cx, cz, cx, cz - pitch, spring, 0.11f, arc, ring, (uint32_t) arch_mat[(row + k) & 3]);
		}
	}
	// objects in the nave: three metal vases (one with the textured brass) and four analytic spheres
	const int vase_mat[3] = {gold, brass_tex, steel};
	for (int i = 0; i < 3; i++) {
		const float cz = -4.0f - 3.2f * (float) i, cx = (i & 1) ? 0.7f : -0.7f;
		auto profile = [](float t) { return 0.12f + 0.30f * sinf(3.0f * t + 0.3f) * sinf(3.0f * t + 0.3f) * (1.0f - 0.55f * t); };
		push_revolved(s, cx, cz, y0, y0 + 1.1f, profile, seg, 16u * detail, (uint32_t) vase_mat[i]);
	}
	for (int i = 0; i < 4; i++) {
		const float o[3] = {-1.2f + 0.8f * (float) i, y0 + 0.28f, -3.2f - 2.3f * (float) i};
		pth_scene_add_sphere(s, o, 0.28f, (uint32_t) ((i & 1) ? copper : gold));
	}
	return pth_scene_finalize(s);
}

// Six seeded 512x512 RGB8 layers for the atrium's textured materials: checker, bricks, stripes, value-noise marble,
// rings, speckle.  out: layers * 512 * 512 * 3 bytes (layers <= 6 are distinct, further ones repeat with another seed).
int pth_make_textures(uint32_t layers, uint32_t seed, uint8_t *out)
{
	if (!out) return PT_EINVAL;
	auto hash = [](uint32_t x, uint32_t y, uint32_t k) {
		uint32_t h = x * 0x9e3779b1u ^ y * 0x85ebca77u ^ k * 0xc2b2ae3du;
		h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
		return h;
	};
	auto noise = [&](float x, float y, uint32_t k) { // bilinear value noise in [0,1]
		const int xi = (int) floorf(x), yi = (int) floorf(y);
		const float fx = x - (float) xi, fy = y - (float) yi;
		auto v = [&](int a, int b) { return (float) (hash((uint32_t) a & 63u, (uint32_t) b & 63u, k) >> 8) * (1.0f / 16777216.0f); };
		const float top = v(xi, yi) * (1 - fx) + v(xi + 1, yi) * fx, bot = v(xi, yi + 1) * (1 - fx) + v(xi + 1, yi + 1) * fx;
		return top * (1 - fy) + bot * fy;
	};
	for (uint32_t l = 0; l < layers; l++) {
		const uint32_t k = seed * 31u + l / 6u * 977u + 1u;
		for (uint32_t y = 0; y < 512; y++)
			for (uint32_t x = 0; x < 512; x++) {
				float r = 0, g = 0, b = 0;
				const float u = (float) x / 512.0f, v = (float) y / 512.0f;
				switch (l % 6u) {
				case 0: { const bool c = ((x >> 6) + (y >> 6)) & 1u; r = c ? 0.92f : 0.25f; g = c ? 0.9f : 0.24f; b = c ? 0.85f : 0.3f; break; }
				case 1: { const uint32_t row = y >> 5, xs = x + ((row & 1u) ? 32u : 0u); const bool mortar = (y & 31u) < 3u || (xs & 63u) < 3u;
				          const float t = 0.75f + 0.25f * (float) (hash(xs >> 6, row, k) >> 24) / 255.0f; r = mortar ? 0.8f : 0.62f * t; g = mortar ? 0.8f : 0.27f * t; b = mortar ? 0.78f : 0.2f * t; break; }
				case 2: { const float t = 0.5f + 0.5f * sinf(40.0f * u + 6.0f * noise(8 * u, 8 * v, k)); r = 0.3f + 0.6f * t; g = 0.35f + 0.4f * t; b = 0.5f; break; }
				case 3: { const float t = noise(6 * u, 6 * v, k) * 0.6f + noise(24 * u, 24 * v, k + 7u) * 0.4f; const float m = 0.5f + 0.5f * sinf(18.0f * (u + v) + 9.0f * t);
				          r = 0.85f - 0.35f * m; g = 0.85f - 0.4f * m; b = 0.9f - 0.3f * m; break; }
				case 4: { const float d = sqrtf((u - 0.5f) * (u - 0.5f) + (v - 0.5f) * (v - 0.5f)); const float t = 0.5f + 0.5f * sinf(70.0f * d); r = 0.2f + 0.7f * t; g = 0.6f; b = 0.9f - 0.6f * t; break; }
				default: { const float t = (float) (hash(x, y, k) >> 24) / 255.0f; r = 0.55f + 0.45f * t; g = 0.5f + 0.4f * t; b = 0.35f + 0.3f * t; break; }
				}
				uint8_t *o = out + (((size_t) l * 512 + y) * 512 + x) * 3;
				o[0] = (uint8_t) (r * 255.0f + 0.5f); o[1] = (uint8_t) (g * 255.0f + 0.5f); o[2] = (uint8_t) (b * 255.0f + 0.5f);
			}
	}
	return PT_OK;
}

int pth_tonemap_rgb8(const float *rgba, uint32_t width, uint32_t height, uint8_t *rgb8)
{
	if (!rgba || !rgb8) return PT_EINVAL;
	for (uint32_t y = 0; y < height; y++)
		for (uint32_t x = 0; x < width; x++) {
			const float *p = rgba + ((size_t) (height - 1 - y) * width + x) * 4; // row 0 is the bottom row
			uint8_t *o = rgb8 + ((size_t) y * width + x) * 3;
			const bool nan = p[0] != p[0] || p[1] != p[1] || p[2] != p[2] || p[3] != p[3];
			const bool inf = std::isinf(p[0]) || std::isinf(p[1]) || std::isinf(p[2]) || std::isinf(p[3]);
			if (nan) { o[0] = 255; o[1] = 0; o[2] = 0; continue; }
			if (inf) { o[0] = 0; o[1] = 255; o[2] = 0; continue; }
			for (int c = 0; c < 3; c++) {
				float g = powf(p[c], 1.0f / 2.2f);
				g = g < 0 ? 0 : (g > 1 ? 1 : g);
				o[c] = (uint8_t) (g * 255.0f + 0.5f);
			}
		}
	return PT_OK;
}

// Image files for a headless build (the reference only ever blits): binary PPM of the tonemapped image, and PFM
// (RGB float, bottom row first -- the framebuffer's own row order) as the accumulator's checkpoint format.
int pth_write_ppm(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height)
{
	if (!path || !rgb8) return PT_EINVAL;
	FILE *f = fopen(path, "wb");
	if (!f) return PT_EINVAL;
	fprintf(f, "P6\n%u %u\n255\n", width, height);
	const size_t n = (size_t) width * height * 3;
	const bool ok = fwrite(rgb8, 1, n, f) == n;
	return (fclose(f) == 0 && ok) ? PT_OK : PT_EINVAL;
}

int pth_write_pfm(const char *path, const float *rgba, uint32_t width, uint32_t height)
{
	if (!path || !rgba) return PT_EINVAL;
	FILE *f = fopen(path, "wb");
	if (!f) return PT_EINVAL;
	fprintf(f, "PF\n%u %u\n-1.0\n", width, height); // negative scale: little-endian
	bool ok = true;
	for (size_t i = 0; i < (size_t) width * height && ok; i++) ok = fwrite(rgba + 4 * i, 4, 3, f) == 3;
	return (fclose(f) == 0 && ok) ? PT_OK : PT_EINVAL;
}

int pth_read_pfm(const char *path, float *rgba, uint32_t width, uint32_t height)
{
	if (!path || !rgba) return PT_EINVAL;
	FILE *f = fopen(path, "rb");
	if (!f) return PT_EINVAL;
	char tag[3] = {0, 0, 0};
	unsigned w = 0, h = 0;
	float scale = 0;
	bool ok = fscanf(f, "%2s %u %u %f", tag, &w, &h, &scale) == 4 && tag[0] == 'P' && tag[1] == 'F' && w == width && h == height &&
	          scale < 0 && fgetc(f) == '\n';
	for (size_t i = 0; i < (size_t) width * height && ok; i++) {
		ok = fread(rgba + 4 * i, 4, 3, f) == 3;
		rgba[4 * i + 3] = 1.0f;
	}
	fclose(f);
	return ok ? PT_OK : PT_EINVAL;
}

} // extern "C"
