// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Thin extern "C" driver around the reference's OWN host translation units
// (src/scene/{bvh,triangle,material,sphere}.cpp, src/math/math.cpp and
// thirdparty/pcg-c-basic-0.9/pcg_basic.c), compiled from where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/libptref.so.
//
// It exists so that tests can pin, bit for bit, the build's own host pieces
// (SSBO packing, Sweep-SAH BVH build, emissive lists, host PCG frame seeds)
// against what the reference itself produces in this container.  Nothing of
// the reference is copied: this file only *calls* it.  /root/reference does not
// exist on the GPU box, so the vectors this library produces are committed as
// fixtures under tests/golden/ (see tests/golden/make_golden.py).
#include <cstdint>
#include <cstring>

#include "scene/bvh.h"        // BVHNodeGLSL, CalculateBVH   (src/scene/bvh.h:7-16)
#include "scene/triangle.hpp" // TriangleGLSL, FindEmissiveTris (src/scene/triangle.hpp:26-45)
#include "scene/material.hpp" // MaterialGLSL                 (src/scene/material.hpp:13-31)
#include "scene/sphere.hpp"   // SphereGLSL, FindEmissiveSpheres (src/scene/sphere.hpp:7-18)
#include "math/math.hpp"      // pixl::octahedral_normal_encoding (src/math/math.hpp)
#include "loader.h"           // LoadGLTF (src/loader.h:4)
#include "scene/model.h"      // Model (src/scene/model.h:7-29)
#include <stb_image.h>        // the reference's vendored decoder (thirdparty/stb), as loader.cpp:209 calls it
#include <stb_image_resize.h> // ... and resizer, as loader.cpp:224 calls it

// the reference's vendored BVH library: an INDEPENDENT closest-hit (own traversal loop, own box and triangle arithmetic)
#include <bvh/bvh.hpp>
#include <bvh/sweep_sah_builder.hpp>
#include <bvh/single_ray_traverser.hpp>
#include <bvh/primitive_intersectors.hpp>
#include <bvh/triangle.hpp>
#include <vector>

extern "C" {

// sizes of the six layouts the boundary freezes
void ref_sizes(uint32_t out[4])
{
	out[0] = (uint32_t) sizeof(TriangleGLSL);
	out[1] = (uint32_t) sizeof(BVHNodeGLSL);
	out[2] = (uint32_t) sizeof(MaterialGLSL);
	out[3] = (uint32_t) sizeof(SphereGLSL);
}

// src/main.cpp:97 -- the per-frame seed is pcg32_random() on the global stream,
// which starts from PCG32_INITIALIZER (pcg_basic.h).  The global is static inside
// pcg_basic.c, so "process start" is restored by solving pcg32_srandom_r's two
// steps (pcg_basic.c:42-49) for the (initstate, initseq) pair that lands on it.
static uint64_t inv_mod_2_64(uint64_t a)
{
	uint64_t x = a; // Newton: doubles the number of correct low bits each step
	for (int i = 0; i < 6; i++) x *= 2 - a * x;
	return x;
}
void ref_pcg32_restart(void)
{
	const uint64_t S = 0x853c49e6748fea9bULL, I = 0xda3e39cb94b95bdbULL, M = 6364136223846793005ULL;
	uint64_t initstate = (S - I) * inv_mod_2_64(M) - I;
	pcg32_srandom(initstate, I >> 1);
}
uint32_t ref_pcg32_next(void) { return pcg32_random(); }

// src/scene/triangle.cpp:49-64
void ref_pack_triangle(const float v[9], const float uv[6], const float n[9], uint32_t mat_index, void *out64)
{
	TriangleGLSL t(glm::vec3(v[0], v[1], v[2]), glm::vec3(v[3], v[4], v[5]), glm::vec3(v[6], v[7], v[8]),
	               glm::vec2(uv[0], uv[1]), glm::vec2(uv[2], uv[3]), glm::vec2(uv[4], uv[5]),
	               glm::vec3(n[0], n[1], n[2]), glm::vec3(n[3], n[4], n[5]), glm::vec3(n[6], n[7], n[8]),
	               mat_index);
	memcpy(out64, &t, sizeof t);
}

// src/math/math.cpp:205-211
void ref_oct_encode(const float n[3], float out[2])
{
	glm::vec2 e = pixl::octahedral_normal_encoding(glm::vec3(n[0], n[1], n[2]));
	out[0] = e.x;
	out[1] = e.y;
}

// src/scene/material.cpp:7-22
void ref_pack_material(const float diffuse[3], const float specular[3], const float le[3], float roughness,
                       int tex_index, int type, void *out48)
{
	MaterialGLSL m(glm::vec3(diffuse[0], diffuse[1], diffuse[2]), glm::vec3(specular[0], specular[1], specular[2]),
	               glm::vec3(le[0], le[1], le[2]), roughness, tex_index, (MaterialType) type);
	memcpy(out48, &m, sizeof m);
}

// src/scene/sphere.cpp:5-9
void ref_pack_sphere(const float origin[3], float radius, uint32_t mat_index, void *out32)
{
	SphereGLSL s(glm::vec3(origin[0], origin[1], origin[2]), radius, mat_index);
	memcpy(out32, &s, sizeof s);
}

// src/scene/bvh.cpp:31-75.  nodes_out must hold 2n+1 nodes, sorted_out n triangles.
uint32_t ref_calculate_bvh(const void *tris_in, uint32_t n, void *nodes_out, void *sorted_out)
{
	Array<TriangleGLSL> in(n);
	memcpy(in._data, tris_in, (size_t) n * sizeof(TriangleGLSL));
	in.size = n;
	Array<TriangleGLSL> sorted;
	Array<BVHNodeGLSL> nodes = CalculateBVH(in, sorted);
	memcpy(nodes_out, nodes._data, (size_t) nodes.size * sizeof(BVHNodeGLSL));
	memcpy(sorted_out, sorted._data, (size_t) sorted.size * sizeof(TriangleGLSL));
	return nodes.size;
}

// src/scene/triangle.cpp:73-86
uint32_t ref_find_emissive_tris(const void *tris, uint32_t ntris, const void *mats, uint32_t nmats, uint32_t *out)
{
	Array<TriangleGLSL> t(ntris);
	memcpy(t._data, tris, (size_t) ntris * sizeof(TriangleGLSL));
	t.size = ntris;
	Array<MaterialGLSL> m(nmats);
	memcpy(m._data, mats, (size_t) nmats * sizeof(MaterialGLSL));
	m.size = nmats;
	Array<uint32> e = FindEmissiveTris(t, m);
	for (uint32_t i = 0; i < e.size; i++) out[i] = e[i];
	return e.size;
}

// src/scene/sphere.cpp:11-24
uint32_t ref_find_emissive_spheres(const void *spheres, uint32_t nspheres, const void *mats, uint32_t nmats,
                                   uint32_t *out)
{
	if (nspheres == 0) return 0;
	Array<SphereGLSL> s(nspheres);
	memcpy(s._data, spheres, (size_t) nspheres * sizeof(SphereGLSL));
	s.size = nspheres;
	Array<MaterialGLSL> m(nmats);
	memcpy(m._data, mats, (size_t) nmats * sizeof(MaterialGLSL));
	m.size = nmats;
	Array<uint32> e = FindEmissiveSpheres(s, m);
	for (uint32_t i = 0; i < e.size; i++) out[i] = e[i];
	return e.size;
}

// ---- CPU twins of the shader's sampling helpers (src/math/math.cpp:132-194 == framebuffer.comp:141-175, 707-723).
// The live path never calls them, but they are the reference author's own statement of the same math, so they pin the
// oracle's restatement of those shader lines independently of the shader text.

// src/math/math.cpp:132-151 (libm sinf/cosf)
void ref_map_to_unit_sphere(const float uv[2], float out[3])
{
	glm::vec3 p = pixl::map_to_unit_sphere(glm::vec2(uv[0], uv[1]));
	out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

// src/math/math.cpp:153-157
void ref_cosine_hemisphere(const float uv[2], const float n[3], float out[3])
{
	glm::vec3 p = pixl::map_to_unit_hemisphere_cosine_weighted_criver(glm::vec2(uv[0], uv[1]), glm::vec3(n[0], n[1], n[2]));
	out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

// src/math/math.cpp:159-175 over a Triangle built by its own constructor (edge1 = v1 - v0, edge2 = v2 - v0, triangle.cpp:12-17)
void ref_map_to_triangle(const float uv[2], const float v[9], float out[3])
{
	glm::vec3 z(0.0f);
	Triangle t(glm::vec3(v[0], v[1], v[2]), glm::vec3(v[3], v[4], v[5]), glm::vec3(v[6], v[7], v[8]), z, z, z, 0);
	glm::vec3 p = pixl::map_to_triangle(glm::vec2(uv[0], uv[1]), t);
	out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

// src/math/math.cpp:178-194: out = rows of the TRANSPOSED mat3(t, n, bt), i.e. t, n, bt -- what framebuffer.comp:907 uses
void ref_construct_tnb(const float n_in[3], float out[9])
{
	glm::vec3 n(n_in[0], n_in[1], n_in[2]);
	glm::mat3 m = pixl::construct_TNB_matrix(n); // = transpose(mat3(t, n, bt)): column j of m holds (t[j], n[j], bt[j])
	// row i of m = (m[0][i], m[1][i], m[2][i]) = basis vector i
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) out[3 * i + j] = m[j][i];
}

// Closest hit by the reference's vendored library itself: Sweep-SAH BVH over the given triangles (same builder call as
// CalculateBVH, src/scene/bvh.cpp:31-45), SingleRayTraverser (thirdparty/bvh/include/bvh/single_ray_traverser.hpp:65-126:
// same near-first child ordering, own slab test) and ClosestPrimitiveIntersector over bvh::Triangle::intersect
// (triangle.hpp:99-126: a different Moeller-Trumbore arrangement, inclusive t bounds, no |det| cut-off).  Not bit-comparable
// with the shader's arithmetic; agrees with it wherever the closest hit is not a near-tie.  prim indexes tris_in.
void ref_bvh_closest_hits(const void *tris_in, uint32_t n_tris, const float *ro, const float *rd, uint32_t n_rays, float tmin, float tmax,
                          uint32_t *prim, float *t_out)
{
	const TriangleGLSL *tg = (const TriangleGLSL *) tris_in;
	std::vector<bvh::Triangle<float>> prims;
	prims.reserve(n_tris);
	for (uint32_t i = 0; i < n_tris; i++) {
		const TriangleGLSL &t = tg[i];
		prims.emplace_back(bvh::Vector3<float>(t.v0().x, t.v0().y, t.v0().z), bvh::Vector3<float>(t.v1().x, t.v1().y, t.v1().z),
		                   bvh::Vector3<float>(t.v2().x, t.v2().y, t.v2().z));
	}
	auto [bboxes, centers] = bvh::compute_bounding_boxes_and_centers(prims.data(), prims.size());
	auto global_bbox = bvh::compute_bounding_boxes_union(bboxes.get(), prims.size());
	bvh::Bvh<float> tree;
	bvh::SweepSahBuilder<bvh::Bvh<float>> builder(tree);
	builder.build(global_bbox, bboxes.get(), centers.get(), prims.size());
	bvh::ClosestPrimitiveIntersector<bvh::Bvh<float>, bvh::Triangle<float>> isect(tree, prims.data());
	bvh::SingleRayTraverser<bvh::Bvh<float>> trav(tree);
	for (uint32_t r = 0; r < n_rays; r++) {
		bvh::Ray<float> ray(bvh::Vector3<float>(ro[3 * r], ro[3 * r + 1], ro[3 * r + 2]), bvh::Vector3<float>(rd[3 * r], rd[3 * r + 1], rd[3 * r + 2]), tmin, tmax);
		auto hit = trav.traverse(ray, isect);
		prim[r] = hit ? (uint32_t) hit->primitive_index : 0xffffffffu;
		t_out[r] = hit ? hit->distance() : 0.0f;
	}
}

// ---- the reference's own glTF loader + Model (src/loader.cpp:13-401, src/scene/model.cpp:15-68) and the scene set-up
// of src/main.cpp:18-33.  Texture-less files only: with textures the loader calls into GL (loader.cpp:48-58, 242-265),
// whose entry points are not loaded here.  ops: a list of (kind, x, y, z) Model calls applied after LoadGLTF --
// kind 0 Translate, 1 Rotate, 2 Scale, 3 ApplyModelMatrixToTris -- so the caller can replay src/main.cpp:26-29.
// Outputs: fat triangles as 25 floats each (v0 v1 v2 n0 n1 n2 uv0 uv1 uv2) + mat index, MaterialGLSL bytes, and the
// TriangleGLSL bytes of Model::ConvertToSSBOFormat.  Returns the triangle count, or -1 when LoadGLTF returns false.
int ref_load_gltf(const char *path, const float *ops, uint32_t n_ops, uint32_t max_tris, float *fat25, uint32_t *mat_index,
                  void *tris64, uint32_t max_mats, void *mats48, uint32_t *n_mats_out)
{
	Model model;
	if (!LoadGLTF(path, model)) return -1;
	for (uint32_t i = 0; i < n_ops; i++) {
		const float *o = ops + 4 * i;
		const glm::vec3 v(o[1], o[2], o[3]);
		switch ((int) o[0]) {
		case 0: model.Translate(v); break;
		case 1: model.Rotate(v); break;
		case 2: model.Scale(v); break;
		case 3: model.ApplyModelMatrixToTris(); break;
		}
	}
	if (model.triangles.size > max_tris || model.materials.size > max_mats) return -2;
	Array<TriangleGLSL> packed = model.ConvertToSSBOFormat();
	for (uint32_t i = 0; i < model.triangles.size; i++) {
		const Triangle &t = model.triangles[i];
		const float rec[25] = {t.v0.x, t.v0.y, t.v0.z, t.v1.x, t.v1.y, t.v1.z, t.v2.x, t.v2.y, t.v2.z,
		                       t.n0.x, t.n0.y, t.n0.z, t.n1.x, t.n1.y, t.n1.z, t.n2.x, t.n2.y, t.n2.z,
		                       t.uv0.x, t.uv0.y, t.uv1.x, t.uv1.y, t.uv2.x, t.uv2.y, 0.0f};
		memcpy(fat25 + 25 * (size_t) i, rec, sizeof rec);
		mat_index[i] = t.mat_index;
	}
	memcpy(tris64, packed._data, (size_t) packed.size * sizeof(TriangleGLSL));
	memcpy(mats48, model.materials._data, (size_t) model.materials.size * sizeof(MaterialGLSL));
	*n_mats_out = model.materials.size;
	return (int) model.triangles.size;
}

// The two image steps of the loader's textured path on their own (they need no GL): src/loader.cpp:209-214 and :224-226.
int ref_stbi_load_rgb8(const uint8_t *file, size_t n, uint8_t *out, size_t capacity, int *w, int *h, int *channels)
{
	stbi_uc *img = stbi_load_from_memory(file, (int) n, w, h, channels, 3);
	if (!img) return 1;
	const size_t bytes = (size_t) *w * *h * 3;
	if (bytes <= capacity) memcpy(out, img, bytes);
	stbi_image_free(img);
	return bytes <= capacity ? 0 : 2;
}
// The env map exactly as Display::InitRenderBuffer gets it (src/display/display.cpp:183-184, minus the file name):
// stbi_hdr_to_ldr_gamma(1.0), then stbi_load with 3 channels.
int ref_env_hdr_load(const char *path, uint8_t *out, size_t capacity, int *w, int *h)
{
	int c = -1;
	stbi_hdr_to_ldr_gamma(1.0);
	uint8_t *data = stbi_load(path, w, h, &c, 3);
	if (!data) return 1;
	const size_t bytes = (size_t) *w * *h * 3;
	if (bytes <= capacity) memcpy(out, data, bytes);
	stbi_image_free(data);
	return bytes <= capacity ? 0 : 2;
}
int ref_stbir_resize_512(const uint8_t *rgb, int w, int h, uint8_t *out512) { return stbir_resize_uint8(rgb, w, h, 0, out512, 512, 512, 0, 3); }

} // extern "C"
