/* pt355 host side -- the CPU pieces of the path that the north star keeps on the host:
 * SSBO packing, the Sweep-SAH BVH build, emissive-primitive lists, the per-frame host PCG seed,
 * procedural stand-in scenes and the display transfer function.  Part of libpt355.so, no GPU needed.
 * Each entry names the reference code whose OUTPUT BYTES it reproduces (tests pin them against the
 * reference's own translation units compiled in the build container: oracle/_ref).
 */
#ifndef PT355_HOST_H
#define PT355_HOST_H
#include <stddef.h>
#include <stdint.h>
#include "pt355.h"

#ifdef __cplusplus
extern "C" {
#endif

/* MaterialType, src/scene/material.hpp:5-11 */
#define PTH_MAT_LIGHT          (-1)
#define PTH_MAT_LAMBERTIAN       0
#define PTH_MAT_OREN_NAYAR       1
#define PTH_MAT_SPECULAR_METAL   2

/* glm::packHalf2x16 as the reference uses it (thirdparty/glm/glm/detail/type_half.inl:105-238:
 * round-half-up on the magnitude, not RTNE) -- src/scene/triangle.cpp:53-62 */
uint32_t pth_pack_half2x16(float x, float y);
/* pixl::octahedral_normal_encoding, src/math/math.cpp:196-211 */
void pth_oct_encode(const float n[3], float out[2]);
/* TriangleGLSL::TriangleGLSL, src/scene/triangle.cpp:49-64.  v, n: 3x xyz; uv: 3x st. */
void pth_pack_triangle(const float v[9], const float uv[6], const float n[9], uint32_t mat_index, void *out64);
/* MaterialGLSL::MaterialGLSL, src/scene/material.cpp:7-22 (metal roughness clamped to >= 0.01) */
void pth_pack_material(const float diffuse[3], const float specular[3], const float le[3], float roughness,
                       int tex_index, int type, void *out48);
/* SphereGLSL::SphereGLSL, src/scene/sphere.cpp:5-9 */
void pth_pack_sphere(const float origin[3], float radius, uint32_t mat_index, void *out32);
/* CameraGLSL::CameraGLSL, src/scene/camera.cpp:75-80 */
void pth_pack_camera(const float origin[3], const float forward[3], const float right[3], float speed, float sens,
                     float out12[12]);

/* CalculateBVH, src/scene/bvh.cpp:31-75 over bvh::SweepSahBuilder (thirdparty/bvh/include/bvh/
 * sweep_sah_builder.hpp:43-254, top_down_builder.hpp:47-72, radix_sort.hpp:21-95), single-threaded
 * order.  nodes_out: room for 2n+1 nodes; sorted_out: n triangles.  Returns the node count (0 on error). */
uint32_t pth_build_bvh(const void *tris, uint32_t n, void *nodes_out, void *sorted_out);

/* FindEmissiveTris, src/scene/triangle.cpp:73-86 (ANY Le component >= 1e-4) */
uint32_t pth_find_emissive_tris(const void *tris, uint32_t n_tris, const void *mats, uint32_t n_mats, uint32_t *out);
/* FindEmissiveSpheres, src/scene/sphere.cpp:11-24 (ALL Le components > 1e-4) */
uint32_t pth_find_emissive_spheres(const void *spheres, uint32_t n_spheres, const void *mats, uint32_t n_mats,
                                   uint32_t *out);

/* pcg32_random() on the default global stream, src/main.cpp:97 with
 * thirdparty/pcg-c-basic-0.9/pcg_basic.c:35,60-72: outputs number first..first+n-1 since process start. */
void pth_frame_seeds(uint32_t first, uint32_t n, uint32_t *out);

/* ---- scene container: what src/main.cpp:18-63 assembles before the uploads -------------------- */
typedef struct pth_scene pth_scene;
pth_scene *pth_scene_create(void);
void pth_scene_destroy(pth_scene *s);
int pth_scene_add_material(pth_scene *s, const float diffuse[3], const float specular[3], const float le[3],
                           float roughness, int tex_index, int type); /* returns the material index */
int pth_scene_add_sphere(pth_scene *s, const float origin[3], float radius, uint32_t mat_index);
/* count "fat" triangles (src/scene/triangle.hpp:8-24): v 9 floats, n 9 floats, uv 6 floats (uv may be NULL) */
int pth_scene_add_triangles(pth_scene *s, const float *v, const float *n, const float *uv, const uint32_t *mat,
                            uint32_t count);
/* the fat triangles added so far, in insertion order (any pointer may be NULL) -- e.g. to write a scene out as glTF */
uint32_t pth_scene_fat_count(const pth_scene *s);
int pth_scene_get_fat(const pth_scene *s, float *v9, float *n9, float *uv6, uint32_t *mat);
/* Model::ConvertToSSBOFormat + CalculateBVH + FindEmissive* (src/main.cpp:31-55) */
int pth_scene_finalize(pth_scene *s);
/* Same, with the BVH built by `build` instead of CalculateBVH: build(tris, n, nodes_out [2n+1 nodes of room],
 * sorted_out [n triangles], user) returns the node count, 0 on failure.  pt_scene_finalize_device (pt355.h) passes
 * the GPU builder. */
typedef uint32_t (*pth_bvh_builder)(const void *tris, uint32_t n, void *nodes_out, void *sorted_out, void *user);
int pth_scene_finalize_with(pth_scene *s, pth_bvh_builder build, void *user);
/* pointers stay valid until the next mutation / destroy */
int pth_scene_desc(const pth_scene *s, pt_scene_desc *out);
uint64_t pth_scene_hash(const pth_scene *s); /* FNV-1a over the six arrays, binding order */

/* `count` MaterialGLSL records (48 bytes each) appended as they are; returns the index of the first one */
uint32_t pth_scene_add_packed_materials(pth_scene *s, const void *materials48, uint32_t count);

/* ---- glTF loader + Model: src/loader.h:4 LoadGLTF(path, Model&), src/scene/model.h:7-29 ------------------------
 * Own parser (JSON, GLB, accessors; PNG textures over zlib) in place of the reference's vendored cgltf / stb; what the
 * loader does with the data is the reference's, quirks included (per-mesh arrays appended per primitive, index byteOffset
 * ignored, one material per primitive, metals -> Lambert roughness 2.0, y <-> z swap of node scale / translation, every
 * node baked into all triangles): output bytes pinned against the reference's own loader, tests/test_gltf_loader.py.
 * The GL texture array of src/loader.cpp:48-58, 242-265 becomes a host RGB8 layer array + the sampler's GL enums.
 * Errors: return code != 0 and pth_last_error() (thread-local text); where the reference would crash (no indices, no
 * normals, index out of range) this returns an error instead. */
typedef struct pth_model pth_model;
pth_model *pth_model_create(void);
void pth_model_destroy(pth_model *m);
const char *pth_last_error(void);
int pth_load_gltf(const char *path, pth_model *out);                /* LoadGLTF, src/loader.cpp:13-401 */
void pth_model_translate(pth_model *m, const float t[3]);           /* Model::Translate, src/scene/model.cpp:48-51 */
void pth_model_rotate(pth_model *m, const float r[3]);              /* Model::Rotate (x, then y, then z; radians), :53-58 */
void pth_model_scale(pth_model *m, const float s[3]);               /* Model::Scale, :60-68 */
void pth_model_apply_matrix(pth_model *m);                          /* Model::ApplyModelMatrixToTris, :27-42 */
uint32_t pth_model_triangle_count(const pth_model *m);
uint32_t pth_model_material_count(const pth_model *m);
/* copies out the fat triangles (v 9, n 9, uv 6 floats and the material index per triangle) and the MaterialGLSL records;
 * any pointer may be NULL */
int pth_model_get(const pth_model *m, float *v9, float *n9, float *uv6, uint32_t *mat, void *materials48);
/* the texture array's content: layers x 512 x 512 x RGB8 (NULL when the file has no textures) and sampler[3] =
 * {wrap_s, wrap_t, mag_filter} as GL enums -> pt_scene_desc.tex_rgb8 / tex_layers / tex_wrap_s / tex_wrap_t / tex_mag_filter */
int pth_model_textures(const pth_model *m, const uint8_t **rgb8, uint32_t *layers, uint32_t sampler[3]);
/* src/main.cpp:31-36: the model's triangles and materials become the scene's (then pth_scene_add_material / _sphere for
 * src/main.cpp:44-51, then pth_scene_finalize) */
int pth_scene_add_model(pth_scene *s, const pth_model *m);
/* the two image steps of the textured path on their own (stbi_load_from_memory with 3 channels; stbir_resize_uint8 to
 * 512 x 512: src/loader.cpp:209-232), exposed so tests can check each against the reference's stb build.  The decoder
 * takes PNG (8/16-bit, non-interlaced) and JPEG (8-bit baseline / extended / progressive, 1 or 3 components), chosen by
 * signature; its bytes equal stb_image's for both */
int pth_decode_image_rgb8(const uint8_t *png, size_t n, uint8_t *rgb_out, size_t rgb_capacity, int *w, int *h, int *channels);
int pth_resize_rgb8_512(const uint8_t *rgb, int w, int h, uint8_t *out512);

/* The environment map as Display::InitRenderBuffer loads it (src/display/display.cpp:183-189): a Radiance .hdr through
 * stbi_load with stbi_hdr_to_ldr_gamma(1.0), 3 channels -> 8 bits per channel, rows in file order = what goes to
 * pt_scene_desc.env_rgb8 / env_w / env_h.  out == NULL: only report the size.  Own RGBE reader. */
int pth_load_env_hdr(const char *path, uint8_t *out, size_t capacity, uint32_t *w, uint32_t *h);

/* Procedural stand-ins (no assets exist offline: the reference git-ignores .glb / .hdr files).
 * Cornell box: 5 walls + ceiling light quad + two boxes = 36 triangles, placed like src/main.cpp:26-28.
 * flags bit0: add the four gold GGX spheres of src/main.cpp:44-51. */
int pth_scene_make_cornell(pth_scene *s, uint32_t flags);
/* Cornell box + a displaced icosphere ("bunny stand-in"): subdiv 6 -> 81,920 triangles; `target_tris`
 * > 0 trims to that count (69,451 for the Stanford bunny).  mesh_material: PTH_MAT_*; flags as above. */
int pth_scene_make_bunny(pth_scene *s, uint32_t subdiv, uint32_t target_tris, uint32_t seed, int mesh_material,
                         float mesh_roughness, uint32_t flags);
/* n_tris random small triangles in the box (BVH / traversal stress; LCG s = s*1664525+1013904223) */
int pth_scene_make_soup(pth_scene *s, uint32_t n_tris, uint32_t seed, float size);
/* "atrium", the Sponza-class stand-in of SURVEY.md 8(d) C4: tiled floor, two rows of eight tessellated columns joined by
 * arches, textured walls, six ceiling light panels of different radiance, metal vases and spheres; 27 materials
 * (Lambert / Oren-Nayar / GGX), six of them sampling texture layers 0..5; ~17,600 * detail^2 triangles (detail 4:
 * 281 k).  pth_make_textures fills `layers` 512x512 RGB8 layers (six distinct procedural patterns) for it. */
int pth_scene_make_atrium(pth_scene *s, uint32_t detail, uint32_t seed);
int pth_make_textures(uint32_t layers, uint32_t seed, uint8_t *out);

/* shaders/framebuffer.frag:8-26: NaN -> red, Inf -> green, else pow(rgb, 1/2.2); 8-bit RGB out, top row first */
int pth_tonemap_rgb8(const float *rgba, uint32_t width, uint32_t height, uint8_t *rgb8);
/* Image files (no reference counterpart: it only blits to a window, src/main.cpp:108-120).  PPM: the tonemapped image.
 * PFM: RGB float, bottom row first = the framebuffer's own row order; the accumulator's checkpoint format
 * (pt_read_framebuffer -> pth_write_pfm ... pth_read_pfm -> pt_write_framebuffer, then continue at the next frame#). */
int pth_write_ppm(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height);
int pth_write_pfm(const char *path, const float *rgba, uint32_t width, uint32_t height);
int pth_read_pfm(const char *path, float *rgba, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif
