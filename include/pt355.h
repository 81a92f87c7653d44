/* pt355 -- C ABI of the MI355X-native render hot path.
 *
 * This library replaces ONE seam of limepixl/pathtracer: the OpenGL upload +
 * compute dispatch of shaders/framebuffer.comp.  Every entry point below names
 * the reference interface it stands in for (paths relative to the reference
 * repository root).  Plain C types only; no exceptions cross this boundary; all
 * functions return 0 on success and a PT_E* code otherwise (pt_last_error gives
 * the text).  One pt_ctx per host thread; a context owns its HIP streams.
 *
 * Buffer layouts are the reference's, byte for byte:
 *   sphere    32 B  src/scene/sphere.hpp:7-15      shaders/framebuffer.comp:54-58
 *   triangle  64 B  src/scene/triangle.hpp:26-43   shaders/framebuffer.comp:46-52  (BVH leaf order)
 *   material  48 B  src/scene/material.hpp:13-31   shaders/framebuffer.comp:60-65
 *   bvh node  32 B  src/scene/bvh.h:7-14           shaders/framebuffer.comp:67-71
 *   camera    48 B  src/scene/camera.hpp:26-39     shaders/framebuffer.comp:14-19
 *   framedata 16 B  src/display/display.hpp:7-13   shaders/framebuffer.comp:21-24
 */
#ifndef PT355_H
#define PT355_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_EINVAL 1   /* bad argument / shape the kernels do not accept */
#define PT_EHIP 2     /* a HIP call failed; pt_last_error has hipGetErrorString */
#define PT_ESTATE 3   /* call order: e.g. render before upload_scene/set_camera */
#define PT_ENOMEM 4

typedef struct pt_ctx pt_ctx;

/* pt_config.flags */
/* NEE visibility.  The reference traces a shadow ray to its CLOSEST hit and compares that point with the light sample,
 * one-sided per component (framebuffer.comp:970-974).  Default: the same verdict for every ray, but the traversal
 * stops at the first hit that already decides it (a hit that loses the compare on a component the ray advances
 * along cannot be rescued by a closer one) -- the image is bit-identical to the full traversal's. */
#define PT_FLAG_SHADOW_ANYHIT      1u /* fast mode: ANY hit at least 0.02 in front of the light sample occludes it; this ignores
                                         the one-sided quirk of the compare, so the image differs where that quirk leaks light */
#define PT_FLAG_SHADOW_FULL_TRAVERSAL 8u /* always traverse to the closest hit (like-for-like work counters against the oracle) */
#define PT_FLAG_TRACE_DEAD_PATHS   2u /* keep tracing paths whose throughput is exactly 0 (the reference does,
                                         framebuffer.comp:899-1078); default retires them: fewer rays, and the same image
                                         unless a later bounce of such a path would have produced inf or NaN (0 * NaN = NaN
                                         in the shader's color += throughput * X, e.g. a light sample at zero distance) --
                                         set the flag to reproduce those NaN pixels too */
#define PT_FLAG_COUNT_TRAVERSAL    4u /* accumulate node-pair visits / triangle tests (slower; for roofline inputs) */

/* pt_config.partition: how the image is shared between the ranks of one node */
#define PT_PARTITION_TILES   0u /* 8x8 tiles dealt round-robin: tile t belongs to rank t % world */
#define PT_PARTITION_FULL    1u /* every rank owns every pixel (single GPU, or sample-interleaved frames) */

/* Replaces: Display::Display / InitRenderBuffer (src/display/display.cpp:34-199) as far as the compute
 * path needs it -- the RGBA32F image (image unit 0), FrameData UBO (binding 4) -- and the constants of
 * src/defines.hpp:15-21. width and height must be multiples of 8 (glDispatchCompute(W/8,H/8,1)). */
typedef struct pt_config {
	uint32_t width, height;
	int32_t  device;       /* HIP device ordinal */
	uint32_t flags;
	uint32_t partition;    /* PT_PARTITION_* */
	uint32_t rank, world;  /* this context's share; world = 0 or 1 means the whole image */
	uint32_t max_batch;    /* frames pt_render traces together (1..32); 0 = default (32, less for images above 2 Mpixel:
	                          at most 2^26 paths = ~23 GB of queues are in flight).
	                          Results do not depend on it: samples are blended into the image in frame order. */
	uint32_t pipelines;    /* independent frame pipelines (streams + queues) batches alternate between (1..4); 0 = default (1).
	                          Results do not depend on it either. */
	uint32_t reserved[7];
} pt_config;

/* Replaces: the six PushDataToSSBO calls, bindings 0..5 in call order (src/main.cpp:57-63,
 * src/core/utils.h:6-19; shader side framebuffer.comp:75-103) and the two sampler units
 * (env map: src/display/display.cpp:174-193; texture array: src/loader.cpp:50-58,242-265).
 * Arrays are copied; the caller keeps ownership.  Empty arrays (count 0) may be NULL. */
typedef struct pt_scene_desc {
	const void     *spheres;        uint32_t n_spheres;
	const void     *tris;           uint32_t n_tris;
	const uint32_t *light_tris;     uint32_t n_light_tris;
	const void     *materials;      uint32_t n_materials;
	const void     *nodes;          uint32_t n_nodes;
	const uint32_t *light_spheres;  uint32_t n_light_spheres;
	const uint8_t  *env_rgb8;       uint32_t env_w, env_h;   /* NULL: texture without storage -> black */
	const uint8_t  *tex_rgb8;       uint32_t tex_layers;     /* 512 x 512 x layers, RGB8 */
	/* sampler state of the texture array: the GL enums the loader takes from the glTF sampler and sets array-wide
	 * (src/loader.cpp:242-251).  0 = the loader's defaults GL_REPEAT (10497) / GL_LINEAR (9729); also understood:
	 * GL_CLAMP_TO_EDGE 33071, GL_MIRRORED_REPEAT 33648, GL_NEAREST 9728.  The minification filter never applies: a compute
	 * shader's texture() samples level 0. */
	uint32_t tex_wrap_s, tex_wrap_t, tex_mag_filter;
} pt_scene_desc;

/* Rays are counted when they are enqueued to a trace kernel. */
typedef struct pt_counters {
	uint64_t frames;
	uint64_t paths;          /* pixel-samples started */
	uint64_t rays_primary;
	uint64_t rays_shadow;
	uint64_t rays_bounce;
	uint64_t node_steps;     /* only with PT_FLAG_COUNT_TRAVERSAL */
	uint64_t tri_tests;      /* only with PT_FLAG_COUNT_TRAVERSAL */
	uint64_t stack_spills;   /* traversal-stack pushes that left the LDS short stack */
	double   ms_trace;       /* HIP-event time of all trace launches since the last reset */
	double   ms_shade;       /* ... of raygen + shade launches */
	uint64_t trace_launches;
	uint64_t wave_iters;     /* only with PT_FLAG_COUNT_TRAVERSAL: wave-iterations of the traversal loop; (node_steps + tri_tests) /
	                            (64 * wave_iters) = the share of lanes doing work per iteration */
	uint64_t top_steps;      /* only with PT_FLAG_COUNT_TRAVERSAL: those of node_steps + tri_tests whose record came from the copy of
	                            the tree top in LDS instead of global memory */
	uint64_t reserved[3];
} pt_counters;

int pt_create(const pt_config *cfg, pt_ctx **out);
int pt_destroy(pt_ctx *ctx);
const char *pt_last_error(const pt_ctx *ctx); /* ctx may be NULL: error of the last failed pt_create */

int pt_upload_scene(pt_ctx *ctx, const pt_scene_desc *scene);

/* Replaces: glNamedBufferSubData(cam.cam_ubo, ...) of src/main.cpp:90-94 (CameraGLSL = 12 floats). */
int pt_set_camera(pt_ctx *ctx, const float cam[12]);

/* Replaces: editing shaders/framebuffer.comp:1109-1111, where the reference selects one of its three estimators by
 * (un)commenting a line.  MIS (the live one) is the default. */
#define PT_ESTIMATOR_MIS  0u /* estimator_path_tracing_mis,  framebuffer.comp:874-1081 */
#define PT_ESTIMATOR_NEE  1u /* estimator_path_tracing_nee,  framebuffer.comp:725-867 */
#define PT_ESTIMATOR_BRDF 2u /* estimator_path_tracing_BRDF, framebuffer.comp:647-693 */
int pt_set_estimator(pt_ctx *ctx, uint32_t estimator);

/* Replaces: one iteration of src/main.cpp:97-101 -- FrameData upload, glDispatchCompute(W/8,H/8,1),
 * glMemoryBarrier.  frame_data = {seed, frame#, bounces, 0}.  One sample per pixel, blended into the
 * running mean with weight 1/(frame#+1) (framebuffer.comp:1124-1129).  Asynchronous on the context's
 * stream; pt_sync or any read-back orders it. */
int pt_render_frame(pt_ctx *ctx, const uint32_t frame_data[4]);

/* Same result as n_frames consecutive pt_render_frame calls (frame numbers first_frame.., seeds[i] per frame), but up
 * to pt_config.max_batch frames share one set of kernel launches; their samples are blended in frame order. */
int pt_render(pt_ctx *ctx, uint32_t first_frame, uint32_t n_frames, const uint32_t *seeds, uint32_t bounces);

int pt_sync(pt_ctx *ctx);

/* The reference never reads the image back (it blits it: src/main.cpp:108-114); a headless build must.
 * rgba receives width*height*4 floats, row 0 = bottom row of the displayed image. */
int pt_read_framebuffer(pt_ctx *ctx, float *rgba, size_t n_floats);
int pt_clear_framebuffer(pt_ctx *ctx);
/* Restores a running mean saved with pt_read_framebuffer (checkpoint/resume: the accumulation state is the image plus
 * the next frame number; the reference keeps it only in the GL image and loses it on any input). */
int pt_write_framebuffer(pt_ctx *ctx, const float *rgba, size_t n_floats);
/* Device address of the RGBA32F image, for callers that move it with RCCL (torch.distributed). */
int pt_framebuffer_device(pt_ctx *ctx, void **dev_ptr);

/* Test / bench boundary for the bit-exact closest-hit contract: intersect() of
 * framebuffer.comp:423-456 for n rays (AoS xyz, host memory).  kind: 0 miss, 1 triangle, 2 sphere;
 * prim: full 32-bit primitive index; t_h: fp16 bits of HitData.t; uv: optional, 2 floats per ray --
 * kind 1: the barycentrics (u, v) of the accepted triangle, from which framebuffer.comp:294-310 derives uvs and
 * normal; kind 2: HitData.uvs itself, widened to float -- the sphere branch (framebuffer.comp:447-451) does not
 * write uvs, so they are those of the last triangle the BVH pass accepted (0 if none), and calc_BRDF textures the
 * sphere's material with them (framebuffer.comp:599-603); kind 0: 0. */
int pt_trace_closest(pt_ctx *ctx, const float *ro, const float *rd, uint32_t n,
                     uint32_t *prim, uint16_t *t_h, uint8_t *kind, float *uv);
/* Same kernel on rays already resident in HBM: ray6 = structure-of-arrays ro.x[n] ro.y[n] ro.z[n] rd.x[n] rd.y[n]
 * rd.z[n]; hit4 = one 16-byte record per ray, 16-byte aligned: {prim u32, t_h | kind<<16, u f32, v f32}; a sphere's
 * record is {sphere u32, t_h | 2<<16, HitData.uvs as fp16 bits x | y<<16, 0}.  Asynchronous. */
int pt_trace_closest_device(pt_ctx *ctx, const void *ray6_dev, uint32_t n, void *hit4_dev);

/* No counterpart in the reference.  pt_trace_closest_device for callers that cannot allocate: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernel, same arguments, same hit records; the
 * kernel's scratch (work-unit counters and statistics block, traversal-stack spill tail: 26 MiB) is a file-scope
 * device array that every GPU the library runs on holds until the process ends, and the call makes no hipMalloc /
 * hipFree / event creation.  At most PT_NOALLOC_MAX_RAYS rays per call (PT_EINVAL beyond).  All calls on one GPU share
 * that GPU's array: calls that overlap on one GPU (any contexts, any streams) race; back-to-back calls on the
 * context's stream are fine.  Asynchronous like pt_trace_closest_device; not timed in pt_counters.  It never synchronises:
 * with renders of this context still in flight it returns PT_ESTATE (call pt_sync first).  Its grid is capped at 2,048 waves
 * (a quarter of pt_create's grid on a 256-CU part -- what the 26 MiB spill tail is sized for): measured 0.83 x the rate of
 * pt_trace_closest_device on 4 M incoherent rays (6.3 against 7.7 Grays/s, tests/test_zz_no_alloc.py prints it). */
#define PT_NOALLOC_MAX_RAYS (1u << 26)
int pt_trace_closest_device_noalloc(pt_ctx *ctx, const void *ray6_dev, uint32_t n, void *hit4_dev);

/* What pt_create settled on (no reference counterpart): info[0] frames per launch set, [1] pipelines, [2] queue stride
 * (slots), [3] slots per queue region, [4] paths per frame on this rank, [5] k_trace grid (blocks), [6] k_trace block size (threads), [7] bytes of the traversal blob's head that k_trace keeps in LDS. */
int pt_get_info(const pt_ctx *ctx, uint32_t info[8]);

int pt_get_counters(pt_ctx *ctx, pt_counters *out);
int pt_reset_counters(pt_ctx *ctx);

/* Multi-GPU (no counterpart in the reference, which is single-GPU): the pixels a rank owns, in the
 * order pt_pack_owned writes them.  Pure host arithmetic -- usable without a GPU. */
uint32_t pt_partition_count(uint32_t width, uint32_t height, uint32_t partition, uint32_t rank, uint32_t world);
int pt_partition_pixels(uint32_t width, uint32_t height, uint32_t partition, uint32_t rank, uint32_t world,
                        uint32_t *pixels /* y*width+x, pt_partition_count entries */);
/* framebuffer -> dst_dev: RGBA of this rank's pixels, packed in pt_partition_pixels order (device memory). */
int pt_pack_owned(pt_ctx *ctx, void *dst_dev);
/* src_dev (packed by rank `from_rank` of `world`) -> this context's framebuffer. */
int pt_unpack_from(pt_ctx *ctx, const void *src_dev, uint32_t from_rank, uint32_t world);

/* ---- Multi-device in one process ("pt_create(device list ...)", "pt_read_framebuffer gathers across
 * GPUs").  No counterpart in the reference, which is single-GPU: this is what lets a C++ caller shaped like its main()
 * (src/main.cpp:57-101: upload, set camera, one render call per frame, read the image) use every GPU of a node.
 * A group holds one pt_ctx per listed device (a device may be listed more than once: "virtual devices", how the
 * partition and gather logic is tested on one GPU); the scene is replicated, rendering needs no exchange, and the
 * finished image is gathered onto the first device with peer copies over xGMI.  cfg: width, height, flags, max_batch,
 * pipelines are used; device / partition / rank / world are set per member.  One group per host thread. */
#define PT_GROUP_TILES   0u /* 8x8 tiles dealt round-robin over the members; image bit-identical to one device's */
#define PT_GROUP_SAMPLES 1u /* frame f -> member f % n over the full image; gather = sum_i (n_i / N) * mean_i (agrees with the
                               sequential running mean to ~1e-6 relative: the summation order differs) */
typedef struct pt_group pt_group;
int pt_group_create(const pt_config *cfg, const int32_t *devices, uint32_t n_devices, uint32_t mode, pt_group **out);
int pt_group_destroy(pt_group *g);
const char *pt_group_last_error(const pt_group *g); /* g may be NULL: error of the last failed pt_group_create.  Right after a SUCCESSFUL
                                                       pt_group_create the text is empty, or names the device pairs whose peer access
                                                       could not be enabled (their gathers then run on staged copies: slower, same result) */
uint32_t pt_group_size(const pt_group *g);
pt_ctx *pt_group_member(pt_group *g, uint32_t i);   /* the member's own context, e.g. for pt_get_counters per device */
int pt_group_upload_scene(pt_group *g, const pt_scene_desc *scene);
int pt_group_set_camera(pt_group *g, const float cam[12]);
int pt_group_set_estimator(pt_group *g, uint32_t estimator);
/* frames first_frame.. with seeds[i] per frame, like pt_render; asynchronous on every member */
int pt_group_render(pt_group *g, uint32_t first_frame, uint32_t n_frames, const uint32_t *seeds, uint32_t bounces);
int pt_group_sync(pt_group *g);
int pt_group_clear(pt_group *g);
/* gathers, then hands out the device address (on the first device) of the full RGBA32F image; valid until the next
 * group call */
int pt_group_gather(pt_group *g, void **image_dev);
/* pt_read_framebuffer across the group: gather + copy to the host */
int pt_group_read_framebuffer(pt_group *g, float *rgba, size_t n_floats);
int pt_group_get_counters(pt_group *g, pt_counters *out); /* sums over the members */

/* Numerics self-test hook: evaluates one primitive of the arithmetic contract on the device
 * (op codes as in oracle/oracle.h) so tests can pin device min/max/div/sqrt/fp16/transcendentals. */
int pt_debug_eval(pt_ctx *ctx, int op, const float *a, const float *b, uint32_t n, float *out);

/* ---- BVH build on the GPU.  The reference builds on one CPU thread (CalculateBVH, src/scene/bvh.cpp:31-75,
 * rebuilt bit for bit as pth_build_bvh); this is the same step for scenes where that build dominates the set-up, on ctx's
 * device, returned in the reference's own layout -- same arguments and output contract as pth_build_bvh: tris = n packed
 * 64-byte triangles in any order, nodes_out = room for 2n+1 32-byte nodes, sorted_out = n triangles in leaf order.
 * Two builders: a linear BVH (Morton order, radix tree, bottom-up bounds: the fastest) and, with PT_BVH_PLOC or-ed into
 * max_leaf, parallel locally-ordered clustering over the same Morton order (bottom-up merging by the area of the merged
 * box: a better tree for a longer build).  Subtrees of <= max_leaf triangles become leaves (max_leaf 1..16, 0 = 2).
 * Either tree differs from the Sweep-SAH tree (the bit-exactness contract covers the CPU builder); it is a valid input
 * of pt_upload_scene, the oracle and the reference's shader.  build_ms (optional) receives the device time of the
 * build proper. */
#define PT_BVH_PLOC 0x100u
int pt_build_bvh_device(pt_ctx *ctx, const void *tris, uint32_t n, uint32_t max_leaf, void *nodes_out, uint32_t *n_nodes_out,
                        void *sorted_out, float *build_ms);
/* pth_scene_finalize (pt355_host.h) with that builder in place of CalculateBVH */
struct pth_scene;
int pt_scene_finalize_device(pt_ctx *ctx, struct pth_scene *scene, uint32_t max_leaf, float *build_ms);

/* Debug read-back of one column group of a path queue (queue 0/1; field: 0 pix, 1 rng, 2 meta, 3 thr, 4 col,
 * 5 ray, 6 shadow ray, 7 nee, 8 bsdf, 9 hit, 10 shadow hit).  n_words = columns * stride.  No reference counterpart. */
int pt_debug_read_queue(pt_ctx *ctx, uint32_t queue, uint32_t field, void *out, size_t n_words, uint32_t *stride, uint32_t *count);

#ifdef __cplusplus
}
#endif
#endif
