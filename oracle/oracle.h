/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's render hot path
 * (shaders/framebuffer.comp), used as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product library
 * (libpt355.so) never includes, links or calls anything declared here.
 *
 * Parity status: the reference ships no golden vectors for this path and its
 * GLSL cannot be executed in the build container, so the
 * integrator half of this oracle has no end-to-end pin by the reference
 * ("parity unpinned" for whole frames).  Piecewise it is pinned by sources that
 * do not share its text (tests/test_oracle_pins.py):
 * (i) the reference's own CPU twins of the sampling helpers (src/math/math.cpp,
 * run through oracle/_ref, tests/golden/ref_math.npz), (ii) closest hits of the
 * reference's vendored BVH library (same file), (iii) the reference's own host
 * TUs for every buffer it consumes (ref_host.npz), (iv) closed forms,
 * quadrature and histograms for the BRDFs, the sampling densities and the
 * radiance of analytic scenes; the rest by line-by-line citation.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The six SSBOs in binding order (src/main.cpp:57-63) plus the two samplers. */
typedef struct oracle_scene {
	const void *spheres;        uint32_t n_spheres;        /* binding 0, 32 B each  */
	const void *tris;           uint32_t n_tris;           /* binding 1, 64 B each, BVH leaf order */
	const uint32_t *light_tris; uint32_t n_light_tris;     /* binding 2 */
	const void *materials;      uint32_t n_materials;      /* binding 3, 48 B each */
	const void *nodes;          uint32_t n_nodes;          /* binding 4, 32 B each */
	const uint32_t *light_spheres; uint32_t n_light_spheres; /* binding 5 */
	const uint8_t *env_rgb8;    uint32_t env_w, env_h;     /* sampler unit 1 (NULL: no storage -> black) */
	const uint8_t *tex_rgb8;    uint32_t tex_layers;       /* sampler unit 2, 512x512xN RGB8 (NULL: none) */
	uint32_t tex_wrap_s, tex_wrap_t, tex_mag_filter;       /* GL enums of loader.cpp:242-251; 0 = GL_REPEAT / GL_LINEAR */
} oracle_scene;

typedef struct oracle_counters {
	uint64_t paths;         /* pixel-samples started */
	uint64_t rays_primary;  /* intersect() calls: framebuffer.comp:880  */
	uint64_t rays_shadow;   /*                    framebuffer.comp:971  */
	uint64_t rays_bounce;   /*                    framebuffer.comp:1023 */
	uint64_t node_steps;    /* S: iterations of while(true) at framebuffer.comp:352 */
	uint64_t tri_tests;     /* T: intersect_triangle calls, framebuffer.comp:371 */
	uint64_t max_stack;     /* deepest traversal stack seen (reference has 16 slots) */
	uint64_t reserved;
} oracle_counters;

#define ORACLE_FLAG_LIBM 1u /* transcendental functions from glibc instead of the spec'd ptm_* forms */
#define ORACLE_FLAG_EST_NEE  2u /* estimator_path_tracing_nee  (framebuffer.comp:725-867) instead of the MIS estimator */
#define ORACLE_FLAG_EST_BRDF 4u /* estimator_path_tracing_BRDF (framebuffer.comp:647-693) */

/* intersect() of framebuffer.comp:423-456 for a batch of rays (AoS xyz).
 * kind: 0 miss, 1 triangle, 2 sphere.  prim: full 32-bit index.  t_h: fp16 bits.
 * uv (2 floats per ray, may be NULL): Moller-Trumbore u,v of the accepted triangle; for a sphere hit HitData.uvs as the
 * shader leaves them (fc:447-451 does not write uvs: the last accepted triangle's, else 0), widened to float.
 * steps/tests (may be NULL): per-ray S and T. */
int oracle_trace_closest(const oracle_scene *sc, const float *ro, const float *rd, uint32_t n,
                         uint32_t *prim, uint16_t *t_h, uint8_t *kind, float *uv,
                         uint32_t *steps, uint32_t *tests);

/* n_frames dispatches of main() (framebuffer.comp:1114-1130) into rgba (W*H*4
 * floats, read-modify-write running mean).  Frame f uses FrameData =
 * {seeds[f - first_frame], f, bounces, 0}. */
int oracle_render(const oracle_scene *sc, const float cam[12], uint32_t width, uint32_t height,
                  uint32_t first_frame, uint32_t n_frames, const uint32_t *seeds, uint32_t bounces,
                  float *rgba, uint32_t flags, uint32_t n_threads, oracle_counters *counters);

/* Unit-test hooks.  op codes are shared by convention with pt_debug_eval(). */
enum {
	ORACLE_OP_MIN = 0, ORACLE_OP_MAX = 1, ORACLE_OP_DIV = 2, ORACLE_OP_SQRT = 3,
	ORACLE_OP_F2H2F = 4, /* float -> fp16 -> float */
	ORACLE_OP_SIN = 5, ORACLE_OP_COS = 6, ORACLE_OP_TAN = 7, ORACLE_OP_ACOS = 8, ORACLE_OP_ATAN2 = 9,
	ORACLE_OP_HADD = 10, ORACLE_OP_HMUL = 11, /* fp16 add / mul of the fp16-rounded inputs */
	ORACLE_OP_RAND = 12, /* a = bit pattern of rng state; out = rand() float, b ignored */
	/* composite BRDF probes: directions drawn from the PCG stream seeded with the bits of a, b = roughness */
	ORACLE_OP_OREN_NAYAR = 13, ORACLE_OP_GGX_NEE = 14, ORACLE_OP_GGX_IS = 15
};
int oracle_math(int op, const float *a, const float *b, uint32_t n, float *out);
/* Vector-valued probes of single shader functions, for tests that pin the restatement against sources that do NOT
 * share its text: the reference's CPU twins in src/math/math.cpp (oracle/_ref), closed forms, quadrature, histograms.
 * in: n records of `in_stride` floats, out: n records of `out_stride` floats (strides per probe below).
 * flags: ORACLE_FLAG_LIBM selects glibc's transcendental functions. */
enum {
	ORACLE_PROBE_MAP_TO_UNIT_SPHERE = 0, /* fc:154-170  in (u, v)                      out xyz                                  2 -> 3 */
	ORACLE_PROBE_TNB_ROWS = 1,           /* fc:141-152,907  in n.xyz                    out rows t, n, bt                        3 -> 9 */
	ORACLE_PROBE_MAP_TO_TRIANGLE = 2,    /* fc:707-723  in (u, v, v0, v1, v2)           out xyz                                 11 -> 3 */
	ORACLE_PROBE_COSINE_HEMISPHERE = 3,  /* fc:172-175  in (u, v, n.xyz)                out normalize(map_to_unit_sphere + n)    5 -> 3 */
	ORACLE_PROBE_NDF = 4,                /* fc:505-514  in (wm.xyz, alpha)              out D(wm), wg = (0,1,0)                  4 -> 1 */
	ORACLE_PROBE_PICK_WI = 5,            /* fc:563-589  in (rng bits, wo.xyz, type, roughness)  out wi, wm, cos, pdf             6 -> 8 */
	ORACLE_PROBE_CALC_BRDF = 6,          /* fc:591-643  in (wo, wm, wi, diffuse rgb, F0 rgb, type, roughness, using_NEE)  out rgb 18 -> 3 */
	ORACLE_PROBE_OCT_DECODE = 7,         /* fc:203-210  in (packed 2x fp16 as float bits)  out the f16vec3 normal widened        1 -> 3 */
	ORACLE_PROBE_INVERSE3 = 8,           /* inverse(mat3), fc:908  in 9 (rows)          out 9 (rows)                             9 -> 9 */
	ORACLE_PROBE_PRIMARY_RAY = 9         /* fc:1088-1107,1120  in (camera 12, W, H, px, py, frame seed as float bits)  out ro, rd, rng state bits after the jitter draws  17 -> 7 */
};
int oracle_probe(int what, const float *in, uint32_t n, float *out, uint32_t flags);

/* Mean of n_samples evaluations of the selected estimator (flags: ORACLE_FLAG_EST_*) for ONE fixed camera ray, sample i
 * using the PCG stream seed3(i, 0, seed) -- a Monte-Carlo radiance estimate that closed forms can be checked against
 * without a camera in between.  out_mean: rgb as doubles. */
int oracle_estimate(const oracle_scene *sc, const float ro[3], const float rd[3], uint32_t n_samples, uint32_t bounces,
                    uint32_t flags, uint32_t seed, double out_mean[3]);

uint16_t oracle_f2h(float f);
float oracle_h2f(uint16_t h);
/* seed3 + n pcg() draws for pixel (x,y) and frame seed: framebuffer.comp:111-128 */
void oracle_pcg_stream(uint32_t x, uint32_t y, uint32_t seed, uint32_t n, uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif
