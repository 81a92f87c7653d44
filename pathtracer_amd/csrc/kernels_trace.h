// k_trace: closest-hit traversal, intersect() of shaders/framebuffer.comp:423-456, as a persistent
// wavefront kernel.
//
//  * one ray per lane; a wave keeps pulling 64-ray units from sharded work counters and REFILLS idle lanes
//    (ballot + mbcnt prefix) so that long rays do not strand a wave;
//  * a lane alternates between node steps and single-triangle tests; both fetch 64 aligned bytes, so every wave
//    iteration is one memory round trip in which every active lane makes progress (see k_trace below);
//  * traversal stack: PT_SHORT_STACK entries per lane in LDS, laid out [level][lane] so that a
//    wave-wide push/pop is bank-conflict free whatever the per-lane depth, between a sentinel level that reads "done"
//    and a dump level that takes the pushes beyond; those spill to a per-wave HBM tail (counted in
//    pt_counters.stack_spills; the reference has 16 slots and no check);
//  * a finished lane leaves the traversal loop with its result in registers; the hit records of all finished lanes
//    are written by one store in front of the next refill;
//  * the kernel walks its own copy of the BVH, built at upload ("traversal blob", below): one 64-byte record per
//    inner node with both children's bounds and integer links, and 64-byte triangle records with the edges already
//    subtracted -> 4 x dwordx4 per lane per step from ONE base pointer with 32-bit offsets;
//    the packed normals / material of a triangle are only used by the shading kernel for the one accepted hit;
//  * arithmetic order follows the shader exactly -- the hit index is part of the bit-exact contract.
#pragma once
#include "device_math.h"

// streaming accesses to the path queues (each slot is read once and written once per stage)
// Measured (profiles/r02_nt_experiment.txt): the `nt` forms bypass L1 only and do not keep the blob in L2 any better --
// k_trace 46.2 -> 47.1 ms, k_shade 23.4 -> 23.7 ms per 64 frames -- so plain accesses stay the default.
#ifndef PT_NT
#define PT_NT 0
#endif
typedef uint32_t pt_u32x4 __attribute__((ext_vector_type(4)));
#if PT_NT
#define PT_NT_LOAD(p) __builtin_nontemporal_load(p)
#define PT_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define PT_NT_LOAD(p) (*(p))
#define PT_NT_STORE(v, p) (*(p) = (v))
#endif

// The top of the tree lives in LDS: the first PT_TOP_BYTES of the traversal blob (build_traversal_blob puts the records
// with the largest boxes there) are copied into every block's LDS at kernel start, and a lane whose record lies in that
// range reads it with ds_read_b128 instead of global_load_dwordx4.  Without it the kernel sits on the vector-memory path's
// rate of divergent 16-byte lane accesses (tools/ubench_gather.hip: 0.69 clk per lane and instruction whether they hit L1
// or not); every ray passes through the top levels, so this takes 63 % of those accesses off that path (bench scene) and
// leaves the kernel bound by its own instruction stream (k_trace).
// 0 = off (every record from global memory).
#ifndef PT_TOP_BYTES
#define PT_TOP_BYTES 16384
#endif
#if PT_TOP_BYTES
#ifndef PT_SHORT_STACK
#define PT_SHORT_STACK 12
#endif
#ifndef PT_TRACE_BLOCK
#define PT_TRACE_BLOCK 1024 // 16 waves share one copy of the tree top: 16 * 3.5 KB of stacks + 16 KB = 72 KB, two blocks per CU
#endif
#define PT_TRACE_MIN_WAVES 8 // two such blocks per CU need 8 waves per SIMD: at most 64 VGPRs
#else
#define PT_SHORT_STACK 16
#define PT_TRACE_BLOCK 256
#define PT_TRACE_MIN_WAVES 7
#endif
#define PT_SPILL_STACK (64 - PT_SHORT_STACK) // together 64 = the builder's max_depth (top_down_builder.hpp:36)
#define PT_TRACE_WAVES (PT_TRACE_BLOCK / 64)
#define PT_TOP_RECORDS (PT_TOP_BYTES / 64)
// Work distribution.  A ray queue is cut into PT_SHARDS regions of `cap` slots (cap a multiple of 256); region r
// holds count[r] live entries at slots [r*cap, r*cap + count[r]).  Every region has its own fill counter and its own
// work-unit counter, each on a 128-byte line of its own, so no single word sees more than 1/PT_SHARDS of the
// atomics (one word saturates at ~88 returning atomics/us on this chip: a single queue tail made the first shading
// kernel atomic-bound at 32k waves per launch, a single work counter did the same to the trace kernel).
// A trace wave pulls 64-ray units from its home region and walks on to the next region when one runs dry.
#define PT_SHARDS 64
#define PT_SHARD_STRIDE 32   // u32 words between two counters = 128 bytes

#ifndef PT_MAX_BATCH
#define PT_MAX_BATCH 32   // frames (samples per pixel) traced together by one set of launches
#endif
#ifndef PT_MAX_PATHS_LOG2
#define PT_MAX_PATHS_LOG2 26 // ... and at most this many paths (2^26 = ~23 GB of queues); k_shade packs (path id << 4) | flags into one word: < 2^28
#endif

// Debug build only (-DPT_DEBUG_ITER_CAP, `make debug`: pathtracer_amd/libpt355_dbg.so).  k_trace ends because pt_upload_scene
// proved the tree to be a tree and because the loop below steps every active lane in every iteration -- nothing in the kernel
// itself bounds it, and a commit that moved the step calls behind an #else once produced a build that never returned
// (b882dcd).  With this macro every lane counts the iterations of its ray; a ray past PT_DEBUG_ITER_CAP x (records of the
// traversal blob) + 64 -- a tree walk visits every record at most once -- is abandoned, the count goes to DevCtl::iter_cap_hits,
// and launch_trace (which then waits for every launch) returns PT_EHIP with a message: a failing test instead of a hung GPU.
// -DPT_DEBUG_INJECT_STALL=n reproduces the fault for the test of the cap itself: a lane's first n iterations skip the step calls.
#ifdef PT_DEBUG_ITER_CAP
#if PT_DEBUG_ITER_CAP + 0 == 0
#undef PT_DEBUG_ITER_CAP
#define PT_DEBUG_ITER_CAP 4
#endif
#endif

// tile-major path -> pixel map shared with pt_partition_pixels(): path p of this rank is pixel
// (p % 64) of its (p / 64)-th own 8x8 tile; with PT_PARTITION_TILES tile t belongs to rank t % world.
PT_DEV uint32_t path_pixel(uint32_t p, uint32_t width, uint32_t partition, uint32_t rank, uint32_t world)
{
	const uint32_t tiles_x = width >> 3;
	uint32_t tile = p >> 6;
	if (partition == 0u && world > 1u) tile = tile * world + rank;
	const uint32_t in = p & 63u;
	const uint32_t x = (tile % tiles_x) * 8u + (in & 7u), y = (tile / tiles_x) * 8u + (in >> 3);
	return y * width + x;
}

// main() + render_function(), fc:1088-1122: seed, pixel jitter, pinhole ray of path p of a batch (frame slot
// p / n_paths_frame, path-in-frame p % n_paths_frame).  ONE function for every kernel that needs the camera ray -- the
// stand-alone k_raygen, and the fused primary stage where k_trace generates the ray at refill and k_shade generates it
// again instead of both reading it from HBM -- so all of them get the same bits.  A: any argument block with the
// members cam, width, height, n_paths_frame, partition, rank, world, seeds.
template <typename A>
PT_DEV void primary_ray(const A &a, uint32_t p, v3 *ro_out, v3 *rd_out, uint32_t *rng_out)
{
	const uint32_t fslot = p / a.n_paths_frame;
	const uint32_t pixel = path_pixel(p - fslot * a.n_paths_frame, a.width, a.partition, a.rank, a.world);
	const uint32_t px = pixel % a.width, py = pixel / a.width;
	uint32_t rng = seed3(px, py, a.seeds[fslot]);
	const v3 cam_origin = V(a.cam[0], a.cam[1], a.cam[2]);
	const v3 cam_forward = V(a.cam[4], a.cam[5], a.cam[6]);
	const v3 cam_right = V(a.cam[8], a.cam[9], a.cam[10]);
	const float sw = (float) a.width, sh = (float) a.height;
	const float grid_height = 2.0f;
	const float grid_width = grid_height * sw / sh;
	const v3 cam_up = normalize(cross(cam_right, cam_forward));
	const v3 grid_x = cam_right * grid_width;
	const v3 grid_y = cam_up * grid_height;
	v3 grid_origin = (cam_origin - (grid_x * 0.5f)) - (grid_y * 0.5f);
	grid_origin = grid_origin + 2.0f * cam_forward;
	const float ox = rnd(rng) - 0.5f;
	const float oy = rnd(rng) - 0.5f;
	const float u = ((float) (int) px + ox) / sw;
	const float v = ((float) (int) py + oy) / sh;
	const v3 point_on_grid = (grid_origin + u * grid_x) + v * grid_y;
	*ro_out = cam_origin;
	*rd_out = normalize(point_on_grid - cam_origin);
	*rng_out = rng;
}

struct DevScene {
	// Traversal blob (pt355.hip build_traversal_blob), addressed by 32-bit byte offsets:
	//   record of an inner node, 64 B: (lo0.xyz, link0) (hi0.xyz, bytes0) (lo1.xyz, link1) (hi1.xyz, bytes1) -- the bounds of
	//     its two children as uploaded; bytesK = 64 * triangle count of child K (0 = inner); linkK = offset of the child's
	//     own record, or of its first triangle record.  The record of node 0 (the pair 1, 2) is at offset 0.
	//   triangle record, 64 B: (v0.xyz, index) (e1.xyz, -) (e2.xyz, -) (-), laid out as the shader's merged leaf loop walks
	//     them (fc:356-368, quirk 5): a left leaf's n_l triangles are followed by the next n_r triangles of the array, so
	//     the two hit leaves of one step are ONE pending range.
	const char *tblob;
	uint32_t n_records;    // inner nodes; 0 = nothing to traverse (a root that is a leaf)
	uint32_t tblob_bytes;
	uint32_t short_rcp;    // 1: every |det| the triangle test can see is below 2^126, where v_rcp_f32 + one Newton step IS 1/det: the host launches k_trace<.., IEEE_RCP = false>
	const float4 *tris;    // triangle i at tris[4*i] (shading)
	const float4 *mats;    // material i at mats[3*i]
	const float4 *spheres; // sphere i at spheres[2*i]
	const uint32_t *light_tris;
	const uint32_t *light_spheres;
	uint32_t n_nodes, n_tris, n_mats, n_spheres, n_light_tris, n_light_spheres;
	const uint8_t *env; uint32_t env_w, env_h;
	const uint8_t *tex; uint32_t tex_layers;
	uint32_t tex_wrap_s, tex_wrap_t, tex_mag_filter; // GL enums, 0 = GL_REPEAT / GL_LINEAR
};

// device control block shared by the kernels of one context
struct DevCtl {
	unsigned long long rays_primary, node_steps, tri_tests, stack_spills, paths, wave_iters, top_steps;
	unsigned long long iter_cap_hits; // PT_DEBUG_ITER_CAP builds: rays abandoned because their traversal did not end
	unsigned long long pad[8];
	uint32_t qcount[2][PT_SHARDS * PT_SHARD_STRIDE];             // live entries of region r of queue q at [q][r*32]
	uint32_t shard[PT_SHARDS * PT_SHARD_STRIDE];                 // trace work-unit counters, zeroed before each trace launch
	unsigned long long rays_bounce[PT_SHARDS * (PT_SHARD_STRIDE / 2)]; // per-region statistics, summed by the host
	unsigned long long rays_shadow[PT_SHARDS * (PT_SHARD_STRIDE / 2)];
};

// One segment of rays.  Rays are structure-of-arrays with a common stride: ray[c*stride + i], c = 0..5 = ro.xyz,
// rd.xyz (wave refills read runs of consecutive slots: coalesced).  A hit is ONE 16-byte record per slot,
// hit[i] = (prim, t_h | kind<<16, u, v) for a triangle (kind 1), (sphere, t_h | 2<<16, HitData.uvs as 2 x fp16 bits, 0)
// for a sphere: lanes finish at different times, so a hit is written by a lone lane and
// one dwordx4 store touches one 32-byte sector where four column stores touched four.
struct RaySeg {
	const float *ray;
	uint4 *hit;
	const uint32_t *valid; // optional: bit0 of valid[i] says the slot holds a ray (shadow segment)
	const float *target;   // optional (shadow segment): target[c*stride + i], c = 0..2 = the light sample y_nee the ray aims at; such a
	                       // segment's ray columns hold the origin only, the direction is normalize(target - origin)
};

struct TraceArgs {
	DevScene sc;
	RaySeg seg[2];
	uint32_t stride;
	uint32_t cap;              // slots per region
	uint32_t n_fixed;          // total ray count when count_ptr == nullptr: region r holds clamp(n_fixed - r*cap, 0, cap)
	const uint32_t *count_ptr; // else count of region r at count_ptr[r * PT_SHARD_STRIDE] (written by the previous kernel)
	uint32_t n_segs;
	uint32_t refill_threshold; // refill when at least this many lanes of the wave are idle
	uint32_t shadow_mode;      // PT_SHADOW_* for rays that carry a target
	uint32_t peek;             // small launches: look at a region's unit counter before spending an atomic on it (see the unit fetch)
	uint32_t unit_shift;       // a wave claims 1 << unit_shift rays (64 .. 512) with one atomic on its region's work counter
	DevCtl *ctl;
	uint32_t *zero_on_entry;   // optional: region counters of the queue the next shading kernel fills, cleared by block 0
	uint32_t *spill;           // [gridDim.x * PT_TRACE_WAVES][PT_SPILL_STACK][64]
	// k_trace<.., GEN = true> only (the fused primary stage): slot s IS path s of the batch and its ray is generated here
	float cam[12];
	uint32_t width, height, n_paths_frame, partition, rank, world;
	uint32_t seeds[PT_MAX_BATCH];
};

PT_DEV void box_pair_test(float4 lo, float4 hi, v3 ro, v3 inv, float tmax, float *entry, float *exit_)
{
	// shaders/framebuffer.comp:320-327
	float t0x = (lo.x - ro.x) * inv.x, t0y = (lo.y - ro.y) * inv.y, t0z = (lo.z - ro.z) * inv.z;
	float t1x = (hi.x - ro.x) * inv.x, t1y = (hi.y - ro.y) * inv.y, t1z = (hi.z - ro.z) * inv.z;
	float nx = fmin_(t0x, t1x), ny = fmin_(t0y, t1y), nz = fmin_(t0z, t1z);
	float fx = fmax_(t0x, t1x), fy = fmax_(t0y, t1y), fz = fmax_(t0z, t1z);
	*entry = fmax_(nx, fmax_(ny, fmax_(nz, PT_TMIN)));
	*exit_ = fmin_(fx, fmin_(fy, fmin_(fz, tmax)));
}

// shaders/framebuffer.comp:226-268
PT_DEV float sphere_t(v3 ro, v3 rd, float4 s)
{
	v3 oc = ro - V(s.x, s.y, s.z);
	float a = dot(rd, rd);
	float b = 2.0f * dot(oc, rd);
	float c = dot(oc, oc) - s.w * s.w;
	float disc = b * b - 4.0f * a * c;
	if (disc >= 0.0f) {
		if (disc <= PT_EPSILON) {
			float t = -b / (2.0f * a);
			if (t >= PT_TMIN && t <= PT_TMAX) return t;
		} else {
			float sq = __builtin_sqrtf(disc);
			float t1 = (-b + sq) / (2.0f * a);
			float t2 = (-b - sq) / (2.0f * a);
			if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
			if (t1 >= PT_TMIN && t1 <= PT_TMAX) return t1;
			if (t2 >= PT_TMIN && t2 <= PT_TMAX) return t2;
		}
	}
	return -1.0f;
}

#define PT_CUR_DONE 0xffffffffu

// data.uvs of an accepted triangle hit (fc:303-306: float16_t arithmetic, left to right), as the two fp16 bit patterns
// uvx | uvy << 16.  One function for the shading kernels (every triangle hit) and for k_trace's flush (a sphere that wins
// over an accepted triangle: fc:447-451 writes t, normal, mat_index, object_index and object_type but NOT uvs, so the
// sphere's material is textured with the uvs of the last triangle intersect_bvh_stack accepted).
PT_DEV uint32_t tri_uv_bits(const DevScene &sc, uint32_t prim, float u, float v)
{
	const float4 *tp = sc.tris + 4u * (size_t) prim;
	const uint32_t p0 = __float_as_uint(tp[0].w), p1 = __float_as_uint(tp[1].w), p2 = __float_as_uint(tp[2].w);
	const float w = q16((1.0f - u) - v), uh = q16(u), vh = q16(v);
	const h16 ux = to_h16(q16(q16(w * (float) h_from_bits(p0 & 0xffffu)) + q16(uh * (float) h_from_bits(p1 & 0xffffu))) +
	                      q16(vh * (float) h_from_bits(p2 & 0xffffu)));
	const h16 uy = to_h16(q16(q16(w * (float) h_from_bits(p0 >> 16)) + q16(uh * (float) h_from_bits(p1 >> 16))) +
	                      q16(vh * (float) h_from_bits(p2 >> 16)));
	return h_bits(ux) | (h_bits(uy) << 16);
}

// Shadow rays (fc:970-974) are judged by the shading kernel from the CLOSEST hit: the light sample is visible iff
// every component of y_nee - (ro + rd * t_hit) is < 0.01 (one-sided).
#define PT_SHADOW_FULL   0u // traverse to the closest hit like the reference, whatever is found on the way
#define PT_SHADOW_EXACT  1u // stop at the first hit that already decides the compare (below): same image, less work
#define PT_SHADOW_ANYHIT 2u // also stop at any hit at least 0.02 in front of the light sample and call it occluded
                            // (kind 3): ignores the one-sided quirk, so the image differs where that quirk leaks light

// Largest fp16 t for which the compare is already lost on component k (rd_k > 0): y - (ro + rd*t) >= 0.01.
// fp32 multiply, add and subtract are monotone, so every t' <= t loses too: once a hit with t <= this bound is
// accepted, no closer hit (triangle or sphere) can make the sample visible.  -inf when there is no such t.
PT_DEV float shadow_kill_component(float y, float ro, float rd)
{
	if (!(rd > 0.0f)) return -__builtin_inff();
	float cand = (float) to_h16(((y - ro) - PT_FLOAT_COMPARE) * __builtin_amdgcn_rcpf(rd));
	// the quotient is only a guess (hence the 1-ulp hardware reciprocal): step down in fp16 until the shading kernel's own
	// expression confirms it
	for (int i = 0; i < 3; i++) {
		if ((y - (ro + rd * cand)) >= PT_FLOAT_COMPARE) return cand;
		cand = (float) h_from_bits(h_bits(to_h16(cand)) - (cand > 0.0f ? 1u : 0u));
		if (!(cand > 0.0f)) break;
	}
	return -__builtin_inff();
}

// Per-lane state machine.  A lane is in one of two modes each iteration:
//   N  fetch the record at `cur` (the two children of an inner node), run both slab tests, make the triangles of the
//      hit leaves the pending range [tri, tri_end) and pick the next `cur` (descend / push / pop) -- one iteration of
//      while(true), fc:352-418, minus its triangle loop;
//   T  test ONE pending triangle (fc:368-378).
// Both modes fetch 64 aligned bytes through the same four dwordx4 loads, so a wave iteration costs ONE memory round
// trip whatever mix of modes its lanes are in, and every active lane makes progress every iteration.  (The first
// version ran the triangle loop inside the node step: a lane with an 8-triangle leaf held 63 others through 8
// dependent round trips -- VALU lane utilisation 24 %, profiles/r01b.)  The order of operations seen by ONE ray is
// unchanged: the pending triangles are tested, in order, before the next pair's slab tests read tmax, and the
// descend/push/pop decision never depended on them (the stack holds record offsets only).
// cur, tri, tri_end and the stack entries are byte offsets into the traversal blob: the address of either mode is
// one select away, and no index is converted or range-checked here (pt_upload_scene proved every link).
template <bool COUNT, bool GEN, bool IEEE_RCP>
__global__ __launch_bounds__(PT_TRACE_BLOCK, PT_TRACE_MIN_WAVES) void k_trace(TraceArgs a)
{
	// level 0 of every lane's stack holds PT_CUR_DONE for good and sp starts at 1: popping the last entry needs no test for an
	// empty stack, the lane just reads "done".  Level PT_SHORT_STACK + 1 is a dump: a push beyond the LDS levels lands there
	// (and, for real, in the spill tail), so the common push is one unconditional ds_write instead of a two-way branch.
	__shared__ uint32_t s_stack[PT_TRACE_WAVES][PT_SHORT_STACK + 2][64];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t *const lds = &s_stack[wave][0][lane];
	lds[0] = PT_CUR_DONE;
	uint32_t *const spill = a.spill + ((size_t) (blockIdx.x * PT_TRACE_WAVES + wave) * PT_SPILL_STACK) * 64 + lane;

	if (blockIdx.x == 0 && threadIdx.x < PT_SHARDS && a.zero_on_entry) a.zero_on_entry[threadIdx.x * PT_SHARD_STRIDE] = 0;

	const char *__restrict__ blob = a.sc.tblob;
#if PT_TOP_BYTES
	__shared__ float4 s_top[PT_TOP_BYTES / 16];
	const uint32_t top_bytes = a.sc.tblob_bytes < (uint32_t) PT_TOP_BYTES ? a.sc.tblob_bytes : (uint32_t) PT_TOP_BYTES; // a multiple of 64
	// LDS image: the four 16-byte pieces of a record in four planes ([piece][record]), so that one ds_read_b128 of the wave
	// (piece j of 64 different records, 16 B apart) spreads over all 64 banks; kept as 64-byte records it would touch only
	// the 16 banks of its piece
	for (uint32_t i = threadIdx.x; i < top_bytes / 16u; i += PT_TRACE_BLOCK) s_top[(i & 3u) * PT_TOP_RECORDS + (i >> 2)] = ((const float4 *) blob)[i];
	__syncthreads();
#endif
	const bool have_tree = a.sc.n_records != 0u;

	// per-lane ray state
	uint4 *hitp = nullptr;
	v3 ro = V(0, 0, 0), rd = V(0, 0, 0), inv = V(0, 0, 0);
	float tmax = PT_TMAX;
	uint32_t cur = PT_CUR_DONE, sp = 0, tri = 0, tri_end = 0;
	uint32_t best_prim = 0xffffffffu, best_tk = 0;
	float best_u = 0.0f, best_v = 0.0f;
	uint32_t sph_tk = 0, sph_prim = 0; // closest analytic sphere, found at refill (see below)
	float t_kill = -__builtin_inff();  // shadow rays: an accepted hit with t <= t_kill ends the traversal
	uint32_t n_steps = 0, n_tests = 0, n_spills = 0, n_iters = 0, n_top = 0; // n_iters: iterations of the traversal loop (wave-uniform)
#ifdef PT_DEBUG_ITER_CAP
	uint32_t dbg_iters = 0; // iterations of this lane's current ray
	const uint32_t dbg_cap = (uint32_t) PT_DEBUG_ITER_CAP * (a.sc.tblob_bytes >> 6) + 64u;
#endif
	// wave-uniform work state
	bool exhausted = false;
	uint32_t loc_next = 0, loc_end = 0, loc_seg = 0; // the unit this wave is handing out to its lanes
	uint32_t shard = (blockIdx.x * PT_TRACE_WAVES + wave) % PT_SHARDS, shards_done = 0;
	uint32_t shard_count = 0, shard_units = 0;
	bool shard_known = false;
	// rays per work unit.  The returning atomic that claims a unit is a round trip of 2-3 us which the wave sits out, and a refill that
	// finds its unit short stays partial (lanes idle until the next one): at 64 rays a unit the two cost 6 % of a batch launch.  256 is
	// the measured optimum there (512 is slower again, with or without small units at a region's end: EXPERIMENTS.md section 6), 64 for
	// a launch of one small frame, where larger units would leave waves without work.  The host chooses (trace_launch_shape, pt355.hip).
	const uint32_t unit_rays = 1u << a.unit_shift;

	// a lane is busy while it has a record to visit or triangles pending; finished, refused and never-filled lanes all look alike
	auto is_active = [&]() { return tri < tri_end || cur != PT_CUR_DONE; };
	// One N step on the record (q0..q3) for this lane's ray: fc:352-418 minus the triangle loop (see k_trace's header).
	auto node_step = [&](const float4 q0, const float4 q1, const float4 q2, const float4 q3) {
		if (COUNT) n_steps++;
#ifdef PT_EXP_DUP_N
		{ // diagnostic only: both slab tests a second time on a perturbed ray, result kept alive but never used
			float a0, a1, a2, a3;
			const v3 ro2 = V(ro.x + 1.0f, ro.y, ro.z);
			box_pair_test(q0, q1, ro2, inv, tmax, &a0, &a1);
			box_pair_test(q2, q3, ro2, inv, tmax, &a2, &a3);
			if (a0 + a1 + a2 + a3 == 12345.678f) best_v = a0;
		}
#endif
		float le, lx, re, rx;
		box_pair_test(q0, q1, ro, inv, tmax, &le, &lx);
		box_pair_test(q2, q3, ro, inv, tmax, &re, &rx);
		const bool in_l = le <= lx, in_r = re <= rx;
		const uint32_t c_l = __float_as_uint(q0.w), c_r = __float_as_uint(q2.w); // child record, or first triangle of a leaf
		const uint32_t b_l = __float_as_uint(q1.w), b_r = __float_as_uint(q3.w); // 64 * triangles of a leaf, 0 = inner
		const bool leaf_l = b_l != 0u, leaf_r = b_r != 0u;
		// fc:364-382: the triangles of the hit leaves become the pending range (empty when no leaf was hit;
		// a lane in N mode has none pending, so nothing is overwritten)
		tri = (in_l && leaf_l) ? c_l : c_r;
		tri_end = tri + ((in_l ? b_l : 0u) + (in_r ? b_r : 0u));
		// fc:384-418, as selects: descend into the nearer hit inner child, push the farther, else pop
		const bool go_l = in_l && !leaf_l, go_r = in_r && !leaf_r;
		const bool both = go_l && go_r, any = go_l || go_r;
		const bool r_first = le > re; // tie -> left first
		const uint32_t nearer = r_first ? c_r : c_l, farther = r_first ? c_l : c_r;
		cur = both ? nearer : (go_l ? c_l : c_r);
		if (both) {
			lds[(sp <= PT_SHORT_STACK ? sp : PT_SHORT_STACK + 1u) * 64u] = farther;
			if (__builtin_expect(sp > PT_SHORT_STACK, 0)) { spill[(sp - (PT_SHORT_STACK + 1u)) * 64u] = farther; n_spills++; } // sp <= 64: pt_upload_scene bounds the depth
			sp++;
		}
		if (!any) {
			// sp >= 1 here: a lane that has popped the sentinel of level 0 retires in the same iteration
			sp--;
			cur = lds[(sp < PT_SHORT_STACK ? sp : PT_SHORT_STACK) * 64u];
			if (__builtin_expect(sp > PT_SHORT_STACK, 0)) cur = spill[(sp - (PT_SHORT_STACK + 1u)) * 64u]; // rare: keeps the common pop a plain ds_read
		}
	};
	// One T step: intersect_triangle, fc:271-292, on the triangle record (q0..q2) -- the attributes are resolved by the shading kernel.
	auto tri_step = [&](const float4 q0, const float4 q1, const float4 q2) {
#ifdef PT_EXP_DUP_T
		{ // diagnostic only: the triangle arithmetic a second time on a perturbed ray, result kept alive but never used
			const v3 rd2 = V(rd.x + 1.0f, rd.y, rd.z), ro2 = V(ro.x, ro.y + 1.0f, ro.z);
			const v3 xv0 = V(q0.x, q0.y, q0.z), xe1 = V(q1.x, q1.y, q1.z), xe2 = V(q2.x, q2.y, q2.z);
			const v3 xpv = cross(rd2, xe2);
			const float xdet = dot(xe1, xpv), xinv = 1.0f / xdet;
			const v3 xtv = ro2 - xv0;
			const float xu = dot(xtv, xpv) * xinv;
			const v3 xqv = cross(xtv, xe1);
			const float xv = xinv * dot(rd2, xqv), xt = xinv * dot(xe2, xqv);
			if (xu + xv + xt == 12345.678f) best_u = xt;
		}
#endif
		// intersect_triangle, fc:271-292 (attributes are resolved by the shading kernel)
		if (COUNT) n_tests++;
		tri += 64u;
		const v3 v0 = V(q0.x, q0.y, q0.z);
		const v3 e1 = V(q1.x, q1.y, q1.z);
		const v3 e2 = V(q2.x, q2.y, q2.z);
		const v3 pv = cross(rd, e2);
		const float det = dot(e1, pv);
		// 1/det.  v_rcp_f32 + one FMA Newton step (3 instructions for the 12 of the correctly rounded division) returns
		// the SAME bits as the IEEE division for every normal |x| < 2^126 -- all 4.26e9 of them checked
		// (tools/verify_rcp.hip, profiles/r01g_verify_rcp.txt); a smaller |det| is rejected below whatever its
		// reciprocal, and pt_upload_scene sets short_rcp only when no triangle's edges can produce a larger one.
		float inv_det;
		if (!IEEE_RCP) {
			const float r = __builtin_amdgcn_rcpf(det);
			inv_det = __builtin_fmaf(__builtin_fmaf(-det, r, 1.0f), r, r);
		} else inv_det = 1.0f / det;
		const v3 tv = ro - v0;
		const float u = dot(tv, pv) * inv_det;
		if (!(__builtin_fabsf(det) < PT_EPSILON || u < 0.0f || u > 1.0f)) {
			const v3 qv = cross(tv, e1);
			const float v = inv_det * dot(rd, qv);
			const float t = inv_det * dot(e2, qv);
			if (v >= 0.0f && u + v <= 1.0f && t > PT_TMIN && t < tmax) {
				const h16 th = to_h16(t);
				tmax = (float) th;           // fc:375: tmax = local_data.t (float16_t)
				best_prim = __float_as_uint(q0.w);
				best_tk = h_bits(th) | (1u << 16);
				best_u = u; best_v = v;
				if (tmax <= t_kill) { // shadow ray decided: nothing closer can change the verdict
					tri = tri_end; cur = PT_CUR_DONE; sph_tk = 0u;
					if (a.shadow_mode == PT_SHADOW_ANYHIT) best_tk = (best_tk & 0xffffu) | (3u << 16);
				}
			}
		}
	};
	// Write out the record of a lane whose ray has finished: the hoisted sphere result competes with the traversal's (fc:447:
	// TMIN <= t <= tmax).  NOT done inside the traversal loop: a finished lane just leaves it with its result in its registers
	// (nothing touches them while it idles) and the records of all of them go out in one store instruction before the next
	// refill.  Inside the loop the store sat in front of the next iteration's loads in the wave's one vector-memory counter, so
	// every s_waitcnt for a record fetch also waited for a write acknowledgement -- in almost every iteration, since about
	// three lanes of a wave finish per iteration.
	auto flush_finished = [&]() {
		if (hitp && !is_active()) {
			if (sph_tk != 0u && (float) h_from_bits(sph_tk & 0xffffu) <= tmax) {
				// fc:447-451 leaves result.uvs alone: a sphere's record carries the uvs of the last accepted triangle (+0 when
				// there was none: the oracle's zeroed `out` record) where a triangle's carries the barycentrics
				best_u = __uint_as_float(best_tk != 0u ? tri_uv_bits(a.sc, best_prim, best_u, best_v) : 0u);
				best_v = 0.0f;
				best_prim = sph_prim; best_tk = sph_tk;
			}
			pt_u32x4 rec;
			rec.x = best_prim; rec.y = best_tk; rec.z = __float_as_uint(best_u); rec.w = __float_as_uint(best_v);
			PT_NT_STORE(rec, (pt_u32x4 *) hitp);
			hitp = nullptr;
		}
	};

	for (;;) {
		// ------------------------------------------------------------------ refill idle lanes
		// The wave's work state is the same in every lane.  The compiler cannot see that through the loops around it and keeps it
		// in vector registers with lane masks for the flags; read back through v_readfirstlane it lives in scalar registers and
		// the unit bookkeeping below is scalar code (8 VGPRs fewer, no spill in the ray-generating variant, k_trace -3.7 %).
		exhausted = __builtin_amdgcn_readfirstlane((int) exhausted) != 0;
		shard_known = __builtin_amdgcn_readfirstlane((int) shard_known) != 0;
		loc_next = (uint32_t) __builtin_amdgcn_readfirstlane((int) loc_next);
		loc_end = (uint32_t) __builtin_amdgcn_readfirstlane((int) loc_end);
		loc_seg = (uint32_t) __builtin_amdgcn_readfirstlane((int) loc_seg);
		shard = (uint32_t) __builtin_amdgcn_readfirstlane((int) shard);
		shards_done = (uint32_t) __builtin_amdgcn_readfirstlane((int) shards_done);
		shard_count = (uint32_t) __builtin_amdgcn_readfirstlane((int) shard_count);
		shard_units = (uint32_t) __builtin_amdgcn_readfirstlane((int) shard_units);
		flush_finished();
		const unsigned long long idle = __ballot(!is_active());
		const uint32_t n_idle = (uint32_t) __popcll(idle);
		if (!exhausted && n_idle > 0u && (n_idle == 64u || n_idle >= a.refill_threshold)) {
			if (loc_next == loc_end) {
				bool found = false;
				while (shards_done < PT_SHARDS) {
					if (!shard_known) {
						uint32_t cnt;
						if (a.count_ptr) cnt = a.count_ptr[shard * PT_SHARD_STRIDE];
						else { const uint32_t lo = shard * a.cap; cnt = a.n_fixed > lo ? (a.n_fixed - lo < a.cap ? a.n_fixed - lo : a.cap) : 0u; }
						shard_count = (uint32_t) __builtin_amdgcn_readfirstlane((int) (cnt < a.cap ? cnt : a.cap));
						shard_units = (shard_count + unit_rays - 1u) >> a.unit_shift;
						shard_known = true;
					}
					// A launch of a frame or two is over before most waves have seen their second unit: towards its end every wave walks
					// the dry regions one returning atomic (a round trip to L2) at a time.  With a.peek a plain load of the counter
					// comes first and a region seen dry costs no atomic: the one-frame launch -16 %.  Off for batches, where the
					// extra round trip in front of EVERY unit costs more than the dry tail saves (EXPERIMENTS.md section 6).
					bool open = shard_units != 0u;
					if (open && a.peek)
						open = __hip_atomic_load(&a.ctl->shard[shard * PT_SHARD_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < shard_units * a.n_segs;
					if (open) {
						uint32_t u = 0;
						if (lane == 0) u = atomicAdd(&a.ctl->shard[shard * PT_SHARD_STRIDE], 1u);
						u = (uint32_t) __builtin_amdgcn_readfirstlane((int) u);
						if (u < shard_units * a.n_segs) {
							loc_seg = u >= shard_units ? 1u : 0u;
							const uint32_t idx = u - loc_seg * shard_units;
							const uint32_t base = shard * a.cap;
							loc_next = base + (idx << a.unit_shift);
							loc_end = loc_next + unit_rays < base + shard_count ? loc_next + unit_rays : base + shard_count;
							found = true;
							break;
						}
					}
					shard = shard + 1u == PT_SHARDS ? 0u : shard + 1u; // this region is dry for good: never revisited
					shards_done++;
					shard_known = false;
				}
				if (!found) exhausted = true;
			}
			const uint32_t avail = loc_end - loc_next;
			const uint32_t take = n_idle < avail ? n_idle : avail;
			const uint32_t base = loc_next;
			loc_next += take;
			if (!is_active()) {
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) idle, 0u));
				if (rank < take) {
					const uint32_t slot = base + rank;
					const RaySeg &rs = a.seg[loc_seg];
					hitp = nullptr; // stays so for a slot without a ray: nothing to write
					// queue columns are read once and streamed: non-temporal, so that they do not push the traversal blob out of
					// the XCD's 4 MiB L2 (PT_NT below; measured in profiles/r02_*)
					bool ok = (!GEN && rs.valid) ? (PT_NT_LOAD(&rs.valid[slot]) & 1u) != 0 : true;
					if (ok) {
						hitp = rs.hit + slot; // where this ray's record goes: kept per lane, so that writing it later needs no look at the segment table
						const uint32_t st = a.stride;
						v3 fro, frd, fy = V(0, 0, 0);
						if (GEN) {
							uint32_t rng_unused;
							primary_ray(a, slot, &fro, &frd, &rng_unused); // the camera ray of path `slot`: nothing to read
						} else {
							const float *r = rs.ray + slot;
							fro = V(PT_NT_LOAD(&r[0]), PT_NT_LOAD(&r[st]), PT_NT_LOAD(&r[2 * st]));
							if (rs.target) {
								// a shadow ray's direction is not in the queue: it is normalize(y_nee - x) (fc:959-960), the expression the
								// shading kernel used and will use again, from the two points that are (12 B less per shadow ray written
								// once and read twice)
								const float *tg = rs.target + slot;
								fy = V(PT_NT_LOAD(&tg[0]), PT_NT_LOAD(&tg[st]), PT_NT_LOAD(&tg[2 * st]));
								frd = normalize(fy - fro);
							} else frd = V(PT_NT_LOAD(&r[3 * st]), PT_NT_LOAD(&r[4 * st]), PT_NT_LOAD(&r[5 * st]));
						}
						// Sphere loop of fc:439-453, hoisted in front of the traversal while the full-precision ray is in
						// registers.  The loop only compares against tmax, so its outcome is: the LAST sphere attaining the
						// smallest fp16 t wins iff that t <= the traversal's final tmax -- evaluated when the lane retires.
						sph_tk = 0; sph_prim = 0;
						{
							float smax = PT_TMAX;
							const uint32_t ns = a.sc.n_spheres & 0xffffu; // uint16_t loop counter, fc:439
							for (uint32_t i = 0; i < ns; i++) {
								const float4 s = a.sc.spheres[2u * i];
								const h16 th = to_h16(sphere_t(fro, frd, s));
								const float ct = (float) th;
								if (ct >= PT_TMIN && ct <= smax) { smax = ct; sph_prim = i; sph_tk = h_bits(th) | (2u << 16); }
							}
						}
						// shadow rays: how early may the traversal stop?
						t_kill = -__builtin_inff();
						bool occluded_now = false;
						if (!GEN && rs.target && a.shadow_mode != PT_SHADOW_FULL) {
							const v3 y = fy;
							t_kill = fmax_(shadow_kill_component(y.x, fro.x, frd.x),
							               fmax_(shadow_kill_component(y.y, fro.y, frd.y), shadow_kill_component(y.z, fro.z, frd.z)));
							if (a.shadow_mode == PT_SHADOW_ANYHIT) {
								const v3 d = y - fro;
								const float t_any = (float) to_h16(__builtin_sqrtf(dot(d, d)) - 2.0f * PT_FLOAT_COMPARE);
								t_kill = fmax_(t_kill, t_any - 0.01f * t_any); // fp16 rounding of t_any may round up: stay clear of the light
							}
							// a sphere that close already decides it: the final hit can only be as close or closer
							occluded_now = sph_tk != 0u && (float) h_from_bits(sph_tk & 0xffffu) <= t_kill;
						}
						// fc:437 -- traversal sees the ray rounded to fp16 and widened again
						ro = q16(fro);
						rd = q16(frd);
						inv = V(rcp_of_q16(rd.x), rcp_of_q16(rd.y), rcp_of_q16(rd.z)); // fc:347: 1.0 / rd, bit for bit
						tmax = PT_TMAX;
						sp = 1; tri = 0; tri_end = 0;
#ifdef PT_DEBUG_ITER_CAP
						dbg_iters = 0;
#endif
						best_prim = 0xffffffffu; best_tk = 0; best_u = 0.0f; best_v = 0.0f;
						// a root that is a leaf has no children to start from: spheres only
						if (occluded_now) {
							if (a.shadow_mode == PT_SHADOW_ANYHIT) sph_tk = (sph_tk & 0xffffu) | (3u << 16);
							*hitp = make_uint4(sph_prim, sph_tk, 0u, 0u);
							hitp = nullptr;
						} else if (have_tree) {
							cur = 0u; // the root pair's record: the lane is active from here on
						} else {
							*hitp = make_uint4(sph_tk ? sph_prim : 0xffffffffu, sph_tk, 0u, 0u);
							hitp = nullptr;
						}
					}
				}
			}
		}
		if (__ballot(is_active()) == 0ull) {
			if (exhausted) break;
			continue;
		}

		// ------------------------------------------------------------------ traverse until enough lanes go idle
		{
			// every active lane has work: pending triangles (T) or a record to visit (N); a lane leaves as soon as it has neither
			// (Holding lanes with a pending triangle back until 4..48 of them had gathered, so that the triangle block
			// runs on a fuller wave, was measured 2-38 % slower in both a stalling and a non-stalling form, and again
			// 1.5-5 % slower at 4/8/16 lanes after the tree top moved to LDS -- DESIGN.md "What was tried".)
			uint32_t my_iters = 0;
			// leave for a refill once no more than this many lanes are still traversing (never, once the queue is dry); one scalar
			// for the whole loop: `exhausted` cannot change inside it
			const int live_min = __builtin_amdgcn_readfirstlane(exhausted ? -1 : 64 - (int) a.refill_threshold);
			if (is_active()) for (;;) {
				if (COUNT) my_iters++;
				const bool tmode = tri < tri_end;
				const uint32_t off = tmode ? tri : cur;
				float4 q0, q1, q2, q3;
#if PT_TOP_BYTES
				// two exec-masked load groups: global_load_dwordx4 for the lanes below the top, ds_read_b128 for the others.  The
				// empty asm keeps the two branches apart: merged into one select of a generic pointer they become flat_load
				// instructions, which take every lane through the vector-memory front end again.  (The compiler cannot know that the
				// groups write disjoint lanes of the same registers and waits for the global loads before it issues the LDS reads;
				// the same fetch written in asm with both groups in flight together measured +-0, DESIGN.md "What was tried".)
				if (off >= top_bytes) {
					const float4 *p = (const float4 *) (blob + off);
					q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
				} else {
					if (COUNT) n_top++;
					const float4 *p = (const float4 *) ((const char *) s_top + (off >> 2));
					q0 = p[0]; q1 = p[PT_TOP_RECORDS]; q2 = p[2 * PT_TOP_RECORDS]; q3 = p[3 * PT_TOP_RECORDS];
					asm volatile("" ::: "memory");
				}
#else
				{
					const float4 *p = (const float4 *) (blob + off);
					q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
				}
#endif
				// all four loads are consumed here, in front of the mode branch: one memory round trip per iteration
				// (left alone, the compiler sinks the fourth load into the node branch behind the first wait)
				asm volatile("" : "+v"(q0.x), "+v"(q1.w), "+v"(q2.x), "+v"(q3.x), "+v"(q3.w));
#ifdef PT_DEBUG_ITER_CAP
				if (++dbg_iters > dbg_cap) { // this ray's traversal does not end: give it up and say so (see the top of this file)
					atomicAdd(&a.ctl->iter_cap_hits, 1ull);
					tri = 0; tri_end = 0; cur = PT_CUR_DONE;
					break;
				}
#ifdef PT_DEBUG_INJECT_STALL
				if (dbg_iters <= (uint32_t) PT_DEBUG_INJECT_STALL) continue; // the fault of b882dcd, bounded: no step calls
#endif
#endif
				if (tmode) tri_step(q0, q1, q2);
				else node_step(q0, q1, q2, q3);
				// a retiring lane leaves the loop (its exec bit drops until the others follow); the rest go round again unless
				// enough of the wave has gone idle for a refill
				if (!is_active()) break;
				if ((int) __popcll(__ballot(true)) <= live_min) break;
			}
			if (COUNT) {
				for (int o = 32; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(my_iters, o); my_iters = other > my_iters ? other : my_iters; }
				n_iters += my_iters;
			}
		}
	}

	if (COUNT) {
		// wave-level reduction, one atomic per wave
		for (int o = 32; o > 0; o >>= 1) {
			n_steps += __shfl_down(n_steps, o);
			n_tests += __shfl_down(n_tests, o);
			n_top += __shfl_down(n_top, o);
		}
		if (lane == 0) {
			atomicAdd(&a.ctl->node_steps, (unsigned long long) n_steps);
			atomicAdd(&a.ctl->tri_tests, (unsigned long long) n_tests);
			atomicAdd(&a.ctl->wave_iters, (unsigned long long) n_iters);
			atomicAdd(&a.ctl->top_steps, (unsigned long long) n_top);
		}
	}
	{
		unsigned long long any_spill = __ballot(n_spills != 0);
		if (any_spill) {
			for (int o = 32; o > 0; o >>= 1) n_spills += __shfl_down(n_spills, o);
			if (lane == 0) atomicAdd(&a.ctl->stack_spills, (unsigned long long) n_spills);
		}
	}
}
