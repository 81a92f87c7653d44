// k_trace: closest-hit traversal, intersect() of shaders/framebuffer.comp:423-456, as a persistent
// wavefront kernel.
//
//  * one ray per lane; a wave keeps pulling rays from a global work counter and REFILLS idle lanes
//    (ballot + mbcnt prefix, one atomic per wave per refill) so that long rays do not strand a wave;
//  * traversal stack: PT_SHORT_STACK entries per lane in LDS, laid out [level][lane] so that a
//    wave-wide push/pop is bank-conflict free whatever the per-lane depth; deeper pushes spill to a
//    per-wave HBM tail (counted in pt_counters.stack_spills; the reference has 16 slots and no check);
//  * node pairs are siblings: 64 contiguous, 64-byte aligned bytes -> 4 x dwordx4 per lane per step;
//    a triangle test reads 48 of the 64 bytes (the packed normals / material are only read by the
//    shading kernel for the one accepted hit);
//  * arithmetic order follows the shader exactly -- the hit index is part of the bit-exact contract.
#pragma once
#include "device_math.h"

#define PT_SHORT_STACK 16
#define PT_SPILL_STACK 48   // 16 + 48 = 64 = the builder's max_depth (top_down_builder.hpp:36)
#define PT_TRACE_BLOCK 256
#define PT_TRACE_WAVES (PT_TRACE_BLOCK / 64)
// Work distribution: the ray range is cut into 64-ray units; the units are dealt to PT_SHARDS contiguous shards,
// each with its own counter on its own 128-byte line.  A wave pulls one unit at a time from its home shard and
// walks on to the next shard when one runs dry, so no single word sees more than total/(64*PT_SHARDS) atomics
// (one word saturates at ~88 atomics/us on this chip; a single counter made the first version atomic-bound).
#define PT_SHARDS 64
#define PT_SHARD_STRIDE 32

struct DevScene {
	const float4 *nodes;   // node i at nodes[2*i]; allocation is shifted so pairs (2k+1, 2k+2) are 64-B aligned
	const float4 *tris;    // triangle i at tris[4*i]
	const float4 *mats;    // material i at mats[3*i]
	const float4 *spheres; // sphere i at spheres[2*i]
	const uint32_t *light_tris;
	const uint32_t *light_spheres;
	uint32_t n_nodes, n_tris, n_mats, n_spheres, n_light_tris, n_light_spheres;
	const uint8_t *env; uint32_t env_w, env_h;
	const uint8_t *tex; uint32_t tex_layers;
};

// device control block shared by the kernels of one context
struct DevCtl {
	uint32_t qcount[2];   // live entries of the two path queues
	uint32_t work;        // unused (kept for layout)
	uint32_t pad;
	unsigned long long rays_primary, rays_shadow, rays_bounce, node_steps, tri_tests, stack_spills, paths, pad2;
	uint32_t shard[PT_SHARDS * PT_SHARD_STRIDE]; // trace work counters, zeroed by the kernel before each trace launch
};

// One segment of rays: structure-of-arrays with a common stride.
//   ray[c*stride + i], c = 0..5 : ro.xyz, rd.xyz        hit[c*stride + i], c = 0..3 : prim, tk, u, v
struct RaySeg {
	const float *ray;
	uint32_t *hit;
	const uint32_t *valid; // optional: bit0 of valid[i] says the slot holds a ray (shadow segment)
};

struct TraceArgs {
	DevScene sc;
	RaySeg seg[2];
	uint32_t stride;
	uint32_t n_fixed;          // ray count per segment when count_ptr == nullptr
	const uint32_t *count_ptr; // else read on the device (queue length written by the previous kernel)
	uint32_t n_segs;
	uint32_t refill_threshold; // refill when at least this many lanes of the wave are idle
	DevCtl *ctl;
	uint32_t *zero_on_entry;   // optional word cleared by block 0 (the queue the next shade kernel fills)
	uint32_t *spill;           // [gridDim.x * PT_TRACE_WAVES][PT_SPILL_STACK][64]
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

template <bool COUNT>
__global__ __launch_bounds__(PT_TRACE_BLOCK) void k_trace(TraceArgs a)
{
	__shared__ uint32_t s_stack[PT_TRACE_WAVES][PT_SHORT_STACK][64];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t *const lds = &s_stack[wave][0][lane];
	uint32_t *const spill = a.spill + ((size_t) (blockIdx.x * PT_TRACE_WAVES + wave) * PT_SPILL_STACK) * 64 + lane;

	const uint32_t n_per_seg = a.count_ptr ? *a.count_ptr : a.n_fixed;
	const uint32_t total = n_per_seg * a.n_segs;
	if (blockIdx.x == 0 && threadIdx.x == 0 && a.zero_on_entry) *a.zero_on_entry = 0;

	const float4 *__restrict__ nodes = a.sc.nodes;
	const float4 *__restrict__ tris = a.sc.tris;

	// per-lane ray state
	bool active = false;
	uint32_t slot = 0, seg = 0;
	v3 ro = V(0, 0, 0), rd = V(0, 0, 0), inv = V(0, 0, 0);
	float tmax = PT_TMAX;
	uint32_t cur = 1, sp = 0;
	uint32_t best_prim = 0xffffffffu, best_tk = 0;
	float best_u = 0.0f, best_v = 0.0f;
	uint32_t n_steps = 0, n_tests = 0, n_spills = 0;
	bool exhausted = (total == 0);
	// wave-uniform work state
	uint32_t loc_next = 0, loc_end = 0; // the unit this wave is handing out to its lanes
	const uint32_t total_units = (total + 63u) >> 6;
	const uint32_t units_per_shard = (total_units + PT_SHARDS - 1u) / PT_SHARDS;
	uint32_t shard = (blockIdx.x * PT_TRACE_WAVES + wave) % PT_SHARDS, shards_done = 0;

	for (;;) {
		// ------------------------------------------------------------------ refill idle lanes
		const unsigned long long idle = __ballot(!active);
		const uint32_t n_idle = (uint32_t) __popcll(idle);
		if (!exhausted && n_idle > 0u && (n_idle == 64u || n_idle >= a.refill_threshold)) {
			if (loc_next == loc_end) {
				bool found = false;
				while (shards_done < PT_SHARDS) {
					uint32_t u = 0;
					if (lane == 0) u = atomicAdd(&a.ctl->shard[shard * PT_SHARD_STRIDE], 1u);
					u = (uint32_t) __builtin_amdgcn_readfirstlane((int) u);
					const uint32_t unit = shard * units_per_shard + u;
					if (u < units_per_shard && unit < total_units) {
						loc_next = unit << 6;
						loc_end = loc_next + 64u < total ? loc_next + 64u : total;
						found = true;
						break;
					}
					shard = shard + 1u == PT_SHARDS ? 0u : shard + 1u; // this shard is dry for good: never revisited
					shards_done++;
				}
				if (!found) exhausted = true;
			}
			const uint32_t avail = loc_end - loc_next;
			const uint32_t take = n_idle < avail ? n_idle : avail;
			const uint32_t base = loc_next;
			loc_next += take;
			if (!active) {
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) idle, 0u));
				const uint32_t w = base + rank;
				if (rank < take) {
					seg = w >= n_per_seg ? 1u : 0u;
					slot = w - seg * n_per_seg;
					const RaySeg &rs = a.seg[seg];
					bool ok = rs.valid ? (rs.valid[slot] & 1u) != 0 : true;
					if (ok) {
						const float *r = rs.ray + slot;
						const uint32_t st = a.stride;
						// fc:437 -- traversal sees the ray rounded to fp16 and widened again
						ro = V(q16(r[0]), q16(r[st]), q16(r[2 * st]));
						rd = V(q16(r[3 * st]), q16(r[4 * st]), q16(r[5 * st]));
						inv = V(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z); // fc:347
						tmax = PT_TMAX;
						cur = 1; sp = 0;
						best_prim = 0xffffffffu; best_tk = 0; best_u = 0.0f; best_v = 0.0f;
						// a root that is a leaf has no children to start from: spheres only
						cur = a.sc.n_nodes >= 3 ? 1u : 0xfffffffeu;
						active = true;
					}
				}
			}
		}
		const unsigned long long act0 = __ballot(active);
		if (act0 == 0ull) {
			if (exhausted) break;
			continue;
		}

		// ------------------------------------------------------------------ traverse until enough lanes go idle
		for (;;) {
			if (active) {
				bool done = false;
				if (cur == 0xfffffffeu) {
					done = true;
				} else {
					// one iteration of while(true), fc:352-418
					const float4 *p = nodes + 2u * (size_t) cur;
					const float4 l_lo = p[0], l_hi = p[1], r_lo = p[2], r_hi = p[3];
					if (COUNT) n_steps++;
					float le, lx, re, rx;
					box_pair_test(l_lo, l_hi, ro, inv, tmax, &le, &lx);
					box_pair_test(r_lo, r_hi, ro, inv, tmax, &re, &rx);
					bool hit_l = le <= lx, hit_r = re <= rx;
					const bool leaf_l = l_hi.w > 0.0f, leaf_r = r_hi.w > 0.0f;
					const uint32_t n_tris = (uint32_t) (hit_l ? l_hi.w : 0.0f) + (uint32_t) (hit_r ? r_hi.w : 0.0f);
					if (n_tris > 0) {
						const uint32_t first = (uint32_t) ((hit_l && leaf_l) ? l_lo.w : r_lo.w);
						for (uint32_t i = 0; i < n_tris; i++) {
							// intersect_triangle, fc:271-292 (attributes are resolved by the shading kernel)
							const uint32_t ti = first + i;
							if (COUNT) n_tests++;
							if (ti >= a.sc.n_tris) continue;
							const float4 *tp = tris + 4u * (size_t) ti;
							const float4 d1 = tp[0], d2 = tp[1], d3 = tp[2];
							const v3 v0 = V(d1.x, d1.y, d1.z);
							const v3 e1 = V(d2.x, d2.y, d2.z) - v0;
							const v3 e2 = V(d3.x, d3.y, d3.z) - v0;
							const v3 pv = cross(rd, e2);
							const float det = dot(e1, pv);
							const float inv_det = 1.0f / det;
							const v3 tv = ro - v0;
							const float u = dot(tv, pv) * inv_det;
							if (__builtin_fabsf(det) < PT_EPSILON || u < 0.0f || u > 1.0f) continue;
							const v3 qv = cross(tv, e1);
							const float v = inv_det * dot(rd, qv);
							const float t = inv_det * dot(e2, qv);
							if (v >= 0.0f && u + v <= 1.0f && t > PT_TMIN && t < tmax) {
								const h16 th = to_h16(t);
								tmax = (float) th;           // fc:375: tmax = local_data.t (float16_t)
								best_prim = ti;
								best_tk = h_bits(th) | (1u << 16);
								best_u = u; best_v = v;
							}
						}
						hit_l = hit_l && !leaf_l;
						hit_r = hit_r && !leaf_r;
					}
					if (hit_l) {
						if (hit_r) {
							uint32_t first = (uint32_t) l_lo.w, second = (uint32_t) r_lo.w;
							if (le > re) { const uint32_t tmp = first; first = second; second = tmp; }
							if (sp < PT_SHORT_STACK) lds[sp * 64u] = second;
							else if (sp < PT_SHORT_STACK + PT_SPILL_STACK) { spill[(sp - PT_SHORT_STACK) * 64u] = second; n_spills++; }
							sp++;
							cur = first;
						} else {
							cur = (uint32_t) l_lo.w;
						}
					} else if (hit_r) {
						cur = (uint32_t) r_lo.w;
					} else {
						if (sp == 0) done = true;
						else {
							sp--;
							cur = sp < PT_SHORT_STACK ? lds[sp * 64u] : spill[(sp - PT_SHORT_STACK) * 64u];
						}
					}
					if (!done && cur + 1u >= a.sc.n_nodes) done = true; // corrupt index: stop rather than fault
				}
				if (done) {
					// sphere loop, fc:439-453, with the full-precision ray
					if (a.sc.n_spheres) {
						const float *r = a.seg[seg].ray + slot;
						const uint32_t st = a.stride;
						const v3 fro = V(r[0], r[st], r[2 * st]);
						const v3 frd = V(r[3 * st], r[4 * st], r[5 * st]);
						const uint32_t ns = a.sc.n_spheres & 0xffffu;
						for (uint32_t i = 0; i < ns; i++) {
							const float4 s = a.sc.spheres[2u * i];
							const h16 th = to_h16(sphere_t(fro, frd, s));
							const float ct = (float) th;
							if (ct >= PT_TMIN && ct <= tmax) {
								tmax = ct;
								best_prim = i;
								best_tk = h_bits(th) | (2u << 16);
							}
						}
					}
					uint32_t *h = a.seg[seg].hit + slot;
					const uint32_t st = a.stride;
					h[0] = best_prim;
					h[st] = best_tk;
					h[2 * st] = __float_as_uint(best_u);
					h[3 * st] = __float_as_uint(best_v);
					active = false;
				}
			}
			const unsigned long long act = __ballot(active);
			if (act == 0ull) break;
			if (!exhausted && (64u - (uint32_t) __popcll(act)) >= a.refill_threshold) break;
		}
	}

	if (COUNT) {
		// wave-level reduction, one atomic per wave
		for (int o = 32; o > 0; o >>= 1) {
			n_steps += __shfl_down(n_steps, o);
			n_tests += __shfl_down(n_tests, o);
		}
		if (lane == 0) {
			atomicAdd(&a.ctl->node_steps, (unsigned long long) n_steps);
			atomicAdd(&a.ctl->tri_tests, (unsigned long long) n_tests);
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
