// GPU BVH build.  The reference builds its BVH on one CPU thread (CalculateBVH,
// src/scene/bvh.cpp:31-75: full-sweep SAH, O(N log N), 0.37 s per 200 k triangles); that build is reproduced bit for
// bit in host/bvh_build.cpp and stays the default.  This file is the set-up step for scenes where it dominates
// (millions of triangles): a linear BVH (Morton order + Karras' radix-tree construction, one thread per node), emitted
// in the reference's own flattened layout -- BVHNodeGLSL array, root at 0, the two children of a node adjacent, leaves
// as (first triangle, count) ranges of a re-ordered triangle array -- so that pt_upload_scene, the oracle and the
// reference's shader can consume it unchanged.  Not bit-identical to the Sweep-SAH tree and not meant to be: the
// bit-exactness contract covers the CPU builder; any valid tree yields the same closest hit up to ties of the
// fp16-quantised t.
//
// Stages (all on `stream`): centroid bounds -> 64-bit keys (39-bit extended Morton code << 24 | triangle index: unique,
// so the radix tree is at most 63 levels deep = inside the traversal stack; see k_keys) -> hipcub radix sort -> hierarchy -> bottom-up
// bounds (second arrival at a node merges its children) -> subtrees of <= max_leaf triangles collapse into leaves ->
// prefix sum numbers the surviving inner nodes -> nodes and sorted triangles are written.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "bvh_device.h"

namespace {

struct Box { float lo[3], hi[3]; };

__device__ inline int ordered(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ inline float unordered(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__device__ inline Box tri_box(const float4 *t)
{
	const float4 a = t[0], b = t[1], c = t[2];
	Box r;
	r.lo[0] = fminf(a.x, fminf(b.x, c.x)); r.hi[0] = fmaxf(a.x, fmaxf(b.x, c.x));
	r.lo[1] = fminf(a.y, fminf(b.y, c.y)); r.hi[1] = fmaxf(a.y, fmaxf(b.y, c.y));
	r.lo[2] = fminf(a.z, fminf(b.z, c.z)); r.hi[2] = fmaxf(a.z, fmaxf(b.z, c.z));
	return r;
}

// centroid bounds of the scene: cb[0..2] = min, cb[3..5] = max, as order-preserving ints
__global__ void k_bounds(const float4 *tris, uint32_t n, int *cb)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	float c[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, d[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
	if (i < n) {
		const Box b = tri_box(tris + 4u * (size_t) i);
		for (int k = 0; k < 3; k++) c[k] = d[k] = 0.5f * (b.lo[k] + b.hi[k]);
	}
	for (int k = 0; k < 3; k++) {
		for (int o = 32; o > 0; o >>= 1) {
			c[k] = fminf(c[k], __shfl_down(c[k], o));
			d[k] = fmaxf(d[k], __shfl_down(d[k], o));
		}
	}
	if ((threadIdx.x & 63u) == 0) {
		for (int k = 0; k < 3; k++) {
			atomicMin(&cb[k], ordered(c[k]));
			atomicMax(&cb[3 + k], ordered(d[k]));
		}
	}
}

// 13 bits -> every third bit of a 39-bit word
__device__ inline unsigned long long spread13(uint32_t v)
{
	unsigned long long x = v & 0x1fffu;
	x = (x | (x << 32)) & 0x001f00000000ffffull;
	x = (x | (x << 16)) & 0x001f0000ff0000ffull;
	x = (x | (x << 8)) & 0x100f00f00f00f00full;
	x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
	x = (x | (x << 2)) & 0x1249249249249249ull;
	return x;
}

// key = 39-bit code << 24 | triangle index: 63 bits, unique.  The code is the Morton code of the centroid (13 bits per
// axis of the centroid bounds), or, with `extended` (Vinkler et al. 2017), 10 bits per axis interleaved with 9 bits of the
// triangle's box diagonal relative to the scene's: large triangles then sort apart from the small ones around them
// instead of stretching the boxes of their Morton neighbours.
__global__ void k_keys(const float4 *tris, uint32_t n, const int *cb, unsigned long long *keys, int extended)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const Box b = tri_box(tris + 4u * (size_t) i);
	float t[3], ext2 = 0.0f, d2 = 0.0f;
	for (int k = 0; k < 3; k++) {
		const float lo = unordered(cb[k]), hi = unordered(cb[3 + k]);
		const float ext = hi - lo;
		t[k] = ext > 0.0f ? (0.5f * (b.lo[k] + b.hi[k]) - lo) / ext : 0.0f;
		t[k] = fminf(fmaxf(t[k], 0.0f), 1.0f);
		ext2 += ext * ext;
		d2 += (b.hi[k] - b.lo[k]) * (b.hi[k] - b.lo[k]);
	}
	unsigned long long m;
	if (!extended) {
		uint32_t q[3];
		for (int k = 0; k < 3; k++) q[k] = (uint32_t) fminf(t[k] * 8192.0f, 8191.0f);
		m = spread13(q[0]) | (spread13(q[1]) << 1) | (spread13(q[2]) << 2);
	} else {
		uint32_t q[4];
		for (int k = 0; k < 3; k++) q[k] = (uint32_t) fminf(t[k] * 1024.0f, 1023.0f);
		const float rel = ext2 > 0.0f ? sqrtf(d2 / ext2) : 0.0f; // centroid bounds are smaller than the scene: may exceed 1
		q[3] = (uint32_t) fminf(fminf(rel, 1.0f) * 512.0f, 511.0f);
		m = 0;
		int out = 38;
		for (int bit = 9; bit >= 0; bit--) { // x y z s | x y z s | ... | x y z
			for (int k = 0; k < 3; k++) m |= (unsigned long long) ((q[k] >> bit) & 1u) << out--;
			if (bit >= 1) m |= (unsigned long long) ((q[3] >> (bit - 1)) & 1u) << out--;
		}
	}
	keys[i] = (m << 24) | i;
}

__device__ inline int delta(const unsigned long long *keys, int n, int i, int j)
{
	if (j < 0 || j >= n) return -1;
	return __clzll((long long) (keys[i] ^ keys[j])); // keys are unique: never 64
}

// Karras 2012: internal node i of the radix tree over the sorted keys.  Children are stored with bit 31 = leaf.
__global__ void k_hierarchy(const unsigned long long *keys, int n, uint32_t *child, uint32_t *first, uint32_t *last,
                            uint32_t *parent_inner, uint32_t *parent_leaf)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n - 1) return;
	const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
	const int dmin = delta(keys, n, i, i - d);
	int lmax = 2;
	while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
	int l = 0;
	for (int t = lmax >> 1; t >= 1; t >>= 1)
		if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
	const int j = i + l * d;
	const int dnode = delta(keys, n, i, j);
	int s = 0;
	for (int t = l; t > 1;) {
		t = (t + 1) >> 1;
		if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
	}
	const int gamma = i + s * d + (d < 0 ? -1 : 0);
	const int lo = i < j ? i : j, hi = i < j ? j : i;
	const bool leaf_l = lo == gamma, leaf_r = hi == gamma + 1;
	child[2 * i] = (uint32_t) gamma | (leaf_l ? 0x80000000u : 0u);
	child[2 * i + 1] = (uint32_t) (gamma + 1) | (leaf_r ? 0x80000000u : 0u);
	first[i] = (uint32_t) lo;
	last[i] = (uint32_t) hi;
	if (leaf_l) parent_leaf[gamma] = (uint32_t) i; else parent_inner[gamma] = (uint32_t) i;
	if (leaf_r) parent_leaf[gamma + 1] = (uint32_t) i; else parent_inner[gamma + 1] = (uint32_t) i;
	if (i == 0) parent_inner[0] = 0xffffffffu;
}

__global__ void k_leaf_boxes(const float4 *tris, const unsigned long long *keys, uint32_t n, Box *lbox)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	lbox[j] = tri_box(tris + 4u * (size_t) ((uint32_t) keys[j] & 0xffffffu));
}

// One thread per leaf climbs towards the root; the SECOND thread to arrive at a node finds both children's boxes
// written (release fence before the counter, acquire fence after it), merges them and goes on; the first one stops.
// Every thread stops after at most 63 levels, so the grid drains whatever the timing.
__global__ void k_fit(const uint32_t *child, const uint32_t *parent_inner, const uint32_t *parent_leaf, const Box *lbox, Box *ibox,
                      uint32_t *arrived, uint32_t n)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	uint32_t node = parent_leaf[j];
	while (node != 0xffffffffu) {
		__threadfence();
		if (atomicAdd(&arrived[node], 1u) == 0u) return;
		__threadfence();
		Box b[2];
		for (int k = 0; k < 2; k++) {
			const uint32_t c = child[2 * node + k];
			const volatile Box *src = (c & 0x80000000u) ? &lbox[c & 0x7fffffffu] : &ibox[c];
			for (int a = 0; a < 3; a++) { b[k].lo[a] = src->lo[a]; b[k].hi[a] = src->hi[a]; }
		}
		Box u;
		for (int a = 0; a < 3; a++) { u.lo[a] = fminf(b[0].lo[a], b[1].lo[a]); u.hi[a] = fmaxf(b[0].hi[a], b[1].hi[a]); }
		ibox[node] = u;
		node = parent_inner[node];
	}
}

// an inner node survives unless its whole subtree fits one leaf (the root always survives: a root that is a leaf is
// never traversed by the shader)
__global__ void k_kept(const uint32_t *first, const uint32_t *last, uint32_t n_inner, uint32_t max_leaf, uint32_t *kept)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_inner) kept[i] = (i == 0u || last[i] - first[i] + 1u > max_leaf) ? 1u : 0u;
}

struct NodeOut { float bmin[3], first, bmax[3], count; }; // BVHNodeGLSL, src/scene/bvh.h:7-14

// leafpos: position of leaf j in the emitted triangle order (nullptr = j itself, the LBVH's Morton order)
__global__ void k_emit(const uint32_t *child, const uint32_t *first, const uint32_t *last, const uint32_t *kept, const uint32_t *slot,
                       const Box *lbox, const Box *ibox, uint32_t n_inner, NodeOut *nodes, const uint32_t *leafpos)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_inner || !kept[i]) return;
	if (i == 0u) {
		NodeOut r;
		for (int a = 0; a < 3; a++) { r.bmin[a] = ibox[0].lo[a]; r.bmax[a] = ibox[0].hi[a]; }
		r.first = 1.0f; r.count = 0.0f;
		nodes[0] = r;
	}
	for (uint32_t k = 0; k < 2; k++) {
		const uint32_t c = child[2 * i + k];
		const bool leaf = (c & 0x80000000u) != 0u;
		const uint32_t ci = c & 0x7fffffffu;
		const Box &b = leaf ? lbox[ci] : ibox[ci];
		NodeOut o;
		for (int a = 0; a < 3; a++) { o.bmin[a] = b.lo[a]; o.bmax[a] = b.hi[a]; }
		if (!leaf && kept[ci]) { o.first = (float) (1u + 2u * slot[ci]); o.count = 0.0f; }
		else if (leaf) { o.first = (float) (leafpos ? leafpos[ci] : ci); o.count = 1.0f; }
		else { o.first = (float) first[ci]; o.count = (float) (last[ci] - first[ci] + 1u); }
		nodes[1u + 2u * slot[i] + k] = o;
	}
}

__global__ void k_gather(const float4 *tris, const unsigned long long *keys, uint32_t n, float4 *sorted)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= 4u * n) return;
	sorted[t] = tris[4u * (size_t) ((uint32_t) keys[t >> 2] & 0xffffffu) + (t & 3u)];
}


// ---------------------------------------------------------------------------------------------------------------------
// PLOC (parallel locally-ordered clustering, Meister & Bittner 2017): the Morton-ordered triangles are the initial
// clusters; every pass each cluster looks PLOC_RADIUS neighbours to either side for the one whose union with it has
// the smallest area, mutual choices merge into a new inner node, and the survivors are compacted in order.  Bottom-up
// and area-driven where the radix tree only looks at key bits: on the bench mesh its trees take a third fewer traversal
// steps than the LBVH's and a tenth fewer than the reference's Sweep-SAH trees (profiles/r02f_lbvh_vs_ploc_vs_sah.txt).  The root is the last node made, so inner-node ids are
// handed out downwards from n - 2 and the root is node 0, as the emit stage expects.
#ifndef PLOC_RADIUS
#define PLOC_RADIUS 16
#endif

__device__ inline float half_area_union(const Box &a, const Box &b)
{
	const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
	const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
	const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
	return dx * dy + dy * dz + dz * dx;
}

__global__ void k_ploc_init(const Box *lbox, uint32_t n, Box *cbox, uint32_t *cnode)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	cbox[i] = lbox[i];
	cnode[i] = i | 0x80000000u;
}

__global__ void k_ploc_nn(const Box *cbox, uint32_t m, uint32_t *nn)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= m) return;
	const Box me = cbox[i];
	const uint32_t lo = i > PLOC_RADIUS ? i - PLOC_RADIUS : 0u, hi = i + PLOC_RADIUS < m ? i + PLOC_RADIUS : m - 1u;
	float best = __builtin_inff();
	uint32_t bj = i == 0u ? 1u : i - 1u; // a NaN area never wins: fall back to a neighbour so that pairs still form
	for (uint32_t j = lo; j <= hi; j++) {
		if (j == i) continue;
		const float a = half_area_union(me, cbox[j]);
		if (a < best) { best = a; bj = j; } // ties: the lower index, on both sides of a pair
	}
	nn[i] = bj;
}

// mutual nearest neighbours merge; the lower index keeps the slot
__global__ void k_ploc_merge(Box *cbox, uint32_t *cnode, const uint32_t *nn, uint32_t m, uint32_t n, uint32_t *counter, uint32_t *child,
                             Box *ibox, uint32_t *size, uint32_t *parent_inner, uint32_t *parent_leaf, uint32_t *valid)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= m) return;
	const uint32_t j = nn[i];
	if (nn[j] != i) { valid[i] = 1u; return; }
	if (i > j) { valid[i] = 0u; return; }
	const uint32_t id = (n - 2u) - atomicAdd(counter, 1u);
	const uint32_t ca = cnode[i], cb = cnode[j];
	const Box a = cbox[i], b = cbox[j];
	Box u;
	for (int k = 0; k < 3; k++) { u.lo[k] = fminf(a.lo[k], b.lo[k]); u.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
	child[2u * id] = ca; child[2u * id + 1u] = cb;
	ibox[id] = u;
	const uint32_t sa = (ca & 0x80000000u) ? 1u : size[ca], sb = (cb & 0x80000000u) ? 1u : size[cb];
	size[id] = sa + sb;
	if (ca & 0x80000000u) parent_leaf[ca & 0x7fffffffu] = id; else parent_inner[ca] = id;
	if (cb & 0x80000000u) parent_leaf[cb & 0x7fffffffu] = id; else parent_inner[cb] = id;
	cbox[i] = u;
	cnode[i] = id;
	valid[i] = 1u;
}

__global__ void k_ploc_compact(const Box *cbox, const uint32_t *cnode, const uint32_t *valid, const uint32_t *pos, uint32_t m, Box *cbox_out,
                               uint32_t *cnode_out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= m || !valid[i]) return;
	cbox_out[pos[i]] = cbox[i];
	cnode_out[pos[i]] = cnode[i];
}

// Where a node's triangles start in the emitted order: the sizes of the left siblings met on the way to the root.
// Thread t < n handles leaf t, thread n + t inner node t.  Also records the deepest leaf.
__global__ void k_ploc_ranges(const uint32_t *child, const uint32_t *size, const uint32_t *parent_inner, const uint32_t *parent_leaf, uint32_t n,
                              uint32_t *leafpos, uint32_t *first, uint32_t *last, uint32_t *max_depth)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= 2u * n - 1u) return;
	const bool leaf = t < n;
	uint32_t code = leaf ? (t | 0x80000000u) : (t - n);
	uint32_t node = leaf ? parent_leaf[t] : parent_inner[t - n];
	uint32_t off = 0, depth = 0;
	while (node != 0xffffffffu && depth < 4096u) {
		if (child[2u * node + 1u] == code) {
			const uint32_t l = child[2u * node];
			off += (l & 0x80000000u) ? 1u : size[l];
		}
		code = node;
		node = parent_inner[node];
		depth++;
	}
	if (leaf) { leafpos[t] = off; atomicMax(max_depth, depth); }
	else { first[t - n] = off; last[t - n] = off + size[t - n] - 1u; }
}

__global__ void k_scatter_tris(const float4 *tris, const unsigned long long *keys, const uint32_t *leafpos, uint32_t n, float4 *sorted)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= 4u * n) return;
	const uint32_t leaf = t >> 2;
	sorted[4u * (size_t) leafpos[leaf] + (t & 3u)] = tris[4u * (size_t) ((uint32_t) keys[leaf] & 0xffffffu) + (t & 3u)];
}

struct Scratch {
	void *p[40];
	int n = 0;
	~Scratch() { for (int i = 0; i < n; i++) (void) hipFree(p[i]); }
	template <typename T> hipError_t get(T **out, size_t bytes)
	{
		void *q = nullptr;
		hipError_t e = hipMalloc(&q, bytes ? bytes : 4);
		if (e == hipSuccess) p[n++] = q;
		*out = (T *) q;
		return e;
	}
};

struct Events {
	hipEvent_t e[2] = {nullptr, nullptr};
	~Events() { for (hipEvent_t x : e) if (x) (void) hipEventDestroy(x); }
};

} // namespace

#define B_TRY(call)                                                                                     \
	do {                                                                                                \
		hipError_t e_ = (call);                                                                         \
		if (e_ != hipSuccess) { snprintf(err, errlen, "%s: %s", #call, hipGetErrorString(e_)); return 2; } \
	} while (0)

int ptb_build_bvh(hipStream_t stream, int method, const void *tris_host, uint32_t n, uint32_t max_leaf, void *nodes_out, uint32_t *n_nodes_out,
                  void *sorted_out, float *build_ms, char *err, size_t errlen)
{
	if (!tris_host || !nodes_out || !n_nodes_out || !sorted_out) { snprintf(err, errlen, "pt_build_bvh_device: null argument"); return 1; }
	if (n < 2u || n > (1u << 24)) { snprintf(err, errlen, "pt_build_bvh_device: needs 2 .. 2^24 triangles (got %u)", n); return 1; }
	if (max_leaf < 1u || max_leaf > 16u) { snprintf(err, errlen, "pt_build_bvh_device: max_leaf must be 1 .. 16 (the reference builder's max_leaf_size)"); return 1; }
	Scratch s;
	const uint32_t ni = n - 1u;
	float4 *d_tris, *d_sorted;
	unsigned long long *d_keys, *d_keys2;
	int *d_cb;
	uint32_t *d_child, *d_first, *d_last, *d_pi, *d_pl, *d_arrived, *d_kept, *d_slot;
	Box *d_lbox, *d_ibox;
	NodeOut *d_nodes;
	B_TRY(s.get(&d_tris, (size_t) n * 64)); B_TRY(s.get(&d_sorted, (size_t) n * 64));
	B_TRY(s.get(&d_keys, (size_t) n * 8)); B_TRY(s.get(&d_keys2, (size_t) n * 8));
	B_TRY(s.get(&d_cb, 6 * sizeof(int)));
	B_TRY(s.get(&d_child, (size_t) ni * 8)); B_TRY(s.get(&d_first, (size_t) ni * 4)); B_TRY(s.get(&d_last, (size_t) ni * 4));
	B_TRY(s.get(&d_pi, (size_t) ni * 4)); B_TRY(s.get(&d_pl, (size_t) n * 4));
	B_TRY(s.get(&d_lbox, (size_t) n * sizeof(Box))); B_TRY(s.get(&d_ibox, (size_t) ni * sizeof(Box)));
	// arrived | kept | slot in one allocation (16 allocations is the Scratch limit)
	B_TRY(s.get(&d_arrived, (size_t) ni * 12 + 12));
	d_kept = d_arrived + ni; d_slot = d_kept + ni;
	B_TRY(s.get(&d_nodes, (size_t) (2u * ni + 1u) * sizeof(NodeOut)));

	B_TRY(hipMemcpyAsync(d_tris, tris_host, (size_t) n * 64, hipMemcpyHostToDevice, stream));
	Events ev;
	B_TRY(hipEventCreate(&ev.e[0])); B_TRY(hipEventCreate(&ev.e[1]));
	const hipEvent_t e0 = ev.e[0], e1 = ev.e[1];
	B_TRY(hipEventRecord(e0, stream));
	const int big = 0x7fffffff;
	const int init[6] = {big, big, big, -big - 1, -big - 1, -big - 1};
	B_TRY(hipMemcpyAsync(d_cb, init, sizeof init, hipMemcpyHostToDevice, stream));
	B_TRY(hipMemsetAsync(d_arrived, 0, (size_t) ni * 4, stream));
	const uint32_t blk = 256, gn = (n + blk - 1) / blk, gi = (ni + blk - 1) / blk;
	hipLaunchKernelGGL(k_bounds, dim3(gn), dim3(blk), 0, stream, d_tris, n, d_cb);
	hipLaunchKernelGGL(k_keys, dim3(gn), dim3(blk), 0, stream, d_tris, n, d_cb, d_keys, getenv("PT355_BVH_PLAIN_MORTON") ? 0 : 1);
	{
		size_t tmp_bytes = 0;
		B_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_keys, d_keys2, (int) n, 0, 63, stream));
		void *tmp;
		B_TRY(s.get(&tmp, tmp_bytes));
		B_TRY(hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, d_keys, d_keys2, (int) n, 0, 63, stream));
	}
	hipLaunchKernelGGL(k_leaf_boxes, dim3(gn), dim3(blk), 0, stream, d_tris, d_keys2, n, d_lbox);
	uint32_t *d_leafpos = nullptr;
	if (method == 0) {
		hipLaunchKernelGGL(k_hierarchy, dim3(gi), dim3(blk), 0, stream, d_keys2, (int) n, d_child, d_first, d_last, d_pi, d_pl);
		hipLaunchKernelGGL(k_fit, dim3(gn), dim3(blk), 0, stream, d_child, d_pi, d_pl, d_lbox, d_ibox, d_arrived, n);
	} else {
		Box *d_cbox[2];
		uint32_t *d_cnode[2], *d_nn, *d_valid, *d_pos, *d_size, *d_misc;
		B_TRY(s.get(&d_cbox[0], (size_t) n * sizeof(Box))); B_TRY(s.get(&d_cbox[1], (size_t) n * sizeof(Box)));
		B_TRY(s.get(&d_cnode[0], (size_t) n * 4)); B_TRY(s.get(&d_cnode[1], (size_t) n * 4));
		B_TRY(s.get(&d_nn, (size_t) n * 4)); B_TRY(s.get(&d_valid, (size_t) n * 4)); B_TRY(s.get(&d_pos, (size_t) n * 4));
		B_TRY(s.get(&d_size, (size_t) ni * 4)); B_TRY(s.get(&d_leafpos, (size_t) n * 4));
		B_TRY(s.get(&d_misc, 16)); // [0] merge counter, [1] deepest leaf
		B_TRY(hipMemsetAsync(d_misc, 0, 16, stream));
		size_t scan_bytes = 0;
		B_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_valid, d_pos, (int) n, stream));
		void *scan_tmp;
		B_TRY(s.get(&scan_tmp, scan_bytes));
		hipLaunchKernelGGL(k_ploc_init, dim3(gn), dim3(blk), 0, stream, d_lbox, n, d_cbox[0], d_cnode[0]);
		uint32_t m = n;
		int cur = 0;
		for (int pass = 0; m > 1u; pass++) {
			if (pass > 4096) { snprintf(err, errlen, "pt_build_bvh_device: PLOC did not converge"); return 2; }
			const uint32_t gm = (m + blk - 1) / blk;
			hipLaunchKernelGGL(k_ploc_nn, dim3(gm), dim3(blk), 0, stream, d_cbox[cur], m, d_nn);
			hipLaunchKernelGGL(k_ploc_merge, dim3(gm), dim3(blk), 0, stream, d_cbox[cur], d_cnode[cur], d_nn, m, n, d_misc, d_child, d_ibox, d_size, d_pi,
			                   d_pl, d_valid);
			B_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, d_valid, d_pos, (int) m, stream));
			hipLaunchKernelGGL(k_ploc_compact, dim3(gm), dim3(blk), 0, stream, d_cbox[cur], d_cnode[cur], d_valid, d_pos, m, d_cbox[cur ^ 1], d_cnode[cur ^ 1]);
			uint32_t merged = 0;
			B_TRY(hipMemcpyAsync(&merged, d_misc, 4, hipMemcpyDeviceToHost, stream));
			B_TRY(hipStreamSynchronize(stream));
			const uint32_t left = n - merged; // every merge removes one cluster
			if (left >= m) { snprintf(err, errlen, "pt_build_bvh_device: PLOC pass %d merged nothing (%u clusters)", pass, m); return 2; }
			m = left;
			cur ^= 1;
		}
		const uint32_t none = 0xffffffffu;
		B_TRY(hipMemcpyAsync(d_pi, &none, 4, hipMemcpyHostToDevice, stream)); // the root (node 0) has no parent
		hipLaunchKernelGGL(k_ploc_ranges, dim3((2u * n - 1u + blk - 1) / blk), dim3(blk), 0, stream, d_child, d_size, d_pi, d_pl, n, d_leafpos, d_first, d_last,
		                   d_misc + 1);
		uint32_t depth = 0;
		B_TRY(hipMemcpyAsync(&depth, d_misc + 1, 4, hipMemcpyDeviceToHost, stream));
		B_TRY(hipStreamSynchronize(stream));
		if (depth > 62u) { // deeper than the traversal stack: leave it to the radix tree, whose depth is bounded by the key length
			return ptb_build_bvh(stream, 0, tris_host, n, max_leaf, nodes_out, n_nodes_out, sorted_out, build_ms, err, errlen);
		}
	}
	hipLaunchKernelGGL(k_kept, dim3(gi), dim3(blk), 0, stream, d_first, d_last, ni, max_leaf, d_kept);
	{
		size_t tmp_bytes = 0;
		B_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_kept, d_slot, (int) ni, stream));
		void *tmp;
		B_TRY(s.get(&tmp, tmp_bytes));
		B_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, d_kept, d_slot, (int) ni, stream));
	}
	hipLaunchKernelGGL(k_emit, dim3(gi), dim3(blk), 0, stream, d_child, d_first, d_last, d_kept, d_slot, d_lbox, d_ibox, ni, d_nodes, (const uint32_t *) d_leafpos);
	if (d_leafpos) hipLaunchKernelGGL(k_scatter_tris, dim3((4u * n + blk - 1) / blk), dim3(blk), 0, stream, d_tris, d_keys2, d_leafpos, n, d_sorted);
	else hipLaunchKernelGGL(k_gather, dim3((4u * n + blk - 1) / blk), dim3(blk), 0, stream, d_tris, d_keys2, n, d_sorted);
	B_TRY(hipEventRecord(e1, stream));
	B_TRY(hipGetLastError());
	uint32_t tail[2] = {0, 0}; // kept[ni-1], slot[ni-1]
	B_TRY(hipMemcpyAsync(&tail[0], d_kept + (ni - 1u), 4, hipMemcpyDeviceToHost, stream));
	B_TRY(hipMemcpyAsync(&tail[1], d_slot + (ni - 1u), 4, hipMemcpyDeviceToHost, stream));
	B_TRY(hipStreamSynchronize(stream));
	const uint32_t n_kept = tail[0] + tail[1];
	const uint32_t n_nodes = 1u + 2u * n_kept;
	// child links are stored as floats (BVHNodeGLSL): odd indices above 2^24 are not representable and would round to the wrong sibling pair
	if (n_nodes > (1u << 24)) { snprintf(err, errlen, "pt_build_bvh_device: %u nodes exceed the 2^24 a float child index can address; raise max_leaf", n_nodes); return 1; }
	B_TRY(hipMemcpyAsync(nodes_out, d_nodes, (size_t) n_nodes * sizeof(NodeOut), hipMemcpyDeviceToHost, stream));
	B_TRY(hipMemcpyAsync(sorted_out, d_sorted, (size_t) n * 64, hipMemcpyDeviceToHost, stream));
	B_TRY(hipStreamSynchronize(stream));
	float ms = 0.0f;
	B_TRY(hipEventElapsedTime(&ms, e0, e1));
	if (build_ms) *build_ms = ms;
	*n_nodes_out = n_nodes;
	return 0;
}
