// Full-sweep SAH BVH build on the host, producing byte-for-byte the node array and triangle order
// that the reference's CalculateBVH (src/scene/bvh.cpp:31-75) gets from madmann91/bvh v1's
// SweepSahBuilder in its single-threaded configuration (the reference never enables OpenMP:
// CMakeLists.txt:97-108).  Own implementation; the behaviours that fix the output bytes are:
//   * primitive box = bounds of {p0, p0-(p0-p1), p0+(p2-p0)}, centre = ((p0+p1')+p2') * (1/3)
//     (thirdparty/bvh/include/bvh/triangle.hpp:34-52)
//   * per axis, primitives sorted once by an order-preserving integer key of the centre, stably
//     (radix_sort.hpp:88-95; LSD radix = any stable sort by key)
//   * split search: suffix sweep then prefix sweep, first strict minimum wins, axis ties go to the
//     lower axis (sweep_sah_builder.hpp:131-146,189-193); half_area = (dx+dy)*dz + dx*dy
//   * no split when best >= half_area(node)*(n-1) unless n > 16, then median on the largest axis
//   * children numbered in DFS order, smaller child first (top_down_builder.hpp:58-69), depth <= 64
//   * nodes hold indices as floats, triangles are reordered into leaf order (src/scene/bvh.cpp:9-12,51-70)
// Must be compiled without FMA contraction (-ffp-contract=off): see SURVEY.md 8a row B1.
#include <algorithm>
#include <cfloat>
#include <cstring>
#include <vector>

#include "../../../include/pt355_host.h"
#include "layouts.h"

namespace {

struct Box {
	float lo[3], hi[3];
	void reset()
	{
		for (int a = 0; a < 3; a++) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; }
	}
	void grow(const Box &o)
	{
		for (int a = 0; a < 3; a++) {
			lo[a] = o.lo[a] < lo[a] ? o.lo[a] : lo[a]; // std::min(lo, o.lo)
			hi[a] = hi[a] < o.hi[a] ? o.hi[a] : hi[a]; // std::max(hi, o.hi)
		}
	}
	void grow_point(const float p[3])
	{
		for (int a = 0; a < 3; a++) {
			lo[a] = p[a] < lo[a] ? p[a] : lo[a];
			hi[a] = hi[a] < p[a] ? p[a] : hi[a];
		}
	}
	float half_area() const
	{
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return (dx + dy) * dz + dx * dy;
	}
	unsigned widest_axis() const
	{
		const float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
		unsigned a = 0;
		if (d[0] < d[1]) a = 1;
		if (d[a] < d[2]) a = 2;
		return a;
	}
};

// order-preserving map float -> uint32 (negative values mirrored, both zeros equal)
inline uint32_t sort_key(float x)
{
	uint32_t y;
	memcpy(&y, &x, 4);
	return (y & 0x80000000u) ? (0u - y) : (y ^ 0x80000000u);
}

struct RawNode {
	Box box;
	uint32_t count;
	uint32_t first;
};

struct Range {
	size_t node, begin, end, depth;
	size_t size() const { return end - begin; }
};

struct Builder {
	size_t n;
	std::vector<Box> prim_box;
	std::vector<float> centre; // 3 per primitive
	std::vector<uint32_t> order[3];
	std::vector<float> suffix_cost[3];
	std::vector<uint8_t> goes_left;
	std::vector<uint32_t> scratch;
	std::vector<RawNode> nodes;
	bool failed = false;

	static constexpr size_t kMaxDepth = 64;    // top_down_builder.hpp:36
	static constexpr size_t kMaxLeaf = 16;     // top_down_builder.hpp:41
	static constexpr float kTraversalCost = 1; // sah_based_algorithm.hpp:16

	void sweep(int axis, size_t begin, size_t end, float *best_cost, size_t *best_at)
	{
		const uint32_t *ord = order[axis].data();
		float *cost = suffix_cost[axis].data();
		Box acc;
		acc.reset();
		for (size_t i = end - 1; i > begin; --i) {
			acc.grow(prim_box[ord[i]]);
			cost[i] = acc.half_area() * (float) (end - i);
		}
		acc.reset();
		float best = FLT_MAX;
		size_t at = end;
		for (size_t i = begin; i < end - 1; ++i) {
			acc.grow(prim_box[ord[i]]);
			const float c = acc.half_area() * (float) (i + 1 - begin) + cost[i + 1];
			if (c < best) { best = c; at = i + 1; }
		}
		*best_cost = best;
		*best_at = at;
	}

	void keep_order_partition(uint32_t *first, uint32_t *last)
	{
		uint32_t *out = first;
		size_t spill = 0;
		for (uint32_t *p = first; p != last; ++p) {
			if (goes_left[*p]) *out++ = *p;
			else scratch[spill++] = *p;
		}
		memcpy(out, scratch.data(), spill * sizeof(uint32_t));
	}

	// returns true and fills the two child ranges when the node was split
	bool split(const Range &r, Range *lo_child, Range *hi_child)
	{
		RawNode &node = nodes[r.node];
		if (r.size() <= 1 || r.depth >= kMaxDepth) {
			node.first = (uint32_t) r.begin;
			node.count = (uint32_t) r.size();
			return false;
		}
		float cost[3];
		size_t at[3];
		for (int a = 0; a < 3; a++) sweep(a, r.begin, r.end, &cost[a], &at[a]);
		unsigned axis = 0;
		if (cost[0] > cost[1]) axis = 1;
		if (cost[axis] > cost[2]) axis = 2;
		size_t cut = at[axis];

		const float leaf_cost = node.box.half_area() * ((float) r.size() - kTraversalCost);
		if (cost[axis] >= leaf_cost) {
			if (r.size() > kMaxLeaf) {
				axis = node.box.widest_axis();
				cut = (r.begin + r.end) / 2;
			} else {
				node.first = (uint32_t) r.begin;
				node.count = (uint32_t) r.size();
				return false;
			}
		}

		const uint32_t *ord = order[axis].data();
		for (size_t i = r.begin; i < cut; ++i) goes_left[ord[i]] = 1;
		for (size_t i = cut; i < r.end; ++i) goes_left[ord[i]] = 0;
		for (unsigned k = 1; k <= 2; k++) {
			uint32_t *o = order[(axis + k) % 3].data();
			keep_order_partition(o + r.begin, o + r.end);
		}
		Box lo_box, hi_box;
		lo_box.reset();
		hi_box.reset();
		for (size_t i = r.begin; i < cut; ++i) lo_box.grow(prim_box[ord[i]]);
		for (size_t i = cut; i < r.end; ++i) hi_box.grow(prim_box[ord[i]]);

		const size_t child = nodes.size();
		// A binary tree over n primitives has at most 2n - 1 nodes, and the caller's array holds 2n + 1 (the library allocates the
		// same, bvh.hpp / sweep_sah_builder.hpp:61).  Only non-finite input gets here with more: with NaN costs no split position
		// wins, the "split" keeps the whole range on one side, and the range recurses to max_depth.  The reference writes past
		// its array then; this builder reports failure.
		if (child + 2 > 2 * n + 1) { failed = true; return false; }
		nodes.resize(child + 2);
		RawNode &self = nodes[r.node]; // resize may have moved the array
		self.first = (uint32_t) child;
		self.count = 0;
		nodes[child].box = lo_box;
		nodes[child + 1].box = hi_box;
		*lo_child = Range{child, r.begin, cut, r.depth + 1};
		*hi_child = Range{child + 1, cut, r.end, r.depth + 1};
		return true;
	}

	void run(const PtTriangle *tris, size_t count)
	{
		n = count;
		prim_box.resize(n);
		centre.resize(3 * n);
		const float third = 1.0f / 3.0f;
		Box all;
		all.reset();
		for (size_t i = 0; i < n; i++) {
			const PtTriangle &t = tris[i];
			float p1[3], p2[3];
			for (int a = 0; a < 3; a++) {
				const float e1 = t.v0[a] - t.v1[a]; // the library keeps p0 and two edges, and
				const float e2 = t.v2[a] - t.v0[a]; // rebuilds p1, p2 from them (not bit-equal to v1, v2)
				p1[a] = t.v0[a] - e1;
				p2[a] = t.v0[a] + e2;
				centre[3 * i + a] = ((t.v0[a] + p1[a]) + p2[a]) * third;
			}
			Box &b = prim_box[i];
			for (int a = 0; a < 3; a++) b.lo[a] = b.hi[a] = t.v0[a];
			b.grow_point(p1);
			b.grow_point(p2);
			all.grow(b);
		}
		std::vector<std::pair<uint32_t, uint32_t>> keyed(n);
		for (int a = 0; a < 3; a++) {
			for (size_t i = 0; i < n; i++) keyed[i] = {sort_key(centre[3 * i + a]), (uint32_t) i};
			std::stable_sort(keyed.begin(), keyed.end(),
			                 [](const std::pair<uint32_t, uint32_t> &x, const std::pair<uint32_t, uint32_t> &y) {
				                 return x.first < y.first;
			                 });
			order[a].resize(n);
			for (size_t i = 0; i < n; i++) order[a][i] = keyed[i].second;
			suffix_cost[a].assign(n, 0.0f);
		}
		goes_left.assign(n, 0);
		scratch.resize(n);
		nodes.reserve(2 * n + 1);
		nodes.resize(1);
		nodes[0].box = all;
		nodes[0].count = 0;
		nodes[0].first = 0;

		std::vector<Range> todo;
		todo.push_back(Range{0, 0, n, 0});
		while (!todo.empty()) {
			const Range r = todo.back();
			todo.pop_back();
			Range a, b;
			if (!split(r, &a, &b)) { if (failed) return; continue; }
			if (a.size() > b.size()) std::swap(a, b);
			todo.push_back(b); // larger half waits; smaller half is expanded (and numbered) first
			todo.push_back(a);
		}
	}
};

} // namespace

extern "C" uint32_t pth_build_bvh(const void *tris_in, uint32_t n, void *nodes_out, void *sorted_out)
{
	if (!tris_in || n == 0 || !nodes_out || !sorted_out) return 0;
	const PtTriangle *tris = (const PtTriangle *) tris_in;
	Builder b;
	b.run(tris, n);
	if (b.failed) return 0; // non-finite vertices: no tree (see Builder::split)
	PtNode *out = (PtNode *) nodes_out;
	PtTriangle *sorted = (PtTriangle *) sorted_out;
	for (size_t i = 0; i < b.nodes.size(); i++) {
		const RawNode &r = b.nodes[i];
		for (int a = 0; a < 3; a++) { out[i].bmin[a] = r.box.lo[a]; out[i].bmax[a] = r.box.hi[a]; }
		out[i].first = (float) r.first;
		out[i].count = (float) r.count;
	}
	const uint32_t *leaf_order = b.order[0].data(); // the x-sorted list is the one the library exports
	for (size_t k = 0; k < n; k++) sorted[k] = tris[leaf_order[k]];
	return (uint32_t) b.nodes.size();
}
