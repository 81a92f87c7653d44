// ubench_gather -- what does one divergent 64-byte-record gather cost on the vector-memory path of an MI355X CU?
// Each lane walks a dependent chain of 64-byte records (next index comes from the loaded data, like a BVH descent),
// fetching NL x dwordx4 per step.  Swept: active lanes per wave, loads per step, table size (L1/L2/MALL residency),
// waves per SIMD.  Prints ns and CU-cycles per wave-step.   build: hipcc --offload-arch=gfx950 -O3 -o ubench_gather ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NL>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ tab, uint32_t mask, int steps, int active_lanes, uint32_t *out)
{
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
	uint32_t acc = 0;
	if ((int) lane < active_lanes) {
		for (int s = 0; s < steps; s++) {
			const uint4 *p = tab + 4u * (size_t) idx;
			uint4 q0 = p[0], q1, q2, q3;
			uint32_t x = q0.x;
			if (NL > 1) { q1 = p[1]; x ^= q1.y; }
			if (NL > 2) { q2 = p[2]; x ^= q2.z; }
			if (NL > 3) { q3 = p[3]; x ^= q3.w; }
			acc += x;
			idx = (x + s) & mask;
		}
	}
	if (acc == 0x12345678u) out[0] = acc;
}


// quad-cooperative form: load k fetches the record of quad-lane k, each lane of the quad one 16-byte chunk of it
// (64 contiguous bytes per quad and instruction instead of four lines' worth of 16-byte pieces)
__global__ __launch_bounds__(256) void k_gather_coop(const uint4 *__restrict__ tab, uint32_t mask, int steps, int active_lanes, uint32_t *out)
{
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
	uint32_t acc = 0;
	if ((int) lane < active_lanes) {
		for (int s = 0; s < steps; s++) {
			uint32_t x = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t ik = __shfl(idx, (int) ((lane & ~3u) + k));
				const uint4 q = tab[4u * (size_t) ik + (lane & 3u)];
				x ^= q.x + q.w;
			}
			acc += x;
			idx = (x + s) & mask;
		}
	}
	if (acc == 0x12345678u) out[0] = acc;
}

// ---- the same measurement as a library call (pathtracer_amd/libptubench.so), for bench.py's roofline block: the rate
// at which this GPU's vector-memory path serves dependent 64-byte record gathers (4 x dwordx4 per lane-step, like one
// k_trace iteration) shaped like the trace kernel's: a table of the traversal blob's footprint, `hot_share` of the
// steps landing in a small hot set (16 KB: L1-resident, standing for the top of the tree), `active_lanes` of 64 lanes
// working, `waves_per_simd` resident waves -- and NO arithmetic between the gathers.  Measurement infrastructure, not
// product code: nothing in libpt355.so calls it.
__global__ __launch_bounds__(256) void k_gather_hot(const uint4 *__restrict__ tab, uint32_t n_rec, uint32_t hot_mask, uint32_t hot_cut, int steps,
                                                    int active_lanes, uint32_t *out)
{
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t idx = (uint32_t) (((unsigned long long) ((blockIdx.x * 256u + threadIdx.x) * 2654435761u) * n_rec) >> 32);
	uint32_t acc = 0;
	if ((int) lane < active_lanes) {
		for (int s = 0; s < steps; s++) {
			const uint4 *p = tab + 4u * (size_t) idx;
			const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
			const uint32_t x = (q0.x ^ q1.y ^ q2.z ^ q3.w) + (uint32_t) s * 0x9e3779b9u;
			acc += x;
			const uint32_t y = x * 0x85ebca6bu;
			idx = (y >> 16) < hot_cut ? (x & hot_mask) : (uint32_t) (((unsigned long long) x * n_rec) >> 32);
		}
	}
	if (acc == 0x12345678u) out[0] = acc;
}

extern "C" int ptu_gather_ceiling(int device, unsigned long long footprint_bytes, double hot_share, int active_lanes, int waves_per_simd,
                                  int steps, double *ns_per_wavestep, double *lane_steps_per_s)
{
	if (hipSetDevice(device) != hipSuccess) return 1;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 1;
	uint32_t n = (uint32_t) (footprint_bytes / 64);
	if (n < 1024u) n = 1024u;
	if (active_lanes < 1) active_lanes = 1;
	if (active_lanes > 64) active_lanes = 64;
	if (waves_per_simd < 1) waves_per_simd = 1;
	if (waves_per_simd > 8) waves_per_simd = 8;
	std::vector<uint32_t> h((size_t) n * 16);
	uint32_t ctr = 12345;
	for (auto &v : h) { uint32_t z = ctr++; z ^= z >> 16; z *= 0x85ebca6bu; z ^= z >> 13; z *= 0xc2b2ae35u; z ^= z >> 16; v = z; }
	uint4 *tab = nullptr; uint32_t *out = nullptr;
	if (hipMalloc(&tab, (size_t) n * 64) != hipSuccess) return 2;
	if (hipMalloc(&out, 4) != hipSuccess) { hipFree(tab); return 2; }
	hipMemcpy(tab, h.data(), (size_t) n * 64, hipMemcpyHostToDevice);
	const uint32_t hot_cut = (uint32_t) (hot_share * 65536.0 + 0.5);
	const int blocks = prop.multiProcessorCount * waves_per_simd; // 256-thread blocks: 4 waves = one per SIMD
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e30f;
	for (int rep = 0; rep < 4; rep++) { // first one warms the caches
		hipEventRecord(e0);
		k_gather_hot<<<blocks, 256>>>(tab, n, 255u, hot_cut, steps, active_lanes, out);
		hipEventRecord(e1);
		if (hipEventSynchronize(e1) != hipSuccess) { hipFree(tab); hipFree(out); return 3; }
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		if (rep > 0 && ms < best) best = ms;
	}
	hipEventDestroy(e0); hipEventDestroy(e1);
	hipFree(tab); hipFree(out);
	const double wavesteps_per_cu = (double) waves_per_simd * 4 * steps;
	if (ns_per_wavestep) *ns_per_wavestep = best * 1e6 / wavesteps_per_cu;
	if (lane_steps_per_s) *lane_steps_per_s = (double) blocks * 4 * active_lanes * steps / (best * 1e-3);
	return 0;
}

#ifndef PTU_LIBRARY
int main(int argc, char **argv)
{
	const int steps = 2000;
	uint32_t *out;
	CHECK(hipMalloc(&out, 4));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	printf("%-10s %-6s %-6s %-6s %10s %12s %14s\n", "table", "lanes", "loads", "w/SIMD", "ms", "ns/wavestep", "CUcyc/wavestep");
	for (uint32_t log_rec : {8u, 12u, 14u, 17u, 20u, 22u}) { // 16 KB (L1), 256 KB, 1 MB, 8 MB, 64 MB, 256 MB
		const uint32_t n = 1u << log_rec;
		std::vector<uint32_t> h((size_t) n * 16);
		uint32_t ctr = 12345;
		for (auto &v : h) { uint32_t z = ctr++; z ^= z >> 16; z *= 0x85ebca6bu; z ^= z >> 13; z *= 0xc2b2ae35u; z ^= z >> 16; v = z; } // murmur3 finaliser
		uint4 *tab;
		CHECK(hipMalloc(&tab, (size_t) n * 64));
		CHECK(hipMemcpy(tab, h.data(), (size_t) n * 64, hipMemcpyHostToDevice));
		for (int wps : {4, 7, 8}) {
			for (int lanes : {64, 48, 32, 16}) {
				for (int nl : {4, 3, 2, 1}) {
					if (log_rec != 17u && (wps != 7 || (lanes != 64 && lanes != 32) || (nl != 4 && nl != 2))) continue;
					const int blocks = 256 * wps;
					auto launch = [&]() {
						switch (nl) {
						case 4: k_gather<4><<<blocks, 256>>>(tab, n - 1, steps, lanes, out); break;
						case 3: k_gather<3><<<blocks, 256>>>(tab, n - 1, steps, lanes, out); break;
						case 2: k_gather<2><<<blocks, 256>>>(tab, n - 1, steps, lanes, out); break;
						default: k_gather<1><<<blocks, 256>>>(tab, n - 1, steps, lanes, out); break;
						}
					};
					launch();
					CHECK(hipDeviceSynchronize());
					CHECK(hipEventRecord(e0));
					launch();
					CHECK(hipEventRecord(e1));
					CHECK(hipEventSynchronize(e1));
					float ms;
					CHECK(hipEventElapsedTime(&ms, e0, e1));
					const double wavesteps_per_cu = (double) wps * 4 * steps; // waves per CU x steps
					const double ns = ms * 1e6 / wavesteps_per_cu;
					char tb[32];
					snprintf(tb, sizeof tb, "%uKB", n * 64 / 1024);
					printf("%-10s %-6d %-6d %-6d %10.3f %12.1f %14.1f\n", tb, lanes, nl, wps, ms, ns, ns * 2.4);
					if (nl == 4) {
						k_gather_coop<<<blocks, 256>>>(tab, n - 1, steps, lanes, out);
						CHECK(hipDeviceSynchronize());
						CHECK(hipEventRecord(e0));
						k_gather_coop<<<blocks, 256>>>(tab, n - 1, steps, lanes, out);
						CHECK(hipEventRecord(e1));
						CHECK(hipEventSynchronize(e1));
						CHECK(hipEventElapsedTime(&ms, e0, e1));
						const double ns2 = ms * 1e6 / wavesteps_per_cu;
						printf("%-10s %-6d %-6s %-6d %10.3f %12.1f %14.1f\n", tb, lanes, "coop4", wps, ms, ns2, ns2 * 2.4);
					}
				}
			}
		}
		CHECK(hipFree(tab));
	}
	return 0;
}
#endif
