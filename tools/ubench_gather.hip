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
