// ubench_gather -- what does one divergent 64-byte-record gather cost on the vector-memory path of an MI355X CU?
// Each lane walks a dependent chain of 64-byte records (next index comes from the loaded data, like a BVH descent),
// fetching NL x dwordx4 per step.  Swept: active lanes per wave, loads per step, table size (L1/L2/MALL residency),
// waves per SIMD.  Prints ns and CU-cycles per wave-step.   build: hipcc --offload-arch=gfx950 -O3 -o ubench_gather ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NL>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ tab, uint32_t mask, int steps, int active_lanes, uint32_t *out)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t salt = (blockIdx.x * 256u + threadIdx.x) * 2654435761u; // keeps lanes that meet on a record from merging for good
	uint32_t idx = salt & mask;
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
			idx = ((x + s) ^ salt) * 0x85ebca6bu >> 7 & mask;
		}
	}
	if (acc == 0x12345678u) out[0] = acc;
}


// quad-cooperative form: load k fetches the record of quad-lane k, each lane of the quad one 16-byte chunk of it
// (64 contiguous bytes per quad and instruction instead of four lines' worth of 16-byte pieces)
__global__ __launch_bounds__(256) void k_gather_coop(const uint4 *__restrict__ tab, uint32_t mask, int steps, int active_lanes, uint32_t *out)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t salt = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
	uint32_t idx = salt & mask;
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
			idx = ((x + s) ^ salt) * 0x85ebca6bu >> 7 & mask;
		}
	}
	if (acc == 0x12345678u) out[0] = acc;
}

// ---- the same measurement as a library call (pathtracer_amd/libptubench.so), for bench.py's roofline block: the rate
// at which this GPU's vector-memory path serves dependent 64-byte record gathers (4 x dwordx4 per lane-step, like one
// k_trace iteration) with the trace kernel's CACHE BEHAVIOUR PER RECORD FETCH and nothing else: a step lands
//   * in a 16 KB hot set (L1-resident: the top of the tree)                      with probability 1 - miss_l1,
//   * in a 2 MB region (misses the 32 KB L1, stays in the XCD's 4 MB L2)          with probability miss_l1 - miss_l2,
//   * in a 64 MB region (misses L1 and L2, served by the Infinity Cache / HBM)    with probability miss_l2,
// where miss_l1 / miss_l2 are the kernel's own TCP->TCC read requests and TCC->fabric read requests per lane-step (PMC),
// `active_lanes` of 64 lanes work, `waves_per_simd` waves are resident, and there is NO arithmetic between the gathers.
// (Round 2's first version took the TCP miss share per ACCESS as the share of steps outside the hot set; a record fetch is
// four accesses of which only the first can miss, so that understated the kernel's misses fourfold and overstated the
// ceiling.  `ubench_gather tiers ...` under rocprofv3 --pmc shows the model reproducing the requested rates:
// profiles/r02g_ubench_calibration.txt.)  Measurement infrastructure, not product code: nothing in libpt355.so calls it.
#define PTU_HOT_REC 256u            /* 16 KB */
#define PTU_L2_REC (2u << 20 >> 6)  /* 2 MB  */
#define PTU_COLD_REC (64u << 20 >> 6) /* 64 MB */
// COOP = 0: each lane fetches its own record with 4 x dwordx4 (64 lanes -> 64 lines per instruction).
// COOP = 1: quad-cooperative -- instruction k fetches the record of quad-lane k, each lane of the quad one 16-byte piece
//           of it (64 contiguous bytes per quad: 16 lines per instruction), then the pieces go to their owner by
//           quad-permute DPP moves (here: the four words the chain needs).
template <int COOP>
__global__ __launch_bounds__(256) void k_gather_tiers(const uint4 *__restrict__ tab, uint32_t cut1, uint32_t cut2, int steps, int active_lanes, uint32_t *out)
{
	const uint32_t lane = threadIdx.x & 63u;
	// the lane's own salt goes into every step: two lanes that meet on one record part again (without it the lanes of a
	// wave merge into a few shared chains within a few hundred steps and the "gather" degenerates into broadcasts)
	const uint32_t salt = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
	uint32_t idx = salt & (PTU_HOT_REC - 1u);
	uint32_t acc = 0;
	if ((int) lane < active_lanes) {
		for (int s = 0; s < steps; s++) {
			uint4 q0, q1, q2, q3;
			if (COOP) {
				const uint32_t sub = lane & 3u, qb = lane & ~3u;
				uint32_t part[4];
#pragma unroll
				for (int k = 0; k < 4; k++) { // piece `sub` of quad-lane k's record, folded to one word
					const uint4 r = tab[4u * (size_t) __shfl(idx, (int) (qb + k)) + sub];
					part[k] = sub == 0 ? r.x : (sub == 1 ? r.y : (sub == 2 ? r.z : r.w));
				}
				// round t: every lane sends the word meant for quad-lane (sub + t) & 3 and receives the one from quad-lane (sub - t) & 3
				uint32_t xx = 0;
#pragma unroll
				for (int t = 0; t < 4; t++) {
					const uint32_t to = (sub + (uint32_t) t) & 3u;
					const uint32_t send = to == 0 ? part[0] : (to == 1 ? part[1] : (to == 2 ? part[2] : part[3]));
					xx ^= __shfl(send, (int) (qb + ((sub - (uint32_t) t) & 3u)));
				}
				q0 = make_uint4(xx, 0, 0, 0); q1 = q2 = q3 = make_uint4(0, 0, 0, 0); // = q0.x ^ q1.y ^ q2.z ^ q3.w of the lane's own record
			} else {
				const uint4 *p = tab + 4u * (size_t) idx;
				q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
			}
			const uint32_t x = ((q0.x ^ q1.y ^ q2.z ^ q3.w) + (uint32_t) s * 0x9e3779b9u) ^ salt;
			acc += x;
			const uint32_t u = (x * 0x85ebca6bu) >> 16, z = x * 0xc2b2ae35u;
			idx = u < cut1 ? (z >> 24) // 256 hot records
			    : (u < cut2 ? PTU_HOT_REC + (uint32_t) (((unsigned long long) z * PTU_L2_REC) >> 32)
			                : PTU_HOT_REC + PTU_L2_REC + (uint32_t) (((unsigned long long) z * PTU_COLD_REC) >> 32));
		}
	}
	if (acc == 0x12345678u) out[0] = acc;
}

static int gather_ceiling(int coop, int device, double miss_l1, double miss_l2, int active_lanes, int waves_per_simd,
                          int steps, double *ns_per_wavestep, double *lane_steps_per_s)
{
	if (hipSetDevice(device) != hipSuccess) return 1;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 1;
	if (!(miss_l1 >= 0.0)) miss_l1 = 0.0;
	if (miss_l1 > 1.0) miss_l1 = 1.0;
	if (!(miss_l2 >= 0.0)) miss_l2 = 0.0;
	if (miss_l2 > miss_l1) miss_l2 = miss_l1;
	if (active_lanes < 1) active_lanes = 1;
	if (active_lanes > 64) active_lanes = 64;
	if (waves_per_simd < 1) waves_per_simd = 1;
	if (waves_per_simd > 8) waves_per_simd = 8;
	const size_t n = (size_t) PTU_HOT_REC + PTU_L2_REC + PTU_COLD_REC;
	std::vector<uint32_t> h(n * 16);
	uint32_t ctr = 12345;
	for (auto &v : h) { uint32_t z = ctr++; z ^= z >> 16; z *= 0x85ebca6bu; z ^= z >> 13; z *= 0xc2b2ae35u; z ^= z >> 16; v = z; }
	uint4 *tab = nullptr; uint32_t *out = nullptr;
	if (hipMalloc(&tab, n * 64) != hipSuccess) return 2;
	if (hipMalloc(&out, 4) != hipSuccess) { (void) hipFree(tab); return 2; }
	(void) hipMemcpy(tab, h.data(), n * 64, hipMemcpyHostToDevice);
	const uint32_t cut1 = (uint32_t) ((1.0 - miss_l1) * 65536.0 + 0.5), cut2 = (uint32_t) ((1.0 - miss_l2) * 65536.0 + 0.5);
	const int blocks = prop.multiProcessorCount * waves_per_simd; // 256-thread blocks: 4 waves = one per SIMD
	hipEvent_t e0, e1;
	if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void) hipFree(tab); (void) hipFree(out); return 3; }
	float best = 1e30f;
	for (int rep = 0; rep < 4; rep++) { // first one warms the caches
		(void) hipEventRecord(e0);
		if (coop) k_gather_tiers<1><<<blocks, 256>>>(tab, cut1, cut2, steps, active_lanes, out);
		else k_gather_tiers<0><<<blocks, 256>>>(tab, cut1, cut2, steps, active_lanes, out);
		(void) hipEventRecord(e1);
		if (hipEventSynchronize(e1) != hipSuccess) { (void) hipFree(tab); (void) hipFree(out); return 3; }
		float ms = 0;
		(void) hipEventElapsedTime(&ms, e0, e1);
		if (rep > 0 && ms < best) best = ms;
	}
	(void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
	(void) hipFree(tab); (void) hipFree(out);
	const double wavesteps_per_cu = (double) waves_per_simd * 4 * steps;
	if (ns_per_wavestep) *ns_per_wavestep = best * 1e6 / wavesteps_per_cu;
	if (lane_steps_per_s) *lane_steps_per_s = (double) blocks * 4 * active_lanes * steps / (best * 1e-3);
	return 0;
}

extern "C" int ptu_gather_ceiling(int device, double miss_l1, double miss_l2, int active_lanes, int waves_per_simd,
                                  int steps, double *ns_per_wavestep, double *lane_steps_per_s)
{
	return gather_ceiling(0, device, miss_l1, miss_l2, active_lanes, waves_per_simd, steps, ns_per_wavestep, lane_steps_per_s);
}
// the quad-cooperative access shape (see k_gather_tiers): what the memory path would serve if a record were fetched by four lanes
extern "C" int ptu_gather_ceiling_coop(int device, double miss_l1, double miss_l2, int active_lanes, int waves_per_simd,
                                       int steps, double *ns_per_wavestep, double *lane_steps_per_s)
{
	return gather_ceiling(1, device, miss_l1, miss_l2, active_lanes, waves_per_simd, steps, ns_per_wavestep, lane_steps_per_s);
}

#ifndef PTU_LIBRARY
int main(int argc, char **argv)
{
	if (argc >= 6 && (!strcmp(argv[1], "tiers") || !strcmp(argv[1], "tiers_coop"))) { // ubench_gather tiers <miss_l1> <miss_l2> <lanes> <waves/SIMD>: one line, for PMC calibration
		double ns = 0, rate = 0;
		const int rc = gather_ceiling(!strcmp(argv[1], "tiers_coop"), 0, atof(argv[2]), atof(argv[3]), atoi(argv[4]), atoi(argv[5]), 2000, &ns, &rate);
		const double launches = 4.0, lane_steps = 256.0 * atoi(argv[5]) * 4 * atoi(argv[4]) * 2000.0;
		printf("%s miss_l1=%s miss_l2=%s lanes=%s waves/SIMD=%s rc=%d: %.2f ns per wave-step, %.1f G lane-steps/s; %.0f launches of %.0f lane-steps each (256 CUs assumed)\n",
		       argv[1], argv[2], argv[3], argv[4], argv[5], rc, ns, rate / 1e9, launches, lane_steps);
		return rc;
	}
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
