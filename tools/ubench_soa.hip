// ubench_soa -- does the queue layout matter for the shading kernel's HBM bandwidth?  One thread per slot reads NC float
// columns and writes NC float columns, (A) structure-of-arrays with a column stride of n slots (the current queues)
// vs (B) tiles of 256 slots holding all their columns back to back.   build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#define NC 32
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool TILED>
__global__ __launch_bounds__(256) void k_copy(const float *__restrict__ in, float *__restrict__ out, size_t n)
{
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	float v[NC];
#pragma unroll
	for (int c = 0; c < NC; c++) v[c] = TILED ? in[(i >> 8) * (NC * 256) + c * 256 + (i & 255)] : in[(size_t) c * n + i];
	float s = 0.0f;
#pragma unroll
	for (int c = 0; c < NC; c++) s += v[c];
#pragma unroll
	for (int c = 0; c < NC; c++) { const float o = v[c] + s; if (TILED) out[(i >> 8) * (NC * 256) + c * 256 + (i & 255)] = o; else out[(size_t) c * n + i] = o; }
}

int main()
{
	const size_t n = (size_t) 32 << 20; // 32 Mi slots x 32 columns x 4 B = 4 GiB in, 4 GiB out
	float *in, *out;
	CHECK(hipMalloc(&in, n * NC * 4)); CHECK(hipMalloc(&out, n * NC * 4));
	CHECK(hipMemset(in, 0, n * NC * 4));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int rep = 0; rep < 2; rep++) {
		for (int tiled = 0; tiled < 2; tiled++) {
			CHECK(hipEventRecord(e0));
			if (tiled) k_copy<true><<<(unsigned) (n / 256), 256>>>(in, out, n); else k_copy<false><<<(unsigned) (n / 256), 256>>>(in, out, n);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			printf("%s: %.3f ms, %.2f TB/s (read + write)\n", tiled ? "tiles of 256 slots" : "SoA, stride n     ", ms, 2.0 * n * NC * 4 / ms / 1e9);
		}
	}
	return 0;
}
