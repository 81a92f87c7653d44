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

// (C) the same 128 bytes per slot as 8 float4 columns: 16-byte accesses, a quarter of the load / store instructions
__global__ __launch_bounds__(256) void k_copy4(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n)
{
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	float4 v[NC / 4];
#pragma unroll
	for (int c = 0; c < NC / 4; c++) v[c] = in[(size_t) c * n + i];
	float s = 0.0f;
#pragma unroll
	for (int c = 0; c < NC / 4; c++) s += v[c].x + v[c].y + v[c].z + v[c].w;
#pragma unroll
	for (int c = 0; c < NC / 4; c++) out[(size_t) c * n + i] = make_float4(v[c].x + s, v[c].y + s, v[c].z + s, v[c].w + s);
}
// (D) the raw ceiling: a flat float4 copy, one element per thread
__global__ __launch_bounds__(256) void k_flat4(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4)
{
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (i < n4) out[i] = in[i];
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
	for (int rep = 0; rep < 2; rep++) {
		float ms;
		CHECK(hipEventRecord(e0));
		k_copy4<<<(unsigned) (n / 256), 256>>>((const float4 *) in, (float4 *) out, n);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		printf("SoA of float4 columns : %.3f ms, %.2f TB/s (read + write)\n", ms, 2.0 * n * NC * 4 / ms / 1e9);
		const size_t n4 = n * NC / 4;
		CHECK(hipEventRecord(e0));
		k_flat4<<<(unsigned) ((n4 + 255) / 256), 256>>>((const float4 *) in, (float4 *) out, n4);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		printf("flat float4 copy      : %.3f ms, %.2f TB/s (read + write)\n", ms, 2.0 * n * NC * 4 / ms / 1e9);
		CHECK(hipEventRecord(e0));
		CHECK(hipMemcpyAsync(out, in, n * NC * 4, hipMemcpyDeviceToDevice, 0));
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		printf("hipMemcpy D2D         : %.3f ms, %.2f TB/s (read + write)\n", ms, 2.0 * n * NC * 4 / ms / 1e9);
	}
	return 0;
}
