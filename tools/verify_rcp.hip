// verify_rcp -- exhaustive check, over every finite non-zero fp32 x, of where a short v_rcp_f32 + FMA sequence returns the correctly rounded 1.0f / x, i.e. the
// same bits as the IEEE division the oracle performs.  Prints the mismatches per candidate and per exponent.
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o verify_rcp verify_rcp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ float cand1(float x) { float r = __builtin_amdgcn_rcpf(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ float cand2(float x) { float r = cand1(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }

__global__ void k_check(unsigned long long *bad, uint32_t *first)
{
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long b1 = 0, b2 = 0, n = 0;
	for (uint32_t hi = 0; hi < 256u; hi++) {
		const uint32_t bits = (hi << 24) | tid; // tid covers 2^24 values
		const float x = __uint_as_float(bits);
		const float ax = __builtin_fabsf(x);
		if (!(ax > 0.0f) || !(ax < __builtin_inff())) continue;
		n++;
		const uint32_t want = __float_as_uint(1.0f / x);
		if (__float_as_uint(cand1(x)) != want) atomicAdd(&bad[8 + ((bits >> 23) & 0xffu)], 1ull);
		if (__float_as_uint(cand1(x)) != want) { b1++; if (atomicAdd(&first[0], 1u) < 8u) printf("cand1 x=%08x want=%08x got=%08x\n", bits, want, __float_as_uint(cand1(x))); }
		if (__float_as_uint(cand2(x)) != want) { b2++; if (atomicAdd(&first[1], 1u) < 8u) printf("cand2 x=%08x want=%08x got=%08x\n", bits, want, __float_as_uint(cand2(x))); }
	}
	atomicAdd(&bad[0], n); atomicAdd(&bad[1], b1); atomicAdd(&bad[2], b2);
}

int main()
{
	unsigned long long *bad; uint32_t *first;
	hipMalloc(&bad, 8 * 264); hipMemset(bad, 0, 8 * 264);
	hipMalloc(&first, 8); hipMemset(first, 0, 8);
	k_check<<<(1u << 24) / 256, 256>>>(bad, first);
	unsigned long long h[264];
	if (hipMemcpy(h, bad, 8 * 264, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
	for (int e = 0; e < 256; e++) if (h[8 + e]) printf("rcp + 1 step: %llu mismatches among x with biased exponent %d (|x| in [2^%d, 2^%d))\n", h[8 + e], e, e - 127, e - 126);
	printf("checked %llu values: rcp + 1 Newton step mismatches %llu, + 2 steps mismatches %llu\n", h[0], h[1], h[2]);
	return 0;
}
