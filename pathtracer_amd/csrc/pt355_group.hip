// pt_group -- several devices behind ONE handle in ONE process (include/pt355.h "Multi-device"), so that a C++ caller
// shaped like the reference's main() (src/main.cpp:57-101) can use every GPU of a node without torch.distributed.
// The members are ordinary pt_ctx contexts, one per listed device, driven from the calling thread (every pt_* call is
// asynchronous on its context's stream, so the devices run concurrently); the scene is replicated; there is no
// data-path exchange while rendering.  The finished framebuffer is gathered onto the first device with peer copies
// over xGMI (hipMemcpyPeerAsync; the single-process counterpart of the RCCL gather / reduce bench.py does between
// processes).  Partition arithmetic, packing and blending are the contexts' own (pt_pack_owned / pt_unpack_from).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt355.h"

struct pt_group {
	std::vector<pt_ctx *> member;
	std::vector<int32_t> device;
	uint32_t mode = PT_GROUP_TILES;
	uint32_t width = 0, height = 0;
	std::string err;
	std::vector<void *> staging_src; // on member i's device: its packed pixels
	std::vector<void *> staging_dst; // on device 0: where member i's pixels land
	float *gathered = nullptr;       // samples mode: the weighted sum, on device 0
	hipStream_t stream0 = nullptr;   // device 0: gather arithmetic
	std::vector<hipStream_t> copy_stream;
	std::vector<uint64_t> frames_done; // samples mode: frames rendered per member (its blend weight)
	uint64_t frames_total = 0;
};

namespace {

thread_local std::string g_group_create_error;

int gfail(pt_group *g, int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (g) g->err = buf;
	else g_group_create_error = buf;
	return code;
}

#define G_TRY(g, call)                                                                                 \
	do {                                                                                               \
		hipError_t e_ = (call);                                                                        \
		if (e_ != hipSuccess) return gfail(g, PT_EHIP, "%s: %s", #call, hipGetErrorString(e_));        \
	} while (0)
#define G_PT(g, i, call)                                                                               \
	do {                                                                                               \
		int rc_ = (call);                                                                              \
		if (rc_ != PT_OK) return gfail(g, rc_, "member %u (device %d): %s", (unsigned) (i), g->device[i], pt_last_error(g->member[i])); \
	} while (0)

// acc = (first ? 0 : acc) + w * src, alpha forced to 1: the sample-interleaved reduce, sum_r (n_r / N) * mean_r
__global__ void k_group_axpy(float4 *acc, const float4 *src, float w, uint32_t n, int first)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4 s = src[i];
	float4 a = first ? make_float4(0.0f, 0.0f, 0.0f, 1.0f) : acc[i];
	a.x += w * s.x; a.y += w * s.y; a.z += w * s.z; a.w = 1.0f;
	acc[i] = a;
}

} // namespace

extern "C" {

const char *pt_group_last_error(const pt_group *g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int pt_group_destroy(pt_group *g)
{
	if (!g) return PT_OK;
	for (size_t i = 0; i < g->member.size(); i++) {
		if (!g->member[i]) continue; // a member that was never created (failed pt_group_create) owns nothing, and its device may not exist
		pt_sync(g->member[i]);
		hipSetDevice(g->device[i]);
		if (i < g->staging_src.size() && g->staging_src[i]) hipFree(g->staging_src[i]);
		if (i < g->copy_stream.size() && g->copy_stream[i]) hipStreamDestroy(g->copy_stream[i]);
	}
	if (!g->member.empty() && g->member[0]) hipSetDevice(g->device[0]);
	for (void *p : g->staging_dst) if (p) hipFree(p);
	if (g->gathered) hipFree(g->gathered);
	if (g->stream0) hipStreamDestroy(g->stream0);
	for (pt_ctx *c : g->member) if (c) pt_destroy(c);
	delete g;
	(void) hipGetLastError(); // nothing above may leave a sticky error for the thread's next launch check
	return PT_OK;
}

int pt_group_create(const pt_config *cfg, const int32_t *devices, uint32_t n_devices, uint32_t mode, pt_group **out)
{
	if (!cfg || !devices || !out || n_devices == 0) return gfail(nullptr, PT_EINVAL, "pt_group_create: null argument or empty device list");
	*out = nullptr;
	if (mode != PT_GROUP_TILES && mode != PT_GROUP_SAMPLES) return gfail(nullptr, PT_EINVAL, "pt_group_create: unknown mode %u", mode);
	if (n_devices > 64) return gfail(nullptr, PT_EINVAL, "pt_group_create: at most 64 members");
	pt_group *g = new pt_group();
	g->mode = mode; g->width = cfg->width; g->height = cfg->height;
	g->device.assign(devices, devices + n_devices);
	g->member.assign(n_devices, nullptr);
	g->staging_src.assign(n_devices, nullptr);
	g->staging_dst.assign(n_devices, nullptr);
	g->copy_stream.assign(n_devices, nullptr);
	g->frames_done.assign(n_devices, 0);
	for (uint32_t i = 0; i < n_devices; i++) {
		pt_config c = *cfg;
		c.device = devices[i];
		if (mode == PT_GROUP_TILES) { c.partition = PT_PARTITION_TILES; c.rank = i; c.world = n_devices; }
		else { c.partition = PT_PARTITION_FULL; c.rank = 0; c.world = 1; }
		const int rc = pt_create(&c, &g->member[i]);
		if (rc != PT_OK) {
			gfail(nullptr, rc, "pt_group_create: member %u (device %d): %s", i, devices[i], pt_last_error(nullptr));
			pt_group_destroy(g);
			return rc;
		}
	}
#define CREATE_TRY(call)                                                                               \
	do {                                                                                               \
		hipError_t e_ = (call);                                                                        \
		if (e_ != hipSuccess) {                                                                        \
			gfail(nullptr, PT_EHIP, "%s: %s", #call, hipGetErrorString(e_));                           \
			pt_group_destroy(g);                                                                       \
			return PT_EHIP;                                                                            \
		}                                                                                              \
	} while (0)
	const size_t full = (size_t) cfg->width * cfg->height * 16;
	for (uint32_t i = 0; i < n_devices; i++) {
		const size_t bytes = mode == PT_GROUP_TILES ? (size_t) pt_partition_count(cfg->width, cfg->height, PT_PARTITION_TILES, i, n_devices) * 16 : full;
		CREATE_TRY(hipSetDevice(devices[i]));
		CREATE_TRY(hipStreamCreate(&g->copy_stream[i]));
		if (devices[i] != devices[0]) {
			// peer access is an optimisation (direct xGMI instead of a bounce through host memory): without it hipMemcpyPeerAsync
			// stages the copy itself, so a failure does not fail the group -- but it is REPORTED: pt_group_last_error of a
			// successfully created group says which pair runs on staged copies and why
			auto enable = [&](int from, int to) {
				int can = 0;
				hipError_t e = hipSetDevice(from);
				if (e == hipSuccess) e = hipDeviceCanAccessPeer(&can, from, to);
				if (e == hipSuccess && can) {
					e = hipDeviceEnablePeerAccess(to, 0);
					if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
				}
				(void) hipGetLastError();
				if (e != hipSuccess || !can) {
					char buf[256];
					snprintf(buf, sizeof buf, "%spt_group_create: no peer access from device %d to device %d (%s): gathers between them use staged copies",
					         g->err.empty() ? "" : "; ", from, to, e != hipSuccess ? hipGetErrorString(e) : "hipDeviceCanAccessPeer says no");
					g->err += buf;
				}
			};
			enable(devices[i], devices[0]);
			enable(devices[0], devices[i]);
			CREATE_TRY(hipSetDevice(devices[i]));
		}
		if (i > 0 && mode == PT_GROUP_TILES) CREATE_TRY(hipMalloc(&g->staging_src[i], bytes ? bytes : 16));
		CREATE_TRY(hipSetDevice(devices[0]));
		if (i > 0 && devices[i] != devices[0]) CREATE_TRY(hipMalloc(&g->staging_dst[i], bytes ? bytes : 16));
	}
	CREATE_TRY(hipSetDevice(devices[0]));
	CREATE_TRY(hipStreamCreate(&g->stream0));
	if (mode == PT_GROUP_SAMPLES) CREATE_TRY(hipMalloc((void **) &g->gathered, full));
#undef CREATE_TRY
	*out = g;
	return PT_OK;
}

uint32_t pt_group_size(const pt_group *g) { return g ? (uint32_t) g->member.size() : 0u; }
pt_ctx *pt_group_member(pt_group *g, uint32_t i) { return g && i < g->member.size() ? g->member[i] : nullptr; }

int pt_group_upload_scene(pt_group *g, const pt_scene_desc *scene)
{
	if (!g || !scene) return PT_EINVAL;
	for (size_t i = 0; i < g->member.size(); i++) G_PT(g, i, pt_upload_scene(g->member[i], scene)); // replicated: a few MB to a few hundred MB per device
	return PT_OK;
}

int pt_group_set_camera(pt_group *g, const float cam[12])
{
	if (!g || !cam) return PT_EINVAL;
	for (size_t i = 0; i < g->member.size(); i++) G_PT(g, i, pt_set_camera(g->member[i], cam));
	return PT_OK;
}

int pt_group_set_estimator(pt_group *g, uint32_t estimator)
{
	if (!g) return PT_EINVAL;
	for (size_t i = 0; i < g->member.size(); i++) G_PT(g, i, pt_set_estimator(g->member[i], estimator));
	return PT_OK;
}

int pt_group_clear(pt_group *g)
{
	if (!g) return PT_EINVAL;
	for (size_t i = 0; i < g->member.size(); i++) {
		G_PT(g, i, pt_clear_framebuffer(g->member[i]));
		g->frames_done[i] = 0;
	}
	g->frames_total = 0;
	return PT_OK;
}

// frames first_frame .. first_frame + n_frames - 1 with seeds[i] per frame, shared between the members:
//   tiles    every member renders every frame for its own 8x8 tiles (same frame numbers, same seeds) -> bit-identical to one device
//   samples  global frame f goes to member f % n over the full image; a member blends its own frames with its LOCAL
//            frame count as weight, so its image is the mean of its own samples
int pt_group_render(pt_group *g, uint32_t first_frame, uint32_t n_frames, const uint32_t *seeds, uint32_t bounces)
{
	if (!g || (n_frames && !seeds)) return PT_EINVAL;
	const uint32_t n = (uint32_t) g->member.size();
	if (g->mode == PT_GROUP_TILES) {
		for (uint32_t i = 0; i < n; i++) G_PT(g, i, pt_render(g->member[i], first_frame, n_frames, seeds, bounces));
	} else {
		std::vector<uint32_t> mine;
		for (uint32_t i = 0; i < n; i++) {
			mine.clear();
			for (uint32_t f = 0; f < n_frames; f++)
				if ((first_frame + f) % n == i) mine.push_back(seeds[f]);
			if (mine.empty()) continue;
			G_PT(g, i, pt_render(g->member[i], (uint32_t) g->frames_done[i], (uint32_t) mine.size(), mine.data(), bounces));
			g->frames_done[i] += mine.size();
		}
	}
	g->frames_total += n_frames;
	return PT_OK;
}

int pt_group_sync(pt_group *g)
{
	if (!g) return PT_EINVAL;
	for (size_t i = 0; i < g->member.size(); i++) G_PT(g, i, pt_sync(g->member[i]));
	return PT_OK;
}

// The finished image on the first device.  tiles: every other member's packed tiles travel to device 0 and are
// unpacked into member 0's framebuffer (whose own tiles are already there).  samples: sum_i (n_i / N) * mean_i into a
// buffer of the group's own -- the members' running means stay untouched, rendering can go on afterwards.
int pt_group_gather(pt_group *g, void **image_dev)
{
	if (!g) return PT_EINVAL;
	const uint32_t n = (uint32_t) g->member.size();
	const uint32_t n_pix = g->width * g->height;
	if (g->mode == PT_GROUP_TILES) {
		for (uint32_t i = 1; i < n; i++) {
			const uint32_t cnt = pt_partition_count(g->width, g->height, PT_PARTITION_TILES, i, n);
			if (cnt == 0) continue;
			G_PT(g, i, pt_pack_owned(g->member[i], g->staging_src[i]));
			G_PT(g, i, pt_sync(g->member[i]));
			const void *src = g->staging_src[i];
			if (g->device[i] != g->device[0]) {
				G_TRY(g, hipSetDevice(g->device[0]));
				G_TRY(g, hipMemcpyPeerAsync(g->staging_dst[i], g->device[0], g->staging_src[i], g->device[i], (size_t) cnt * 16, g->stream0));
				G_TRY(g, hipStreamSynchronize(g->stream0));
				src = g->staging_dst[i];
			}
			G_PT(g, 0, pt_unpack_from(g->member[0], src, i, n));
		}
		G_PT(g, 0, pt_sync(g->member[0]));
		if (image_dev) G_PT(g, 0, pt_framebuffer_device(g->member[0], image_dev));
		return PT_OK;
	}
	uint64_t total = 0;
	for (uint32_t i = 0; i < n; i++) total += g->frames_done[i];
	bool first = true;
	for (uint32_t i = 0; i < n; i++) {
		if (g->frames_done[i] == 0) continue;
		G_PT(g, i, pt_sync(g->member[i]));
		void *fb = nullptr;
		G_PT(g, i, pt_framebuffer_device(g->member[i], &fb));
		const void *src = fb;
		G_TRY(g, hipSetDevice(g->device[0]));
		if (g->device[i] != g->device[0]) {
			G_TRY(g, hipMemcpyPeerAsync(g->staging_dst[i], g->device[0], fb, g->device[i], (size_t) n_pix * 16, g->stream0));
			src = g->staging_dst[i];
		}
		const float w = (float) g->frames_done[i] / (float) (total ? total : 1);
		hipLaunchKernelGGL(k_group_axpy, dim3((n_pix + 255) / 256), dim3(256), 0, g->stream0, (float4 *) g->gathered, (const float4 *) src, w, n_pix, first ? 1 : 0);
		G_TRY(g, hipGetLastError());
		first = false;
	}
	G_TRY(g, hipSetDevice(g->device[0]));
	if (first) G_TRY(g, hipMemsetAsync(g->gathered, 0, (size_t) n_pix * 16, g->stream0));
	G_TRY(g, hipStreamSynchronize(g->stream0));
	if (image_dev) *image_dev = g->gathered;
	return PT_OK;
}

int pt_group_read_framebuffer(pt_group *g, float *rgba, size_t n_floats)
{
	if (!g || !rgba) return PT_EINVAL;
	const size_t need = (size_t) g->width * g->height * 4;
	if (n_floats != need) return gfail(g, PT_EINVAL, "pt_group_read_framebuffer: expected %zu floats, got %zu", need, n_floats);
	void *img = nullptr;
	int rc = pt_group_gather(g, &img);
	if (rc) return rc;
	if (g->mode == PT_GROUP_TILES) { G_PT(g, 0, pt_read_framebuffer(g->member[0], rgba, n_floats)); return PT_OK; }
	G_TRY(g, hipSetDevice(g->device[0]));
	G_TRY(g, hipMemcpy(rgba, img, need * 4, hipMemcpyDeviceToHost));
	return PT_OK;
}

int pt_group_get_counters(pt_group *g, pt_counters *out)
{
	if (!g || !out) return PT_EINVAL;
	memset(out, 0, sizeof *out);
	for (size_t i = 0; i < g->member.size(); i++) {
		pt_counters c;
		G_PT(g, i, pt_get_counters(g->member[i], &c));
		out->paths += c.paths; out->rays_primary += c.rays_primary; out->rays_shadow += c.rays_shadow; out->rays_bounce += c.rays_bounce;
		out->node_steps += c.node_steps; out->tri_tests += c.tri_tests; out->stack_spills += c.stack_spills; out->wave_iters += c.wave_iters; out->top_steps += c.top_steps;
		out->ms_trace += c.ms_trace; out->ms_shade += c.ms_shade; out->trace_launches += c.trace_launches; // sums over devices that ran concurrently
	}
	out->frames = g->frames_total;
	return PT_OK;
}

} // extern "C"
