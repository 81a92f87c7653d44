// libpt355.so -- context management and the C ABI of include/pt355.h over the gfx950 kernels.
// One context = one device, scene copies in HBM, one RGBA32F image, and 1..4 frame pipelines: each pipeline has its
// own stream, two ping-pong path queues, control block and sample store, and renders every n-th batch of frames.
// Batches are blended into the image in frame order (an event chain between the pipelines' resolve kernels).
// Default is ONE pipeline: with two, one pipeline's shading does run beside the other's traversal, but both stretch
// (they queue on the same vector-memory pipeline) and the frame time does not improve.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <queue>
#include <vector>

#include "../../include/pt355.h"
#include "../../include/pt355_host.h"
#include "kernels_shade.h"
#include "bvh_device.h"

static_assert(sizeof(pt_config) == 64 && sizeof(pt_scene_desc) == 136 && sizeof(pt_counters) == 128, "C ABI structs changed size: update include/pt355.h, pathtracer_amd/__init__.py and tests/test_abi.py together");

namespace {

thread_local std::string g_create_error;

struct EventPair { hipEvent_t a, b; int kind; };

#define PT_MAX_PIPES 4

struct Pipe {
	hipStream_t stream = nullptr;
	void *queue_alloc = nullptr;
	PathQ q[2];
	DevCtl *ctl = nullptr;
	uint32_t *spill = nullptr;
	float4 *sample = nullptr;
	hipEvent_t resolved = nullptr; // recorded behind this pipeline's latest k_resolve
	std::vector<EventPair> pending;
};

} // namespace

// ---- scratch of the allocation-free trace entry (pt_trace_closest_device_noalloc).  What k_trace needs beyond its
// inputs and outputs -- the control block (work-unit counters, statistics) and the traversal-stack spill tail of every
// wave of a capped grid -- lives in ONE file-scope device array instead of hipMalloc'ed memory: 41,088 B of DevCtl
// (padded to 256-byte granules) + 2,048 waves (PT_NOALLOC_MAX_BLOCKS blocks) x 52 spill levels x 64 lanes x 4 B = 26 MiB.  Every GPU the
// library is loaded on holds its own copy at its own address until the process ends; the array's NAME is a host-side
// stand-in only and is never handed to a kernel or a hip* call: the per-device address from hipGetSymbolAddress is.
#define PT_NOALLOC_MAX_BLOCKS (2048u / PT_TRACE_WAVES) /* 2048 waves = 2 waves per SIMD on 256 CUs: a QUARTER of pt_create's own grid (n_cu * 2 blocks of
                                                          16 waves = 8 waves per SIMD); a spill tail for the full grid would be 104 MiB of static device memory per GPU */
struct NoAllocScratch {
	DevCtl ctl;
	unsigned char pad[(256u - sizeof(DevCtl) % 256u) % 256u];
	uint32_t spill[(size_t) PT_NOALLOC_MAX_BLOCKS * PT_TRACE_WAVES * PT_SPILL_STACK * 64];
};
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];

#define PT_MAX_DEVICES 64
namespace {
std::mutex g_scratch_mutex;
NoAllocScratch *g_scratch_ptr[PT_MAX_DEVICES]; // device address of `scratch` on each device, resolved once with that device current
} // namespace

struct pt_ctx {
	pt_config cfg;
	hipStream_t stream = nullptr; // = pipe[0].stream: uploads, image copies, pack/unpack, the stand-alone trace entry
	std::string err;
	int n_cu = 0;
	int trace_grid = 0;
	bool fused_primary = true;      // MIS: k_raygen folded into the primary k_trace / k_shade (PT355_FUSED_PRIMARY=0 turns it off for A/B runs)
	uint32_t refill_threshold = 32; // idle lanes that trigger a refill: sweep 8..64 on the bench scene peaks at 32 (a refill costs ~3 loop iterations)

	// scene
	DevScene sc{};
	std::vector<void *> scene_allocs;
	bool have_scene = false, have_camera = false;
	uint32_t estimator = PT_EST_MIS;
	float cam[12];

	// queues
	uint32_t n_paths = 0;   // paths per frame = pixels this rank owns
	uint32_t max_batch = 1; // frames traced together (pt_config.max_batch)
	uint32_t stride = 0, cap = 0, blocks_per_region = 0;
	Pipe pipe[PT_MAX_PIPES];
	uint32_t n_pipes = 1, next_pipe = 0;
	int last_resolved = -1;  // pipeline whose k_resolve was enqueued last (the next one waits for it)
	bool in_flight = false;  // renders enqueued since the last join
	float *fb = nullptr;
	NoAllocScratch *noalloc = nullptr; // this device's copy of `scratch` (resolved in pt_create)

	// timing
	std::vector<EventPair> free_events;
	double ms_trace = 0, ms_shade = 0;
	uint64_t trace_launches = 0, frames = 0;
};

namespace {

int fail(pt_ctx *c, int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (c) c->err = buf;
	else g_create_error = buf;
	return code;
}

#define HIP_TRY(c, call)                                                                              \
	do {                                                                                              \
		hipError_t e_ = (call);                                                                       \
		if (e_ != hipSuccess) return fail(c, PT_EHIP, "%s: %s", #call, hipGetErrorString(e_));        \
	} while (0)

uint32_t own_tiles(uint32_t width, uint32_t height, uint32_t partition, uint32_t rank, uint32_t world)
{
	const uint32_t tiles = (width / 8) * (height / 8);
	if (partition != PT_PARTITION_TILES || world <= 1) return tiles;
	return rank < tiles ? (tiles - rank + world - 1) / world : 0;
}

uint32_t host_path_pixel(uint32_t p, uint32_t width, uint32_t partition, uint32_t rank, uint32_t world)
{
	const uint32_t tiles_x = width >> 3;
	uint32_t tile = p >> 6;
	if (partition == PT_PARTITION_TILES && world > 1) tile = tile * world + rank;
	const uint32_t in = p & 63u;
	return ((tile / tiles_x) * 8u + (in >> 3)) * width + (tile % tiles_x) * 8u + (in & 7u);
}

// Profiler ranges around the host-side phases, the counterpart of the reference's glPushDebugGroup("COMPUTE")
// (src/main.cpp:78).  roctx is looked up at run time so that the library carries no link dependency on a profiler
// component; without it (or without a tool attached) a range is two null checks.  rocprofv3 --marker-trace shows them.
struct Range {
	typedef int (*push_fn)(const char *);
	typedef int (*pop_fn)(void);
	static push_fn &push() { static push_fn f = nullptr; return f; }
	static pop_fn &pop() { static pop_fn f = nullptr; return f; }
	static void resolve()
	{
		static std::once_flag once;
		std::call_once(once, [] {
		for (const char *name : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4"}) {
			if (void *h = dlopen(name, RTLD_LAZY | RTLD_LOCAL)) {
				push() = (push_fn) dlsym(h, "roctxRangePushA");
				pop() = (pop_fn) dlsym(h, "roctxRangePop");
				if (push() && pop()) return;
				push() = nullptr; pop() = nullptr;
			}
		}
		});
	}
	bool open = false;
	explicit Range(const char *name) { resolve(); if (push()) { push()(name); open = true; } }
	~Range() { if (open && pop()) pop()(); }
};

// Timing events around a launch.  A failure here only loses a timing sample, never a launch: the pair is dropped and
// the error text kept for pt_last_error (timed = false tells end_timed there is nothing to close).
bool begin_timed(pt_ctx *c, Pipe &pp, int kind)
{
	EventPair ev;
	ev.a = ev.b = nullptr;
	if (!c->free_events.empty()) { ev = c->free_events.back(); c->free_events.pop_back(); }
	else {
		hipError_t e = hipEventCreate(&ev.a);
		if (e == hipSuccess && (e = hipEventCreate(&ev.b)) != hipSuccess) { hipEventDestroy(ev.a); }
		if (e != hipSuccess) { fail(c, PT_EHIP, "hipEventCreate: %s (launch not timed)", hipGetErrorString(e)); return false; }
	}
	ev.kind = kind;
	hipError_t e = hipEventRecord(ev.a, pp.stream);
	if (e != hipSuccess) { c->free_events.push_back(ev); fail(c, PT_EHIP, "hipEventRecord: %s (launch not timed)", hipGetErrorString(e)); return false; }
	pp.pending.push_back(ev);
	return true;
}
void end_timed(pt_ctx *c, Pipe &pp, bool timed)
{
	if (!timed) return;
	hipError_t e = hipEventRecord(pp.pending.back().b, pp.stream);
	if (e != hipSuccess) {
		c->free_events.push_back(pp.pending.back());
		pp.pending.pop_back();
		fail(c, PT_EHIP, "hipEventRecord: %s (launch not timed)", hipGetErrorString(e));
	}
}

// every pipeline idle: the image and the counters are complete
int join(pt_ctx *c)
{
	for (uint32_t i = 0; i < c->n_pipes; i++)
		if (c->pipe[i].stream) HIP_TRY(c, hipStreamSynchronize(c->pipe[i].stream));
	c->in_flight = false;
	return PT_OK;
}

// Kernel times are summed per kind; with two pipelines their kernels overlap, so the sums can exceed wall time.
int drain_events(pt_ctx *c)
{
	int rc = join(c);
	if (rc) return rc;
	for (uint32_t i = 0; i < c->n_pipes; i++) {
		for (EventPair &ev : c->pipe[i].pending) {
			float ms = 0;
			hipEventElapsedTime(&ms, ev.a, ev.b);
			if (ev.kind == 0) c->ms_trace += ms; else c->ms_shade += ms;
			c->free_events.push_back(ev);
		}
		c->pipe[i].pending.clear();
	}
	return PT_OK;
}

template <typename T> int upload(pt_ctx *c, const void *src, size_t bytes, T **dst, size_t pad_front = 0)
{
	*dst = nullptr;
	if (bytes == 0) return PT_OK;
	char *p = nullptr;
	HIP_TRY(c, hipMalloc((void **) &p, bytes + pad_front + 64));
	c->scene_allocs.push_back(p);
	HIP_TRY(c, hipMemsetAsync(p, 0, bytes + pad_front + 64, c->stream));
	HIP_TRY(c, hipMemcpyAsync(p + pad_front, src, bytes, hipMemcpyHostToDevice, c->stream));
	*dst = (T *) (p + pad_front);
	return PT_OK;
}

void free_scene(pt_ctx *c)
{
	for (void *p : c->scene_allocs) hipFree(p);
	c->scene_allocs.clear();
	memset(&c->sc, 0, sizeof c->sc);
	c->have_scene = false;
}

void carve_queue(PathQ &q, char *&p, uint32_t stride)
{
	auto take = [&](uint32_t cols) { void *r = p; p += (size_t) cols * stride * 4; return r; };
	q.pix = (uint32_t *) take(1); q.rng = (uint32_t *) take(1); q.meta = (uint32_t *) take(1);
	q.thr = (float *) take(3); q.col = (float *) take(3);
	q.ray = (float *) take(6); q.sray = (float *) take(6);
	q.nee = (float *) take(8); q.bs = (float *) take(5);
	q.hit = (uint4 *) take(4); q.shit = (uint4 *) take(4);
}
const uint32_t kQueueCols = 3 + 3 + 3 + 6 + 6 + 8 + 5 + 4 + 4;

// the variant of k_trace for a launch: traversal counters, fused primary stage, IEEE division for 1/det (scenes whose
// determinants can leave the range of the short reciprocal, see pt_upload_scene)
typedef void (*trace_kernel_t)(TraceArgs);
trace_kernel_t trace_kernel(bool count, bool gen, bool ieee_rcp)
{
	static const trace_kernel_t table[8] = {
		k_trace<false, false, false>, k_trace<true, false, false>, k_trace<false, true, false>, k_trace<true, true, false>,
		k_trace<false, false, true>,  k_trace<true, false, true>,  k_trace<false, true, true>,  k_trace<true, true, true>,
	};
	return table[(count ? 1 : 0) | (gen ? 2 : 0) | (ieee_rcp ? 4 : 0)];
}

// PT_DEBUG_ITER_CAP builds (kernels_trace.h): wait for the trace launch just enqueued on `stream` and turn abandoned rays
// into an error.  The debug library trades the asynchrony of every trace launch for that; the product build compiles to nothing.
int check_iter_cap(pt_ctx *c, hipStream_t stream, DevCtl *ctl)
{
#ifdef PT_DEBUG_ITER_CAP
	unsigned long long hits = 0;
	HIP_TRY(c, hipMemcpyAsync(&hits, &ctl->iter_cap_hits, sizeof hits, hipMemcpyDeviceToHost, stream));
	HIP_TRY(c, hipStreamSynchronize(stream));
	if (hits) {
		HIP_TRY(c, hipMemsetAsync(&ctl->iter_cap_hits, 0, sizeof hits, stream));
		return fail(c, PT_EHIP, "k_trace: %llu ray(s) exceeded the traversal iteration cap (%u x %u records + 64) and were abandoned: "
		            "the traversal loop does not terminate for them (malformed tree, or a loop that no longer steps)",
		            hits, (unsigned) PT_DEBUG_ITER_CAP, (unsigned) (c->sc.tblob_bytes >> 6));
	}
#else
	(void) c; (void) stream; (void) ctl;
#endif
	return PT_OK;
}

// How a trace launch hands out its rays, from an upper bound of their number (the fill counts themselves live on the device).
// Work-unit size and the dry-region peek only move work between waves: results do not depend on them.
void trace_launch_shape(TraceArgs &ta, uint32_t rays_bound /* paths of the batch, or rays of a trace call */)
{
	// measured on the bench scene, launch sets of 1..32 frames at 720p and 1080p (tools/experiments/exp_small_launch_sweep.sh)
	ta.unit_shift = rays_bound <= (1u << 20) ? 6u : rays_bound <= 3u * (1u << 23) ? 7u : 8u;
	ta.peek = rays_bound <= (1u << 22) ? 1u : 0u;
	if (const char *e = getenv("PT355_EXP_UNIT")) { int v = atoi(e); if (v >= 6 && v <= 9) ta.unit_shift = (uint32_t) v; }
	if (const char *e = getenv("PT355_EXP_PEEK")) ta.peek = atoi(e) != 0;
}

int launch_trace(pt_ctx *c, Pipe &pp, const TraceArgs &ta, bool gen_primary = false)
{
	const bool timed = begin_timed(c, pp, 0);
	const bool count = (c->cfg.flags & PT_FLAG_COUNT_TRAVERSAL) != 0;
	const dim3 grid(c->trace_grid), block(PT_TRACE_BLOCK);
	hipLaunchKernelGGL(trace_kernel(count, gen_primary, ta.sc.short_rcp == 0u), grid, block, 0, pp.stream, ta);
	end_timed(c, pp, timed);
	c->trace_launches++;
	HIP_TRY(c, hipGetLastError());
	return check_iter_cap(c, pp.stream, ta.ctl);
}

TraceArgs queue_trace_args(pt_ctx *c, Pipe &pp, uint32_t qi, bool with_shadow, uint32_t n_paths_bound)
{
	TraceArgs ta;
	memset(&ta, 0, sizeof ta);
	ta.sc = c->sc;
	ta.seg[0].ray = pp.q[qi].ray; ta.seg[0].hit = pp.q[qi].hit; ta.seg[0].valid = nullptr;
	ta.seg[1].ray = pp.q[qi].sray; ta.seg[1].hit = pp.q[qi].shit; ta.seg[1].valid = pp.q[qi].meta;
	ta.seg[1].target = pp.q[qi].nee + 5 * (size_t) c->stride;
	ta.shadow_mode = (c->cfg.flags & PT_FLAG_SHADOW_ANYHIT) ? PT_SHADOW_ANYHIT
	               : (c->cfg.flags & PT_FLAG_SHADOW_FULL_TRAVERSAL) ? PT_SHADOW_FULL : PT_SHADOW_EXACT;
	ta.stride = c->stride;
	ta.cap = c->cap;
	ta.count_ptr = pp.ctl->qcount[qi];
	ta.n_segs = with_shadow ? 2 : 1;
	ta.refill_threshold = c->refill_threshold;
	ta.ctl = pp.ctl;
	ta.zero_on_entry = pp.ctl->qcount[qi ^ 1u];
	ta.spill = pp.spill;
	trace_launch_shape(ta, n_paths_bound);
	return ta;
}

} // namespace

extern "C" {

const char *pt_last_error(const pt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int pt_create(const pt_config *cfg, pt_ctx **out)
{
	if (!cfg || !out) return fail(nullptr, PT_EINVAL, "pt_create: null argument");
	*out = nullptr;
	if (cfg->width == 0 || cfg->height == 0 || (cfg->width & 7u) || (cfg->height & 7u))
		return fail(nullptr, PT_EINVAL, "pt_create: width and height must be positive multiples of 8 (got %ux%u)", cfg->width, cfg->height);
	if ((uint64_t) cfg->width * cfg->height > (1ull << 28))
		return fail(nullptr, PT_EINVAL, "pt_create: image too large");
	if (cfg->partition > PT_PARTITION_FULL) return fail(nullptr, PT_EINVAL, "pt_create: unknown partition %u", cfg->partition);
	const uint32_t world = cfg->world ? cfg->world : 1;
	if (cfg->rank >= world) return fail(nullptr, PT_EINVAL, "pt_create: rank %u outside world %u", cfg->rank, world);

	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev == 0)
		return fail(nullptr, PT_EHIP, "pt_create: no HIP device (%s); this library has no CPU fallback", hipGetErrorString(e));
	if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, PT_EINVAL, "pt_create: device %d of %d", cfg->device, ndev);

	pt_ctx *c = new pt_ctx();
	c->cfg = *cfg;
	c->cfg.world = world;
#define CREATE_TRY(call)                                                                              \
	do {                                                                                              \
		hipError_t e_ = (call);                                                                       \
		if (e_ != hipSuccess) {                                                                       \
			fail(nullptr, PT_EHIP, "%s: %s", #call, hipGetErrorString(e_));                          \
			pt_destroy(c);                                                                            \
			return PT_EHIP;                                                                           \
		}                                                                                             \
	} while (0)
	CREATE_TRY(hipSetDevice(cfg->device));
	if (cfg->device < PT_MAX_DEVICES) {
		// once per device, with that device current: every GPU holds its own copy of `scratch` at its own address
		std::lock_guard<std::mutex> lock(g_scratch_mutex);
		if (!g_scratch_ptr[cfg->device]) {
			void *p = nullptr;
			CREATE_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
			g_scratch_ptr[cfg->device] = (NoAllocScratch *) p;
		}
		c->noalloc = g_scratch_ptr[cfg->device];
	}
	hipDeviceProp_t prop;
	CREATE_TRY(hipGetDeviceProperties(&prop, cfg->device));
	c->n_cu = prop.multiProcessorCount;
	c->n_pipes = cfg->pipelines ? cfg->pipelines : 1u; // 2..4 measured no faster: DESIGN.md "What was tried"
	if (const char *e = getenv("PT355_PIPES")) { int v = atoi(e); if (v >= 1) c->n_pipes = (uint32_t) v; }
	if (c->n_pipes > PT_MAX_PIPES) c->n_pipes = PT_MAX_PIPES;
	for (uint32_t i = 0; i < c->n_pipes; i++) {
		CREATE_TRY(hipStreamCreate(&c->pipe[i].stream));
		CREATE_TRY(hipEventCreateWithFlags(&c->pipe[i].resolved, hipEventDisableTiming));
	}
	c->stream = c->pipe[0].stream;

	int blocks_per_cu = 0;
	CREATE_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_trace<false, false, false>, PT_TRACE_BLOCK, 0));
	if (blocks_per_cu < 1) blocks_per_cu = 1;
	if (blocks_per_cu > 8) blocks_per_cu = 8;
	// tuning knobs for experiments (tools/experiments/exp_*.py), read once per pt_create; defaults are the measured optimum (DESIGN.md
	// section 7): PT355_PIPES (1..4), PT355_TRACE_BLOCKS_PER_CU, PT355_REFILL (1..64), PT355_BATCH (1..32), PT355_FUSED_PRIMARY (0/1).  None changes results.
	if (const char *e = getenv("PT355_TRACE_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= blocks_per_cu) blocks_per_cu = v; }
	if (const char *e = getenv("PT355_FUSED_PRIMARY")) c->fused_primary = atoi(e) != 0;
	if (const char *e = getenv("PT355_REFILL")) { int v = atoi(e); if (v >= 1 && v <= 64) c->refill_threshold = (uint32_t) v; }
	c->trace_grid = c->n_cu * blocks_per_cu;

	c->n_paths = own_tiles(cfg->width, cfg->height, cfg->partition, cfg->rank, world) * 64u;
	// Frames in flight per set of launches.  A trace launch costs ~0.25 ms of ramp-up and tail whatever its size
	// (DESIGN.md "Batching"), so a batch is made as large as a modest memory budget allows: <= 2^26 paths (~23 GB of
	// the 288 GB).  Measured at 1080p: 8 -> 6.89, 16 -> 7.24, 32 -> 7.40 Grays/s (r01e kernel); 16 -> 8.39, 24 -> 8.51,
	// 32 -> 8.59 (r01g kernel).
	c->max_batch = cfg->max_batch ? cfg->max_batch : (uint32_t) PT_MAX_BATCH;
	if (c->max_batch > PT_MAX_BATCH) c->max_batch = PT_MAX_BATCH;
	if (const char *e = getenv("PT355_BATCH")) { int v = atoi(e); if (v >= 1 && v <= PT_MAX_BATCH) c->max_batch = (uint32_t) v; }
	// applied to the override too: k_shade keeps a path's id as (id << 4) | flags in one word, so ids must stay below 2^28
	while (c->max_batch > 1 && (uint64_t) c->max_batch * c->n_paths > (1ull << PT_MAX_PATHS_LOG2)) c->max_batch--;
	// queue geometry: PT_SHARDS regions of `cap` slots, cap a whole number of shading blocks (kernels_trace.h)
	{
		const uint32_t n_blocks = (uint32_t) (((uint64_t) c->max_batch * c->n_paths + PT_SHADE_BLOCK - 1) / PT_SHADE_BLOCK);
		c->blocks_per_region = n_blocks ? (n_blocks + PT_SHARDS - 1) / PT_SHARDS : 1;
		c->cap = c->blocks_per_region * PT_SHADE_BLOCK;
		c->stride = c->cap * PT_SHARDS;
	}
	const size_t qbytes = (size_t) kQueueCols * c->stride * 4;
	{
		// say what is needed before the first hipMalloc fails half way: per pipeline two queues, the sample store and the
		// spill tail; the image once
		const size_t per_pipe = 2 * qbytes + ((size_t) c->max_batch * c->n_paths + 1) * 16 + (size_t) c->trace_grid * PT_TRACE_WAVES * PT_SPILL_STACK * 64 * 4 + sizeof(DevCtl);
		const size_t need = per_pipe * c->n_pipes + (size_t) cfg->width * cfg->height * 16;
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b) {
			fail(nullptr, PT_ENOMEM, "pt_create: %ux%u with %u frames per launch set and %u pipeline(s) needs %.2f GB of device memory, %.2f GB are free; lower pt_config.max_batch",
			     cfg->width, cfg->height, c->max_batch, c->n_pipes, need / 1e9, free_b / 1e9);
			pt_destroy(c);
			return PT_ENOMEM;
		}
	}
	for (uint32_t i = 0; i < c->n_pipes; i++) {
		Pipe &pp = c->pipe[i];
		CREATE_TRY(hipMalloc(&pp.queue_alloc, 2 * qbytes));
		CREATE_TRY(hipMemsetAsync(pp.queue_alloc, 0, 2 * qbytes, pp.stream));
		char *p = (char *) pp.queue_alloc;
		carve_queue(pp.q[0], p, c->stride);
		carve_queue(pp.q[1], p, c->stride);
		CREATE_TRY(hipMalloc((void **) &pp.ctl, sizeof(DevCtl)));
		CREATE_TRY(hipMemsetAsync(pp.ctl, 0, sizeof(DevCtl), pp.stream));
		CREATE_TRY(hipMalloc((void **) &pp.spill, (size_t) c->trace_grid * PT_TRACE_WAVES * PT_SPILL_STACK * 64 * 4));
		CREATE_TRY(hipMalloc((void **) &pp.sample, ((size_t) c->max_batch * c->n_paths + 1) * 16));
	}
	const size_t fb_bytes = (size_t) cfg->width * cfg->height * 16;
	CREATE_TRY(hipMalloc((void **) &c->fb, fb_bytes));
	CREATE_TRY(hipMemsetAsync(c->fb, 0, fb_bytes, c->stream));
	for (uint32_t i = 0; i < c->n_pipes; i++) CREATE_TRY(hipStreamSynchronize(c->pipe[i].stream));
#undef CREATE_TRY
	*out = c;
	return PT_OK;
}

int pt_destroy(pt_ctx *c)
{
	if (!c) return PT_OK;
	hipSetDevice(c->cfg.device);
	for (uint32_t i = 0; i < PT_MAX_PIPES; i++) {
		Pipe &pp = c->pipe[i];
		if (pp.stream) hipStreamSynchronize(pp.stream);
		for (EventPair &ev : pp.pending) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
		if (pp.queue_alloc) hipFree(pp.queue_alloc);
		if (pp.ctl) hipFree(pp.ctl);
		if (pp.spill) hipFree(pp.spill);
		if (pp.sample) hipFree(pp.sample);
		if (pp.resolved) hipEventDestroy(pp.resolved);
	}
	for (EventPair &ev : c->free_events) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
	free_scene(c);
	if (c->fb) hipFree(c->fb);
	for (uint32_t i = 0; i < PT_MAX_PIPES; i++) if (c->pipe[i].stream) hipStreamDestroy(c->pipe[i].stream);
	delete c;
	return PT_OK;
}

// The traversal's own copy of the BVH (kernels_trace.h "Traversal blob"): one 64-byte record per INNER node holding
// its two children (bounds as uploaded, links and triangle counts as integers in bytes), then 64-byte triangle
// records (v0, e1 = v1 - v0, e2 = v2 - v0 -- the same fp32 subtractions fc:273-274 performs -- and the triangle's
// index) in the order the records reference them: the triangles of a left leaf, then those of its right sibling.
// Also proves what the kernel relies on: every link lands inside the blob and every node has at most one parent
// (so the walk from the root is a tree walk and ends).
static int build_traversal_blob(pt_ctx *c, const pt_scene_desc *s, std::vector<uint32_t> &blob, uint32_t *n_records, bool *short_rcp)
{
	// |det| = |e1 . (rd x e2)| <= 2 |rd|_inf |e1|_1 |e2|_1.  The traversal ray is q16(rd) (fc:437), so |rd|_inf <= 65504 whatever
	// the caller hands over: the renderer's directions are normalised, but pt_trace_closest* trace the caller's own rays and
	// nothing normalises those.  The bound takes the fp16 maximum (2^16 for short) and a factor 16 for the roundings: 2^21 |e1|_1
	// |e2|_1.  Still below 2^126 for anything but absurd coordinates (edges of ~2^52).
	double det_bound = 0.0;
	const float *nd = (const float *) s->nodes;
	const float *tr = (const float *) s->tris;
	const uint32_t n = s->n_nodes;
	std::vector<uint32_t> rec_of(n, 0xffffffffu);
	uint32_t n_inner = 0;
	std::vector<uint8_t> claimed(n, 0);
	// the shader starts at node 1 whatever node 0 says (fc:343-350): node 0 is "inner with children 1, 2" when there are any
	auto inner = [&](uint32_t i) { return i == 0 ? n >= 3u : !(nd[8 * i + 7] > 0.0f); };
	auto child = [&](uint32_t i) { return i == 0 ? 1u : (uint32_t) nd[8 * i + 3]; };
	for (uint32_t i = 0; i < n; i++) {
		if (!inner(i)) continue;
		n_inner++;
		const uint32_t ch = child(i);
		for (uint32_t k = 0; k < 2; k++) {
			if (claimed[ch + k]) return fail(c, PT_EINVAL, "pt_upload_scene: node %u is the child of two nodes (not a tree)", ch + k);
			claimed[ch + k] = 1;
		}
	}
	// Record order.  The first PT_TOP_RECORDS records are the part of the tree a ray is most likely to visit -- taken from
	// the root in order of decreasing box area (the SAH's own visit probability) -- because k_trace keeps that prefix of
	// the blob in LDS; the other records follow in node order.  The order is invisible to the traversal: links are byte
	// offsets, and the root's record is first either way.
	{
		uint32_t next = 0;
		if (n_inner) {
			auto area = [&](uint32_t i) {
				if (i == 0) return __builtin_inff();
				const float *b = nd + 8 * (size_t) i;
				const float dx = b[4] - b[0], dy = b[5] - b[1], dz = b[6] - b[2];
				const float a = dx * dy + dy * dz + dz * dx;
				return a == a ? a : 0.0f;
			};
			std::priority_queue<std::pair<float, uint32_t>> heap;
			heap.push({area(0), 0u});
			while (!heap.empty() && next < (uint32_t) PT_TOP_RECORDS) {
				const uint32_t i = heap.top().second;
				heap.pop();
				rec_of[i] = next++;
				for (uint32_t k = 0; k < 2; k++)
					if (inner(child(i) + k)) heap.push({area(child(i) + k), child(i) + k});
			}
		}
		for (uint32_t i = 0; i < n; i++)
			if (inner(i) && rec_of[i] == 0xffffffffu) rec_of[i] = next++;
	}
	// the traversal stack holds one entry per level below the root: 16 in LDS + 48 spilled = the builder's max_depth
	if (n_inner) {
		std::vector<std::pair<uint32_t, uint32_t>> walk{{0u, 1u}};
		while (!walk.empty()) {
			const auto [i, depth] = walk.back();
			walk.pop_back();
			if (depth > PT_SHORT_STACK + PT_SPILL_STACK)
				return fail(c, PT_EINVAL, "pt_upload_scene: BVH deeper than %d levels (the traversal stack's size)", PT_SHORT_STACK + PT_SPILL_STACK);
			for (uint32_t k = 0; k < 2; k++)
				if (inner(child(i) + k)) walk.push_back({child(i) + k, depth + 1u});
		}
	}
	// triangle records, in the order the shader's merged leaf loop walks them (fc:356-368): when the left
	// child is a leaf the loop starts at ITS first triangle and runs over n_left + n_right consecutive triangles of the
	// array -- the right leaf's own triangles only if the builder laid the two ranges out back to back, as the
	// reference's does.  A foreign tree that did not gets exactly what the shader would do with it: the run after the
	// left leaf (entries past the end of the array are never-hit records: the shader's loop reads out of bounds there),
	// and the right leaf's own range as a second copy for the steps that hit only the right leaf.
	std::vector<uint32_t> tt;
	tt.reserve((size_t) s->n_tris * 16u);
	auto f2u = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
	auto emit_range = [&](uint32_t first, uint32_t count) {
		const uint64_t off = (uint64_t) n_inner * 64u + (uint64_t) tt.size() * 4u;
		for (uint64_t t = first; t < (uint64_t) first + count; t++) {
			uint32_t d[16] = {0};
			d[3] = 0xffffffffu;
			if (t < s->n_tris) {
				const float *src = tr + 16 * (size_t) t;
				const float v0[3] = {src[0], src[1], src[2]};
				for (int a = 0; a < 3; a++) {
					const float e1 = src[4 + a] - v0[a], e2 = src[8 + a] - v0[a];
					d[a] = f2u(v0[a]); d[4 + a] = f2u(e1); d[8 + a] = f2u(e2);
				}
				{
					double l1[2] = {0.0, 0.0};
					for (int a = 0; a < 3; a++) { l1[0] += fabs((double) (src[4 + a] - v0[a])); l1[1] += fabs((double) (src[8 + a] - v0[a])); }
					const double b = 16.0 * 2.0 * 65536.0 * l1[0] * l1[1];
					if (!(b <= det_bound)) det_bound = b; // NaN / inf make the bound NaN / inf: the short form is then off
				}
				d[3] = (uint32_t) t;
			}
			tt.insert(tt.end(), d, d + 16);
		}
		return off;
	};
	blob.assign((size_t) n_inner * 16u, 0u);
	for (uint32_t i = 0; i < n; i++) {
		if (rec_of[i] == 0xffffffffu) continue;
		const uint32_t ch = child(i);
		uint32_t cnt[2], first[2];
		uint64_t link[2] = {0, 0};
		for (uint32_t k = 0; k < 2; k++) {
			const float *cn = nd + 8 * (size_t) (ch + k);
			cnt[k] = cn[7] > 0.0f ? (uint32_t) cn[7] : 0u;
			first[k] = (uint32_t) cn[3];
			if (cnt[k] == 0u) link[k] = (uint64_t) rec_of[ch + k] * 64u; // an inner child always has a record
		}
		if (cnt[0]) {
			link[0] = emit_range(first[0], cnt[0] + cnt[1]);
			if (cnt[1]) link[1] = first[1] == first[0] + cnt[0] ? link[0] + (uint64_t) cnt[0] * 64u : emit_range(first[1], cnt[1]);
		} else if (cnt[1]) link[1] = emit_range(first[1], cnt[1]);
		if ((uint64_t) n_inner * 64u + (uint64_t) tt.size() * 4u >= 0xffffff00ull)
			return fail(c, PT_EINVAL, "pt_upload_scene: traversal data exceeds the 32-bit offsets of the trace kernel");
		uint32_t *r = &blob[(size_t) rec_of[i] * 16u];
		for (uint32_t k = 0; k < 2; k++) {
			const float *cn = nd + 8 * (size_t) (ch + k);
			uint32_t *o = r + 8 * k;
			o[0] = f2u(cn[0]); o[1] = f2u(cn[1]); o[2] = f2u(cn[2]); o[3] = (uint32_t) link[k];
			o[4] = f2u(cn[4]); o[5] = f2u(cn[5]); o[6] = f2u(cn[6]); o[7] = cnt[k] * 64u;
		}
	}
	blob.insert(blob.end(), tt.begin(), tt.end());
	*n_records = n_inner;
	*short_rcp = det_bound < 0x1p126;
	return PT_OK;
}

int pt_upload_scene(pt_ctx *c, const pt_scene_desc *s)
{
	if (!c || !s) return PT_EINVAL;
	Range range("pt355 upload_scene");
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	if (!s->tris || s->n_tris == 0 || !s->nodes || s->n_nodes == 0 || !s->materials || s->n_materials == 0)
		return fail(c, PT_EINVAL, "pt_upload_scene: triangles, nodes and materials must be non-empty");
	if (s->n_nodes > (1u << 24) || s->n_tris > (1u << 24))
		return fail(c, PT_EINVAL, "pt_upload_scene: node/triangle indices are stored as floats and stop being exact above 2^24");
	if ((s->n_spheres && !s->spheres) || (s->n_light_tris && !s->light_tris) || (s->n_light_spheres && !s->light_spheres))
		return fail(c, PT_EINVAL, "pt_upload_scene: null array with non-zero count");
	for (uint32_t m : {s->tex_wrap_s, s->tex_wrap_t})
		if (m != 0u && m != 10497u && m != 33071u && m != 33648u) return fail(c, PT_EINVAL, "pt_upload_scene: texture wrap mode %u is not GL_REPEAT, GL_CLAMP_TO_EDGE or GL_MIRRORED_REPEAT", m);
	if (s->tex_mag_filter != 0u && s->tex_mag_filter != 9728u && s->tex_mag_filter != 9729u)
		return fail(c, PT_EINVAL, "pt_upload_scene: texture magnification filter %u is not GL_NEAREST or GL_LINEAR", s->tex_mag_filter);
	if ((s->n_nodes & 1u) == 0) return fail(c, PT_EINVAL, "pt_upload_scene: a sibling-pair BVH has an odd node count (got %u)", s->n_nodes);
	// host-side shape checks so that no kernel can index out of bounds
	{
		const float *nd = (const float *) s->nodes;
		for (uint32_t i = 0; i < s->n_nodes; i++) {
			const float first = nd[8 * i + 3], count = nd[8 * i + 7];
			if (!(first >= 0.0f) || !(count >= 0.0f)) return fail(c, PT_EINVAL, "pt_upload_scene: node %u has a negative/NaN index", i);
			if (count > 0.0f) {
				if ((uint64_t) first + (uint64_t) count > s->n_tris) return fail(c, PT_EINVAL, "pt_upload_scene: leaf %u runs past the triangle array", i);
			} else if (i + 0 < s->n_nodes && s->n_nodes > 1) {
				if ((uint64_t) first + 1 >= s->n_nodes || (uint32_t) first == 0) return fail(c, PT_EINVAL, "pt_upload_scene: node %u has child index %g outside the array", i, (double) first);
			}
		}
		const uint32_t *tr = (const uint32_t *) s->tris;
		for (uint32_t i = 0; i < s->n_tris; i++)
			if (tr[16 * i + 15] >= s->n_materials) return fail(c, PT_EINVAL, "pt_upload_scene: triangle %u uses material %u of %u", i, tr[16 * i + 15], s->n_materials);
		const uint32_t *sp = (const uint32_t *) s->spheres;
		for (uint32_t i = 0; i < s->n_spheres; i++)
			if (sp[8 * i + 4] >= s->n_materials) return fail(c, PT_EINVAL, "pt_upload_scene: sphere %u uses material %u of %u", i, sp[8 * i + 4], s->n_materials);
		for (uint32_t i = 0; i < s->n_light_tris; i++)
			if (s->light_tris[i] >= s->n_tris) return fail(c, PT_EINVAL, "pt_upload_scene: light triangle index out of range");
		for (uint32_t i = 0; i < s->n_light_spheres; i++)
			if (s->light_spheres[i] >= s->n_spheres) return fail(c, PT_EINVAL, "pt_upload_scene: light sphere index out of range");
	}
	{ int rc = join(c); if (rc) return rc; }
	free_scene(c);
	DevScene d;
	memset(&d, 0, sizeof d);
	int rc;
	std::vector<uint32_t> blob;
	uint32_t n_records = 0;
	bool short_rcp = false;
	if ((rc = build_traversal_blob(c, s, blob, &n_records, &short_rcp))) return rc;
	char *tb = nullptr;
	if ((rc = upload(c, blob.data(), blob.size() * 4, &tb))) return rc; // hipMalloc is 256-B aligned: records sit on 64-B lines
	d.tblob = tb;
	d.n_records = n_records;
	d.tblob_bytes = (uint32_t) (blob.size() * 4);
	d.short_rcp = short_rcp && !getenv("PT355_IEEE_RCP") ? 1u : 0u;
	char *tmp = nullptr;
	if ((rc = upload(c, s->tris, (size_t) s->n_tris * 64, &tmp))) return rc;
	d.tris = (const float4 *) tmp;
	if ((rc = upload(c, s->materials, (size_t) s->n_materials * 48, &tmp))) return rc;
	d.mats = (const float4 *) tmp;
	if ((rc = upload(c, s->spheres, (size_t) s->n_spheres * 32, &tmp))) return rc;
	d.spheres = (const float4 *) tmp;
	uint32_t *u = nullptr;
	if ((rc = upload(c, s->light_tris, (size_t) s->n_light_tris * 4, &u))) return rc;
	d.light_tris = u;
	if ((rc = upload(c, s->light_spheres, (size_t) s->n_light_spheres * 4, &u))) return rc;
	d.light_spheres = u;
	uint8_t *b = nullptr;
	if (s->env_rgb8 && s->env_w && s->env_h) {
		if ((rc = upload(c, s->env_rgb8, (size_t) s->env_w * s->env_h * 3, &b))) return rc;
		d.env = b; d.env_w = s->env_w; d.env_h = s->env_h;
	}
	if (s->tex_rgb8 && s->tex_layers) {
		if ((rc = upload(c, s->tex_rgb8, (size_t) s->tex_layers * 512 * 512 * 3, &b))) return rc;
		d.tex = b; d.tex_layers = s->tex_layers;
		d.tex_wrap_s = s->tex_wrap_s; d.tex_wrap_t = s->tex_wrap_t; d.tex_mag_filter = s->tex_mag_filter;
	}
	d.n_nodes = s->n_nodes; d.n_tris = s->n_tris; d.n_mats = s->n_materials; d.n_spheres = s->n_spheres;
	d.n_light_tris = s->n_light_tris; d.n_light_spheres = s->n_light_spheres;
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	c->sc = d;
	c->have_scene = true;
	return PT_OK;
}

int pt_set_camera(pt_ctx *c, const float cam[12])
{
	if (!c || !cam) return PT_EINVAL;
	memcpy(c->cam, cam, sizeof c->cam);
	c->have_camera = true;
	return PT_OK;
}

// n_batch consecutive frames through one set of launches on the next pipeline:
// raygen, (trace, shade) x (bounces + 1), resolve
static int render_batch(pt_ctx *c, uint32_t first_frame, uint32_t n_batch, const uint32_t *seeds, uint32_t bounces)
{
	if (c->n_paths == 0) { c->frames += n_batch; return PT_OK; }
	char range_name[64];
	snprintf(range_name, sizeof range_name, "pt355 batch: frames %u..%u", first_frame, first_frame + n_batch - 1u);
	Range range(range_name);
	const uint32_t pi = c->next_pipe;
	Pipe &pp = c->pipe[pi];
	c->next_pipe = (pi + 1u) % c->n_pipes;
	if (pp.pending.size() > 4096) { int rc = drain_events(c); if (rc) return rc; }
	c->in_flight = true;

	ShadeArgs sa;
	memset(&sa, 0, sizeof sa);
	sa.sc = c->sc;
	sa.stride = c->stride;
	sa.cap = c->cap;
	sa.blocks_per_region = c->blocks_per_region;
	sa.ctl = pp.ctl;
	memcpy(sa.cam, c->cam, sizeof sa.cam);
	sa.width = c->cfg.width; sa.height = c->cfg.height;
	sa.bounces = bounces;
	sa.estimator = c->estimator;
	sa.flags = c->cfg.flags;
	sa.sample = pp.sample;
	sa.fb = c->fb;
	sa.n_batch = n_batch; sa.first_frame = first_frame; sa.n_paths_frame = c->n_paths;
	const uint32_t n_total = n_batch * c->n_paths;
	sa.per_region = (((n_total + 63u) / 64u + PT_SHARDS - 1) / PT_SHARDS) * 64u; // whole 64-ray units, <= cap
	for (uint32_t i = 0; i < n_batch; i++) sa.seeds[i] = seeds[i];
	sa.rank = c->cfg.rank; sa.world = c->cfg.world; sa.partition = c->cfg.partition;
	const dim3 sblock(PT_SHADE_BLOCK);
	// shading blocks that cover the occupied prefix of every region would need a 2-D launch; regions past the batch's
	// own size are empty and their blocks return after one load, so the grid simply covers every slot
	// A region's fill never exceeds the slots of the blocks that append to it.  k_raygen fills region r with <= per_region paths;
	// the fused primary stage runs ceil(n_total / block) blocks of which every 64th appends to region r; a queue-driven stage
	// of 64 * L blocks gives every region exactly L appenders.  So L blocks per region cover every fill of this batch:
	const uint32_t primary_blocks = (n_total + PT_SHADE_BLOCK - 1) / PT_SHADE_BLOCK;
	uint32_t L = (sa.per_region + PT_SHADE_BLOCK - 1) / PT_SHADE_BLOCK;
	if ((primary_blocks + PT_SHARDS - 1) / PT_SHARDS > L) L = (primary_blocks + PT_SHARDS - 1) / PT_SHARDS;
	if (L > c->blocks_per_region) L = c->blocks_per_region; // = cap / block: the bound of the largest batch
	sa.launch_bpr = L;
	const dim3 sgrid(PT_SHARDS * L);

	sa.out = pp.q[0];
	// Fused primary stage (MIS): no k_raygen launch.  k_trace generates the camera ray of path p at refill and k_shade
	// generates it again (same function, same bits) instead of both reading what k_raygen would have written: slots are
	// dense (slot p = path p), hits land in queue 0's hit column.
	const bool fused = c->estimator == PT_EST_MIS && c->fused_primary;
	if (!fused) {
		const bool timed_ = begin_timed(c, pp, 1);
		hipLaunchKernelGGL(k_raygen, dim3((n_total + PT_SHADE_BLOCK - 1) / PT_SHADE_BLOCK), sblock, 0, pp.stream, sa);
		end_timed(c, pp, timed_);
	} else HIP_TRY(c, hipMemsetAsync(pp.ctl->shard, 0, sizeof(pp.ctl->shard), pp.stream)); // k_raygen's other job: fresh work-unit counters
	uint32_t qi = 0;
	if (c->estimator == PT_EST_MIS) {
		for (uint32_t stage = 0; stage <= sa.bounces; stage++) {
			TraceArgs ta = queue_trace_args(c, pp, qi, stage > 0, n_total);
			const bool primary = fused && stage == 0;
			if (primary) {
				ta.seg[0].ray = nullptr;
				ta.count_ptr = nullptr; ta.n_fixed = n_total;
				ta.cap = (((n_total + 63u) / 64u + PT_SHARDS - 1) / PT_SHARDS) * 64u; // whole 64-ray units per region, dense slots
				memcpy(ta.cam, c->cam, sizeof ta.cam);
				ta.width = c->cfg.width; ta.height = c->cfg.height; ta.n_paths_frame = c->n_paths;
				ta.partition = c->cfg.partition; ta.rank = c->cfg.rank; ta.world = c->cfg.world;
				for (uint32_t i = 0; i < n_batch; i++) ta.seeds[i] = seeds[i];
			}
			int rc = launch_trace(c, pp, ta, primary);
			if (rc) return rc;
			sa.in = pp.q[qi]; sa.out = pp.q[qi ^ 1u]; sa.in_q = qi; sa.stage = stage;
			const bool timed_ = begin_timed(c, pp, 1);
			if (primary) hipLaunchKernelGGL(k_shade<true>, dim3((n_total + PT_SHADE_BLOCK - 1) / PT_SHADE_BLOCK), sblock, 0, pp.stream, sa);
			else hipLaunchKernelGGL(k_shade<false>, sgrid, sblock, 0, pp.stream, sa);
			end_timed(c, pp, timed_);
			qi ^= 1u;
		}
	} else {
		// fc:647-693 / fc:725-867: `bounces` intersect() calls per path; the NEE estimator's last shadow ray needs one
		// more (shadow-only) trace and a resolve-only stage
		const bool nee = c->estimator == PT_EST_NEE;
		const uint32_t n_stages = sa.bounces ? sa.bounces + (nee ? 1u : 0u) : 0u;
		for (uint32_t stage = 0; stage < n_stages; stage++) {
			TraceArgs ta = queue_trace_args(c, pp, qi, nee && stage > 0, n_total);
			ta.shadow_mode = PT_SHADOW_FULL; // this estimator's compare is two-sided (fc:821): no early-out rule built for it
			if (nee && stage == sa.bounces) { ta.seg[0] = ta.seg[1]; ta.n_segs = 1; } // only shadow rays are left
			int rc = launch_trace(c, pp, ta);
			if (rc) return rc;
			sa.in = pp.q[qi]; sa.out = pp.q[qi ^ 1u]; sa.in_q = qi; sa.stage = stage;
			const bool timed_ = begin_timed(c, pp, 1);
			hipLaunchKernelGGL(k_shade_alt, sgrid, sblock, 0, pp.stream, sa);
			end_timed(c, pp, timed_);
			qi ^= 1u;
		}
	}
	// the running mean is order-dependent: this batch is blended after the previous one, whichever pipeline ran it
	if (c->last_resolved >= 0 && (uint32_t) c->last_resolved != pi)
		HIP_TRY(c, hipStreamWaitEvent(pp.stream, c->pipe[c->last_resolved].resolved, 0));
	const bool timed_r = begin_timed(c, pp, 1);
	hipLaunchKernelGGL(k_resolve, dim3((c->n_paths + PT_SHADE_BLOCK - 1) / PT_SHADE_BLOCK), sblock, 0, pp.stream, sa);
	end_timed(c, pp, timed_r);
	HIP_TRY(c, hipEventRecord(pp.resolved, pp.stream));
	c->last_resolved = (int) pi;
	HIP_TRY(c, hipGetLastError());
	c->frames += n_batch;
	return PT_OK;
}

// image written on pipeline 0's stream outside a render (clear, unpack): later blends must come after it
static int image_touched(pt_ctx *c)
{
	HIP_TRY(c, hipEventRecord(c->pipe[0].resolved, c->stream));
	c->last_resolved = 0;
	return PT_OK;
}

int pt_set_estimator(pt_ctx *c, uint32_t estimator)
{
	if (!c) return PT_EINVAL;
	if (estimator > PT_EST_BRDF) return fail(c, PT_EINVAL, "pt_set_estimator: unknown estimator %u", estimator);
	c->estimator = estimator;
	return PT_OK;
}

int pt_render_frame(pt_ctx *c, const uint32_t frame_data[4])
{
	if (!c || !frame_data) return PT_EINVAL;
	if (!c->have_scene || !c->have_camera) return fail(c, PT_ESTATE, "pt_render_frame: upload a scene and set the camera first");
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	return render_batch(c, frame_data[1], 1, &frame_data[0], frame_data[2]);
}

int pt_render(pt_ctx *c, uint32_t first_frame, uint32_t n_frames, const uint32_t *seeds, uint32_t bounces)
{
	if (!c || (n_frames && !seeds)) return PT_EINVAL;
	if (!c->have_scene || !c->have_camera) return fail(c, PT_ESTATE, "pt_render: upload a scene and set the camera first");
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	for (uint32_t f = 0; f < n_frames;) {
		const uint32_t nb = n_frames - f < c->max_batch ? n_frames - f : c->max_batch;
		int rc = render_batch(c, first_frame + f, nb, seeds + f, bounces);
		if (rc) return rc;
		f += nb;
	}
	return PT_OK;
}

int pt_sync(pt_ctx *c)
{
	if (!c) return PT_EINVAL;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	return join(c);
}

int pt_read_framebuffer(pt_ctx *c, float *rgba, size_t n_floats)
{
	if (!c || !rgba) return PT_EINVAL;
	const size_t need = (size_t) c->cfg.width * c->cfg.height * 4;
	if (n_floats != need) return fail(c, PT_EINVAL, "pt_read_framebuffer: expected %zu floats, got %zu", need, n_floats);
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	{ int rc = join(c); if (rc) return rc; }
	HIP_TRY(c, hipMemcpyAsync(rgba, c->fb, need * 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return PT_OK;
}

int pt_clear_framebuffer(pt_ctx *c)
{
	if (!c) return PT_EINVAL;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	{ int rc = join(c); if (rc) return rc; }
	HIP_TRY(c, hipMemsetAsync(c->fb, 0, (size_t) c->cfg.width * c->cfg.height * 16, c->stream));
	return image_touched(c);
}

int pt_write_framebuffer(pt_ctx *c, const float *rgba, size_t n_floats)
{
	if (!c || !rgba) return PT_EINVAL;
	const size_t need = (size_t) c->cfg.width * c->cfg.height * 4;
	if (n_floats != need) return fail(c, PT_EINVAL, "pt_write_framebuffer: expected %zu floats, got %zu", need, n_floats);
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	{ int rc = join(c); if (rc) return rc; }
	HIP_TRY(c, hipMemcpyAsync(c->fb, rgba, need * 4, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream)); // the caller's buffer may go away
	return image_touched(c);
}

int pt_framebuffer_device(pt_ctx *c, void **dev_ptr)
{
	if (!c || !dev_ptr) return PT_EINVAL;
	*dev_ptr = c->fb;
	return PT_OK;
}

int pt_trace_closest_device(pt_ctx *c, const void *ray6_dev, uint32_t n, void *hit4_dev)
{
	if (!c || !ray6_dev || !hit4_dev) return PT_EINVAL;
	if (!c->have_scene) return fail(c, PT_ESTATE, "pt_trace_closest: upload a scene first");
	if (n == 0) return PT_OK;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	if (c->in_flight) { int rc = join(c); if (rc) return rc; } // borrows pipeline 0's control block and spill tail
	Pipe &pp = c->pipe[0];
	HIP_TRY(c, hipMemsetAsync(pp.ctl->shard, 0, sizeof(pp.ctl->shard), c->stream));
	TraceArgs ta;
	memset(&ta, 0, sizeof ta);
	ta.sc = c->sc;
	ta.seg[0].ray = (const float *) ray6_dev; ta.seg[0].hit = (uint4 *) hit4_dev;
	ta.stride = n; ta.n_fixed = n; ta.n_segs = 1;
	ta.cap = (((n + 63u) / 64u + PT_SHARDS - 1) / PT_SHARDS) * 64u; // whole 64-ray units per region
	ta.refill_threshold = c->refill_threshold;
	ta.ctl = pp.ctl;
	ta.spill = pp.spill;
	trace_launch_shape(ta, n);
	return launch_trace(c, pp, ta);
}

// Same kernel, same results as pt_trace_closest_device, for callers that cannot allocate (HIP graph capture, tight
// serving loops): control block and spill tail come from this device's copy of the file-scope `scratch` array, no
// timing events are taken (the event pool may have to grow), and nothing on this path calls hipMalloc/hipFree.
// The array is shared by every call on the device and keeps the last call's contents: the control block is cleared
// on the stream in front of the launch (work-unit counters must start at zero; the spill tail is written before it is
// read), and calls that overlap on one GPU -- from this or any other context of the process -- race on it.
int pt_trace_closest_device_noalloc(pt_ctx *c, const void *ray6_dev, uint32_t n, void *hit4_dev)
{
	if (!c || !ray6_dev || !hit4_dev) return PT_EINVAL;
	if (!c->have_scene) return fail(c, PT_ESTATE, "pt_trace_closest_device_noalloc: upload a scene first");
	if (n > PT_NOALLOC_MAX_RAYS)
		return fail(c, PT_EINVAL, "pt_trace_closest_device_noalloc: %u rays exceed the entry point's maximum of %u per call", n, (unsigned) PT_NOALLOC_MAX_RAYS);
	if (!c->noalloc) return fail(c, PT_ESTATE, "pt_trace_closest_device_noalloc: no static scratch on device %d (ordinal above %d)", c->cfg.device, PT_MAX_DEVICES - 1);
	if (n == 0) return PT_OK;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	// no implicit join here: a stream synchronisation is not something a caller under graph capture can take, and renders in
	// flight own the streams this call would have to wait for.  pt_sync first.
	if (c->in_flight) return fail(c, PT_ESTATE, "pt_trace_closest_device_noalloc: renders are in flight on this context: call pt_sync first (this entry point never synchronises)");
	NoAllocScratch *const sp = c->noalloc; // the device address, never the array's name
	HIP_TRY(c, hipMemsetAsync(&sp->ctl, 0, sizeof(DevCtl), c->stream));
	TraceArgs ta;
	memset(&ta, 0, sizeof ta);
	ta.sc = c->sc;
	ta.seg[0].ray = (const float *) ray6_dev; ta.seg[0].hit = (uint4 *) hit4_dev;
	ta.stride = n; ta.n_fixed = n; ta.n_segs = 1;
	ta.cap = (((n + 63u) / 64u + PT_SHARDS - 1) / PT_SHARDS) * 64u; // whole 64-ray units per region
	ta.refill_threshold = c->refill_threshold;
	ta.ctl = &sp->ctl;
	ta.spill = sp->spill;
	trace_launch_shape(ta, n);
	// the spill tail is sized for PT_NOALLOC_MAX_BLOCKS blocks: k_trace is persistent, any grid drains the work counters
	const uint32_t grid = (uint32_t) c->trace_grid < PT_NOALLOC_MAX_BLOCKS ? (uint32_t) c->trace_grid : PT_NOALLOC_MAX_BLOCKS;
	hipLaunchKernelGGL(trace_kernel(false, false, ta.sc.short_rcp == 0u), dim3(grid), dim3(PT_TRACE_BLOCK), 0, c->stream, ta);
	HIP_TRY(c, hipGetLastError());
	return check_iter_cap(c, c->stream, ta.ctl); // debug builds only; allocates nothing either way
}

int pt_trace_closest(pt_ctx *c, const float *ro, const float *rd, uint32_t n, uint32_t *prim, uint16_t *t_h, uint8_t *kind, float *uv)
{
	if (!c || (n && (!ro || !rd || !prim || !t_h || !kind))) return PT_EINVAL;
	if (!c->have_scene) return fail(c, PT_ESTATE, "pt_trace_closest: upload a scene first");
	if (n == 0) return PT_OK;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	std::vector<float> soa((size_t) 6 * n);
	for (uint32_t i = 0; i < n; i++)
		for (int k = 0; k < 3; k++) { soa[(size_t) k * n + i] = ro[3 * (size_t) i + k]; soa[(size_t) (3 + k) * n + i] = rd[3 * (size_t) i + k]; }
	float *d_ray = nullptr; uint32_t *d_hit = nullptr;
	HIP_TRY(c, hipMalloc((void **) &d_ray, (size_t) 6 * n * 4));
	hipError_t e = hipMalloc((void **) &d_hit, (size_t) 4 * n * 4);
	if (e != hipSuccess) { hipFree(d_ray); return fail(c, PT_EHIP, "hipMalloc: %s", hipGetErrorString(e)); }
	std::vector<uint32_t> hit((size_t) 4 * n);
	int rc = PT_OK;
	do {
		if ((e = hipMemcpyAsync(d_ray, soa.data(), soa.size() * 4, hipMemcpyHostToDevice, c->stream)) != hipSuccess) break;
		if ((rc = pt_trace_closest_device(c, d_ray, n, d_hit))) break;
		if ((e = hipMemcpyAsync(hit.data(), d_hit, hit.size() * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) break;
		e = hipStreamSynchronize(c->stream);
	} while (0);
	hipFree(d_ray); hipFree(d_hit);
	if (rc) return rc;
	if (e != hipSuccess) return fail(c, PT_EHIP, "pt_trace_closest: %s", hipGetErrorString(e));
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t *h = &hit[4 * (size_t) i];
		const uint32_t tk = h[1];
		kind[i] = (uint8_t) (tk >> 16);
		prim[i] = kind[i] ? h[0] : 0xffffffffu;
		t_h[i] = kind[i] ? (uint16_t) (tk & 0xffffu) : 0;
		if (uv) {
			memcpy(&uv[2 * (size_t) i], &h[2], 4);
			memcpy(&uv[2 * (size_t) i + 1], &h[3], 4);
			if (kind[i] == 2) { // a sphere's record holds HitData.uvs (2 x fp16 bits, kept from the last accepted triangle) in its third word
				uv[2 * (size_t) i] = (float) __builtin_bit_cast(_Float16, (uint16_t) (h[2] & 0xffffu));
				uv[2 * (size_t) i + 1] = (float) __builtin_bit_cast(_Float16, (uint16_t) (h[2] >> 16));
			} else if (kind[i] != 1) uv[2 * (size_t) i] = uv[2 * (size_t) i + 1] = 0.0f;
		}
	}
	return PT_OK;
}

int pt_get_info(const pt_ctx *c, uint32_t info[8])
{
	if (!c || !info) return PT_EINVAL;
	memset(info, 0, 8 * sizeof(uint32_t));
	info[0] = c->max_batch; info[1] = c->n_pipes; info[2] = c->stride; info[3] = c->cap; info[4] = c->n_paths;
	info[5] = (uint32_t) c->trace_grid; info[6] = PT_TRACE_BLOCK; info[7] = PT_TOP_BYTES;
	return PT_OK;
}

int pt_get_counters(pt_ctx *c, pt_counters *out)
{
	if (!c || !out) return PT_EINVAL;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	int rc = drain_events(c);
	if (rc) return rc;
	memset(out, 0, sizeof *out);
	out->frames = c->frames;
	for (uint32_t i = 0; i < c->n_pipes; i++) {
		static thread_local DevCtl h;
		HIP_TRY(c, hipMemcpyAsync(&h, c->pipe[i].ctl, sizeof h, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		out->paths += h.paths; out->rays_primary += h.rays_primary;
		for (uint32_t r = 0; r < PT_SHARDS; r++) {
			out->rays_shadow += h.rays_shadow[r * (PT_SHARD_STRIDE / 2)];
			out->rays_bounce += h.rays_bounce[r * (PT_SHARD_STRIDE / 2)];
		}
		out->node_steps += h.node_steps; out->tri_tests += h.tri_tests; out->stack_spills += h.stack_spills;
		out->wave_iters += h.wave_iters; out->top_steps += h.top_steps;
	}
	out->ms_trace = c->ms_trace; out->ms_shade = c->ms_shade; out->trace_launches = c->trace_launches;
	return PT_OK;
}

int pt_reset_counters(pt_ctx *c)
{
	if (!c) return PT_EINVAL;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	int rc = drain_events(c);
	if (rc) return rc;
	for (uint32_t i = 0; i < c->n_pipes; i++) HIP_TRY(c, hipMemsetAsync(c->pipe[i].ctl, 0, sizeof(DevCtl), c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	c->ms_trace = c->ms_shade = 0; c->trace_launches = 0; c->frames = 0;
	return PT_OK;
}

uint32_t pt_partition_count(uint32_t width, uint32_t height, uint32_t partition, uint32_t rank, uint32_t world)
{
	if (world == 0) world = 1;
	if ((width & 7u) || (height & 7u) || rank >= world) return 0;
	return own_tiles(width, height, partition, rank, world) * 64u;
}

int pt_partition_pixels(uint32_t width, uint32_t height, uint32_t partition, uint32_t rank, uint32_t world, uint32_t *pixels)
{
	if (world == 0) world = 1;
	if (!pixels || (width & 7u) || (height & 7u) || rank >= world) return PT_EINVAL;
	const uint32_t n = own_tiles(width, height, partition, rank, world) * 64u;
	for (uint32_t p = 0; p < n; p++) pixels[p] = host_path_pixel(p, width, partition, rank, world);
	return PT_OK;
}

int pt_pack_owned(pt_ctx *c, void *dst_dev)
{
	if (!c || !dst_dev) return PT_EINVAL;
	if (c->n_paths == 0) return PT_OK;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	if (c->in_flight) { int rc = join(c); if (rc) return rc; }
	hipLaunchKernelGGL(k_pack_owned, dim3((c->n_paths + 255) / 256), dim3(256), 0, c->stream, (const float4 *) c->fb, (float4 *) dst_dev,
	                   c->n_paths, c->cfg.width, c->cfg.partition, c->cfg.rank, c->cfg.world);
	HIP_TRY(c, hipGetLastError());
	return PT_OK;
}

int pt_unpack_from(pt_ctx *c, const void *src_dev, uint32_t from_rank, uint32_t world)
{
	if (!c || !src_dev) return PT_EINVAL;
	if (world == 0) world = 1;
	if (from_rank >= world) return fail(c, PT_EINVAL, "pt_unpack_from: rank %u outside world %u", from_rank, world);
	const uint32_t n = own_tiles(c->cfg.width, c->cfg.height, PT_PARTITION_TILES, from_rank, world) * 64u;
	if (n == 0) return PT_OK;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	if (c->in_flight) { int rc = join(c); if (rc) return rc; }
	hipLaunchKernelGGL(k_unpack_from, dim3((n + 255) / 256), dim3(256), 0, c->stream, (float4 *) c->fb, (const float4 *) src_dev, n,
	                   c->cfg.width, (uint32_t) PT_PARTITION_TILES, from_rank, world);
	HIP_TRY(c, hipGetLastError());
	return image_touched(c);
}

int pt_debug_read_queue(pt_ctx *c, uint32_t queue, uint32_t field, void *out, size_t n_words, uint32_t *stride, uint32_t *count)
{
	if (!c || queue > 1 || field > 10 || !out) return PT_EINVAL;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	{ int rc = join(c); if (rc) return rc; }
	const PathQ &q = c->pipe[0].q[queue];
	const void *base[11] = {q.pix, q.rng, q.meta, q.thr, q.col, q.ray, q.sray, q.nee, q.bs, q.hit, q.shit};
	const uint32_t cols[11] = {1, 1, 1, 3, 3, 6, 6, 8, 5, 4, 4};
	const size_t need = (size_t) cols[field] * c->stride;
	if (n_words != need) return fail(c, PT_EINVAL, "pt_debug_read_queue: field %u needs %zu words", field, need);
	HIP_TRY(c, hipMemcpyAsync(out, base[field], need * 4, hipMemcpyDeviceToHost, c->stream));
	std::vector<uint32_t> qc(PT_SHARDS * PT_SHARD_STRIDE);
	HIP_TRY(c, hipMemcpyAsync(qc.data(), c->pipe[0].ctl->qcount[queue], qc.size() * 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	if (stride) *stride = c->stride;
	if (count) { *count = 0; for (uint32_t r = 0; r < PT_SHARDS; r++) *count += qc[r * PT_SHARD_STRIDE]; }
	return PT_OK;
}

int pt_build_bvh_device(pt_ctx *c, const void *tris, uint32_t n, uint32_t max_leaf, void *nodes_out, uint32_t *n_nodes_out,
                        void *sorted_out, float *build_ms)
{
	if (!c) return PT_EINVAL;
	Range range("pt355 build_bvh_device");
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	char err[256] = "";
	const int method = (max_leaf & PT_BVH_PLOC) ? 1 : 0;
	max_leaf &= ~(uint32_t) PT_BVH_PLOC;
	const int rc = ptb_build_bvh(c->stream, method, tris, n, max_leaf ? max_leaf : 2u, nodes_out, n_nodes_out, sorted_out, build_ms, err, sizeof err);
	if (rc) return fail(c, rc == 1 ? PT_EINVAL : PT_EHIP, "%s", err);
	return PT_OK;
}

static uint32_t scene_builder_thunk(const void *tris, uint32_t n, void *nodes_out, void *sorted_out, void *user)
{
	void **u = (void **) user;
	uint32_t nn = 0;
	const int rc = pt_build_bvh_device((pt_ctx *) u[0], tris, n, (uint32_t) (uintptr_t) u[1], nodes_out, &nn, sorted_out, (float *) u[2]);
	return rc == PT_OK ? nn : 0u;
}

int pt_scene_finalize_device(pt_ctx *c, pth_scene *s, uint32_t max_leaf, float *build_ms)
{
	if (!c || !s) return PT_EINVAL;
	void *user[3] = {c, (void *) (uintptr_t) max_leaf, build_ms};
	const int rc = pth_scene_finalize_with(s, scene_builder_thunk, user);
	if (rc != PT_OK && c->err.empty()) return fail(c, rc, "pt_scene_finalize_device: the scene could not be finalized");
	return rc;
}

int pt_debug_eval(pt_ctx *c, int op, const float *a, const float *b, uint32_t n, float *out)
{
	if (!c || !a || !out) return PT_EINVAL;
	if (n == 0) return PT_OK;
	HIP_TRY(c, hipSetDevice(c->cfg.device));
	float *da = nullptr, *db = nullptr, *dout = nullptr;
	hipError_t e = hipMalloc((void **) &da, (size_t) n * 4);
	if (e == hipSuccess) e = hipMalloc((void **) &dout, (size_t) n * 4);
	if (e == hipSuccess && b) e = hipMalloc((void **) &db, (size_t) n * 4);
	if (e == hipSuccess) e = hipMemcpyAsync(da, a, (size_t) n * 4, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess && b) e = hipMemcpyAsync(db, b, (size_t) n * 4, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_debug_eval, dim3((n + 255) / 256), dim3(256), 0, c->stream, op, (const float *) da, (const float *) db, n, dout);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t) n * 4, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (da) hipFree(da);
	if (dout) hipFree(dout);
	if (db) hipFree(db);
	if (e != hipSuccess) return fail(c, PT_EHIP, "pt_debug_eval: %s", hipGetErrorString(e));
	return PT_OK;
}

} // extern "C"
