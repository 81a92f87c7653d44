/* ptbench -- minimal native driver of libpt355.so for profiling runs (rocprofv3 wants the program itself after
 * "--", and a PMC pass should not pay a Python start-up).  Renders the bench.py workload: bunny-class scene,
 * 1920x1080, MIS, 5 bounces.   usage: ptbench [frames] [width] [height] [scene: 0 bunny ON, 1 bunny GGX, 2 cornell, 3 atrium (untextured)] [flags]
 * The warm-up is one full batch (pt_get_info: frames per launch set), so that every launch a profiler sees has the size
 * of the timed ones (give a multiple of the batch as `frames`) and its per-kernel averages are those of the timed region. */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "../include/pt355_host.h"

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
	const uint32_t frames = argc > 1 ? (uint32_t) atoi(argv[1]) : 8;
	const uint32_t w = argc > 2 ? (uint32_t) atoi(argv[2]) : 1920, h = argc > 3 ? (uint32_t) atoi(argv[3]) : 1080;
	const int which = argc > 4 ? atoi(argv[4]) : 0;
	const uint32_t flags = argc > 5 ? (uint32_t) strtoul(argv[5], 0, 0) : 0;
	pth_scene *s = pth_scene_create();
	int rc = which == 3 ? pth_scene_make_atrium(s, 4, 1)
	       : which == 2 ? pth_scene_make_cornell(s, 0)
	       : which == 1 ? pth_scene_make_bunny(s, 59, 69451, 1, PTH_MAT_SPECULAR_METAL, 0.2f, 1)
	                    : pth_scene_make_bunny(s, 59, 69451, 1, PTH_MAT_OREN_NAYAR, 0.175f, 0);
	if (rc) { fprintf(stderr, "scene generator failed: %d\n", rc); return 1; }
	pt_scene_desc d;
	pth_scene_desc(s, &d);
	pt_config cfg = {0};
	cfg.width = w; cfg.height = h; cfg.flags = flags; cfg.partition = PT_PARTITION_FULL; cfg.world = 1;
	pt_ctx *c = 0;
	if (pt_create(&cfg, &c)) { fprintf(stderr, "pt_create: %s\n", pt_last_error(0)); return 1; }
	if (pt_upload_scene(c, &d)) { fprintf(stderr, "upload: %s\n", pt_last_error(c)); return 1; }
	const float o[3] = {0, 0, 0}, f[3] = {0, 0, -1}, r[3] = {1, 0, 0};
	float cam[12];
	pth_pack_camera(o, f, r, 0.005f, 0.05f, cam);
	pt_set_camera(c, cam);
	uint32_t info[8] = {0};
	pt_get_info(c, info);
	const uint32_t warm = info[0] ? info[0] : 16;
	uint32_t *seeds = (uint32_t *) malloc(sizeof(uint32_t) * (frames + warm));
	pth_frame_seeds(0, frames + warm, seeds);
	if (pt_render(c, 0, warm, seeds, 5) || pt_sync(c)) { fprintf(stderr, "render: %s\n", pt_last_error(c)); return 1; }
	pt_reset_counters(c);
	const double t0 = now();
	if (pt_render(c, warm, frames, seeds + warm, 5) || pt_sync(c)) { fprintf(stderr, "render: %s\n", pt_last_error(c)); return 1; }
	const double dt = now() - t0;
	pt_counters k;
	pt_get_counters(c, &k);
	const double rays = (double) (k.rays_primary + k.rays_shadow + k.rays_bounce);
	printf("{\"tris\": %u, \"nodes\": %u, \"frames\": %u, \"ms_per_frame\": %.4f, \"mrays_per_s\": %.1f, \"rays\": %.0f, "
	       "\"ms_trace\": %.3f, \"ms_shade\": %.3f, \"trace_launches\": %llu, \"node_steps\": %llu, \"tri_tests\": %llu, \"spills\": %llu, \"wave_iters\": %llu, \"top_steps\": %llu}\n",
	       d.n_tris, d.n_nodes, frames, dt / frames * 1e3, rays / dt / 1e6, rays, k.ms_trace, k.ms_shade,
	       (unsigned long long) k.trace_launches, (unsigned long long) k.node_steps, (unsigned long long) k.tri_tests,
	       (unsigned long long) k.stack_spills, (unsigned long long) k.wave_iters, (unsigned long long) k.top_steps);
	pt_destroy(c);
	pth_scene_destroy(s);
	free(seeds);
	return 0;
}
