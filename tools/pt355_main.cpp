// pt355_main -- the reference's main() (src/main.cpp:11-121) with the GL upload + dispatch replaced by libpt355.
// Everything the reference does on the host before the seam keeps its shape: build the model's triangles (here the
// seeded stand-in mesh, the .glb assets being absent offline), CalculateBVH, append the four gold spheres and their
// materials (src/main.cpp:44-51), FindEmissive*, upload the six arrays, set the camera of src/main.cpp:67, then one
// pt_render_frame per displayed frame with FrameData { pcg32_random(), frame_count++, BOUNCE_COUNT, 0 } (src/main.cpp:97).
// Instead of blitting to a window it writes the tonemapped image (shaders/framebuffer.frag) and the float accumulator.
//   usage: pt355_main [frames] [width] [height] [out.ppm] [out.pfm]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/pt355_host.h"

#define BOUNCE_COUNT 5 /* src/defines.hpp:18 */

int main(int argc, char **argv)
{
	const uint32_t frames = argc > 1 ? (uint32_t) atoi(argv[1]) : 64;
	const uint32_t width = argc > 2 ? (uint32_t) atoi(argv[2]) : 1280, height = argc > 3 ? (uint32_t) atoi(argv[3]) : 720; // src/defines.hpp:15-16
	const char *ppm = argc > 4 ? argv[4] : "pt355_main.ppm";
	const char *pfm = argc > 5 ? argv[5] : nullptr;

	// scene build, src/main.cpp:18-55 (the last argument adds the four gold spheres of :44-51)
	pth_scene *scene = pth_scene_create();
	if (pth_scene_make_bunny(scene, 59, 69451, 1, PTH_MAT_OREN_NAYAR, 0.175f, 1)) { fprintf(stderr, "scene build failed\n"); return -1; }
	pt_scene_desc desc;
	pth_scene_desc(scene, &desc);
	printf("Calculated BVH for scene, using %u nodes.\n", desc.n_nodes); // src/scene/bvh.cpp:73

	// Display::Display + InitRenderBuffer + the six PushDataToSSBO calls, src/main.cpp:16, 57-63
	pt_config cfg = {};
	cfg.width = width; cfg.height = height; cfg.partition = PT_PARTITION_FULL;
	pt_ctx *pt = nullptr;
	if (pt_create(&cfg, &pt) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error(nullptr)); return -1; }
	if (pt_upload_scene(pt, &desc) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error(pt)); return -1; }

	// Camera cam(origin 0, forward -z, right +x, 0.005, 0.05), src/main.cpp:67; uploaded when frame_count == 0 (:90-94)
	const float origin[3] = {0, 0, 0}, forward[3] = {0, 0, -1}, right[3] = {1, 0, 0};
	float cam_glsl[12];
	pth_pack_camera(origin, forward, right, 0.005f, 0.05f, cam_glsl);
	pt_set_camera(pt, cam_glsl);

	// the frame loop, src/main.cpp:69-101: one sample per pixel per frame, blended into the running mean
	std::vector<uint32_t> seed(frames);
	pth_frame_seeds(0, frames, seed.data()); // successive pcg32_random() of the default global stream
	for (uint32_t frame_count = 0; frame_count < frames; frame_count++) {
		const uint32_t frame_data[4] = {seed[frame_count], frame_count, BOUNCE_COUNT, 0};
		if (pt_render_frame(pt, frame_data) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error(pt)); return -1; }
	}

	// the blit, src/main.cpp:103-118 + shaders/framebuffer.frag
	std::vector<float> rgba((size_t) width * height * 4);
	if (pt_read_framebuffer(pt, rgba.data(), rgba.size()) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error(pt)); return -1; }
	std::vector<uint8_t> rgb8((size_t) width * height * 3);
	pth_tonemap_rgb8(rgba.data(), width, height, rgb8.data());
	if (pth_write_ppm(ppm, rgb8.data(), width, height)) { fprintf(stderr, "cannot write %s\n", ppm); return -1; }
	if (pfm && pth_write_pfm(pfm, rgba.data(), width, height)) { fprintf(stderr, "cannot write %s\n", pfm); return -1; }
	pt_counters k;
	pt_get_counters(pt, &k);
	printf("%u frames of %ux%u: %.1f M rays, trace %.1f ms, shade %.1f ms -> %s\n", frames, width, height,
	       (double) (k.rays_primary + k.rays_shadow + k.rays_bounce) / 1e6, k.ms_trace, k.ms_shade, ppm);
	pt_destroy(pt);
	pth_scene_destroy(scene);
	return 0;
}
