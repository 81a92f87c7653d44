// pt355_main -- the reference's main() (src/main.cpp:11-121) with the GL upload + dispatch replaced by libpt355.
// Everything the reference does on the host before the seam keeps its shape: LoadGLTF + the model set-up of
// src/main.cpp:18-29 when a model file is given (the reference hard-codes res/models/CornellBox_lit.glb, which like all
// its assets is absent offline; without a file the seeded stand-in mesh is built instead), CalculateBVH, the four gold
// spheres and their materials (src/main.cpp:44-51), FindEmissive*, upload of the six arrays (and the texture array),
// the camera of src/main.cpp:67, then one pt_render_frame per displayed frame with FrameData { pcg32_random(),
// frame_count++, BOUNCE_COUNT, 0 } (src/main.cpp:97).
// Instead of blitting to a window it writes the tonemapped image (shaders/framebuffer.frag) and the float accumulator.
//   usage: pt355_main [frames] [width] [height] [out.ppm] [out.pfm] [model.glb | -] [devices, e.g. 0,1,2,3] [tiles | samples]
// With a device list the same sequence runs on a pt_group: the scene is replicated, the image partitioned over the
// devices (8x8 tiles by default) and gathered onto the first one for the read-back.
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

	const char *model_path = argc > 6 && argv[6][0] != '-' ? argv[6] : nullptr;
	std::vector<int32_t> devices;
	if (argc > 7)
		for (const char *p = argv[7]; *p;) { devices.push_back((int32_t) strtol(p, (char **) &p, 10)); if (*p == ',') p++; }
	const uint32_t group_mode = argc > 8 && argv[8][0] == 's' ? PT_GROUP_SAMPLES : PT_GROUP_TILES;

	// scene build, src/main.cpp:18-55
	pth_scene *scene = pth_scene_create();
	pth_model *model = nullptr;
	if (model_path) {
		model = pth_model_create();
		if (pth_load_gltf(model_path, model)) { printf("Failed to load model!\n%s\n", pth_last_error()); return -1; } // src/main.cpp:18-23
		const float t[3] = {0.0f, -2.0f, -6.0f}, r[3] = {0.0f, -90.0f * 0.01745329251994329576923690768489f, 0.0f}, s[3] = {2.0f, 2.0f, 2.0f};
		pth_model_translate(model, t); // src/main.cpp:26-29
		pth_model_rotate(model, r);
		pth_model_scale(model, s);
		pth_model_apply_matrix(model);
		pth_scene_add_model(scene, model); // ConvertToSSBOFormat happens in pth_scene_finalize; materials_ssbo = model.materials (:36)
		const float gold[3] = {0.944f, 0.776f, 0.373f}, zero[3] = {0, 0, 0}, rough[4] = {0.0f, 0.1f, 0.15f, 0.2f}, cx[4] = {-1.0f, -0.4f, 0.2f, 0.8f};
		for (int i = 0; i < 4; i++) { // src/main.cpp:44-51
			const int m = pth_scene_add_material(scene, zero, gold, zero, rough[i], -1, PTH_MAT_SPECULAR_METAL);
			const float o[3] = {cx[i], 1.0f, -5.0f};
			pth_scene_add_sphere(scene, o, 0.3f, (uint32_t) m);
		}
		if (pth_scene_finalize(scene)) { fprintf(stderr, "scene build failed\n"); return -1; }
	} else if (pth_scene_make_bunny(scene, 59, 69451, 1, PTH_MAT_OREN_NAYAR, 0.175f, 1)) { fprintf(stderr, "scene build failed\n"); return -1; } // last argument: the gold spheres
	pt_scene_desc desc;
	pth_scene_desc(scene, &desc);
	if (model) { // the texture array bound to unit 2, src/main.cpp:83-88, with the sampler state the loader set
		uint32_t smp[3];
		pth_model_textures(model, &desc.tex_rgb8, &desc.tex_layers, smp);
		desc.tex_wrap_s = smp[0]; desc.tex_wrap_t = smp[1]; desc.tex_mag_filter = smp[2];
	}
	printf("Calculated BVH for scene, using %u nodes.\n", desc.n_nodes); // src/scene/bvh.cpp:73

	// Display::Display + InitRenderBuffer + the six PushDataToSSBO calls, src/main.cpp:16, 57-63
	pt_config cfg = {};
	cfg.width = width; cfg.height = height; cfg.partition = PT_PARTITION_FULL;
	const float origin[3] = {0, 0, 0}, forward[3] = {0, 0, -1}, right[3] = {1, 0, 0};
	float cam_glsl[12];
	pth_pack_camera(origin, forward, right, 0.005f, 0.05f, cam_glsl);
	std::vector<uint32_t> seed(frames);
	pth_frame_seeds(0, frames, seed.data()); // successive pcg32_random() of the default global stream
	std::vector<float> rgba((size_t) width * height * 4);
	if (!devices.empty()) {
		// the same sequence on several devices at once (no counterpart in the reference)
		pt_group *grp = nullptr;
		if (pt_group_create(&cfg, devices.data(), (uint32_t) devices.size(), group_mode, &grp) != PT_OK) { fprintf(stderr, "%s\n", pt_group_last_error(nullptr)); return -1; }
		if (pt_group_upload_scene(grp, &desc) != PT_OK || pt_group_set_camera(grp, cam_glsl) != PT_OK) { fprintf(stderr, "%s\n", pt_group_last_error(grp)); return -1; }
		if (pt_group_render(grp, 0, frames, seed.data(), BOUNCE_COUNT) != PT_OK) { fprintf(stderr, "%s\n", pt_group_last_error(grp)); return -1; }
		if (pt_group_read_framebuffer(grp, rgba.data(), rgba.size()) != PT_OK) { fprintf(stderr, "%s\n", pt_group_last_error(grp)); return -1; }
		pt_counters k;
		pt_group_get_counters(grp, &k);
		std::vector<uint8_t> rgb8((size_t) width * height * 3);
		pth_tonemap_rgb8(rgba.data(), width, height, rgb8.data());
		if (pth_write_ppm(ppm, rgb8.data(), width, height)) { fprintf(stderr, "cannot write %s\n", ppm); return -1; }
		if (pfm && pth_write_pfm(pfm, rgba.data(), width, height)) { fprintf(stderr, "cannot write %s\n", pfm); return -1; }
		printf("%u frames of %ux%u on %zu devices (%s): %.1f M rays -> %s\n", frames, width, height, devices.size(), group_mode == PT_GROUP_TILES ? "tiles" : "samples",
		       (double) (k.rays_primary + k.rays_shadow + k.rays_bounce) / 1e6, ppm);
		pt_group_destroy(grp);
		pth_scene_destroy(scene);
		if (model) pth_model_destroy(model);
		return 0;
	}
	pt_ctx *pt = nullptr;
	if (pt_create(&cfg, &pt) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error(nullptr)); return -1; }
	if (pt_upload_scene(pt, &desc) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error(pt)); return -1; }

	// Camera cam(origin 0, forward -z, right +x, 0.005, 0.05), src/main.cpp:67; uploaded when frame_count == 0 (:90-94)
	pt_set_camera(pt, cam_glsl);

	// the frame loop, src/main.cpp:69-101: one sample per pixel per frame, blended into the running mean
	for (uint32_t frame_count = 0; frame_count < frames; frame_count++) {
		const uint32_t frame_data[4] = {seed[frame_count], frame_count, BOUNCE_COUNT, 0};
		if (pt_render_frame(pt, frame_data) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error(pt)); return -1; }
	}

	// the blit, src/main.cpp:103-118 + shaders/framebuffer.frag
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
	if (model) pth_model_destroy(model);
	return 0;
}
