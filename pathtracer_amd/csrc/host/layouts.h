// Frozen buffer layouts of the drop-in boundary (byte-identical to the reference's *GLSL structs).
#pragma once
#include <cstdint>

struct PtTriangle {      // TriangleGLSL, src/scene/triangle.hpp:26-43
	float v0[3]; uint32_t uv0;   // xyz + packHalf2x16(uv) as bits
	float v1[3]; uint32_t uv1;
	float v2[3]; uint32_t uv2;
	uint32_t n0, n1, n2, mat;    // packHalf2x16(oct(n_i)), material index
};
struct PtNode {          // BVHNodeGLSL, src/scene/bvh.h:7-14
	float bmin[3]; float first;  // first child, or first triangle of a leaf (stored as float)
	float bmax[3]; float count;  // triangle count, 0 for inner nodes (stored as float)
};
struct PtMaterial {      // MaterialGLSL, src/scene/material.hpp:13-31
	float diffuse[3]; float roughness;
	float specular[3]; float type;
	float le[3]; float tex_index;
};
struct PtSphere {        // SphereGLSL, src/scene/sphere.hpp:7-15
	float origin[3]; float radius;
	uint32_t mat, pad[3];
};
struct PtCamera {        // CameraGLSL, src/scene/camera.hpp:26-39
	float origin[3]; float speed;
	float forward[3]; float sens;
	float right[3]; float zero;
};
struct PtFrameData {     // FrameData, src/display/display.hpp:7-13
	uint32_t seed, frame, bounces, pad;
};
static_assert(sizeof(PtTriangle) == 64, "triangle layout");
static_assert(sizeof(PtNode) == 32, "node layout");
static_assert(sizeof(PtMaterial) == 48, "material layout");
static_assert(sizeof(PtSphere) == 32, "sphere layout");
static_assert(sizeof(PtCamera) == 48, "camera layout");
static_assert(sizeof(PtFrameData) == 16, "frame data layout");
