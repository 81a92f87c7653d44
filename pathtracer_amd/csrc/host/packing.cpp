// Host-side packing of scene records into the boundary layouts (include/pt355_host.h).
// Output bytes are pinned against the reference's own TUs by tests/test_host_vs_reference.py.
#include <cmath>
#include <cstring>

#include "../../../include/pt355_host.h"
#include "layouts.h"

namespace {

inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// fp32 -> fp16 the way the reference's glm does it (type_half.inl:105-238): the 13 dropped
// mantissa bits round HALF UP on the magnitude (bit 12 decides), subnormals are built by
// shifting the significand, magnitudes below 2^-25 flush to signed zero.
uint16_t half_round_half_up(float f)
{
	const uint32_t bits = f32_bits(f);
	const uint32_t sign = (bits >> 16) & 0x8000u;
	int exp = (int) ((bits >> 23) & 0xffu) - 112;
	uint32_t man = bits & 0x007fffffu;

	if (exp <= 0) {
		if (exp < -10) return (uint16_t) sign;
		man = (man | 0x00800000u) >> (1 - exp);
		if (man & 0x1000u) man += 0x2000u;
		return (uint16_t) (sign | (man >> 13));
	}
	if (exp == 0xff - 112) {
		if (man == 0) return (uint16_t) (sign | 0x7c00u);
		man >>= 13;
		return (uint16_t) (sign | 0x7c00u | man | (man == 0 ? 1u : 0u));
	}
	if (man & 0x1000u) {
		man += 0x2000u;
		if (man & 0x00800000u) { man = 0; exp += 1; }
	}
	if (exp > 30) return (uint16_t) (sign | 0x7c00u);
	return (uint16_t) (sign | ((uint32_t) exp << 10) | (man >> 13));
}

} // namespace

extern "C" {

uint32_t pth_pack_half2x16(float x, float y)
{
	return (uint32_t) half_round_half_up(x) | ((uint32_t) half_round_half_up(y) << 16);
}

void pth_oct_encode(const float n[3], float out[2])
{
	const float l1 = fabsf(n[0]) + fabsf(n[1]) + fabsf(n[2]);
	const float x = n[0] / l1, y = n[1] / l1, z = n[2] / l1;
	if (z >= 0.0f) {
		out[0] = x;
		out[1] = y;
	} else { // fold the lower hemisphere over the diagonals
		out[0] = (1.0f - fabsf(y)) * (x >= 0.0f ? 1.0f : -1.0f);
		out[1] = (1.0f - fabsf(x)) * (y >= 0.0f ? 1.0f : -1.0f);
	}
}

void pth_pack_triangle(const float v[9], const float uv[6], const float n[9], uint32_t mat_index, void *out64)
{
	PtTriangle t;
	memcpy(t.v0, v, 12);
	memcpy(t.v1, v + 3, 12);
	memcpy(t.v2, v + 6, 12);
	t.uv0 = pth_pack_half2x16(uv[0], uv[1]);
	t.uv1 = pth_pack_half2x16(uv[2], uv[3]);
	t.uv2 = pth_pack_half2x16(uv[4], uv[5]);
	float e[2];
	pth_oct_encode(n, e);     t.n0 = pth_pack_half2x16(e[0], e[1]);
	pth_oct_encode(n + 3, e); t.n1 = pth_pack_half2x16(e[0], e[1]);
	pth_oct_encode(n + 6, e); t.n2 = pth_pack_half2x16(e[0], e[1]);
	t.mat = mat_index;
	memcpy(out64, &t, sizeof t);
}

void pth_pack_material(const float diffuse[3], const float specular[3], const float le[3], float roughness,
                       int tex_index, int type, void *out48)
{
	PtMaterial m;
	memcpy(m.diffuse, diffuse, 12);
	memcpy(m.specular, specular, 12);
	memcpy(m.le, le, 12);
	m.roughness = roughness;
	m.type = (float) type;
	m.tex_index = (float) tex_index;
	if (type == PTH_MAT_SPECULAR_METAL && 0.01f > m.roughness) m.roughness = 0.01f; // pixl::max(0.01f, r) = a > b ? a : b
	memcpy(out48, &m, sizeof m);
}

void pth_pack_sphere(const float origin[3], float radius, uint32_t mat_index, void *out32)
{
	PtSphere s;
	memcpy(s.origin, origin, 12);
	s.radius = radius;
	s.mat = mat_index;
	s.pad[0] = s.pad[1] = s.pad[2] = 0;
	memcpy(out32, &s, sizeof s);
}

void pth_pack_camera(const float origin[3], const float forward[3], const float right[3], float speed, float sens,
                     float out12[12])
{
	PtCamera c;
	memcpy(c.origin, origin, 12);
	memcpy(c.forward, forward, 12);
	memcpy(c.right, right, 12);
	c.speed = speed;
	c.sens = sens;
	c.zero = 0.0f;
	memcpy(out12, &c, sizeof c);
}

} // extern "C"
