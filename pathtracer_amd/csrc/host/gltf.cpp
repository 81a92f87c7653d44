// glTF / GLB loader and Model -- the host pieces of src/loader.cpp:13-401 and src/scene/model.cpp:15-68, rebuilt.
// Own parser (JSON + GLB container + accessors) instead of the reference's vendored cgltf; own PNG decoder over zlib
// instead of stb_image; GL texture-array calls (loader.cpp:48-58, 242-265) replaced by a host RGB8 layer array + sampler
// enums that go to pt_scene_desc.  What the reference's loader DOES with the data is reproduced statement by statement,
// its quirks included, because they change pixels:
//   * per-MESH attribute / index arrays that are appended per PRIMITIVE, and a de-index loop that re-walks the whole
//     index array for every primitive (loader.cpp:62-65, 304-345): primitive k re-emits the triangles of primitives
//     0..k-1 with its own material;
//   * the index accessor's byteOffset is ignored (loader.cpp:156-160) while attribute accessors honour theirs (:96-97);
//   * every TEXCOORD_n set lands in the one tex_coords array; attribute component types are assumed float;
//   * one MaterialGLSL per primitive; emissive only with KHR_materials_emissive_strength (:177-185); metallic materials
//     come out Lambertian with roughness 2.0 (:289-295); rough dielectrics Oren-Nayar with roughness * 0.35 (:276-283);
//     default diffuse magenta (:169);
//   * every non-camera node's transform is baked into ALL triangles, scale and translation with y and z swapped
//     (:349-393); each bake re-normalises every normal (model.cpp:27-42), identity or not.
// The glm arithmetic involved (mat4 * vec4, inverse, transpose, translate / rotate / scale, mat4_cast, normalize) is
// restated in glm 0.9.9.8's own operation order (thirdparty/glm/glm/detail/type_mat4x4.inl:536-581, 630-648,
// func_matrix.inl:294-351, ext/matrix_transform.inl:10-86, gtc/quaternion.inl:41-72, func_geometric.inl:82-90), so
// the output bytes equal the reference's: tests/test_gltf_loader.py against tests/golden/ref_gltf.npz.
// Where the reference would crash (missing indices / normals, out-of-range index) this loader returns an error.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/pt355_host.h"
#include "image_decode.h"
#include "layouts.h"

namespace {

thread_local std::string g_err;
int fail(const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = buf;
	return PT_EINVAL;
}

// ------------------------------------------------------------------------------------------------ JSON
struct JVal {
	enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
	double num = 0;
	bool b = false;
	std::string str;
	std::vector<JVal> arr;
	std::vector<std::pair<std::string, JVal>> obj; // key order preserved: attribute order decides append order
	const JVal *get(const char *key) const
	{
		if (type != Obj) return nullptr;
		for (const auto &kv : obj)
			if (kv.first == key) return &kv.second;
		return nullptr;
	}
	double number(const char *key, double dflt) const
	{
		const JVal *v = get(key);
		return v && v->type == Num ? v->num : dflt;
	}
	// A JSON number is a double, and a hostile file puts anything there: every conversion to an integer type checks the
	// value BEFORE the cast (negative, huge or non-finite doubles cast straight to an integer are undefined behaviour).
	static long to_index(double v) { return v >= 0.0 && v < 2147483648.0 ? (long) v : -1; }
	long index(const char *key) const
	{
		const JVal *v = get(key);
		return v && v->type == Num ? to_index(v->num) : -1;
	}
	// non-negative size / offset / count; false when the file holds something that is not one (then *out is untouched)
	bool size(const char *key, size_t dflt, size_t *out) const
	{
		const JVal *v = get(key);
		if (!v || v->type != Num) { *out = dflt; return true; }
		if (!(v->num >= 0.0) || !(v->num <= 9007199254740992.0)) return false;
		*out = (size_t) v->num;
		return true;
	}
	// small integers (enums, modes): anything outside int's range reads as -1, which no caller accepts
	int small(const char *key, int dflt) const
	{
		const JVal *v = get(key);
		if (!v || v->type != Num) return dflt;
		return v->num >= -2147483648.0 && v->num <= 2147483647.0 ? (int) v->num : -1;
	}
};

struct JParser {
	const char *p, *end;
	bool ok = true;
	void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
	bool lit(const char *s)
	{
		size_t n = strlen(s);
		if ((size_t) (end - p) >= n && memcmp(p, s, n) == 0) { p += n; return true; }
		return false;
	}
	std::string string()
	{
		std::string out;
		p++; // opening quote
		while (p < end && *p != '"') {
			if (*p == '\\' && p + 1 < end) {
				p++;
				switch (*p) {
				case 'n': out += '\n'; break;
				case 't': out += '\t'; break;
				case 'r': out += '\r'; break;
				case 'b': out += '\b'; break;
				case 'f': out += '\f'; break;
				case 'u': { // \uXXXX: kept as UTF-8 (only names and URIs carry strings here)
					unsigned cp = 0;
					for (int i = 0; i < 4 && p + 1 < end; i++) { p++; cp = cp * 16 + (unsigned) (isdigit((unsigned char) *p) ? *p - '0' : (tolower(*p) - 'a' + 10)); }
					if (cp < 0x80) out += (char) cp;
					else if (cp < 0x800) { out += (char) (0xc0 | (cp >> 6)); out += (char) (0x80 | (cp & 0x3f)); }
					else { out += (char) (0xe0 | (cp >> 12)); out += (char) (0x80 | ((cp >> 6) & 0x3f)); out += (char) (0x80 | (cp & 0x3f)); }
					break;
				}
				default: out += *p;
				}
				p++;
			} else out += *p++;
		}
		if (p >= end) ok = false;
		else p++;
		return out;
	}
	JVal value(int depth = 0)
	{
		JVal v;
		ws();
		if (p >= end || depth > 64) { ok = false; return v; }
		if (*p == '{') {
			v.type = JVal::Obj;
			p++;
			ws();
			if (p < end && *p == '}') { p++; return v; }
			while (ok) {
				ws();
				if (p >= end || *p != '"') { ok = false; break; }
				std::string k = string();
				ws();
				if (p >= end || *p != ':') { ok = false; break; }
				p++;
				v.obj.emplace_back(std::move(k), value(depth + 1));
				ws();
				if (p < end && *p == ',') { p++; continue; }
				if (p < end && *p == '}') { p++; break; }
				ok = false;
			}
		} else if (*p == '[') {
			v.type = JVal::Arr;
			p++;
			ws();
			if (p < end && *p == ']') { p++; return v; }
			while (ok) {
				v.arr.push_back(value(depth + 1));
				ws();
				if (p < end && *p == ',') { p++; continue; }
				if (p < end && *p == ']') { p++; break; }
				ok = false;
			}
		} else if (*p == '"') {
			v.type = JVal::Str;
			v.str = string();
		} else if (lit("true")) { v.type = JVal::Bool; v.b = true; }
		else if (lit("false")) { v.type = JVal::Bool; v.b = false; }
		else if (lit("null")) { v.type = JVal::Null; }
		else {
			// the token goes through atof like cgltf_json_to_float does; callers narrow the double to float
			char tmp[128];
			size_t n = 0;
			while (p < end && n + 1 < sizeof tmp && (isdigit((unsigned char) *p) || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) tmp[n++] = *p++;
			tmp[n] = 0;
			if (n == 0) { ok = false; return v; }
			v.type = JVal::Num;
			v.num = atof(tmp);
		}
		return v;
	}
};

// ------------------------------------------------------------------------------------------------ glTF document
struct Buffer { std::vector<uint8_t> data; };
struct View { long buffer = -1; size_t offset = 0, size = 0, stride = 0; };
enum AType { A_SCALAR, A_VEC2, A_VEC3, A_VEC4, A_OTHER };
struct Accessor { long view = -1; size_t offset = 0, count = 0, stride = 0; int comp = 0; AType type = A_OTHER; };

// do `count` (>= 1) elements of `need` bytes, `stride` apart from base + offset, all lie inside `len` bytes?  Overflow-free.
bool span_inside(size_t base, size_t offset, size_t stride, size_t count, size_t need, size_t len)
{
	if (base > len || offset > len - base || need > len - base - offset) return false;
	if (count == 1) return true;
	if (stride == 0) return false; // more than one element in the same place: never a valid accessor
	return (count - 1) <= (len - base - offset - need) / stride;
}
size_t comp_size(int c) { return c == 5120 || c == 5121 ? 1 : (c == 5122 || c == 5123 ? 2 : (c == 5125 || c == 5126 ? 4 : 0)); }
size_t type_count(AType t) { return t == A_SCALAR ? 1 : (t == A_VEC2 ? 2 : (t == A_VEC3 ? 3 : (t == A_VEC4 ? 4 : 0))); }

bool read_file(const std::string &path, std::vector<uint8_t> &out)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	out.resize(n > 0 ? (size_t) n : 0);
	bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
	fclose(f);
	return ok;
}

bool base64(const char *s, size_t n, std::vector<uint8_t> &out)
{
	auto val = [](char c) { return c >= 'A' && c <= 'Z' ? c - 'A' : (c >= 'a' && c <= 'z' ? c - 'a' + 26 : (c >= '0' && c <= '9' ? c - '0' + 52 : (c == '+' ? 62 : (c == '/' ? 63 : -1)))); };
	uint32_t acc = 0;
	int bits = 0;
	for (size_t i = 0; i < n; i++) {
		if (s[i] == '=') break;
		int v = val(s[i]);
		if (v < 0) return false;
		acc = (acc << 6) | (uint32_t) v;
		bits += 6;
		if (bits >= 8) { bits -= 8; out.push_back((uint8_t) (acc >> bits)); }
	}
	return true;
}

struct Doc {
	JVal root;
	std::vector<Buffer> buffers;
	std::vector<View> views;
	std::vector<Accessor> accessors;
};

int parse_document(const char *path, Doc &d)
{
	std::vector<uint8_t> file;
	if (!read_file(path, file)) return fail("LoadGLTF: cannot read %s", path);
	const uint8_t *json = file.data();
	size_t json_len = file.size();
	const uint8_t *bin = nullptr;
	size_t bin_len = 0;
	auto u32 = [&](size_t o) { uint32_t v; memcpy(&v, file.data() + o, 4); return v; };
	if (file.size() >= 12 && u32(0) == 0x46546C67u) { // "glTF": binary container
		if (u32(4) != 2u) return fail("LoadGLTF: GLB version %u is not 2", u32(4));
		size_t o = 12;
		bool have_json = false;
		while (o + 8 <= file.size()) {
			const size_t len = u32(o);
			const uint32_t kind = u32(o + 4);
			if (o + 8 + len > file.size()) return fail("LoadGLTF: GLB chunk runs past the end of the file");
			if (kind == 0x4E4F534Au && !have_json) { json = file.data() + o + 8; json_len = len; have_json = true; }
			else if (kind == 0x004E4942u && !bin) { bin = file.data() + o + 8; bin_len = len; }
			o += 8 + ((len + 3) & ~(size_t) 3);
		}
		if (!have_json) return fail("LoadGLTF: GLB without a JSON chunk");
	}
	JParser jp{(const char *) json, (const char *) json + json_len};
	d.root = jp.value();
	if (!jp.ok || d.root.type != JVal::Obj) return fail("LoadGLTF: malformed JSON");
	std::string dir = path;
	size_t slash = dir.find_last_of("/\\");
	dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
	if (const JVal *bs = d.root.get("buffers"))
		for (size_t i = 0; i < bs->arr.size(); i++) {
			Buffer b;
			const JVal &jb = bs->arr[i];
			size_t want = 0;
			if (!jb.size("byteLength", 0, &want)) return fail("LoadGLTF: buffer %zu: byteLength is not a size", i);
			const JVal *uri = jb.get("uri");
			if (!uri) {
				if (i != 0 || !bin) return fail("LoadGLTF: buffer %zu has no uri and there is no GLB BIN chunk for it", i);
				b.data.assign(bin, bin + bin_len);
			} else if (uri->str.compare(0, 5, "data:") == 0) {
				size_t comma = uri->str.find(',');
				if (comma == std::string::npos || uri->str.find(";base64") == std::string::npos || !base64(uri->str.c_str() + comma + 1, uri->str.size() - comma - 1, b.data))
					return fail("LoadGLTF: buffer %zu: unsupported data URI", i);
			} else if (!read_file(dir + uri->str, b.data)) return fail("LoadGLTF: cannot read buffer file %s", (dir + uri->str).c_str());
			if (b.data.size() < want) return fail("LoadGLTF: buffer %zu holds %zu bytes, byteLength says %zu", i, b.data.size(), want);
			d.buffers.push_back(std::move(b));
		}
	if (const JVal *vs = d.root.get("bufferViews"))
		for (const JVal &jv : vs->arr) {
			View v;
			v.buffer = jv.index("buffer");
			if (!jv.size("byteOffset", 0, &v.offset) || !jv.size("byteLength", 0, &v.size) || !jv.size("byteStride", 0, &v.stride))
				return fail("LoadGLTF: buffer view %zu: offset, length or stride is not a size", d.views.size());
			// size <= len && offset <= len - size: the sum of two hostile sizes may wrap
			if (v.buffer < 0 || (size_t) v.buffer >= d.buffers.size() || v.size > d.buffers[v.buffer].data.size() ||
			    v.offset > d.buffers[v.buffer].data.size() - v.size)
				return fail("LoadGLTF: buffer view %zu lies outside its buffer", d.views.size());
			d.views.push_back(v);
		}
	if (const JVal *as = d.root.get("accessors"))
		for (const JVal &ja : as->arr) {
			Accessor a;
			a.view = ja.index("bufferView");
			if (!ja.size("byteOffset", 0, &a.offset) || !ja.size("count", 0, &a.count))
				return fail("LoadGLTF: accessor %zu: offset or count is not a size", d.accessors.size());
			a.comp = ja.small("componentType", 0);
			const JVal *t = ja.get("type");
			const std::string ts = t ? t->str : "";
			a.type = ts == "SCALAR" ? A_SCALAR : (ts == "VEC2" ? A_VEC2 : (ts == "VEC3" ? A_VEC3 : (ts == "VEC4" ? A_VEC4 : A_OTHER)));
			a.stride = comp_size(a.comp) * type_count(a.type); // cgltf: tightly packed element size
			if (a.view >= (long) d.views.size()) return fail("LoadGLTF: accessor %zu names buffer view %ld of %zu", d.accessors.size(), a.view, d.views.size());
			d.accessors.push_back(a);
		}
	return PT_OK;
}

// ------------------------------------------------------------------------------------------------ glm, restated
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct M4 { V4 c[4]; }; // column-major like glm::mat4
inline V4 mul(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V4 add(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 sub(V4 a, V4 b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 mulv(V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
M4 identity() { return M4{{V4{1, 0, 0, 0}, V4{0, 1, 0, 0}, V4{0, 0, 1, 0}, V4{0, 0, 0, 1}}}; }
inline float el(const M4 &m, int c, int r) { return (&m.c[c].x)[r]; }

// type_mat4x4.inl:561-572: (m0*x + m1*y) + (m2*z + m3*w)
V4 mat_vec(const M4 &m, V4 v) { return add(add(mul(m.c[0], v.x), mul(m.c[1], v.y)), add(mul(m.c[2], v.z), mul(m.c[3], v.w))); }
// type_mat4x4.inl:630-648: left to right
M4 mat_mat(const M4 &a, const M4 &b)
{
	M4 r;
	for (int i = 0; i < 4; i++)
		r.c[i] = add(add(add(mul(a.c[0], b.c[i].x), mul(a.c[1], b.c[i].y)), mul(a.c[2], b.c[i].z)), mul(a.c[3], b.c[i].w));
	return r;
}
// ext/matrix_transform.inl:10-15
M4 translate(const M4 &m, V3 v)
{
	M4 r = m;
	r.c[3] = add(add(add(mul(m.c[0], v.x), mul(m.c[1], v.y)), mul(m.c[2], v.z)), m.c[3]);
	return r;
}
// ext/matrix_transform.inl:78-86
M4 scale(const M4 &m, V3 v) { return M4{{mul(m.c[0], v.x), mul(m.c[1], v.y), mul(m.c[2], v.z), m.c[3]}}; }
// func_geometric.inl:82-90 + func_exponential.inl:134-139: v * (1 / sqrt(dot)), dot = x*x + y*y + z*z left to right
V3 normalize(V3 v)
{
	const float d = v.x * v.x + v.y * v.y + v.z * v.z;
	const float inv = 1.0f / sqrtf(d);
	return V3{v.x * inv, v.y * inv, v.z * inv};
}
// ext/matrix_transform.inl:18-46
M4 rotate(const M4 &m, float angle, V3 v)
{
	const float c = cosf(angle), s = sinf(angle);
	const V3 axis = normalize(v);
	const V3 temp = V3{(1.0f - c) * axis.x, (1.0f - c) * axis.y, (1.0f - c) * axis.z};
	float R[3][3];
	R[0][0] = c + temp.x * axis.x;
	R[0][1] = temp.x * axis.y + s * axis.z;
	R[0][2] = temp.x * axis.z - s * axis.y;
	R[1][0] = temp.y * axis.x - s * axis.z;
	R[1][1] = c + temp.y * axis.y;
	R[1][2] = temp.y * axis.z + s * axis.x;
	R[2][0] = temp.z * axis.x + s * axis.y;
	R[2][1] = temp.z * axis.y - s * axis.x;
	R[2][2] = c + temp.z * axis.z;
	M4 r;
	for (int i = 0; i < 3; i++) r.c[i] = add(add(mul(m.c[0], R[i][0]), mul(m.c[1], R[i][1])), mul(m.c[2], R[i][2]));
	r.c[3] = m.c[3];
	return r;
}
// gtc/quaternion.inl:41-72 (q given as x, y, z, w)
M4 mat4_cast(float qx, float qy, float qz, float qw)
{
	const float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz, qxz = qx * qz, qxy = qx * qy, qyz = qy * qz, qwx = qw * qx, qwy = qw * qy, qwz = qw * qz;
	M4 r = identity();
	r.c[0] = V4{1.0f - 2.0f * (qyy + qzz), 2.0f * (qxy + qwz), 2.0f * (qxz - qwy), 0.0f};
	r.c[1] = V4{2.0f * (qxy - qwz), 1.0f - 2.0f * (qxx + qzz), 2.0f * (qyz + qwx), 0.0f};
	r.c[2] = V4{2.0f * (qxz + qwy), 2.0f * (qyz - qwx), 1.0f - 2.0f * (qxx + qyy), 0.0f};
	return r;
}
// func_matrix.inl:294-351
M4 inverse(const M4 &m)
{
#define E(c_, r_) el(m, c_, r_)
	const float Coef00 = E(2, 2) * E(3, 3) - E(3, 2) * E(2, 3);
	const float Coef02 = E(1, 2) * E(3, 3) - E(3, 2) * E(1, 3);
	const float Coef03 = E(1, 2) * E(2, 3) - E(2, 2) * E(1, 3);
	const float Coef04 = E(2, 1) * E(3, 3) - E(3, 1) * E(2, 3);
	const float Coef06 = E(1, 1) * E(3, 3) - E(3, 1) * E(1, 3);
	const float Coef07 = E(1, 1) * E(2, 3) - E(2, 1) * E(1, 3);
	const float Coef08 = E(2, 1) * E(3, 2) - E(3, 1) * E(2, 2);
	const float Coef10 = E(1, 1) * E(3, 2) - E(3, 1) * E(1, 2);
	const float Coef11 = E(1, 1) * E(2, 2) - E(2, 1) * E(1, 2);
	const float Coef12 = E(2, 0) * E(3, 3) - E(3, 0) * E(2, 3);
	const float Coef14 = E(1, 0) * E(3, 3) - E(3, 0) * E(1, 3);
	const float Coef15 = E(1, 0) * E(2, 3) - E(2, 0) * E(1, 3);
	const float Coef16 = E(2, 0) * E(3, 2) - E(3, 0) * E(2, 2);
	const float Coef18 = E(1, 0) * E(3, 2) - E(3, 0) * E(1, 2);
	const float Coef19 = E(1, 0) * E(2, 2) - E(2, 0) * E(1, 2);
	const float Coef20 = E(2, 0) * E(3, 1) - E(3, 0) * E(2, 1);
	const float Coef22 = E(1, 0) * E(3, 1) - E(3, 0) * E(1, 1);
	const float Coef23 = E(1, 0) * E(2, 1) - E(2, 0) * E(1, 1);
	const V4 Fac0{Coef00, Coef00, Coef02, Coef03}, Fac1{Coef04, Coef04, Coef06, Coef07}, Fac2{Coef08, Coef08, Coef10, Coef11};
	const V4 Fac3{Coef12, Coef12, Coef14, Coef15}, Fac4{Coef16, Coef16, Coef18, Coef19}, Fac5{Coef20, Coef20, Coef22, Coef23};
	const V4 Vec0{E(1, 0), E(0, 0), E(0, 0), E(0, 0)}, Vec1{E(1, 1), E(0, 1), E(0, 1), E(0, 1)};
	const V4 Vec2{E(1, 2), E(0, 2), E(0, 2), E(0, 2)}, Vec3{E(1, 3), E(0, 3), E(0, 3), E(0, 3)};
#undef E
	const V4 Inv0 = add(sub(mulv(Vec1, Fac0), mulv(Vec2, Fac1)), mulv(Vec3, Fac2));
	const V4 Inv1 = add(sub(mulv(Vec0, Fac0), mulv(Vec2, Fac3)), mulv(Vec3, Fac4));
	const V4 Inv2 = add(sub(mulv(Vec0, Fac1), mulv(Vec1, Fac3)), mulv(Vec3, Fac5));
	const V4 Inv3 = add(sub(mulv(Vec0, Fac2), mulv(Vec1, Fac4)), mulv(Vec2, Fac5));
	const V4 SignA{+1, -1, +1, -1}, SignB{-1, +1, -1, +1};
	M4 inv{{mulv(Inv0, SignA), mulv(Inv1, SignB), mulv(Inv2, SignA), mulv(Inv3, SignB)}};
	const V4 Row0{inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x};
	const V4 Dot0 = mulv(m.c[0], Row0);
	const float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
	const float OneOverDeterminant = 1.0f / Dot1;
	for (int i = 0; i < 4; i++) inv.c[i] = mul(inv.c[i], OneOverDeterminant);
	return inv;
}
// type_mat3x3.inl:468-474 on mat3(transpose(inverse(M)))
V3 normal_mul(const M4 &it /* inverse, NOT yet transposed */, V3 v)
{
	// N = transpose(it): N[c][r] = it[r][c]; result_r = N[0][r]*v.x + N[1][r]*v.y + N[2][r]*v.z = it[r][0]*v.x + it[r][1]*v.y + it[r][2]*v.z
	return V3{el(it, 0, 0) * v.x + el(it, 0, 1) * v.y + el(it, 0, 2) * v.z,
	          el(it, 1, 0) * v.x + el(it, 1, 1) * v.y + el(it, 1, 2) * v.z,
	          el(it, 2, 0) * v.x + el(it, 2, 1) * v.y + el(it, 2, 2) * v.z};
}

// ------------------------------------------------------------------------------------------------ PNG (zlib)
uint32_t be32(const uint8_t *p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }

// 8/16-bit grey, RGB, palette, grey+alpha, RGBA, non-interlaced -> tightly packed RGB8 (alpha dropped, 16 -> 8 bit by
// the high byte: what stbi_load_from_memory(..., req_comp = 3) returns, loader.cpp:209-214)
int decode_png_rgb8(const uint8_t *d, size_t n, std::vector<uint8_t> &rgb, int *w, int *h, int *channels)
{
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
	if (n < 8 || memcmp(d, sig, 8) != 0) return fail("LoadGLTF: texture is neither a PNG nor a JPEG (other formats are not supported by this build)");
	size_t o = 8;
	uint32_t W = 0, H = 0;
	int depth = 0, ctype = 0, interlace = 0;
	std::vector<uint8_t> idat, pal;
	while (o + 12 <= n) {
		const uint32_t len = be32(d + o);
		const uint8_t *ty = d + o + 4, *body = d + o + 8;
		if (o + 12 + (size_t) len > n) return fail("LoadGLTF: truncated PNG");
		if (!memcmp(ty, "IHDR", 4) && len >= 13) { W = be32(body); H = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
		else if (!memcmp(ty, "PLTE", 4)) pal.assign(body, body + len);
		else if (!memcmp(ty, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
		else if (!memcmp(ty, "IEND", 4)) break;
		o += 12 + (size_t) len;
	}
	if (W == 0 || H == 0 || W > 16384 || H > 16384) return fail("LoadGLTF: bad PNG size %ux%u", W, H);
	if (interlace) return fail("LoadGLTF: interlaced PNG textures are not supported");
	const int nch = ctype == 0 ? 1 : (ctype == 2 ? 3 : (ctype == 3 ? 1 : (ctype == 4 ? 2 : (ctype == 6 ? 4 : 0))));
	if (!nch || (depth != 8 && depth != 16) || (ctype == 3 && depth != 8)) return fail("LoadGLTF: PNG colour type %d / bit depth %d is not supported", ctype, depth);
	const size_t bpp = (size_t) nch * (depth / 8), row = (size_t) W * bpp;
	std::vector<uint8_t> raw((row + 1) * H);
	uLongf out_len = (uLongf) raw.size();
	if (uncompress(raw.data(), &out_len, idat.data(), (uLong) idat.size()) != Z_OK || out_len != raw.size()) return fail("LoadGLTF: PNG data does not inflate to the image size");
	std::vector<uint8_t> img(row * H);
	for (uint32_t y = 0; y < H; y++) {
		const uint8_t *src = raw.data() + (row + 1) * y;
		uint8_t *cur = img.data() + row * y;
		const uint8_t *up = y ? cur - row : nullptr;
		const int f = src[0];
		for (size_t x = 0; x < row; x++) {
			const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
			int pred = 0;
			if (f == 1) pred = a;
			else if (f == 2) pred = b;
			else if (f == 3) pred = (a + b) >> 1;
			else if (f == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
			else if (f != 0) return fail("LoadGLTF: bad PNG filter %d", f);
			cur[x] = (uint8_t) (src[1 + x] + pred);
		}
	}
	rgb.resize((size_t) W * H * 3);
	const size_t step = depth / 8;
	for (size_t i = 0; i < (size_t) W * H; i++) {
		const uint8_t *p = img.data() + i * bpp;
		uint8_t r, g, b;
		if (ctype == 3) { const size_t k = (size_t) p[0] * 3; if (k + 3 > pal.size()) { r = g = b = 0; } else { r = pal[k]; g = pal[k + 1]; b = pal[k + 2]; } }
		else if (nch <= 2) r = g = b = p[0];
		else { r = p[0]; g = p[step]; b = p[2 * step]; }
		rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
	}
	*w = (int) W; *h = (int) H;
	*channels = ctype == 3 ? 3 : nch; // stbi reports the file's channel count (palette images count as RGB)
	return PT_OK;
}

// A texture's bytes -> RGB8: PNG or JPEG by signature (stbi_load_from_memory tries JPEG first, then PNG: stb_image.h:1127-1140)
int decode_image_rgb8(const uint8_t *d, size_t n, std::vector<uint8_t> &rgb, int *w, int *h, int *channels)
{
	if (n >= 3 && d[0] == 0xff && d[1] == 0xd8) {
		std::string err;
		if (pth_internal_decode_jpeg_rgb8(d, n, rgb, w, h, channels, err)) return fail("LoadGLTF: JPEG texture: %s", err.c_str());
		return PT_OK;
	}
	return decode_png_rgb8(d, n, rgb, w, h, channels);
}

// stbir_resize_uint8(in, w, h, 0, out, 512, 512, 0, 3) as src/loader.cpp:224-232 calls it (stb_image_resize v0.97), restated so
// that every float operation happens in the same order: clamped edges, linear colour space, Catmull-Rom when an axis is
// enlarged and Mitchell-Netravali when it is not (an axis of exactly 512 is "not enlarged": it is filtered, not copied),
// coefficients normalised per output pixel, gather form for enlarged axes and scatter form (input pixel -> the output
// pixels it touches) for the others, horizontal pass first, vertical accumulation in scanline order, bytes by
// (int)(saturate(v) * 255 + 0.5).  The reference's ring buffers only cache rows; dense accumulators give the same sums.
// Even its one out-of-bounds habit is kept: a scatter coefficient group is 4 wide but 5 values can be written, the 5th
// landing in the next group's first slot until that group is computed (stb_image_resize.h:1092-1124, 1126-1199).
// tests/test_gltf_loader.py::test_resize_to_512 compares with the reference's own stb build byte for byte.
struct ResizeAxis {
	bool up = false;
	float scale = 1.0f;
	int in_size = 0, out_size = 0, margin = 0, coef_w = 4, n_contrib = 0;
	std::vector<int> n0, n1;
	std::vector<float> coef; // [n_contrib][coef_w], flat, zero-padded at the end
	float kernel(float x) const
	{
		x = (float) fabs(x);
		if (up) { // Catmull-Rom (stb_image_resize.h:811-823)
			if (x < 1.0f) return 1 - x * x * (2.5f - 1.5f * x);
			if (x < 2.0f) return 2 - x * (4 + x * (0.5f * x - 2.5f));
			return 0.0f;
		}
		if (x < 1.0f) return (16 + x * x * (21 * x - 36)) / 18; // Mitchell (:825-837)
		if (x < 2.0f) return (32 + x * (-60 + x * (36 - 7 * x))) / 18;
		return 0.0f;
	}
	void build(int in, int out)
	{
		in_size = in; out_size = out;
		scale = ((float) out / in) / (1.0f - 0.0f);
		up = scale > 1;
		const float support = 2.0f;
		const int pixel_width = up ? (int) ceil(support * 2) : (int) ceil(support * 2 / scale);
		margin = pixel_width / 2;
		coef_w = (int) ceil(support * 2);
		n_contrib = up ? out : in + margin * 2;
		n0.assign((size_t) n_contrib, 0); n1.assign((size_t) n_contrib, 0);
		coef.assign((size_t) n_contrib * coef_w + 8, 0.0f);
		const float shift = 0.0f;
		if (up) {
			const float out_radius = support * scale;
			for (int n = 0; n < n_contrib; n++) {
				const float out_center = (float) n + 0.5f;
				const float lo = (out_center - out_radius + shift) / scale, hi = (out_center + out_radius + shift) / scale;
				const float in_center_of_out = (out_center + shift) / scale;
				int first = (int) (floor(lo + 0.5)), last = (int) (floor(hi - 0.5));
				float *g = &coef[(size_t) n * coef_w];
				float total = 0;
				n0[n] = first; n1[n] = last;
				for (int i = 0; i <= last - first; i++) {
					const float in_pixel_center = (float) (i + first) + 0.5f;
					g[i] = kernel(in_center_of_out - in_pixel_center);
					if (i == 0 && !g[i]) { n0[n] = ++first; i--; continue; } // a leading zero is dropped, slot 0 rewritten
					total += g[i];
				}
				const float filter_scale = 1 / total;
				for (int i = 0; i <= last - first; i++) g[i] *= filter_scale;
				for (int i = last - first; i >= 0; i--) {
					if (g[i]) break;
					n1[n] = n0[n] + i - 1;
				}
			}
			return;
		}
		const float in_radius = support / scale;
		for (int n = 0; n < n_contrib; n++) {
			const float in_center = (float) (n - margin) + 0.5f;
			const float lo = (in_center - in_radius) * scale - shift, hi = (in_center + in_radius) * scale - shift;
			const float out_center_of_in = in_center * scale - shift;
			const int first = (int) (floor(lo + 0.5)), last = (int) (floor(hi - 0.5));
			float *g = &coef[(size_t) n * coef_w];
			n0[n] = first; n1[n] = last;
			for (int i = 0; i <= last - first; i++) {
				const float out_pixel_center = (float) (i + first) + 0.5f;
				g[i] = kernel((out_pixel_center - out_center_of_in) * 1.0f) * scale; // kernel(x, scale): Mitchell ignores the scale argument
			}
			for (int i = last - first; i >= 0; i--) {
				if (g[i]) break;
				n1[n] = n0[n] + i - 1;
			}
		}
		// per output pixel: the coefficients of every input pixel reaching it sum to 1
		for (int i = 0; i < out_size; i++) {
			float total = 0;
			for (int j = 0; j < n_contrib; j++) {
				if (i >= n0[j] && i <= n1[j]) total += coef[(size_t) j * coef_w + (i - n0[j])];
				else if (i < n0[j]) break;
			}
			const float sc = 1 / total;
			for (int j = 0; j < n_contrib; j++) {
				if (i >= n0[j] && i <= n1[j]) coef[(size_t) j * coef_w + (i - n0[j])] *= sc;
				else if (i < n0[j]) break;
			}
		}
		// leading zeros and outputs left of the image are skipped, the group shifted down accordingly
		for (int j = 0; j < n_contrib; j++) {
			int skip = 0;
			while ((size_t) j * coef_w + skip < coef.size() && coef[(size_t) j * coef_w + skip] == 0) skip++;
			n0[j] += skip;
			while (n0[j] < 0) { n0[j]++; skip++; }
			const int range = n1[j] - n0[j] + 1, mx = coef_w < range ? coef_w : range;
			for (int i = 0; i < mx; i++) {
				if (i + skip >= coef_w) break;
				coef[(size_t) j * coef_w + i] = coef[(size_t) j * coef_w + i + skip];
			}
		}
		for (int j = 0; j < n_contrib; j++) n1[j] = n1[j] < out_size - 1 ? n1[j] : out_size - 1;
	}
};

void resize_rgb8_512(const std::vector<uint8_t> &in, int w, int h, uint8_t *out)
{
	if (w == 512 && h == 512) { memcpy(out, in.data(), (size_t) 512 * 512 * 3); return; } // loader.cpp:222: no resize call at all
	const int ch = 3, ow = 512, oh = 512;
	ResizeAxis H, V;
	H.build(w, ow);
	V.build(h, oh);
	std::vector<float> decode((size_t) (w + 2 * H.margin) * ch), hrow((size_t) ow * ch);
	float *const dec = decode.data() + (size_t) H.margin * ch; // index 0 = first image pixel, negative indices = the left margin
	// input scanline n (clamped to the image) decoded to floats and resampled horizontally into hrow
	auto horizontal = [&](int n) {
		const int row = n < 0 ? 0 : (n >= h ? h - 1 : n);
		const uint8_t *src = in.data() + (size_t) row * w * ch;
		for (int x = -H.margin; x < w + H.margin; x++) {
			const int xx = x < 0 ? 0 : (x >= w ? w - 1 : x);
			for (int c = 0; c < ch; c++) dec[x * ch + c] = ((float) src[xx * ch + c]) / 255.0f;
		}
		std::fill(hrow.begin(), hrow.end(), 0.0f);
		if (H.up) {
			for (int x = 0; x < ow; x++) {
				int counter = 0;
				for (int k = H.n0[x]; k <= H.n1[x]; k++) {
					const float cf = H.coef[(size_t) x * H.coef_w + counter++];
					for (int c = 0; c < ch; c++) hrow[(size_t) x * ch + c] += dec[k * ch + c] * cf;
				}
			}
		} else {
			for (int x = 0; x < w + 2 * H.margin; x++) {
				const int in_x = x - H.margin;
				for (int k = H.n0[x]; k <= H.n1[x]; k++) {
					const float cf = H.coef[(size_t) x * H.coef_w + k - H.n0[x]];
					for (int c = 0; c < ch; c++) hrow[(size_t) k * ch + c] += dec[in_x * ch + c] * cf;
				}
			}
		}
	};
	auto encode = [&](const float *row, int y) {
		for (int i = 0; i < ow * ch; i++) {
			float v = row[i];
			v = v < 0 ? 0 : (v > 1 ? 1 : v);
			out[(size_t) y * ow * ch + i] = (uint8_t) (int) ((v * 255.0f) + 0.5);
		}
	};
	if (V.up) {
		// every needed scanline is resampled once (the reference keeps them in a ring buffer), then each output row sums its
		// contributors in order
		std::vector<std::vector<float>> cache((size_t) h + 2 * V.margin + 8);
		const int base = V.margin + 4;
		std::vector<float> acc((size_t) ow * ch);
		for (int y = 0; y < oh; y++) {
			std::fill(acc.begin(), acc.end(), 0.0f);
			int counter = 0;
			for (int k = V.n0[y]; k <= V.n1[y]; k++) {
				std::vector<float> &r = cache[(size_t) (k + base)];
				if (r.empty()) { horizontal(k); r = hrow; }
				const float cf = V.coef[(size_t) y * V.coef_w + counter++];
				for (int i = 0; i < ow * ch; i++) acc[i] += r[i] * cf;
			}
			encode(acc.data(), y);
			for (int k = 0; k + 1 < V.n0[y] + base && k < (int) cache.size(); k++) if (!cache[(size_t) k].empty()) std::vector<float>().swap(cache[(size_t) k]);
		}
		return;
	}
	std::vector<float> acc((size_t) oh * ow * ch, 0.0f);
	const float in_radius = 2.0f / V.scale;
	for (int y = -V.margin; y < h + V.margin; y++) {
		const float in_center = (float) y + 0.5f;
		const float lo = (in_center - in_radius) * V.scale - 0.0f, hi = (in_center + in_radius) * V.scale - 0.0f;
		const int first = (int) (floor(lo + 0.5)), last = (int) (floor(hi - 0.5));
		if (last < 0 || first >= oh) continue;
		horizontal(y);
		const int j = y + V.margin;
		for (int k = V.n0[j]; k <= V.n1[j]; k++) {
			const float cf = V.coef[(size_t) j * V.coef_w + k - V.n0[j]];
			float *r = &acc[(size_t) k * ow * ch];
			for (int i = 0; i < ow * ch; i++) r[i] += hrow[i] * cf;
		}
	}
	for (int y = 0; y < oh; y++) encode(&acc[(size_t) y * ow * ch], y);
}

} // namespace

// ------------------------------------------------------------------------------------------------ Model
struct pth_model {
	// fat triangles, src/scene/triangle.hpp:8-24
	std::vector<V3> v, n;           // 3 per triangle
	std::vector<float> uv;          // 6 per triangle
	std::vector<uint32_t> mat;
	std::vector<PtMaterial> materials;
	M4 model_matrix = identity();
	// what loader.cpp:48-58, 242-265 puts into the GL texture array: `layers` 512 x 512 RGB8 images + the array-wide sampler state
	std::vector<uint8_t> tex;
	uint32_t tex_layers = 0, wrap_s = 0, wrap_t = 0, mag_filter = 0, min_filter = 0;
};

namespace {

// Model::ApplyModelMatrixToTris, src/scene/model.cpp:27-42
void apply_model_matrix(pth_model *m)
{
	const M4 inv = inverse(m->model_matrix);
	for (size_t i = 0; i < m->v.size(); i++) {
		const V4 p = mat_vec(m->model_matrix, V4{m->v[i].x, m->v[i].y, m->v[i].z, 1.0f});
		m->v[i] = V3{p.x, p.y, p.z};
		m->n[i] = normalize(normal_mul(inv, m->n[i]));
	}
	m->model_matrix = identity();
}

int load_material(const Doc &d, const JVal *jm, pth_model *out, int *num_loaded_textures, uint32_t num_textures)
{
	float diffuse[3] = {1.0f, 0.0f, 1.0f}, specular[3] = {0, 0, 0}, le[3] = {0, 0, 0};
	float roughness = 0.0f;
	int tex_index = -1, type = PTH_MAT_LAMBERTIAN;
	if (jm) {
		const JVal *ext = jm->get("extensions");
		const JVal *es = ext ? ext->get("KHR_materials_emissive_strength") : nullptr;
		const JVal *pbr = jm->get("pbrMetallicRoughness");
		if (es) { // loader.cpp:177-185
			float ef[3] = {0, 0, 0};
			if (const JVal *e = jm->get("emissiveFactor"))
				for (size_t k = 0; k < 3 && k < e->arr.size(); k++) ef[k] = (float) e->arr[k].num;
			const float strength = (float) es->number("emissiveStrength", 1.0);
			for (int k = 0; k < 3; k++) le[k] = ef[k] * strength;
			type = PTH_MAT_LIGHT;
		} else if (pbr) { // loader.cpp:187-298
			float base[4] = {1, 1, 1, 1};
			if (const JVal *bc = pbr->get("baseColorFactor"))
				for (size_t k = 0; k < 4 && k < bc->arr.size(); k++) base[k] = (float) bc->arr[k].num;
			const float metallic = (float) pbr->number("metallicFactor", 1.0), rough = (float) pbr->number("roughnessFactor", 1.0);
			if (base[3] < 0.99f) return fail("LoadGLTF: base colour with transparency is not supported (loader.cpp:192-197)");
			const JVal *bt = pbr->get("baseColorTexture");
			const long ti = bt ? bt->index("index") : -1;
			if (ti >= 0) { // loader.cpp:199-268
				const JVal *texs = d.root.get("textures");
				if (!texs || (size_t) ti >= texs->arr.size()) return fail("LoadGLTF: texture index %ld out of range", ti);
				const JVal &jt = texs->arr[ti];
				const long src = jt.index("source"), smp = jt.index("sampler");
				const JVal *imgs = d.root.get("images");
				if (src < 0 || !imgs || (size_t) src >= imgs->arr.size()) return fail("LoadGLTF: texture %ld has no image", ti);
				const long iv = imgs->arr[src].index("bufferView");
				if (iv < 0 || (size_t) iv >= d.views.size()) return fail("LoadGLTF: image %ld is not stored in a buffer view (the reference reads image->buffer_view only, loader.cpp:203)", src);
				const View &view = d.views[iv];
				std::vector<uint8_t> rgb;
				int w = 0, h = 0, channels = 0;
				if (int rc = decode_image_rgb8(d.buffers[view.buffer].data.data() + view.offset, view.size, rgb, &w, &h, &channels)) return rc;
				const int layer = (*num_loaded_textures)++;
				tex_index = layer;
				if ((uint32_t) layer >= num_textures) return fail("LoadGLTF: more textured primitives (%d) than textures (%u): the reference's array has one layer per glTF texture (loader.cpp:52-57)", layer + 1, num_textures);
				// (the reference hands stbir the FILE's channel count for a 3-channel buffer, loader.cpp:224-226; that is only
				//  well-defined for 3-channel files, which is what is reproduced here)
				resize_rgb8_512(rgb, w, h, out->tex.data() + (size_t) layer * 512 * 512 * 3);
				const JVal *smps = d.root.get("samplers");
				const JVal *js = (smp >= 0 && smps && (size_t) smp < smps->arr.size()) ? &smps->arr[smp] : nullptr;
				// loader.cpp:242-251: array-wide, so the LAST textured primitive's sampler wins; cgltf's defaults for absent
				// keys are 0 filters and GL_REPEAT wraps; no sampler at all = GL_LINEAR / GL_REPEAT
				out->min_filter = js ? (uint32_t) js->small("minFilter", 0) : 9729u;
				out->mag_filter = js ? (uint32_t) js->small("magFilter", 0) : 9729u;
				out->wrap_s = js ? (uint32_t) js->small("wrapS", 10497) : 10497u;
				out->wrap_t = js ? (uint32_t) js->small("wrapT", 10497) : 10497u;
			}
			for (int k = 0; k < 3; k++) diffuse[k] = base[k];
			if (metallic < 0.0001f) { // loader.cpp:270-287 (EPSILON, src/math/math.hpp:11)
				roughness = rough;
				if (roughness > 0.0001f) { type = PTH_MAT_OREN_NAYAR; roughness *= 0.35f; }
				else type = PTH_MAT_LAMBERTIAN;
			} else roughness = (float) PTH_MAT_SPECULAR_METAL; // loader.cpp:289-295: the TYPE's value lands in roughness; type stays Lambertian
		}
	}
	PtMaterial pm;
	pth_pack_material(diffuse, specular, le, roughness, tex_index, type, &pm);
	out->materials.push_back(pm);
	return PT_OK;
}

} // namespace

extern "C" {

const char *pth_last_error(void) { return g_err.c_str(); }

pth_model *pth_model_create(void) { return new pth_model(); }
void pth_model_destroy(pth_model *m) { delete m; }

int pth_load_gltf(const char *path, pth_model *out)
{
	if (!path || !out) return fail("LoadGLTF: null argument");
	Doc d;
	if (int rc = parse_document(path, d)) return rc;
	const JVal *meshes = d.root.get("meshes");
	const JVal *jtex = d.root.get("textures");
	const JVal *jmats = d.root.get("materials");
	const uint32_t num_textures = jtex ? (uint32_t) jtex->arr.size() : 0u;
	int num_loaded_textures = 0;
	if (num_textures > 0) { // loader.cpp:46-58: one zero-initialised 512 x 512 RGB8 layer per glTF texture
		out->tex.assign((size_t) num_textures * 512 * 512 * 3, 0);
		out->tex_layers = num_textures;
	}
	for (size_t mi = 0; meshes && mi < meshes->arr.size(); mi++) {
		std::vector<V3> positions, normals;
		std::vector<float> tex_coords; // 2 per entry
		std::vector<uint16_t> indices;
		const JVal *prims = meshes->arr[mi].get("primitives");
		for (size_t pi = 0; prims && pi < prims->arr.size(); pi++) {
			const JVal &prim = prims->arr[pi];
			if (prim.small("mode", 4) != 4) return fail("LoadGLTF: only triangles are supported as primitives (loader.cpp:73-79)");
			// ---- attributes, in file order (loader.cpp:81-141)
			if (const JVal *attrs = prim.get("attributes"))
				for (const auto &kv : attrs->obj) {
					const bool is_pos = kv.first == "POSITION", is_nrm = kv.first == "NORMAL", is_uv = kv.first.compare(0, 8, "TEXCOORD") == 0;
					if (!is_pos && !is_nrm && !is_uv) continue;
					const long ai = kv.second.type == JVal::Num ? JVal::to_index(kv.second.num) : -1;
					if (ai < 0 || (size_t) ai >= d.accessors.size()) return fail("LoadGLTF: attribute %s names accessor %ld", kv.first.c_str(), ai);
					const Accessor &a = d.accessors[ai];
					if (a.view < 0) return fail("LoadGLTF: attribute %s has no buffer view", kv.first.c_str());
					const View &view = d.views[a.view];
					const std::vector<uint8_t> &buf = d.buffers[view.buffer].data;
					const size_t stride = view.stride != 0 ? view.stride : a.stride;
					const size_t need = a.type == A_VEC3 ? 12 : (a.type == A_VEC2 ? 8 : 0);
					// all `count` elements must lie inside the buffer, proven before the first is read: a stride of 0 (unknown component
					// type) with a count of 1e18 would otherwise append the same element until memory runs out, and offset sums may wrap
					if (need && a.count && !span_inside(view.offset, a.offset, stride, a.count, need, buf.size()))
						return fail("LoadGLTF: attribute %s reads past its buffer", kv.first.c_str());
					for (size_t i = 0; i < a.count && need; i++) {
						const size_t at = view.offset + a.offset + stride * i;
						float f[3] = {0, 0, 0};
						memcpy(f, buf.data() + at, need);
						if (a.type == A_VEC3) {
							if (is_pos) positions.push_back(V3{f[0], f[1], f[2]});
							else if (is_nrm) normals.push_back(V3{f[0], f[1], f[2]});
							else return fail("LoadGLTF: this attribute type is unsupported (vec3 texture coordinates, loader.cpp:111-116)");
						} else {
							if (is_uv) { tex_coords.push_back(f[0]); tex_coords.push_back(f[1]); }
							else return fail("LoadGLTF: this attribute type is unsupported (vec2 position / normal, loader.cpp:128-133)");
						}
					}
				}
			// ---- indices (loader.cpp:144-162)
			{
				const long ii = prim.index("indices");
				if (ii < 0 || (size_t) ii >= d.accessors.size()) return fail("LoadGLTF: primitive without indices (the reference dereferences a null accessor here, loader.cpp:154)");
				const Accessor &a = d.accessors[ii];
				if (a.type != A_SCALAR || a.comp != 5123) return fail("LoadGLTF: indices accessor type or component type is wrong (uint16 scalars only, loader.cpp:146-152)");
				if (a.view < 0) return fail("LoadGLTF: index accessor without a buffer view");
				const View &view = d.views[a.view];
				const std::vector<uint8_t> &buf = d.buffers[view.buffer].data;
				const size_t stride = view.stride != 0 ? view.stride : a.stride;
				if (a.count && !span_inside(view.offset, 0, stride, a.count, 2, buf.size())) return fail("LoadGLTF: indices read past their buffer");
				for (size_t i = 0; i < a.count; i++) {
					const size_t at = view.offset + stride * i; // NOT + accessor offset: loader.cpp:159
					uint16_t val;
					memcpy(&val, buf.data() + at, 2);
					indices.push_back(val);
				}
			}
			// ---- material: one MaterialGLSL per primitive (loader.cpp:164-301)
			const uint32_t mat_index = (uint32_t) out->materials.size();
			{
				const long mref = prim.index("material");
				const JVal *jm = (mref >= 0 && jmats && (size_t) mref < jmats->arr.size()) ? &jmats->arr[mref] : nullptr;
				if (int rc = load_material(d, jm, out, &num_loaded_textures, num_textures)) return rc;
			}
			// ---- de-index: over the WHOLE mesh-level index array, every time (loader.cpp:303-345)
			if (indices.size() < 3) return fail("LoadGLTF: fewer than 3 indices (the reference's unsigned loop bound wraps here, loader.cpp:304)");
			if (normals.empty()) return fail("LoadGLTF: mesh without normals (the reference reads an empty array here, triangle.cpp:19-21)");
			for (size_t i = 0; i + 3 <= indices.size(); i += 3) {
				const uint16_t id[3] = {indices[i], indices[i + 1], indices[i + 2]};
				for (int k = 0; k < 3; k++) {
					if (id[k] >= positions.size() || id[k] >= normals.size() || (!tex_coords.empty() && (size_t) id[k] * 2 + 1 >= tex_coords.size()))
						return fail("LoadGLTF: index %u outside the mesh's accumulated attribute arrays", id[k]);
					out->v.push_back(positions[id[k]]);
					out->n.push_back(normals[id[k]]);
					out->uv.push_back(tex_coords.empty() ? 0.0f : tex_coords[(size_t) id[k] * 2]);
					out->uv.push_back(tex_coords.empty() ? 0.0f : tex_coords[(size_t) id[k] * 2 + 1]);
				}
				out->mat.push_back(mat_index);
			}
		}
	}
	// ---- node transforms, baked into ALL triangles node by node (loader.cpp:346-393)
	if (const JVal *nodes = d.root.get("nodes"))
		for (const JVal &node : nodes->arr) {
			if (node.get("camera")) continue;
			auto f = [](const JVal *a, size_t k) { return a && k < a->arr.size() ? (float) a->arr[k].num : 0.0f; };
			if (const JVal *mx = node.get("matrix")) {
				M4 m;
				for (int c = 0; c < 4; c++) m.c[c] = V4{f(mx, 4 * c), f(mx, 4 * c + 1), f(mx, 4 * c + 2), f(mx, 4 * c + 3)};
				out->model_matrix = m;
				apply_model_matrix(out);
				continue;
			}
			if (const JVal *s = node.get("scale")) out->model_matrix = scale(out->model_matrix, V3{f(s, 0), f(s, 2), f(s, 1)});       // y <-> z
			if (const JVal *r = node.get("rotation")) out->model_matrix = mat_mat(out->model_matrix, mat4_cast(f(r, 0), f(r, 1), f(r, 2), f(r, 3)));
			if (const JVal *t = node.get("translation")) out->model_matrix = translate(out->model_matrix, V3{f(t, 0), f(t, 2), f(t, 1)}); // y <-> z
			apply_model_matrix(out);
		}
	return PT_OK;
}

// Model::Translate / Rotate / Scale / ApplyModelMatrixToTris, src/scene/model.cpp:27-68
void pth_model_translate(pth_model *m, const float t[3]) { m->model_matrix = translate(m->model_matrix, V3{t[0], t[1], t[2]}); }
void pth_model_rotate(pth_model *m, const float r[3])
{
	m->model_matrix = rotate(m->model_matrix, r[0], V3{1.0f, 0.0f, 0.0f});
	m->model_matrix = rotate(m->model_matrix, r[1], V3{0.0f, 1.0f, 0.0f});
	m->model_matrix = rotate(m->model_matrix, r[2], V3{0.0f, 0.0f, 1.0f});
}
void pth_model_scale(pth_model *m, const float s[3]) { m->model_matrix = scale(m->model_matrix, V3{s[0], s[1], s[2]}); }
void pth_model_apply_matrix(pth_model *m) { apply_model_matrix(m); }

uint32_t pth_model_triangle_count(const pth_model *m) { return (uint32_t) m->mat.size(); }
uint32_t pth_model_material_count(const pth_model *m) { return (uint32_t) m->materials.size(); }

int pth_model_get(const pth_model *m, float *v9, float *n9, float *uv6, uint32_t *mat, void *materials48)
{
	if (!m) return PT_EINVAL;
	const size_t n = m->mat.size();
	if (n == 0 && m->materials.empty()) return PT_OK; // nothing to copy: memcpy from an empty vector's null data() is undefined even for 0 bytes
	if (v9) memcpy(v9, m->v.data(), n * 36);
	if (n9) memcpy(n9, m->n.data(), n * 36);
	if (uv6) memcpy(uv6, m->uv.data(), n * 24);
	if (mat) memcpy(mat, m->mat.data(), n * 4);
	if (materials48) memcpy(materials48, m->materials.data(), m->materials.size() * sizeof(PtMaterial));
	return PT_OK;
}

int pth_model_textures(const pth_model *m, const uint8_t **rgb8, uint32_t *layers, uint32_t sampler[3])
{
	if (rgb8) *rgb8 = m->tex.empty() ? nullptr : m->tex.data();
	if (layers) *layers = m->tex_layers;
	if (sampler) { sampler[0] = m->wrap_s; sampler[1] = m->wrap_t; sampler[2] = m->mag_filter; }
	return PT_OK;
}

// src/main.cpp:31-36: the model's triangles (Model::ConvertToSSBOFormat happens in pth_scene_finalize) and its materials
// become the scene's; materials appended later (the spheres' of src/main.cpp:44-51) follow them.
int pth_scene_add_model(pth_scene *s, const pth_model *m)
{
	if (!s || !m) return PT_EINVAL;
	const uint32_t base = pth_scene_add_packed_materials(s, m->materials.data(), (uint32_t) m->materials.size());
	std::vector<uint32_t> mat(m->mat);
	for (uint32_t &x : mat) x += base;
	if (mat.empty()) return PT_OK; // a file without meshes, or a model nothing was loaded into: materials only
	return pth_scene_add_triangles(s, &m->v[0].x, &m->n[0].x, m->uv.data(), mat.data(), (uint32_t) mat.size());
}

// The environment map as Display::InitRenderBuffer loads it (src/display/display.cpp:183-189): a Radiance .hdr read by
// stbi_load with stbi_hdr_to_ldr_gamma(1.0) and 3 channels, i.e. converted to 8 bits per channel as
// (uint8) clamp(v * 255 + 0.5, 0, 255) with v = mantissa * 2^(e - 136) in float -- and handed to glTextureSubImage2D as
// GL_UNSIGNED_BYTE.  Own RGBE reader (header, "-Y h +X w", flat or run-length scanlines); rows in file order.
// Two-call pattern: out == NULL reports the size.
int pth_load_env_hdr(const char *path, uint8_t *out, size_t capacity, uint32_t *w_out, uint32_t *h_out)
{
	if (!path) return fail("pth_load_env_hdr: null path");
	std::vector<uint8_t> f;
	if (!read_file(path, f)) return fail("pth_load_env_hdr: cannot read %s", path);
	size_t o = 0;
	auto line = [&]() { // one header line without its newline
		std::string l;
		while (o < f.size() && f[o] != '\n') l += (char) f[o++];
		if (o < f.size()) o++;
		return l;
	};
	const std::string magic = line();
	if (magic != "#?RADIANCE" && magic != "#?RGBE") return fail("pth_load_env_hdr: %s is not a Radiance HDR file", path);
	bool valid = false;
	for (;;) {
		const std::string l = line();
		if (l.empty()) break;
		if (l == "FORMAT=32-bit_rle_rgbe") valid = true;
		if (o >= f.size()) break;
	}
	if (!valid) return fail("pth_load_env_hdr: unsupported HDR format (FORMAT=32-bit_rle_rgbe expected)");
	const std::string dims = line();
	int height = 0, width = 0;
	if (sscanf(dims.c_str(), "-Y %d +X %d", &height, &width) != 2 || width <= 0 || height <= 0 || width > (1 << 24) || height > (1 << 24))
		return fail("pth_load_env_hdr: unsupported data layout '%s' (-Y h +X w expected)", dims.c_str());
	if (w_out) *w_out = (uint32_t) width;
	if (h_out) *h_out = (uint32_t) height;
	const size_t need = (size_t) width * height * 3;
	if (!out) return PT_OK;
	if (capacity < need) return fail("pth_load_env_hdr: output needs %zu bytes", need);
	auto convert = [&](const uint8_t *rgbe, uint8_t *dst) {
		float v[3] = {0.0f, 0.0f, 0.0f};
		if (rgbe[3] != 0) {
			const float f1 = (float) ldexp(1.0f, (int) rgbe[3] - (128 + 8));
			for (int k = 0; k < 3; k++) v[k] = (float) rgbe[k] * f1;
		}
		for (int k = 0; k < 3; k++) {
			float z = (float) pow((double) (v[k] * 1.0f), (double) 1.0f) * 255 + 0.5f; // scale 1, gamma 1 (display.cpp:183)
			if (z < 0) z = 0;
			if (z > 255) z = 255;
			dst[k] = (uint8_t) (int) z;
		}
	};
	std::vector<uint8_t> scan((size_t) width * 4);
	bool flat = width < 8 || width >= 32768;
	for (int j = 0; j < height; j++) {
		if (!flat) {
			if (o + 4 > f.size()) return fail("pth_load_env_hdr: truncated file");
			if (f[o] != 2 || f[o + 1] != 2 || (f[o + 2] & 0x80)) flat = true; // not run-length encoded: this and everything after it is flat RGBE
			else {
				if ((((int) f[o + 2]) << 8 | f[o + 3]) != width) return fail("pth_load_env_hdr: invalid decoded scanline length (corrupt HDR)");
				o += 4;
				for (int k = 0; k < 4; k++) {
					int i = 0;
					while (i < width) {
						if (o >= f.size()) return fail("pth_load_env_hdr: truncated file");
						int count = f[o++];
						if (count > 128) {
							count -= 128;
							if (o >= f.size() || count == 0 || count > width - i) return fail("pth_load_env_hdr: corrupt run (bad RLE data in HDR)");
							const uint8_t value = f[o++];
							for (int z = 0; z < count; z++) scan[(size_t) (i++) * 4 + k] = value;
						} else {
							if (count == 0 || count > width - i || o + (size_t) count > f.size()) return fail("pth_load_env_hdr: corrupt dump (bad RLE data in HDR)");
							for (int z = 0; z < count; z++) scan[(size_t) (i++) * 4 + k] = f[o++];
						}
					}
				}
				for (int i = 0; i < width; i++) convert(&scan[(size_t) i * 4], out + ((size_t) j * width + i) * 3);
				continue;
			}
		}
		if (o + (size_t) width * 4 > f.size()) return fail("pth_load_env_hdr: truncated file");
		for (int i = 0; i < width; i++, o += 4) convert(&f[o], out + ((size_t) j * width + i) * 3);
	}
	return PT_OK;
}

// Test hooks for the two image steps of the textured path (stbi_load_from_memory with 3 channels, stbir_resize_uint8 to
// 512 x 512: loader.cpp:209-232), so that each can be checked against the reference's own stb build.
int pth_decode_image_rgb8(const uint8_t *png, size_t n, uint8_t *rgb_out, size_t rgb_capacity, int *w, int *h, int *channels)
{
	std::vector<uint8_t> rgb;
	int ww = 0, hh = 0, ch = 0;
	if (int rc = decode_image_rgb8(png, n, rgb, &ww, &hh, &ch)) return rc;
	if (w) *w = ww;
	if (h) *h = hh;
	if (channels) *channels = ch;
	if (rgb_out) {
		if (rgb.size() > rgb_capacity) return fail("pth_decode_image_rgb8: output needs %zu bytes", rgb.size());
		memcpy(rgb_out, rgb.data(), rgb.size());
	}
	return PT_OK;
}
int pth_resize_rgb8_512(const uint8_t *rgb, int w, int h, uint8_t *out512)
{
	if (!rgb || !out512 || w <= 0 || h <= 0) return PT_EINVAL;
	std::vector<uint8_t> in(rgb, rgb + (size_t) w * h * 3);
	resize_rgb8_512(in, w, h, out512);
	return PT_OK;
}

} // extern "C"
