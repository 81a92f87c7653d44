// host_fuzz -- mutation fuzzer for the library's own parsers of untrusted bytes (csrc/host/gltf.cpp: GLB / JSON / PNG / resize /
// Radiance HDR / PFM; csrc/host/jpeg.cpp), built with the host translation units under AddressSanitizer + UBSan
// (`make -C pathtracer_amd/csrc host-asan` -> pathtracer_amd/host_fuzz_asan; CPU only, no HIP).  The reference leans on
// cgltf / stb for these bytes and has no sanitizer target at all (CMakeLists.txt:72-81); the rebuild parses them itself, so
// it carries the target.  Any memory error or undefined behaviour aborts the process (non-zero exit); a mutated file being
// REFUSED with an error is the expected outcome and counts as "rejected".
//
//   host_fuzz <seed> <mutations per file> <file>...      (.glb .gltf | .jpg .jpeg .png | .hdr | .pfm, by extension)
//
// Deterministic: xorshift64* seeded from <seed> and the file's index.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>

#include "../include/pt355_host.h"

namespace {

struct Rng {
	uint64_t s;
	uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
	uint32_t below(uint32_t n) { return n ? (uint32_t) (next() % n) : 0u; }
};

std::vector<uint8_t> slurp(const char *path)
{
	std::vector<uint8_t> d;
	FILE *f = fopen(path, "rb");
	if (!f) return d;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	d.resize(n > 0 ? (size_t) n : 0);
	if (n > 0 && fread(d.data(), 1, d.size(), f) != d.size()) d.clear();
	fclose(f);
	return d;
}

bool spit(const std::string &path, const std::vector<uint8_t> &d)
{
	FILE *f = fopen(path.c_str(), "wb");
	if (!f) return false;
	const bool ok = d.empty() || fwrite(d.data(), 1, d.size(), f) == d.size();
	fclose(f);
	return ok;
}

// numbers a parser is most likely to mishandle, as text (JSON) and as little-endian words (binary headers, accessors)
const char *const kTextNumbers[] = {"0", "-1", "1e18", "-1e18", "4294967295", "4294967296", "18446744073709551615", "1e400", "-0", "2147483648",
                                    "0.5", "NaN", "1e-400", "9007199254740993", "65536", "5121", "5126", "99999"};
const uint32_t kWords[] = {0u, 1u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0xfffffffeu, 0x00010000u, 0xffffu, 0x7f800000u, 0xffc00000u, 16u, 0x40000000u};

void mutate(std::vector<uint8_t> &d, Rng &r, bool text_aware)
{
	const uint32_t ops = 1u + r.below(6);
	for (uint32_t o = 0; o < ops && !d.empty(); o++) {
		const uint32_t n = (uint32_t) d.size();
		switch (r.below(text_aware ? 9u : 8u)) {
		case 0: d[r.below(n)] ^= (uint8_t) (1u << r.below(8)); break;
		case 1: d[r.below(n)] = (uint8_t) r.next(); break;
		case 2: { const uint8_t v[] = {0x00, 0xff, 0x7f, 0x80, 0x01}; d[r.below(n)] = v[r.below(5)]; break; }
		case 3: if (n >= 4) { const uint32_t at = r.below(2) ? (r.below(n - 3) & ~3u) : r.below(n - 3); const uint32_t w = kWords[r.below(sizeof kWords / 4)]; memcpy(&d[at], &w, 4); } break;
		case 4: d.resize(r.below(n)); break;                                               // truncate
		case 5: { const uint32_t a = r.below(n), len = 1u + r.below(n - a < 64u ? n - a : 64u); d.erase(d.begin() + a, d.begin() + a + len); break; }
		case 6: { const uint32_t a = r.below(n), len = 1u + r.below(n - a < 64u ? n - a : 64u); std::vector<uint8_t> c(d.begin() + a, d.begin() + a + len); d.insert(d.begin() + r.below(n), c.begin(), c.end()); break; }
		case 7: if (n >= 2) { const uint32_t at = r.below(n - 1); const uint16_t w = (uint16_t) (r.below(2) ? 0xffffu : r.next()); memcpy(&d[at], &w, 2); } break;
		case 8: { // replace a run of digits by a hostile number (JSON chunk of a .glb / a .gltf)
			uint32_t at = r.below(n), scan = 0;
			while (scan < n && !(d[at] >= '0' && d[at] <= '9')) { at = at + 1 == n ? 0 : at + 1; scan++; }
			if (scan == n) break;
			uint32_t end = at;
			while (end < n && ((d[end] >= '0' && d[end] <= '9') || d[end] == '.' || d[end] == 'e' || d[end] == '-' || d[end] == '+')) end++;
			const char *t = kTextNumbers[r.below(sizeof kTextNumbers / sizeof kTextNumbers[0])];
			d.erase(d.begin() + at, d.begin() + end);
			d.insert(d.begin() + at, (const uint8_t *) t, (const uint8_t *) t + strlen(t));
			break;
		}
		}
	}
}

bool ends_with(const std::string &s, const char *suf)
{
	const size_t n = strlen(suf);
	return s.size() >= n && strcasecmp(s.c_str() + s.size() - n, suf) == 0;
}

} // namespace

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: host_fuzz <seed> <mutations per file> <file>...\n"); return 2; }
	const uint64_t seed = strtoull(argv[1], nullptr, 0);
	const uint32_t per_file = (uint32_t) strtoul(argv[2], nullptr, 0);
	char tmpl[] = "/tmp/pt355_fuzz_XXXXXX";
	if (!mkdtemp(tmpl)) { perror("mkdtemp"); return 2; }
	const std::string dir = tmpl;
	unsigned long long accepted = 0, rejected = 0, total = 0;
	std::vector<uint8_t> rgb((size_t) 4096 * 4096 * 3), big512((size_t) 512 * 512 * 3);
	std::vector<float> pfm((size_t) 256 * 256 * 4);
	for (int fi = 3; fi < argc; fi++) {
		const std::string path = argv[fi];
		const std::vector<uint8_t> orig = slurp(argv[fi]);
		if (orig.empty()) { fprintf(stderr, "host_fuzz: cannot read %s\n", argv[fi]); return 2; }
		const bool is_gltf = ends_with(path, ".glb") || ends_with(path, ".gltf");
		const bool is_img = ends_with(path, ".jpg") || ends_with(path, ".jpeg") || ends_with(path, ".png");
		const bool is_hdr = ends_with(path, ".hdr"), is_pfm = ends_with(path, ".pfm");
		if (!is_gltf && !is_img && !is_hdr && !is_pfm) { fprintf(stderr, "host_fuzz: unknown kind of file %s\n", argv[fi]); return 2; }
		Rng r{seed * 0x9E3779B97F4A7C15ull + (uint64_t) fi * 0xD1B54A32D192ED03ull + 1ull};
		const std::string tmp = dir + (ends_with(path, ".gltf") ? "/m.gltf" : is_gltf ? "/m.glb" : is_hdr ? "/m.hdr" : "/m.pfm");
		for (uint32_t it = 0; it <= per_file; it++) {
			std::vector<uint8_t> d = orig;
			if (it) mutate(d, r, is_gltf); // iteration 0: the file as it is (must be accepted)
			int rc = 0;
			if (is_img) {
				int w = 0, h = 0, ch = 0;
				rc = pth_decode_image_rgb8(d.data(), d.size(), rgb.data(), rgb.size(), &w, &h, &ch);
				if (rc == 0 && w > 0 && h > 0 && (size_t) w * h <= (size_t) 2048 * 2048) pth_resize_rgb8_512(rgb.data(), w, h, big512.data());
			} else {
				if (!spit(tmp, d)) { perror("write"); return 2; }
				if (is_gltf) {
					pth_model *m = pth_model_create();
					rc = pth_load_gltf(tmp.c_str(), m);
					if (rc == 0) { // what src/main.cpp:26-36 does with a loaded model
						const float t[3] = {0.0f, -2.0f, -6.0f}, s3[3] = {2.0f, 2.0f, 2.0f};
						pth_model_translate(m, t);
						pth_model_scale(m, s3);
						pth_model_apply_matrix(m);
						pth_scene *s = pth_scene_create();
						if (pth_scene_add_model(s, m) == 0 && pth_model_triangle_count(m) <= 200000u) pth_scene_finalize(s);
						pth_scene_destroy(s);
					}
					pth_model_destroy(m);
				} else if (is_hdr) {
					uint32_t w = 0, h = 0;
					rc = pth_load_env_hdr(tmp.c_str(), rgb.data(), rgb.size(), &w, &h);
				} else {
					rc = pth_read_pfm(tmp.c_str(), pfm.data(), 256, 256);
					if (rc) rc = pth_read_pfm(tmp.c_str(), pfm.data(), 8, 8);
				}
			}
			total++;
			if (rc == 0) accepted++; else rejected++;
			if (it == 0 && rc != 0 && !is_pfm) { fprintf(stderr, "host_fuzz: the unmutated %s was refused: %s\n", argv[fi], pth_last_error()); return 1; }
		}
	}
	unlink((dir + "/m.glb").c_str()); unlink((dir + "/m.gltf").c_str()); unlink((dir + "/m.hdr").c_str()); unlink((dir + "/m.pfm").c_str());
	rmdir(dir.c_str());
	printf("host_fuzz: %llu inputs (%d files x (1 + %u mutations)), %llu accepted, %llu rejected, no sanitizer report\n", total, argc - 3, per_file, accepted, rejected);
	return 0;
}
