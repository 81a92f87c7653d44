// JPEG decoder for the loader's base-colour textures: what stbi_load_from_memory(..., req_comp = 3) returns for a JPEG
// (src/loader.cpp:209-214 through the reference's vendored stb_image 2.27).  Own implementation of the format (markers,
// Huffman / progressive entropy decoding written from ITU-T T.81); the parts whose ARITHMETIC decides the output bytes
// follow stb_image's choices so that the texels equal the reference's: the integer IDCT (jidctint-style, 12-bit
// constants, +2 bits between the passes: stb_image.h:2380-2480), dequantisation in 16-bit (`(short)(v * q)`), the
// chroma upsampling filters (3:1 vertical, 3:1 horizontal, 9:3:3:1 for 2x2: stb_image.h:3411-3472; nearest otherwise)
// with stb's row pairing at the image edges (stb_image.h:3866-3885), and the 20-bit fixed-point YCbCr -> RGB
// (stb_image.h:3604-3630).  8-bit baseline, extended-sequential and progressive frames, 1 or 3 components, restart
// intervals; Adobe / "RGB"-labelled files are passed through untransformed like stb does.  4-component (CMYK / YCCK),
// arithmetic-coded, lossless and 12-bit files are refused.  tests/test_gltf_loader.py checks the output against the
// reference's own stb build on baseline / progressive / subsampled / greyscale / restart-interval files.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "image_decode.h"

namespace {

const uint8_t kDezigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                    6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                    39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
	// canonical Huffman table, T.81 annex C / F.2.2.3: codes of each length are consecutive
	uint8_t values[256];
	int mincode[18], maxcode[18], valptr[18]; // per code length 1..16
	bool valid = false;
	bool build(const uint8_t counts[16], const uint8_t *vals, int n)
	{
		memcpy(values, vals, (size_t) n);
		int code = 0, k = 0;
		for (int len = 1; len <= 16; len++) {
			valptr[len] = k;
			mincode[len] = code;
			code += counts[len - 1];
			k += counts[len - 1];
			maxcode[len] = counts[len - 1] ? code - 1 : -1;
			if (code > (1 << len)) return false;
			code <<= 1;
		}
		valid = true;
		return true;
	}
};

struct Comp {
	int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
	int x = 0, y = 0, w2 = 0, h2 = 0;
	std::vector<uint8_t> data;   // w2 x h2 samples
	std::vector<int16_t> coeff;  // progressive: 64 per block, coeff_w x coeff_h blocks
	int coeff_w = 0, coeff_h = 0;
};

struct Jpeg {
	const uint8_t *p = nullptr, *end = nullptr;
	std::string err;
	Huff hdc[4], hac[4];
	uint16_t dequant[4][64];
	Comp comp[4];
	int img_x = 0, img_y = 0, img_n = 0, h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
	bool progressive = false, jfif = false;
	int app14 = -1, rgb = 0, restart_interval = 0, todo = 0;
	int scan_n = 0, order[4] = {0, 0, 0, 0}, spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;
	// bit reader over the entropy-coded segment: byte stuffing removed, zeros fed once a marker is met
	uint32_t bitbuf = 0;
	int bits = 0, marker = 0xff;
	bool nomore = false;

	bool fail(const char *m) { if (err.empty()) err = m; return false; }
	int get8() { return p < end ? *p++ : 0; }
	int get16() { int a = get8(); return (a << 8) | get8(); }

	void fill()
	{
		while (bits <= 24) {
			int b = nomore ? 0 : get8();
			if (b == 0xff && !nomore) {
				int c = get8();
				while (c == 0xff) c = get8();
				if (c != 0) { marker = c; nomore = true; b = 0; }
			}
			bitbuf |= (uint32_t) b << (24 - bits);
			bits += 8;
		}
	}
	int getbits(int n)
	{
		if (n == 0) return 0;
		if (bits < n) fill();
		const int v = (int) (bitbuf >> (32 - n));
		bitbuf <<= n;
		bits -= n;
		return v;
	}
	int getbit() { return getbits(1); }
	int decode(const Huff &h)
	{
		if (bits < 16) fill();
		int code = 0;
		for (int len = 1; len <= 16; len++) {
			code = (code << 1) | (int) ((bitbuf >> (32 - len)) & 1u);
			if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) {
				bitbuf <<= len;
				bits -= len;
				return h.values[h.valptr[len] + code - h.mincode[len]];
			}
		}
		return -1;
	}
	// T.81 F.2.2.1 EXTEND(RECEIVE(n), n)
	int receive_extend(int n)
	{
		if (n == 0) return 0;
		const int v = getbits(n);
		return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
	}
	void reset()
	{
		bitbuf = 0; bits = 0; nomore = false; marker = 0xff;
		for (int i = 0; i < 4; i++) comp[i].dc_pred = 0;
		todo = restart_interval ? restart_interval : 0x7fffffff;
		eob_run = 0;
	}
};

inline uint8_t clamp8(int x) { return (uint8_t) (x < 0 ? 0 : (x > 255 ? 255 : x)); }

// The integer IDCT whose rounding stb_image uses (stb_image.h:2392-2478; the LLM / jidctint factorisation): 12-bit
// constants c = (int)(x * 4096 + 0.5) of the same float literals, even part from (s0 +- s4) << 12 and the s2/s6 rotation,
// odd part from the four-term rotation; the column pass keeps 2 extra bits (>> 10 after + 512), the row pass removes
// them with the +128 level shift folded in (>> 17 after + 65536 + (128 << 17)).  Integer adds commute, so only the
// constants, the shifts and the rounding terms decide the result.
// Sums and products are formed in uint32_t (wrap-around is defined there; it is what the reference's build does on
// damaged files too), the shifts are arithmetic shifts of the int32_t value.
constexpr uint32_t fx(double x) { return (uint32_t) (int) (x * 4096 + 0.5); }
struct Butterfly { uint32_t e[4], o[4]; }; // outputs k and 7-k are e[k] +- o[k]
inline Butterfly idct8(int i0, int i1, int i2, int i3, int i4, int i5, int i6, int i7)
{
	constexpr uint32_t C_0_541 = fx(0.5411961f), C_m1_847 = fx(-1.847759065f), C_0_765 = fx(0.765366865f), C_1_175 = fx(1.175875602f),
	                   C_0_298 = fx(0.298631336f), C_2_053 = fx(2.053119869f), C_3_072 = fx(3.072711026f), C_1_501 = fx(1.501321110f),
	                   C_m0_899 = fx(-0.899976223f), C_m2_562 = fx(-2.562915447f), C_m1_961 = fx(-1.961570560f), C_m0_390 = fx(-0.390180644f);
	const uint32_t s0 = (uint32_t) i0, s1 = (uint32_t) i1, s2 = (uint32_t) i2, s3 = (uint32_t) i3, s4 = (uint32_t) i4, s5 = (uint32_t) i5, s6 = (uint32_t) i6, s7 = (uint32_t) i7;
	Butterfly r;
	// even part
	const uint32_t rot = (s2 + s6) * C_0_541;
	const uint32_t ev2 = rot + s6 * C_m1_847, ev3 = rot + s2 * C_0_765;
	const uint32_t sum = (s0 + s4) * 4096u, dif = (s0 - s4) * 4096u;
	r.e[0] = sum + ev3; r.e[3] = sum - ev3; r.e[1] = dif + ev2; r.e[2] = dif - ev2;
	// odd part
	const uint32_t a73 = s7 + s3, a51 = s5 + s1, a71 = s7 + s1, a53 = s5 + s3;
	const uint32_t z5 = (a73 + a51) * C_1_175;
	const uint32_t q1 = z5 + a71 * C_m0_899, q2 = z5 + a53 * C_m2_562, q3 = a73 * C_m1_961, q4 = a51 * C_m0_390;
	r.o[0] = s1 * C_1_501 + q1 + q4;
	r.o[1] = s3 * C_3_072 + q2 + q3;
	r.o[2] = s5 * C_2_053 + q2 + q4;
	r.o[3] = s7 * C_0_298 + q1 + q3;
	return r;
}
inline int sar(uint32_t x, int n) { return (int) (int32_t) x >> n; }

void idct_block(uint8_t *out, int out_stride, const int16_t data[64])
{
	int val[64];
	for (int i = 0; i < 8; i++) { // columns
		const int16_t *d = data + i;
		int *v = val + i;
		if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
			const int dcterm = d[0] * 4; // a constant column, same scale as the general case
			for (int k = 0; k < 8; k++) v[8 * k] = dcterm;
			continue;
		}
		const Butterfly b = idct8(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56]);
		for (int k = 0; k < 4; k++) { v[8 * k] = sar(b.e[k] + 512u + b.o[k], 10); v[8 * (7 - k)] = sar(b.e[k] + 512u - b.o[k], 10); }
	}
	for (int i = 0; i < 8; i++) { // rows
		const int *v = val + 8 * i;
		uint8_t *px = out + (size_t) i * out_stride;
		const Butterfly b = idct8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
		const uint32_t bias = 65536u + (128u << 17);
		for (int k = 0; k < 4; k++) { px[k] = clamp8(sar(b.e[k] + bias + b.o[k], 17)); px[7 - k] = clamp8(sar(b.e[k] + bias - b.o[k], 17)); }
	}
}

// one 8x8 block of a sequential scan (T.81 F.2.2), dequantised in 16 bits like stb (stb_image.h:2182-2235)
bool decode_block(Jpeg &j, int16_t data[64], int c)
{
	Comp &cp = j.comp[c];
	const Huff &hdc = j.hdc[cp.hd], &hac = j.hac[cp.ha];
	const uint16_t *dq = j.dequant[cp.tq];
	const int t = j.decode(hdc);
	if (t < 0 || t > 15) return j.fail("bad huffman code");
	memset(data, 0, 64 * sizeof(int16_t));
	const int diff = t ? j.receive_extend(t) : 0;
	// a crafted stream can run the predictor out of int: wrap (as unsigned) instead of overflowing; a valid file never gets near
	const int dc = (int) ((uint32_t) cp.dc_pred + (uint32_t) diff);
	cp.dc_pred = dc;
	data[0] = (int16_t) ((uint32_t) dc * (uint32_t) dq[0]);
	int k = 1;
	do {
		const int rs = j.decode(hac);
		if (rs < 0) return j.fail("bad huffman code");
		const int s = rs & 15, r = rs >> 4;
		if (s == 0) {
			if (rs != 0xf0) break; // end of block
			k += 16;
		} else {
			k += r;
			const int zig = kDezigzag[k++];
			data[zig] = (int16_t) (j.receive_extend(s) * dq[zig]);
		}
	} while (k < 64);
	return true;
}

// progressive DC scan (T.81 G.1.2.1)
bool decode_block_prog_dc(Jpeg &j, int16_t data[64], int c)
{
	Comp &cp = j.comp[c];
	if (j.spec_end != 0) return j.fail("can't merge dc and ac");
	if (j.succ_high == 0) {
		memset(data, 0, 64 * sizeof(int16_t));
		const int t = j.decode(j.hdc[cp.hd]);
		if (t < 0 || t > 15) return j.fail("can't merge dc and ac");
		const int diff = t ? j.receive_extend(t) : 0;
		const int dc = (int) ((uint32_t) cp.dc_pred + (uint32_t) diff); // wraps instead of overflowing on a crafted stream
		cp.dc_pred = dc;
		data[0] = (int16_t) ((uint32_t) dc * (1u << j.succ_low));
	} else if (j.getbit()) data[0] = (int16_t) ((uint32_t) (uint16_t) data[0] + (1u << j.succ_low));
	return true;
}

// progressive AC scan: first pass (G.1.2.2) and refinement (G.1.2.3)
bool decode_block_prog_ac(Jpeg &j, int16_t data[64], const Huff &hac)
{
	if (j.spec_start == 0) return j.fail("can't merge dc and ac");
	if (j.succ_high == 0) {
		const int shift = j.succ_low;
		if (j.eob_run) { --j.eob_run; return true; }
		int k = j.spec_start;
		do {
			const int rs = j.decode(hac);
			if (rs < 0) return j.fail("bad huffman code");
			const int s = rs & 15, r = rs >> 4;
			if (s == 0) {
				if (r < 15) {
					j.eob_run = 1 << r;
					if (r) j.eob_run += j.getbits(r);
					--j.eob_run;
					break;
				}
				k += 16;
			} else {
				k += r;
				const int zig = kDezigzag[k++];
				data[zig] = (int16_t) (j.receive_extend(s) * (1 << shift));
			}
		} while (k <= j.spec_end);
	} else {
		const int16_t bit = (int16_t) (1 << j.succ_low);
		auto refine = [&](int16_t *p) {
			if (*p != 0 && j.getbit() && (*p & bit) == 0) { if (*p > 0) *p += bit; else *p -= bit; }
		};
		if (j.eob_run) {
			--j.eob_run;
			for (int k = j.spec_start; k <= j.spec_end; k++) refine(&data[kDezigzag[k]]);
		} else {
			int k = j.spec_start;
			do {
				const int rs = j.decode(hac);
				if (rs < 0) return j.fail("bad huffman code");
				int s = rs & 15, r = rs >> 4;
				if (s == 0) {
					if (r < 15) {
						j.eob_run = (1 << r) - 1;
						if (r) j.eob_run += j.getbits(r);
						r = 64; // force end of block
					}
					// r == 15: a run of 16 zero coefficients, refined like any run
				} else {
					if (s != 1) return j.fail("bad huffman code");
					s = j.getbit() ? bit : -bit;
				}
				while (k <= j.spec_end) {
					int16_t *p = &data[kDezigzag[k++]];
					if (*p != 0) refine(p);
					else {
						if (r == 0) { *p = (int16_t) s; break; }
						--r;
					}
				}
			} while (k <= j.spec_end);
		}
	}
	return true;
}

#define IS_RESTART(x) ((x) >= 0xd0 && (x) <= 0xd7)

// the MCU walk of one scan (T.81 A.2), restart intervals included (stb_image.h:2915-3033)
bool parse_scan(Jpeg &j)
{
	j.reset();
	int16_t block[64];
	auto after_mcu = [&]() -> int { // 0 go on, 1 stop quietly (no restart marker where one is due)
		if (--j.todo <= 0) {
			if (j.bits < 24) j.fill();
			if (!IS_RESTART(j.marker)) return 1;
			j.reset();
		}
		return 0;
	};
	if (j.scan_n == 1) {
		const int n = j.order[0];
		Comp &cp = j.comp[n];
		const int w = (cp.x + 7) >> 3, h = (cp.y + 7) >> 3;
		for (int by = 0; by < h; by++)
			for (int bx = 0; bx < w; bx++) {
				if (!j.progressive) {
					if (!decode_block(j, block, n)) return false;
					idct_block(cp.data.data() + (size_t) cp.w2 * by * 8 + bx * 8, cp.w2, block);
				} else {
					int16_t *data = cp.coeff.data() + 64 * ((size_t) bx + (size_t) by * cp.coeff_w);
					if (j.spec_start == 0) { if (!decode_block_prog_dc(j, data, n)) return false; }
					else if (!decode_block_prog_ac(j, data, j.hac[cp.ha])) return false;
				}
				if (after_mcu()) return true;
			}
		return true;
	}
	for (int my = 0; my < j.mcu_y; my++)
		for (int mx = 0; mx < j.mcu_x; mx++) {
			for (int k = 0; k < j.scan_n; k++) {
				const int n = j.order[k];
				Comp &cp = j.comp[n];
				for (int y = 0; y < cp.v; y++)
					for (int x = 0; x < cp.h; x++) {
						const int bx = mx * cp.h + x, by = my * cp.v + y;
						if (!j.progressive) {
							if (!decode_block(j, block, n)) return false;
							idct_block(cp.data.data() + (size_t) cp.w2 * by * 8 + bx * 8, cp.w2, block);
						} else if (!decode_block_prog_dc(j, cp.coeff.data() + 64 * ((size_t) bx + (size_t) by * cp.coeff_w), n)) return false;
					}
			}
			if (after_mcu()) return true;
		}
	return true;
}

bool process_marker(Jpeg &j, int m)
{
	switch (m) {
	case 0xdd: // DRI
		if (j.get16() != 4) return j.fail("bad DRI len");
		j.restart_interval = j.get16();
		return true;
	case 0xdb: { // DQT
		int L = j.get16() - 2;
		while (L > 0) {
			const int q = j.get8(), prec = q >> 4, t = q & 15;
			if ((prec != 0 && prec != 1) || t > 3) return j.fail("bad DQT");
			for (int i = 0; i < 64; i++) j.dequant[t][kDezigzag[i]] = (uint16_t) (prec ? j.get16() : j.get8());
			L -= prec ? 129 : 65;
		}
		return L == 0 || j.fail("bad DQT len");
	}
	case 0xc4: { // DHT
		int L = j.get16() - 2;
		while (L > 0) {
			const int q = j.get8(), tc = q >> 4, th = q & 15;
			if (tc > 1 || th > 3) return j.fail("bad DHT header");
			uint8_t counts[16], vals[256];
			int n = 0;
			for (int i = 0; i < 16; i++) { counts[i] = (uint8_t) j.get8(); n += counts[i]; }
			if (n > 256) return j.fail("bad DHT header");
			L -= 17;
			for (int i = 0; i < n; i++) vals[i] = (uint8_t) j.get8();
			if (!(tc == 0 ? j.hdc[th] : j.hac[th]).build(counts, vals, n)) return j.fail("bad code lengths");
			L -= n;
		}
		return L == 0 || j.fail("bad DHT len");
	}
	default: break;
	}
	if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) { // APPn, COM
		int L = j.get16();
		if (L < 2) return j.fail(m == 0xfe ? "bad COM len" : "bad APP len");
		L -= 2;
		if (m == 0xe0 && L >= 5) {
			static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
			bool ok = true;
			for (int i = 0; i < 5; i++) if (j.get8() != tag[i]) ok = false;
			L -= 5;
			if (ok) j.jfif = true;
		} else if (m == 0xee && L >= 12) {
			static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
			bool ok = true;
			for (int i = 0; i < 6; i++) if (j.get8() != tag[i]) ok = false;
			L -= 6;
			if (ok) {
				j.get8(); j.get16(); j.get16(); // version, flags0, flags1
				j.app14 = j.get8();
				L -= 6;
			}
		}
		j.p = (j.p + L <= j.end) ? j.p + L : j.end;
		return true;
	}
	return j.fail("unknown marker");
}

bool frame_header(Jpeg &j)
{
	const int Lf = j.get16();
	if (Lf < 11) return j.fail("bad SOF len");
	if (j.get8() != 8) return j.fail("only 8-bit JPEG is supported");
	j.img_y = j.get16();
	j.img_x = j.get16();
	if (j.img_y == 0 || j.img_x == 0) return j.fail("0-sized JPEG");
	if (j.img_x > 16384 || j.img_y > 16384) return j.fail("JPEG larger than 16384 pixels a side");
	const int c = j.get8();
	if (c == 4) return j.fail("4-component (CMYK / YCCK) JPEG textures are not supported");
	if (c != 1 && c != 3) return j.fail("bad component count");
	j.img_n = c;
	if (Lf != 8 + 3 * c) return j.fail("bad SOF len");
	j.rgb = 0;
	static const uint8_t rgb[3] = {'R', 'G', 'B'};
	for (int i = 0; i < c; i++) {
		Comp &cp = j.comp[i];
		cp.id = j.get8();
		if (c == 3 && cp.id == rgb[i]) ++j.rgb;
		const int q = j.get8();
		cp.h = q >> 4; cp.v = q & 15;
		if (!cp.h || cp.h > 4 || !cp.v || cp.v > 4) return j.fail("bad sampling factors");
		cp.tq = j.get8();
		if (cp.tq > 3) return j.fail("bad TQ");
	}
	j.h_max = j.v_max = 1;
	for (int i = 0; i < c; i++) { if (j.comp[i].h > j.h_max) j.h_max = j.comp[i].h; if (j.comp[i].v > j.v_max) j.v_max = j.comp[i].v; }
	for (int i = 0; i < c; i++) if (j.h_max % j.comp[i].h != 0 || j.v_max % j.comp[i].v != 0) return j.fail("bad sampling factors");
	const int mcu_w = j.h_max * 8, mcu_h = j.v_max * 8;
	j.mcu_x = (j.img_x + mcu_w - 1) / mcu_w;
	j.mcu_y = (j.img_y + mcu_h - 1) / mcu_h;
	for (int i = 0; i < c; i++) {
		Comp &cp = j.comp[i];
		cp.x = (j.img_x * cp.h + j.h_max - 1) / j.h_max;
		cp.y = (j.img_y * cp.v + j.v_max - 1) / j.v_max;
		cp.w2 = j.mcu_x * cp.h * 8;
		cp.h2 = j.mcu_y * cp.v * 8;
		cp.data.assign((size_t) cp.w2 * cp.h2, 0);
		if (j.progressive) {
			cp.coeff_w = cp.w2 / 8; cp.coeff_h = cp.h2 / 8;
			cp.coeff.assign((size_t) cp.w2 * cp.h2, 0);
		}
	}
	return true;
}

bool scan_header(Jpeg &j)
{
	const int Ls = j.get16();
	j.scan_n = j.get8();
	if (j.scan_n < 1 || j.scan_n > 4 || j.scan_n > j.img_n) return j.fail("bad SOS component count");
	if (Ls != 6 + 2 * j.scan_n) return j.fail("bad SOS len");
	for (int i = 0; i < j.scan_n; i++) {
		const int id = j.get8(), q = j.get8();
		int which = 0;
		for (; which < j.img_n; which++) if (j.comp[which].id == id) break;
		if (which == j.img_n) return j.fail("bad SOS component");
		j.comp[which].hd = q >> 4; j.comp[which].ha = q & 15;
		if (j.comp[which].hd > 3 || j.comp[which].ha > 3) return j.fail("bad huffman table index");
		j.order[i] = which;
	}
	j.spec_start = j.get8();
	j.spec_end = j.get8();
	const int aa = j.get8();
	j.succ_high = aa >> 4; j.succ_low = aa & 15;
	if (j.progressive) {
		if (j.spec_start > 63 || j.spec_end > 63 || j.spec_start > j.spec_end || j.succ_high > 13 || j.succ_low > 13) return j.fail("bad SOS");
	} else {
		if (j.spec_start != 0 || j.succ_high != 0 || j.succ_low != 0) return j.fail("bad SOS");
		j.spec_end = 63;
	}
	for (int i = 0; i < j.scan_n; i++) { // a scan may only name tables that a DHT segment defined
		const Comp &cp = j.comp[j.order[i]];
		const bool need_dc = j.progressive ? (j.spec_start == 0 && j.succ_high == 0) : true;
		const bool need_ac = j.progressive ? j.spec_start != 0 : true;
		if ((need_dc && !j.hdc[cp.hd].valid) || (need_ac && !j.hac[cp.ha].valid)) return j.fail("scan uses a Huffman table that was never defined");
	}
	return true;
}

// next marker: skip fill bytes, like stb's stbi__get_marker
int get_marker(Jpeg &j)
{
	if (j.marker != 0xff) { const int m = j.marker; j.marker = 0xff; return m; }
	int x = j.get8();
	if (x != 0xff) return 0xff;
	while (x == 0xff) x = j.get8();
	return x;
}

} // namespace

int pth_internal_decode_jpeg_rgb8(const uint8_t *d, size_t n, std::vector<uint8_t> &rgb, int *w, int *h, int *channels, std::string &err)
{
	Jpeg j;
	j.p = d; j.end = d + n;
	memset(j.dequant, 0, sizeof j.dequant);
	auto bad = [&](const char *m) { err = j.err.empty() ? m : j.err; return 1; };
	if (get_marker(j) != 0xd8) return bad("no SOI");
	int m = get_marker(j);
	while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
		if (m == 0xd9) return bad("no SOF");
		if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) return bad("lossless, hierarchical or arithmetic-coded JPEG is not supported");
		if (!process_marker(j, m)) return bad("bad marker");
		m = get_marker(j);
		while (m == 0xff) {
			if (j.p >= j.end) return bad("no SOF");
			m = get_marker(j);
		}
	}
	j.progressive = m == 0xc2;
	if (!frame_header(j)) return bad("bad frame header");
	m = get_marker(j);
	while (m != 0xd9) { // EOI
		if (m == 0xda) {
			if (!scan_header(j)) return bad("bad scan header");
			if (!parse_scan(j)) return bad("bad entropy-coded data");
			if (j.marker == 0xff) {
				// skip to the next marker (stb: "handle 0s at the end of image data from IP Kamera 9060")
				while (j.p < j.end) {
					const int x = j.get8();
					if (x == 0xff) { j.marker = j.get8(); break; }
				}
				if (j.marker == 0xff && j.p >= j.end) break;
			}
		} else if (m == 0xdc) { // DNL
			const int Ld = j.get16(), NL = j.get16();
			if (Ld != 4 || NL != j.img_y) return bad("bad DNL");
		} else if (!process_marker(j, m)) return bad("bad marker");
		m = get_marker(j);
		if (m == 0xff && j.p >= j.end) break;
	}
	if (j.progressive) // dequantise in 16 bits, then the IDCT (stb_image.h:3037-3063)
		for (int c = 0; c < j.img_n; c++) {
			Comp &cp = j.comp[c];
			const int bw = (cp.x + 7) >> 3, bh = (cp.y + 7) >> 3;
			for (int by = 0; by < bh; by++)
				for (int bx = 0; bx < bw; bx++) {
					int16_t *data = cp.coeff.data() + 64 * ((size_t) bx + (size_t) by * cp.coeff_w);
					for (int i = 0; i < 64; i++) data[i] = (int16_t) (data[i] * j.dequant[cp.tq][i]);
					idct_block(cp.data.data() + (size_t) cp.w2 * by * 8 + bx * 8, cp.w2, data);
				}
		}

	// ---- upsample row by row and convert (stb_image.h:3841-3930 for req_comp = 3)
	const bool is_rgb = j.img_n == 3 && (j.rgb == 3 || (j.app14 == 0 && !j.jfif));
	struct Res { int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } res[3];
	for (int k = 0; k < j.img_n; k++) {
		Res &r = res[k];
		r.hs = j.h_max / j.comp[k].h; r.vs = j.v_max / j.comp[k].v;
		r.ystep = r.vs >> 1;
		r.w_lores = (j.img_x + r.hs - 1) / r.hs;
		r.ypos = 0;
		r.line0 = r.line1 = j.comp[k].data.data();
		r.buf.assign((size_t) j.img_x + 3 + 8, 0);
	}
	rgb.assign((size_t) j.img_x * j.img_y * 3, 0);
	const uint8_t *cout[3] = {nullptr, nullptr, nullptr};
	for (int row = 0; row < j.img_y; row++) {
		for (int k = 0; k < j.img_n; k++) {
			Res &r = res[k];
			const bool y_bot = r.ystep >= (r.vs >> 1);
			const uint8_t *nr = y_bot ? r.line1 : r.line0, *fr = y_bot ? r.line0 : r.line1;
			uint8_t *o = r.buf.data();
			const int wl = r.w_lores;
			if (r.hs == 1 && r.vs == 1) cout[k] = nr;
			else {
				if (r.hs == 1 && r.vs == 2) {
					for (int i = 0; i < wl; i++) o[i] = (uint8_t) ((3 * nr[i] + fr[i] + 2) >> 2);
				} else if (r.hs == 2 && r.vs == 1) {
					if (wl == 1) o[0] = o[1] = nr[0];
					else {
						o[0] = nr[0];
						o[1] = (uint8_t) ((nr[0] * 3 + nr[1] + 2) >> 2);
						int i;
						for (i = 1; i < wl - 1; i++) {
							const int t = 3 * nr[i] + 2;
							o[i * 2 + 0] = (uint8_t) ((t + nr[i - 1]) >> 2);
							o[i * 2 + 1] = (uint8_t) ((t + nr[i + 1]) >> 2);
						}
						o[i * 2 + 0] = (uint8_t) ((nr[wl - 2] * 3 + nr[wl - 1] + 2) >> 2);
						o[i * 2 + 1] = nr[wl - 1];
					}
				} else if (r.hs == 2 && r.vs == 2) {
					if (wl == 1) o[0] = o[1] = (uint8_t) ((3 * nr[0] + fr[0] + 2) >> 2);
					else {
						int t1 = 3 * nr[0] + fr[0];
						o[0] = (uint8_t) ((t1 + 2) >> 2);
						for (int i = 1; i < wl; i++) {
							const int t0 = t1;
							t1 = 3 * nr[i] + fr[i];
							o[i * 2 - 1] = (uint8_t) ((3 * t0 + t1 + 8) >> 4);
							o[i * 2] = (uint8_t) ((3 * t1 + t0 + 8) >> 4);
						}
						o[wl * 2 - 1] = (uint8_t) ((t1 + 2) >> 2);
					}
				} else {
					if ((size_t) wl * r.hs > r.buf.size()) r.buf.resize((size_t) wl * r.hs);
					o = r.buf.data();
					for (int i = 0; i < wl; i++)
						for (int q = 0; q < r.hs; q++) o[i * r.hs + q] = nr[i];
				}
				cout[k] = o;
			}
			if (++r.ystep >= r.vs) {
				r.ystep = 0;
				r.line0 = r.line1;
				if (++r.ypos < j.comp[k].y) r.line1 += j.comp[k].w2;
			}
		}
		uint8_t *out = rgb.data() + (size_t) row * j.img_x * 3;
		if (j.img_n == 1) for (int i = 0; i < j.img_x; i++) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = cout[0][i];
		else if (is_rgb) for (int i = 0; i < j.img_x; i++) { out[3 * i] = cout[0][i]; out[3 * i + 1] = cout[1][i]; out[3 * i + 2] = cout[2][i]; }
		else {
#define F2FIX(x) (((int) ((x) * 4096.0f + 0.5f)) << 8)
			for (int i = 0; i < j.img_x; i++) {
				const int y_fixed = (cout[0][i] << 20) + (1 << 19);
				const int cr = cout[2][i] - 128, cb = cout[1][i] - 128;
				int r = y_fixed + cr * F2FIX(1.40200f);
				int g = y_fixed + (cr * -F2FIX(0.71414f)) + ((cb * -F2FIX(0.34414f)) & 0xffff0000);
				int b = y_fixed + cb * F2FIX(1.77200f);
				r >>= 20; g >>= 20; b >>= 20;
				out[3 * i] = clamp8(r); out[3 * i + 1] = clamp8(g); out[3 * i + 2] = clamp8(b);
			}
#undef F2FIX
		}
	}
	*w = j.img_x; *h = j.img_y;
	*channels = j.img_n >= 3 ? 3 : 1;
	return 0;
}
