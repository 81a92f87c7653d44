// Internal: image decoders shared by the glTF loader's translation units (not part of the C ABI).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

// JPEG -> tightly packed RGB8 as stbi_load_from_memory(..., req_comp = 3) gives it; *channels = the file's component
// count (1 or 3).  Returns 0, or 1 with `err` set.
int pth_internal_decode_jpeg_rgb8(const uint8_t *d, size_t n, std::vector<uint8_t> &rgb, int *w, int *h, int *channels, std::string &err);
