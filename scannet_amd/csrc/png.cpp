// png.cpp -- PNG reader / writer for the annotation images of the 2-D annotation tools: 8-bit grey (instance), 16-bit grey
// (label), 8-bit RGB / RGBA (read only; debug renderings).  Replaces FreeImageWrapper::loadImage / saveImage as
// AnnotationTools/Filter2dAnnotations/Filter2dAnnotations.cpp:340-341,400-401 and ProjectAnnotations/Visualizer.cpp:185-186 use
// them (FreeImage is an external binary dependency of mLib).  Non-interlaced images, bit depth 8 or 16, colour types 0, 2, 4, 6;
// the five scan-line filters of the PNG specification on the way in, filter 2 (Up) on the way out; zlib through this library's
// own codec (zlib_codec.cpp).  Samples of 16-bit images are big-endian in the file and host-endian (little) in memory.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

uint32_t crc_table[256];
bool crc_ready = false;
void crc_init() {
  for (uint32_t n = 0; n < 256; n++) {
    uint32_t c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    crc_table[n] = c;
  }
  crc_ready = true;
}
uint32_t crc32(const uint8_t* p, size_t n, uint32_t c = 0xFFFFFFFFu) {
  if (!crc_ready) crc_init();
  for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c;
}
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

// Decodes a PNG file.  *channels = 1, 2, 3 or 4; *bits = 8 or 16.  *data (malloc'ed, caller frees with sf_free) holds
// width * height * channels samples of 1 or 2 bytes, row-major, 16-bit samples in host byte order.
SF_API int sf_png_read(const char* path, uint32_t* width, uint32_t* height, int* channels, int* bits, void** data) {
  if (!path || !width || !height || !channels || !bits || !data) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return sf::fail(SF_ERR_IO, "could not open %s", path);
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), fp)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(fp);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) return sf::fail(SF_ERR_FORMAT, "%s is not a PNG file", path);
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0;
  std::vector<uint8_t> idat;
  size_t pos = 8;
  bool have_ihdr = false, end = false;
  while (!end && pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t* type = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) return sf::fail(SF_ERR_FORMAT, "%s: truncated chunk", path);
    const uint8_t* body = &file[pos + 8];
    if ((crc32(type, 4 + len) ^ 0xFFFFFFFFu) != be32(body + len)) return sf::fail(SF_ERR_FORMAT, "%s: chunk CRC mismatch", path);
    if (std::memcmp(type, "IHDR", 4) == 0) {
      if (len != 13) return sf::fail(SF_ERR_FORMAT, "%s: bad IHDR", path);
      w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9];
      if (body[10] != 0 || body[11] != 0) return sf::fail(SF_ERR_FORMAT, "%s: unknown compression / filter method", path);
      if (body[12] != 0) return sf::fail(SF_ERR_UNSUPPORTED, "%s: interlaced PNG is not supported", path);
      have_ihdr = true;
    } else if (std::memcmp(type, "IDAT", 4) == 0) idat.insert(idat.end(), body, body + len);
    else if (std::memcmp(type, "IEND", 4) == 0) end = true;
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || idat.empty()) return sf::fail(SF_ERR_FORMAT, "%s: missing IHDR / IDAT", path);
  int ch;
  switch (ctype) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: return sf::fail(SF_ERR_UNSUPPORTED, "%s: colour type %d (palette) is not supported", path, ctype);
  }
  if ((depth != 8 && depth != 16) || w == 0 || h == 0 || (uint64_t)w * h > (1ull << 30)) return sf::fail(SF_ERR_UNSUPPORTED, "%s: %ux%u at bit depth %d is not supported", path, w, h, depth);
  const size_t bpp = (size_t)ch * (depth / 8), stride = (size_t)w * bpp;
  std::vector<uint8_t> raw((stride + 1) * h);
  uint64_t got = 0;
  if (sf_zlib_inflate(idat.data(), idat.size(), raw.data(), raw.size(), &got) != SF_OK) return sf::fail(SF_ERR_FORMAT, "%s: %s", path, sf_last_error());
  if (got != raw.size()) return sf::fail(SF_ERR_FORMAT, "%s: image data holds %llu bytes, expected %zu", path, (unsigned long long)got, raw.size());
  uint8_t* out = (uint8_t*)std::malloc(stride * h);
  if (!out) return sf::fail(SF_ERR_IO, "out of memory");
  std::vector<uint8_t> zero(stride, 0);
  for (uint32_t y = 0; y < h; y++) {
    const uint8_t* in = &raw[(stride + 1) * y];
    const int ft = in[0];
    in++;
    uint8_t* cur = out + stride * y;
    const uint8_t* up = y ? out + stride * (y - 1) : zero.data();
    if (ft > 4) { std::free(out); return sf::fail(SF_ERR_FORMAT, "%s: unknown filter type %d", path, ft); }
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
      int v = in[i];
      switch (ft) {
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: break;
      }
      cur[i] = (uint8_t)v;
    }
  }
  if (depth == 16)
    for (size_t i = 0; i < stride * h; i += 2) { const uint8_t t = out[i]; out[i] = out[i + 1]; out[i + 1] = t; }
  *width = w; *height = h; *channels = ch; *bits = depth; *data = out;
  return SF_OK;
}

SF_API void sf_free(void* p) { std::free(p); }

// Writes a grey image (bits = 8: uint8 samples, bits = 16: uint16 samples in host byte order).
SF_API int sf_png_write_gray(const char* path, const void* data, uint32_t width, uint32_t height, int bits) {
  if (!path || !data || width == 0 || height == 0 || (bits != 8 && bits != 16)) return sf::fail(SF_ERR_INVALID_ARG, "sf_png_write_gray: bad argument");
  const size_t bpp = bits / 8, stride = (size_t)width * bpp;
  std::vector<uint8_t> raw((stride + 1) * height);
  const uint8_t* src = (const uint8_t*)data;
  // filter 2 (Up) on every row but the first: label and instance images are piecewise constant, so the difference to the row above is
  // almost all zeros -- smaller files and a faster deflate than filter 0
  std::vector<uint8_t> cur(stride), prev(stride, 0);
  for (uint32_t y = 0; y < height; y++) {
    uint8_t* o = &raw[(stride + 1) * y];
    if (bits == 8) std::memcpy(cur.data(), src + stride * y, stride);
    else for (size_t i = 0; i < stride; i += 2) { cur[i] = src[stride * y + i + 1]; cur[i + 1] = src[stride * y + i]; }
    *o++ = y == 0 ? 0 : 2;
    if (y == 0) std::memcpy(o, cur.data(), stride);
    else for (size_t i = 0; i < stride; i++) o[i] = (uint8_t)(cur[i] - prev[i]);
    cur.swap(prev);
  }
  std::vector<uint8_t> z((size_t)sf_zlib_deflate_bound(raw.size()));
  uint64_t zn = 0;
  if (sf_zlib_deflate(raw.data(), raw.size(), z.data(), z.size(), &zn) != SF_OK) return SF_ERR_IO;
  std::vector<uint8_t> f = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  auto chunk = [&](const char* type, const uint8_t* body, size_t len) {
    put32(f, (uint32_t)len);
    const size_t at = f.size();
    f.insert(f.end(), type, type + 4);
    f.insert(f.end(), body, body + len);
    put32(f, crc32(&f[at], 4 + len) ^ 0xFFFFFFFFu);
  };
  std::vector<uint8_t> ihdr;
  put32(ihdr, width); put32(ihdr, height);
  ihdr.push_back((uint8_t)bits); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk("IHDR", ihdr.data(), ihdr.size());
  chunk("IDAT", z.data(), (size_t)zn);
  chunk("IEND", nullptr, 0);
  FILE* fp = std::fopen(path, "wb");
  if (!fp) return sf::fail(SF_ERR_IO, "could not open %s for writing", path);
  const bool ok = std::fwrite(f.data(), 1, f.size(), fp) == f.size();
  std::fclose(fp);
  return ok ? SF_OK : sf::fail(SF_ERR_IO, "short write to %s", path);
}
