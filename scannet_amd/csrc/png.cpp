// png.cpp -- PNG reader / writer for the annotation images of the 2-D annotation tools: 8-bit grey (instance), 16-bit grey
// (label), 8-bit RGB / RGBA, palette, 1/2/4-bit, Adam7-interlaced (read only); also the decoder of TYPE_PNG colour frames of a .sens.  Replaces FreeImageWrapper::loadImage / saveImage as
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

// the CRC-32 table of the PNG specification, built once (a function-local static: the decode pool of sf_fuse_run reads PNG colour frames on
// many threads, a lazily filled global table raced its own "ready" flag)
struct CrcTable {
  uint32_t t[256];
  CrcTable() {
    for (uint32_t n = 0; n < 256; n++) {
      uint32_t c = n;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[n] = c;
    }
  }
};
uint32_t crc32(const uint8_t* p, size_t n, uint32_t c = 0xFFFFFFFFu) {
  static const CrcTable table;
  for (size_t i = 0; i < n; i++) c = table.t[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c;
}
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

namespace {

struct PngImage {
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, channels = 0;   // as stored: depth 1, 2, 4, 8, 16; channels per pixel before palette expansion
  bool interlaced = false;
  std::vector<uint8_t> samples;             // w * h * channels samples, one byte each (depth <= 8, unscaled) or two (16, big-endian)
  uint8_t palette[256][3];
  uint32_t pal_len = 0;
};

// one (sub-)image of rows "filter byte + packed samples": undo the scan-line filter, unpack to one sample per byte (two at depth 16)
// into dst at dst[(y * ystep + y0) * W + x * xstep + x0]
bool unfilter_pass(const uint8_t* raw, size_t raw_len, size_t& used, uint32_t pw, uint32_t ph, int depth, int ch, PngImage& img, uint32_t x0, uint32_t y0,
                   uint32_t xstep, uint32_t ystep, const char** why) {
  const size_t row_bytes = ((size_t)pw * ch * depth + 7) / 8;
  const size_t bpp = (size_t)ch * depth >= 8 ? (size_t)ch * depth / 8 : 1;
  if (raw_len - used < (row_bytes + 1) * ph) { *why = "not enough image data"; return false; }
  std::vector<uint8_t> prev(row_bytes, 0), cur(row_bytes);
  const size_t sb = depth == 16 ? 2 : 1;   // bytes per sample in `samples`
  for (uint32_t y = 0; y < ph; y++) {
    const uint8_t* in = raw + used;
    used += row_bytes + 1;
    const int ft = in[0];
    in++;
    if (ft > 4) { *why = "unknown filter type"; return false; }
    for (size_t i = 0; i < row_bytes; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
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
    uint8_t* out_row = img.samples.data() + ((size_t)(y * ystep + y0) * img.w) * ch * sb;
    for (uint32_t x = 0; x < pw; x++) {
      uint8_t* o = out_row + (size_t)(x * xstep + x0) * ch * sb;
      if (depth >= 8) std::memcpy(o, &cur[(size_t)x * ch * sb], ch * sb);
      else {   // one channel (grey or palette index), MSB first
        const size_t bit = (size_t)x * depth;
        o[0] = (uint8_t)((cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1));
      }
    }
    cur.swap(prev);
  }
  return true;
}

// PNG in memory -> samples.  check_crc: verify the chunk CRCs (files of the annotation tools; the reference's stb reader does not).
// expect_w / expect_h (0 = any size) and max_depth are enforced on the IHDR itself, before anything is allocated: a colour frame of a .sens is
// untrusted input, and a 40-byte file may announce 2^30 pixels.
int png_decode(const uint8_t* file, size_t n, bool check_crc, PngImage& img, const char* name, uint32_t expect_w = 0, uint32_t expect_h = 0, int max_depth = 16) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (n < 33 || std::memcmp(file, sig, 8) != 0) return sf::fail(SF_ERR_FORMAT, "%s is not a PNG file", name);
  std::vector<uint8_t> idat;
  size_t pos = 8;
  bool have_ihdr = false, end = false;
  while (!end && pos + 12 <= n) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t* type = &file[pos + 4];
    if (pos + 12 + (size_t)len > n) return sf::fail(SF_ERR_FORMAT, "%s: truncated chunk", name);
    const uint8_t* body = &file[pos + 8];
    if (check_crc && (crc32(type, 4 + len) ^ 0xFFFFFFFFu) != be32(body + len)) return sf::fail(SF_ERR_FORMAT, "%s: chunk CRC mismatch", name);
    if (std::memcmp(type, "IHDR", 4) == 0) {
      if (len != 13 || have_ihdr) return sf::fail(SF_ERR_FORMAT, "%s: bad IHDR", name);
      img.w = be32(body); img.h = be32(body + 4); img.depth = body[8]; img.ctype = body[9];
      if (body[10] != 0 || body[11] != 0) return sf::fail(SF_ERR_FORMAT, "%s: unknown compression / filter method", name);
      if (body[12] > 1) return sf::fail(SF_ERR_FORMAT, "%s: unknown interlace method", name);
      img.interlaced = body[12] == 1;
      if ((expect_w && img.w != expect_w) || (expect_h && img.h != expect_h))
        return sf::fail(SF_ERR_FORMAT, "%s is %ux%u, header says %ux%u", name, img.w, img.h, expect_w, expect_h);
      if (img.depth > max_depth) return sf::fail(SF_ERR_UNSUPPORTED, "%s: %d-bit samples (limit %d)", name, img.depth, max_depth);
      have_ihdr = true;
    } else if (!have_ihdr) return sf::fail(SF_ERR_FORMAT, "%s: first chunk is not IHDR", name);
    else if (std::memcmp(type, "PLTE", 4) == 0) {
      if (len > 768 || len % 3) return sf::fail(SF_ERR_FORMAT, "%s: invalid PLTE", name);
      img.pal_len = len / 3;
      std::memcpy(img.palette, body, len);
    } else if (std::memcmp(type, "IDAT", 4) == 0) idat.insert(idat.end(), body, body + len);
    else if (std::memcmp(type, "IEND", 4) == 0) end = true;
    else if (!(type[0] & 0x20)) return sf::fail(SF_ERR_UNSUPPORTED, "%s: unknown critical chunk %.4s", name, (const char*)type);
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || idat.empty()) return sf::fail(SF_ERR_FORMAT, "%s: missing IHDR / IDAT", name);
  switch (img.ctype) {
    case 0: img.channels = 1; break;
    case 2: img.channels = 3; break;
    case 3: img.channels = 1; break;
    case 4: img.channels = 2; break;
    case 6: img.channels = 4; break;
    default: return sf::fail(SF_ERR_FORMAT, "%s: bad colour type %d", name, img.ctype);
  }
  const int d = img.depth;
  const bool depth_ok = (d == 8) || (d == 16 && img.ctype != 3) || ((d == 1 || d == 2 || d == 4) && (img.ctype == 0 || img.ctype == 3));
  if (!depth_ok || img.w == 0 || img.h == 0 || (uint64_t)img.w * img.h > (1ull << 30)) return sf::fail(SF_ERR_UNSUPPORTED, "%s: %ux%u, colour type %d at bit depth %d is not supported", name, img.w, img.h, img.ctype, d);
  if (img.ctype == 3 && img.pal_len == 0) return sf::fail(SF_ERR_FORMAT, "%s: palette image without PLTE", name);
  // the seven Adam7 passes (or the one pass of a non-interlaced image): origin and spacing
  static const uint8_t XO[7] = {0, 4, 0, 2, 0, 1, 0}, YO[7] = {0, 0, 4, 0, 2, 0, 1}, XS[7] = {8, 8, 4, 4, 2, 2, 1}, YS[7] = {8, 8, 8, 4, 4, 2, 2};
  size_t raw_len = 0;
  for (int p = 0; p < (img.interlaced ? 7 : 1); p++) {
    const uint32_t pw = img.interlaced ? (img.w - XO[p] + XS[p] - 1) / XS[p] : img.w, ph = img.interlaced ? (img.h - YO[p] + YS[p] - 1) / YS[p] : img.h;
    if (pw && ph) raw_len += (((size_t)pw * img.channels * d + 7) / 8 + 1) * ph;
  }
  // deflate cannot expand by more than 1032 : 1 (a 258-byte match from two bits): an image the IDAT bytes cannot possibly fill is
  // rejected before its buffer is allocated and zero-filled
  if (raw_len > idat.size() * 1032 + 1024) return sf::fail(SF_ERR_FORMAT, "%s: %zu bytes of image data announced, %zu compressed bytes present", name, raw_len, idat.size());
  std::vector<uint8_t> raw(raw_len);
  uint64_t got = 0;
  if (sf_zlib_inflate(idat.data(), idat.size(), raw.data(), raw.size(), &got) != SF_OK) return sf::fail(SF_ERR_FORMAT, "%s: %s", name, sf_last_error());
  if (got != raw.size()) return sf::fail(SF_ERR_FORMAT, "%s: image data holds %llu bytes, expected %zu", name, (unsigned long long)got, raw.size());
  img.samples.assign((size_t)img.w * img.h * img.channels * (d == 16 ? 2 : 1), 0);
  size_t used = 0;
  const char* why = "";
  for (int p = 0; p < (img.interlaced ? 7 : 1); p++) {
    const uint32_t pw = img.interlaced ? (img.w - XO[p] + XS[p] - 1) / XS[p] : img.w, ph = img.interlaced ? (img.h - YO[p] + YS[p] - 1) / YS[p] : img.h;
    if (!pw || !ph) continue;
    const bool ok = img.interlaced ? unfilter_pass(raw.data(), raw.size(), used, pw, ph, d, img.channels, img, XO[p], YO[p], XS[p], YS[p], &why)
                                   : unfilter_pass(raw.data(), raw.size(), used, pw, ph, d, img.channels, img, 0, 0, 1, 1, &why);
    if (!ok) return sf::fail(SF_ERR_FORMAT, "%s: %s", name, why);
  }
  return SF_OK;
}

}  // namespace

// PNG colour frame of a .sens (TYPE_PNG, sensorData.h:346-351) -> RGB8, what stbi_load_from_memory(..., 3) returns for it
// (sensorData.h:609-616; stb_image v2.08 reads 1/2/4/8-bit PNGs of every colour type, interlaced or not, and ignores chunk CRCs):
// grey of depth < 8 scaled to 0..255 (x 255 / 85 / 17), grey -> r = g = b, palette expanded, alpha dropped.
int png_decode_rgb(const uint8_t* data, uint64_t n, uint8_t* dst, uint32_t expect_w, uint32_t expect_h) {
  PngImage img;
  // size and depth (the reference decoder reads 1/2/4/8-bit PNGs only, stb_image.h:4350) are checked on the IHDR, before any allocation
  const int rc = png_decode(data, (size_t)n, false, img, "png colour frame", expect_w, expect_h, 8);
  if (rc != SF_OK) return rc;
  const size_t npx = (size_t)img.w * img.h;
  const uint8_t* s = img.samples.data();
  const int scale = img.ctype == 0 ? (img.depth == 1 ? 255 : img.depth == 2 ? 85 : img.depth == 4 ? 17 : 1) : 1;
  for (size_t i = 0; i < npx; i++) {
    uint8_t* o = dst + 3 * i;
    switch (img.ctype) {
      case 0: o[0] = o[1] = o[2] = (uint8_t)(s[i] * scale); break;
      case 4: o[0] = o[1] = o[2] = s[2 * i]; break;
      case 2: o[0] = s[3 * i]; o[1] = s[3 * i + 1]; o[2] = s[3 * i + 2]; break;
      case 6: o[0] = s[4 * i]; o[1] = s[4 * i + 1]; o[2] = s[4 * i + 2]; break;
      default: {   // palette
        if (s[i] >= img.pal_len) return sf::fail(SF_ERR_FORMAT, "png colour frame: palette index %d beyond the %u entries of PLTE", s[i], img.pal_len);
        o[0] = img.palette[s[i]][0]; o[1] = img.palette[s[i]][1]; o[2] = img.palette[s[i]][2];
      }
    }
  }
  return SF_OK;
}

// Decodes a PNG file.  *channels = 1, 2, 3 or 4; *bits = 8 or 16.  *data (malloc'ed, caller frees with sf_free) holds
// width * height * channels samples of 1 or 2 bytes, row-major, 16-bit samples in host byte order.  Palette images come back as RGB,
// grey of depth 1 / 2 / 4 as 8-bit samples scaled to 0..255.
SF_API int sf_png_read(const char* path, uint32_t* width, uint32_t* height, int* channels, int* bits, void** data) {
  if (!path || !width || !height || !channels || !bits || !data) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return sf::fail(SF_ERR_IO, "could not open %s", path);
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), fp)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(fp);
  PngImage img;
  const int rc = png_decode(file.data(), file.size(), true, img, path);
  if (rc != SF_OK) return rc;
  const size_t npx = (size_t)img.w * img.h;
  const int out_ch = img.ctype == 3 ? 3 : img.channels;
  const size_t sb = img.depth == 16 ? 2 : 1;
  uint8_t* out = (uint8_t*)std::malloc(npx * out_ch * sb);
  if (!out) return sf::fail(SF_ERR_IO, "out of memory");
  const uint8_t* s = img.samples.data();
  if (img.ctype == 3) {
    for (size_t i = 0; i < npx; i++) {
      if (s[i] >= img.pal_len) { std::free(out); return sf::fail(SF_ERR_FORMAT, "%s: palette index beyond PLTE", path); }
      std::memcpy(out + 3 * i, img.palette[s[i]], 3);
    }
  } else if (img.depth == 16) {
    for (size_t i = 0; i < npx * out_ch * 2; i += 2) { out[i] = s[i + 1]; out[i + 1] = s[i]; }
  } else {
    const int scale = img.ctype == 0 ? (img.depth == 1 ? 255 : img.depth == 2 ? 85 : img.depth == 4 ? 17 : 1) : 1;
    for (size_t i = 0; i < npx * out_ch; i++) out[i] = (uint8_t)(s[i] * scale);
  }
  *width = img.w; *height = img.h; *channels = out_ch; *bits = img.depth == 16 ? 16 : 8; *data = out;
  return SF_OK;
}

SF_API void sf_free(void* p) { std::free(p); }

// Writes a grey (channels = 1) or RGB (channels = 3) image; bits = 8: uint8 samples, bits = 16: uint16 samples in host byte order.
SF_API int sf_png_write(const char* path, const void* data, uint32_t width, uint32_t height, int channels, int bits) {
  if (!path || !data || width == 0 || height == 0 || (bits != 8 && bits != 16) || (channels != 1 && channels != 3))
    return sf::fail(SF_ERR_INVALID_ARG, "sf_png_write: bad argument (grey or RGB, 8 or 16 bits)");
  const size_t bpp = bits / 8, stride = (size_t)width * bpp * channels;
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
  ihdr.push_back((uint8_t)bits); ihdr.push_back(channels == 3 ? 2 : 0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk("IHDR", ihdr.data(), ihdr.size());
  chunk("IDAT", z.data(), (size_t)zn);
  chunk("IEND", nullptr, 0);
  FILE* fp = std::fopen(path, "wb");
  if (!fp) return sf::fail(SF_ERR_IO, "could not open %s for writing", path);
  const bool ok = std::fwrite(f.data(), 1, f.size(), fp) == f.size();
  std::fclose(fp);
  return ok ? SF_OK : sf::fail(SF_ERR_IO, "short write to %s", path);
}

SF_API int sf_png_write_gray(const char* path, const void* data, uint32_t width, uint32_t height, int bits) { return sf_png_write(path, data, width, height, 1, bits); }
