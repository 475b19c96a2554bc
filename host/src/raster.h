// Minimal 8-bit raster + PNG writer for NDTFrame::dumpMap (the reference draws with OpenCV, ndtframe.cpp:297-421;
// this build has no OpenCV dependency).  Shutdown-time map export only: nothing here is on the alignment path.
#ifndef NDTPSO_SLAM_AMD_RASTER_H
#define NDTPSO_SLAM_AMD_RASTER_H

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace ndtpso_host {

struct Rgb {
  uint8_t r, g, b;
};

class Raster {
 public:
  Raster(int rows, int cols, int channels, uint8_t fill)
      : rows_(rows), cols_(cols), ch_(channels), px_((size_t)rows * cols * channels, fill) {}
  int rows() const { return rows_; }
  int cols() const { return cols_; }

  void put(int x, int y, Rgb c) {  // x = column, y = row (cv::Point convention); clipped
    if (x < 0 || y < 0 || x >= cols_ || y >= rows_) return;
    uint8_t* p = &px_[((size_t)y * cols_ + x) * ch_];
    p[0] = c.r;
    if (ch_ == 3) p[1] = c.g, p[2] = c.b;
  }

  void line(int x0, int y0, int x1, int y1, Rgb c) {  // Bresenham, both end points included
    const int dx = std::abs(x1 - x0), sx = x0 < x1 ? 1 : -1;
    const int dy = -std::abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
    int err = dx + dy;
    for (;;) {
      put(x0, y0, c);
      if (x0 == x1 && y0 == y1) break;
      const int e2 = 2 * err;
      if (e2 >= dy) err += dy, x0 += sx;
      if (e2 <= dx) err += dx, y0 += sy;
    }
  }

  void circle(int cx, int cy, int r, Rgb c) {  // midpoint circle outline
    int x = r, y = 0, err = 1 - r;
    while (x >= y) {
      put(cx + x, cy + y, c), put(cx - x, cy + y, c), put(cx + x, cy - y, c), put(cx - x, cy - y, c);
      put(cx + y, cy + x, c), put(cx - y, cy + x, c), put(cx + y, cy - x, c), put(cx - y, cy - x, c);
      ++y;
      if (err < 0) {
        err += 2 * y + 1;
      } else {
        --x;
        err += 2 * (y - x) + 1;
      }
    }
  }

  // PNG with stored (uncompressed) deflate blocks: valid for every reader, no zlib needed.
  bool writePng(const char* path) const {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13];
    be32(ihdr, (uint32_t)cols_);
    be32(ihdr + 4, (uint32_t)rows_);
    ihdr[8] = 8;
    ihdr[9] = ch_ == 3 ? 2 : 0;  // truecolour / greyscale
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk(f, "IHDR", ihdr, 13);
    std::vector<uint8_t> raw;  // filter byte 0 + scanline
    const size_t stride = (size_t)cols_ * ch_;
    raw.reserve((stride + 1) * rows_);
    for (int y = 0; y < rows_; ++y) {
      raw.push_back(0);
      raw.insert(raw.end(), px_.begin() + (size_t)y * stride, px_.begin() + (size_t)(y + 1) * stride);
    }
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78), z.push_back(0x01);
    size_t pos = 0;
    do {
      const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
      z.push_back(pos + n == raw.size() ? 1 : 0);
      z.push_back((uint8_t)(n & 255)), z.push_back((uint8_t)(n >> 8));
      z.push_back((uint8_t)(~n & 255)), z.push_back((uint8_t)((~n >> 8) & 255));
      z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
      pos += n;
    } while (pos < raw.size());
    uint32_t a = 1, b = 0;  // adler32
    for (uint8_t v : raw) a = (a + v) % 65521u, b = (b + a) % 65521u;
    uint8_t ad[4];
    be32(ad, (b << 16) | a);
    z.insert(z.end(), ad, ad + 4);
    chunk(f, "IDAT", z.data(), z.size());
    chunk(f, "IEND", nullptr, 0);
    return std::fclose(f) == 0;
  }

 private:
  int rows_, cols_, ch_;
  std::vector<uint8_t> px_;

  static void be32(uint8_t* p, uint32_t v) { p[0] = v >> 24, p[1] = v >> 16, p[2] = v >> 8, p[3] = v; }
  static uint32_t crc(uint32_t c, const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      c ^= d[i];
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    }
    return c;
  }
  static void chunk(FILE* f, const char* tag, const uint8_t* d, size_t n) {
    uint8_t w[4];
    be32(w, (uint32_t)n);
    std::fwrite(w, 1, 4, f);
    std::fwrite(tag, 1, 4, f);
    if (n) std::fwrite(d, 1, n, f);
    uint32_t c = crc(0xFFFFFFFFu, (const uint8_t*)tag, 4);
    c = crc(c, d, n) ^ 0xFFFFFFFFu;
    be32(w, c);
    std::fwrite(w, 1, 4, f);
  }
};

}  // namespace ndtpso_host
#endif
