// Host-side PNG decoding for the GPU input pipeline: the reference reads every sample with
// PIL — Image.open(path).convert("RGB") for images, .convert("L") for masks (utils/dataset.py:55, 101-102) — inside four
// DataLoader worker processes.  Here a batch of PNG files is decoded by a pool of native threads straight into ONE pinned
// uint8 [N][H][W][C] buffer that the GPU transforms (input_pipeline.hip) consume; no Python object per sample.
//
// Scope = what PIL's PngImagePlugin yields for every colour type / bit depth of the PNG specification: gray of 1, 2, 4, 8 and
// 16 bits, gray+alpha / RGB / RGBA of 8 and 16 bits, palette images of 1-8 bits, non-interlaced and Adam7-interlaced; tRNS is
// ignored by convert("RGB" / "L").  16-bit samples follow PIL: colour types 2, 4, 6 are opened through the "...;16B" raw modes,
// which keep the HIGH byte; 16-bit gray is opened as mode I;16, whose conversion to "L" / "RGB" SATURATES (value > 255 -> 255,
// Convert.c I16L_L) instead of scaling.  DEFLATE is zlib's inflate; chunk walk, un-filtering (PNG filter types 0-4, per
// interlace pass) and the mode conversions are below.  convert("L") uses PIL's ITU-R 601-2 integer luma
// (19595 R + 38470 G + 7471 B + 0x8000) >> 16.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <exception>
#include <thread>
#include <vector>

#include "../../include/mi355conv.h"

void mi355_set_error(const char* fmt, ...);

namespace {

struct PngHead {
  uint32_t w = 0, h = 0;
  int depth = 0, color = 0, interlace = 0;
  int channels() const { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4; }
};

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

// Walks the chunks: header, palette, and the concatenated IDAT payload ranges.  Returns 0 or an error code (message set).
int walk(const uint8_t* f, long long n, PngHead& hd, const uint8_t*& plte, int& nplte, std::vector<std::pair<const uint8_t*, uint32_t>>* idat) {
  if (!f || n < 8 + 25 || memcmp(f, kSig, 8) != 0) {
    mi355_set_error("png: not a PNG stream");
    return MI355_ERR_ARG;
  }
  long long o = 8;
  bool have_ihdr = false;
  plte = nullptr;
  nplte = 0;
  while (o + 12 <= n) {
    const uint32_t len = be32(f + o);
    const uint8_t* type = f + o + 4;
    const uint8_t* data = f + o + 8;
    if (o + 12 + (long long)len > n) {
      mi355_set_error("png: truncated chunk");
      return MI355_ERR_ARG;
    }
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) {
        mi355_set_error("png: bad IHDR");
        return MI355_ERR_ARG;
      }
      hd.w = be32(data);
      hd.h = be32(data + 4);
      hd.depth = data[8];
      hd.color = data[9];
      hd.interlace = data[12];
      have_ihdr = true;
      if (data[10] != 0 || data[11] != 0) {
        mi355_set_error("png: unknown compression / filter method");
        return MI355_ERR_ARG;
      }
    } else if (!memcmp(type, "PLTE", 4)) {
      plte = data;
      nplte = (int)(len / 3);
    } else if (!memcmp(type, "IDAT", 4)) {
      if (idat) idat->push_back({data, len});
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    o += 12 + (long long)len;
  }
  if (!have_ihdr || hd.w == 0 || hd.h == 0) {
    mi355_set_error("png: missing IHDR");
    return MI355_ERR_ARG;
  }
  if (hd.w > 32768 || hd.h > 32768) {                 // (a corrupt header must not turn into a multi-gigabyte allocation)
    mi355_set_error("png: %ux%u exceeds the decoder's 32768 x 32768 limit", hd.w, hd.h);
    return MI355_ERR_UNSUPPORTED;
  }
  const bool depth_ok = (hd.color == 0 && (hd.depth == 1 || hd.depth == 2 || hd.depth == 4 || hd.depth == 8 || hd.depth == 16)) ||
                        (hd.color == 3 && (hd.depth == 1 || hd.depth == 2 || hd.depth == 4 || hd.depth == 8)) ||
                        ((hd.color == 2 || hd.color == 4 || hd.color == 6) && (hd.depth == 8 || hd.depth == 16));
  if (hd.interlace > 1) {
    mi355_set_error("png: unknown interlace method %d", hd.interlace);
    return MI355_ERR_UNSUPPORTED;
  }
  if (!depth_ok) {
    mi355_set_error("png: invalid colour type %d / bit depth %d", hd.color, hd.depth);
    return MI355_ERR_ARG;
  }
  if (hd.color == 3 && !plte) {
    mi355_set_error("png: palette image without PLTE");
    return MI355_ERR_ARG;
  }
  return MI355_OK;
}

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

inline uint8_t luma(int r, int g, int b) { return (uint8_t)((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16); }

int decode_one(const uint8_t* f, long long n, int want, uint8_t* out, long long cap, int expect_w, int expect_h) {
  PngHead hd;
  const uint8_t* plte;
  int nplte;
  std::vector<std::pair<const uint8_t*, uint32_t>> idat;
  int rc = walk(f, n, hd, plte, nplte, &idat);
  if (rc) return rc;
  if (want != 1 && want != 3) {
    mi355_set_error("png: channels must be 1 (convert('L')) or 3 (convert('RGB'))");
    return MI355_ERR_ARG;
  }
  if ((expect_w > 0 && (int)hd.w != expect_w) || (expect_h > 0 && (int)hd.h != expect_h)) {
    mi355_set_error("png: image is %ux%u, the batch expects %dx%d", hd.w, hd.h, expect_w, expect_h);
    return MI355_ERR_ARG;
  }
  if ((long long)hd.w * hd.h * want > cap) {
    mi355_set_error("png: output buffer too small");
    return MI355_ERR_ARG;
  }
  const int bits = hd.depth * hd.channels();
  const int bpp = bits >= 8 ? bits / 8 : 1;
  // interlace passes: (x0, y0, dx, dy); a non-interlaced image is one pass over every pixel
  static const int kAdam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const int kWhole[1][4] = {{0, 0, 1, 1}};
  const int (*passes)[4] = hd.interlace ? kAdam7 : kWhole;
  const int npass = hd.interlace ? 7 : 1;
  auto pass_dim = [&](int p, uint32_t& pw, uint32_t& ph) {
    pw = hd.w > (uint32_t)passes[p][0] ? (hd.w - passes[p][0] + passes[p][2] - 1) / passes[p][2] : 0;
    ph = hd.h > (uint32_t)passes[p][1] ? (hd.h - passes[p][1] + passes[p][3] - 1) / passes[p][3] : 0;
  };
  size_t total = 0;
  for (int p = 0; p < npass; ++p) {
    uint32_t pw, ph;
    pass_dim(p, pw, ph);
    if (pw && ph) total += (((size_t)pw * bits + 7) / 8 + 1) * ph;      // (an empty pass contributes no bytes, not even filter bytes)
  }
  std::vector<uint8_t> raw(total);
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit(&zs) != Z_OK) {
    mi355_set_error("png: inflateInit failed");
    return MI355_ERR_ARG;
  }
  zs.next_out = raw.data();
  zs.avail_out = (uInt)raw.size();
  int zr = Z_OK;
  for (size_t i = 0; i < idat.size() && zr != Z_STREAM_END; ++i) {
    zs.next_in = const_cast<Bytef*>(idat[i].first);
    zs.avail_in = idat[i].second;
    zr = inflate(&zs, Z_NO_FLUSH);
    if (zr != Z_OK && zr != Z_STREAM_END && zr != Z_BUF_ERROR) break;
  }
  const bool complete = zs.total_out == raw.size();
  inflateEnd(&zs);
  if (!complete) {
    mi355_set_error("png: corrupt or truncated image data (zlib %d)", zr);
    return MI355_ERR_ARG;
  }
  const int scale = hd.depth == 1 ? 255 : hd.depth == 2 ? 85 : hd.depth == 4 ? 17 : 1;
  const int sb = hd.depth == 16 ? 2 : 1;              // bytes per sample (16-bit: big-endian, the high byte first)
  uint8_t* base = raw.data();
  for (int p = 0; p < npass; ++p) {
    uint32_t pw, ph;
    pass_dim(p, pw, ph);
    if (!pw || !ph) continue;
    const size_t stride = ((size_t)pw * bits + 7) / 8;
    // un-filter the pass in place (row r: filter byte + stride bytes)
    for (uint32_t y = 0; y < ph; ++y) {
      uint8_t* row = base + (size_t)y * (stride + 1);
      const int ft = row[0];
      uint8_t* cur = row + 1;
      const uint8_t* up = y ? row - stride : nullptr;      // previous row's data (its filter byte sits in front of it)
      switch (ft) {
        case 0: break;
        case 1:
          for (size_t i = bpp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
          break;
        case 2:
          if (up) for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + up[i]);
          break;
        case 3:
          for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
            cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
          }
          break;
        case 4:
          for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
          }
          break;
        default:
          mi355_set_error("png: unknown filter type %d", ft);
          return MI355_ERR_ARG;
      }
    }
    // mode conversion (PIL: Image.open(...).convert("RGB" | "L")), scattered to the pass's pixel grid
    for (uint32_t y = 0; y < ph; ++y) {
      const uint8_t* src = base + (size_t)y * (stride + 1) + 1;
      uint8_t* drow = out + ((size_t)(passes[p][1] + y * passes[p][3]) * hd.w) * want;
      for (uint32_t x = 0; x < pw; ++x) {
        int r, g, b;
        if (hd.color == 0 || hd.color == 3) {
          int v;
          if (hd.depth == 16) v = src[2 * x] ? 255 : src[2 * x + 1];      // mode I;16 -> "L" / "RGB": saturates (PIL Convert.c)
          else if (hd.depth == 8) v = src[x];
          else {
            const int per = 8 / hd.depth, sh = (per - 1 - (int)(x % per)) * hd.depth;
            v = (src[x / per] >> sh) & ((1 << hd.depth) - 1);
          }
          if (hd.color == 3) {
            if (v < nplte) { r = plte[3 * v]; g = plte[3 * v + 1]; b = plte[3 * v + 2]; }
            else r = g = b = 0;
          } else {
            r = g = b = v * scale;
          }
        } else if (hd.color == 4) {
          r = g = b = src[2 * sb * x];                  // (16-bit: the high byte, raw mode LA;16B)
        } else {
          const int c = (hd.color == 2 ? 3 : 4) * sb;
          r = src[c * x]; g = src[c * x + sb]; b = src[c * x + 2 * sb];
        }
        uint8_t* dst = drow + (size_t)(passes[p][0] + x * passes[p][2]) * want;
        if (want == 3) { dst[0] = (uint8_t)r; dst[1] = (uint8_t)g; dst[2] = (uint8_t)b; }
        else dst[0] = (hd.color == 0 || hd.color == 4) ? (uint8_t)r : luma(r, g, b);
      }
    }
    base += (stride + 1) * ph;
  }
  return MI355_OK;
}

}  // namespace

extern "C" int mi355_png_info(const uint8_t* file, long long nbytes, int* W, int* H, int* color_type, int* bit_depth) {
  PngHead hd;
  const uint8_t* plte;
  int nplte;
  const int rc = walk(file, nbytes, hd, plte, nplte, nullptr);
  if (rc && rc != MI355_ERR_UNSUPPORTED) return rc;
  if (W) *W = (int)hd.w;
  if (H) *H = (int)hd.h;
  if (color_type) *color_type = hd.color;
  if (bit_depth) *bit_depth = hd.depth;
  return rc;
}

extern "C" int mi355_png_decode(const uint8_t* file, long long nbytes, int channels, uint8_t* out, long long out_bytes) {
  if (!out) {
    mi355_set_error("png_decode: null output");
    return MI355_ERR_ARG;
  }
  try {
    return decode_one(file, nbytes, channels, out, out_bytes, 0, 0);
  } catch (const std::exception& e) {                 // no C++ exception crosses the C ABI
    mi355_set_error("png_decode: %s", e.what());
    return MI355_ERR_ARG;
  }
}

extern "C" int mi355_png_decode_batch(const uint8_t* const* files, const long long* nbytes, int n, int channels, uint8_t* out,
                                      long long stride, int W, int H, int threads) {
  if (!files || !nbytes || !out || n <= 0 || W <= 0 || H <= 0 || stride < (long long)W * H * channels) {
    mi355_set_error("png_decode_batch: bad arguments");
    return MI355_ERR_ARG;
  }
  if (threads < 1) threads = 1;
  if (threads > n) threads = n;
  std::atomic<int> next(0), first_err(0), err_index(-1);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) return;
      int rc;
      try {
        rc = decode_one(files[i], nbytes[i], channels, out + (size_t)i * stride, stride, W, H);
      } catch (const std::exception&) {
        rc = MI355_ERR_ARG;
      }
      if (rc) {
        int zero = 0;
        if (first_err.compare_exchange_strong(zero, rc)) err_index = i;
      }
    }
  };
  std::vector<std::thread> pool;
  try {
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
  } catch (const std::exception&) {
    // could not start every worker: the ones that run (and this thread) still drain the queue
  }
  work();
  for (auto& t : pool) t.join();
  if (first_err.load()) {
    // (the message of the failing worker lives in ITS thread; restate it for the caller)
    mi355_set_error("png_decode_batch: image %d of the batch failed to decode (rc %d); decode it alone for the reason",
                    err_index.load(), first_err.load());
    return first_err.load();
  }
  return MI355_OK;
}
