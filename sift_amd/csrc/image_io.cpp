// Image files and the result overlay of the reference's command line program (/root/reference/main.cpp:52-54 image
// ingest, :59-76 overlay, :75 imwrite) — SURVEY.md §8(f) rows 2 and 3.  Host code only (no HIP): the reference does these
// steps with vigra::importImage, cv::imread, cv::RotatedRect, cv::line and cv::imwrite; none of those libraries exists
// here, so their documented behaviour is restated:
//   * PGM / PPM (P2, P3, P5, P6) and PNG (all colour types, 1-16 bit, Adam7) decoding, zlib for the inflate; JPEG
//     (baseline and progressive Huffman files) in jpeg_decode.cpp, libjpeg's default decode restated;
//   * vigra::importImage into a scalar float array: band 0 of multi-band files (red of RGB / palette files, grey of
//     grey+alpha), sample values unscaled (0..255, 0..65535 for 16-bit files), grey samples below 8 bit expanded to
//     0..255 (png_set_expand_gray_1_2_4_to_8), SURVEY App. B-15;
//   * cv::imread(CV_LOAD_IMAGE_COLOR): three 8-bit channels in B, G, R order, grey replicated, alpha dropped, the
//     high byte of 16-bit samples;
//   * cv::RotatedRect::points, the Point2f -> Point rounding of cv::line, its clipping and 8-connected Bresenham walk
//     (OpenCV 3.2 drawing.cpp: LineIterator), cv::Size's float -> int truncation and main.cpp's u16_t coordinates.
#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sift_hip.h"

namespace sift_hip {
// jpeg_decode.cpp: libjpeg's default decode (islow IDCT, fancy upsampling, YCbCr -> RGB) restated
bool decode_jpeg(const uint8_t* data, size_t n, int& w, int& h, int& bands, std::vector<uint16_t>& px, std::string& msg);
}

namespace {

// Largest image a file may announce (2^26 pixels = 8192 x 8192; BASELINE's largest input is 3840 x 2160): a header is
// untrusted, and every decoder also checks that the file is long enough for its header BEFORE allocating for it.
constexpr long long kMaxPixels = 1LL << 26;

struct Raster {          // decoded file: interleaved samples, 8 or 16 bit per sample, as stored
    int w = 0, h = 0, bands = 0, bits = 8;
    std::vector<uint16_t> px;   // w * h * bands
};

void set_err(char* err, int errlen, const std::string& m) {
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", m.c_str());
}

bool read_file(const char* path, std::vector<uint8_t>& buf, std::string& msg) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { msg = std::string("Unable to open file '") + path + "'."; return false; }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size()) { msg = "short read"; return false; }
    return true;
}

// ---- PNM ---------------------------------------------------------------------------------------------------
struct PnmCursor {
    const std::vector<uint8_t>& b;
    size_t i;
    bool token(long& v) {   // next unsigned integer (saturating at 2^40: no signed overflow on a damaged file), skipping whitespace and # comments
        for (;;) {
            while (i < b.size() && std::isspace(b[i])) ++i;
            if (i < b.size() && b[i] == '#') { while (i < b.size() && b[i] != '\n') ++i; continue; }
            break;
        }
        if (i >= b.size() || !std::isdigit(b[i])) return false;
        v = 0;
        while (i < b.size() && std::isdigit(b[i])) {
            const long d = b[i++] - '0';
            v = v > (1L << 40) ? v : v * 10 + d;
        }
        return true;
    }
};

bool decode_pnm(const std::vector<uint8_t>& b, Raster& r, std::string& msg) {
    const int kind = b[1] - '0';   // 2 grey ascii, 3 rgb ascii, 5 grey raw, 6 rgb raw
    PnmCursor c{b, 2};
    long w, h, maxv;
    if (!c.token(w) || !c.token(h) || !c.token(maxv) || w <= 0 || h <= 0 || w > (1L << 20) || h > (1L << 20) || w * h > kMaxPixels || maxv <= 0 || maxv > 65535) { msg = "bad PNM header"; return false; }
    r.w = (int)w; r.h = (int)h; r.bands = (kind == 3 || kind == 6) ? 3 : 1; r.bits = maxv < 256 ? 8 : 16;
    const size_t n = (size_t)w * (size_t)h * (size_t)r.bands;
    // the file must be able to hold what its header announces before anything is allocated: a raw sample takes 1 or 2 bytes,
    // an ASCII sample at least 2 (a digit and a separator; the last one may lack it)
    const size_t need = (kind == 2 || kind == 3) ? 2 * n - 1 : n * (maxv < 256 ? 1 : 2);
    if (c.i >= b.size() || b.size() - c.i < need) { msg = "truncated PNM"; return false; }
    r.px.resize(n);
    if (kind == 2 || kind == 3) {
        for (size_t k = 0; k < n; ++k) { long v; if (!c.token(v)) { msg = "truncated PNM"; return false; } r.px[k] = (uint16_t)v; }
        return true;
    }
    size_t i = c.i + 1;   // exactly one whitespace byte after maxval
    const size_t bytes = n * (r.bits == 16 ? 2 : 1);
    if (i + bytes > b.size()) { msg = "truncated PNM"; return false; }
    for (size_t k = 0; k < n; ++k) r.px[k] = r.bits == 16 ? (uint16_t)((b[i + 2 * k] << 8) | b[i + 2 * k + 1]) : b[i + k];
    return true;
}

// ---- PNG ---------------------------------------------------------------------------------------------------
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the per-scanline filters of one (sub-)image in place; `raw` holds h rows of (1 + stride) bytes
bool unfilter(uint8_t* raw, int h, size_t stride, int bpp, std::string& msg) {
    std::vector<uint8_t> zero(stride, 0);
    const uint8_t* prev = zero.data();
    for (int y = 0; y < h; ++y) {
        uint8_t* row = raw + (size_t)y * (stride + 1);
        const int ft = row[0];
        uint8_t* cur = row + 1;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, bb = prev[i], cc = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int v = cur[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += bb; break;
                case 3: v += (a + bb) >> 1; break;
                case 4: v += paeth(a, bb, cc); break;
                default: msg = "bad PNG filter type"; return false;
            }
            cur[i] = (uint8_t)v;
        }
        prev = cur;
    }
    return true;
}

bool decode_png(const std::vector<uint8_t>& b, Raster& r, bool expand_low_grey, std::string& msg) {
    size_t i = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_hdr = false;
    while (i + 12 <= b.size()) {
        const uint32_t len = be32(&b[i]);
        const char* type = reinterpret_cast<const char*>(&b[i + 4]);
        if (i + 12 + len > b.size()) { msg = "truncated PNG chunk"; return false; }
        const uint8_t* d = &b[i + 8];
        if (!std::memcmp(type, "IHDR", 4) && len == 13) {
            w = (int)be32(d); h = (int)be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
            if (d[10] != 0 || d[11] != 0 || interlace > 1) { msg = "unsupported PNG method"; return false; }
            have_hdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(d, d + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        i += 12 + len;
    }
    if (!have_hdr || w <= 0 || h <= 0 || (long long)w * (long long)h > kMaxPixels) { msg = "bad PNG header"; return false; }
    const int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!chans || !depth_ok) { msg = "bad PNG colour type / bit depth"; return false; }
    const int bits_px = chans * depth;
    const int bpp = (bits_px + 7) / 8;   // filter unit
    // passes: (x0, y0, dx, dy); one pass for a non-interlaced file
    static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    struct Pass { int x0, y0, dx, dy, pw, ph; size_t stride; };
    std::vector<Pass> passes;
    size_t total = 0;
    for (int p = 0; p < (interlace ? 7 : 1); ++p) {
        Pass q{interlace ? adam7[p][0] : 0, interlace ? adam7[p][1] : 0, interlace ? adam7[p][2] : 1, interlace ? adam7[p][3] : 1, 0, 0, 0};
        q.pw = (w - q.x0 + q.dx - 1) / q.dx;
        q.ph = (h - q.y0 + q.dy - 1) / q.dy;
        if (q.pw <= 0 || q.ph <= 0) continue;
        q.stride = ((size_t)q.pw * (size_t)bits_px + 7) / 8;
        total += (size_t)q.ph * (q.stride + 1);
        passes.push_back(q);
    }
    // deflate cannot expand by more than 1032 : 1: a short IDAT stream cannot justify the buffers its header asks for
    if (idat.empty() || (unsigned long long)idat.size() * 1032ull < (unsigned long long)total) { msg = "PNG inflate failed"; return false; }
    std::vector<uint8_t> raw(total);
    uLongf out_len = (uLongf)total;
    const int zr = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || out_len != total) { msg = "PNG inflate failed"; return false; }

    // samples of the file, before palette expansion
    std::vector<uint16_t> smp((size_t)w * (size_t)h * (size_t)chans);
    size_t off = 0;
    for (const Pass& q : passes) {
        if (!unfilter(&raw[off], q.ph, q.stride, bpp, msg)) return false;
        for (int py = 0; py < q.ph; ++py) {
            const uint8_t* row = &raw[off + (size_t)py * (q.stride + 1) + 1];
            const int y = q.y0 + py * q.dy;
            for (int px = 0; px < q.pw; ++px) {
                const int x = q.x0 + px * q.dx;
                for (int c = 0; c < chans; ++c) {
                    const size_t s = (size_t)px * (size_t)chans + (size_t)c;   // sample index in the row
                    uint16_t v;
                    if (depth == 16) v = (uint16_t)((row[2 * s] << 8) | row[2 * s + 1]);
                    else if (depth == 8) v = row[s];
                    else {
                        const size_t bit = s * (size_t)depth;
                        v = (uint16_t)((row[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1));
                    }
                    smp[((size_t)y * (size_t)w + (size_t)x) * (size_t)chans + (size_t)c] = v;
                }
            }
        }
        off += (size_t)q.ph * (q.stride + 1);
    }
    r.w = w; r.h = h;
    if (ctype == 3) {   // palette -> RGB (png_set_palette_to_rgb)
        r.bands = 3; r.bits = 8;
        r.px.resize((size_t)w * (size_t)h * 3);
        for (size_t k = 0; k < (size_t)w * (size_t)h; ++k) {
            const size_t e = (size_t)smp[k] * 3;
            for (int c = 0; c < 3; ++c) r.px[3 * k + (size_t)c] = e + 2 < plte.size() ? plte[e + (size_t)c] : 0;
        }
        return true;
    }
    r.bands = chans;
    r.bits = depth == 16 ? 16 : 8;
    if (ctype == 0 && depth < 8 && expand_low_grey) {   // png_set_expand_gray_1_2_4_to_8: 0 .. 2^d - 1 -> 0 .. 255
        const int mul = 255 / ((1 << depth) - 1);
        for (auto& v : smp) v = (uint16_t)(v * mul);
    }
    r.px.swap(smp);
    return true;
}

bool decode_file_unguarded(const char* path, Raster& r, std::string& msg);

// nothing may leave the C ABI as a C++ exception: a header that asks for more memory than there is becomes an error text
bool decode_file(const char* path, Raster& r, std::string& msg) {
    try {
        return decode_file_unguarded(path, r, msg);
    } catch (const std::exception& e) {
        msg = std::string("cannot decode '") + path + "': " + e.what();
        return false;
    }
}

bool decode_file_unguarded(const char* path, Raster& r, std::string& msg) {
    std::vector<uint8_t> b;
    if (!read_file(path, b, msg)) return false;
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (b.size() >= 8 && !std::memcmp(b.data(), png_sig, 8)) return decode_png(b, r, true, msg);
    if (b.size() >= 3 && b[0] == 'P' && (b[1] == '2' || b[1] == '3' || b[1] == '5' || b[1] == '6')) return decode_pnm(b, r, msg);
    if (b.size() >= 2 && b[0] == 0xff && b[1] == 0xd8) {
        r.bits = 8;
        return sift_hip::decode_jpeg(b.data(), b.size(), r.w, r.h, r.bands, r.px, msg);
    }
    msg = "did not find a matching codec for the given file (PGM, PPM, PNG and JPEG are read)";
    return false;
}

// ---- cv::line ------------------------------------------------------------------------------------------------
int cv_round(float v) { return (int)std::lrintf(v); }   // saturate_cast<int>(float): round to nearest, ties to even

// cv::clipLine(Size2l, Point2l&, Point2l&) (OpenCV 3.2 drawing.cpp)
bool clip_line(long long width, long long height, long long& x1, long long& y1, long long& x2, long long& y2) {
    if (width <= 0 || height <= 0) return false;
    const long long right = width - 1, bottom = height - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// cv::line(img, pt1, pt2, color) with the defaults thickness 1, lineType 8, shift 0: LineIterator(img, pt1, pt2, 8, true)
void draw_line(uint8_t* bgr, int w, int h, int x1, int y1, int x2, int y2, const uint8_t color[3]) {
    long long ax = x1, ay = y1, bx = x2, by = y2;
    if ((unsigned)x1 >= (unsigned)w || (unsigned)x2 >= (unsigned)w || (unsigned)y1 >= (unsigned)h || (unsigned)y2 >= (unsigned)h)
        if (!clip_line(w, h, ax, ay, bx, by)) return;
    int px = (int)ax, py = (int)ay;
    int dx = (int)bx - (int)ax, dy = (int)by - (int)ay;
    if (dx < 0) { dx = -dx; dy = -dy; px = (int)bx; py = (int)by; }   // left to right
    int ystep = 1;
    if (dy < 0) { dy = -dy; ystep = -1; }
    const bool steep = dy > dx;
    if (steep) std::swap(dx, dy);
    int err = dx - (dy + dy);
    const int plus_delta = dx + dx, minus_delta = -(dy + dy);
    for (int k = 0; k <= dx; ++k) {
        uint8_t* p = bgr + ((size_t)py * (size_t)w + (size_t)px) * 3;
        p[0] = color[0]; p[1] = color[1]; p[2] = color[2];
        const bool neg = err < 0;
        err += minus_delta + (neg ? plus_delta : 0);
        // the iterator always advances along the major axis and, when the error went negative, along the minor one too
        if (steep) { py += ystep; if (neg) px += 1; } else { px += 1; if (neg) py += ystep; }
    }
}

uint16_t to_u16(double v) {   // main.cpp:61-62: double -> u16_t as x86 compiles it (cvttsd2si, low 16 bits)
    if (!(v > -2147483649.0 && v < 2147483648.0)) return 0;
    return (uint16_t)(uint32_t)(int32_t)v;
}

}  // namespace

extern "C" {

int sift_hip_image_info(const char* path, int* w, int* h, int* bands, int* bits, char* err, int errlen) {
    if (!path) return SIFT_HIP_EINVAL;
    Raster r;
    std::string msg;
    if (!decode_file(path, r, msg)) { set_err(err, errlen, msg); return SIFT_HIP_EPRECONDITION; }
    if (w) *w = r.w;
    if (h) *h = r.h;
    if (bands) *bands = r.bands;
    if (bits) *bits = r.bits;
    return SIFT_HIP_OK;
}

int sift_hip_image_read_band0(const char* path, float* out, long long cap, char* err, int errlen) {
    if (!path || !out) return SIFT_HIP_EINVAL;
    Raster r;
    std::string msg;
    if (!decode_file(path, r, msg)) { set_err(err, errlen, msg); return SIFT_HIP_EPRECONDITION; }
    const size_t n = (size_t)r.w * (size_t)r.h;
    if ((long long)n > cap) return SIFT_HIP_EINVAL;
    for (size_t k = 0; k < n; ++k) out[k] = (float)r.px[k * (size_t)r.bands];
    return SIFT_HIP_OK;
}

int sift_hip_image_read_bgr8(const char* path, uint8_t* out, long long cap, char* err, int errlen) {
    if (!path || !out) return SIFT_HIP_EINVAL;
    Raster r;
    std::string msg;
    if (!decode_file(path, r, msg)) { set_err(err, errlen, msg); return SIFT_HIP_EPRECONDITION; }
    const size_t n = (size_t)r.w * (size_t)r.h;
    if ((long long)(n * 3) > cap) return SIFT_HIP_EINVAL;
    const int sh = r.bits == 16 ? 8 : 0;
    for (size_t k = 0; k < n; ++k) {
        const uint16_t* p = &r.px[k * (size_t)r.bands];
        const bool colour = r.bands >= 3;
        out[3 * k + 0] = (uint8_t)((colour ? p[2] : p[0]) >> sh);
        out[3 * k + 1] = (uint8_t)((colour ? p[1] : p[0]) >> sh);
        out[3 * k + 2] = (uint8_t)(p[0] >> sh);
    }
    return SIFT_HIP_OK;
}

int sift_hip_png_write_bgr8(const char* path, const uint8_t* bgr, int w, int h, char* err, int errlen) try {
    if (!path || !bgr || w <= 0 || h <= 0) return SIFT_HIP_EINVAL;
    const size_t stride = (size_t)w * 3;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    for (int y = 0; y < h; ++y) {
        uint8_t* row = &raw[(size_t)y * (stride + 1)];
        row[0] = 0;   // filter type None
        for (int x = 0; x < w; ++x) {
            const uint8_t* s = bgr + ((size_t)y * (size_t)w + (size_t)x) * 3;
            row[1 + 3 * x] = s[2]; row[2 + 3 * x] = s[1]; row[3 + 3 * x] = s[0];
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { set_err(err, errlen, "PNG deflate failed"); return SIFT_HIP_EHIP; }
    FILE* f = std::fopen(path, "wb");
    if (!f) { set_err(err, errlen, std::string("cannot write '") + path + "'"); return SIFT_HIP_EPRECONDITION; }
    auto chunk = [&](const char* type, const uint8_t* d, uint32_t len) {
        uint8_t hdr[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len, (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(d, 1, len, f);
        uLong c = crc32(0L, hdr + 4, 4);
        if (len) c = crc32(c, d, len);
        const uint8_t cb[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
        std::fwrite(cb, 1, 4, f);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::fwrite(sig, 1, 8, f);
    const uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 2, 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zlen);
    chunk("IEND", nullptr, 0);
    const bool ok = std::fclose(f) == 0;
    if (!ok) set_err(err, errlen, "write failed");
    return ok ? SIFT_HIP_OK : SIFT_HIP_EPRECONDITION;
} catch (const std::exception& e) {
    set_err(err, errlen, e.what());
    return SIFT_HIP_EHIP;
}

// cv::RotatedRect::points (OpenCV 3.2 matrix.cpp): bottomLeft, topLeft, topRight, bottomRight
void sift_hip_rotated_rect_points(float cx, float cy, float width, float height, float angle, float* pts /* 8: x0 y0 .. x3 y3 */) {
    const double a_ = (double)angle * 3.1415926535897932384626433832795 / 180.;
    const float b = (float)std::cos(a_) * 0.5f;
    const float a = (float)std::sin(a_) * 0.5f;
    pts[0] = cx - a * height - b * width;
    pts[1] = cy + b * height - a * width;
    pts[2] = cx + a * height - b * width;
    pts[3] = cy - b * height - a * width;
    pts[4] = 2 * cx - pts[0];
    pts[5] = 2 * cy - pts[1];
    pts[6] = 2 * cx - pts[2];
    pts[7] = 2 * cy - pts[3];
}

// The box main.cpp:60-67 builds for one keypoint: centre ((loc * 2^octave) / divisor as u16_t), side (int)(scale * 10)
void sift_hip_overlay_box(const sift_hip_keypoint* kp, int subpixel, uint16_t* cx, uint16_t* cy, int* side, float* pts) {
    const int div = subpixel ? 2 : 1;
    const uint16_t x = to_u16(((double)kp->x * std::pow(2.0, (double)kp->octave)) / (double)div);
    const uint16_t y = to_u16(((double)kp->y * std::pow(2.0, (double)kp->octave)) / (double)div);
    const float s10 = kp->scale * 10;              // float product, truncated by cv::Size's int fields
    const int sd = (s10 > -2147483904.0f && s10 < 2147483648.0f) ? (int)s10 : (int)0x80000000;
    if (cx) *cx = x;
    if (cy) *cy = y;
    if (side) *side = sd;
    if (pts) sift_hip_rotated_rect_points((float)x, (float)y, (float)sd, (float)sd, kp->orientation, pts);
}

// main.cpp:59-74 on a B,G,R image: four 1-px blue lines per keypoint, in the reference's order
int sift_hip_overlay_draw(uint8_t* bgr, int w, int h, const sift_hip_keypoint* kps, long long n, int subpixel) {
    if (!bgr || w <= 0 || h <= 0 || (n > 0 && !kps)) return SIFT_HIP_EINVAL;
    static const uint8_t blue[3] = {255, 0, 0};   // cv::Scalar(255, 0, 0) on a BGR image
    for (long long k = 0; k < n; ++k) {
        float p[8];
        sift_hip_overlay_box(&kps[k], subpixel, nullptr, nullptr, nullptr, p);
        int ix[4], iy[4];
        bool finite = true;
        for (int j = 0; j < 4; ++j) {
            finite = finite && std::isfinite(p[2 * j]) && std::isfinite(p[2 * j + 1]);
            ix[j] = cv_round(p[2 * j]);
            iy[j] = cv_round(p[2 * j + 1]);
        }
        if (!finite) continue;   // NaN orientation (App. B-9, h0 = 0): cvRound of NaN is INT_MIN on x86, every line is clipped away
        draw_line(bgr, w, h, ix[0], iy[0], ix[1], iy[1], blue);
        draw_line(bgr, w, h, ix[0], iy[0], ix[3], iy[3], blue);
        draw_line(bgr, w, h, ix[2], iy[2], ix[3], iy[3], blue);
        draw_line(bgr, w, h, ix[1], iy[1], ix[2], iy[2], blue);
    }
    return SIFT_HIP_OK;
}

}  // extern "C"
