// JPEG files for the image ingest of the reference's command line program (/root/reference/main.cpp:52-54 vigra::importImage,
// :59 cv::imread; the reference's own example input, example/parrot.jpg, is a JPEG) — SURVEY.md §8(f) row 2.  Host code only.
//
// Neither libjpeg nor a header of it exists in this image, so the decoder is written out here.  A JPEG file does not fix its
// decoded pixels: they depend on the decoder's inverse DCT, chroma upsampling and colour conversion.  Vigra and OpenCV both
// call libjpeg with its default settings; on the distributions that ship the reference's dependencies "libjpeg.so.8" is
// libjpeg-turbo, whose defaults are the classic IJG ones restated here, integer for integer:
//   * inverse DCT "islow" (jidctint.c): 13-bit fixed-point constants, two passes with 2 extra bits between them, output
//     range-limited after the level shift;
//   * "fancy" upsampling of subsampled chroma (jdsample.c): the triangle filters h2v1 (3/4, 1/4), h2v2 (9/16, 3/16, 3/16,
//     1/16) and h1v2 with libjpeg's alternating rounding biases, edge samples replicated, plain replication for other ratios
//     and for components at most two samples wide;
//   * YCbCr -> RGB with 16-bit fixed-point tables (jdcolor.c).
// Covered: baseline and extended sequential (SOF0 / SOF1) and progressive (SOF2) Huffman files, 8-bit samples, one
// (greyscale) or three components, restart intervals.  Not covered (an error text says so): arithmetic coding, lossless and
// hierarchical processes, 12-bit samples, four-component (CMYK / YCCK) files.  tests/test_cli_io.py compares the output with
// what libjpeg-turbo (through PIL, in the build container) returns for files of every covered kind
// (tests/golden/make_jpeg_fixtures.py), pixel for pixel.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace sift_hip {

namespace {

const uint8_t kZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                  41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                  30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  // a corrupt run can step past 63: libjpeg pads its table the same way
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffTable {
    bool defined = false;
    std::vector<uint16_t> look;   // 16-bit prefix -> length << 8 | symbol (0: no code)
    void build(const uint8_t counts[16], const uint8_t* symbols) {
        look.assign(65536, 0);
        unsigned code = 0;
        int k = 0;
        for (int len = 1; len <= 16; ++len) {
            for (int i = 0; i < counts[len - 1]; ++i, ++k) {
                const unsigned first = code << (16 - len), span = 1u << (16 - len);
                if (first + span > 65536u) return;   // over-subscribed table: the remaining prefixes stay undefined
                for (unsigned p = 0; p < span; ++p) look[first + p] = (uint16_t)((len << 8) | symbols[k]);
                ++code;
            }
            code <<= 1;
        }
        defined = true;
    }
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int n = 0;
    int marker = 0;   // marker met inside the entropy-coded data: no byte is read past it
    // jdhuff.c jpeg_fill_bit_buffer: a code or a value that needs more bits than the segment holds gets zero bits and sets
    // `insufficient_data`; the MCU in progress is finished with those zeros, the following ones up to the next restart are
    // left as they are (zero coefficients: uniform grey)
    bool insufficient = false;
    void fill() {
        while (n <= 56 && !marker) {
            unsigned byte = 0;
            if (p < end) {
                byte = *p++;
                if (byte == 0xff) {
                    while (p < end && *p == 0xff) ++p;   // fill bytes
                    if (p < end && *p == 0) ++p;         // stuffed zero: a data byte 0xff
                    else { marker = p < end ? *p++ : 0xd9; break; }
                }
            } else {
                marker = 0xd9;
                break;
            }
            acc |= (uint64_t)byte << (56 - n);
            n += 8;
        }
    }
    void starve() { insufficient = true; n = 57; }   // the accumulator's bits below the valid ones are zero already
    unsigned peek16() { if (n < 16) fill(); return (unsigned)(acc >> 48); }
    void skip(int k) { acc <<= k; n -= k; }
    unsigned receive(int k) {   // k <= 16
        if (k == 0) return 0;
        if (n < k) { fill(); if (n < k) starve(); }
        const unsigned v = (unsigned)(acc >> (64 - k));
        skip(k);
        return v;
    }
    int decode(const HuffTable& t, bool& ok) {
        const unsigned e = t.look[peek16()];
        if (!e) { if (n < 16 && marker) { starve(); return 0; } ok = false; return 0; }
        if ((int)(e >> 8) > n) starve();
        skip((int)(e >> 8));
        return (int)(e & 0xff);
    }
    void restart() { acc = 0; n = 0; marker = 0; insufficient = false; }
};

inline int extend(unsigned v, int s) { return v < (1u << (s - 1)) ? (int)v - (1 << s) + 1 : (int)v; }

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;           // tables of the current scan
    int dw = 0, dh = 0;           // real size after subsampling: ceil(W * h / hmax), ceil(H * v / vmax)
    int bw = 0, bh = 0;           // blocks allocated (whole MCUs)
    int quant[64];                // latched when the component's first scan starts
    bool quant_latched = false;
    int pred = 0;
    std::vector<int16_t> coef;    // bw * bh * 64, natural order
    std::vector<uint8_t> plane;   // (bw * 8) x (bh * 8)
};

// jidctint.c, jpeg_idct_islow: one 8x8 block of dequantised coefficients -> samples
void idct_islow(const int16_t* in, const int* q, uint8_t* out, int pitch) {
    constexpr int CB = 13, P1 = 2;
    constexpr int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                  F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    auto descale = [](long x, int n) { return (int)((x + (1L << (n - 1))) >> n); };
    int ws[64];
    for (int c = 0; c < 8; ++c) {
        const int16_t* ip = in + c;
        const int* qp = q + c;
        int* wp = ws + c;
        if (!(ip[8] | ip[16] | ip[24] | ip[32] | ip[40] | ip[48] | ip[56])) {
            const int dc = (int)(((unsigned)ip[0] * (unsigned)qp[0]) << P1);   // same bits as the signed product; no overflow on a damaged file
            for (int r = 0; r < 8; ++r) wp[8 * r] = dc;
            continue;
        }
        long z2 = (long)ip[16] * qp[16], z3 = (long)ip[48] * qp[48];
        long z1 = (z2 + z3) * F0541;
        long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        z2 = (long)ip[0] * qp[0];
        z3 = (long)ip[32] * qp[32];
        long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = (long)ip[56] * qp[56]; tmp1 = (long)ip[40] * qp[40]; tmp2 = (long)ip[24] * qp[24]; tmp3 = (long)ip[8] * qp[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        wp[0] = descale(tmp10 + tmp3, CB - P1);  wp[56] = descale(tmp10 - tmp3, CB - P1);
        wp[8] = descale(tmp11 + tmp2, CB - P1);  wp[48] = descale(tmp11 - tmp2, CB - P1);
        wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
        wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
    }
    // post-IDCT range limit: level shift by 128 and clamp, on the value's low 10 bits (libjpeg's table lookup)
    auto limit = [](int x) {
        x &= 1023;
        const int v = (x < 512 ? x : x - 1024) + 128;
        return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    };
    for (int r = 0; r < 8; ++r) {
        const int* wp = ws + 8 * r;
        uint8_t* op = out + (size_t)r * (size_t)pitch;
        long z2 = wp[2], z3 = wp[6];
        long z1 = (z2 + z3) * F0541;
        long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        long tmp0 = ((long)wp[0] + wp[4]) * (1L << CB), tmp1 = ((long)wp[0] - wp[4]) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        op[0] = limit(descale(tmp10 + tmp3, S)); op[7] = limit(descale(tmp10 - tmp3, S));
        op[1] = limit(descale(tmp11 + tmp2, S)); op[6] = limit(descale(tmp11 - tmp2, S));
        op[2] = limit(descale(tmp12 + tmp1, S)); op[5] = limit(descale(tmp12 - tmp1, S));
        op[3] = limit(descale(tmp13 + tmp0, S)); op[4] = limit(descale(tmp13 - tmp0, S));
    }
}

// jdsample.c: one component of dw x dh real samples (row pitch `pitch`) -> W x H, ratios rh = hmax / h, rv = vmax / v
void upsample(const uint8_t* src, int pitch, int dw, int dh, int rh, int rv, int W, int H, std::vector<uint8_t>& dst) {
    dst.assign((size_t)W * (size_t)H, 0);
    const bool fancy = dw > 2;
    std::vector<uint8_t> line((size_t)dw * (size_t)rh + 2);
    auto put_row = [&](int y, const uint8_t* l) { if (y < H) std::memcpy(&dst[(size_t)y * (size_t)W], l, (size_t)W); };
    if (rh == 1 && rv == 1) {
        for (int y = 0; y < H; ++y) std::memcpy(&dst[(size_t)y * (size_t)W], src + (size_t)y * (size_t)pitch, (size_t)W);
        return;
    }
    if (fancy && rh == 2 && rv == 1) {   // h2v1_fancy_upsample
        for (int y = 0; y < dh && y < H; ++y) {
            const uint8_t* in = src + (size_t)y * (size_t)pitch;
            uint8_t* o = line.data();
            int v = in[0];
            *o++ = (uint8_t)v;
            *o++ = (uint8_t)((v * 3 + in[1] + 2) >> 2);
            for (int x = 1; x < dw - 1; ++x) {
                v = in[x] * 3;
                *o++ = (uint8_t)((v + in[x - 1] + 1) >> 2);
                *o++ = (uint8_t)((v + in[x + 1] + 2) >> 2);
            }
            v = in[dw - 1];
            *o++ = (uint8_t)((v * 3 + in[dw - 2] + 1) >> 2);
            *o++ = (uint8_t)v;
            put_row(y, line.data());
        }
        return;
    }
    if (fancy && rh == 2 && rv == 2) {   // h2v2_fancy_upsample: the row above for the upper output row, the row below for the lower
        for (int y = 0; y < dh; ++y) {
            const uint8_t* in0 = src + (size_t)y * (size_t)pitch;
            for (int v = 0; v < 2; ++v) {
                const int yn = v == 0 ? (y > 0 ? y - 1 : 0) : (y < dh - 1 ? y + 1 : dh - 1);
                const uint8_t* in1 = src + (size_t)yn * (size_t)pitch;
                uint8_t* o = line.data();
                int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
                *o++ = (uint8_t)((thiscol * 4 + 8) >> 4);
                *o++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol; thiscol = nextcol;
                for (int x = 2; x < dw; ++x) {
                    nextcol = in0[x] * 3 + in1[x];
                    *o++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                    *o++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                    lastcol = thiscol; thiscol = nextcol;
                }
                *o++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                *o++ = (uint8_t)((thiscol * 4 + 7) >> 4);
                put_row(2 * y + v, line.data());
            }
        }
        return;
    }
    if (rh == 1 && rv == 2) {   // h1v2_fancy_upsample (no width condition in libjpeg-turbo)
        for (int y = 0; y < dh; ++y) {
            const uint8_t* in0 = src + (size_t)y * (size_t)pitch;
            for (int v = 0; v < 2; ++v) {
                const int yn = v == 0 ? (y > 0 ? y - 1 : 0) : (y < dh - 1 ? y + 1 : dh - 1);
                const uint8_t* in1 = src + (size_t)yn * (size_t)pitch;
                const int bias = v == 0 ? 1 : 2;
                for (int x = 0; x < dw; ++x) line[(size_t)x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
                put_row(2 * y + v, line.data());
            }
        }
        return;
    }
    // h2v1_upsample / h2v2_upsample / int_upsample: replication
    for (int y = 0; y < dh; ++y) {
        const uint8_t* in = src + (size_t)y * (size_t)pitch;
        uint8_t* o = line.data();
        for (int x = 0; x < dw; ++x)
            for (int k = 0; k < rh; ++k) *o++ = in[x];
        for (int k = 0; k < rv; ++k) put_row(y * rv + k, line.data());
    }
}

struct Decoder {
    const uint8_t* b;
    size_t n;
    std::string& msg;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_sof = false;
    bool adobe = false;
    int adobe_transform = -1;
    bool jfif = false;
    int restart_interval = 0;
    int qt[4][64];
    bool qt_defined[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    Component comp[3];
    int eobrun = 0;

    bool fail(const char* m) { msg = m; return false; }

    bool dqt(const uint8_t* p, size_t len) {
        size_t i = 0;
        while (i < len) {
            const int pq = p[i] >> 4, tq = p[i] & 15;
            ++i;
            if (tq > 3 || pq > 1) return fail("bad JPEG quantisation table");
            if (i + (size_t)(pq ? 128 : 64) > len) return fail("truncated JPEG quantisation table");
            for (int k = 0; k < 64; ++k) {
                const int v = pq ? (p[i] << 8 | p[i + 1]) : p[i];
                i += pq ? 2 : 1;
                qt[tq][kZigzag[k]] = v;
            }
            qt_defined[tq] = true;
        }
        return true;
    }

    bool dht(const uint8_t* p, size_t len) {
        size_t i = 0;
        while (i + 17 <= len) {
            const int tc = p[i] >> 4, th = p[i] & 15;
            if (tc > 1 || th > 3) return fail("bad JPEG Huffman table");
            int total = 0;
            for (int k = 0; k < 16; ++k) total += p[i + 1 + k];
            if (total > 256 || i + 17 + (size_t)total > len) return fail("bad JPEG Huffman table");
            (tc ? ac[th] : dc[th]).build(p + i + 1, p + i + 17);
            if (!(tc ? ac[th] : dc[th]).defined) return fail("bad JPEG Huffman table");
            i += 17 + (size_t)total;
        }
        return true;
    }

    bool sof(const uint8_t* p, size_t len) {
        if (len < 6 || p[0] != 8) return fail("only 8-bit JPEG samples are decoded");
        H = p[1] << 8 | p[2];
        W = p[3] << 8 | p[4];
        ncomp = p[5];
        if (W <= 0 || H <= 0) return fail("bad JPEG size");
        if ((long long)W * (long long)H > (1LL << 26)) return fail("JPEG image too large");   // a header is untrusted (image_io.cpp kMaxPixels)
        if (ncomp != 1 && ncomp != 3) return fail("only greyscale and three-component JPEG files are decoded");
        if (len < 6 + 3 * (size_t)ncomp) return fail("truncated JPEG frame header");
        for (int c = 0; c < ncomp; ++c) {
            Component& k = comp[c];
            k.id = p[6 + 3 * c];
            k.h = p[7 + 3 * c] >> 4;
            k.v = p[7 + 3 * c] & 15;
            k.tq = p[8 + 3 * c];
            if (k.h < 1 || k.h > 4 || k.v < 1 || k.v > 4 || k.tq > 3) return fail("bad JPEG component");
            hmax = k.h > hmax ? k.h : hmax;
            vmax = k.v > vmax ? k.v : vmax;
        }
        if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }   // a single component is never subsampled
        const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (int c = 0; c < ncomp; ++c) {
            Component& k = comp[c];
            if (hmax % k.h || vmax % k.v) return fail("fractional JPEG sampling ratios are not decoded");
            k.dw = (W * k.h + hmax - 1) / hmax;
            k.dh = (H * k.v + vmax - 1) / vmax;
            k.bw = mcux * k.h;
            k.bh = mcuy * k.v;   // the coefficient arrays are allocated by the first scan, once entropy-coded data is seen to exist
        }
        have_sof = true;
        return true;
    }

    // ---- entropy-coded blocks (jdhuff.c, jdphuff.c) ------------------------------------------------------------
    bool block_sequential(BitReader& br, Component& k, int16_t* blk) {
        bool ok = true;
        int s = br.decode(dc[k.td], ok);
        if (!ok || s > 16) return fail("corrupt JPEG data");
        if (s) k.pred = (int)((unsigned)k.pred + (unsigned)extend(br.receive(s), s));   // wraps on a damaged file, never overflows
        blk[0] = (int16_t)k.pred;
        for (int i = 1; i < 64; ++i) {
            const int rs = br.decode(ac[k.ta], ok);
            if (!ok) return fail("corrupt JPEG data");
            const int r = rs >> 4;
            s = rs & 15;
            if (s) {
                i += r;
                blk[kZigzag[i]] = (int16_t)extend(br.receive(s), s);
            } else {
                if (r != 15) break;
                i += 15;
            }
        }
        return true;
    }

    bool block_dc_first(BitReader& br, Component& k, int16_t* blk, int al) {
        bool ok = true;
        const int s = br.decode(dc[k.td], ok);
        if (!ok || s > 16) return fail("corrupt JPEG data");
        if (s) k.pred = (int)((unsigned)k.pred + (unsigned)extend(br.receive(s), s));
        blk[0] = (int16_t)((unsigned)k.pred << al);
        return true;
    }

    void block_dc_refine(BitReader& br, int16_t* blk, int al) {
        if (br.receive(1)) blk[0] |= (int16_t)(1 << al);
    }

    bool block_ac_first(BitReader& br, Component& k, int16_t* blk, int ss, int se, int al) {
        if (eobrun > 0) { --eobrun; return true; }
        bool ok = true;
        for (int i = ss; i <= se; ++i) {
            const int rs = br.decode(ac[k.ta], ok);
            if (!ok) return fail("corrupt JPEG data");
            const int r = rs >> 4, s = rs & 15;
            if (s) {
                i += r;
                blk[kZigzag[i]] = (int16_t)(extend(br.receive(s), s) * (1 << al));
            } else if (r == 15) {
                i += 15;
            } else {
                eobrun = 1 << r;
                if (r) eobrun += (int)br.receive(r);
                --eobrun;
                break;
            }
        }
        return true;
    }

    bool block_ac_refine(BitReader& br, Component& k, int16_t* blk, int ss, int se, int al) {
        const int p1 = 1 << al, m1 = -(1 << al);
        bool ok = true;
        int i = ss;
        if (eobrun == 0) {
            for (; i <= se; ++i) {
                const int rs = br.decode(ac[k.ta], ok);
                if (!ok) return fail("corrupt JPEG data");
                int r = rs >> 4, s = rs & 15;
                if (s) {
                    s = br.receive(1) ? p1 : m1;   // the size of a newly nonzero coefficient is always 1
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += (int)br.receive(r);
                    break;   // the rest of the band only carries correction bits
                }
                // skip r still-zero coefficients, absorbing the correction bits of the nonzero ones on the way
                do {
                    int16_t& c = blk[kZigzag[i]];
                    if (c != 0) {
                        if (br.receive(1) && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
                    } else if (--r < 0) {
                        break;
                    }
                    ++i;
                } while (i <= se);
                if (s && i <= se) blk[kZigzag[i]] = (int16_t)s;
            }
        }
        if (eobrun > 0) {
            for (; i <= se; ++i) {
                int16_t& c = blk[kZigzag[i]];
                if (c != 0 && br.receive(1) && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
            }
            --eobrun;
        }
        return true;
    }

    // one scan; `p` points at the SOS segment's payload, `len` is its length; returns the position after the scan's data
    bool scan(const uint8_t* p, size_t len, const uint8_t* data_end, const uint8_t*& next) {
        if (!have_sof) return fail("JPEG scan before the frame header");
        if (len < 1) return fail("truncated JPEG scan header");
        const int ns = p[0];
        if (ns < 1 || ns > ncomp || len < 1 + 2 * (size_t)ns + 3) return fail("bad JPEG scan header");
        Component* sc[3];
        for (int i = 0; i < ns; ++i) {
            const int id = p[1 + 2 * i];
            sc[i] = nullptr;
            for (int c = 0; c < ncomp; ++c)
                if (comp[c].id == id) sc[i] = &comp[c];
            if (!sc[i]) return fail("bad JPEG scan component");
            sc[i]->td = p[2 + 2 * i] >> 4;
            sc[i]->ta = p[2 + 2 * i] & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("bad JPEG scan tables");
            if (!sc[i]->quant_latched) {
                if (!qt_defined[sc[i]->tq]) return fail("JPEG quantisation table missing");
                std::memcpy(sc[i]->quant, qt[sc[i]->tq], sizeof(sc[i]->quant));
                sc[i]->quant_latched = true;
            }
        }
        const int ss = p[1 + 2 * ns], se = p[2 + 2 * ns], ah = p[3 + 2 * ns] >> 4, al = p[3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) return fail("bad JPEG progressive scan");
        }
        const bool need_dc = !progressive || ss == 0, need_ac = !progressive || ss > 0;
        for (int i = 0; i < ns; ++i) {
            if (need_dc && !(progressive && ah) && !dc[sc[i]->td].defined) return fail("JPEG Huffman table missing");
            if (need_ac && !ac[sc[i]->ta].defined) return fail("JPEG Huffman table missing");
        }

        // Nothing is allocated for a header alone: a block takes at least one bit of entropy-coded data in a scan, so a file
        // shorter than that cannot hold the image its frame header announces.
        {
            unsigned long long blocks = 0;
            for (int i = 0; i < ns; ++i) blocks += (unsigned long long)sc[i]->bw * (unsigned long long)sc[i]->bh;
            if ((unsigned long long)(data_end - (p + len)) * 8ull < blocks / 2) return fail("truncated JPEG data");
            for (int c = 0; c < ncomp; ++c)
                if (comp[c].coef.empty()) comp[c].coef.assign((size_t)comp[c].bw * (size_t)comp[c].bh * 64, 0);
        }
        BitReader br{p + len, data_end};
        for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
        eobrun = 0;
        auto one_block = [&](Component& k, int bx, int by) -> bool {
            int16_t* blk = &k.coef[((size_t)by * (size_t)k.bw + (size_t)bx) * 64];
            if (!progressive) return block_sequential(br, k, blk);
            if (ss == 0) {
                if (ah == 0) return block_dc_first(br, k, blk, al);
                block_dc_refine(br, blk, al);
                return true;
            }
            return ah == 0 ? block_ac_first(br, k, blk, ss, se, al) : block_ac_refine(br, k, blk, ss, se, al);
        };
        int mx, my;
        if (ns == 1) {   // non-interleaved: the component's own blocks, only those that hold real samples
            mx = (sc[0]->dw + 7) / 8;
            my = (sc[0]->dh + 7) / 8;
        } else {
            mx = (W + 8 * hmax - 1) / (8 * hmax);
            my = (H + 8 * vmax - 1) / (8 * vmax);
        }
        int until_restart = restart_interval, next_rst = 0;
        for (int y = 0; y < my; ++y)
            for (int x = 0; x < mx; ++x) {
                if (restart_interval && until_restart == 0) {
                    // jdhuff.c process_restart + jdmarker.c read_restart_marker: drop the bits left, take the marker the reader ran
                    // into (or the next one in the stream); the expected RSTn is swallowed, anything else goes through libjpeg's
                    // default jpeg_resync_to_restart.  A damaged file is decoded, not refused (libjpeg only warns).
                    int m = br.marker;
                    const uint8_t* q = br.p;
                    auto next_marker = [&]() -> int {   // jdmarker.c next_marker: the next 0xff that is followed by neither 0x00 nor 0xff
                        for (;;) {
                            while (q < data_end && *q != 0xff) ++q;
                            while (q < data_end && *q == 0xff) ++q;
                            if (q >= data_end) return 0xd9;
                            if (*q != 0) return *q++;
                            ++q;
                        }
                    };
                    if (!m) m = next_marker();
                    bool starved = false;   // the marker stays unread: the segment up to the next restart has no data (zero blocks)
                    if (m != 0xd0 + next_rst) {
                        for (;;) {
                            int action;
                            if (m < 0xc0) action = 2;                                   // not a marker libjpeg knows: skip it
                            else if (m < 0xd0 || m > 0xd7) action = 3;                  // a real non-restart marker: stop in front of it
                            else if (m == 0xd0 + ((next_rst + 1) & 7) || m == 0xd0 + ((next_rst + 2) & 7)) action = 3;   // one of the next two restarts
                            else if (m == 0xd0 + ((next_rst - 1) & 7) || m == 0xd0 + ((next_rst - 2) & 7)) action = 2;   // an earlier restart: advance
                            else action = 1;                                            // the expected one, or too far away: resume after it
                            if (action == 2) { m = next_marker(); continue; }
                            starved = action == 3;
                            break;
                        }
                    }
                    br.restart();
                    br.p = q;
                    // jdhuff.c process_restart: with the marker left unread the segment is empty - its first MCU is decoded from
                    // zero bits (which sets insufficient_data), the rest is skipped
                    if (starved) br.marker = m;
                    next_rst = (next_rst + 1) & 7;
                    for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                    eobrun = 0;
                    until_restart = restart_interval;
                }
                if (br.insufficient) {          // jdhuff.c: "if (!entropy->pub.insufficient_data)" - the MCU's blocks keep their zeros
                    --until_restart;
                    continue;
                }
                if (ns == 1) {
                    if (!one_block(*sc[0], x, y)) return false;
                } else {
                    for (int i = 0; i < ns; ++i)
                        for (int v = 0; v < sc[i]->v; ++v)
                            for (int h = 0; h < sc[i]->h; ++h)
                                if (!one_block(*sc[i], x * sc[i]->h + h, y * sc[i]->v + v)) return false;
                }
                --until_restart;
            }
        // position after the entropy-coded segment: the marker the reader ran into, or the next one in the stream
        if (br.marker) {
            next = br.p - 2;
        } else {
            const uint8_t* q = br.p - (br.n / 8);   // bytes still in the accumulator were not consumed
            if (q < p + len) q = p + len;
            while (q + 1 < data_end && !(q[0] == 0xff && q[1] != 0 && q[1] != 0xff && !(q[1] >= 0xd0 && q[1] <= 0xd7))) ++q;
            next = q;
        }
        return true;
    }

    bool run(int& w, int& h, int& bands, std::vector<uint16_t>& px) {
        if (n < 4 || b[0] != 0xff || b[1] != 0xd8) return fail("not a JPEG file");
        const uint8_t* p = b + 2;
        const uint8_t* end = b + n;
        bool done = false, any_scan = false;
        while (!done) {
            while (p < end && *p != 0xff) ++p;   // garbage before a marker is skipped
            while (p < end && *p == 0xff) ++p;
            if (p >= end) break;
            const int m = *p++;
            if (m == 0xd9) break;                                  // EOI
            if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;   // standalone markers
            if (p + 2 > end) break;
            const size_t len = (size_t)(p[0] << 8 | p[1]);
            if (len < 2 || p + len > end) return fail("truncated JPEG segment");
            const uint8_t* seg = p + 2;
            const size_t sl = len - 2;
            switch (m) {
                case 0xc0: case 0xc1: case 0xc2:
                    if (have_sof) return fail("several JPEG frame headers");
                    progressive = m == 0xc2;
                    if (!sof(seg, sl)) return false;
                    break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    return fail("this JPEG process (lossless, hierarchical or arithmetic coding) is not decoded");
                case 0xc4: if (!dht(seg, sl)) return false; break;
                case 0xcc: return fail("arithmetic-coded JPEG files are not decoded");
                case 0xdb: if (!dqt(seg, sl)) return false; break;
                case 0xdd:
                    if (sl < 2) return fail("truncated JPEG restart interval");
                    restart_interval = seg[0] << 8 | seg[1];
                    break;
                case 0xe0: if (sl >= 5 && !std::memcmp(seg, "JFIF", 5)) jfif = true; break;
                case 0xee:
                    if (sl >= 12 && !std::memcmp(seg, "Adobe", 5)) { adobe = true; adobe_transform = seg[11]; }
                    break;
                case 0xda: {
                    const uint8_t* next = nullptr;
                    if (!scan(seg, sl, end, next)) return false;
                    any_scan = true;
                    p = next;
                    continue;
                }
                default: break;
            }
            p += len;
        }
        if (!have_sof || !any_scan) return fail("JPEG file without image data");

        // coefficients -> samples, component by component
        for (int c = 0; c < ncomp; ++c) {
            Component& k = comp[c];
            if (!k.quant_latched) return fail("JPEG component without a scan");
            const int pitch = k.bw * 8;
            k.plane.assign((size_t)pitch * (size_t)k.bh * 8, 0);
            for (int by = 0; by < k.bh; ++by)
                for (int bx = 0; bx < k.bw; ++bx)
                    idct_islow(&k.coef[((size_t)by * (size_t)k.bw + (size_t)bx) * 64], k.quant, &k.plane[(size_t)by * 8 * (size_t)pitch + (size_t)bx * 8], pitch);
            std::vector<int16_t>().swap(k.coef);
        }
        w = W; h = H;
        if (ncomp == 1) {
            bands = 1;
            px.resize((size_t)W * (size_t)H);
            const int pitch = comp[0].bw * 8;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) px[(size_t)y * (size_t)W + (size_t)x] = comp[0].plane[(size_t)y * (size_t)pitch + (size_t)x];
            return true;
        }
        std::vector<uint8_t> full[3];
        for (int c = 0; c < 3; ++c) {
            upsample(comp[c].plane.data(), comp[c].bw * 8, comp[c].dw, comp[c].dh, hmax / comp[c].h, vmax / comp[c].v, W, H, full[c]);
            std::vector<uint8_t>().swap(comp[c].plane);
        }
        // colour space (jdapimin.c default_decompress_parms): JFIF => YCbCr; Adobe transform 0 => RGB, 1 => YCbCr; neither:
        // component ids 1,2,3 => YCbCr, 'R','G','B' => RGB, anything else YCbCr
        bool ycc = true;
        if (!jfif && adobe) ycc = adobe_transform != 0;
        else if (!jfif && !adobe && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') ycc = false;
        bands = 3;
        px.resize((size_t)W * (size_t)H * 3);
        if (!ycc) {
            for (size_t i = 0; i < (size_t)W * (size_t)H; ++i) { px[3 * i] = full[0][i]; px[3 * i + 1] = full[1][i]; px[3 * i + 2] = full[2][i]; }
            return true;
        }
        // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert
        int cr_r[256], cb_b[256];
        long cr_g[256], cb_g[256];
        auto fix = [](double x) { return (long)(x * 65536.0 + 0.5); };
        for (int i = 0; i < 256; ++i) {
            const long x = i - 128;
            cr_r[i] = (int)((fix(1.40200) * x + 32768) >> 16);
            cb_b[i] = (int)((fix(1.77200) * x + 32768) >> 16);
            cr_g[i] = -fix(0.71414) * x;
            cb_g[i] = -fix(0.34414) * x + 32768;
        }
        auto clamp = [](int v) { return (uint16_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
        for (size_t i = 0; i < (size_t)W * (size_t)H; ++i) {
            const int y = full[0][i], cb = full[1][i], cr = full[2][i];
            px[3 * i] = clamp(y + cr_r[cr]);
            px[3 * i + 1] = clamp(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
            px[3 * i + 2] = clamp(y + cb_b[cb]);
        }
        return true;
    }
};

}  // namespace

// interleaved 8-bit samples: one band (greyscale file) or three (R, G, B)
bool decode_jpeg(const uint8_t* data, size_t n, int& w, int& h, int& bands, std::vector<uint16_t>& px, std::string& msg) {
    Decoder d{data, n, msg};
    return d.run(w, h, bands, px);
}

}  // namespace sift_hip
