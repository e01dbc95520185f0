// ssim_cli.cpp -- `rmgr-ssim`, the command-line front end.
//
// Same command line, output format and exit codes as the reference's tool (src/ssim-cli.cpp:74-83 help,
// :130-213 per-channel / luminance / single-channel modes and their printf formats, :216-389 argument
// handling and map export), on top of the C ABI of this repo:
//   all channels  -> rmgr_ssim_hip_compute_ssim_channels_host   (one staging copy, one launch)
//   -y luminance  -> rmgr_ssim_hip_compute_ssim_luminance_host  (BT.601 conversion on the GPU)
//   -0..-3        -> rmgr_ssim_compute_ssim                      (the plain drop-in call)
// The reference decodes images with stb_image, which it downloads at configure time and which is not
// available here; this tool carries its own small codecs instead: PNG (8/16-bit, non-interlaced), JPEG
// (baseline and progressive Huffman), binary/ASCII PNM, uncompressed BMP and TGA for input; PNG, PNM, BMP, TGA and PFM for the map.
#include <rmgr/ssim.h>
#include <rmgr/ssim-hip.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <strings.h>
#include <vector>

namespace {

typedef std::vector<unsigned char> Bytes;

struct Image {
    int width, height, channels;
    Bytes px;   // interleaved, top-down
    Image() : width(0), height(0), channels(0) {}
};

// Decoders refuse absurd headers before allocating for them (stb_image, which the reference uses, caps
// dimensions at 2^24 as well).
const size_t kMaxDimension = size_t(1) << 24;
const size_t kMaxImageBytes = size_t(1) << 32;
bool sane_size(size_t w, size_t h, size_t channels)
{
    return w > 0 && h > 0 && channels > 0 && w <= kMaxDimension && h <= kMaxDimension && w * h <= kMaxImageBytes / channels;
}

bool read_file(const char* path, Bytes& out)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return true;
}

// ------------------------------------------------------------------------------ inflate (RFC 1951)
struct BitReader {
    const unsigned char* p; size_t n, pos; unsigned bitbuf; int bitcnt; bool bad;
    BitReader(const unsigned char* d, size_t len) : p(d), n(len), pos(0), bitbuf(0), bitcnt(0), bad(false) {}
    unsigned bits(int need) {
        unsigned v = bitbuf;
        while (bitcnt < need) {
            if (pos >= n) { bad = true; return 0; }
            v |= unsigned(p[pos++]) << bitcnt;
            bitcnt += 8;
        }
        bitbuf = need < 32 ? v >> need : 0;
        bitcnt -= need;
        return need < 32 ? v & ((1u << need) - 1u) : v;
    }
};

struct Huffman { short count[16]; short symbol[288]; };

void build_huffman(Huffman& h, const short* lengths, int n)
{
    memset(h.count, 0, sizeof(h.count));
    for (int i = 0; i < n; ++i) h.count[lengths[i]]++;
    short offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + h.count[l];
    for (int i = 0; i < n; ++i)
        if (lengths[i]) h.symbol[offs[lengths[i]]++] = short(i);
}

int decode_symbol(BitReader& br, const Huffman& h)
{
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
        code |= int(br.bits(1));
        if (br.bad) return -1;
        const int count = h.count[len];
        if (code - count < first) return h.symbol[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
}

bool inflate_codes(BitReader& br, Bytes& out, const Huffman& lit, const Huffman& dist, size_t limit)
{
    static const short lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const short lext[29]  = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    static const short dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    static const short dext[30]  = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    for (;;) {
        int sym = decode_symbol(br, lit);
        if (sym < 0) return false;
        if (out.size() > limit) return false;                 // more data than the image can hold: corrupt or hostile
        if (sym < 256) { out.push_back((unsigned char)sym); continue; }
        if (sym == 256) return true;
        sym -= 257;
        if (sym >= 29) return false;
        const int len = lbase[sym] + int(br.bits(lext[sym]));
        const int ds = decode_symbol(br, dist);
        if (ds < 0 || ds >= 30) return false;
        const size_t d = size_t(dbase[ds]) + br.bits(dext[ds]);
        if (br.bad || d > out.size()) return false;
        for (int i = 0; i < len; ++i) out.push_back(out[out.size() - d]);
    }
}

bool inflate(const unsigned char* data, size_t n, Bytes& out, size_t limit)
{
    BitReader br(data, n);
    int last;
    do {
        last = int(br.bits(1));
        const int type = int(br.bits(2));
        if (br.bad) return false;
        if (type == 0) {
            br.bitbuf = 0; br.bitcnt = 0;
            if (br.pos + 4 > n) return false;
            const unsigned len = data[br.pos] | (data[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > n || out.size() + len > limit + 65536) return false;
            out.insert(out.end(), data + br.pos, data + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            short lengths[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lengths[i] = 8;
                for (; i < 256; ++i) lengths[i] = 9;
                for (; i < 280; ++i) lengths[i] = 7;
                for (; i < 288; ++i) lengths[i] = 8;
                build_huffman(lit, lengths, 288);
                for (i = 0; i < 30; ++i) lengths[i] = 5;
                build_huffman(dist, lengths, 30);
            } else {
                static const short order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                const int nlen = int(br.bits(5)) + 257, ndist = int(br.bits(5)) + 1, ncode = int(br.bits(4)) + 4;
                if (br.bad || nlen > 286 || ndist > 30) return false;
                short cl[19];
                memset(cl, 0, sizeof(cl));
                for (int i = 0; i < ncode; ++i) cl[order[i]] = short(br.bits(3));
                Huffman lencode;
                build_huffman(lencode, cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = decode_symbol(br, lencode);
                    if (sym < 0) return false;
                    if (sym < 16) { lengths[idx++] = short(sym); continue; }
                    int prev = 0, rep;
                    if (sym == 16) { if (idx == 0) return false; prev = lengths[idx - 1]; rep = 3 + int(br.bits(2)); }
                    else if (sym == 17) rep = 3 + int(br.bits(3));
                    else rep = 11 + int(br.bits(7));
                    if (idx + rep > nlen + ndist) return false;
                    while (rep--) lengths[idx++] = short(prev);
                }
                build_huffman(lit, lengths, nlen);
                build_huffman(dist, lengths + nlen, ndist);
            }
            if (!inflate_codes(br, out, lit, dist, limit)) return false;
        } else return false;
    } while (!last);
    return true;
}

// ------------------------------------------------------------------------------ PNG
unsigned be32(const unsigned char* p) { return (unsigned(p[0]) << 24) | (unsigned(p[1]) << 16) | (unsigned(p[2]) << 8) | p[3]; }

bool decode_png(const Bytes& f, Image& img, std::string& err)
{
    if (f.size() < 33) { err = "truncated PNG"; return false; }
    size_t pos = 8;
    unsigned w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    Bytes idat, plte;
    while (pos + 12 <= f.size()) {
        const unsigned len = be32(&f[pos]);
        const char* tag = reinterpret_cast<const char*>(&f[pos + 4]);
        if (pos + 12 + len > f.size()) { err = "truncated PNG chunk"; return false; }
        const unsigned char* d = &f[pos + 8];
        if (!memcmp(tag, "IHDR", 4)) {
            if (len < 13) { err = "corrupt PNG header"; return false; }
            w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
        }
        else if (!memcmp(tag, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(tag, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (interlace) { err = "interlaced PNG is not supported"; return false; }
    // bit depths the PNG specification allows per colour type; anything else (0, 3, 5, 6, 7, 32 ...) is refused
    // before any buffer is sized or any shift is formed
    const bool depth_ok = (ctype == 0) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : (ctype == 3) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                        : (ctype == 2 || ctype == 4 || ctype == 6) ? (depth == 8 || depth == 16) : false;
    if (!depth_ok) { err = "unsupported PNG colour type / bit depth"; return false; }
    const int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!samples || !sane_size(w, h, size_t(samples) * 2) || idat.size() < 6) { err = "unsupported or corrupt PNG"; return false; }
    const size_t bpp = std::max<size_t>(1, size_t(samples) * depth / 8), rowBytes = (size_t(w) * samples * depth + 7) / 8;
    Bytes raw;
    raw.reserve(std::min((rowBytes + 1) * size_t(h), idat.size() * 1032 + 64));   // deflate cannot expand more than ~1032x
    if (!inflate(&idat[2], idat.size() - 2, raw, (rowBytes + 1) * size_t(h))) { err = "corrupt PNG data stream"; return false; }
    if (raw.size() < (rowBytes + 1) * h) { err = "short PNG data stream"; return false; }
    Bytes prev(rowBytes, 0), cur(rowBytes);
    img.width = int(w); img.height = int(h); img.channels = (ctype == 3) ? 3 : samples;
    img.px.resize(size_t(w) * h * img.channels);
    for (unsigned y = 0; y < h; ++y) {
        const unsigned char* src = &raw[y * (rowBytes + 1)];
        const int ft = src[0];
        for (size_t i = 0; i < rowBytes; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) { err = "bad PNG filter"; return false; }
            cur[i] = (unsigned char)(src[1 + i] + pred);
        }
        unsigned char* dst = &img.px[size_t(y) * w * img.channels];
        for (unsigned x = 0; x < w; ++x) {
            for (int s = 0; s < samples; ++s) {
                unsigned v;
                if (depth == 8) v = cur[x * samples + s];
                else if (depth == 16) v = cur[(x * samples + s) * 2];          // most significant byte
                else { const unsigned bit = x * depth; v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u); if (ctype == 0) v = v * 255u / ((1u << depth) - 1u); }
                if (ctype == 3) {
                    if (3 * v + 2 >= plte.size()) { err = "bad PNG palette index"; return false; }
                    dst[x * 3] = plte[3 * v]; dst[x * 3 + 1] = plte[3 * v + 1]; dst[x * 3 + 2] = plte[3 * v + 2];
                } else dst[x * samples + s] = (unsigned char)v;
            }
        }
        prev.swap(cur);
    }
    return true;
}

// ------------------------------------------------------------------------------ PNM / BMP / TGA input
bool pnm_token(const Bytes& f, size_t& pos, int& v)
{
    for (;;) {
        while (pos < f.size() && isspace(f[pos])) ++pos;
        if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') ++pos; continue; }
        break;
    }
    if (pos >= f.size() || !isdigit(f[pos])) return false;
    v = 0;
    while (pos < f.size() && isdigit(f[pos])) {
        if (v > 100000000) return false;
        v = v * 10 + (f[pos++] - '0');
    }
    return true;
}

bool decode_pnm(const Bytes& f, Image& img, std::string& err)
{
    const int kind = f[1] - '0';
    if (kind != 2 && kind != 3 && kind != 5 && kind != 6) { err = "unsupported PNM type"; return false; }
    size_t pos = 2;
    int w, h, maxv;
    if (!pnm_token(f, pos, w) || !pnm_token(f, pos, h) || !pnm_token(f, pos, maxv) || maxv <= 0 || maxv > 255) { err = "bad PNM header (only maxval <= 255)"; return false; }
    img.width = w; img.height = h; img.channels = (kind == 3 || kind == 6) ? 3 : 1;
    if (!sane_size(size_t(w), size_t(h), size_t(img.channels))) { err = "bad PNM dimensions"; return false; }
    const size_t n = size_t(w) * h * img.channels;
    img.px.resize(n);
    if (kind >= 5) {
        ++pos;   // single whitespace after maxval
        if (pos + n > f.size()) { err = "truncated PNM"; return false; }
        memcpy(&img.px[0], &f[pos], n);
    } else {
        for (size_t i = 0; i < n; ++i) { int v; if (!pnm_token(f, pos, v)) { err = "truncated PNM"; return false; } img.px[i] = (unsigned char)v; }
    }
    if (maxv != 255) for (size_t i = 0; i < n; ++i) img.px[i] = (unsigned char)(img.px[i] * 255 / maxv);
    return true;
}

unsigned le16(const unsigned char* p) { return p[0] | (p[1] << 8); }
unsigned le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (unsigned(p[3]) << 24); }

bool decode_bmp(const Bytes& f, Image& img, std::string& err)
{
    if (f.size() < 54) { err = "truncated BMP"; return false; }
    const unsigned off = le32(&f[10]), hdr = le32(&f[14]);
    const int w = int(le32(&f[18])), hs = int(le32(&f[22])), bpp = int(le16(&f[28]));
    const unsigned comp = le32(&f[30]);
    if ((comp != 0 && !(comp == 3 && bpp == 32)) || !(bpp == 8 || bpp == 24 || bpp == 32) || w <= 0 || hs == 0) { err = "unsupported BMP (need uncompressed 8/24/32 bpp)"; return false; }
    if (hs == int(0x80000000u) || !sane_size(size_t(w), size_t(hs < 0 ? -hs : hs), 4)) { err = "bad BMP dimensions"; return false; }
    const int h = hs < 0 ? -hs : hs;
    const size_t stride = ((size_t(w) * bpp + 31) / 32) * 4;
    if (off + stride * h > f.size() || (bpp == 8 && size_t(14) + hdr + 1024 > f.size())) { err = "truncated BMP"; return false; }
    const unsigned char* pal = &f[14 + hdr];
    img.width = w; img.height = h; img.channels = bpp == 32 ? 4 : 3;
    img.px.resize(size_t(w) * h * img.channels);
    for (int y = 0; y < h; ++y) {
        const unsigned char* src = &f[off + stride * size_t(hs < 0 ? y : h - 1 - y)];
        unsigned char* dst = &img.px[size_t(y) * w * img.channels];
        for (int x = 0; x < w; ++x) {
            if (bpp == 8) { const unsigned char* e = pal + 4 * src[x]; dst[3 * x] = e[2]; dst[3 * x + 1] = e[1]; dst[3 * x + 2] = e[0]; }
            else { const int n = bpp / 8; dst[img.channels * x] = src[n * x + 2]; dst[img.channels * x + 1] = src[n * x + 1]; dst[img.channels * x + 2] = src[n * x]; if (n == 4) dst[4 * x + 3] = src[4 * x + 3]; }
        }
    }
    return true;
}

bool decode_tga(const Bytes& f, Image& img, std::string& err)
{
    if (f.size() < 18) { err = "truncated TGA"; return false; }
    const int idlen = f[0], type = f[2], w = int(le16(&f[12])), h = int(le16(&f[14])), bpp = f[16];
    const bool topDown = (f[17] & 0x20) != 0;
    if (!((type == 2 && (bpp == 24 || bpp == 32)) || (type == 3 && bpp == 8)) || f[1] != 0 || w <= 0 || h <= 0) { err = "unsupported TGA (need uncompressed true-colour or grey)"; return false; }
    const int n = bpp / 8;
    const size_t off = 18 + idlen;
    if (off + size_t(w) * h * n > f.size()) { err = "truncated TGA"; return false; }
    img.width = w; img.height = h; img.channels = n;
    img.px.resize(size_t(w) * h * n);
    for (int y = 0; y < h; ++y) {
        const unsigned char* src = &f[off + size_t(topDown ? y : h - 1 - y) * w * n];
        unsigned char* dst = &img.px[size_t(y) * w * n];
        for (int x = 0; x < w; ++x) {
            if (n == 1) dst[x] = src[x];
            else { dst[n * x] = src[n * x + 2]; dst[n * x + 1] = src[n * x + 1]; dst[n * x + 2] = src[n * x]; if (n == 4) dst[4 * x + 3] = src[4 * x + 3]; }
        }
    }
    return true;
}

// ------------------------------------------------------------------------------ JPEG (baseline + progressive)
// ITU T.81 Huffman-coded 8-bit JPEG: sequential (SOF0/SOF1) and progressive (SOF2) scans, restart intervals,
// 1 or 3 components, sampling factors 1 or 2.  Reconstruction follows the IJG conventions every mainstream
// decoder shares -- the 13-bit fixed-point "slow integer" inverse DCT, triangle-filter ("fancy") chroma
// upsampling for 2:1 factors, 16-bit fixed-point YCbCr -> RGB -- so that decoded pixels agree with libjpeg
// (checked against PIL in tests/test_cli.py).  Arithmetic coding, 12-bit, lossless and CMYK files are refused.
const unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct JpegHuff {
    bool defined;
    int maxcode[18], valptr[17], mincode[17];
    unsigned char vals[256];
    JpegHuff() : defined(false) {}
    bool build(const unsigned char* bits /* [1..16] */, const unsigned char* v, int n)
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            if (code > (1 << l)) return false;                 // over-subscribed
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        if (k != n || n > 256) return false;
        memcpy(vals, v, size_t(n));
        defined = true;
        return true;
    }
};

struct JpegBits {
    const unsigned char* p; size_t n, pos; unsigned acc; int cnt; bool hit_marker;
    JpegBits(const unsigned char* d, size_t len, size_t start) : p(d), n(len), pos(start), acc(0), cnt(0), hit_marker(false) {}
    void fill()
    {
        while (cnt <= 24) {
            unsigned b = 0;
            if (!hit_marker && pos < n) {
                b = p[pos];
                if (b == 0xFF) {
                    if (pos + 1 < n && p[pos + 1] == 0x00) pos += 2;            // stuffed zero
                    else { hit_marker = true; b = 0; }                             // a marker: feed zeros from here on
                } else ++pos;
            }
            acc |= b << (24 - cnt);
            cnt += 8;
        }
    }
    int bit() { if (cnt < 1) fill(); const int v = int(acc >> 31); acc <<= 1; --cnt; return v; }
    int bits(int k) { if (k == 0) return 0; if (cnt < k) fill(); const int v = int(acc >> (32 - k)); acc <<= k; cnt -= k; return v; }
    int decode(const JpegHuff& h)
    {
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = (code << 1) | bit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        return -1;
    }
    void reset() { acc = 0; cnt = 0; }
};

inline int jpeg_extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

struct JpegComp {
    int id, h, v, tq, td, ta;
    int bw, bh;            // blocks per row / column of the coefficient array (padded to whole MCUs)
    int cw, chh;           // blocks per row / column that a non-interleaved scan visits
    int pred;
    std::vector<short> coef;
    Bytes plane;           // bw*8 x bh*8 samples after the inverse DCT
};

// jidctint.c-style 8x8 inverse DCT (CONST_BITS 13, PASS1_BITS 2) on dequantised coefficients.
void jpeg_idct(const short* in, const unsigned short* q, unsigned char* out, size_t stride)
{
    typedef long long W;      // 64-bit intermediates: same values as the 32-bit original on real images, no overflow on corrupt ones
    W ws[64];
    for (int c = 0; c < 8; ++c) {
        const W i0 = W(in[c]) * q[c], i1 = W(in[8 + c]) * q[8 + c], i2 = W(in[16 + c]) * q[16 + c], i3 = W(in[24 + c]) * q[24 + c],
                i4 = W(in[32 + c]) * q[32 + c], i5 = W(in[40 + c]) * q[40 + c], i6 = W(in[48 + c]) * q[48 + c], i7 = W(in[56 + c]) * q[56 + c];
        if (!(i1 | i2 | i3 | i4 | i5 | i6 | i7)) {
            const W dc = i0 * 4;
            for (int r = 0; r < 8; ++r) ws[r * 8 + c] = dc;
            continue;
        }
        W z1 = (i2 + i6) * 4433, t2 = z1 + i6 * -15137, t3 = z1 + i2 * 6270;
        W t0 = (i0 + i4) * 8192, t1 = (i0 - i4) * 8192;
        const W t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = i7; t1 = i5; t2 = i3; t3 = i1;
        z1 = t0 + t3; W z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
        const W z5 = (z3 + z4) * 9633;
        t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
        z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
        z3 += z5; z4 += z5;
        t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        const W rnd = 1 << 10;
        ws[0 * 8 + c] = (t10 + t3 + rnd) >> 11; ws[7 * 8 + c] = (t10 - t3 + rnd) >> 11;
        ws[1 * 8 + c] = (t11 + t2 + rnd) >> 11; ws[6 * 8 + c] = (t11 - t2 + rnd) >> 11;
        ws[2 * 8 + c] = (t12 + t1 + rnd) >> 11; ws[5 * 8 + c] = (t12 - t1 + rnd) >> 11;
        ws[3 * 8 + c] = (t13 + t0 + rnd) >> 11; ws[4 * 8 + c] = (t13 - t0 + rnd) >> 11;
    }
    for (int r = 0; r < 8; ++r) {
        const W* w = ws + r * 8;
        unsigned char* o = out + size_t(r) * stride;
        W res[8];
        if (!(w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7])) {
            const W dc = (w[0] + 16) >> 5;
            for (int c = 0; c < 8; ++c) res[c] = dc;
        } else {
            W z1 = (w[2] + w[6]) * 4433, t2 = z1 + w[6] * -15137, t3 = z1 + w[2] * 6270;
            W t0 = (w[0] + w[4]) * 8192, t1 = (w[0] - w[4]) * 8192;
            const W t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            t0 = w[7]; t1 = w[5]; t2 = w[3]; t3 = w[1];
            z1 = t0 + t3; W z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
            const W z5 = (z3 + z4) * 9633;
            t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
            z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
            z3 += z5; z4 += z5;
            t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
            const W rnd = 1 << 17;
            res[0] = (t10 + t3 + rnd) >> 18; res[7] = (t10 - t3 + rnd) >> 18;
            res[1] = (t11 + t2 + rnd) >> 18; res[6] = (t11 - t2 + rnd) >> 18;
            res[2] = (t12 + t1 + rnd) >> 18; res[5] = (t12 - t1 + rnd) >> 18;
            res[3] = (t13 + t0 + rnd) >> 18; res[4] = (t13 - t0 + rnd) >> 18;
        }
        for (int c = 0; c < 8; ++c) { const W v = res[c] + 128; o[c] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }
    }
}

struct JpegDecoder {
    const Bytes& f;
    std::string& err;
    int width, height, ncomp, hmax, vmax, mcux, mcuy, restart;
    bool progressive, adobe_rgb;
    JpegComp comp[3];
    unsigned short qt[4][64];
    bool qt_set[4];
    JpegHuff dc[4], ac[4];
    JpegDecoder(const Bytes& file, std::string& e) : f(file), err(e), width(0), height(0), ncomp(0), hmax(1), vmax(1), mcux(0), mcuy(0), restart(0), progressive(false), adobe_rgb(false)
    { memset(qt_set, 0, sizeof(qt_set)); }

    bool fail(const char* m) { err = m; return false; }

    bool read_sof(const unsigned char* d, size_t len)
    {
        if (len < 6 || d[0] != 8) return fail("unsupported JPEG sample precision");
        height = (d[1] << 8) | d[2]; width = (d[3] << 8) | d[4]; ncomp = d[5];
        if ((ncomp != 1 && ncomp != 3) || len < size_t(6 + 3 * ncomp)) return fail("unsupported JPEG component count (need greyscale or YCbCr)");
        // the coefficient store is allocated from the header alone: cap it (2^27 pixels = 16K x 8K, 0.8 GB)
        if (!sane_size(size_t(width), size_t(height), 3) || size_t(width) * size_t(height) > (size_t(1) << 27)) return fail("bad or oversized JPEG dimensions");
        for (int i = 0; i < ncomp; ++i) {
            JpegComp& c = comp[i];
            c.id = d[6 + 3 * i]; c.h = d[7 + 3 * i] >> 4; c.v = d[7 + 3 * i] & 15; c.tq = d[8 + 3 * i];
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return fail("unsupported JPEG sampling factors");
            hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
        }
        if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }
        mcux = (width + 8 * hmax - 1) / (8 * hmax); mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            JpegComp& c = comp[i];
            c.bw = mcux * c.h; c.bh = mcuy * c.v;
            const int sw = (width * c.h + hmax - 1) / hmax, sh = (height * c.v + vmax - 1) / vmax;
            c.cw = (sw + 7) / 8; c.chh = (sh + 7) / 8;
            c.coef.assign(size_t(c.bw) * c.bh * 64, 0);
        }
        return true;
    }

    bool read_dht(const unsigned char* d, size_t len)
    {
        while (len >= 17) {
            const int tc = d[0] >> 4, th = d[0] & 15;
            unsigned char bits[17]; bits[0] = 0;
            int n = 0;
            for (int i = 1; i <= 16; ++i) { bits[i] = d[i]; n += d[i]; }
            if (tc > 1 || th > 3 || n > 256 || len < size_t(17 + n)) return fail("corrupt JPEG Huffman table");
            if (!(tc ? ac[th] : dc[th]).build(bits, d + 17, n)) return fail("corrupt JPEG Huffman table");
            d += 17 + n; len -= size_t(17 + n);
        }
        return len == 0 || fail("corrupt JPEG Huffman table");
    }

    bool read_dqt(const unsigned char* d, size_t len)
    {
        while (len >= 1) {
            const int pq = d[0] >> 4, tq = d[0] & 15;
            const size_t need = pq ? 129 : 65;
            if (pq > 1 || tq > 3 || len < need) return fail("corrupt JPEG quantisation table");
            for (int i = 0; i < 64; ++i) qt[tq][kZigzag[i]] = pq ? (unsigned short)((d[1 + 2 * i] << 8) | d[2 + 2 * i]) : d[1 + i];
            qt_set[tq] = true;
            d += need; len -= need;
        }
        return true;
    }

    // One entropy-coded segment.  Returns the position just behind it (at the next marker).
    bool read_scan(const unsigned char* d, size_t len, size_t data_pos, size_t& next)
    {
        if (len < 1) return fail("corrupt JPEG scan header");
        const int ns = d[0];
        if (ns < 1 || ns > ncomp || len < size_t(4 + 2 * ns)) return fail("corrupt JPEG scan header");
        JpegComp* sc[3];
        for (int i = 0; i < ns; ++i) {
            sc[i] = NULL;
            for (int k = 0; k < ncomp; ++k) if (comp[k].id == d[1 + 2 * i]) sc[i] = &comp[k];
            if (!sc[i]) return fail("JPEG scan names an unknown component");
            sc[i]->td = d[2 + 2 * i] >> 4; sc[i]->ta = d[2 + 2 * i] & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("corrupt JPEG scan header");
        }
        const int ss = d[1 + 2 * ns], se = d[2 + 2 * ns], ah = d[3 + 2 * ns] >> 4, al = d[3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13 || ah > 13) return fail("corrupt JPEG progressive scan");
        } else if (ss != 0 || se != 63 || ah != 0 || al != 0) return fail("corrupt JPEG scan header");
        for (int i = 0; i < ns; ++i) {
            if ((ss == 0 && ah == 0 && !dc[sc[i]->td].defined) || (se > 0 && !ac[sc[i]->ta].defined)) return fail("JPEG scan uses an undefined Huffman table");
            sc[i]->pred = 0;
        }

        JpegBits br(&f[0], f.size(), data_pos);
        const bool interleaved = ns > 1;
        const int nx = interleaved ? mcux : sc[0]->cw, ny = interleaved ? mcuy : sc[0]->chh;
        int eobrun = 0, todo = restart, rst = 0;
        for (int my = 0; my < ny; ++my) {
            for (int mx = 0; mx < nx; ++mx) {
                if (restart && todo == 0) {
                    // expect RSTn at the current byte position
                    br.reset();
                    size_t p = br.pos;
                    while (p + 1 < f.size() && !(f[p] == 0xFF && f[p + 1] >= 0xD0 && f[p + 1] <= 0xD7)) {
                        if (f[p] == 0xFF && f[p + 1] != 0 && f[p + 1] != 0xFF) return fail("JPEG restart marker missing");
                        ++p;
                    }
                    if (p + 1 >= f.size() || f[p + 1] != 0xD0 + rst) return fail("JPEG restart marker missing");
                    rst = (rst + 1) & 7;
                    br.pos = p + 2; br.hit_marker = false;
                    for (int i = 0; i < ns; ++i) sc[i]->pred = 0;
                    eobrun = 0; todo = restart;
                }
                for (int i = 0; i < ns; ++i) {
                    JpegComp& c = *sc[i];
                    const int bxn = interleaved ? c.h : 1, byn = interleaved ? c.v : 1;
                    for (int by = 0; by < byn; ++by) for (int bx = 0; bx < bxn; ++bx) {
                        const int gx = interleaved ? mx * c.h + bx : mx, gy = interleaved ? my * c.v + by : my;
                        short* blk = &c.coef[(size_t(gy) * c.bw + gx) * 64];
                        if (!decode_block(br, c, blk, ss, se, ah, al, eobrun)) return fail("corrupt JPEG entropy-coded data");
                    }
                }
                if (restart) --todo;
            }
        }
        // the next marker
        size_t p = br.pos;
        while (p + 1 < f.size() && !(f[p] == 0xFF && f[p + 1] != 0x00 && f[p + 1] != 0xFF && !(f[p + 1] >= 0xD0 && f[p + 1] <= 0xD7))) ++p;
        next = p;
        return true;
    }

    bool decode_block(JpegBits& br, JpegComp& c, short* blk, int ss, int se, int ah, int al, int& eobrun)
    {
        if (!progressive) {
            int t = br.decode(dc[c.td]);
            if (t < 0 || t > 11) return false;
            c.pred = int(short(c.pred + jpeg_extend(br.bits(t), t)));
            blk[0] = short(c.pred);
            for (int k = 1; k < 64; ++k) {
                const int rs = br.decode(ac[c.ta]);
                if (rs < 0) return false;
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) { if (r == 15) { k += 15; continue; } break; }
                k += r;
                if (k > 63) return false;
                blk[kZigzag[k]] = short(jpeg_extend(br.bits(s), s));
            }
            return true;
        }
        if (ss == 0) {
            if (ah == 0) {
                const int t = br.decode(dc[c.td]);
                if (t < 0 || t > 11) return false;
                c.pred = int(short(c.pred + jpeg_extend(br.bits(t), t)));
                blk[0] = short(c.pred * (1 << al));
            } else if (br.bit()) blk[0] = short(blk[0] | (1 << al));
            return true;
        }
        if (ah == 0) {                                      // AC first pass
            if (eobrun > 0) { --eobrun; return true; }
            for (int k = ss; k <= se; ++k) {
                const int rs = br.decode(ac[c.ta]);
                if (rs < 0) return false;
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.bits(r); break; }
                    k += 15;
                    continue;
                }
                k += r;
                if (k > 63) return false;
                blk[kZigzag[k]] = short(jpeg_extend(br.bits(s), s) * (1 << al));
            }
            return true;
        }
        // AC refinement pass
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; ++k) {
                const int rs = br.decode(ac[c.ta]);
                if (rs < 0) return false;
                int r = rs >> 4, s = rs & 15;
                if (s) { if (s != 1) return false; s = br.bit() ? p1 : m1; }
                else if (r != 15) { eobrun = 1 << r; if (r) eobrun += br.bits(r); break; }
                do {
                    short& co = blk[kZigzag[k]];
                    if (co != 0) { if (br.bit() && (co & p1) == 0) co = short(co + (co >= 0 ? p1 : m1)); }
                    else if (--r < 0) break;
                    ++k;
                } while (k <= se);
                if (s) { if (k > 63) return false; blk[kZigzag[k]] = short(s); }
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) {
                short& co = blk[kZigzag[k]];
                if (co != 0 && br.bit() && (co & p1) == 0) co = short(co + (co >= 0 ? p1 : m1));
            }
            --eobrun;
        }
        return true;
    }

    bool reconstruct(Image& img)
    {
        for (int i = 0; i < ncomp; ++i) {
            JpegComp& c = comp[i];
            if (!qt_set[c.tq]) return fail("JPEG component uses an undefined quantisation table");
            const size_t stride = size_t(c.bw) * 8;
            c.plane.assign(stride * c.bh * 8, 0);
            for (int by = 0; by < c.bh; ++by) for (int bx = 0; bx < c.bw; ++bx)
                jpeg_idct(&c.coef[(size_t(by) * c.bw + bx) * 64], qt[c.tq], &c.plane[(size_t(by) * 8) * stride + size_t(bx) * 8], stride);
            std::vector<short>().swap(c.coef);
        }
        img.width = width; img.height = height; img.channels = ncomp;
        img.px.resize(size_t(width) * height * ncomp);
        if (ncomp == 1) {
            for (int y = 0; y < height; ++y) memcpy(&img.px[size_t(y) * width], &comp[0].plane[size_t(y) * comp[0].bw * 8], size_t(width));
            return true;
        }
        // full-resolution planes of the chroma components
        Bytes up[3];
        for (int i = 0; i < 3; ++i) upsample(comp[i], up[i]);
        // YCbCr -> RGB, jdcolor.c fixed point (SCALEBITS 16)
        for (int y = 0; y < height; ++y) {
            const unsigned char* Y = &up[0][size_t(y) * width]; const unsigned char* Cb = &up[1][size_t(y) * width]; const unsigned char* Cr = &up[2][size_t(y) * width];
            unsigned char* o = &img.px[size_t(y) * width * 3];
            for (int x = 0; x < width; ++x, o += 3) {
                if (adobe_rgb) { o[0] = Y[x]; o[1] = Cb[x]; o[2] = Cr[x]; continue; }
                const int yy = Y[x], cb = Cb[x] - 128, cr = Cr[x] - 128;
                const int r = yy + ((91881 * cr + 32768) >> 16);
                const int g = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
                const int b = yy + ((116130 * cb + 32768) >> 16);
                o[0] = (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r);
                o[1] = (unsigned char)(g < 0 ? 0 : g > 255 ? 255 : g);
                o[2] = (unsigned char)(b < 0 ? 0 : b > 255 ? 255 : b);
            }
        }
        return true;
    }

    // Component plane -> width x height samples.  2:1 factors use libjpeg's triangle filters (jdsample.c
    // h2v1_fancy_upsample / h2v2_fancy_upsample / h1v2_fancy_upsample); rows and columns beyond the
    // component's real extent are never read (edges replicate).
    void upsample(const JpegComp& c, Bytes& out) const
    {
        const int fx = hmax / c.h, fy = vmax / c.v;
        const int sw = (width * c.h + hmax - 1) / hmax, sh = (height * c.v + vmax - 1) / vmax;
        const size_t stride = size_t(c.bw) * 8;
        out.resize(size_t(width) * height);
        if (fx == 1 && fy == 1) {
            for (int y = 0; y < height; ++y) memcpy(&out[size_t(y) * width], &c.plane[size_t(y) * stride], size_t(width));
            return;
        }
        std::vector<int> colsum(size_t(sw) + 1);
        Bytes row(size_t(sw) * 2 + 2);
        for (int y = 0; y < height; ++y) {
            const int sy = fy == 2 ? y >> 1 : y;
            const unsigned char* r0 = &c.plane[size_t(sy) * stride];
            // vertical step: 3/4 nearer + 1/4 further source row (fy == 2), else the row itself (scaled by 4)
            if (fy == 2) {
                int ny = (y & 1) ? sy + 1 : sy - 1;
                ny = ny < 0 ? 0 : (ny > sh - 1 ? sh - 1 : ny);
                const unsigned char* r1 = &c.plane[size_t(ny) * stride];
                for (int x = 0; x < sw; ++x) colsum[x] = 3 * r0[x] + r1[x];
            }
            unsigned char* o = &out[size_t(y) * width];
            if (fx == 2 && fy == 2) {
                // h2v2: (3*this + neighbour + 8) >> 4 and (+7), edges (4*this + 8) >> 4 and (+7)
                for (int x = 0; x < sw; ++x) {
                    const int t = colsum[x], l = x > 0 ? colsum[x - 1] : t, r = x < sw - 1 ? colsum[x + 1] : t;
                    const int a = (x > 0 ? 3 * t + l + 8 : 4 * t + 8) >> 4, b = (x < sw - 1 ? 3 * t + r + 7 : 4 * t + 7) >> 4;
                    if (2 * x < width) o[2 * x] = (unsigned char)a;
                    if (2 * x + 1 < width) o[2 * x + 1] = (unsigned char)b;
                }
            } else if (fx == 2) {
                // h2v1: (3*this + left + 1) >> 2 and (3*this + right + 2) >> 2, edges copy
                for (int x = 0; x < sw; ++x) {
                    const int t = r0[x];
                    const int a = x > 0 ? (3 * t + r0[x - 1] + 1) >> 2 : t, b = x < sw - 1 ? (3 * t + r0[x + 1] + 2) >> 2 : t;
                    if (2 * x < width) o[2 * x] = (unsigned char)a;
                    if (2 * x + 1 < width) o[2 * x + 1] = (unsigned char)b;
                }
            } else {
                // h1v2: (3*near + far + bias) >> 2 with bias 1 for the upper and 2 for the lower output row
                const int bias = (y & 1) ? 2 : 1;
                for (int x = 0; x < width; ++x) o[x] = (unsigned char)((colsum[x] + bias) >> 2);
            }
        }
    }

    bool run(Image& img)
    {
        if (f.size() < 4 || f[0] != 0xFF || f[1] != 0xD8) return fail("not a JPEG file");
        size_t pos = 2;
        bool have_sof = false, have_scan = false;
        while (pos + 4 <= f.size()) {
            if (f[pos] != 0xFF) { ++pos; continue; }
            const int m = f[pos + 1];
            if (m == 0xFF) { ++pos; continue; }
            if (m == 0xD9) break;
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0x00) { pos += 2; continue; }
            const size_t len = (size_t(f[pos + 2]) << 8) | f[pos + 3];
            if (len < 2 || pos + 2 + len > f.size()) return fail("truncated JPEG segment");
            const unsigned char* d = &f[pos + 4];
            const size_t n = len - 2;
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_sof) return fail("JPEG with several frames");
                progressive = (m == 0xC2);
                if (!read_sof(d, n)) return false;
                have_sof = true;
            } else if ((m >= 0xC3 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) return fail("unsupported JPEG coding process (lossless, hierarchical or arithmetic)");
            else if (m == 0xCC) return fail("arithmetic-coded JPEG is not supported");
            else if (m == 0xC4) { if (!read_dht(d, n)) return false; }
            else if (m == 0xDB) { if (!read_dqt(d, n)) return false; }
            else if (m == 0xDD) { if (n < 2) return fail("corrupt JPEG restart interval"); restart = (d[0] << 8) | d[1]; }
            else if (m == 0xEE) { if (n >= 12 && !memcmp(d, "Adobe", 5) && d[11] == 0) adobe_rgb = true; }
            else if (m == 0xDA) {
                if (!have_sof) return fail("JPEG scan before frame header");
                size_t next = 0;
                if (!read_scan(d, n, pos + 2 + len, next)) return false;
                have_scan = true;
                pos = next;
                continue;
            }
            pos += 2 + len;
        }
        if (!have_sof || !have_scan) return fail("JPEG without image data");
        return reconstruct(img);
    }
};

bool decode_jpeg(const Bytes& f, Image& img, std::string& err)
{
    JpegDecoder d(f, err);
    return d.run(img);
}

bool load_image(const char* path, Image& img)
{
    Bytes f;
    if (!read_file(path, f)) { fprintf(stderr, "Failed to open file \"%s\"\n", path); return false; }
    std::string err = "unknown image format (supported: PNG, JPEG, PNM, BMP, TGA)";
    bool ok = false;
    static const unsigned char pngSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    const char* ext = strrchr(path, '.');
    if (f.size() >= 8 && !memcmp(&f[0], pngSig, 8)) ok = decode_png(f, img, err);
    else if (f.size() >= 3 && f[0] == 'P' && f[1] >= '1' && f[1] <= '6') ok = decode_pnm(f, img, err);
    else if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') ok = decode_bmp(f, img, err);
    else if (f.size() >= 3 && f[0] == 0xFF && f[1] == 0xD8 && f[2] == 0xFF) ok = decode_jpeg(f, img, err);
    else if (ext && !strcasecmp(ext, ".tga")) ok = decode_tga(f, img, err);
    if (!ok) fprintf(stderr, "Failed to load image \"%s\":\n%s\n", path, err.c_str());
    return ok;
}

// ------------------------------------------------------------------------------ map output
unsigned crc32_of(const unsigned char* p, size_t n, unsigned crc)
{
    static unsigned table[256];
    if (!table[1]) for (unsigned i = 0; i < 256; ++i) { unsigned c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 255] ^ (crc >> 8);
    return ~crc;
}

void put_be32(Bytes& b, unsigned v) { b.push_back(v >> 24); b.push_back(v >> 16); b.push_back(v >> 8); b.push_back(v); }

void png_chunk(FILE* f, const char* tag, const Bytes& data)
{
    Bytes c;
    put_be32(c, unsigned(data.size()));
    c.insert(c.end(), tag, tag + 4);
    c.insert(c.end(), data.begin(), data.end());
    put_be32(c, crc32_of(&c[4], c.size() - 4, 0));
    fwrite(&c[0], 1, c.size(), f);
}

bool write_png(FILE* f, int w, int h, int ch, const unsigned char* px)
{
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    Bytes ihdr;
    put_be32(ihdr, unsigned(w)); put_be32(ihdr, unsigned(h));
    ihdr.push_back(8); ihdr.push_back(ch == 1 ? 0 : ch == 2 ? 4 : ch == 3 ? 2 : 6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(f, "IHDR", ihdr);
    Bytes raw;                      // filter byte 0 + row
    for (int y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), px + size_t(y) * w * ch, px + size_t(y + 1) * w * ch); }
    Bytes z;                        // zlib stream of stored blocks
    z.push_back(0x78); z.push_back(0x01);
    unsigned a = 1, b = 0;
    for (size_t i = 0; i < raw.size(); ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back(n & 255); z.push_back(n >> 8); z.push_back(~n & 255); z.push_back((~n >> 8) & 255);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (n == 0) break;
    }
    put_be32(z, (b << 16) | a);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", Bytes());
    return !ferror(f);
}

bool write_bmp(FILE* f, int w, int h, int ch, const unsigned char* px)
{
    const int stride = (w * 3 + 3) & ~3;
    unsigned char hdr[54];
    memset(hdr, 0, sizeof(hdr));
    hdr[0] = 'B'; hdr[1] = 'M';
    const unsigned size = 54 + unsigned(stride) * h;
    hdr[2] = size; hdr[3] = size >> 8; hdr[4] = size >> 16; hdr[5] = size >> 24;
    hdr[10] = 54; hdr[14] = 40;
    hdr[18] = w; hdr[19] = w >> 8; hdr[20] = w >> 16; hdr[21] = w >> 24;
    hdr[22] = h; hdr[23] = h >> 8; hdr[24] = h >> 16; hdr[25] = h >> 24;
    hdr[26] = 1; hdr[28] = 24;
    fwrite(hdr, 1, 54, f);
    Bytes row(stride, 0);
    for (int y = h - 1; y >= 0; --y) {
        for (int x = 0; x < w; ++x) {
            const unsigned char* s = px + (size_t(y) * w + x) * ch;
            row[3 * x + 2] = s[0]; row[3 * x + 1] = ch >= 3 ? s[1] : s[0]; row[3 * x] = ch >= 3 ? s[2] : s[0];
        }
        fwrite(&row[0], 1, stride, f);
    }
    return !ferror(f);
}

bool write_tga(FILE* f, int w, int h, int ch, const unsigned char* px)
{
    unsigned char hdr[18];
    memset(hdr, 0, sizeof(hdr));
    const int n = (ch == 1) ? 1 : (ch == 4 ? 4 : 3);
    hdr[2] = n == 1 ? 3 : 2; hdr[12] = w; hdr[13] = w >> 8; hdr[14] = h; hdr[15] = h >> 8; hdr[16] = 8 * n; hdr[17] = 0x20 | (n == 4 ? 8 : 0);
    fwrite(hdr, 1, 18, f);
    Bytes row(size_t(w) * n);
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            const unsigned char* s = px + (size_t(y) * w + x) * ch;
            if (n == 1) row[x] = s[0];
            else { row[n * x] = ch >= 3 ? s[2] : s[0]; row[n * x + 1] = ch >= 3 ? s[1] : s[0]; row[n * x + 2] = s[0]; if (n == 4) row[4 * x + 3] = s[3]; }
        }
        fwrite(&row[0], 1, row.size(), f);
    }
    return !ferror(f);
}

void print_help(FILE* file)
{
    fprintf(file, "Usage: rmgr-ssim [options] img1 img2 [map]\n"
                  "Options:\n"
                  "  -#  Compute SSIM only for channel #\n"
                  "  -y  Compute SSIM on luminance\n"
                  "      For images with <= 2 channels, only channel 0's SSIM will be computed\n"
                  "      For images with >= 3 channels, first three channels are converted from RGB to Y\n\n");
}

int report(rmgr_int32_t rc)
{
    if (rc == ENODEV) fprintf(stderr, "No usable gfx950 (MI355X) device: this build of rmgr-ssim computes on the GPU only\n");
    else fprintf(stderr, "SSIM computation failed: %s (errno %d)\n", strerror(rc), int(rc));
    return EXIT_FAILURE;
}

// ---- the command, as data ----
// What to compare: every channel (the default), one channel (-0 .. -3) or the BT.601 luminance (-y).  Behaviour and
// printed text are the reference tool's (src/ssim-cli.cpp:130-213, :216-389); the structure is this file's own.
struct Selection {
    enum Kind { EVERY_CHANNEL, ONE_CHANNEL, LUMA } kind;
    int channel;
};

struct Command {
    Selection   what;
    const char* file[2];
    const char* mapFile;     // NULL: no map requested
};

const struct { const char* flag; Selection sel; } kFlags[] = {
    {"-0", {Selection::ONE_CHANNEL, 0}}, {"-1", {Selection::ONE_CHANNEL, 1}}, {"-2", {Selection::ONE_CHANNEL, 2}},
    {"-3", {Selection::ONE_CHANNEL, 3}}, {"-y", {Selection::LUMA, 0}},
};

// argv -> Command.  false: usage error (the message has been printed).
bool parse_command(int argc, char* argv[], Command& cmd)
{
    if (argc < 3 || argc > 5) { print_help(stderr); return false; }
    cmd.what.kind = Selection::EVERY_CHANNEL;
    cmd.what.channel = 0;
    int first = 1;
    if (argc >= 4 && argv[1][0] == '-') {
        size_t k = 0;
        while (k < sizeof(kFlags) / sizeof(kFlags[0]) && strcmp(argv[1], kFlags[k].flag) != 0) ++k;
        if (k == sizeof(kFlags) / sizeof(kFlags[0])) { fprintf(stderr, "Unknown option: %s\n", argv[1]); return false; }
        cmd.what = kFlags[k].sel;
        first = 2;
    }
    cmd.file[0] = argv[first];
    cmd.file[1] = argv[first + 1];
    cmd.mapFile = (argc - first == 3) ? argv[first + 2] : NULL;
    return true;
}

// Map file formats by extension; `anyChannels` false = only 1 or 3 channels can be stored.
enum MapFormat { MAP_TGA, MAP_BMP, MAP_PNG, MAP_PFM, MAP_PNM };
const struct { const char* ext; MapFormat fmt; bool anyChannels; const char* label; } kMapFormats[] = {
    {".bmp", MAP_BMP, true, "BMP"}, {".png", MAP_PNG, true, "PNG"}, {".tga", MAP_TGA, true, "TGA"},
    {".pgm", MAP_PNM, false, "PNM"}, {".ppm", MAP_PNM, false, "PNM"}, {".pnm", MAP_PNM, false, "PNM"}, {".pfm", MAP_PFM, false, "PFM"},
};

// Runs the selected comparison and prints it in the reference's formats ("% 7.4f", "Channel %u: ...", "Average  : ...").
int compare(const Image& a, const Image& b, Selection what, float* map, int mapChannels)
{
    const rmgr_uint32_t w = rmgr_uint32_t(a.width), h = rmgr_uint32_t(a.height), ch = rmgr_uint32_t(a.channels);
    const ptrdiff_t pitch = ptrdiff_t(w) * ch;
    if (what.kind == Selection::LUMA && ch < 3)            // nothing to weigh: the first channel is the luminance
        what.kind = Selection::ONE_CHANNEL, what.channel = 0;
    std::vector<float> value(what.kind == Selection::EVERY_CHANNEL ? ch : 1u);
    rmgr_int32_t rc;
    switch (what.kind) {
    case Selection::ONE_CHANNEL: {
        rmgr_ssim_Params p;
        memset(&p, 0, sizeof(p));
        p.width = w; p.height = h;
        rmgr_ssim_init_interleaved(&p.imgA, &a.px[0], pitch, ch, rmgr_uint32_t(what.channel));
        rmgr_ssim_init_interleaved(&p.imgB, &b.px[0], pitch, ch, rmgr_uint32_t(what.channel));
        p.ssimMap = map; p.ssimStep = mapChannels; p.ssimStride = ptrdiff_t(w) * mapChannels;
        rc = rmgr_ssim_compute_ssim(&value[0], &p, NULL);
        break;
    }
    case Selection::LUMA:
        rc = rmgr_ssim_hip_compute_ssim_luminance_host(NULL, &value[0], &a.px[0], pitch, &b.px[0], pitch, w, h, ch, map);
        break;
    default:
        rc = rmgr_ssim_hip_compute_ssim_channels_host(NULL, &value[0], &a.px[0], pitch, &b.px[0], pitch, w, h, ch, map);
        break;
    }
    if (rc != 0) return report(rc);
    if (what.kind != Selection::EVERY_CHANNEL) {
        printf("% 7.4f\n", value[0]);
        return EXIT_SUCCESS;
    }
    float total = 0.0f;
    for (rmgr_uint32_t c = 0; c < ch; ++c) {
        printf("Channel %u: % 7.4f\n", unsigned(c), value[c]);
        total += value[c];
    }
    printf("Average  : % 7.4f\n", total / float(ch));
    return EXIT_SUCCESS;
}

// Stores the map: PFM keeps the floats (bottom row first, little endian, src/ssim-cli.cpp:355-374); every other
// format stores max(0, v) * 255 truncated to 8 bits (:341-342).
int save_map(const char* path, const std::vector<float>& map, int w, int h, int channels)
{
    const char* ext = strrchr(path, '.');
    MapFormat fmt = MAP_TGA;
    if (!ext) {
        fprintf(stderr, "Cannot deduce file format from extension, saving as tga\n");
    } else {
        size_t k = 0;
        while (k < sizeof(kMapFormats) / sizeof(kMapFormats[0]) && strcasecmp(ext, kMapFormats[k].ext) != 0) ++k;
        if (k == sizeof(kMapFormats) / sizeof(kMapFormats[0])) return EXIT_FAILURE;         // unknown extension
        fmt = kMapFormats[k].fmt;
        if (!kMapFormats[k].anyChannels && channels != 1 && channels != 3) {
            fprintf(stderr, "%s images can only contain 1 or 3 channels but the map contains %d channels\n", kMapFormats[k].label, channels);
            return EXIT_FAILURE;
        }
    }
    FILE* f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "Failed to open file \"%s\" for writing\n", path); return EXIT_FAILURE; }
    bool ok = true;
    if (fmt == MAP_PFM) {
        fprintf(f, "P%c\n%d %d\n-1.0\n", channels == 1 ? 'f' : 'F', w, h);
        const size_t row = size_t(w) * channels;
        for (int y = h; --y >= 0;) ok = ok && fwrite(&map[y * row], sizeof(float), row, f) == row;
    } else {
        Bytes q(map.size());
        for (size_t i = 0; i < map.size(); ++i) q[i] = (unsigned char)(std::max(0.0f, map[i]) * 255.0f);
        switch (fmt) {
        case MAP_PNG: ok = write_png(f, w, h, channels, &q[0]); break;
        case MAP_BMP: ok = write_bmp(f, w, h, channels, &q[0]); break;
        case MAP_TGA: ok = write_tga(f, w, h, channels, &q[0]); break;
        default:
            fprintf(f, "P%c\n%d %d\n255\n", channels == 1 ? '5' : '6', w, h);
            ok = fwrite(&q[0], 1, q.size(), f) == q.size();
        }
    }
    fclose(f);
    if (!ok) { fprintf(stderr, "Error writing to file \"%s\"\n", path); return EXIT_FAILURE; }
    return EXIT_SUCCESS;
}

// test hook (not in the reference): decode an image with the built-in codecs, dump the raw pixels
int decode_to_file(const char* in, const char* out)
{
    Image img;
    if (!load_image(in, img)) return EXIT_FAILURE;
    FILE* f = fopen(out, "wb");
    if (!f || fwrite(&img.px[0], 1, img.px.size(), f) != img.px.size()) return EXIT_FAILURE;
    fclose(f);
    printf("%d %d %d\n", img.width, img.height, img.channels);
    return EXIT_SUCCESS;
}

} // namespace

static int run(int argc, char* argv[])
{
    if (argc == 2 && (!strcmp(argv[1], "-h") || !strcmp(argv[1], "--help"))) { print_help(stdout); return EXIT_SUCCESS; }
    if (argc == 4 && !strcmp(argv[1], "--decode")) return decode_to_file(argv[2], argv[3]);
    Command cmd;
    if (!parse_command(argc, argv, cmd)) return EXIT_FAILURE;

    Image img[2];
    for (int i = 0; i < 2; ++i)
        if (!load_image(cmd.file[i], img[i])) return EXIT_FAILURE;
    if (img[0].width != img[1].width || img[0].height != img[1].height) {
        fprintf(stderr, "Images do not have the same dimensions: %ux%u vs %ux%u\n", img[0].width, img[0].height, img[1].width, img[1].height);
        return EXIT_FAILURE;
    }
    if (img[0].channels != img[1].channels) {
        fprintf(stderr, "Images do not have the same number of channels: %u vs %u\n", img[0].channels, img[1].channels);
        return EXIT_FAILURE;
    }
    if (cmd.what.kind == Selection::ONE_CHANNEL && cmd.what.channel >= img[0].channels) {
        fprintf(stderr, "Cannot compute SSIM for channel %u, images have only %u channels\n", cmd.what.channel, img[0].channels);
        return EXIT_FAILURE;
    }

    std::vector<float> map;
    const int mapChannels = !cmd.mapFile ? 0 : (cmd.what.kind == Selection::EVERY_CHANNEL ? img[0].channels : 1);
    if (cmd.mapFile) map.resize(size_t(img[0].width) * img[0].height * mapChannels);
    const int rc = compare(img[0], img[1], cmd.what, cmd.mapFile ? &map[0] : NULL, mapChannels);
    if (rc != EXIT_SUCCESS || !cmd.mapFile) return rc;
    return save_map(cmd.mapFile, map, img[0].width, img[0].height, mapChannels);
}

int main(int argc, char* argv[])
{
    try { return run(argc, argv); }
    catch (const std::exception& e) { fprintf(stderr, "rmgr-ssim: %s\n", e.what()); }   // bad_alloc / length_error on hostile headers
    catch (...) { fprintf(stderr, "rmgr-ssim: unexpected failure\n"); }
    return EXIT_FAILURE;
}
